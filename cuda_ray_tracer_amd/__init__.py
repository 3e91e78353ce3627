"""cuda_ray_tracer_amd -- MI355X-native LBVH ray tracer (hot path of GJ0407790/cuda_ray_tracer).

Host side: api.py (ctypes over include/mirt.h).  Device side: csrc/*.hip, built by build.py into _build/libmirt.so.
"""
from .api import (MirtError, StlConfig, RawConfig, parseInput, parseText, syntheticScene, initRawConfigFromStl,
                  copyConfigDataToDevice, freeRawConfigDeviceMemory, build_lbvh_karas, render, render_params,
                  num_pixels, scatter_part, write_png, lib, render_accumulate, finalize)

__all__ = ["MirtError", "StlConfig", "RawConfig", "parseInput", "parseText", "syntheticScene", "initRawConfigFromStl",
           "copyConfigDataToDevice", "freeRawConfigDeviceMemory", "build_lbvh_karas", "render", "render_params",
           "num_pixels", "scatter_part", "write_png", "lib", "render_accumulate", "finalize"]
