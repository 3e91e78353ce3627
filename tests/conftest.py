import os
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))

SCENES = os.path.join(ROOT, "scenes")


def pytest_configure(config):
    config.addinivalue_line("markers", "gpu: needs a real MI355X (run with -m gpu on the GPU box)")


def scene_path(name):
    return os.path.join(SCENES, name + ".txt")


@pytest.fixture(scope="session")
def pyscenes():
    import pyscene
    cache = {}

    def get(name):
        if name not in cache:
            cache[name] = pyscene.parse_file(scene_path(name))
        return cache[name]
    return get


@pytest.fixture(scope="session")
def oracle_scenes(pyscenes):
    """Oracle scenes built with the true scene bounds (bounds_mode 0)."""
    import oracle_lib
    cache = {}

    def get(name, bounds_mode=0):
        key = (name, bounds_mode)
        if key not in cache:
            cache[key] = oracle_lib.OracleScene(pyscenes(name), bounds_mode=bounds_mode)
        return cache[key]
    return get


@pytest.fixture(scope="session")
def gpu_scenes():
    """Device scenes (parsed by the product's C++ parser, uploaded, LBVH built)."""
    import cuda_ray_tracer_amd as m
    cache = {}

    def get(name):
        if name not in cache:
            stl = m.parseInput(scene_path(name))
            raw = m.initRawConfigFromStl(stl, 0)
            m.build_lbvh_karas(raw)
            cache[name] = (stl, raw)
        return cache[name]
    yield get
    for stl, raw in cache.values():
        raw.close()
