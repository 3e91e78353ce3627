#!/bin/bash
# usage: ab_run.sh "scene ..." lib1 lib2 ...  -- tools/perf3.py per library variant on the same box ("-" = the default build), twice (A B A B)
R=$GRAFT_REPO_ROOT; cd $R
SCENES=$1; shift
for rep in 1 2; do
for lib in "$@"; do
  if [ "$lib" = "-" ]; then unset MIRT_LIB; echo "== default"; else export MIRT_LIB=$R/cuda_ray_tracer_amd/_build/ab/$lib/libmirt.so; echo "== $lib"; fi
  python tools/perf3.py $SCENES 2>&1 | grep -v Warning | grep -v amdgpu.ids | sed 's/\[.*\]//'
done; done
