#!/bin/bash
# usage: ab_run.sh "scene ..." lib1 lib2 ...  -- tools/perf3.py per library variant on the same box ("-" = the default build)
R=$GRAFT_REPO_ROOT; cd $R
SCENES=$1; shift
for lib in "$@"; do
  if [ "$lib" = "-" ]; then unset MIRT_LIB; echo "== default"; else export MIRT_LIB=$R/$lib; echo "== $lib"; fi
  python tools/perf3.py $SCENES 2>&1 | grep -v Warning
done
