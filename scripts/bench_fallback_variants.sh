#!/bin/bash
# generic-kernel latency (scripts/bench_fallback.py) for every experiment build libsai2b_v<name>.so
REPO=${GRAFT_REPO_ROOT:-$(pwd)}
echo "baseline"; timeout -k 10 200 python $REPO/scripts/bench_fallback.py | tail -2
for lib in $REPO/sai2-primitives-perso_amd/csrc/libsai2b_v*.so; do
  echo $(basename $lib .so); SAI2B_LIB=$lib timeout -k 10 200 python $REPO/scripts/bench_fallback.py | tail -2
done
