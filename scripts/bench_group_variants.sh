#!/bin/bash
# times the generic (lanes-per-robot) kernel built for 1..4 wavefronts per SIMD, 16 and 8 lanes per robot
cd "$(dirname "$0")/.."
for w in ${WAVES:-1 2}; do
  for g in 16 8; do
    echo "== waves/SIMD=$w lanes=$g"
    SAI2B_LIB=$PWD/sai2-primitives-perso_amd/csrc/libsai2b_w$w.so SAI2B_GENERIC_LANES=$g python scripts/bench_hierarchy.py 2>&1 | grep -v amdgpu.ids
    SAI2B_LIB=$PWD/sai2-primitives-perso_amd/csrc/libsai2b_w$w.so SAI2B_GENERIC_LANES=$g python scripts/bench_fallback.py 2>&1 | grep -v amdgpu.ids
  done
done
