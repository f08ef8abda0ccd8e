#!/bin/bash
# A/B of compile-time variants of sai2b_cert.hip (7 joints) on one box.
#   build (here, before gpurun):  scripts/micro/cert_variants.sh build name1:-DFLAG1 name2:"-DFLAG2 -DFLAG3" ...
#   run (on the GPU box):         scripts/micro/cert_variants.sh run [bench.py arguments]
set -e
cd "$(dirname "$0")/../../sai2-primitives-perso_amd/csrc"
mode=$1; shift
if [ "$mode" = build ]; then
  rm -rf build_var && mkdir -p build_var
  objs=$(ls *.o | grep -v sai2b_cert_n7.o | tr '\n' ' ')
  for spec in "$@"; do
    name=${spec%%:*}; flags=${spec#*:}; [ "$flags" = "$spec" ] && flags=""
    ( /opt/rocm/bin/hipcc --offload-arch=gfx950 -O3 -std=c++17 -fPIC -DSAI2B_N=7 -include sai2b_dof_rename.h $flags -c sai2b_cert.hip -o build_var/cert_$name.o \
      && /opt/rocm/bin/hipcc --offload-arch=gfx950 -shared -o build_var/libsai2b_$name.so $objs build_var/cert_$name.o && rm build_var/cert_$name.o ) &
  done
  wait
  ls build_var
else
  cd ../..
  for lib in sai2-primitives-perso_amd/csrc/build_var/libsai2b_*.so; do
    for rep in 1 2; do
      SAI2B_LIB=$PWD/$lib python bench.py --no-cpu-baseline "$@" 2>/dev/null | python -c "import json,sys; d=json.loads(sys.stdin.read()); r=d['roofline']; print('$(basename $lib)', round(d['value']/1e6,1), 'M ticks/s  step', round(d['ms_per_step']*1e3,2), 'us  kernel', round(r['kernel_ms']*1e3,2), 'us')"
    done
  done
fi
