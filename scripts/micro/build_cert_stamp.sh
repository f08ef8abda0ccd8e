#!/bin/bash
# builds csrc/build_stamp/libsai2b_stamp.so: the product library with sai2b_cert.hip (7 joints) compiled with
# -DSAI2B_CERT_STAMP, for scripts/micro/cert_stamps.py (SAI2B_LIB=<that file>). Run after `make` in csrc/.
set -e
cd "$(dirname "$0")/../../sai2-primitives-perso_amd/csrc"
mkdir -p build_stamp
/opt/rocm/bin/hipcc --offload-arch=gfx950 -O3 -std=c++17 -fPIC -DSAI2B_N=7 -include sai2b_dof_rename.h -DSAI2B_CERT_STAMP -c sai2b_cert.hip -o build_stamp/sai2b_cert_n7.o
objs=$(ls *.o | grep -v sai2b_cert_n7.o | tr '\n' ' ')
/opt/rocm/bin/hipcc --offload-arch=gfx950 -shared -o build_stamp/libsai2b_stamp.so $objs build_stamp/sai2b_cert_n7.o
