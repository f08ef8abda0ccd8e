#!/bin/bash
# times every experiment build libsai2b_v<n>.so with bench.py (and checks it against the oracle)
REPO=${GRAFT_REPO_ROOT:-$(pwd)}
for lib in $REPO/sai2-primitives-perso_amd/csrc/libsai2b_v*.so; do
  v=$(basename $lib .so)
  SAI2B_LIB=$lib timeout -k 10 120 python $REPO/__graft_entry__.py smoke 2>&1 | grep "smoke:" | sed "s/^/$v /"
  SAI2B_LIB=$lib timeout -k 10 200 python $REPO/bench.py --steps 300 --warmup 30 --no-cpu-baseline > $REPO/gpurun_out/bench_$v.log 2>&1
  python3 -c "
import json
d=json.loads(open('$REPO/gpurun_out/bench_$v.log').read().strip().splitlines()[-1]); print('$v', round(d['value']/1e6,1), 'Mticks/s  step_ms', round(d['ms_per_step'],4))"
done
