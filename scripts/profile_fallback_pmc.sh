#!/bin/bash
# Instruction-fetch counters of the generic kernel (scripts/bench_fallback.py <B>), separate --pmc passes.
set -o pipefail
B=${1:-4096}
REPO=${GRAFT_REPO_ROOT:-$(pwd)}
OUT=$REPO/gpurun_out/prof_fallback_$B
mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp
CMD="python3 $REPO/scripts/bench_fallback.py $B"
for pass in "SQ_IFETCH SQ_IFETCH_LEVEL SQ_WAVE_CYCLES SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_WAVES SQ_BUSY_CYCLES" \
            "SQC_ICACHE_REQ SQC_ICACHE_HITS SQC_ICACHE_MISSES SQC_ICACHE_MISSES_DUPLICATE SQC_TC_INST_REQ SQC_TC_STALL" \
            "SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_SMEM SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR SQ_INSTS_LDS SQ_WAIT_ANY"; do
  name=$(echo $pass | cut -d' ' -f1)
  rocprofv3 --pmc $pass --kernel-trace --output-format csv -d $OUT/pmc_$name -- $CMD > $OUT/pmc_$name.log 2>&1 || echo "pass $name failed" >> $OUT/errors.log
done
python3 - <<PY
import csv, glob, collections
for d in sorted(glob.glob("$OUT/pmc_*/*/*_counter_collection.csv")):
    acc = collections.defaultdict(list)
    for r in csv.DictReader(open(d)):
        if "tick_kernel" in r['Kernel_Name']:
            acc[r['Counter_Name']].append(float(r['Counter_Value']))
    for k, v in acc.items():
        # four cases x 35 launches each, in order
        n = len(v) // 4
        print(k, [round(sum(v[i*n:(i+1)*n])/n) for i in range(4)])
PY
