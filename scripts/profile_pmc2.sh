#!/bin/bash
# extra PMC passes (instruction/scalar cache, vector memory) for the bench command
set -o pipefail
TAG=${1:-r01b}
REPO=${GRAFT_REPO_ROOT:-$(pwd)}
OUT=$REPO/gpurun_out/prof_$TAG
mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp
BENCH="python3 $REPO/bench.py --steps 30 --warmup 5 --no-cpu-baseline"
i=0
for pass in "SQC_ICACHE_REQ SQC_ICACHE_HITS SQC_ICACHE_MISSES SQC_ICACHE_MISSES_DUPLICATE SQC_TC_INST_REQ SQ_IFETCH SQ_WAVE_CYCLES" \
            "SQC_DCACHE_REQ SQC_DCACHE_HITS SQC_DCACHE_MISSES SQC_TC_DATA_READ_REQ SQC_TC_STALL SQ_WAIT_ANY SQ_INSTS_SMEM" \
            "SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR SQ_INSTS_FLAT SQ_INSTS_LDS SQ_WAIT_INST_ANY SQ_ACTIVE_INST_VMEM SQ_INST_CYCLES_VMEM" \
            "SQ_IFETCH_LEVEL SQ_IFETCH SQ_BUSY_CYCLES SQ_WAVES SQ_INSTS_VALU SQ_ACTIVE_INST_VALU"; do
  i=$((i+1))
  rocprofv3 --pmc $pass --kernel-trace --output-format csv -d $OUT/pmc2_$i -- $BENCH > $OUT/pmc2_$i.log 2>&1 || echo "pass $i failed: $pass" >> $OUT/errors.log
done
python3 - <<PY
import csv, glob, collections
for d in sorted(glob.glob("$OUT/pmc2_*/*/*_counter_collection.csv")):
    acc = collections.defaultdict(list)
    for r in csv.DictReader(open(d)):
        if 'tick_kernel' in r['Kernel_Name']:
            acc[r['Counter_Name']].append(float(r['Counter_Value']))
    print(d.split('/')[-3], {k: round(sum(v)/len(v)) for k, v in acc.items()})
PY
cat $OUT/errors.log 2>/dev/null
