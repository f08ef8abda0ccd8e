#!/bin/bash
# Counters of the SVD-free kernel for general hierarchies (tick_cert_kernel) on the C4 hierarchy (scripts/bench_hierarchy.py: regular poses,
# 1 in 1000 near-singular, the C4 workload), separate --pmc passes. Usage: profile_group_pmc.sh <tag> [lanes]
set -o pipefail
TAG=${1:-c4} LANES=${2:-8}
REPO=${GRAFT_REPO_ROOT:-$(pwd)}
OUT=$REPO/gpurun_out/prof_cert_$TAG
mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp
export SAI2B_GENERIC_LANES=$LANES
CMD="python3 $REPO/scripts/bench_hierarchy.py"
rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/trace -- $CMD > $OUT/trace.log 2>&1 || echo "trace failed" >> $OUT/errors.log
for pass in "SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR SQ_WAVES" \
            "SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_VALU" \
            "FETCH_SIZE" "WRITE_SIZE"; do
  name=$(echo $pass | cut -d' ' -f1)
  rocprofv3 --pmc $pass --kernel-trace --output-format csv -d $OUT/pmc_$name -- $CMD > $OUT/pmc_$name.log 2>&1 || echo "pass $name failed" >> $OUT/errors.log
done
python3 - <<PY
import csv, glob, collections
for d in sorted(glob.glob("$OUT/pmc_*/*/*_counter_collection.csv")):
    acc = collections.defaultdict(list)
    for r in csv.DictReader(open(d)):
        if "tick_cert_kernel" in r["Kernel_Name"]:
            acc[r['Counter_Name']].append(float(r['Counter_Value']))
    for k, v in acc.items():
        n = len(v) // 3  # three cases x 35 launches each, in order
        print(k, [round(sum(v[i*n:(i+1)*n])/n) for i in range(3)])
PY
