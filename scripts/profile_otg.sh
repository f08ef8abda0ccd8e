#!/bin/bash
# rocprofv3 of the all-moving phase of scripts/bench_otg.py: kernel stats, then HBM traffic counters in
# separate passes (FETCH_SIZE, WRITE_SIZE), per /opt/skills/guides/MI355X_MICROARCH.md
set -o pipefail
REPO=${GRAFT_REPO_ROOT:-$(pwd)}
OUT=$REPO/gpurun_out/prof_otg_moving
mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp
CMD="python3 $REPO/scripts/bench_otg.py 65536 moving"
rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/trace -- $CMD > $OUT/trace.log 2>&1 || exit 1
for c in FETCH_SIZE WRITE_SIZE; do
  rocprofv3 --pmc $c --kernel-trace --output-format csv -d $OUT/pmc_$c -- $CMD > $OUT/pmc_$c.log 2>&1 || echo "pass $c failed" >> $OUT/errors.log
done
python3 - <<PY
import csv, glob, collections
for d in sorted(glob.glob("$OUT/pmc_*/*/*_counter_collection.csv")):
    acc = collections.defaultdict(list)
    for r in csv.DictReader(open(d)):
        acc[(r['Kernel_Name'][:40], r['Counter_Name'])].append(float(r['Counter_Value']))
    for k, v in acc.items():
        v = sorted(v)
        print(k, 'n', len(v), 'median', v[len(v)//2], 'max', v[-1])
PY
