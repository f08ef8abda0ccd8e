#!/bin/bash
# Runs on the GPU box (via gpurun): rocprofv3 kernel-trace/stats plus separate PMC passes of the
# bench command. Summaries land in gpurun_out/prof_<tag>/ ; copy what is to be judged to profiles/.
set -o pipefail
TAG=${1:-r01}
shift || true
EXTRA="$@"  # e.g. --config 4
REPO=${GRAFT_REPO_ROOT:-$(pwd)}
OUT=$REPO/gpurun_out/prof_$TAG
mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp
BENCH="python3 $REPO/bench.py --steps 50 --warmup 5 --no-cpu-baseline $EXTRA"
rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/trace -- $BENCH > $OUT/trace.log 2>&1 || exit 1
for pass in "SQ_WAVES SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_SMEM SQ_WAVE_CYCLES SQ_BUSY_CYCLES GRBM_GUI_ACTIVE" \
            "SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_WAIT_ANY SQ_ACTIVE_INST_VALU SQ_INST_CYCLES_SALU SQ_IFETCH SQ_ACTIVE_INST_SCA" \
            "FETCH_SIZE" "WRITE_SIZE"; do
  name=$(echo $pass | cut -d' ' -f1)
  rocprofv3 --pmc $pass --kernel-trace --output-format csv -d $OUT/pmc_$name -- $BENCH > $OUT/pmc_$name.log 2>&1 || echo "pass $name failed" >> $OUT/errors.log
done
rocprofv3 -L 2>/dev/null | grep -iE "icache|SQC_|IFETCH|INST_CYCLES|LEVEL_WAVES" | head -60 > $OUT/counter_list.txt
find $OUT -name "*.csv" | head -50 > $OUT/files.txt
echo done
