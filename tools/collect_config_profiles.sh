#!/bin/bash
# Profiles of the path-tracing kernel on one of the larger configurations (a hierarchy scene), on a GPU box:
#   tools/collect_config_profiles.sh <tag> scifi|bubbletea [size]
# 1. rocprofv3 --kernel-trace --stats over 50 passes (two 25-pass launches) -> gpurun_out/<tag>/kernel_stats.csv
# 2. hardware counters of one 25-pass launch, one rocprofv3 run per counter set -> gpurun_out/<tag>/pmc.json
# Copy kernel_stats.csv, run.log and pmc.json from there to profiles/<tag>_* afterwards.
set -e
TAG=${1:?tag}
SCENE=${2:?scene}
SIZE=${3:-1024}
ROOT=$(cd "$(dirname "$0")/.." && pwd)
OUT=$ROOT/gpurun_out/$TAG
mkdir -p "$OUT"
cd /tmp && export TMPDIR=/tmp
timeout -k 10 300 rocprofv3 --kernel-trace --stats --output-format csv -d "$OUT/trace" -- \
    python3 "$ROOT/tools/run_config.py" "$SCENE" "$SIZE" 50 > "$OUT/run.log" 2>&1
STATS=$(find "$OUT/trace" -name '*kernel_stats.csv' | head -1)
cp "$STATS" "$OUT/kernel_stats.csv"
i=0
for C in "SQ_WAVES SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_SMEM SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR SQ_WAVE_CYCLES SQ_BUSY_CYCLES" \
         "SQ_ACTIVE_INST_VALU SQ_THREAD_CYCLES_VALU SQ_ACTIVE_INST_ANY SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_SCA SQ_INSTS_LDS SQ_INSTS_VALU_FMA_F64" \
         "FETCH_SIZE" "WRITE_SIZE" "TCC_HIT_sum TCC_MISS_sum TCC_REQ_sum"; do
  i=$((i+1))
  timeout -s KILL 180 rocprofv3 --pmc $C --output-format csv -d "$OUT/p$i" -- \
      python3 "$ROOT/tools/run_config.py" "$SCENE" "$SIZE" 25 > "$OUT/p$i.log" 2>&1 || { echo "pass $i failed"; tail -5 "$OUT/p$i.log"; exit 1; }
done
python3 "$ROOT/tools/pmc_summary.py" "$OUT" "$SCENE $SIZE" > "$OUT/pmc.json"
cat "$OUT/run.log" | tail -2
cat "$OUT/pmc.json"
