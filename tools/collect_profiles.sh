#!/bin/bash
# Collects the profiles that back bench.py's roofline figures, on a GPU box:
#   tools/collect_profiles.sh <tag>
# 1. rocprofv3 --kernel-trace --stats over the default bench.py run -> profiles/<tag>_kernel_stats.csv
# 2. hardware counters of one render launch of the same workload, one rocprofv3 run per counter set
#    (FETCH_SIZE and WRITE_SIZE do not fit one pass) -> profiles/<tag>_pmc.json via tools/pmc_summary.py
# Everything is written under gpurun_out/<tag>/ (the only directory that comes back from a GPU box); copy
# kernel_stats.csv, bench_traced.json and pmc.json from there to profiles/<tag>_* afterwards.
set -e
TAG=${1:?tag}
ROOT=$(cd "$(dirname "$0")/.." && pwd)
OUT=$ROOT/gpurun_out/$TAG
mkdir -p "$OUT"
cd /tmp && export TMPDIR=/tmp
timeout -k 10 600 rocprofv3 --kernel-trace --stats --output-format csv -d "$OUT/trace" -- \
    python3 "$ROOT/bench.py" --no-cpu-baseline > "$OUT/bench_traced.log" 2>&1
STATS=$(find "$OUT/trace" -name '*kernel_stats.csv' | head -1)
cp "$STATS" "$OUT/kernel_stats.csv"
grep '^{"metric"' "$OUT/bench_traced.log" > "$OUT/bench_traced.json"
i=0
for C in "SQ_WAVES SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_SMEM SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR SQ_WAVE_CYCLES SQ_BUSY_CYCLES" \
         "SQ_ACTIVE_INST_VALU SQ_THREAD_CYCLES_VALU SQ_ACTIVE_INST_ANY SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_SCA SQ_INSTS_LDS SQ_INSTS_VALU_FMA_F64" \
         "SQ_INSTS_VALU_ADD_F64 SQ_INSTS_VALU_MUL_F64 SQ_INSTS_VALU_TRANS_F64 SQ_INSTS_VALU_INT32 SQ_INSTS_VALU_INT64 SQ_INSTS_VALU_CVT" \
         "FETCH_SIZE" "WRITE_SIZE" "GRBM_GUI_ACTIVE"; do
  i=$((i+1))
  timeout -s KILL 180 rocprofv3 --pmc $C --output-format csv -d "$OUT/p$i" -- \
      python3 "$ROOT/bench.py" --steps 1 --warmup 0 --no-cpu-baseline > "$OUT/p$i.log" 2>&1 || { echo "pass $i failed"; tail -5 "$OUT/p$i.log"; exit 1; }
done
python3 "$ROOT/tools/pmc_summary.py" "$OUT" > "$OUT/pmc.json"
cat "$OUT/pmc.json"
