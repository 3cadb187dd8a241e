#!/bin/bash
# Instruction- and scalar-cache counters of one 25-pass launch on a hierarchy scene (on a GPU box):
#   tools/collect_icache.sh <tag> scifi|bubbletea [size]   -> gpurun_out/<tag>/icache_*.csv
set -e
TAG=${1:?tag}
SCENE=${2:?scene}
SIZE=${3:-1024}
ROOT=$(cd "$(dirname "$0")/.." && pwd)
OUT=$ROOT/gpurun_out/$TAG
mkdir -p "$OUT"
cd /tmp && export TMPDIR=/tmp
i=0
for C in "SQC_ICACHE_REQ SQC_ICACHE_HITS SQC_ICACHE_MISSES SQ_IFETCH SQ_WAVE_CYCLES SQ_WAIT_INST_ANY" \
         "SQC_DCACHE_REQ SQC_DCACHE_HITS SQC_DCACHE_MISSES SQ_INSTS_SMEM SQ_WAIT_ANY SQ_BUSY_CYCLES"; do
  i=$((i+1))
  timeout -s KILL 180 rocprofv3 --pmc $C --output-format csv -d "$OUT/ic$i" -- \
      python3 "$ROOT/tools/run_config.py" "$SCENE" "$SIZE" 25 > "$OUT/ic$i.log" 2>&1 || { echo "pass $i failed"; tail -5 "$OUT/ic$i.log"; exit 1; }
  F=$(find "$OUT/ic$i" -name '*counter_collection.csv' | head -1)
  python3 - "$F" <<'PY'
import csv, sys, collections
tot = collections.defaultdict(float)
for r in csv.DictReader(open(sys.argv[1])):
    if "k_trace_samples" in r["Kernel_Name"]:
        tot[r["Counter_Name"]] += float(r["Counter_Value"])
print(dict(tot))
PY
done
