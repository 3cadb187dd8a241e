#!/bin/bash
# Tiled (PTX_TEXEL_LAYOUT=tiled) against linear (default) texel layout on SciFiCube (textured and normal-mapped walls): run time of 100 spp at 2048^2 and the L2
# requests of one 25-pass launch at 1024^2 (rocprofv3 --pmc TCC_REQ_sum TCC_HIT_sum TCC_MISS_sum, one run per layout).
#   tools/texel_layout_ab.sh  -> gpurun_out/texel_layout_ab.txt
ROOT=$(cd "$(dirname "$0")/.." && pwd)
OUT=$ROOT/gpurun_out
mkdir -p "$OUT"
cd /tmp && export TMPDIR=/tmp
: > "$OUT/texel_layout_ab.txt"
for L in linear tiled; do
  export PTX_TEXEL_LAYOUT=$L
  for rep in 1 2; do
    timeout -k 10 120 python3 "$ROOT/tools/run_config.py" scifi 2048 100 | tail -1 | sed "s/^/$L: /" >> "$OUT/texel_layout_ab.txt" || exit 1
  done
  rm -rf "$OUT/texel_$L"
  timeout -s KILL 180 rocprofv3 --pmc TCC_REQ_sum TCC_HIT_sum TCC_MISS_sum --output-format csv -d "$OUT/texel_$L" -- \
      python3 "$ROOT/tools/run_config.py" scifi 1024 25 > "$OUT/texel_$L.log" 2>&1 || { echo "pmc run failed"; exit 1; }
  python3 - "$OUT/texel_$L" "$L" >> "$OUT/texel_layout_ab.txt" <<'PY'
import csv, glob, sys, collections
tot = collections.defaultdict(float)
for f in glob.glob(sys.argv[1] + "/**/*counter_collection.csv", recursive=True):
    for r in csv.DictReader(open(f)):
        if "k_trace_samples" in r["Kernel_Name"]:
            tot[r["Counter_Name"]] += float(r["Counter_Value"])
print(sys.argv[2] + ": L2 requests of one 25-pass launch at 1024^2:", dict(tot))
PY
done
cat "$OUT/texel_layout_ab.txt"
