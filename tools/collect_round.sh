#!/bin/bash
# Everything the round's figures are quoted from, on a GPU box:  tools/collect_round.sh <tag>
#  - tools/collect_profiles.sh <tag> (kernel trace and counters of the benchmark)
#  - the benchmark kernel's phase table (instrumented library), the roofline figures of the three hierarchy configurations,
#    and a plain bench.py run, under gpurun_out/<tag>_cfg/
set -e
TAG=${1:?tag}
ROOT=$(cd "$(dirname "$0")/.." && pwd)
bash "$ROOT/tools/collect_profiles.sh" "$TAG" > "$ROOT/gpurun_out/${TAG}_collect.log" 2>&1
echo "benchmark profiles done"
cd "$ROOT"
OUT=gpurun_out/${TAG}_cfg
mkdir -p $OUT
timeout -k 10 300 python3 tools/phase_timers.py cornell 1024 1000 > $OUT/cornell_phases.txt 2>&1
for s in scifi bubbletea meshstress; do
  W=""; [ $s = meshstress ] && W=4.63   # (four-wide visits per ray of the mesh scene, from tools/phase_timers.py meshstress)
  PTX_WIDE_VISITS_PER_RAY=$W timeout -k 10 300 python3 tools/config_roofline.py $s 100 $OUT/roofline_$s.json > $OUT/roofline_$s.log 2>&1
  tail -2 $OUT/roofline_$s.log
done
timeout -k 10 600 python3 bench.py > $OUT/bench.json 2> $OUT/bench.err
cat $OUT/bench.json
timeout -k 10 300 python3 tools/build_time.py 1,8 > $OUT/build_time.txt 2>&1
cat $OUT/build_time.txt
for s in scifi bubbletea meshstress; do
  timeout -k 10 300 python3 tools/phase_timers.py $s 2048 25 > $OUT/${s}_phases.txt 2>&1
done
for s in scifi bubbletea meshstress; do
  bash tools/collect_config_profiles.sh ${TAG}_$s $s 1024 > gpurun_out/${TAG}_${s}_collect.log 2>&1 || { echo "counters of $s failed"; tail -3 gpurun_out/${TAG}_${s}_collect.log; }
done
