#!/bin/bash
# The benchmark's workload under several settings of one environment variable: tools/sweep_bench.sh VAR v1 v2 ... [-- lib.so ...]
ROOT=$(cd "$(dirname "$0")/.." && pwd)
VAR=$1; shift
VALS=()
while [ $# -gt 0 ] && [ "$1" != "--" ]; do VALS+=("$1"); shift; done
[ "$1" == "--" ] && shift
for v in "${VALS[@]}"; do
  echo "== $VAR=$v"
  env $VAR=$v timeout -k 10 120 python3 $ROOT/tools/bench_lib.py "$@" || exit 1
done
