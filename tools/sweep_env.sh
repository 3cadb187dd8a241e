#!/bin/bash
# times the hierarchy scenes under several values of one environment variable: tools/sweep_env.sh VAR size spp v1 v2 ...
ROOT=$(cd "$(dirname "$0")/.." && pwd)
VAR=$1; SIZE=$2; SPP=$3; shift 3
for V in "$@"; do
  echo "== $VAR=$V"
  env $VAR=$V timeout -k 10 300 python3 $ROOT/tools/cmp_kernels.py $SIZE $SPP sliced 2>&1 | tail -3 || exit 1
done
