#!/bin/bash
# A/B timing of alternative builds of the library on the hierarchy scenes: tools/ab_libs.sh size spp lib1.so lib2.so ...
# (paths relative to pathtracer_amd/lib; each is timed with tools/cmp_kernels.py sliced sliced)
set -e
ROOT=$(cd "$(dirname "$0")/.." && pwd)
SIZE=$1; SPP=$2; shift 2
for L in "$@"; do
  echo "== $L"
  PTX_LIB=$ROOT/pathtracer_amd/lib/$L timeout -k 10 300 python3 $ROOT/tools/cmp_kernels.py $SIZE $SPP whole sliced sliced 2>&1 | tail -9
done
