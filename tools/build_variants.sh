#!/bin/bash
# Alternative builds of the render kernels for A/B timing: tools/build_variants.sh name1 'flags1' name2 'flags2' ...
# -> pathtracer_amd/lib/libptx_hip_<name>.so (the flags are added to the Makefile's; time them with tools/bench_lib.py)
# The four-wide instances of the hierarchy kernel (ptx_wide.hip) are the product's unless VARIANT_WIDE=1 is set, which
# compiles them with the variant's flags too.
set -e
ROOT=$(cd "$(dirname "$0")/.." && pwd)
P=$ROOT/pathtracer_amd
mkdir -p $P/build/var
BASE="--offload-arch=gfx950 -O3 -std=c++17 -fPIC -I$ROOT/include -I$P/csrc -mllvm -disable-machine-licm -ffp-contract=on"
pids=()
while [ $# -ge 2 ]; do
  N=$1; F=$2; shift 2
  W=$P/build/ptx_wide.o
  if [ -n "$VARIANT_WIDE" ]; then W=$P/build/var/${N}_wide.o; fi
  ( /opt/rocm/bin/hipcc $BASE $F -c $P/csrc/ptx_hip.hip -o $P/build/var/$N.o 2> $P/build/var/$N.log \
    && { [ -z "$VARIANT_WIDE" ] || /opt/rocm/bin/hipcc $BASE $F -c $P/csrc/ptx_wide.hip -o $W 2>> $P/build/var/$N.log; } \
    && /opt/rocm/bin/hipcc --offload-arch=gfx950 -shared -fPIC $P/build/var/$N.o $W $P/build/ptx_bvh.o $P/build/ptx_tonemap.o -o $P/lib/libptx_hip_$N.so \
    && echo "built $N" || echo "FAILED $N (see build/var/$N.log)" ) &
  pids+=($!)
done
for p in "${pids[@]}"; do wait $p; done
