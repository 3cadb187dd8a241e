#!/bin/bash
# Disassembly of one kernel of the library's device code: tools/kernel_isa.sh 'k_trace_samples_bvh<7, false>' [out.s]
# (default object pathtracer_amd/build/ptx_hip.o; PTX_OBJ overrides)
set -e
REPO=$(cd "$(dirname "$0")/.." && pwd)
OBJ=${PTX_OBJ:-$REPO/pathtracer_amd/build/ptx_hip.o}
LLVM=/opt/rocm/lib/llvm/bin
TMP=$(mktemp -d)
cp "$OBJ" "$TMP/o.o"
(cd "$TMP" && $LLVM/llvm-objdump --offloading o.o >/dev/null)
CO=$(ls "$TMP" | grep amdgcn | head -1)
$LLVM/llvm-objdump -d --no-show-raw-insn "$TMP/$CO" | c++filt > "$TMP/all.s"
awk -v name="<void $1(" 'index($0, name) && /^[0-9a-f]+ </ {on=1} on && /^$/ {exit} on {print}' "$TMP/all.s" > "${2:-/dev/stdout}"
rm -rf "$TMP"
