#!/usr/bin/env python3
"""Registers and scratch memory of the kernels in a compiled object, read from the code object's metadata.

usage: kernel_regs.py [--check] [object file, default pathtracer_amd/build/ptx_hip.o]

--check: exit non-zero when the benchmark's tracing kernel (k_trace_samples<7, false, true>: small scene, zero
aperture, IS+ES+RR) uses more than its budget of 128 vector registers or 80 bytes of scratch per lane.  That
kernel depends on a compiler-internal switch (-mllvm -disable-machine-licm, see pathtracer_amd/Makefile); a
toolchain that ignores the switch brings back about 500 bytes of scratch per lane and a quarter of the run time,
and this check is what notices.
"""
import os
import re
import subprocess
import sys
import tempfile

REPO = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
LLVM = "/opt/rocm/lib/llvm/bin"
BUDGET = {"vgpr": 128, "scratch": 80}


def kernels(obj):
    """[(demangled name, vgprs, agprs, sgprs, scratch bytes per lane, lds bytes)] of every kernel in obj"""
    with tempfile.TemporaryDirectory() as tmp:
        local = os.path.join(tmp, "o.o")
        with open(obj, "rb") as f, open(local, "wb") as g:
            g.write(f.read())
        subprocess.run([f"{LLVM}/llvm-objdump", "--offloading", local], check=True, capture_output=True, cwd=tmp)
        cos = [os.path.join(tmp, n) for n in os.listdir(tmp) if "amdgcn" in n]
        if not cos:
            raise SystemExit(f"{obj}: no device code object found")
        notes = subprocess.run([f"{LLVM}/llvm-readelf", "--notes", cos[0]], check=True, capture_output=True, text=True).stdout
    out = []
    for block in notes.split("- .agpr_count:")[1:]:
        def field(name, default="0"):
            m = re.search(rf"\.{name}:\s+(\S+)", block)
            return m.group(1) if m else default
        sym = field("name", "?")
        agpr = int(re.match(r"\s*(\d+)", block).group(1))
        name = subprocess.run(["c++filt", sym], capture_output=True, text=True).stdout.strip()
        out.append((name, int(field("vgpr_count")), agpr, int(field("sgpr_count")), int(field("private_segment_fixed_size")),
                    int(field("group_segment_fixed_size"))))
    return out


def main():
    args = [a for a in sys.argv[1:] if a != "--check"]
    obj = args[0] if args else os.path.join(REPO, "pathtracer_amd", "build", "ptx_hip.o")
    ks = kernels(obj)
    bad = []
    for name, vgpr, agpr, sgpr, scratch, lds in sorted(ks):
        short = re.sub(r"\(.*", "", name).replace("void ", "")
        print(f"{short:60s} vgpr {vgpr:3d} agpr {agpr:3d} sgpr {sgpr:3d} scratch {scratch:4d} B/lane  lds {lds:6d} B")
        if "--check" in sys.argv and short.replace(" ", "") == "k_trace_samples<7,false,true>":
            if vgpr > BUDGET["vgpr"] or scratch > BUDGET["scratch"]:
                bad.append(f"{short}: vgpr {vgpr} (budget {BUDGET['vgpr']}), scratch {scratch} B/lane (budget {BUDGET['scratch']})")
    if "--check" in sys.argv:
        if not any("k_trace_samples<7, false, true>" in k[0] for k in ks):
            bad.append("k_trace_samples<7, false, true> not found in the object")
        for b in bad:
            print("BUDGET EXCEEDED:", b, file=sys.stderr)
        sys.exit(1 if bad else 0)


if __name__ == "__main__":
    main()
