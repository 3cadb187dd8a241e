#!/usr/bin/env python3
"""Registers and scratch memory of the kernels in a compiled object, read from the code object's metadata.
--check exits 0 (within budget), 1 (budget exceeded) or 3 (the tools that read the code object failed: no verdict).

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

import shutil

REPO = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
BUDGET = {"vgpr": 128, "scratch": 80}
# exit codes of --check: 0 within budget, 1 budget exceeded, 3 the tools that read the code object could not be run or
# understood (a different toolchain layout: not a verdict on the kernel)
EXIT_EXCEEDED, EXIT_TOOL = 1, 3


class ToolFailure(Exception):
    pass


def llvm_tool(name):
    """llvm-objdump / llvm-readelf of the ROCm in use: $ROCM_PATH, hipcc's own directory, /opt/rocm, then the PATH"""
    roots = [os.environ.get("ROCM_PATH"), os.environ.get("HIP_PATH")]
    hipcc = shutil.which("hipcc")
    if hipcc:
        roots.append(os.path.dirname(os.path.dirname(os.path.realpath(hipcc))))
    roots.append("/opt/rocm")
    for root in roots:
        if root:
            for sub in ("lib/llvm/bin", "llvm/bin", "bin"):
                cand = os.path.join(root, sub, name)
                if os.path.isfile(cand) and os.access(cand, os.X_OK):
                    return cand
    found = shutil.which(name)
    if found:
        return found
    raise ToolFailure(f"{name} not found (looked under ROCM_PATH, beside hipcc, in /opt/rocm and on the PATH)")


def kernels(obj):
    """[(demangled name, vgprs, agprs, sgprs, scratch bytes per lane, lds bytes)] of every kernel in obj"""
    with tempfile.TemporaryDirectory() as tmp:
        local = os.path.join(tmp, "o.o")
        with open(obj, "rb") as f, open(local, "wb") as g:
            g.write(f.read())
        try:
            subprocess.run([llvm_tool("llvm-objdump"), "--offloading", local], check=True, capture_output=True, cwd=tmp)
            cos = [os.path.join(tmp, n) for n in os.listdir(tmp) if "amdgcn" in n]
            if not cos:
                raise ToolFailure(f"{obj}: llvm-objdump --offloading left no device code object")
            notes = subprocess.run([llvm_tool("llvm-readelf"), "--notes", cos[0]], check=True, capture_output=True, text=True).stdout
        except (OSError, subprocess.CalledProcessError) as e:
            raise ToolFailure(f"reading the code object of {obj} failed: {e}")
    blocks = notes.split("- .agpr_count:")[1:]
    if not blocks:
        raise ToolFailure(f"{obj}: the code object's notes are not laid out as expected (no '.agpr_count' entries)")
    demangle = shutil.which("c++filt")
    out = []
    for block in blocks:
        def field(name, default="0"):
            m = re.search(rf"\.{name}:\s+(\S+)", block)
            return m.group(1) if m else default
        sym = field("name", "?")
        agpr = int(re.match(r"\s*(\d+)", block).group(1))
        name = subprocess.run([demangle, sym], capture_output=True, text=True).stdout.strip() if demangle else sym
        out.append((name, int(field("vgpr_count")), agpr, int(field("sgpr_count")), int(field("private_segment_fixed_size")),
                    int(field("group_segment_fixed_size"))))
    return out


def main():
    args = [a for a in sys.argv[1:] if a != "--check"]
    obj = args[0] if args else os.path.join(REPO, "pathtracer_amd", "build", "ptx_hip.o")
    try:
        ks = kernels(obj)
        if "--check" in sys.argv and not shutil.which("c++filt"):
            raise ToolFailure("c++filt not found: kernel names cannot be demangled")
    except ToolFailure as e:
        print("kernel_regs: TOOL FAILURE (no verdict on the kernel):", e, file=sys.stderr)
        sys.exit(EXIT_TOOL)
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
        sys.exit(EXIT_EXCEEDED if bad else 0)


if __name__ == "__main__":
    main()
