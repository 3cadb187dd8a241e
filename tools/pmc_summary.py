#!/usr/bin/env python3
"""Sums the rocprofv3 counter CSVs of tools/collect_profiles.sh for the path-tracing kernel and prints one JSON
object.  usage: pmc_summary.py <directory with p*/ passes> [workload label, default "bench"]

FETCH_SIZE and WRITE_SIZE are in KiB.  On gfx950 FETCH_SIZE counts a 128-byte read request as 64 bytes for wide
streaming reads; the kernel's accumulator reads are 8 bytes per lane, a width for which the counter is not
calibrated, so both the raw and the doubled figure are given.  The kernel's reads are scene records and its own
spilled registers; its writes are the traced samples (24 bytes per path) and spills."""
import collections
import csv
import glob
import hashlib
import json
import os
import re
import sys

REPO = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def library_stamp():
    """First 16 hex digits of the SHA-256 of the library the profiled process loaded: bench.py prints the same stamp of
    the library it runs, so a counter summary taken with another build of the kernels is recognisable as stale."""
    try:
        with open(os.path.join(REPO, "pathtracer_amd", "lib", "libptx_hip.so"), "rb") as f:
            return hashlib.sha256(f.read()).hexdigest()[:16]
    except OSError:
        return None


d = sys.argv[1]
tot = collections.defaultdict(float)
launches = collections.defaultdict(int)
names = collections.Counter()
for f in glob.glob(f"{d}/p*/**/*counter_collection.csv", recursive=True):
    seen = set()
    for r in csv.DictReader(open(f)):
        if "k_trace_samples" in r["Kernel_Name"]:
            tot[r["Counter_Name"]] += float(r["Counter_Value"])
            seen.add(r["Dispatch_Id"])
            names[re.sub(r"^void |\(.*$", "", r["Kernel_Name"])] += 1
    for c in {r["Counter_Name"] for r in csv.DictReader(open(f)) if "k_trace_samples" in r["Kernel_Name"]}:
        launches[c] = max(launches[c], len(seen))
# "kernel": the family the counters were filtered by (what bench.py looks for); "kernel_names": the instantiations seen
family = "k_trace_samples_bvh" if names and all(n.startswith("k_trace_samples_bvh") for n in names) else "k_trace_samples"
out = {"kernel": family, "kernel_names": sorted(names), "workload": sys.argv[2] if len(sys.argv) > 2 else "bench",
       "library_sha16": library_stamp(), "launches_per_run": max(launches.values()) if launches else 0,
       "counters": {k: tot[k] for k in sorted(tot)}}
n = max(1, out["launches_per_run"])
if "FETCH_SIZE" in tot and "WRITE_SIZE" in tot:
    out["hbm_bytes_per_launch"] = {
        "fetch_raw": tot["FETCH_SIZE"] * 1024 / n,
        "fetch_doubled": 2 * tot["FETCH_SIZE"] * 1024 / n,
        "write": tot["WRITE_SIZE"] * 1024 / n,
        "total": (2 * tot["FETCH_SIZE"] + tot["WRITE_SIZE"]) * 1024 / n,
    }
g = tot
if g.get("SQ_INSTS_VALU") and g.get("SQ_THREAD_CYCLES_VALU"):
    out["active_lanes_per_valu_inst"] = g["SQ_THREAD_CYCLES_VALU"] / g["SQ_INSTS_VALU"]
if g.get("SQ_INSTS_VALU") and g.get("SQ_WAVES"):
    out["valu_insts_per_wave"] = g["SQ_INSTS_VALU"] / g["SQ_WAVES"]
if g.get("SQ_ACTIVE_INST_VALU") and g.get("SQ_WAVE_CYCLES"):
    out["valu_active_share_of_wave_cycles"] = g["SQ_ACTIVE_INST_VALU"] / g["SQ_WAVE_CYCLES"]
if g.get("SQ_WAIT_ANY") and g.get("SQ_WAVE_CYCLES"):
    out["wait_share_of_wave_cycles"] = g["SQ_WAIT_ANY"] / g["SQ_WAVE_CYCLES"]
f64 = sum(g.get(k, 0) for k in ("SQ_INSTS_VALU_FMA_F64", "SQ_INSTS_VALU_ADD_F64", "SQ_INSTS_VALU_MUL_F64", "SQ_INSTS_VALU_TRANS_F64"))
if f64 and g.get("SQ_INSTS_VALU"):
    out["f64_share_of_valu_insts"] = f64 / g["SQ_INSTS_VALU"]
    out["f64_flop_counted"] = 2 * g.get("SQ_INSTS_VALU_FMA_F64", 0) + g.get("SQ_INSTS_VALU_ADD_F64", 0) + g.get("SQ_INSTS_VALU_MUL_F64", 0)
print(json.dumps(out, indent=1))
