#!/usr/bin/env python3
"""Renders one of the larger benchmark configurations on one GPU in 25-pass launches (as the command line does) and
prints throughput.  usage: run_config.py scifi|bubbletea|meshstress [size spp]"""
import os
import sys
import time

REPO = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, REPO)
import numpy as np

import pathtracer_amd as pt

if os.environ.get("PTX_LIB"):  # an alternative build of the library (a file name under pathtracer_amd/lib), for A/B measurements
    from pathtracer_amd import device
    device._LIB = device.load_library(os.path.join(REPO, "pathtracer_amd", "lib", os.environ["PTX_LIB"]))

CONFIGS = {"scifi": (2048, 12, 400, 0.05), "bubbletea": (2048, 12, 2000, 0.15), "meshstress": (2048, 8, 100, 0.0)}
name = sys.argv[1]
sx, depth, spp, ap = CONFIGS[name]
if len(sys.argv) > 3:
    sx, spp = int(sys.argv[2]), int(sys.argv[3])
sys.path.insert(0, os.path.join(REPO, "tools"))
from walk_stats import load_scene
scene = load_scene(name)
dev = pt.Device(0)
dev.scene_upload(scene)
prm = pt.make_params(sx, depth, 7, ap, seed=1)
dev.render_begin(prm)
t0 = time.perf_counter()
k = 0
while k < spp:
    nk = min(25, spp - k)
    dev.render_passes(k, nk)
    k += nk
    if k % 100 == 0 or k == spp:
        dev.sync()
        print(f"  {k}/{spp} passes, {time.perf_counter() - t0:.1f} s", flush=True)
dev.sync()
dt = time.perf_counter() - t0
st = dev.stats()
rays = st["closest_hits"] + st["shadow_rays"]
bits, info = dev.tonemap()
print(f"{name} {sx}x{sx} depth {depth} {spp} spp aperture {ap}: {dt:.2f} s, {sx*sx*spp/dt/1e6:.1f} M paths/s, "
      f"{rays/dt/1e6:.1f} M rays/s, rays/path {rays/st['paths']:.2f}, r_index out-of-range accesses {st['rindex_oob']}, "
      f"image mean byte {bits.mean():.1f}, finite {bool(np.isfinite([info['min'], info['max']]).all())}")
dev.close()
