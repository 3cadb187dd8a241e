#!/usr/bin/env python3
"""Sweeps the settings of the render kernel for hierarchy scenes (PTX_WALK_*, read at every launch) on one scene that
is uploaded once.  usage: sweep_walk.py scene size spp  VAR=v1,v2,...  [VAR=...]     e.g.
    sweep_walk.py meshstress 2048 50 PTX_WALK_BURST=1,2 PTX_WALK_EXIT_LANES=32,48
Prints M paths/s for every combination (best of two renders), best first at the end."""
import itertools
import os
import sys
import time

REPO = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, REPO)
sys.path.insert(0, os.path.join(REPO, "tools"))
import pathtracer_amd as pt
from walk_stats import load_scene

CONFIGS = {"scifi": (12, 0.05), "bubbletea": (12, 0.15), "meshstress": (8, 0.0)}
name, sx, spp = sys.argv[1], int(sys.argv[2]), int(sys.argv[3])
axes = [(a.split("=")[0], a.split("=")[1].split(",")) for a in sys.argv[4:]]
depth, ap = CONFIGS[name]
dev = pt.Device(0)
dev.scene_upload(load_scene(name))
prm = pt.make_params(sx, depth, 7, ap, seed=1)


def render():
    dev.render_begin(prm)
    dev.sync()
    t0 = time.perf_counter()
    k = 0
    while k < spp:
        nk = min(25, spp - k)
        dev.render_passes(k, nk)
        k += nk
    dev.sync()
    return sx * sx * spp / (time.perf_counter() - t0) / 1e6


render()
results = []
for combo in itertools.product(*[v for _, v in axes]):
    for (var, _), val in zip(axes, combo):
        os.environ[var] = val
    rate = max(render(), render())
    label = " ".join(f"{var.replace('PTX_WALK_', '').lower()}={val}" for (var, _), val in zip(axes, combo))
    results.append((rate, label))
    print(f"{rate:8.1f} M paths/s  {label}", flush=True)
print("best:")
for rate, label in sorted(results, reverse=True)[:8]:
    print(f"{rate:8.1f} M paths/s  {label}")
dev.close()
