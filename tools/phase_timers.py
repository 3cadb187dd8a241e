#!/usr/bin/env python3
"""Where the waves of the whole-query render kernel spend their cycles (debug build of the library with
-DPTX_PHASE_TIMERS: pathtracer_amd/lib/libptx_hip_dbg.so).  usage: phase_timers.py scene size spp"""
import os
import sys

REPO = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, REPO)
import pathtracer_amd as pt
from pathtracer_amd import device

device._LIB = device.load_library(os.path.join(REPO, "pathtracer_amd", "lib", "libptx_hip_dbg.so"))
os.environ["PTX_BVH_KERNEL"] = "whole"
CONFIGS = {"cornell": (6, 0.0), "scifi": (12, 0.05), "bubbletea": (12, 0.15)}
name, size, spp = sys.argv[1], int(sys.argv[2]), int(sys.argv[3])
depth, ap = CONFIGS[name]
path = "cornell.ptxscene" if name == "cornell" else f"{name}_scene.npz"
sc = pt.Scene.load(os.path.join(REPO, "tests", "golden", path))
dev = pt.Device(0)
dev.scene_upload(sc)
dev.render_begin(pt.make_params(size, depth, 7, ap, seed=3))
dev.render_passes(0, spp)
dev.sync()
w = dev.last_walk_stats()
st = dev.stats()
total, lst, walk, iters = w["queries"], w["hits"], w["list_tests"], w["node_visits"]
steps_wave, leaf_passes, nodes, queries, cyc_leaf, leaf_tests = w["leaf_tests"][:6]
wave_queries = (st["closest_hits"] + st["shadow_rays"]) / 64.0  # upper bound on lane occupancy: every lane in a query
print(f"{name} {size}^2 {spp} spp: wave cycles total {total:.3e}; always-tested list {lst / total:.1%}, hierarchy walk {walk / total:.1%} "
      f"(of which primitive tests at leaves {cyc_leaf / total:.1%}), rest (shading, sampling, regeneration) {1 - (lst + walk) / total:.1%}; "
      f"loop iterations summed over waves {iters}, paths {st['paths']}; per lane and query: {nodes / max(queries, 1):.2f} node visits, "
      f"{leaf_tests / max(queries, 1):.3f} leaf tests; per wave and query phase (2 per iteration): {steps_wave / (2 * iters):.1f} walk steps "
      f"with {nodes / max(steps_wave, 1):.1f} lanes active, {leaf_passes / (2 * iters):.2f} leaf-test passes with {leaf_tests / max(leaf_passes, 1):.1f} lanes active")
dev.close()
