#!/usr/bin/env python3
"""Where the waves of the render kernels for hierarchy scenes spend their cycles (instrumented build of the library,
-DPTX_PHASE_TIMERS: `make -C pathtracer_amd dbg` -> pathtracer_amd/lib/libptx_hip_dbg.so).
usage: phase_timers.py scene size spp [whole|sliced]     (settings of the sliced kernel: PTX_WALK_* in the environment)"""
import ctypes
import os
import sys

REPO = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, REPO)
import pathtracer_amd as pt
from pathtracer_amd import device

device._LIB = device.load_library(os.path.join(REPO, "pathtracer_amd", "lib", "libptx_hip_dbg.so"))
CONFIGS = {"cornell": (6, 0.0), "scifi": (12, 0.05), "bubbletea": (12, 0.15), "meshstress": (8, 0.0)}
name, size, spp = sys.argv[1], int(sys.argv[2]), int(sys.argv[3])
kernel = sys.argv[4] if len(sys.argv) > 4 else "sliced"
os.environ["PTX_BVH_KERNEL"] = kernel
depth, ap = CONFIGS[name]
sys.path.insert(0, os.path.join(REPO, "tools"))
from walk_stats import load_scene
sc = load_scene(name)
dev = pt.Device(0)
dev.scene_upload(sc)
dev.render_begin(pt.make_params(size, depth, 7, ap, seed=3))
dev.render_passes(0, spp)
dev.sync()
st = dev.stats()
N_DBG = 36
device._LIB.ptx_debug_counters.argtypes = [ctypes.c_void_p, ctypes.POINTER(ctypes.c_ulonglong), ctypes.c_int]
if name == "cornell":
    # a scene without a hierarchy: the parts of k_trace_samples' loop (cycles as the waves see them, summed over the waves)
    buf = (ctypes.c_ulonglong * N_DBG)()
    assert device._LIB.ptx_debug_counters(dev.ctx, buf, 1) == 0
    v = list(buf)
    labels = ["closest-hit query", "surface (hit record, material, roulette)", "diffuse direction + end of the bounce", "light sample", "shadow query",
              "shadow result", "mirror", "refraction", "new path", "glossy deviates", "end of sample (store)", "whole loop"]
    total, iters = v[3 * 11], v[3 * 11 + 1]
    print(f"{name} {size}^2 {spp} spp: wave cycles total {total:.3e}, loop iterations {iters}, {total / iters:.0f} cycles per iteration, paths {st['paths']}")
    acc = 0
    for q, label in enumerate(labels[:-1]):
        cyc, runs, lanes = v[3 * q:3 * q + 3]
        acc += cyc
        print(f"  {label:42s} {cyc / total:6.1%} of cycles, in {runs / iters:5.1%} of iterations, {cyc / max(runs, 1):7.0f} cycles per run, "
              f"{lanes / max(runs, 1):5.1f} lanes per run")
    print(f"  {'loop control and the rest':42s} {1 - acc / total:6.1%}")
    dev.close()
    sys.exit(0)
if kernel == "whole":
    w = dev.last_walk_stats()
    total, lst, walk, iters = w["queries"], w["hits"], w["list_tests"], w["node_visits"]
    steps_wave, leaf_passes, nodes, queries, cyc_leaf, leaf_tests = w["leaf_tests"][:6]
    print(f"{name} {size}^2 {spp} spp (whole queries): wave cycles total {total:.3e}; always-tested list {lst / total:.1%}, hierarchy walk {walk / total:.1%} "
          f"(of which primitive tests at leaves {cyc_leaf / total:.1%}), rest (shading, sampling, regeneration) {1 - (lst + walk) / total:.1%}; "
          f"loop iterations summed over waves {iters}, paths {st['paths']}; per lane and query: {nodes / max(queries, 1):.2f} node visits, "
          f"{leaf_tests / max(queries, 1):.3f} leaf tests; per wave and query phase (2 per iteration): {steps_wave / (2 * iters):.1f} walk steps "
          f"with {nodes / max(steps_wave, 1):.1f} lanes active, {leaf_passes / (2 * iters):.2f} leaf-test passes with {leaf_tests / max(leaf_passes, 1):.1f} lanes active")
else:
    buf = (ctypes.c_ulonglong * N_DBG)()
    rc = device._LIB.ptx_debug_counters(dev.ctx, buf, 1)
    assert rc == 0, rc
    (total, shade, shadow, setup, steps, leaf, regen, n_iter, n_shade, l_shade, n_shadow, l_shadow, n_setup, l_setup, n_burst, l_burst,
     n_leaf, l_leaf, n_regen, l_regen, l_steps, l_active) = list(buf)[:22]
    queries = st["closest_hits"] + st["shadow_rays"]
    other = total - (shade + shadow + setup + steps + leaf + regen)
    print(f"{name} {size}^2 {spp} spp (sliced queries): wave cycles total {total:.3e}, {total / queries * 64:.0f} per 64 queries; "
          f"paths {st['paths']}, queries {queries}, outer iterations {n_iter} ({queries / n_iter:.1f} queries each, "
          f"{l_active / n_iter:.1f} lanes with samples left)")
    for label, cyc, n, lanes in (("shade_hit", shade, n_shade, l_shade), ("apply_shadow", shadow, n_shadow, l_shadow),
                                 ("sample end + regeneration", regen, n_regen, l_regen), ("query set-up (list)", setup, n_setup, l_setup),
                                 ("walk bursts", steps, n_burst, l_burst), ("leaf tests", leaf, n_leaf, l_leaf)):
        print(f"  {label:26s} {cyc / total:6.1%} of cycles, {n / n_iter:6.2f} runs per iteration, {cyc / max(n, 1):8.0f} cycles per run, "
              f"{lanes / max(n, 1):5.1f} lanes per run")
    print(f"  {'bookkeeping between':26s} {other / total:6.1%} of cycles; node visits per query {l_steps / queries:.2f}, per burst and lane {l_steps / max(l_burst, 1):.2f}")
dev.close()
