#!/usr/bin/env python3
"""Roofline figures of one of the larger configurations (a scene with a hierarchy), from counters and a timed render.

    FLOP/ray  = 24 * node visits + sum over types (cost * primitive tests, list and leaves) + 40 * (share of rays that hit)
    bytes/ray = 32 * node visits + 160 * leaf primitive tests + 48 * texel look-ups   (what a lane must fetch per ray: a node
                is 32 bytes, a primitive's record 160, a bilinear look-up four texels of three bytes; the always-tested list
                is read through the scalar cache, once per wave, and is not counted)
with the per-test costs of the model: box 24, plane 43, sphere 58, cylinder 80, triangle 45, cube 65 FLOP.  The counts
come from the instrumented path kernel (ptx_trace_paths + ptx_last_walk_stats) on 2^18 random paths of the
configuration; rays/s from rendering the configuration at full size in 25-pass launches, as the command line does.
Peaks: 157.3 TFLOP/s (vector FP32; the arithmetic is FP64, whose vector peak is half) and 8 TB/s (HBM).

usage: config_roofline.py scifi|bubbletea|meshstress [spp, default 100] [json_out]"""
import json
import os
import sys
import time

import numpy as np

REPO = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, REPO)
sys.path.insert(0, os.path.join(REPO, "tools"))
import pathtracer_amd as pt
from walk_stats import CONFIGS, PRIM_FLOP, load_scene

TYPE_NAMES = ["plane", "sphere", "cylinder", "triangle", "cube", "room"]


def main():
    name = sys.argv[1]
    spp = int(sys.argv[2]) if len(sys.argv) > 2 else 100
    sx, depth, ap = CONFIGS[name]
    sc = load_scene(name)
    dev = pt.Device(0)
    dev.scene_upload(sc)
    # 1. what a query tests
    n = 1 << 18
    rng = np.random.default_rng(1)
    pi, pj, pk = rng.integers(0, sx, n), rng.integers(0, sx, n), rng.integers(0, 64, n)
    dev.trace_paths(pt.make_params(sx, depth, 7, ap, seed=1), pi, pj, pk)
    w = dev.last_walk_stats()
    q = w["queries"]
    nodes, hits, texels = w["node_visits"] / q, w["hits"] / q, w["texel_lookups"] / q
    leaf = [t / q for t in w["leaf_tests"]]
    n_list = round(w["list_tests"] / q)
    # the always-tested primitives by type: the largest boxes of the scene, as ptx_scene_upload picks them
    o = sc.objects
    ext = (o["box"][:, 1] - o["box"][:, 0]) + (o["box"][:, 3] - o["box"][:, 2]) + (o["box"][:, 5] - o["box"][:, 4])
    med = np.sort(ext)[len(ext) // 2]
    on_list = [k for k in np.argsort(-ext, kind="stable") if ext[k] > (8.0 if o["is_light"][k] else 16.0) * med][:96] if len(ext) > 48 else range(len(ext))
    list_by_type = [int(sum(1 for k in on_list if o["type"][k] == t)) for t in range(6)]
    assert sum(list_by_type) == n_list, (list_by_type, n_list)
    flop = 24 * nodes + sum(c * (a + b) for c, a, b in zip(PRIM_FLOP, leaf, list_by_type)) + 40 * hits
    byts = 32 * nodes + 160 * sum(leaf) + 48 * texels
    # The render kernel walks the four-wide form of the tree where the node arrays are far larger than the L2 caches: a
    # visit there is one 128-byte node and four box tests.  The path kernel that counts visits walks the binary form, so
    # the wide visits per ray come from the instrumented render kernel (tools/phase_timers.py: "node visits per query"),
    # given as PTX_WIDE_VISITS_PER_RAY.
    wide_visits = float(os.environ["PTX_WIDE_VISITS_PER_RAY"]) if os.environ.get("PTX_WIDE_VISITS_PER_RAY") else None
    if wide_visits is not None:
        flop = 4 * 24 * wide_visits + sum(c * (a + b) for c, a, b in zip(PRIM_FLOP, leaf, list_by_type)) + 40 * hits
        byts = 128 * wide_visits + 160 * sum(leaf) + 48 * texels
    # 2. rays per second at full size
    prm = pt.make_params(sx, depth, 7, ap, seed=1)
    dev.render_begin(prm)
    dev.render_passes(0, 4)
    dev.sync()
    dev.render_begin(prm)
    t0 = time.perf_counter()
    k = 0
    while k < spp:
        nk = min(25, spp - k)
        dev.render_passes(k, nk)
        k += nk
    dev.sync()
    dt = time.perf_counter() - t0
    st = dev.stats()
    rays = st["closest_hits"] + st["shadow_rays"]
    out = {
        "config": name, "size": sx, "depth": depth, "aperture": ap, "spp_timed": spp, "objects": int(len(o)),
        "kernel": os.environ.get("PTX_BVH_KERNEL", "sliced"),
        "walk": "binary" if wide_visits is None else "four-wide (visits per ray from tools/phase_timers.py)",
        "per_ray": {"node_visits": nodes, "wide_node_visits": wide_visits, "leaf_tests_by_type": dict(zip(TYPE_NAMES, leaf)), "list_tests_by_type": dict(zip(TYPE_NAMES, list_by_type)),
                    "hit_share": hits, "texel_lookups": texels},
        "flop_per_ray": flop, "bytes_per_ray": byts,
        "seconds": dt, "paths_per_s": sx * sx * spp / dt, "rays_per_s": rays / dt, "rays_per_path": rays / st["paths"],
        "roofline_valu": {"achieved_tflops": rays / dt * flop / 1e12, "peak_tflops": 157.3, "frac": rays / dt * flop / 157.3e12,
                          "frac_of_fp64_vector_peak": rays / dt * flop / 78.6e12},
        "roofline_hbm": {"achieved_gbs": rays / dt * byts / 1e9, "peak_gbs": 8000.0, "frac": rays / dt * byts / 8e12},
    }
    print(json.dumps(out))
    if len(sys.argv) > 3:
        with open(sys.argv[3], "w") as f:
            json.dump(out, f, indent=1)
    dev.close()


if __name__ == "__main__":
    main()
