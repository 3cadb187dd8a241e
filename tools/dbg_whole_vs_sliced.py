#!/usr/bin/env python3
"""Where the whole-query and the sliced-query kernels differ: dbg_whole_vs_sliced.py scene size depth flags spp aperture"""
import os
import sys

import numpy as np

REPO = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, REPO)
sys.path.insert(0, os.path.join(REPO, "tools"))
import pathtracer_amd as pt
from walk_stats import load_scene

name, sx, depth, flags, spp, ap = sys.argv[1], int(sys.argv[2]), int(sys.argv[3]), int(sys.argv[4]), int(sys.argv[5]), float(sys.argv[6])
if os.environ.get("PTX_LIB"):
    from pathtracer_amd import device
    device._LIB = device.load_library(os.environ["PTX_LIB"])
dev = pt.Device(0)
scene = load_scene(name)
dev.scene_upload(scene)
out = {}
for kernel in ("whole", "sliced"):
    os.environ["PTX_BVH_KERNEL"] = kernel
    dev.render_begin(pt.make_params(sx, depth, flags, ap, seed=5))
    dev.render_passes(0, spp)
    out[kernel] = dev.download()
a, b = out["whole"][0].reshape(sx, sx, 3), out["sliced"][0].reshape(sx, sx, 3)
bad = np.argwhere(~((a == b) | (np.isnan(a) & np.isnan(b))).all(axis=2))
print(f"{len(bad)} of {sx * sx} pixels differ")
for j, i in bad[:10]:
    print(j, i, a[j, i], b[j, i], np.abs(a[j, i] - b[j, i]).max() / max(1e-300, np.abs(a[j, i]).max()))

if os.environ.get("WITH_ORACLE"):
    sys.path.insert(0, os.path.join(REPO, "tests"))
    from harness import Oracle
    want = Oracle(scene).render(pt.make_params(sx, depth, flags, ap, seed=5), 0, spp)[0].reshape(sx, sx, 3)
    for kernel, img in (("whole", a), ("sliced", b)):
        with np.errstate(invalid="ignore"):
            d = np.abs(img - want).max(axis=2) > 1e-9 + 1e-7 * np.abs(want).max(axis=2)
        print(f"{kernel}: {int(d.sum())} pixels differ from the oracle's")
        if kernel == "sliced":
            for j, i in np.argwhere(d)[:12]:
                print("   ", j, i, "sliced", img[j, i], "oracle", want[j, i], "whole", a[j, i])
