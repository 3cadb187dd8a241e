#!/usr/bin/env python3
"""Calls the GPU tone mapping a few times on a rendered image (for rocprofv3 --kernel-trace --stats)."""
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import pathtracer_amd as pt

REPO = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sx = int(os.environ.get("SX", "2048"))
dev = pt.Device(0)
dev.scene_upload(pt.Scene.from_file(os.path.join(REPO, "tests", "golden", "cornell.ptxscene")))
dev.render_begin(pt.make_params(sx, 6, 7, 0.0, seed=1))
dev.render_passes(0, 4)
for _ in range(10):
    bits, info = dev.tonemap()
print(info)
dev.close()
