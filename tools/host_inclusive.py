#!/usr/bin/env python3
"""The benchmark with everything that crosses the host boundary inside the clock: scene upload, 1000 passes, download of
the accumulators into host memory (what the command line does before its final dataOutput).  usage: host_inclusive.py"""
import os
import sys
import time

REPO = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, REPO)
import pathtracer_amd as pt

scene = pt.Scene.load(os.path.join(REPO, "tests", "golden", "cornell.ptxscene"))
dev = pt.Device(0)
sx, spp = 1024, 1000
for rep in range(3):
    t0 = time.perf_counter()
    dev.scene_upload(scene)
    dev.render_begin(pt.make_params(sx, 6, 7, 0.0, seed=1 + rep))
    t1 = time.perf_counter()
    dev.render_passes(0, spp)
    dev.sync()
    t2 = time.perf_counter()
    rgb, w = dev.download()
    t3 = time.perf_counter()
    print(f"upload + begin {1e3 * (t1 - t0):.2f} ms, {spp} passes {1e3 * (t2 - t1):.2f} ms, download of {rgb.nbytes + w.nbytes} bytes {1e3 * (t3 - t2):.2f} ms "
          f"({(rgb.nbytes + w.nbytes) / (t3 - t2) / 1e9:.1f} GB/s): {sx * sx * spp / (t3 - t0) / 1e9:.3f} G paths/s with all of it, "
          f"{sx * sx * spp / (t2 - t1) / 1e9:.3f} G paths/s for the passes alone")
dev.close()
