#!/usr/bin/env python3
"""The benchmark's workload (Cornell 1024^2, depth 6, 1000 spp in one launch, IS+ES+RR) timed on an alternative build of
the library, with a checksum of the image: bench_lib.py [lib.so ...]   (paths relative to pathtracer_amd/lib)"""
import hashlib
import os
import sys
import time

REPO = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, REPO)
import pathtracer_amd as pt
from pathtracer_amd import device

sc = pt.Scene.load(os.path.join(REPO, "tests", "golden", "cornell.ptxscene"))
for name in sys.argv[1:] or ["libptx_hip.so"]:
    device._LIB = device.load_library(os.path.join(REPO, "pathtracer_amd", "lib", name))
    dev = pt.Device(0)
    dev.scene_upload(sc)
    prm = pt.make_params(1024, 6, 7, 0.0, seed=1)
    dev.render_begin(prm); dev.render_passes(0, 100); dev.sync()
    best = 1e9
    for rep in range(3):
        dev.render_begin(prm)
        t0 = time.perf_counter()
        dev.render_passes(0, 1000)
        dev.sync()
        best = min(best, time.perf_counter() - t0)
    rgb, w = dev.download()
    print(f"{name}: {best * 1e3:.1f} ms, {1024 * 1024 * 1000 / best / 1e9:.3f} G paths/s, image sha {hashlib.sha1(rgb.tobytes() + w.tobytes()).hexdigest()[:12]}", flush=True)
    dev.close()
