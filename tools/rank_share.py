#!/usr/bin/env python3
"""What one GPU's share of an N-GPU render of the benchmark costs, measured on one GPU: rank 0's row bands of an N-rank
split (Cornell 1024^2, depth 6, 1000 spp in one launch), kernels only, no gather.  usage: rank_share.py [N ...]"""
import os
import sys
import time

REPO = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, REPO)
import pathtracer_amd as pt

sc = pt.Scene.load(os.path.join(REPO, "tests", "golden", "cornell.ptxscene"))
dev = pt.Device(0)
dev.scene_upload(sc)
base = None
for n in [int(a) for a in sys.argv[1:]] or [1, 2, 4, 8]:
    prm = pt.make_params(1024, 6, 7, 0.0, seed=1, rank=0, n_ranks=n, band_rows=8)
    dev.render_begin(prm); dev.render_passes(0, 100); dev.sync()
    best = 1e9
    for rep in range(3):
        dev.render_begin(prm)
        t0 = time.perf_counter()
        dev.render_passes(0, 1000)
        dev.sync()
        best = min(best, time.perf_counter() - t0)
    base = base or best
    print(f"N={n}: {best * 1e3:.1f} ms (ideal {base * 1e3 / n:.1f}, {base / n / best:.1%} of linear)", flush=True)
dev.close()
