#!/usr/bin/env python3
"""Time of the scene upload (hierarchy build included) of the mesh scene of a million triangles: with 1, 4 and 16 host
threads for the surface-area topology, and for eight contexts in a row (what PTX_GPUS=8 does; the topology is then made
once and kept).  usage: build_time.py [threads,threads,...]"""
import os
import sys
import time

REPO = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, REPO)
sys.path.insert(0, os.path.join(REPO, "tools"))
import pathtracer_amd as pt
from walk_stats import load_scene

t = time.time()
sc = load_scene("meshstress")
print(f"scene made on the host in {time.time() - t:.2f} s, {len(sc.objects)} objects")
pt.Device(0).close()  # the first context pays for the runtime's start
for thr in (sys.argv[1] if len(sys.argv) > 1 else "1,4,16").split(","):
    os.environ["PTX_BUILD_THREADS"] = thr
    os.environ["PTX_BVH_NO_CACHE"] = "1"
    dev = pt.Device(0)
    t = time.time()
    dev.scene_upload(sc)
    print(f"{thr:>2} threads: upload {time.time() - t:.2f} s")
    dev.close()
os.environ.pop("PTX_BVH_NO_CACHE")
os.environ.pop("PTX_BUILD_THREADS")
devs = []
t = time.time()
for r in range(8):
    d = pt.Device(0)
    d.scene_upload(sc)
    devs.append(d)
print(f"eight contexts, one after the other: {time.time() - t:.2f} s")
for d in devs:
    d.close()
