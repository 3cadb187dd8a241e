import sys, glob, os, time, numpy as np, ctypes as C
sys.path.insert(0, "/root/repo")
import pathtracer_amd as pt
from pathtracer_amd import device
scene = pt.Scene.from_file("tests/golden/cornell.ptxscene")
libs = sorted(glob.glob(sys.argv[1])) if len(sys.argv)>1 else sorted(glob.glob("scratch/variants/*.so"))
sx = int(os.environ.get("SX","1024")); spp = int(os.environ.get("SPP","100"))
ref = None
for path in libs:
    device._LIB = device.load_library(path)
    dev = pt.Device(0); dev.scene_upload(scene)
    prm = pt.make_params(sx, 6, 7, 0.0, seed=1)
    dev.render_begin(prm); dev.render_passes(0, 4); dev.sync()
    ts=[]
    for r in range(3):
        dev.render_begin(prm); dev.render_passes(0, spp); dev.sync(); ts.append(dev.last_pass_ms())
    rgb,_ = dev.download()
    if ref is None: ref = rgb
    print(f"{os.path.basename(path):40s} ms {min(ts):9.2f}  Gpaths/s {sx*sx*spp/min(ts)/1e6:7.3f}  maxdiff_vs_first {np.abs(rgb-ref).max():.3e}", flush=True)
    dev.close()
