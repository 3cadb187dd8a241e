import sys, numpy as np
sys.path.insert(0, "/root/repo"); sys.path.insert(0, "/root/repo/tests")
import pathtracer_amd as pt
from harness import Oracle
scene = pt.Scene.from_file("tests/golden/cornell.ptxscene")
orc = Oracle(scene); dev = pt.Device(0); dev.scene_upload(scene)
prm = pt.make_params(64, 6, 7, 0.0, seed=99)
n = 20000
rng = np.random.default_rng(5)
pi = rng.integers(0,64,n); pj = rng.integers(0,64,n); pk = rng.integers(0,50,n)
g = dev.trace_paths(prm, pi, pj, pk); c = orc.trace_pixels(prm, pi, pj, pk)
err = np.abs(g-c).max(axis=1)
bad = np.where(err>1e-9)[0]
print(len(bad), "bad of", n)
for q in bad[:40]: print(pi[q], pj[q], pk[q], g[q], c[q])
