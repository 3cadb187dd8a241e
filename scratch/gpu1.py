import sys, time, numpy as np
sys.path.insert(0, "/root/repo"); sys.path.insert(0, "/root/repo/tests")
import pathtracer_amd as pt
from harness import Oracle, random_rays
scene = pt.Scene.from_file(sys.argv[1])
orc = Oracle(scene)
dev = pt.Device(0)
dev.scene_upload(scene)
rays = random_rays(scene, 20000, 3)
h_gpu = dev.first_hit(rays); h_cpu = orc.first_hit(rays)
print("first-hit obj mismatches:", int((h_gpu["obj"] != h_cpu["obj"]).sum()), "of", len(rays), "hits", int((h_cpu["obj"]>=0).sum()))
m = (h_cpu["obj"] >= 0) & (h_gpu["obj"] == h_cpu["obj"])
rel = np.abs(h_gpu["lambda"][m]-h_cpu["lambda"][m])/np.abs(h_cpu["lambda"][m])
print("lambda max rel", rel.max(), "n max abs", np.abs(h_gpu["n"][m]-h_cpu["n"][m]).max(), "p max abs", np.abs(h_gpu["p"][m]-h_cpu["p"][m]).max(),
      "a", np.abs(h_gpu["a"][m]-h_cpu["a"][m]).max(), "b", np.abs(h_gpu["b"][m]-h_cpu["b"][m]).max())
for flags in (7, 0):
    prm = pt.make_params(64, 6, flags, 0.0, seed=99)
    n = 4096
    rng = np.random.default_rng(5)
    pi = rng.integers(0,64,n); pj = rng.integers(0,64,n); pk = rng.integers(0,50,n)
    g = dev.trace_paths(prm, pi, pj, pk); c = orc.trace_pixels(prm, pi, pj, pk)
    err = np.abs(g-c).max(axis=1)
    print("flags", flags, "paths: exact-ish (<1e-9):", int((err<1e-9).sum()), "of", n, "max err", err.max(), "mean col", g.mean(), c.mean())
    sx, spp = 64, 16
    prm = pt.make_params(sx, 6, flags, 0.0, seed=7)
    dev.render_begin(prm); dev.render_passes(0, spp); rgb, w = dev.download()
    rgb2, w2, st = orc.render(prm, 0, spp)
    d = np.abs(rgb-rgb2)
    print("render", sx, spp, "max abs diff", d.max(), "frac>1e-9", (d>1e-9).mean(), "mean", rgb.mean(), rgb2.mean(), "stats", dev.stats(), st, "ms", dev.last_pass_ms())
t=time.time()
prm = pt.make_params(1024, 6, 7, 0.0, seed=1)
dev.render_begin(prm); dev.render_passes(0, 8); dev.sync(); print("1024^2 x8spp ms", dev.last_pass_ms(), "paths/s", 1024*1024*8/dev.last_pass_ms()*1e3, dev.stats())
dev.render_passes(8, 32); dev.sync(); print("1024^2 x32spp ms", dev.last_pass_ms(), "paths/s", 1024*1024*32/dev.last_pass_ms()*1e3)
