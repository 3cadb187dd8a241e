import os, sys
import numpy as np
REPO = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, REPO)
import pathtracer_amd as pt
from pathtracer_amd.scene import Scene
for name, ap in (("scifi", 0.05), ("bubbletea", 0.15), ("scifi", 0.0)):
    sc = Scene.load(os.path.join(REPO, "tests/golden", f"{name}_scene.npz"))
    dev = pt.Device(0); dev.scene_upload(sc)
    for depth in (1, 2, 3, 12):
      for flags in (7, 0):
        res = {}
        for kern in ("whole", "sliced"):
            os.environ["PTX_BVH_KERNEL"] = kern
            prm = pt.make_params(64, depth, flags, ap, seed=3)
            dev.render_begin(prm); dev.render_passes(0, 4); dev.sync()
            rgb, w = dev.download(); res[kern] = (rgb, w, dev.stats())
        a, b = res["whole"], res["sliced"]
        diff = ~np.isclose(a[0], b[0], rtol=0, atol=0, equal_nan=True)
        print(name, ap, "depth", depth, "flags", flags, "diff values", int(diff.sum()), "of", diff.size, a[2], b[2], flush=True)
    dev.close()
