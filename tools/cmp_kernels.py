"""Times the render kernels for hierarchy scenes against each other and checks that their images are equal.
usage: cmp_kernels.py size spp [whole | sliced[:exit_lanes,leaf_batch,shade_min,burst,small_run]] ..."""
import hashlib, os, sys, time
import numpy as np
REPO = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, REPO)
import pathtracer_amd as pt
from pathtracer_amd.scene import Scene
if os.environ.get("PTX_LIB"):  # an alternative build of the library, for A/B measurements
    from pathtracer_amd import device
    device._LIB = device.load_library(os.environ["PTX_LIB"])
size = int(sys.argv[1]) if len(sys.argv) > 1 else 512
spp = int(sys.argv[2]) if len(sys.argv) > 2 else 8
variants = sys.argv[3:] or ["whole", "sliced"]
sys.path.insert(0, os.path.join(REPO, "tools"))
from walk_stats import load_scene
ALL = {"scifi": (12, 0.05), "bubbletea": (12, 0.15), "meshstress": (8, 0.0)}
for name in os.environ.get("PTX_CMP_SCENES", "scifi,bubbletea").split(","):  # PTX_CMP_SCENES=meshstress for the mesh configuration
    depth, ap = ALL[name]
    sc = load_scene(name)
    dev = pt.Device(0); dev.scene_upload(sc)
    ref = None
    for v in variants:
        kern, _, rest = v.partition(":")
        os.environ["PTX_BVH_KERNEL"] = kern
        for var, val in zip(("PTX_WALK_EXIT_LANES", "PTX_WALK_LEAF_BATCH", "PTX_WALK_SHADE_MIN", "PTX_WALK_BURST", "PTX_WALK_SMALL_RUN"), rest.split(",") if rest else ()):
            os.environ[var] = val
        prm = pt.make_params(size, depth, 7, ap, seed=3)
        dev.render_begin(prm); dev.render_passes(0, 4); dev.sync()  # warm
        dev.render_begin(prm)
        t0 = time.perf_counter()
        k = 0
        while k < spp:
            nk = min(25, spp - k); dev.render_passes(k, nk); k += nk
        dev.sync(); dt = time.perf_counter() - t0
        rgb, w = dev.download(); st = dev.stats()
        same = "" if ref is None else f" same_as_first={np.array_equal(rgb, ref[0], equal_nan=True) and np.array_equal(w, ref[1], equal_nan=True) and st == ref[2]}"
        if ref is None: ref = (rgb, w, st)
        print(f"{name} {size}^2 {spp}spp {v}: {dt*1e3:.1f} ms, {size*size*spp/dt/1e6:.1f} M paths/s, image sha {hashlib.sha1(rgb.tobytes() + w.tobytes()).hexdigest()[:10]}{same}", flush=True)
    dev.close()
