import sys, os, numpy as np
sys.path.insert(0, "/root/repo")
import pathtracer_amd as pt
scene = pt.Scene.from_file("tests/golden/cornell.ptxscene")
dev = pt.Device(0); dev.scene_upload(scene)
sx, spp = 1024, 60
for flags in (0,1,2,4,5,6,7):
    for depth in (6,):
        prm = pt.make_params(sx, depth, flags, 0.0, seed=1)
        dev.render_begin(prm); dev.render_passes(0, 4); dev.sync()
        dev.render_begin(prm); dev.render_passes(0, spp); dev.sync()
        ms = dev.last_pass_ms(); st = dev.stats()
        it = st["closest_hits"]/64  # wave-iterations if fully packed
        print(f"flags {flags} ms {ms:8.2f} Gpaths/s {sx*sx*spp/ms/1e6:6.3f} closest/path {st['closest_hits']/st['paths']:.2f} shadow/path {st['shadow_rays']/st['paths']:.2f} ns per closest-iteration-wave {ms*1e6/it*1024*2/1:.0f} (SIMD-ns)")
