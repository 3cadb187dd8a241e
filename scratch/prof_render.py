# one render launch of the benchmark scene, for counter collection
import sys, os
sys.path.insert(0, "/root/repo")
import pathtracer_amd as pt
scene = pt.Scene.from_file("/root/repo/tests/golden/cornell.ptxscene")
sx = int(os.environ.get("SX","1024")); spp = int(os.environ.get("SPP","40"))
dev = pt.Device(0); dev.scene_upload(scene)
prm = pt.make_params(sx, 6, 7, 0.0, seed=1)
dev.render_begin(prm); dev.render_passes(0, spp); dev.sync()
print("ms", dev.last_pass_ms(), dev.stats())
dev.close()
