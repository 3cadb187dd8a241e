#!/usr/bin/env python3
"""What the four-wide walk does on the mesh scene (a library built with -DPTX_WIDE_DIAG: tools/build_variants.sh diag
'-DPTX_WIDE_DIAG -DPTX_ONLY_FLAGS7').  usage: wide_diag.py [size spp]"""
import ctypes
import os
import sys

REPO = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, REPO)
sys.path.insert(0, os.path.join(REPO, "tools"))
import pathtracer_amd as pt
from pathtracer_amd import device
from walk_stats import load_scene

device._LIB = device.load_library(os.path.join(REPO, "pathtracer_amd", "lib", "libptx_hip_diag.so"))
size = int(sys.argv[1]) if len(sys.argv) > 1 else 1024
spp = int(sys.argv[2]) if len(sys.argv) > 2 else 25
dev = pt.Device(0)
dev.scene_upload(load_scene("meshstress"))
dev.render_begin(pt.make_params(size, 8, 7, 0.0, seed=1))
dev.render_passes(0, spp)
dev.sync()
st = dev.stats()
buf = (ctypes.c_ulonglong * 16)()
assert device._LIB.ptx_debug_wide_diag(buf) == 0
v = list(buf)
rays = st["closest_hits"] + st["shadow_rays"]
print(f"rays {rays}; node visits {v[0]} ({v[0] / rays:.2f} per ray), of which no child entered {v[1]} ({v[1] / v[0]:.1%}), three or four entered {v[7] / v[0]:.1%}")
print(f"children pushed {v[2]} ({v[2] / v[0]:.2f} per visit); visits that pushed beyond the LDS part {v[3]} ({v[3] / v[0]:.2%}); pops from the spill {v[4]}")
print(f"leaf pops {v[5]} ({v[5] / rays:.2f} per ray), of which the ray's own surface {v[6]}")
tot = sum(v[8:16])
print("stack depth at a pop (<4, <8, <12, <16, <20, <28, <40, more): " + " ".join(f"{x / tot:.1%}" for x in v[8:16]))
dev.close()
