#!/usr/bin/env python3
"""What the closest-hit queries of a configuration test, from the instrumented path kernel (ptx_trace_paths +
ptx_last_walk_stats), turned into the per-ray FLOP and byte figures of the roofline model:
    FLOP/ray  = 24 * node visits + sum over types (cost * primitive tests) + 40 (hit attributes)
    bytes/ray = 32 * node visits + 160 * leaf primitive tests (the geometry record read per lane) + 12 * 4 * texel look-ups / rays
usage: walk_stats.py cornell|scifi|bubbletea|meshstress [n_paths] [json_out]"""
import ctypes
import json
import os
import subprocess
import sys
import tempfile

import numpy as np

REPO = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, REPO)
import pathtracer_amd as pt

CONFIGS = {"cornell": (1024, 6, 0.0), "scifi": (2048, 12, 0.05), "bubbletea": (2048, 12, 0.15), "meshstress": (2048, 8, 0.0)}
# per-test costs of the model: plane, sphere, cylinder, triangle, cube, room
PRIM_FLOP = [43, 58, 80, 45, 65, 65]


def load_scene(name):
    if name == "cornell":
        return pt.Scene.load(os.path.join(REPO, "tests", "golden", "cornell.ptxscene"))
    if name != "meshstress":
        return pt.Scene.load(os.path.join(REPO, "tests", "golden", f"{name}_scene.npz"))
    out = os.path.join(tempfile.mkdtemp(), "mesh.ptxscene")
    code = ("import ctypes,sys; lib=ctypes.CDLL(sys.argv[1]); lib.ptxh_build_scene_file.argtypes=[ctypes.c_char_p,ctypes.c_char_p];"
            "sys.exit(0 if lib.ptxh_build_scene_file(b'meshstress', sys.argv[2].encode())==0 else 1)")
    subprocess.run([sys.executable, "-c", code, os.path.join(REPO, "pathtracer_amd", "lib", "libptx_host.so"), out], check=True,
                   env=dict(os.environ, PTX_FLAT_ORDER="1", PTX_MESH_N=os.environ.get("PTX_MESH_N", "512")))
    sc = pt.Scene.from_file(out)
    os.unlink(out)
    return sc


def main():
    name = sys.argv[1]
    n = int(sys.argv[2]) if len(sys.argv) > 2 else 1 << 20
    sx, depth, ap = CONFIGS[name]
    sc = load_scene(name)
    dev = pt.Device(0)
    dev.scene_upload(sc)
    rng = np.random.default_rng(1)
    pi, pj, pk = rng.integers(0, sx, n), rng.integers(0, sx, n), rng.integers(0, 64, n)
    dev.trace_paths(pt.make_params(sx, depth, 7, ap, seed=1), pi, pj, pk)
    w = dev.last_walk_stats()
    q = w["queries"]
    types = sc.objects["type"]
    per_ray = {
        "list_tests": w["list_tests"] / q, "node_visits": w["node_visits"] / q,
        "leaf_tests": [t / q for t in w["leaf_tests"]], "hit_fraction": w["hits"] / q,
        "texel_lookups": w["texel_lookups"] / q,
    }
    # the always-tested list: its primitives by type (each query tests all of them)
    n_list = round(per_ray["list_tests"])
    flop = 24 * per_ray["node_visits"] + sum(c * t for c, t in zip(PRIM_FLOP, per_ray["leaf_tests"])) + 40 * per_ray["hit_fraction"]
    out = {"config": name, "size": sx, "depth": depth, "aperture": ap, "paths_sampled": n, "queries_per_path": q / n,
           "objects": int(len(types)), "per_query": per_ray, "list_primitives": n_list,
           "flop_per_query_without_list": flop,
           "bytes_per_query": 32 * per_ray["node_visits"] + 160 * sum(per_ray["leaf_tests"]) + 48 * per_ray["texel_lookups"]}
    print(json.dumps(out))
    if len(sys.argv) > 3:
        with open(sys.argv[3], "w") as f:
            json.dump(out, f, indent=1)
    dev.close()


if __name__ == "__main__":
    main()
