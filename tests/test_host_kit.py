"""The host scene kit (pathtracer_amd/host) against scene descriptions dumped from the reference.

The kit re-implements the reference's scene API, BoxTree builder and light bookkeeping; the flat scene it produces
must list the same surviving objects in the same visiting order with the same transforms, boxes, tree and lights
as the dump taken from the reference's own structures (tests/golden/*_scene.*).
"""
import ctypes as C
import os
import subprocess

import numpy as np
import pytest

from assets import synthetic_assets
from pathtracer_amd import abi
from pathtracer_amd.scene import Scene

REPO = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
HOST_LIB = os.path.join(REPO, "pathtracer_amd", "lib", "libptx_host.so")


@pytest.fixture(scope="module")
def asset_dir(tmp_path_factory):
    d = tmp_path_factory.mktemp("assets")
    have_mesh = synthetic_assets(str(d))
    return str(d), have_mesh


def build_with_kit(name, cwd, out_path, env=None):
    # a child process: the kit keeps its scene in globals and chdir is process-wide
    code = (
        "import ctypes,sys,os; os.chdir(sys.argv[3]); lib=ctypes.CDLL(sys.argv[1]);"
        "lib.ptxh_build_scene_file.argtypes=[ctypes.c_char_p,ctypes.c_char_p];"
        "sys.exit(0 if lib.ptxh_build_scene_file(sys.argv[2].encode(), sys.argv[4].encode())==0 else 1)"
    )
    subprocess.run(["python3", "-c", code, HOST_LIB, name, cwd, out_path], check=True, capture_output=True,
                   env=dict(os.environ, **(env or {})))
    return Scene.from_file(out_path)


def image_bytes(scene, idx):
    if idx < 0:
        return None
    im = scene.images[idx]
    n = int(im["sx"]) * int(im["sy"]) * int(im["channels"])
    off = int(im["data_offset"])
    return (int(im["sx"]), int(im["sy"]), int(im["channels"]), float(im["divisor"]), scene.texels[off:off + n].tobytes())


@pytest.mark.parametrize("name", ["cornell", "scifi", "bubbletea"])
def test_kit_scene_equals_reference_dump(name, asset_dir, golden_dir, tmp_path):
    cwd, have_mesh = asset_dir
    if name == "bubbletea" and not have_mesh:
        pytest.skip("objs/teapot.obj needs the reference checkout")
    want = Scene.load(os.path.join(golden_dir, "cornell.ptxscene" if name == "cornell" else f"{name}_scene.npz"))
    got = build_with_kit(name, cwd, str(tmp_path / f"{name}.bin"))
    assert len(got.objects) == len(want.objects)
    for f in ("type", "is_light", "src_id", "col", "v", "T", "box", "diffPct", "reflPct", "tranPct", "refl_sig", "r_index",
              "surfaceArea", "LSweight"):
        assert np.array_equal(got.objects[f], want.objects[f]), f
    # the reference inverts through an SVD, the kit by elimination: same matrix to rounding
    scale = np.maximum(1.0, np.abs(want.objects["Tinv"]).max(axis=1, keepdims=True))
    assert np.all(np.abs(got.objects["Tinv"] - want.objects["Tinv"]) <= 1e-13 * scale)
    for f in ("box", "child1", "child2", "first_obj", "n_objs"):
        assert np.array_equal(got.nodes[f], want.nodes[f]), f
    assert np.array_equal(got.lights, want.lights)
    for f in ("e", "u", "v", "w", "f", "focus_distance", "wl", "wt", "wsize", "C2W", "W2C"):
        assert np.array_equal(got.camera[f], want.camera[f]), f
    for k in range(len(want.objects)):
        for slot in ("tex_img", "normal_img", "alpha_img"):
            assert image_bytes(got, got.objects[slot][k]) == image_bytes(want, want.objects[slot][k]), (k, slot)


def test_invert_matches_numpy_and_flags_singular():
    lib = C.CDLL(HOST_LIB)
    dp = C.POINTER(C.c_double)
    lib.invert.argtypes = [dp, dp]
    rng = np.random.default_rng(3)
    for _ in range(50):
        M = np.eye(4)
        M[:3, :3] = rng.normal(size=(3, 3)) * rng.uniform(0.1, 500)
        M[:3, 3] = rng.normal(size=3) * 100
        out = np.zeros(16)
        lib.invert(np.ascontiguousarray(M).ctypes.data_as(dp), out.ctypes.data_as(dp))
        assert np.allclose(out.reshape(4, 4) @ M, np.eye(4), atol=1e-9)
    M = np.diag([1.0, 1.0, 1e-12, 1.0])  # a singular value below 1e-9: the reference hands back the identity
    out = np.zeros(16)
    lib.invert(np.ascontiguousarray(M).ctypes.data_as(dp), out.ctypes.data_as(dp))
    assert np.array_equal(out.reshape(4, 4), np.eye(4))


def test_tone_mapped_output_matches_reference_run(tmp_path):
    """dataOutput: same PFM bytes and same 8-bit image as the reference writes for the same accumulator."""
    from harness import REF_DIR, REFERENCE_ROOT, ref_available

    if not ref_available(7):
        pytest.skip("compiled reference not available")
    out = str(tmp_path / "bt.ppm")
    env = dict(os.environ, OMP_NUM_THREADS="1")
    subprocess.run([os.path.join(REF_DIR, "ref_cli_f7"), "5", "24", "5", "3", out, "0.15"], cwd=REFERENCE_ROOT, env=env,
                   check=True, capture_output=True)
    ref_ppm = open(out, "rb").read()
    with open(out + ".pfm", "rb") as f:
        header = f.readline() + f.readline() + f.readline()
        acc = np.frombuffer(f.read(), dtype=np.float64).copy()
    mine = str(tmp_path / "mine.ppm")
    lib = C.CDLL(HOST_LIB)
    lib.dataOutput.argtypes = [C.POINTER(C.c_double), C.c_int, C.c_char_p]
    lib.dataOutput(acc.ctypes.data_as(C.POINTER(C.c_double)), 24, mine.encode())
    assert open(mine, "rb").read() == ref_ppm
    assert open(mine + ".pfm", "rb").read() == header + acc.tobytes()


def test_flat_order_keeps_every_object_in_list_order(tmp_path):
    """PTX_FLAT_ORDER=1 (scenes too large for the reference's BoxTree builder): one node, nothing lost, objects in
    the order of the reference's object list (first inserted, then the others from last to second,
    utils_path.c:125-149); without it the BoxTree restatement runs and reorders."""
    n = 6
    flat = build_with_kit("meshstress", str(tmp_path), str(tmp_path / "flat.ptxscene"),
                          env={"PTX_FLAT_ORDER": "1", "PTX_MESH_N": str(n)})
    n_obj = 6 + 2 * 2 * n * n
    assert len(flat.objects) == n_obj and len(flat.nodes) == 1
    assert int(flat.nodes[0]["n_objs"]) == n_obj and int(flat.nodes[0]["child1"]) == -1
    assert flat.objects["src_id"].tolist() == [0] + list(range(n_obj - 1, 0, -1))
    assert (flat.objects["type"] == abi.TRIANGLE).sum() == n_obj - 6
    tree = build_with_kit("meshstress", str(tmp_path), str(tmp_path / "tree.ptxscene"),
                          env={"PTX_FLAT_ORDER": "0", "PTX_MESH_N": str(n)})
    assert len(tree.nodes) > 1 and len(tree.objects) <= n_obj
    assert sorted(tree.objects["src_id"].tolist()) == sorted(set(tree.objects["src_id"].tolist()))
    b = flat.objects["box"]
    assert np.allclose(flat.nodes[0]["box"], [b[:, 0].min(), b[:, 1].max(), b[:, 2].min(), b[:, 3].max(), b[:, 4].min(), b[:, 5].max()])


def test_load_obj_reads_plain_and_slashed_faces(tmp_path):
    """loadObj: `f a b c` as the reference reads it (meshes.c:81-99), and `f a/t/n b/t/n c/t/n`, which the reference's
    sscanf cannot read, give the same triangles; vt/vn lines between the blocks are skipped."""
    verts = [(0, 0, 5), (1, 0, 5), (1, 1, 6), (0, 1, 6), (0.5, 2, 7)]
    faces = [(1, 2, 3), (1, 3, 4), (4, 3, 5)]
    plain = tmp_path / "plain.obj"
    slashed = tmp_path / "slashed.obj"
    head = "# test mesh\n" + "".join(f"v {x} {y} {z}\n" for x, y, z in verts)
    extra = "vt 0 0\nvt 1 0\nvn 0 0 1\n"
    plain.write_text(head + extra + "".join(f"f {a} {b} {c}\n" for a, b, c in faces))
    slashed.write_text(head + extra + "".join(f"f {a}/1/1 {b}/2/1 {c}//1\n" for a, b, c in faces))
    # beyond the reference: a comment of 500 characters (its 128-byte line buffer would read the tail as lines of their
    # own: one piece here begins with "v " and would become a sixth vertex), and indices counted from the last vertex
    relative = tmp_path / "relative.obj"
    long_comment = "# " + "x" * 123 + "v 9 9 9 " + "y" * 360 + "\n"
    n = len(verts)
    relative.write_text(long_comment + head + extra + "".join(f"f {a - n - 1} {b - n - 1}/1 {c}\n" for a, b, c in faces))
    lib = C.CDLL(HOST_LIB)
    lib.ptxh_obj_scene_file.argtypes = [C.c_char_p, C.c_char_p]
    scenes = []
    for src in (plain, slashed, relative):
        out = str(tmp_path / (src.stem + ".ptxscene"))
        code = ("import ctypes,sys; lib=ctypes.CDLL(sys.argv[1]); lib.ptxh_obj_scene_file.argtypes=[ctypes.c_char_p,ctypes.c_char_p];"
                "sys.exit(0 if lib.ptxh_obj_scene_file(sys.argv[2].encode(), sys.argv[3].encode())==0 else 1)")
        subprocess.run(["python3", "-c", code, HOST_LIB, str(src), out], check=True, capture_output=True)
        scenes.append(Scene.from_file(out))
    a, b, c = scenes
    assert len(a.objects) == 1 + len(faces) == len(b.objects) == len(c.objects)
    assert np.array_equal(a.objects["v"], b.objects["v"]) and np.array_equal(a.objects["box"], b.objects["box"])
    assert np.array_equal(a.objects["v"], c.objects["v"]) and np.array_equal(a.objects["box"], c.objects["box"])
    tris = a.objects[a.objects["type"] == abi.TRIANGLE]
    by_id = {int(t["src_id"]): t for t in tris}
    for k, f in enumerate(faces):
        got = by_id[k + 1]["v"][:, :3]
        assert np.array_equal(got, np.array([verts[i - 1] for i in f], dtype=float))
