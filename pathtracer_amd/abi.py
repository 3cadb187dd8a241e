"""ctypes / numpy mirrors of the C structs in include/ptx_scene.h and include/ptx_hip.h.

Nothing here computes anything: these are the layouts that cross the C ABI.
"""
import ctypes as C

import numpy as np

PTX_SCENE_MAGIC = 0x31435358545450

PLANE, SPHERE, CYL, TRIANGLE, CUBE, ROOM = range(6)
USE_IS, USE_ES, USE_RR = 1, 2, 4
CLEAN = 8  # PTX_CLEAN of ptx_scene.h: not the reference's estimator
RNG_COUNTER, RNG_LCG48 = 0, 1

OBJECT_DTYPE = np.dtype(
    [
        ("type", "<i4"), ("is_light", "<i4"), ("tex_img", "<i4"), ("normal_img", "<i4"), ("alpha_img", "<i4"),
        ("src_id", "<i4"), ("col", "<f8", (3,)), ("v", "<f8", (3, 4)), ("T", "<f8", (16,)), ("Tinv", "<f8", (16,)),
        ("box", "<f8", (6,)), ("diffPct", "<f8"), ("reflPct", "<f8"), ("tranPct", "<f8"), ("refl_sig", "<f8"),
        ("r_index", "<f8"), ("surfaceArea", "<f8"), ("LSweight", "<f8"),
    ],
    align=True,
)
NODE_DTYPE = np.dtype(
    [("box", "<f8", (6,)), ("child1", "<i4"), ("child2", "<i4"), ("first_obj", "<i4"), ("n_objs", "<i4")], align=True
)
IMAGE_DTYPE = np.dtype(
    [("sx", "<i4"), ("sy", "<i4"), ("channels", "<i4"), ("pad", "<i4"), ("divisor", "<f8"), ("data_offset", "<u8")],
    align=True,
)
CAMERA_DTYPE = np.dtype(
    [
        ("e", "<f8", (4,)), ("u", "<f8", (4,)), ("v", "<f8", (4,)), ("w", "<f8", (4,)), ("f", "<f8"),
        ("focus_distance", "<f8"), ("wl", "<f8"), ("wt", "<f8"), ("wsize", "<f8"), ("C2W", "<f8", (16,)),
        ("W2C", "<f8", (16,)),
    ],
    align=True,
)
RAY_DTYPE = np.dtype([("p0", "<f8", (3,)), ("d", "<f8", (3,)), ("os", "<i4"), ("pad", "<i4")], align=True)
HIT_DTYPE = np.dtype(
    [("lambda", "<f8"), ("p", "<f8", (3,)), ("n", "<f8", (3,)), ("a", "<f8"), ("b", "<f8"), ("obj", "<i4"), ("pad", "<i4")],
    align=True,
)
assert OBJECT_DTYPE.itemsize == 504 and NODE_DTYPE.itemsize == 64 and IMAGE_DTYPE.itemsize == 32
assert CAMERA_DTYPE.itemsize == 424 and RAY_DTYPE.itemsize == 56 and HIT_DTYPE.itemsize == 80


class Camera(C.Structure):
    _fields_ = [
        ("e", C.c_double * 4), ("u", C.c_double * 4), ("v", C.c_double * 4), ("w", C.c_double * 4),
        ("f", C.c_double), ("focus_distance", C.c_double), ("wl", C.c_double), ("wt", C.c_double),
        ("wsize", C.c_double), ("C2W", C.c_double * 16), ("W2C", C.c_double * 16),
    ]


class SceneDesc(C.Structure):
    _fields_ = [
        ("magic", C.c_uint64), ("n_objects", C.c_int32), ("n_nodes", C.c_int32), ("n_lights", C.c_int32),
        ("n_images", C.c_int32), ("n_texel_bytes", C.c_uint64), ("objects", C.c_void_p), ("nodes", C.c_void_p),
        ("lights", C.c_void_p), ("images", C.c_void_p), ("texels", C.c_void_p), ("cam", Camera),
    ]


class RenderParams(C.Structure):
    _fields_ = [
        ("sx", C.c_int32), ("max_depth", C.c_int32), ("flags", C.c_int32), ("rng_mode", C.c_int32),
        ("aperture", C.c_double), ("seed", C.c_uint64), ("rank", C.c_int32), ("n_ranks", C.c_int32),
        ("band_rows", C.c_int32), ("pad", C.c_int32),
    ]


class Stats(C.Structure):
    _fields_ = [("paths", C.c_uint64), ("closest_hits", C.c_uint64), ("shadow_rays", C.c_uint64), ("rindex_oob", C.c_uint64)]


class WalkStats(C.Structure):
    _fields_ = [("queries", C.c_uint64), ("hits", C.c_uint64), ("list_tests", C.c_uint64), ("node_visits", C.c_uint64),
                ("leaf_tests", C.c_uint64 * 6), ("texel_lookups", C.c_uint64)]


class TonemapInfo(C.Structure):
    _fields_ = [("min", C.c_double), ("max", C.c_double), ("low_cut", C.c_double), ("high_cut", C.c_double),
                ("recomputed_on_host", C.c_uint64)]


assert C.sizeof(Camera) == CAMERA_DTYPE.itemsize


def rows_owned(sx, rank, n_ranks, band_rows):
    """ptx_rows_owned of include/ptx_scene.h."""
    if n_ranks <= 1:
        return sx
    n = 0
    for j0 in range(0, sx, band_rows):
        if (j0 // band_rows) % n_ranks == rank:
            n += min(band_rows, sx - j0)
    return n


def global_rows(sx, rank, n_ranks, band_rows):
    """Global row index of every local row of (rank, n_ranks, band_rows), in local order."""
    if n_ranks <= 1:
        return np.arange(sx)
    rows = [j for j in range(sx) if (j // band_rows) % n_ranks == rank]
    return np.asarray(rows, dtype=np.int64)
