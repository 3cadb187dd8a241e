"""Test-side bindings of the two checkers.

Oracle  -- oracle/libptx_oracle.so, the CPU restatement of the hot path
           (travels with the repository, available everywhere).
Ref     -- oracle/_ref/libptref_f<flags>.so, the reference's own sources
           compiled by oracle/Makefile (only where that build was possible).

Neither is ever imported by the product package.
"""
import ctypes as C
import os

import numpy as np

from pathtracer_amd import abi
from pathtracer_amd.scene import Scene

REPO = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
ORACLE_LIB = os.path.join(REPO, "oracle", "libptx_oracle.so")
REF_DIR = os.path.join(REPO, "oracle", "_ref")
GOLDEN = os.path.join(REPO, "tests", "golden")
REFERENCE_ROOT = "/root/reference"

dp = C.POINTER(C.c_double)


def _p(a):
    return a.ctypes.data_as(dp)


class OrcStats(C.Structure):
    _fields_ = [(n, C.c_uint64) for n in
                ("paths", "closest_hits", "shadow_rays", "rindex_oob", "end_depth", "end_miss", "end_light", "end_rr")]

    def asdict(self):
        return {n: getattr(self, n) for n, _ in self._fields_}


class Oracle:
    def __init__(self, scene: Scene):
        if not os.path.exists(ORACLE_LIB):
            raise FileNotFoundError(f"{ORACLE_LIB}: build with `make -C oracle oracle`")
        lib = C.CDLL(ORACLE_LIB)
        lib.orc_scene_create.restype = C.c_void_p
        lib.orc_scene_create.argtypes = [C.POINTER(abi.SceneDesc)]
        lib.orc_scene_destroy.argtypes = [C.c_void_p]
        lib.orc_first_hit.argtypes = [C.c_void_p, C.c_void_p, C.c_int, C.c_void_p]
        lib.orc_render.argtypes = [C.c_void_p, C.POINTER(abi.RenderParams), C.c_int, C.c_int, dp, dp, C.c_int,
                                   C.POINTER(C.c_uint64), C.POINTER(OrcStats)]
        lib.orc_trace_pixels.argtypes = [C.c_void_p, C.POINTER(abi.RenderParams), C.c_void_p, C.c_void_p, C.c_void_p,
                                         C.c_int, dp]
        lib.orc_kat_cos_weighted.argtypes = [dp, dp, dp]
        lib.orc_kat_hemisphere_reorient.argtypes = [dp, dp, dp]
        lib.orc_kat_random_normal.argtypes = [dp]
        lib.orc_kat_random_normal.restype = C.c_double
        lib.orc_kat_refract.argtypes = [C.c_void_p, C.c_int, dp, dp, dp, dp, C.POINTER(C.c_int), dp, dp, dp]
        lib.orc_kat_reflect.argtypes = [dp, dp, dp, dp]
        lib.orc_kat_texmap.argtypes = [C.c_void_p, C.c_int, C.c_int, C.c_double, C.c_double, dp]
        lib.orc_kat_normal_deform.argtypes = [C.c_void_p, C.c_int, C.c_double, C.c_double, dp, dp]
        lib.orc_kat_primary_ray.argtypes = [C.c_void_p, C.c_int, C.c_int, C.c_int, C.c_double, dp, dp, dp]
        lib.orc_kat_light_sample.argtypes = [C.c_void_p, C.c_int, dp, dp]
        self.lib = lib
        self.scene = scene
        d = scene.desc()
        self.h = C.c_void_p(lib.orc_scene_create(C.byref(d)))

    def __del__(self):
        try:
            self.lib.orc_scene_destroy(self.h)
        except Exception:
            pass

    def first_hit(self, rays):
        rays = np.ascontiguousarray(rays, dtype=abi.RAY_DTYPE)
        hits = np.zeros(len(rays), dtype=abi.HIT_DTYPE)
        self.lib.orc_first_hit(self.h, rays.ctypes.data, len(rays), hits.ctypes.data)
        return hits

    def render(self, params, k0, nk, rgb=None, wght=None, threads=0, lcg_seed=None):
        sx = params.sx
        rgb = np.zeros(sx * sx * 3) if rgb is None else rgb
        wght = np.ones(sx * sx) if wght is None else wght
        st = OrcStats()
        lcg = C.c_uint64((((lcg_seed or 0) << 16) + 0x330E) & 0xFFFFFFFFFFFF)
        self.lib.orc_render(self.h, C.byref(params), k0, nk, _p(rgb), _p(wght), threads, C.byref(lcg), C.byref(st))
        return rgb, wght, st.asdict()

    def trace_pixels(self, params, pix_i, pix_j, sample):
        pi = np.ascontiguousarray(pix_i, dtype=np.int32)
        pj = np.ascontiguousarray(pix_j, dtype=np.int32)
        pk = np.ascontiguousarray(sample, dtype=np.int32)
        out = np.zeros((len(pi), 3))
        self.lib.orc_trace_pixels(self.h, C.byref(params), pi.ctypes.data, pj.ctypes.data, pk.ctypes.data, len(pi), _p(out))
        return out


class RefXform(C.Structure):
    _fields_ = [("op", C.c_int32), ("pad", C.c_int32), ("a", C.c_double), ("b", C.c_double), ("c", C.c_double)]


class RefObjSpec(C.Structure):
    _fields_ = [("type", C.c_int32), ("is_light", C.c_int32), ("n_ops", C.c_int32), ("pad", C.c_int32),
                ("diff", C.c_double), ("refl", C.c_double), ("tran", C.c_double), ("r", C.c_double), ("g", C.c_double),
                ("b", C.c_double), ("refl_sig", C.c_double), ("r_index", C.c_double), ("verts", C.c_double * 9),
                ("ops", RefXform * 8), ("tex", C.c_char * 128), ("nrm", C.c_char * 128), ("alpha", C.c_char * 128)]


class RefCamSpec(C.Structure):
    _fields_ = [("e", C.c_double * 3), ("g", C.c_double * 3), ("up", C.c_double * 3), ("f", C.c_double),
                ("wl", C.c_double), ("wt", C.c_double), ("wsize", C.c_double), ("focus", C.c_double)]


def ref_available(flags=7):
    return os.path.exists(os.path.join(REF_DIR, f"libptref_f{flags}.so")) and os.path.isdir(REFERENCE_ROOT)


class Ref:
    """The compiled reference.  Scenes load their assets relative to the working directory, so methods
    that build a scene run with the reference checkout (or `asset_dir`) as cwd."""

    def __init__(self, flags=7):
        path = os.path.join(REF_DIR, f"libptref_f{flags}.so")
        lib = C.CDLL(path)
        lib.ref_select_scene.argtypes = [C.c_char_p, C.c_int]
        lib.ref_build_custom.argtypes = [C.POINTER(RefObjSpec), C.c_int, C.POINTER(RefCamSpec), C.c_int]
        lib.ref_dump_scene.argtypes = [C.c_char_p]
        lib.ref_first_hit.argtypes = [C.c_void_p, C.c_int, C.c_void_p]
        lib.ref_render.argtypes = [C.c_int, C.c_int, C.c_int, C.c_double, dp, dp, C.c_int]
        lib.ref_trace_pixels.argtypes = [C.c_int, C.c_void_p, C.c_void_p, C.c_void_p, C.c_int, C.c_double, dp]
        lib.ref_rng_lcg.argtypes = [C.c_uint64]
        lib.ref_rng_counter.argtypes = [C.c_uint64]
        lib.ref_rng_script.argtypes = [dp, C.c_int]
        lib.ref_kat_cos_weighted.argtypes = [dp, dp]
        lib.ref_kat_hemisphere_reorient.argtypes = [dp, dp, dp]
        lib.ref_kat_random_normal.restype = C.c_double
        lib.ref_kat_refract.argtypes = [C.c_int, dp, dp, dp, dp, C.POINTER(C.c_int), dp, dp]
        lib.ref_kat_reflect.argtypes = [dp, dp, dp, dp]
        lib.ref_kat_texmap.argtypes = [C.c_int, C.c_int, C.c_double, C.c_double, dp]
        lib.ref_kat_normal_deform.argtypes = [C.c_int, C.c_double, C.c_double, dp, dp]
        lib.ref_kat_primary_ray.argtypes = [C.c_int, C.c_int, C.c_int, C.c_double, dp, dp]
        lib.ref_kat_light_sample.argtypes = [C.c_int, dp]
        self.lib = lib
        self.flags = flags

    def _in_dir(self, fn, asset_dir):
        cwd = os.getcwd()
        os.chdir(asset_dir or REFERENCE_ROOT)
        try:
            return fn()
        finally:
            os.chdir(cwd)

    def select_scene(self, name, max_depth, asset_dir=None):
        n = self._in_dir(lambda: self.lib.ref_select_scene(name.encode(), max_depth), asset_dir)
        if n < 0:
            raise ValueError(name)
        return n

    def build_custom(self, specs, cam, max_depth, asset_dir=None):
        arr = (RefObjSpec * len(specs))(*specs)
        n = self._in_dir(lambda: self.lib.ref_build_custom(arr, len(specs), C.byref(cam), max_depth), asset_dir)
        if n < 0:
            raise ValueError("bad custom scene")
        return n

    def dump_scene(self, path):
        rc = self.lib.ref_dump_scene(path.encode())
        if rc:
            raise RuntimeError(f"ref_dump_scene: {rc}")
        return Scene.from_file(path)

    def first_hit(self, rays):
        rays = np.ascontiguousarray(rays, dtype=abi.RAY_DTYPE)
        hits = np.zeros(len(rays), dtype=abi.HIT_DTYPE)
        self.lib.ref_first_hit(rays.ctypes.data, len(rays), hits.ctypes.data)
        return hits

    def render(self, sx, k0, nk, aperture, rgb=None, wght=None, threads=0, lcg_seed=None, counter_seed=None):
        rgb = np.zeros(sx * sx * 3) if rgb is None else rgb
        wght = np.ones(sx * sx) if wght is None else wght
        if counter_seed is not None:
            self.lib.ref_rng_counter(counter_seed)
        else:
            self.lib.ref_rng_lcg(lcg_seed or 0)
        self.lib.ref_render(sx, k0, nk, aperture, _p(rgb), _p(wght), threads)
        return rgb, wght

    def trace_pixels(self, sx, pix_i, pix_j, sample, aperture, counter_seed):
        pi = np.ascontiguousarray(pix_i, dtype=np.int32)
        pj = np.ascontiguousarray(pix_j, dtype=np.int32)
        pk = np.ascontiguousarray(sample, dtype=np.int32)
        out = np.zeros((len(pi), 3))
        self.lib.ref_rng_counter(counter_seed)
        self.lib.ref_trace_pixels(sx, pi.ctypes.data, pj.ctypes.data, pk.ctypes.data, len(pi), aperture, _p(out))
        return out


def random_rays(scene: Scene, n, seed, with_os=True):
    """Rays from random points inside the scene's extent plus camera rays, unit directions."""
    rng = np.random.default_rng(seed)
    rays = np.zeros(n, dtype=abi.RAY_DTYPE)
    lo, hi = -9.0, 9.0
    rays["p0"] = rng.uniform(lo, hi, (n, 3))
    d = rng.normal(size=(n, 3))
    rays["d"] = d / np.linalg.norm(d, axis=1, keepdims=True)
    rays["os"] = -1
    if with_os:
        pick = rng.random(n) < 0.3
        rays["os"][pick] = rng.integers(0, len(scene.objects), pick.sum())
    e = scene.camera["e"][0][:3]
    half = n // 2
    rays["p0"][:half] = e
    return rays
