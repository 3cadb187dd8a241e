"""Python binding of libptx_hip.so (include/ptx_hip.h) through ctypes.

The method names follow the C entry points one to one; arguments have the
meaning the header gives them.  There is no CPU fallback: if the shared
library is missing or a call fails, an exception is raised.
"""
import ctypes as C
import os

import numpy as np

from . import abi

_LIB = None
LIB_PATH = os.path.join(os.path.dirname(os.path.abspath(__file__)), "lib", "libptx_hip.so")

EXPORTS = [
    "ptx_init", "ptx_shutdown", "ptx_scene_upload", "ptx_render_begin", "ptx_render_begin_external",
    "ptx_render_passes", "ptx_sync", "ptx_download", "ptx_accumulators", "ptx_first_hit", "ptx_first_hit_device",
    "ptx_trace_paths", "ptx_get_stats", "ptx_last_pass_ms", "ptx_last_trace_ms", "ptx_set_stream", "ptx_last_error", "ptx_version",
    "ptx_tonemap", "ptx_tonemap_device", "ptx_selftest", "ptx_upload", "ptx_last_walk_stats",
    "ptx_render_passes_static", "ptx_static_max_pixels", "ptx_device_count", "ptx_gather",
]


class PtxError(RuntimeError):
    def __init__(self, code, msg):
        super().__init__(f"ptx error {code}: {msg}")
        self.code = code


def load_library(path=None):
    """Loads libptx_hip.so (built by pathtracer_amd/Makefile).  Raises if it is absent."""
    global _LIB
    if _LIB is not None and path is None:
        return _LIB
    p = path or LIB_PATH
    if not os.path.exists(p):
        raise FileNotFoundError(f"{p} not found: build it with `make -C pathtracer_amd` (hipcc, gfx950)")
    lib = C.CDLL(p)
    vp, i32p, dp = C.c_void_p, C.POINTER(C.c_int32), C.POINTER(C.c_double)
    lib.ptx_init.argtypes = [C.c_int, C.POINTER(vp)]
    lib.ptx_shutdown.argtypes = [vp]
    lib.ptx_shutdown.restype = None
    lib.ptx_scene_upload.argtypes = [vp, C.POINTER(abi.SceneDesc)]
    lib.ptx_render_begin.argtypes = [vp, C.POINTER(abi.RenderParams)]
    lib.ptx_render_begin_external.argtypes = [vp, C.POINTER(abi.RenderParams), vp, vp]
    lib.ptx_render_passes.argtypes = [vp, C.c_int, C.c_int]
    lib.ptx_render_passes_static.argtypes = [vp, C.c_int, C.c_int]
    lib.ptx_static_max_pixels.restype = C.c_size_t
    lib.ptx_device_count.argtypes = [C.POINTER(C.c_int)]
    lib.ptx_gather.argtypes = [C.POINTER(vp), C.c_int, C.POINTER(vp), C.POINTER(vp)]
    lib.ptx_sync.argtypes = [vp]
    lib.ptx_download.argtypes = [vp, vp, vp]
    lib.ptx_upload.argtypes = [vp, vp, vp]
    lib.ptx_accumulators.argtypes = [vp, C.POINTER(vp), C.POINTER(vp), i32p]
    lib.ptx_first_hit.argtypes = [vp, vp, C.c_int64, vp]
    lib.ptx_first_hit_device.argtypes = [vp, vp, C.c_int64, vp]
    lib.ptx_trace_paths.argtypes = [vp, C.POINTER(abi.RenderParams), vp, vp, vp, C.c_int64, vp]
    lib.ptx_get_stats.argtypes = [vp, C.POINTER(abi.Stats)]
    lib.ptx_last_walk_stats.argtypes = [vp, C.POINTER(abi.WalkStats)]
    lib.ptx_last_pass_ms.argtypes = [vp, C.POINTER(C.c_float)]
    lib.ptx_last_trace_ms.argtypes = [vp, C.POINTER(C.c_float), C.POINTER(C.c_int)]
    lib.ptx_set_stream.argtypes = [vp, vp]
    lib.ptx_selftest.argtypes = [vp, dp]
    lib.ptx_tonemap.argtypes = [vp, vp, C.POINTER(abi.TonemapInfo)]
    lib.ptx_tonemap_device.argtypes = [vp, vp, C.c_uint64, vp, C.POINTER(abi.TonemapInfo)]
    lib.ptx_last_error.restype = C.c_char_p
    lib.ptx_version.restype = C.c_char_p
    if path is None:
        _LIB = lib
    return lib


def make_params(sx, max_depth, flags, aperture=0.0, seed=0, rank=0, n_ranks=1, band_rows=8):
    return abi.RenderParams(sx, max_depth, flags, abi.RNG_COUNTER, aperture, seed, rank, n_ranks, band_rows, 0)


def gather(devices):
    """ptx_gather: the rows of devices[r] (rank r of len(devices)) collected on devices[0]'s GPU.  Returns the device
    addresses of the whole-image rgb and weight buffers."""
    lib = load_library()
    arr = (C.c_void_p * len(devices))(*[d.ctx for d in devices])
    a, b = C.c_void_p(), C.c_void_p()
    devices[0]._check(lib.ptx_gather(arr, len(devices), C.byref(a), C.byref(b)))
    return a.value, b.value


class Device:
    """One ptx_context: one GPU."""

    def __init__(self, device=0):
        self.lib = load_library()
        self.ctx = C.c_void_p()
        self._check(self.lib.ptx_init(device, C.byref(self.ctx)))
        self.params = None

    def _check(self, rc):
        if rc != 0:
            raise PtxError(rc, self.lib.ptx_last_error().decode())

    def close(self):
        if self.ctx:
            self.lib.ptx_shutdown(self.ctx)
            self.ctx = C.c_void_p()

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass

    def scene_upload(self, scene):
        d = scene.desc()
        self._check(self.lib.ptx_scene_upload(self.ctx, C.byref(d)))

    def render_begin(self, params, d_rgb=None, d_wght=None):
        self.params = params
        if d_rgb is None:
            self._check(self.lib.ptx_render_begin(self.ctx, C.byref(params)))
        else:
            self._check(self.lib.ptx_render_begin_external(self.ctx, C.byref(params), d_rgb, d_wght))

    def render_passes(self, k0, nk):
        self._check(self.lib.ptx_render_passes(self.ctx, k0, nk))

    def render_passes_static(self, k0, nk):
        """render_passes without any allocation: the samples go through the library's fixed device array."""
        self._check(self.lib.ptx_render_passes_static(self.ctx, k0, nk))

    def sync(self):
        self._check(self.lib.ptx_sync(self.ctx))

    def set_stream(self, stream_ptr):
        self._check(self.lib.ptx_set_stream(self.ctx, stream_ptr))

    def download(self, rgb=None, wght=None):
        sx = self.params.sx
        if rgb is None:
            rgb = np.zeros(sx * sx * 3)
        if wght is None:
            wght = np.ones(sx * sx)
        self._check(self.lib.ptx_download(self.ctx, rgb.ctypes.data, wght.ctypes.data))
        return rgb, wght

    def upload(self, rgb=None, wght=None):
        """Sets the accumulators from whole-image arrays (sx*sx*3 and sx*sx doubles)."""
        rgb = None if rgb is None else np.ascontiguousarray(rgb, dtype=np.float64)
        wght = None if wght is None else np.ascontiguousarray(wght, dtype=np.float64)
        self._check(self.lib.ptx_upload(self.ctx, None if rgb is None else rgb.ctypes.data,
                                        None if wght is None else wght.ctypes.data))

    def accumulators(self):
        a, b, n = C.c_void_p(), C.c_void_p(), C.c_int32()
        self._check(self.lib.ptx_accumulators(self.ctx, C.byref(a), C.byref(b), C.byref(n)))
        return a.value, b.value, n.value

    def first_hit(self, rays):
        rays = np.ascontiguousarray(rays, dtype=abi.RAY_DTYPE)
        hits = np.zeros(len(rays), dtype=abi.HIT_DTYPE)
        self._check(self.lib.ptx_first_hit(self.ctx, rays.ctypes.data, len(rays), hits.ctypes.data))
        return hits

    def trace_paths(self, params, pix_i, pix_j, sample):
        pi = np.ascontiguousarray(pix_i, dtype=np.int32)
        pj = np.ascontiguousarray(pix_j, dtype=np.int32)
        pk = np.ascontiguousarray(sample, dtype=np.int32)
        out = np.zeros((len(pi), 3))
        self._check(self.lib.ptx_trace_paths(self.ctx, C.byref(params), pi.ctypes.data, pj.ctypes.data, pk.ctypes.data,
                                             len(pi), out.ctypes.data))
        return out

    def tonemap(self, d_im=None, n_values=None):
        """Bytes of the reference's P6 output for this context's accumulator (or for a device buffer of n_values
        doubles) and what dataOutput() reports while making them."""
        info = abi.TonemapInfo()
        if d_im is None:
            n_values = self.params.sx * self.params.sx * 3
            bits = np.zeros(n_values, dtype=np.uint8)
            self._check(self.lib.ptx_tonemap(self.ctx, bits.ctypes.data, C.byref(info)))
        else:
            bits = np.zeros(n_values, dtype=np.uint8)
            self._check(self.lib.ptx_tonemap_device(self.ctx, d_im, n_values, bits.ctypes.data, C.byref(info)))
        return bits, {"min": info.min, "max": info.max, "low_cut": info.low_cut, "high_cut": info.high_cut,
                      "recomputed_on_host": info.recomputed_on_host}

    def selftest(self):
        r = (C.c_double * 5)()
        self._check(self.lib.ptx_selftest(self.ctx, r))
        return {"texel_mismatches": r[0], "log_ulp": r[1], "sqrt_ulp": r[2], "rcp_ulp": r[3], "rsqrt_ulp": r[4]}

    def stats(self):
        s = abi.Stats()
        self._check(self.lib.ptx_get_stats(self.ctx, C.byref(s)))
        return {"paths": s.paths, "closest_hits": s.closest_hits, "shadow_rays": s.shadow_rays, "rindex_oob": s.rindex_oob}

    def last_walk_stats(self):
        """What the queries of the last trace_paths call tested (the instrumented path kernel)."""
        w = abi.WalkStats()
        self._check(self.lib.ptx_last_walk_stats(self.ctx, C.byref(w)))
        return {"queries": w.queries, "hits": w.hits, "list_tests": w.list_tests, "node_visits": w.node_visits,
                "leaf_tests": list(w.leaf_tests), "texel_lookups": w.texel_lookups}

    def last_pass_ms(self):
        ms = C.c_float()
        self._check(self.lib.ptx_last_pass_ms(self.ctx, C.byref(ms)))
        return ms.value

    def last_trace_ms(self):
        """(milliseconds in the path-tracing kernel during the last render_passes call, number of its launches)"""
        ms, n = C.c_float(), C.c_int()
        self._check(self.lib.ptx_last_trace_ms(self.ctx, C.byref(ms), C.byref(n)))
        return ms.value, n.value
