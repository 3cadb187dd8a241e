"""GPU tone mapping (ptx_tonemap*) against the host dataOutput(), which is itself pinned byte for byte to the
reference's output (tests/test_host_kit.py::test_tone_mapped_output_matches_reference_run)."""
import ctypes as C
import os

import numpy as np
import pytest

import pathtracer_amd as pt

pytestmark = pytest.mark.gpu

REPO = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
HOST_LIB = os.path.join(REPO, "pathtracer_amd", "lib", "libptx_host.so")


def host_data_output(acc, sx, tmp_path, tag):
    """P6 payload written by the host dataOutput() for the accumulator acc (sx*sx*3 doubles)."""
    lib = C.CDLL(HOST_LIB)
    lib.dataOutput.argtypes = [C.POINTER(C.c_double), C.c_int, C.c_char_p]
    out = str(tmp_path / f"{tag}.ppm")
    acc = np.ascontiguousarray(acc, dtype=np.float64).copy()
    lib.dataOutput(acc.ctypes.data_as(C.POINTER(C.c_double)), sx, out.encode())
    data = open(out, "rb").read()
    header = f"P6\n# Output from PathTracer.c\n{sx} {sx}\n255\n".encode()
    assert data.startswith(header)
    return np.frombuffer(data[len(header):], dtype=np.uint8)


@pytest.fixture(scope="module")
def device():
    dev = pt.Device(0)
    dev.scene_upload(pt.Scene.from_file(os.path.join(REPO, "tests", "golden", "cornell.ptxscene")))
    yield dev
    dev.close()


@pytest.mark.parametrize("sx,spp", [(64, 1), (128, 40), (250, 7)])
def test_rendered_image_bytes_equal_host_output(device, sx, spp, tmp_path):
    prm = pt.make_params(sx, 6, 7, 0.0, seed=11)
    device.render_begin(prm)
    device.render_passes(0, spp)
    bits, info = device.tonemap()
    rgb, _ = device.download()
    want = host_data_output(rgb, sx, tmp_path, f"r{sx}")
    assert np.array_equal(bits, want)
    assert info["min"] == rgb.min() and info["max"] == rgb.max()
    d_rgb, _, rows = device.accumulators()  # the same through the entry point for any device buffer
    bits2, info2 = device.tonemap(d_rgb, rows * sx * 3)
    assert np.array_equal(bits2, want) and info2 == info
    assert 0.0 <= info["low_cut"] < info["high_cut"] <= 1.000001


@pytest.mark.parametrize("kind", ["uniform", "heavy_tail", "two_level", "mostly_black", "ramp", "with_nan"])
def test_synthetic_buffers_bytes_equal_host_output(device, kind, tmp_path):
    rng = np.random.default_rng(5)
    sx = 97
    n = sx * sx * 3
    if kind == "uniform":
        acc = rng.random(n) * 3.0
    elif kind == "heavy_tail":
        acc = rng.exponential(0.2, n) ** 2
    elif kind == "two_level":
        acc = np.where(rng.random(n) < 0.5, 0.25, 0.75) + rng.random(n) * 1e-3
    elif kind == "mostly_black":
        acc = np.where(rng.random(n) < 0.97, 0.0, rng.random(n) * 40.0)
    elif kind == "with_nan":
        # paths through degenerate materials leave NaN in a few accumulators (the reference's do too); they take no
        # part in minimum, maximum and histogram, and the x86 conversion of the reference makes their byte 0
        acc = rng.random(n) * 2.0
        acc[rng.integers(0, n, 50)] = np.nan
        acc[rng.integers(0, n, 5)] = -1e-5
    else:
        acc = np.linspace(0.0, 1.0, n)  # many values land exactly on bin edges and byte boundaries
    device.render_begin(pt.make_params(sx, 6, 7, 0.0, seed=1))
    device.upload(acc)
    back, _ = device.download()
    assert np.array_equal(back, acc, equal_nan=True)
    bits, info = device.tonemap()
    want = host_data_output(acc, sx, tmp_path, kind)
    assert np.array_equal(bits, want)
    assert info["min"] == np.nanmin(acc) and info["max"] == np.nanmax(acc)


def test_tonemap_at_benchmark_size_is_consistent(device):
    """1024x1024: the bytes depend on the accumulator only, so two calls agree, and a byte is the truncated
    gamma curve of its value between the reported cuts."""
    prm = pt.make_params(1024, 6, 7, 0.0, seed=3)
    device.render_begin(prm)
    device.render_passes(0, 2)
    a, info = device.tonemap()
    b, _ = device.tonemap()
    assert np.array_equal(a, b)
    rgb, _ = device.download()
    v = ((rgb - info["min"]) / (info["max"] - info["min"]) - info["low_cut"]) / (info["high_cut"] - info["low_cut"])
    approx = (255.0 * np.clip(v, 0, 1) ** 0.75).astype(np.uint8)
    assert np.abs(approx.astype(int) - a.astype(int)).max() <= 1 and (approx != a).mean() < 1e-6


def test_tonemap_needs_the_whole_image(device):
    prm = pt.make_params(64, 6, 7, 0.0, seed=1, rank=0, n_ranks=2, band_rows=8)
    device.render_begin(prm)
    with pytest.raises(pt.PtxError):
        device.tonemap()
