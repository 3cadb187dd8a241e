"""ptx_render_passes_static, the render entry point that allocates nothing (its sample buffer is a fixed array in
the library's device image), against ptx_render_passes: same accumulators bit for bit, on two calls in a row, the
second on a smaller image, and a clear error beyond the fixed buffer's size."""
import numpy as np
import pytest

import pathtracer_amd as pt

pytestmark = pytest.mark.gpu

ALL = pt.abi.USE_IS | pt.abi.USE_ES | pt.abi.USE_RR


def _render(dev, prm, spp, static):
    dev.render_begin(prm)
    if static:
        dev.render_passes_static(0, spp)
    else:
        dev.render_passes(0, spp)
    rgb, w = dev.download()
    return rgb, w, dev.stats()


def test_static_sample_buffer_equals_allocated_one(gpu):
    dev = gpu["cornell"]
    # 1024 x 1024 fills the fixed buffer with eight passes a round, so 20 passes take three rounds; then a smaller
    # image whose 20 passes fit one round and leave most of the buffer holding the first call's samples
    for sx, spp, ap in ((1024, 20, 0.0), (200, 20, 0.0), (96, 7, 0.2)):
        prm = pt.make_params(sx, 6, ALL, ap, seed=41)
        want = _render(dev, prm, spp, False)
        got = _render(dev, prm, spp, True)
        assert np.array_equal(got[0], want[0]) and np.array_equal(got[1], want[1]) and got[2] == want[2], sx


def test_static_sample_buffer_on_a_hierarchy_scene_and_two_contexts(gpu, scenes):
    prm = pt.make_params(256, 12, ALL, 0.05, seed=5)
    want = _render(gpu["scifi"], prm, 6, False)
    got = _render(gpu["scifi"], prm, 6, True)
    assert np.array_equal(got[0], want[0], equal_nan=True) and got[2] == want[2]
    # a second context on the same GPU shares the array (calls must not overlap; these do not)
    got = _render(gpu["bubbletea"], pt.make_params(128, 12, ALL, 0.15, seed=5), 5, True)
    want = _render(gpu["bubbletea"], pt.make_params(128, 12, ALL, 0.15, seed=5), 5, False)
    assert np.array_equal(got[0], want[0], equal_nan=True) and got[2] == want[2]


def test_static_sample_buffer_rejects_images_beyond_its_size(gpu):
    dev = gpu["cornell"]
    lib = dev.lib
    assert lib.ptx_static_max_pixels() == 2 * 1024 * 1024
    dev.render_begin(pt.make_params(2048, 6, ALL, 0.0, seed=1))
    rc = lib.ptx_render_passes_static(dev.ctx, 0, 4)
    assert rc == -1 and b"2097152 pixels" in lib.ptx_last_error()
    # half of it, as one of two ranks, fits
    dev.render_begin(pt.make_params(2048, 6, ALL, 0.0, seed=1, rank=1, n_ranks=2, band_rows=8))
    assert lib.ptx_render_passes_static(dev.ctx, 0, 4) == 0
    dev.sync()
    assert dev.stats()["paths"] == 2048 * 1024 * 4


def test_static_entry_point_captured_into_a_graph_and_replayed(gpu):
    """What the entry point is for: a caller under HIP graph capture.  The call is captured on a PyTorch stream (the
    library launches on the stream it is given), nothing runs during the capture, and two replays continue the
    render as two direct calls do.  The timing getters have nothing to report after such a call."""
    import torch

    from pathtracer_amd import parallel

    if len(parallel.hip_runtime_copies()) != 1:
        pytest.skip("the library and PyTorch use two copies of the HIP runtime: a PyTorch stream means nothing to the library")
    dev = gpu["cornell"]
    prm = pt.make_params(256, 6, ALL, 0.0, seed=77)
    dev.render_begin(prm)
    dev.render_passes_static(0, 8)
    dev.render_passes_static(0, 8)
    want_rgb, want_w = dev.download()
    import ctypes

    assert dev.lib.ptx_last_pass_ms(dev.ctx, ctypes.byref(ctypes.c_float())) == -3  # PTX_ESTATE

    stream = torch.cuda.Stream()
    try:
        dev.set_stream(stream.cuda_stream)
        dev.render_begin(prm)
        dev.sync()
        graph = torch.cuda.CUDAGraph()
        with torch.cuda.graph(graph, stream=stream):
            dev.render_passes_static(0, 8)
        assert dev.stats()["paths"] == 0  # captured, not run
        graph.replay()
        graph.replay()
        torch.cuda.synchronize()
        got_rgb, got_w = dev.download()
    finally:
        dev.set_stream(None)
    assert np.array_equal(got_rgb, want_rgb) and np.array_equal(got_w, want_w)
