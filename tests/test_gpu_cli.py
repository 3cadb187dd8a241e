"""The command line of the reference, `./PathTracer size depth spp out [aperture]` (PathTracer.c:486-672), as built
by `make -C pathtracer_amd`: its raw accumulator file and its tone-mapped image against the same render made
through the Python binding."""
import os
import subprocess

import numpy as np
import pytest

import pathtracer_amd as pt

pytestmark = pytest.mark.gpu

REPO = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
CLI = os.path.join(REPO, "pathtracer_amd", "bin", "PathTracer")


def _flags_of_the_build():
    """The IS/ES/RR switches the binary was made with: bin/.config holds the -D__USE_* options of the build."""
    with open(os.path.join(REPO, "pathtracer_amd", "bin", ".config")) as f:
        made_with = f.read()
    return ((pt.abi.USE_IS if "__USE_IS" in made_with else 0) | (pt.abi.USE_ES if "__USE_ES" in made_with else 0) |
            (pt.abi.USE_RR if "__USE_RR" in made_with else 0))


@pytest.mark.skipif(not os.path.exists(CLI), reason="command-line binary not built")
def test_command_line_render_equals_library_render(tmp_path, scenes):
    sx, depth, spp, seed = 40, 6, 30, 5
    out = str(tmp_path / "cli.ppm")
    r = subprocess.run([CLI, str(sx), str(depth), str(spp), out], env=dict(os.environ, PTX_SEED=str(seed)),
                       capture_output=True, text=True, timeout=100)
    assert r.returncode == 0, r.stderr[-2000:]
    assert f"Total number of rays created: {sx * sx * spp}" in r.stderr
    raw = open(out + ".pfm", "rb").read()
    header = f"PF\n{sx} {sx}\n-1.0\n".encode()
    assert raw.startswith(header)
    acc = np.frombuffer(raw[len(header):], dtype=np.float64)
    assert acc.size == sx * sx * 3

    dev = pt.Device(0)
    dev.scene_upload(scenes["cornell"])
    dev.render_begin(pt.make_params(sx, depth, _flags_of_the_build(), 0.0, seed=seed))
    dev.render_passes(0, spp)
    rgb, _ = dev.download()
    assert np.array_equal(acc, rgb.reshape(-1))
    bits, _ = dev.tonemap()
    ppm = open(out, "rb").read()
    assert ppm.endswith(bits.tobytes()) and ppm.startswith(b"P6\n")
    dev.close()


@pytest.mark.skipif(not os.path.exists(CLI), reason="command-line binary not built")
@pytest.mark.parametrize("n_contexts,sx", [(3, 40), (8, 100), (5, 9)])
def test_command_line_with_several_gpu_contexts_writes_the_same_files(n_contexts, sx, tmp_path):
    """PTX_GPUS=N splits the rows over N contexts (on one device here: context r runs on device r modulo the device
    count), gathers them device to device for the tone mapping and downloads them for the raw file: both files must
    equal those of the single-context run byte for byte, uneven shares and a partial last band included."""
    outs = []
    for n in (1, n_contexts):
        out = str(tmp_path / f"cli{n}.ppm")
        r = subprocess.run([CLI, str(sx), "6", "27", out], env=dict(os.environ, PTX_SEED="7", PTX_GPUS=str(n)),
                           capture_output=True, text=True, timeout=100)
        assert r.returncode == 0 and f"Total number of rays created: {sx * sx * 27}" in r.stderr, r.stderr[-2000:]
        outs.append((open(out, "rb").read(), open(out + ".pfm", "rb").read()))
    assert outs[0][0] == outs[1][0] and outs[0][1] == outs[1][1]


def test_gather_of_row_bands_on_the_device(gpu):
    """ptx_gather through the binding: four contexts render the four ranks' bands of one image; the gathered
    device image, tone-mapped, equals the tone-mapped single-context render."""
    from pathtracer_amd import device

    sx, spp = 72, 5
    whole = gpu["cornell"]
    whole.render_begin(pt.make_params(sx, 6, 7, 0.0, seed=3))
    whole.render_passes(0, spp)
    want_bits, want_info = whole.tonemap()
    devs = []
    for r in range(4):
        d = pt.Device(0)
        d.scene_upload(whole_scene(gpu))
        d.render_begin(pt.make_params(sx, 6, 7, 0.0, seed=3, rank=r, n_ranks=4, band_rows=8))
        d.render_passes(0, spp)
        devs.append(d)
    d_rgb, d_w = device.gather(devs)
    bits, info = devs[0].tonemap(d_rgb, sx * sx * 3)
    assert np.array_equal(bits, want_bits) and info == want_info
    with pytest.raises(pt.PtxError):
        device.gather(devs[1:])  # not ranks 0..n-1 of n
    for d in devs:
        d.close()


def whole_scene(gpu):
    import pathtracer_amd as pt

    return pt.Scene.from_file(os.path.join(REPO, "tests", "golden", "cornell.ptxscene"))


@pytest.mark.skipif(not os.path.exists(CLI), reason="command-line binary not built")
def test_command_line_usage_message():
    r = subprocess.run([CLI], capture_output=True, text=True, timeout=30)
    assert r.returncode == 0  # the reference leaves with exit(0) on a usage error (PathTracer.c:511-520)
    assert "PathTracer" in r.stderr + r.stdout
