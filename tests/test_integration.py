"""The reference program with its sample loop swapped for the GPU library (integration/ref_flatten.c and
integration/ref_gpu_loop.c compiled against the reference's own headers, linked with the reference's own objects and
libptx_hip.so by oracle/Makefile into oracle/_ref/ref_gpu_cli_f7).  Without a GPU: that it builds and links.  With
one: that the reference's scene construction + BoxTree + dataOutput around the GPU loop give the files the
library's own render gives."""
import os
import subprocess

import numpy as np
import pytest

import pathtracer_amd as pt

REPO = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
BIN = os.path.join(REPO, "oracle", "_ref", "ref_gpu_cli_f7")


@pytest.mark.skipif(not os.path.isdir("/root/reference"), reason="needs the reference checkout to compile against")
def test_reference_with_gpu_loop_compiles_and_links():
    subprocess.run(["make", "-C", os.path.join(REPO, "oracle"), "ref"], check=True, capture_output=True)
    assert os.path.exists(BIN)
    deps = subprocess.run(["ldd", BIN], check=True, capture_output=True, text=True).stdout
    assert "libptx_hip.so" in deps and "libptref_f7.so" in deps and "not found" not in deps
    # the GPU loop and the flattening are in the binary / the harness library, the CPU sample loop is not called
    syms = subprocess.run(["nm", "-D", "--undefined-only", BIN], check=True, capture_output=True, text=True).stdout
    for name in ("ptx_init", "ptx_scene_upload", "ptx_render_passes", "ptx_download", "dataOutput", "ref_select_scene"):
        assert name in syms, name
    assert "PathTrace" not in syms


@pytest.mark.gpu
@pytest.mark.skipif(not os.path.exists(BIN), reason="oracle/_ref/ref_gpu_cli_f7 was not built (needs the reference checkout)")
def test_reference_scene_code_around_the_gpu_loop(tmp_path, scenes):
    sx, depth, spp, seed = 40, 6, 30, 5
    out = str(tmp_path / "ref_gpu.ppm")
    r = subprocess.run([BIN, "cornell", str(sx), str(depth), str(spp), out], env=dict(os.environ, PTX_SEED=str(seed)),
                       capture_output=True, text=True, timeout=100, cwd=str(tmp_path))
    assert r.returncode == 0, r.stderr[-2000:]
    assert f"Total number of rays created: {sx * sx * spp}" in r.stderr
    raw = open(out + ".pfm", "rb").read()
    header = f"PF\n{sx} {sx}\n-1.0\n".encode()
    assert raw.startswith(header)
    acc = np.frombuffer(raw[len(header):], dtype=np.float64)
    dev = pt.Device(0)
    dev.scene_upload(scenes["cornell"])
    dev.render_begin(pt.make_params(sx, depth, 7, 0.0, seed=seed))
    dev.render_passes(0, spp)
    rgb, _ = dev.download()
    assert np.array_equal(acc, rgb.reshape(-1))
    bits, _ = dev.tonemap()
    ppm = open(out, "rb").read()  # written by the reference's own dataOutput
    assert ppm.endswith(bits.tobytes()) and ppm.startswith(b"P6\n")
    dev.close()
