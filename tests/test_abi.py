"""The C-ABI library loads and exports every function include/ptx_hip.h declares (no GPU needed)."""
import ctypes
import os
import re

import pytest

REPO = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def declared_functions():
    text = open(os.path.join(REPO, "include", "ptx_hip.h")).read()
    text = re.sub(r"/\*.*?\*/", "", text, flags=re.S)
    return sorted(set(re.findall(r"\b(ptx_[a-z_]+)\s*\(", text)))


def test_header_and_binding_agree():
    from pathtracer_amd.device import EXPORTS

    assert declared_functions() == sorted(EXPORTS)


def test_library_exports_every_declared_symbol():
    from pathtracer_amd.device import LIB_PATH

    assert os.path.exists(LIB_PATH), "libptx_hip.so is not built (make -C pathtracer_amd)"
    lib = ctypes.CDLL(LIB_PATH)
    for name in declared_functions():
        assert hasattr(lib, name), name
    lib.ptx_version.restype = ctypes.c_char_p
    assert lib.ptx_version().decode().count(".") == 1


def test_missing_library_fails_loudly(tmp_path):
    from pathtracer_amd.device import load_library

    with pytest.raises(FileNotFoundError):
        load_library(str(tmp_path / "libptx_hip.so"))


def test_struct_layouts_match_the_c_headers(tmp_path):
    """Compiles a tiny C program against include/ptx_scene.h and compares sizeof with the numpy/ctypes mirrors."""
    import subprocess

    from pathtracer_amd import abi

    src = tmp_path / "sz.c"
    src.write_text(
        '#include <stdio.h>\n#include "ptx_scene.h"\n#include "ptx_hip.h"\nint main(){printf("%zu %zu %zu %zu %zu %zu %zu %zu %zu\\n",'
        "sizeof(ptx_object),sizeof(ptx_node),sizeof(ptx_image),sizeof(ptx_camera),sizeof(ptx_ray),sizeof(ptx_hit),"
        "sizeof(ptx_scene_desc),sizeof(ptx_render_params),sizeof(ptx_stats));return 0;}\n"
    )
    exe = tmp_path / "sz"
    subprocess.run(["gcc", "-I", os.path.join(REPO, "include"), str(src), "-o", str(exe)], check=True)
    sizes = list(map(int, subprocess.run([str(exe)], check=True, capture_output=True, text=True).stdout.split()))
    assert sizes == [abi.OBJECT_DTYPE.itemsize, abi.NODE_DTYPE.itemsize, abi.IMAGE_DTYPE.itemsize, abi.CAMERA_DTYPE.itemsize,
                     abi.RAY_DTYPE.itemsize, abi.HIT_DTYPE.itemsize, ctypes.sizeof(abi.SceneDesc),
                     ctypes.sizeof(abi.RenderParams), ctypes.sizeof(abi.Stats)]


def test_row_band_helpers():
    from pathtracer_amd import abi

    for sx, world, band in [(1024, 8, 8), (100, 3, 8), (20, 3, 4), (7, 2, 1), (64, 1, 8)]:
        rows = [abi.global_rows(sx, r, world, band) for r in range(world)]
        assert sorted(int(x) for r in rows for x in r) == list(range(sx))
        for r in range(world):
            assert len(rows[r]) == abi.rows_owned(sx, r, world, band)


def test_benchmark_kernel_stays_within_its_register_budget():
    """k_trace_samples<7, false, true> (small scene, zero aperture, IS+ES+RR) is compiled for four waves per SIMD:
    128 vector registers and at most 80 bytes of scratch per lane.  It gets there only with machine-level
    loop-invariant code motion switched off (pathtracer_amd/Makefile); with the pass on, its sixty double-precision
    literals land in 500 bytes of scratch per lane and the benchmark loses a quarter.  The code object says which."""
    import subprocess
    import sys
    repo = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    obj = os.path.join(repo, "pathtracer_amd", "build", "ptx_hip.o")
    if not os.path.exists(obj):
        pytest.skip("object file not built here")
    r = subprocess.run([sys.executable, os.path.join(repo, "tools", "kernel_regs.py"), "--check", obj], capture_output=True, text=True)
    if r.returncode == 3:
        pytest.skip("the code object could not be read with the tools at hand: " + r.stderr.strip()[-200:])
    assert r.returncode == 0, r.stderr

