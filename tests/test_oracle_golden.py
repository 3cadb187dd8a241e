"""The CPU oracle against the golden vectors the compiled reference produced (tests/golden/make_golden.py).

The vectors were made by the reference's own routines on the same scene descriptions, so agreement here pins the
oracle to the reference wherever this repository is checked out, with or without the reference sources.
"""
import ctypes as C
import os

import numpy as np
import pytest

import pathtracer_amd as pt
from checks import compare_hits
from harness import _p

SCENES = ["cornell", "prims", "scifi", "bubbletea"]
# identical libm, compiler flags and inputs give identical bits; the small slack only covers a libm that
# picks different code paths on another CPU model
PATH_ATOL = 1e-12


@pytest.mark.parametrize("name", SCENES)
def test_first_hit_matches_reference(name, scenes, oracles, golden_dir):
    z = np.load(os.path.join(golden_dir, f"{name}_firsthit.npz"))
    got = oracles[name].first_hit(z["rays"])
    compare_hits(got, z["hits"], scenes[name], exact=True)


def test_bubbletea_keeps_the_objects_the_reference_can_reach(scenes, golden_dir):
    # build_boxtree loses 15 of the 6358 inserted objects (magic.c:224-231); the flat scene holds the survivors
    z = np.load(os.path.join(golden_dir, "bubbletea_firsthit.npz"))
    assert int(z["n_inserted"]) == 6358
    assert len(scenes["bubbletea"].objects) == 6343
    ids = scenes["bubbletea"].objects["src_id"]
    assert len(np.unique(ids)) == len(ids) and ids.max() < 6358


@pytest.mark.parametrize("name,flags", [("cornell", 0), ("cornell", 7), ("prims", 0), ("prims", 7), ("scifi", 7),
                                        ("bubbletea", 7)])
def test_path_radiance_matches_reference(name, flags, oracles, golden_dir):
    z = np.load(os.path.join(golden_dir, f"{name}_paths_f{flags}.npz"))
    ap = float(z["aperture"]) if "aperture" in z else 0.0
    prm = pt.make_params(int(z["sx"]), int(z["depth"]), flags, ap, seed=int(z["seed"]))
    got = oracles[name].trace_pixels(prm, z["pi"], z["pj"], z["pk"])
    assert np.abs(got - z["col"]).max() <= PATH_ATOL
    if "col_dof" in z:
        prm = pt.make_params(int(z["sx"]), int(z["depth"]), flags, float(z["aperture_dof"]), seed=int(z["seed"]))
        got = oracles[name].trace_pixels(prm, z["pi"], z["pj"], z["pk"])
        assert np.abs(got - z["col_dof"]).max() <= PATH_ATOL


@pytest.mark.parametrize("flags", [1, 2, 3, 4, 5, 6])
@pytest.mark.parametrize("name", ["cornell", "prims"])
def test_path_radiance_matches_reference_mixed_switches(name, flags, oracles, golden_dir):
    """The six builds of the reference with some but not all of IS, ES and RR."""
    z = np.load(os.path.join(golden_dir, "mixed_flags_paths.npz"))
    prm = pt.make_params(int(z[f"{name}_sx"]), int(z[f"{name}_depth"]), flags, float(z[f"{name}_aperture"]), seed=int(z[f"{name}_seed"]))
    got = oracles[name].trace_pixels(prm, z[f"{name}_pi"], z[f"{name}_pj"], z[f"{name}_pk"])
    assert np.abs(got - z[f"{name}_col_f{flags}"]).max() <= PATH_ATOL


@pytest.mark.parametrize("name,flags", [("cornell", 0), ("cornell", 7), ("prims", 0), ("prims", 7)])
def test_render_matches_reference(name, flags, oracles, golden_dir):
    z = np.load(os.path.join(golden_dir, f"{name}_paths_f{flags}.npz"))
    ap = float(z["aperture"]) if "aperture" in z else 0.0
    sx, spp = int(z["render_sx"]), int(z["render_spp"])
    prm = pt.make_params(sx, int(z["depth"]), flags, ap, seed=int(z["render_seed"]))
    rgb, w, st = oracles[name].render(prm, 0, spp)
    assert np.abs(rgb - z["rgb_counter"]).max() <= PATH_ATOL
    assert np.abs(w - z["w_counter"]).max() <= PATH_ATOL
    assert st["paths"] == sx * sx * spp
    assert st["end_depth"] + st["end_miss"] + st["end_light"] + st["end_rr"] == st["paths"]
    if "rgb_lcg" in z:  # one global drand48-style stream consumed in pixel order, as OMP_NUM_THREADS=1 would
        prm.rng_mode = pt.abi.RNG_LCG48
        rgb, w, _ = oracles[name].render(prm, 0, spp, threads=1, lcg_seed=int(z["lcg_seed"]))
        assert np.abs(rgb - z["rgb_lcg"]).max() <= PATH_ATOL


def test_render_in_two_calls_equals_one_call(oracles):
    prm = pt.make_params(24, 6, 7, 0.0, seed=3)
    a, wa, _ = oracles["cornell"].render(prm, 0, 6)
    b, wb, _ = oracles["cornell"].render(prm, 0, 2)
    oracles["cornell"].render(prm, 2, 4, b, wb)
    assert np.array_equal(a, b) and np.array_equal(wa, wb)


def test_row_bands_partition_the_image(oracles):
    sx = 20
    whole = pt.make_params(sx, 6, 7, 0.0, seed=3)
    a, wa, _ = oracles["cornell"].render(whole, 0, 3)
    rgb, w = np.zeros(sx * sx * 3), np.ones(sx * sx)
    for r in range(3):
        prm = pt.make_params(sx, 6, 7, 0.0, seed=3, rank=r, n_ranks=3, band_rows=4)
        oracles["cornell"].render(prm, 0, 3, rgb, w)
    assert np.array_equal(a, rgb) and np.array_equal(wa, w)


def test_known_answers(oracles, golden_dir):
    z = np.load(os.path.join(golden_dir, "prims_kat.npz"))
    orc = oracles["prims"]
    lib, h = orc.lib, orc.h
    tol = 1e-13
    out = np.zeros(3)
    for k in range(len(z["cosw_n"])):
        lib.orc_kat_cos_weighted(_p(z["cosw_n"][k].copy()), _p(z["cosw_script"][k].copy()), _p(out))
        assert np.allclose(out, z["cosw_out"][k], rtol=0, atol=tol, equal_nan=True)
    for k in range(len(z["hemi_n"])):
        lib.orc_kat_hemisphere_reorient(_p(z["hemi_n"][k].copy()), _p(z["hemi_v"][k].copy()), _p(out))
        assert np.allclose(out, z["hemi_out"][k], rtol=0, atol=tol)
    for k in range(len(z["randn_out"])):
        assert abs(lib.orc_kat_random_normal(_p(z["randn_script"][k].copy())) - z["randn_out"][k]) <= tol
    n_out = np.zeros(3)
    for k in range(len(z["refr_d"])):
        stack = z["refr_stack"][k].copy()
        idx = C.c_int(int(z["refr_idx"][k]))
        lib.orc_kat_refract(h, int(z["refr_obj"]), _p(z["refr_d"][k].copy()), _p(np.zeros(3)), _p(z["refr_n"][k].copy()),
                            _p(stack), C.byref(idx), _p(z["refr_script"][k].copy()), _p(out), _p(n_out))
        assert np.allclose(out, z["refr_d_out"][k], rtol=0, atol=tol)
        assert np.allclose(stack, z["refr_stack_out"][k]) and idx.value == int(z["refr_idx_out"][k])
        lib.orc_kat_reflect(_p(z["refr_d"][k].copy()), _p(np.zeros(3)), _p(z["refr_n"][k].copy()), _p(out))
        assert np.allclose(out, z["refl_d_out"][k], rtol=0, atol=tol)
    for k in range(len(z["tex_ab"])):
        a, b = map(float, z["tex_ab"][k])
        lib.orc_kat_texmap(h, int(z["tex_obj"]), 0, a, b, _p(out))
        assert np.allclose(out, z["tex_out"][k], rtol=0, atol=tol)
        lib.orc_kat_texmap(h, int(z["tex_obj"]), 1, a, b, _p(out))
        assert np.allclose(out, z["nrm_out"][k], rtol=0, atol=tol)
        lib.orc_kat_texmap(h, int(z["alpha_obj"]), 2, a, b, _p(out))
        assert abs(out[0] - z["alpha_out"][k]) <= tol
        lib.orc_kat_normal_deform(h, int(z["tex_obj"]), a, b, _p(z["ndef_n"][k].copy()), _p(out))
        assert np.allclose(out, z["ndef_out"][k], rtol=0, atol=tol, equal_nan=True)
    p0 = np.zeros(3)
    for k in range(len(z["prim_pix"])):
        i, j = map(int, z["prim_pix"][k])
        lib.orc_kat_primary_ray(h, int(z["prim_sx"]), i, j, 0.0, _p(z["prim_script"][k].copy()), _p(p0), _p(out))
        assert np.allclose(p0, z["prim_p0"][k], rtol=0, atol=tol) and np.allclose(out, z["prim_d"][k], rtol=0, atol=tol)
        lib.orc_kat_primary_ray(h, int(z["prim_sx"]), i, j, float(z["prim_ap"]), _p(z["prim_script"][k].copy()), _p(p0), _p(out))
        assert np.allclose(p0, z["prim_p0_ap"][k], rtol=0, atol=tol) and np.allclose(out, z["prim_d_ap"][k], rtol=0, atol=tol)
    for li, oi in enumerate(z["light_objs"]):
        for k in range(len(z["light_script"])):
            lib.orc_kat_light_sample(h, int(oi), _p(z["light_script"][k].copy()), _p(out))
            assert np.allclose(out, z["light_out"][li, k], rtol=0, atol=tol)


def test_clean_mode_of_the_oracle(oracles):
    """PTX_CLEAN (the estimator without the reference's defects, behind a flag): the accumulator is the plain sum of
    the paths' colours, the weight counts samples, no path is negative, and the flag changes nothing unless set."""
    import pathtracer_amd as pt
    orc = oracles["cornell"]
    sx, spp = 24, 6
    prm = pt.make_params(sx, 6, 7 | pt.abi.CLEAN, 0.0, seed=4)
    rgb, w, st = orc.render(prm, 0, spp)
    assert np.array_equal(w, np.full(sx * sx, 1.0 + spp))
    jj, ii = np.meshgrid(np.arange(sx), np.arange(sx), indexing="ij")
    total = np.zeros((sx * sx, 3))
    for k in range(spp):
        total += orc.trace_pixels(prm, ii.ravel(), jj.ravel(), np.full(sx * sx, k)).reshape(-1, 3)
    assert np.allclose(rgb.reshape(-1, 3), total, rtol=1e-13, atol=0)
    assert rgb.min() >= 0
    # jittered primary rays: two samples of a pixel on a wall's edge see different things more often than centred ones
    plain = pt.make_params(sx, 6, 7, 0.0, seed=4)
    assert not np.array_equal(orc.render(plain, 0, 1)[0], orc.render(prm, 0, 1)[0])

