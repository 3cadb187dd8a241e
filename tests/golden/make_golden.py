#!/usr/bin/env python3
"""Regenerates the golden vectors under tests/golden/ from the compiled reference.

Needs the reference checkout at /root/reference and `make -C oracle ref`
(oracle/_ref/libptref_f*.so).  Everything written here is data: scene
descriptions dumped from the reference's own scene construction, rays, and the
values the reference's routines returned for them.  Run from the repository
root:  python tests/golden/make_golden.py
"""
import ctypes as C
import os
import shutil
import sys
import tempfile

import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))
REPO = os.path.dirname(os.path.dirname(HERE))
sys.path.insert(0, REPO)
sys.path.insert(0, os.path.join(REPO, "tests"))

from assets import synthetic_assets  # noqa: E402
from harness import REFERENCE_ROOT, Ref, RefCamSpec, RefObjSpec, RefXform, _p, random_rays  # noqa: E402
from pathtracer_amd import abi  # noqa: E402
from pathtracer_amd.scene import Scene  # noqa: E402

PI = 3.14159265354


def spec(typ, mat, col, refl_sig, r_index, ops, light=0, verts=None, tex="", nrm="", alpha=""):
    s = RefObjSpec()
    s.type, s.is_light, s.n_ops = typ, light, len(ops)
    s.diff, s.refl, s.tran = mat
    s.r, s.g, s.b = col
    s.refl_sig, s.r_index = refl_sig, r_index
    if verts is not None:
        s.verts[:] = verts
    for k, (op, a, b, c) in enumerate(ops):
        s.ops[k] = RefXform(op, 0, a, b, c)
    s.tex, s.nrm, s.alpha = tex.encode(), nrm.encode(), alpha.encode()
    return s


RX, RY, RZ, TR, SC = range(5)


def prims_scene():
    """Every primitive type, most under non-uniform scale and rotation, with textures, normal maps, an alpha map,
    refraction and two kinds of light."""
    T = "./Textures/"
    objs = [
        # tilted: with the ceiling normal exactly antiparallel to (0,1,0) the uniform-hemisphere rotation axis of
        # hemisphereReorient is pure rounding noise (utils_path.c:88-89)
        spec(abi.ROOM, (1, 0, 0), (.8, .8, .7), 0, 1, [(SC, 12, 9, 14), (RY, .3, 0, 0), (RX, .07, 0, 0), (TR, 0, 1, 3)]),
        spec(abi.PLANE, (.7, .3, 0), (1, 1, 1), .03, 1, [(RX, -PI / 2, 0, 0), (SC, 9, 1, 9), (TR, 0, -5, 3)],
             tex=T + "checker.ppm", nrm=T + "checker_normal.ppm"),
        spec(abi.SPHERE, (.3, .2, .5), (.9, .8, .7), .02, 1.45, [(SC, 2, 1.5, 2), (RZ, .4, 0, 0), (TR, -3, -2, 4)],
             tex=T + "granite.ppm"),
        spec(abi.CYL, (.6, .4, 0), (.6, .9, .6), .1, 1, [(SC, 1, 1.3, 2), (RX, 1.1, 0, 0), (RY, .5, 0, 0), (TR, 3, -2, 5)],
             tex=T + "woodfloor.ppm"),
        spec(abi.CUBE, (.5, .1, .4), (.9, .9, .9), .05, 1.3, [(SC, 1.5, 1, 1.2), (RY, .7, 0, 0), (RX, .2, 0, 0), (TR, 0, -3.5, 1)],
             tex=T + "granite.ppm", nrm=T + "granite_normal.ppm"),
        spec(abi.TRIANGLE, (.8, .2, 0), (.9, .3, .3), 0, 1, [], verts=[-6, -4, 8, 6, -4, 8, 0, 5, 9]),
        spec(abi.TRIANGLE, (0, 0, 1), (.95, .95, 1), 0, 1.2, [], verts=[-2, -1, 0, 2, -1.5, 0.5, 0, 2.5, -0.5]),
        spec(abi.PLANE, (1, 0, 0), (1, 1, 1), 0, 1, [(SC, 2, 3, 1), (RY, -.6, 0, 0), (TR, 4, 1, 2)],
             tex=T + "card_ace_spade.ppm", alpha=T + "blob_alpha.pgm"),
        spec(abi.SPHERE, (1, 0, 0), (1, .9, .8), 0, 1, [(SC, .8, .8, .8), (TR, -4, 5, 6)], light=1),
        spec(abi.PLANE, (1, 0, 0), (1, 1, 1), 0, 1, [(SC, 2, 1.5, 1), (RX, PI / 2, 0, 0), (TR, 2, 7.5, 3)], light=1),
        spec(abi.SPHERE, (0, 0, 1), (1, 1, 1), 0, 1.5, [(SC, 1.2, 1.2, 1.2), (TR, 1, 0, -2)]),
        spec(abi.CYL, (0, 1, 0), (.9, .9, .9), 0, 1, [(SC, .7, .7, 3), (TR, -6, 0, 7)]),
    ]
    cam = RefCamSpec()
    cam.e[:] = (0.5, 1.0, -9.0)
    cam.g[:] = (-0.5, -1.0, 12.0)
    cam.up[:] = (0, 1, 0)
    cam.f, cam.wl, cam.wt, cam.wsize, cam.focus = -3.0, -2.0, 2.0, 4.0, -12.0
    return objs, cam


def special_rays(scene, rng):
    """Rays that start on surfaces (self-hit logic), graze, start inside objects, or exclude the object they hit."""
    rays = random_rays(scene, 600, 77, with_os=False)
    n = len(rays)
    # axis-parallel directions (division by zero in the slab tests) and exact zeros
    for k in range(0, 60):
        d = np.zeros(3)
        d[k % 3] = 1.0 if k % 2 else -1.0
        rays["d"][k] = d
    rays["p0"][60:120] = rng.uniform(-1, 1, (60, 3)) * 0.5 + np.array([-3, -2, 4])  # inside the textured sphere
    rays["p0"][120:180] = rng.uniform(-1, 1, (60, 3)) * 0.4 + np.array([0, -3.5, 1])  # inside the cube
    rays["p0"][180:240] = rng.uniform(-1, 1, (60, 3)) * 0.3 + np.array([3, -2, 5])  # inside the cylinder
    return rays


def save_hits(path, rays, hits, **extra):
    np.savez_compressed(path, rays=rays, hits=hits, **extra)


def scene_to_npz(scene, path):
    np.savez_compressed(path, objects=scene.objects, nodes=scene.nodes, lights=scene.lights, images=scene.images,
                        texels=scene.texels, camera=scene.camera)


def mixed_flag_vectors(tmp):
    """Path radiances from the six builds of the reference with some but not all of IS / ES / RR
    (reference Makefile:6-16), on the Cornell box and on the scene with every primitive type."""
    rng = np.random.default_rng(606)
    out = {}
    objs, cam = prims_scene()
    for name, sx, depth, ap in (("cornell", 48, 6, 0.0), ("prims", 40, 8, 0.1)):
        n = 1200
        pi, pj, pk = rng.integers(0, sx, n), rng.integers(0, sx, n), rng.integers(0, 32, n)
        out.update({f"{name}_sx": sx, f"{name}_depth": depth, f"{name}_aperture": ap, f"{name}_seed": 515,
                    f"{name}_pi": pi, f"{name}_pj": pj, f"{name}_pk": pk})
        for f in range(1, 7):
            r = Ref(f)
            if name == "cornell":
                r.select_scene("cornell", depth)
            else:
                r.build_custom(objs, cam, depth, asset_dir=tmp)
            out[f"{name}_col_f{f}"] = r.trace_pixels(sx, pi, pj, pk, ap, 515)
    np.savez_compressed(os.path.join(HERE, "mixed_flags_paths.npz"), **out)


def main():
    tmp = tempfile.mkdtemp(prefix="ptx_golden_")
    synthetic_assets(tmp)
    if "--mixed-only" in sys.argv:
        mixed_flag_vectors(tmp)
        shutil.rmtree(tmp)
        return
    rng = np.random.default_rng(2025)
    refs = {f: Ref(f) for f in (0, 7)}
    ref = refs[7]
    scratch_scene = os.path.join(tmp, "scene.bin")

    # ---------------------------------------------------------------- Cornell (config 1 and 2 of the baseline)
    for f in refs:
        refs[f].select_scene("cornell", 6)
    cornell = ref.dump_scene(os.path.join(HERE, "cornell.ptxscene"))
    rays = random_rays(cornell, 4096 + 4096, 11)
    save_hits(os.path.join(HERE, "cornell_firsthit.npz"), rays, ref.first_hit(rays))
    for f in refs:
        sx = 48
        n = 3000
        pi, pj, pk = rng.integers(0, sx, n), rng.integers(0, sx, n), rng.integers(0, 64, n)
        col = refs[f].trace_pixels(sx, pi, pj, pk, 0.0, 4242)
        col_dof = refs[f].trace_pixels(sx, pi, pj, pk, 0.2, 4242)
        rgb_c, w_c = refs[f].render(32, 0, 6, 0.0, counter_seed=31)
        rgb_l, w_l = refs[f].render(32, 0, 6, 0.0, lcg_seed=1234, threads=1)
        np.savez_compressed(os.path.join(HERE, f"cornell_paths_f{f}.npz"), sx=sx, depth=6, seed=4242, pi=pi, pj=pj, pk=pk,
                            col=col, col_dof=col_dof, aperture_dof=0.2, render_sx=32, render_spp=6, render_seed=31,
                            rgb_counter=rgb_c, w_counter=w_c, lcg_seed=1234, rgb_lcg=rgb_l, w_lcg=w_l)

    # ---------------------------------------------------------------- every primitive, textures, alpha, lights
    objs, cam = prims_scene()
    for f in refs:
        refs[f].build_custom(objs, cam, 8, asset_dir=tmp)
    prims = ref.dump_scene(scratch_scene)
    scene_to_npz(prims, os.path.join(HERE, "prims_scene.npz"))
    rays = np.concatenate([random_rays(prims, 6000, 5), special_rays(prims, rng)])
    hits = ref.first_hit(rays)
    # rays leaving from hit points with Os set to the object hit (the reference's self-hit exclusion)
    ok = hits["obj"] >= 0
    bounce = np.zeros(ok.sum(), dtype=abi.RAY_DTYPE)
    bounce["p0"] = hits["p"][ok]
    d = rng.normal(size=(len(bounce), 3))
    bounce["d"] = d / np.linalg.norm(d, axis=1, keepdims=True)
    bounce["os"] = hits["obj"][ok]
    bounce["os"][::3] = -1  # and without exclusion: only TOL guards the origin surface
    rays = np.concatenate([rays, bounce])
    save_hits(os.path.join(HERE, "prims_firsthit.npz"), rays, ref.first_hit(rays))
    for f in refs:
        sx = 40
        n = 2500
        pi, pj, pk = rng.integers(0, sx, n), rng.integers(0, sx, n), rng.integers(0, 32, n)
        col = refs[f].trace_pixels(sx, pi, pj, pk, 0.1, 77)
        rgb_c, w_c = refs[f].render(24, 0, 5, 0.1, counter_seed=5)
        np.savez_compressed(os.path.join(HERE, f"prims_paths_f{f}.npz"), sx=sx, depth=8, seed=77, aperture=0.1, pi=pi, pj=pj,
                            pk=pk, col=col, render_sx=24, render_spp=5, render_seed=5, rgb_counter=rgb_c, w_counter=w_c)

    # known-answer vectors for the samplers and look-ups, random numbers scripted
    lib = ref.lib
    kat = {}
    m = 64
    normals = rng.normal(size=(m, 3))
    normals /= np.linalg.norm(normals, axis=1, keepdims=True)
    normals[0] = (0, 0, 1); normals[1] = (0, 0, -1); normals[2] = (0, 1, 0); normals[3] = (1, 0, 0); normals[4] = (0, -1, 0)
    script = rng.random((m, 8))
    out = np.zeros((m, 3))
    for k in range(m):
        lib.ref_rng_script(_p(script[k]), 8)
        lib.ref_kat_cos_weighted(_p(normals[k]), _p(out[k]))
    kat["cosw_n"], kat["cosw_script"], kat["cosw_out"] = normals, script, out.copy()
    vs = rng.normal(size=(m, 3))
    vs /= np.linalg.norm(vs, axis=1, keepdims=True)
    out = np.zeros((m, 3))
    for k in range(5, m):  # (0,1,0) x n is degenerate for n = +-y; start past the axis cases
        lib.ref_kat_hemisphere_reorient(_p(normals[k]), _p(vs[k]), _p(out[k]))
    kat["hemi_n"], kat["hemi_v"], kat["hemi_out"] = normals[5:], vs[5:], out[5:].copy()
    rn = np.zeros(m)
    for k in range(m):
        lib.ref_rng_script(_p(script[k]), 8)
        rn[k] = lib.ref_kat_random_normal()
    kat["randn_script"], kat["randn_out"] = script, rn
    # refraction / reflection against the glass sphere (flat index looked up by type and r_index)
    glass = int(np.where((prims.objects["type"] == abi.SPHERE) & (prims.objects["r_index"] == 1.5))[0][0])
    d_in = rng.normal(size=(m, 3)); d_in /= np.linalg.norm(d_in, axis=1, keepdims=True)
    stacks = np.zeros((m, 10)); stacks[:, 0] = 1.0; stacks[:, 1] = rng.choice([1.0, 1.33, 1.5], m)
    idx_in = rng.integers(0, 2, m).astype(np.int32)
    d_out = np.zeros((m, 3)); n_out = np.zeros((m, 3)); stacks_out = stacks.copy(); idx_out = idx_in.copy()
    pzero = np.zeros(3)
    for k in range(m):
        lib.ref_rng_script(_p(script[k]), 8)
        ii = C.c_int(int(idx_in[k]))
        lib.ref_kat_refract(glass, _p(d_in[k]), _p(pzero), _p(normals[k]), _p(stacks_out[k]), C.byref(ii), _p(d_out[k]), _p(n_out[k]))
        idx_out[k] = ii.value
    kat.update(refr_obj=glass, refr_d=d_in, refr_n=normals, refr_stack=stacks, refr_idx=idx_in, refr_script=script,
               refr_d_out=d_out.copy(), refr_n_out=n_out, refr_stack_out=stacks_out, refr_idx_out=idx_out)
    for k in range(m):
        lib.ref_kat_reflect(_p(d_in[k]), _p(pzero), _p(normals[k]), _p(d_out[k]))
    kat["refl_d_out"] = d_out.copy()
    # texture, normal map, alpha look-ups: interior, edges, and coordinates outside [0,1) whose texel indices still
    # fall inside the image buffer (the reference reads whatever memory lies beyond it otherwise)
    ab = rng.random((m, 2)); ab[0] = (0, 0); ab[1] = (1, 1); ab[2] = (0.5, 1.0); ab[3] = (-0.2, 0.4); ab[4] = (0.999, 0.999)
    ab[5] = (1.2, 0.5)
    textured = int(np.where((prims.objects["tex_img"] >= 0) & (prims.objects["normal_img"] >= 0))[0][0])
    alpha_obj = int(np.where(prims.objects["alpha_img"] >= 0)[0][0])
    tex = np.zeros((m, 3)); nrm = np.zeros((m, 3)); alp = np.zeros((m, 3)); nd = np.zeros((m, 3))
    for k in range(m):
        lib.ref_kat_texmap(textured, 0, ab[k, 0], ab[k, 1], _p(tex[k]))
        lib.ref_kat_texmap(textured, 1, ab[k, 0], ab[k, 1], _p(nrm[k]))
        lib.ref_kat_texmap(alpha_obj, 2, ab[k, 0], ab[k, 1], _p(alp[k]))
        lib.ref_kat_normal_deform(textured, ab[k, 0], ab[k, 1], _p(normals[k]), _p(nd[k]))
    kat.update(tex_obj=textured, alpha_obj=alpha_obj, tex_ab=ab, tex_out=tex, nrm_out=nrm, alpha_out=alp[:, 0],
               ndef_n=normals, ndef_out=nd)
    # primary rays with and without aperture, light point samples
    pix = rng.integers(0, 40, (m, 2)).astype(np.int32)
    p0 = np.zeros((m, 3)); dd = np.zeros((m, 3)); p0a = np.zeros((m, 3)); dda = np.zeros((m, 3))
    for k in range(m):
        lib.ref_rng_script(_p(script[k]), 8)
        lib.ref_kat_primary_ray(40, int(pix[k, 0]), int(pix[k, 1]), 0.0, _p(p0[k]), _p(dd[k]))
        lib.ref_rng_script(_p(script[k]), 8)
        lib.ref_kat_primary_ray(40, int(pix[k, 0]), int(pix[k, 1]), 0.25, _p(p0a[k]), _p(dda[k]))
    kat.update(prim_pix=pix, prim_script=script, prim_p0=p0, prim_d=dd, prim_p0_ap=p0a, prim_d_ap=dda, prim_sx=40, prim_ap=0.25)
    lights = prims.lights
    ls = np.zeros((len(lights), m, 3))
    for li, oi in enumerate(lights):
        for k in range(m):
            lib.ref_rng_script(_p(script[k]), 8)
            lib.ref_kat_light_sample(int(oi), _p(ls[li, k]))
    kat.update(light_objs=np.asarray(lights), light_script=script, light_out=ls)
    np.savez_compressed(os.path.join(HERE, "prims_kat.npz"), **kat)

    # ---------------------------------------------------------------- SciFiCube and BubbleTea, synthetic textures
    for name, depth, ap in (("scifi", 12, 0.05), ("bubbletea", 12, 0.15)):
        for f in refs:
            refs[f].select_scene(name, depth, asset_dir=tmp)
        sc = ref.dump_scene(scratch_scene)
        scene_to_npz(sc, os.path.join(HERE, f"{name}_scene.npz"))
        rays = random_rays(sc, 6000, 21)
        save_hits(os.path.join(HERE, f"{name}_firsthit.npz"), rays, ref.first_hit(rays), n_inserted=ref.lib.ref_num_inserted())
        sx = 32
        n = 1500
        pi, pj, pk = rng.integers(0, sx, n), rng.integers(0, sx, n), rng.integers(0, 16, n)
        col = ref.trace_pixels(sx, pi, pj, pk, ap, 9)
        np.savez_compressed(os.path.join(HERE, f"{name}_paths_f7.npz"), sx=sx, depth=depth, seed=9, aperture=ap, pi=pi, pj=pj,
                            pk=pk, col=col)
    mixed_flag_vectors(tmp)
    shutil.rmtree(tmp)
    for fn in sorted(os.listdir(HERE)):
        print(f"{os.path.getsize(os.path.join(HERE, fn)):9d}  {fn}")


if __name__ == "__main__":
    main()
