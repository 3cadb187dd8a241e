#!/usr/bin/env python3
"""Renders the benchmark configurations with the compiled reference and stores compact summaries of the raw
accumulators under tests/golden/ (data only: block means and a few whole rows of the images the reference's
own PathTrace produced).

  cornell_1024_1000spp_f7.npz  Cornell box 1024 x 1024, depth 6, 1000 samples per pixel, IS+ES+RR, aperture 0
                               (reference README.md:24 `1024 6 1000`): about 40 minutes on 8 cores
  cornell_256_64spp_f0.npz     Cornell box 256 x 256, depth 6, 64 samples, default build (no IS/ES/RR)
  cornell_1024_1000spp_f7_seed32.npz  (argument `floor`) the first render again with another seed
  scifi_256_128spp_f7.npz, bubbletea_256_128spp_f7.npz  (argument `hierarchy`) the two hierarchy scenes of the baseline
                               with the synthetic textures of tests/assets.py (the scenes of scifi_scene.npz and
                               bubbletea_scene.npz), 256 x 256, depth 12, 128 samples, their apertures: a minute each

Needs /root/reference and `make -C oracle ref`.  Run from the repository root:
    python tests/golden/make_golden_big.py [threads]
The random numbers are the counter stream (seed stored in the file), so a render with the same seed on any other
implementation of the same stream can be compared pixel by pixel.
"""
import os
import sys
import time

import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))
REPO = os.path.dirname(os.path.dirname(HERE))
sys.path.insert(0, REPO)
sys.path.insert(0, os.path.join(REPO, "tests"))

from harness import Ref  # noqa: E402


def block_means(img, b):
    """img (sx, sx, c) -> (sx/b, sx/b, c) means over b x b blocks"""
    sx = img.shape[0]
    return img.reshape(sx // b, b, sx // b, b, -1).mean(axis=(1, 3))


def nan_block_means(img, b):
    sx = img.shape[0]
    return np.nanmean(img.reshape(sx // b, b, sx // b, b, -1), axis=(1, 3))


def summarise(rgb, wght, sx, rows):
    rgb = rgb.reshape(sx, sx, 3)
    wght = wght.reshape(sx, sx, 1)
    # (a pixel whose weight has turned negative is NaN in the reference from then on: block means skip such pixels)
    return dict(blocks8=nan_block_means(rgb, 8),
                wblocks8=block_means(wght, 8)[..., 0], rows=np.asarray(rows),
                rgb_rows=rgb[rows].copy(), w_rows=wght[rows, :, 0].copy(), mean=np.nanmean(rgb, axis=(0, 1)),
                n_nan=int(np.isnan(rgb).sum()), nan_pixels=np.argwhere(np.isnan(rgb).any(axis=2)))


def render(flags, sx, depth, spp, seed, threads, chunk, out, rows, scene="cornell", aperture=0.0, asset_dir=None):
    ref = Ref(flags)
    ref.select_scene(scene, depth, asset_dir=asset_dir)
    rgb = np.zeros(sx * sx * 3)
    wght = np.ones(sx * sx)
    t0 = time.time()
    for k0 in range(0, spp, chunk):
        nk = min(chunk, spp - k0)
        ref.lib.ref_rng_counter(seed)
        ref.lib.ref_render(sx, k0, nk, aperture, rgb.ctypes.data_as(ref.lib.ref_render.argtypes[4]),
                           wght.ctypes.data_as(ref.lib.ref_render.argtypes[5]), threads)
        print(f"  flags {flags} {sx}x{sx}: {k0 + nk}/{spp} passes, {time.time() - t0:.0f} s", flush=True)
    np.savez_compressed(out, sx=sx, depth=depth, spp=spp, flags=flags, seed=seed, aperture=aperture,
                        **summarise(rgb, wght, sx, rows))
    print(f"wrote {out} ({os.path.getsize(out)} bytes), {sx * sx * spp / (time.time() - t0):.0f} paths/s", flush=True)


def main():
    threads = int(sys.argv[1]) if len(sys.argv) > 1 and sys.argv[1].isdigit() else 0
    if "hierarchy" in sys.argv:
        import shutil
        import tempfile

        from assets import synthetic_assets

        tmp = tempfile.mkdtemp(prefix="ptx_golden_")
        synthetic_assets(tmp)
        for name, ap in (("scifi", 0.05), ("bubbletea", 0.15)):
            render(7, 256, 12, 128, 31, threads, 16, os.path.join(HERE, f"{name}_256_128spp_f7.npz"), [20, 100, 128, 230], scene=name,
                   aperture=ap, asset_dir=tmp)
        shutil.rmtree(tmp)
        return
    if "floor" in sys.argv:
        # the same render with another seed: how far two reference renders lie apart (the noise floor the image
        # statistic is read against)
        render(7, 1024, 6, 1000, 32, threads, 20, os.path.join(HERE, "cornell_1024_1000spp_f7_seed32.npz"), [37, 400, 512, 900])
        return
    render(0, 256, 6, 64, 31, threads, 16, os.path.join(HERE, "cornell_256_64spp_f0.npz"), [5, 64, 128, 200])
    render(7, 1024, 6, 1000, 31, threads, 20, os.path.join(HERE, "cornell_1024_1000spp_f7.npz"), [37, 400, 512, 900])


if __name__ == "__main__":
    main()
