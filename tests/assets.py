"""Deterministic stand-ins for the reference's asset files (its 1024x1024 textures are 3 MB each): small textures,
normal maps and alpha maps written as binary PPM/PGM, under the file names the shipped scenes ask for."""
import os
import shutil

import numpy as np

REFERENCE_ROOT = "/root/reference"


def write_pnm(path, arr):
    arr = np.ascontiguousarray(arr, dtype=np.uint8)
    h, w = arr.shape[:2]
    magic = b"P6" if arr.ndim == 3 else b"P5"
    with open(path, "wb") as f:
        f.write(magic + b"\n# synthetic test texture\n" + f"{w} {h}\n255\n".encode())
        f.write(arr.tobytes())


def synthetic_assets(root):
    """Small deterministic textures standing in for the reference's 1024^2 ones (3 MB each)."""
    rng = np.random.default_rng(1234)
    tex = os.path.join(root, "Textures")
    os.makedirs(tex, exist_ok=True)

    def smooth(h, w, c):
        base = rng.integers(0, 256, (h // 4 + 2, w // 4 + 2, c)).astype(np.float64)
        up = np.kron(base, np.ones((4, 4, 1)))[:h, :w]
        yy, xx = np.mgrid[0:h, 0:w]
        return np.clip(0.6 * up + 50 * np.sin(xx / 5.0)[..., None] + 40 * np.cos(yy / 7.0)[..., None] + 40, 0, 255)

    def normal_map(h, w):
        n = smooth(h, w, 3)
        n[..., 2] = np.clip(170 + 0.3 * n[..., 2], 0, 255)
        n[..., 0] = np.clip(128 + 0.35 * (n[..., 0] - 128), 0, 255)
        n[..., 1] = np.clip(128 + 0.35 * (n[..., 1] - 128), 0, 255)
        return n

    names3 = ["granite", "woodfloor", "card_back_red", "card_ace_spade", "card_9_club", "card_queen_heart",
              "card_back_black", "card_4_diamond", "card_king_club", "card_8_diamond", "aluminium", "scifi_wall",
              "checker"]
    for k, nme in enumerate(names3):
        h, w = (48, 64) if k % 2 == 0 else (40, 56)
        write_pnm(os.path.join(tex, nme + ".ppm"), smooth(h, w, 3))
    for nme in ["granite_normal", "woodfloor_normal", "ceiling_normal", "aluminium_normal", "scifi_wall_normal",
                "checker_normal"]:
        write_pnm(os.path.join(tex, nme + ".ppm"), normal_map(48, 64))
    for nme in ["card_alpha", "blob_alpha"]:
        a = smooth(44, 36, 1)[..., 0]
        a[8:30, 6:28] = 255
        a[:4] = 0
        write_pnm(os.path.join(tex, nme + ".pgm"), a)
    os.makedirs(os.path.join(root, "objs"), exist_ok=True)
    teapot = os.path.join(REFERENCE_ROOT, "objs", "teapot.obj")
    if os.path.exists(teapot):  # a mesh is data the reference ships; without the checkout BubbleTea cannot be built
        shutil.copy(teapot, os.path.join(root, "objs", "teapot.obj"))
        return True
    return False


