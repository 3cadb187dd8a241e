"""Live, bit-for-bit comparison of the CPU oracle with the compiled reference (oracle/_ref).

Runs only where the reference checkout and its harness build exist; elsewhere the golden vectors of
test_oracle_golden.py carry the same pin.
"""
import os
import tempfile

import numpy as np
import pytest

import pathtracer_amd as pt
from checks import compare_hits
from harness import Oracle, Ref, random_rays, ref_available

pytestmark = pytest.mark.skipif(not all(ref_available(f) for f in range(8)), reason="compiled reference not available")

CASES = [("cornell", 6, 0.0), ("scifi", 12, 0.05), ("bubbletea", 12, 0.15)]


@pytest.fixture(scope="module")
def tmpdir():
    with tempfile.TemporaryDirectory() as d:
        yield d


@pytest.mark.parametrize("name,depth,aperture", CASES)
@pytest.mark.parametrize("flags", list(range(8)))  # every build of the reference Makefile: IS | ES << 1 | RR << 2
def test_first_hit_and_render_bit_exact(name, depth, aperture, flags, tmpdir):
    ref = Ref(flags)
    ref.select_scene(name, depth)  # shipped scene with whatever assets the checkout holds
    scene = ref.dump_scene(os.path.join(tmpdir, f"{name}.bin"))
    orc = Oracle(scene)
    rays = random_rays(scene, 3000, 8)
    compare_hits(orc.first_hit(rays), ref.first_hit(rays), scene, exact=True)
    sx, spp = 20, 3
    prm = pt.make_params(sx, depth, flags, aperture, seed=17)
    a, wa = ref.render(sx, 0, spp, aperture, counter_seed=17)
    b, wb, _ = orc.render(prm, 0, spp)
    assert np.array_equal(a, b) and np.array_equal(wa, wb)
    prm.rng_mode = pt.abi.RNG_LCG48
    a, wa = ref.render(sx, 0, spp, aperture, lcg_seed=5, threads=1)
    b, wb, _ = orc.render(prm, 0, spp, threads=1, lcg_seed=5)
    assert np.array_equal(a, b) and np.array_equal(wa, wb)


def test_harness_loop_equals_reference_main(tmpdir):
    """ref_render() repeats the pixel loop of the reference's main(); check it against main() itself (run as a child
    process on the scene main() is compiled with), so that renders taken through the harness are reference renders."""
    import subprocess

    from harness import REF_DIR, REFERENCE_ROOT

    out = os.path.join(tmpdir, "bt.ppm")
    env = dict(os.environ, OMP_NUM_THREADS="1")
    subprocess.run([os.path.join(REF_DIR, "ref_cli_f7"), "77", "16", "5", "2", out, "0.15"], cwd=REFERENCE_ROOT, env=env,
                   check=True, capture_output=True)
    with open(out + ".pfm", "rb") as f:
        assert f.readline() == b"PF\n" and f.readline() == b"16 16\n" and f.readline() == b"-1.0\n"
        main_rgb = np.frombuffer(f.read(), dtype=np.float64)
    ref = Ref(7)
    ref.select_scene("bubbletea", 5)
    rgb, _ = ref.render(16, 0, 2, 0.15, lcg_seed=77, threads=1)
    assert np.array_equal(rgb, main_rgb)
