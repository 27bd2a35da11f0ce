"""GPU parity over a sweep of generated scenes: other generator seeds (sphere positions, radii, materials, which solids
are glass or quads), lattices, path depths, lens radii, samples per pixel and per-pixel RNG seeds than the benchmark
configurations fix.  Every case goes through the scan of every record and through the grid walk; bar as everywhere:
radiance, camera-ray hit ids and ray counts equal to the oracle's bit for bit."""
import pytest

import pyoracle as po
from helpers import bits_equal
from raytracer_amd import api, synth

pytestmark = pytest.mark.gpu

CASES = [
    # config, generator seed, lattice, depth, spp, lens, frame seed, width
    ("c3", 1, (6, 5, 4), 3, 1, 0.0, 0, 144),
    ("c3", 2, (9, 3, 7), 6, 2, 2.5, 7, 112),
    ("c3", 3, (2, 2, 2), 1, 4, 0.3, 123456789, 160),
    ("c3", 4, (12, 10, 3), 5, 1, 1.0, 1, 96),
    ("c5", 5, (8, 6, 8), 4, 2, 1.0, 99, 96),
    ("c5", 6, (5, 9, 4), 2, 3, 0.0, 3, 120),
    ("c5", 7, (14, 4, 6), 7, 1, 4.0, 2**31, 80),
    ("c2", 8, (3, 3, 5), 2, 2, 0.5, 42, 200),
]


@pytest.mark.parametrize("cfg,seed,lattice,depth,spp,lens,frame_seed,width", CASES, ids=[f"{c[0]}-seed{c[1]}" for c in CASES])
def test_generated_scene_matches_oracle_in_both_modes(renderer, cfg, seed, lattice, depth, spp, lens, frame_seed, width):
    _, sc = synth.make(cfg, width=width, spp=spp, depth=depth, lattice=lattice, seed=seed, lens=lens, frame_seed=frame_seed)
    o = po.render(sc, policy=po.POLICY_DEV, rng_mode=po.RNG_PIXEL, want_hits=True)
    want_rays = int(o["rays"].sum())
    try:
        for accel in (0, 1):
            renderer.set_option(api.OPT_ACCEL, accel)
            renderer.upload(sc)
            g = renderer.render(want_hits=True)
            assert g["rays"] == want_rays, f"accel {accel}: ray count {g['rays']} != oracle {want_rays}"
            assert (g["hits"] == o["hits"]).all(), f"accel {accel}: hit ids differ"
            eq = bits_equal(g["rgb"], o["rgb"])
            assert eq.all(), f"accel {accel}: {int((~eq).sum())} of {eq.size} radiance words differ"
    finally:
        renderer.set_option(api.OPT_ACCEL, 0)
