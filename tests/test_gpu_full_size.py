"""GPU tests at the full sizes of the benchmark configurations (C3 = 3840x2160 / 1031 solids / depth 4,
C4 = 7680x4320 / 4103 solids / depth 8, C5 = 3840x2160 / 16384 mixed solids / 16 samples per pixel).

The CPU oracle cannot render such frames in seconds, so these tests use what does not depend on size:
  * rows: the oracle renders a handful of rows of the full-size frame (same scene, same per-pixel seeds) and
    those rows of the GPU frame must match bit for bit (radiance, hit ids);
  * partition: the frame rendered as the row tiles of 2 (or 3) ranks, one call per rank, re-assembled, is
    bit-identical to the frame of one call, and the ray counters add up;
  * determinism: two renders give the same bits.
"""
import numpy as np
import pytest

import pyoracle as po
from helpers import bits_equal
from raytracer_amd import api, dist as D, synth

torch = pytest.importorskip("torch")
pytestmark = pytest.mark.gpu


def _render_device(renderer, sc, tile_rows, rank, world, want_hits=False):
    """One rank's row tiles through rt_render_rows_device; returns (rows, rgb[rows,W,3], hits|None, rays)."""
    rows = D.rows_of_rank(sc.height, tile_rows, rank, world).numpy()
    n = len(rows)
    rgb = torch.empty((max(n, 1), sc.width, 3), dtype=torch.float64, device="cuda:0")
    hits = torch.empty((max(n, 1), sc.width, sc.spp), dtype=torch.int32, device="cuda:0") if want_hits else None
    torch.cuda.synchronize()
    renderer.render_rows_device(rgb.data_ptr(), tile_rows, rank, world, hits.data_ptr() if want_hits else None,
                                torch.cuda.current_stream().cuda_stream)
    rays = renderer.finish()
    return rows, rgb[:n].cpu().numpy(), (hits[:n].cpu().numpy() if want_hits else None), rays


def _check_rows_against_oracle(sc, rows, rgb, hits, sample_rows):
    """Oracle (device arithmetic policy, per-pixel seeds) on single rows of the full-size frame."""
    pos = {int(r): i for i, r in enumerate(rows)}
    for y in sample_rows:
        o = po.render(sc, policy=po.POLICY_DEV, rng_mode=po.RNG_PIXEL, rows=(y, y + 1), want_hits=True, workers=1)
        i = pos[y]
        assert (hits[i] == o["hits"][y]).all(), f"row {y}: hit ids differ"
        eq = bits_equal(rgb[i], o["rgb"][y])
        assert eq.all(), f"row {y}: {int((~eq).sum())} radiance words differ"


@pytest.mark.timeout(300)
def test_c3_full_size_rows_partition_determinism(renderer):
    _, sc = synth.make("c3")
    assert (sc.width, sc.height) == (3840, 2160)
    renderer.upload(sc)
    rows, rgb, hits, rays = _render_device(renderer, sc, 8, 0, 1, want_hits=True)
    assert rays > 25_000_000
    _check_rows_against_oracle(sc, rows, rgb, hits, [0, 701, 1080, 1619, 2159])
    # determinism
    _, rgb2, hits2, rays2 = _render_device(renderer, sc, 8, 0, 1, want_hits=True)
    assert rays2 == rays and bits_equal(rgb, rgb2).all() and (hits == hits2).all()
    # partition over 3 ranks (2160 rows = 270 tiles of 8: ranks get 90 tiles each) and over 2 with ragged tiles of 7
    for tile, world in ((8, 3), (7, 2)):
        frame = np.empty_like(rgb)
        total = 0
        for rank in range(world):
            r, part, _, n = _render_device(renderer, sc, tile, rank, world)
            frame[r] = part
            total += n
        assert total == rays
        assert bits_equal(frame, rgb).all(), f"tile {tile} world {world}: partitioned frame differs"


@pytest.mark.timeout(600)
def test_c4_full_size_rows_and_partition(renderer):
    _, sc = synth.make("c4")
    assert (sc.width, sc.height) == (7680, 4320)
    renderer.upload(sc)
    rows, rgb, hits, rays = _render_device(renderer, sc, 16, 0, 1, want_hits=True)
    assert rays > 100_000_000
    _check_rows_against_oracle(sc, rows, rgb, hits, [1234, 3000])
    frame = np.empty_like(rgb)
    total = 0
    for rank in range(2):
        r, part, _, n = _render_device(renderer, sc, 16, rank, 2)
        frame[r] = part
        total += n
    assert total == rays
    assert bits_equal(frame, rgb).all()


@pytest.mark.timeout(600)
def test_c5_full_size_partition_and_batches(renderer):
    """16384 mixed solids (2 LDS windows), 16 samples per pixel: one call vs two ranks vs a run forced into
    pixel batches; the first and last rows of a 1/8-width strip are checked against the oracle elsewhere
    (reduced size), here the full-size frame is checked against itself."""
    _, sc = synth.make("c5")
    assert (sc.width, sc.height, sc.spp) == (3840, 2160, 16)
    renderer.upload(sc)
    rows, rgb, hits, rays = _render_device(renderer, sc, 8, 0, 1, want_hits=True)
    assert rays > 300_000_000
    frame = np.empty_like(rgb)
    total = 0
    for rank in range(2):
        r, part, _, n = _render_device(renderer, sc, 8, rank, 2)
        frame[r] = part
        total += n
    assert total == rays
    assert bits_equal(frame, rgb).all()
    try:
        renderer.set_option(api.OPT_MAX_SLOTS, 3_000_000)
        renderer.upload(sc)
        _, rgb3, hits3, rays3 = _render_device(renderer, sc, 8, 0, 1, want_hits=True)
    finally:
        renderer.set_option(api.OPT_MAX_SLOTS, 1 << 25)
    assert rays3 == rays and bits_equal(rgb3, rgb).all() and (hits3 == hits).all()
