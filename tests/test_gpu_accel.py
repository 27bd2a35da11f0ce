"""Grid walk (RT_OPT_ACCEL = 1): candidate solids from a front-to-back walk of a uniform grid instead of the scan of every
record.  It must not change a bit: same bar as the other parity tests (radiance, hit ids, ray counts bit-exact against
the oracle), plus -- at the full benchmark sizes, where the oracle cannot follow -- frame equality with the scan mode,
which the other tests pin against the oracle."""
import os

import numpy as np
import pytest

import pyoracle as po
from helpers import bits_equal
from raytracer_amd import api, scene as S, synth

from edge_scenes import SCENES, text_of

torch = pytest.importorskip("torch")
pytestmark = pytest.mark.gpu


@pytest.fixture()
def walker(renderer):
    renderer.set_option(api.OPT_ACCEL, 1)
    yield renderer
    renderer.set_option(api.OPT_ACCEL, 0)


def _against_oracle(ren, sc, name):
    ren.upload(sc)
    g = ren.render(want_hits=True)
    o = po.render(sc, policy=po.POLICY_DEV, rng_mode=po.RNG_PIXEL, want_hits=True)
    assert g["rays"] == int(o["rays"].sum()), f"{name}: ray count {g['rays']} != oracle {int(o['rays'].sum())}"
    assert (g["hits"] == o["hits"]).all(), f"{name}: hit ids differ"
    eq = bits_equal(g["rgb"], o["rgb"])
    assert eq.all(), f"{name}: {int((~eq).sum())} of {eq.size} radiance words differ"
    return g


@pytest.mark.parametrize("cfg,width,kw", [("c2", 480, {}), ("c3", 256, {}), ("c4", 128, {}), ("c5", 96, dict(spp=2))])
def test_walk_baseline_configs_reduced(walker, cfg, width, kw):
    _, sc = synth.make(cfg, width=width, **kw)
    _against_oracle(walker, sc, cfg)


@pytest.mark.parametrize("name", ["scene.json", "small_room.scene.json", "small_mixed.scene.json"])
def test_walk_golden_scenes(walker, golden_dir, name):
    _against_oracle(walker, S.read_json_scene(os.path.join(golden_dir, name)), name)


@pytest.mark.parametrize("name,header,objs", SCENES, ids=[s[0] for s in SCENES])
def test_walk_edge_scenes(walker, name, header, objs):
    _against_oracle(walker, S.loads(text_of(header, objs)), name)


def test_walk_grid_resolutions_and_batches(walker, monkeypatch):
    """Coarse and fine grids (RT_GRID_DENSITY is read when a context is created), pixel batches, a moved camera far from
    the solids: the frame never changes."""
    _, sc = synth.make("c5", width=128, spp=2)
    o = po.render(sc, policy=po.POLICY_DEV, rng_mode=po.RNG_PIXEL, want_hits=True)
    for density, slots in (("0.05", 1 << 25), ("16", 1 << 25), ("2", 3000)):
        monkeypatch.setenv("RT_GRID_DENSITY", density)
        r = api.Renderer(0)
        try:
            r.set_option(api.OPT_ACCEL, 1)
            r.set_option(api.OPT_MAX_SLOTS, slots)
            r.upload(sc)
            g = r.render(want_hits=True)
        finally:
            r.close()
        assert g["rays"] == int(o["rays"].sum()) and (g["hits"] == o["hits"]).all(), (density, slots)
        assert bits_equal(g["rgb"], o["rgb"]).all(), (density, slots)
    text = open(os.path.join(os.path.dirname(__file__), "golden", "small_room.scene.json")).read()
    far_cam = S.loads(text.replace('{"num_threads"', '{"camera": [3, 2.5, 4500], "num_threads"', 1))
    _against_oracle(walker, far_cam, "small_room[far camera]")


def _frame(ren, sc, tile_rows=8, rank=0, world=1):
    from raytracer_amd import dist as D
    rows = D.rows_of_rank(sc.height, tile_rows, rank, world).numpy()
    rgb = torch.empty((max(len(rows), 1), sc.width, 3), dtype=torch.float64, device="cuda:0")
    hits = torch.empty((max(len(rows), 1), sc.width, sc.spp), dtype=torch.int32, device="cuda:0")
    torch.cuda.synchronize()
    ren.render_rows_device(rgb.data_ptr(), tile_rows, rank, world, hits.data_ptr(), torch.cuda.current_stream().cuda_stream)
    rays = ren.finish()
    return rows, rgb[:len(rows)].cpu().numpy(), hits[:len(rows)].cpu().numpy(), rays


@pytest.mark.timeout(600)
@pytest.mark.parametrize("cfg", ["c3", "c4", "c5"])
def test_walk_full_size_equals_scan(renderer, cfg):
    """Full-size frames of the benchmark configurations: walk == scan bit for bit (radiance, hit ids, rays), and the
    walk over two ranks' row tiles re-assembles to the same frame."""
    _, sc = synth.make(cfg)
    renderer.upload(sc)
    _, rgb0, hits0, rays0 = _frame(renderer, sc)
    try:
        renderer.set_option(api.OPT_ACCEL, 1)
        _, rgb1, hits1, rays1 = _frame(renderer, sc)
        assert rays1 == rays0 and (hits1 == hits0).all()
        eq = bits_equal(rgb1, rgb0)
        assert eq.all(), f"{cfg}: {int((~eq).sum())} radiance words differ between walk and scan"
        if cfg == "c3":
            frame = np.empty_like(rgb0)
            total = 0
            for rank in range(2):
                r, part, _, n = _frame(renderer, sc, 8, rank, 2)
                frame[r] = part
                total += n
            assert total == rays0 and bits_equal(frame, rgb0).all()
    finally:
        renderer.set_option(api.OPT_ACCEL, 0)


def test_library_picks_the_candidate_source_per_scene(monkeypatch):
    """RT_ACCEL_AUTO (the default of a fresh context): the scan up to 3072 bounded solids, the grid walk beyond;
    rt_timing.accel_used names the one that ran, and the frame is the same bit for bit either way."""
    monkeypatch.delenv("RT_ACCEL", raising=False)
    r = api.Renderer(0)
    try:
        for lattice, want in (((8, 8, 8), api.ACCEL_SCAN), ((16, 8, 8), api.ACCEL_SCAN), ((16, 16, 12), api.ACCEL_SCAN), ((16, 16, 16), api.ACCEL_WALK)):
            _, sc = synth.make("c3", width=192, lattice=lattice)
            r.upload(sc)
            auto = r.render(want_hits=True)
            assert r.timing()["accel_used"] == want, lattice
            r.set_option(api.OPT_ACCEL, api.ACCEL_WALK if want == api.ACCEL_SCAN else api.ACCEL_SCAN)
            other = r.render(want_hits=True)
            assert r.timing()["accel_used"] != want
            r.set_option(api.OPT_ACCEL, api.ACCEL_AUTO)
            assert auto["rays"] == other["rays"] and (auto["hits"] == other["hits"]).all()
            assert np.array_equal(auto["rgb"].view(np.uint64), other["rgb"].view(np.uint64))
    finally:
        r.close()


def test_library_splits_large_renders_only(monkeypatch):
    """RT_OPT_OVERLAP = 2 (the default) applies from RT_OVERLAP_MIN_PIXELS pixels on (3 M unless set): a 1920 x 1080 frame
    goes through the pipeline in one piece, a 2560 x 1440 frame in two halves on two streams; rt_timing.sub_pipelines says
    which, and the frame is the same either way."""
    monkeypatch.delenv("RT_OVERLAP_MIN_PIXELS", raising=False)
    own = api.Renderer(0)
    monkeypatch.setenv("RT_OVERLAP_MIN_PIXELS", "0")
    split = api.Renderer(0)
    try:
        for width, want in ((1920, 1), (2560, 2)):
            _, sc = synth.make("c2", width=width)
            own.upload(sc); split.upload(sc)
            a = own.render()
            assert own.timing()["sub_pipelines"] == want, width
            b = split.render()
            assert split.timing()["sub_pipelines"] == 2
            assert a["rays"] == b["rays"]
            assert np.array_equal(a["rgb"].view(np.uint64), b["rgb"].view(np.uint64))
        own.set_option(api.OPT_OVERLAP, 1)
        own.render()
        assert own.timing()["sub_pipelines"] == 1
    finally:
        own.close(); split.close()
    # left alone, the library takes three parts for a grid-walk render of 4.7 M pixels or more (same frame again)
    monkeypatch.delenv("RT_OVERLAP_MIN_PIXELS", raising=False)
    auto, two = api.Renderer(0), api.Renderer(0)
    try:
        _, sc = synth.make("c2", width=3072)
        for r in (auto, two):
            r.set_option(api.OPT_ACCEL, api.ACCEL_WALK)
            r.upload(sc)
        two.set_option(api.OPT_OVERLAP, 2)
        a, b = auto.render(), two.render()
        assert auto.timing()["sub_pipelines"] == 3 and two.timing()["sub_pipelines"] == 2
        assert np.array_equal(a["rgb"].view(np.uint64), b["rgb"].view(np.uint64))
        auto.set_option(api.OPT_ACCEL, api.ACCEL_SCAN)
        auto.upload(sc)
        auto.render()
        assert auto.timing()["sub_pipelines"] == 2
    finally:
        auto.close(); two.close()
