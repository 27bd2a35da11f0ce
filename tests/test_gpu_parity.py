"""GPU parity: the HIP kernels (through the C ABI) against the CPU oracle in the device
arithmetic policy (the reference's double arithmetic with portable sin/cos/acos/pow), same RNG
seeding.  Bar: bit-exact radiance (float64), bit-exact hit ids (hit/miss mask), equal ray
counts -- which implies the stated tolerance of <= 1 LSB per 8-bit channel on >= 99.9 % of
pixels with zero slack.  In the reference's own RNG order the 8-bit frame is also compared with
the golden image of the compiled reference."""
import json
import os

import numpy as np
import pytest

import pyoracle as po
from helpers import bits_equal, lsb_stats
from raytracer_amd import api, scene as S, synth

pytestmark = pytest.mark.gpu


def _compare(renderer, sc, name, streams=False):
    renderer.upload(sc)
    g = renderer.render(want_hits=True, streams=streams)
    o = po.render(sc, policy=po.POLICY_DEV, rng_mode=po.RNG_STREAM if streams else po.RNG_PIXEL, want_hits=True)
    assert g["rays"] == int(o["rays"].sum()), f"{name}: ray count {g['rays']} != oracle {int(o['rays'].sum())}"
    if streams:
        assert np.array_equal(g["stream_rays"].astype(np.int64), o["rays"]), f"{name}: per-stream ray counters differ"
    hit_bad = int((g["hits"] != o["hits"]).sum())
    assert hit_bad == 0, f"{name}: {hit_bad} camera-ray hit ids differ"
    eq = bits_equal(g["rgb"], o["rgb"])
    assert eq.all(), f"{name}: {int((~eq).sum())} of {eq.size} radiance words differ"
    h = sc.header
    ia = api.expose(g["rgb"], h["exposure"], 0)
    ib, _ = po.expose(o["rgb"], h["exposure"], 0)
    frac, mx = lsb_stats(ia, ib)
    assert frac >= 0.999 and mx <= 1   # the north-star tolerance; here it holds with equality everywhere
    return g, o


@pytest.mark.parametrize("fn", ["sin", "cos", "acos"])
def test_math_unary_bit_exact(renderer, fn):
    rng = np.random.default_rng(1)
    if fn == "acos":
        x = np.concatenate([rng.uniform(-1, 1, 400000), [0, 1, -1, 0.5, -0.5, 1.0000001, -1.0000001, np.nan, 1 - 2**-53, -1 + 2**-53, 1e-300]])
    else:
        x = np.concatenate([rng.uniform(0, 2 * np.pi, 400000), rng.uniform(-100, 100, 50000),
                            [0, np.pi, np.pi / 2, 2 * np.pi, 1e5, 2e6, np.inf, -np.inf, np.nan, 1e-300]])
    assert bits_equal(renderer.selftest_math(fn, x), po.math_fn(fn, x)).all()


def test_math_pow_bit_exact(renderer):
    rng = np.random.default_rng(2)
    x = np.concatenate([rng.uniform(0, 1, 100000), rng.uniform(-1.01, 1.01, 100000), rng.uniform(0, 1, 50000),
                        [0, -0.0, 1, -1, np.inf, -np.inf, np.nan, 0.5, 2, 1e-300, 1e300, 0, 0, -2, -2, 1e-310, 0.5, 2.0, 0.999]])
    y = np.concatenate([1.0 / rng.choice([1, 50, 1000], 100000), rng.choice([1, 2, 3, 50, 1000], 100000).astype(float),
                        rng.uniform(-2000, 2000, 50000),
                        [2, 3, np.inf, np.inf, 2, 3, 1, 0, 0.5, 4, 4, -1, 0, 0.5, 3, 2, 1074, 1023.5, 5e18]])
    assert bits_equal(renderer.selftest_math("pow", x, y), po.math_fn("pow", x, y)).all()


def test_ieee_div_sqrt_and_rng(renderer):
    """The narrow phase and the shading rely on IEEE-exact double divide and sqrt on the device."""
    rng = np.random.default_rng(3)
    x = np.concatenate([rng.standard_normal(400000) * 10.0 ** rng.integers(-150, 150, 400000), [0, 1, 1e-310, 1.7e308, 2, 3]])
    y = np.concatenate([rng.standard_normal(400000) * 10.0 ** rng.integers(-150, 150, 400000), [1, 3, 3, 1e-3, 3, 7]])
    with np.errstate(all="ignore"):
        assert bits_equal(renderer.selftest_math("div", x, y), x / y).all()
        assert bits_equal(renderer.selftest_math("sqrt", np.abs(x)), np.sqrt(np.abs(x))).all()
    s = rng.integers(1, 2147483646, 100000).astype(np.uint64)
    nxt = renderer.selftest_math("rng", s.view(np.float64)).view(np.uint64)
    ref = (s * np.uint64(16807)) % np.uint64(2147483647)
    assert np.array_equal(nxt, ref)


def test_broad_phase_is_conservative(renderer):
    """No (ray, sphere) pair with a non-negative exact discriminant may be dropped by the FP32 test."""
    rng = np.random.default_rng(4)
    n = 400000
    o = rng.uniform(-60, 60, (n, 3))
    c = rng.uniform(-40, 40, (n, 3))
    r = rng.uniform(0.05, 3.0, n)
    # aim at a point near the sphere's silhouette so that about half the rays graze it
    tgt = rng.standard_normal((n, 3)); tgt /= np.linalg.norm(tgt, axis=1, keepdims=True)
    aim = c + tgt * (r * rng.uniform(0.9, 1.1, n))[:, None]
    d = aim - o; d /= np.linalg.norm(d, axis=1, keepdims=True)
    back = rng.random(n) < 0.2
    d[back] *= -1                                     # some rays point away: exercises the "behind" cull
    flags = renderer.selftest_broad(o, d, c, r)
    oc = o - c
    hb = (d * oc).sum(1)
    disc = hb * hb - ((oc * oc).sum(1) - r * r)
    front = ~((hb > 0) & ((oc * oc).sum(1) - r * r > 0))
    must_keep = (disc >= 0) & front
    assert (flags[must_keep] == 1).all(), f"{int((flags[must_keep] != 1).sum())} true hits dropped"
    # the filter still filters: its margin at these magnitudes (|o| <= 104, |c| <= 70) is 24 u (|o|+cmax)^2 < 0.05
    assert not ((flags == 1) & (disc < -0.5)).any()


def test_matrix_core_broad_phase_is_conservative(renderer):
    """The broad phase on the matrix cores (f16 pieces, two MFMAs per 32 x 32 pairs): no (ray, sphere) pair with a
    non-negative exact discriminant may be dropped -- every ray against every sphere of a cloud, plus rays aimed at the
    silhouettes, from inside the cloud and from thousands of units away -- and it must still filter."""
    rng = np.random.default_rng(11)
    n_rec, n_rays = 700, 4096                         # not multiples of 32: padded groups and a ragged last ray tile
    c = rng.uniform(-40, 40, (n_rec, 3))
    r = rng.uniform(0.05, 3.0, n_rec)
    o = rng.uniform(-60, 60, (n_rays, 3))
    # half of the rays graze a chosen sphere, the rest point anywhere
    pick = rng.integers(0, n_rec, n_rays)
    tgt = rng.standard_normal((n_rays, 3)); tgt /= np.linalg.norm(tgt, axis=1, keepdims=True)
    aim = c[pick] + tgt * (r[pick] * rng.uniform(0.95, 1.05, n_rays))[:, None]
    d = aim - o
    anyway = rng.random(n_rays) < 0.5
    d[anyway] = rng.standard_normal((int(anyway.sum()), 3))
    d /= np.linalg.norm(d, axis=1, keepdims=True)
    keep = renderer.selftest_broad_mfma(o, d, c, r)
    assert keep.shape == (n_rays, n_rec)
    oc = o[:, None, :] - c[None, :, :]
    hb = (d[:, None, :] * oc).sum(2)
    disc = hb * hb - ((oc * oc).sum(2) - (r * r)[None, :])
    dropped = (disc >= 0) & ~keep
    assert not dropped.any(), f"{int(dropped.sum())} true hits dropped, worst discriminant {disc[dropped].max()}"
    # margin at these magnitudes (|o| <= 104, |c| <= 70): about 0.1 + 2e-5 |hb|^2 / ... well below 1
    assert not (keep & (disc < -1.0)).any()
    assert keep.mean() < 0.02
    # the grazing rays: the picked sphere is a true hit for about half of them, all kept
    own = keep[np.arange(n_rays), pick]
    own_disc = disc[np.arange(n_rays), pick]
    assert own[(own_disc >= 0)].all() and (own_disc >= 0).sum() > 500
    # the same silhouettes seen from far outside (the test moves the origin along the line; the f16 range must hold)
    far_o = aim + tgt * 0.0 - d * rng.uniform(2e3, 2e4, n_rays)[:, None]
    keep_far = renderer.selftest_broad_mfma(far_o, d, c, r)
    oc = far_o[:, None, :] - c[None, :, :]
    hb = (d[:, None, :] * oc).sum(2)
    disc_far = hb * hb - ((oc * oc).sum(2) - (r * r)[None, :])
    assert not ((disc_far >= 1e-6) & ~keep_far).any()      # (the discriminant itself is computed here with |o| ~ 1e4: ~1e-8 noise)
    assert not keep_far.all(1).any() and keep_far.mean() < 0.02
    # rays that are not of ordinary size keep everything
    o2 = o[:64].copy(); o2[::2, 0] = 1e30; o2[1::2, 1] = np.nan
    assert renderer.selftest_broad_mfma(o2, d[:64], c, r).all()


def test_scene_json_matches_oracle(renderer, golden_dir):
    sc = S.read_json_scene(os.path.join(golden_dir, "scene.json"))
    g, o = _compare(renderer, sc, "scene.json")
    assert g["rays"] > 4_000_000


def test_scene_json_reference_order_matches_reference_image(renderer, golden_dir):
    """One lane per reference thread, the reference's RNG order: the 8-bit frame must equal the golden image
    of the compiled reference (num_threads=4, bloom included), and the ray counters its num_rays[]."""
    sc = S.read_json_scene(os.path.join(golden_dir, "scene.json"))
    gold = [e for e in json.load(open(os.path.join(golden_dir, "golden.json")))
            if e["scene"] == "scene.json" and e["threads"] == 4 and e["blurs"] == 3][0]
    g, o = _compare(renderer, sc, "scene.json[streams]", streams=True)
    assert [int(x) for x in g["stream_rays"]] == gold["rays"]
    img, _ = po.expose(g["rgb"], sc.header["exposure"], sc.header["num_blurs"])
    ref8 = np.load(os.path.join(golden_dir, gold["pixels"]))["rgb8"]
    got8 = po.unpack_rgb(img)
    d = np.abs(got8.astype(int) - ref8.astype(int))
    assert (d.max(-1) <= 1).mean() >= 0.999
    assert int((d > 0).sum()) == 0, "frame differs from the reference binary's image"


@pytest.mark.parametrize("name", ["small_room", "small_mixed"])
def test_golden_synthetic_scenes_match_oracle(renderer, golden_dir, name):
    sc = S.read_json_scene(os.path.join(golden_dir, f"{name}.scene.json"))
    _compare(renderer, sc, name)
    _compare(renderer, sc, name + "[streams]", streams=True)


def test_single_kernel_and_pipeline_agree(renderer):
    """Both renderers (three-kernel pipeline, persistent single kernel) and a pipeline forced to work in
    several pixel batches and several LDS windows give the same bits."""
    _, sc = synth.make("c5", width=128, spp=2)
    o = po.render(sc, policy=po.POLICY_DEV, rng_mode=po.RNG_PIXEL, want_hits=True)
    try:
        for mode, slots, window in ((api.MODE_PIPELINE, 1 << 25, 8192), (api.MODE_SINGLE_KERNEL, 1 << 25, 8192),
                                    (api.MODE_PIPELINE, 3000, 1000), (api.MODE_SINGLE_KERNEL, 1 << 25, 1000)):
            renderer.set_option(api.OPT_MODE, mode)
            renderer.set_option(api.OPT_MAX_SLOTS, slots)
            renderer.set_option(api.OPT_CHUNK_RECORDS, window)
            renderer.upload(sc)
            g = renderer.render(want_hits=True)
            assert g["rays"] == int(o["rays"].sum())
            assert (g["hits"] == o["hits"]).all()
            assert bits_equal(g["rgb"], o["rgb"]).all(), (mode, slots, window)
    finally:
        renderer.set_option(api.OPT_MODE, api.MODE_PIPELINE)
        renderer.set_option(api.OPT_MAX_SLOTS, 1 << 25)
        renderer.set_option(api.OPT_CHUNK_RECORDS, 8192)


@pytest.mark.parametrize("cfg,width,kw", [("c2", 480, {}), ("c3", 256, {}), ("c4", 128, {}), ("c5", 96, dict(spp=2))])
def test_baseline_configs_reduced(renderer, cfg, width, kw):
    _, sc = synth.make(cfg, width=width, **kw)
    _compare(renderer, sc, cfg)


def test_group_record_pass_drops_groups_not_hits(renderer, monkeypatch):
    """With its table in its own order (compact groups of 32) the matrix-core scan asks the bounding spheres of the 32-record groups first (per
    tile in the camera rays' round, per ray afterwards).  It must compute clearly fewer (32 rays x 32 records) products
    than rays x groups and leave every output bit as it is without that (RT_MFMA_CULL=0: file order, every group scanned)."""
    _, sc = synth.make("c3", width=960)
    renderer.upload(sc)
    a = renderer.render(want_hits=True)
    ta = renderer.timing()
    assert ta["broad_records"] == 1024 and ta["broad_tiles"] > 0          # 1024 spheres: the walls are not in the table
    monkeypatch.setenv("RT_MFMA_CULL", "0")
    plain = api.Renderer(0)
    try:
        plain.upload(sc)
        b = plain.render(want_hits=True)
        tb = plain.timing()
    finally:
        plain.close()
    assert tb["broad_records"] == 1056                                     # file order: 1031 solids padded to whole groups
    # every ray against every group there; here the groups the group records leave: well under half the products
    assert tb["broad_tiles"] >= a["rays"] // 32 * 33 and ta["broad_tiles"] < 0.5 * tb["broad_tiles"], (ta["broad_tiles"], tb["broad_tiles"])
    assert a["rays"] == b["rays"] and np.array_equal(a["hits"], b["hits"]) and bits_equal(a["rgb"], b["rgb"]).all()


def test_groups_behind_the_origin_are_dropped_not_hits(renderer, monkeypatch):
    """The binned scan does not list a ray for a group whose bounding sphere lies wholly behind the ray's origin (both
    line parameters negative, decided from a lower bound of d.(o - c) and an upper bound of the discriminant).  It must
    compute fewer products and keep fewer candidates than with RT_MFMA_BEHIND=0, and leave every output bit alone."""
    _, sc = synth.make("c3", width=960)
    renderer.upload(sc)
    a = renderer.render(want_hits=True)
    ta = renderer.timing()
    monkeypatch.setenv("RT_MFMA_BEHIND", "0")
    other = api.Renderer(0)
    try:
        other.upload(sc)
        b = other.render(want_hits=True)
        tb = other.timing()
    finally:
        other.close()
    assert ta["broad_tiles"] < 0.97 * tb["broad_tiles"] and ta["candidates"] < tb["candidates"], (ta, tb)
    assert a["rays"] == b["rays"] and np.array_equal(a["hits"], b["hits"]) and bits_equal(a["rgb"], b["rgb"]).all()


@pytest.mark.parametrize("env", [{"RT_MFMA_LIST": "0"}, {"RT_MFMA_LIST": "24"}, {"RT_MFMA_BINNED": "0"}])
def test_binned_scan_variants_do_not_change_bits(renderer, monkeypatch, env):
    """The scan of the later rounds lists every group's rays and computes only listed (group, rays) products.  A group
    that more rays keep than its list holds is scanned for all rays of the pass instead: forced here for every group
    (0 places) and for the busier ones (24), and compared with the default and with the plain scan, bit for bit."""
    _, sc = synth.make("c3", width=640)
    renderer.upload(sc)
    a = renderer.render(want_hits=True)
    for k, v in env.items():
        monkeypatch.setenv(k, v)
    other = api.Renderer(0)
    try:
        other.upload(sc)
        b = other.render(want_hits=True)
    finally:
        other.close()
    assert a["rays"] == b["rays"] and np.array_equal(a["hits"], b["hits"]) and bits_equal(a["rgb"], b["rgb"]).all()


def test_reordered_table_with_unbounded_records_matches_oracle(monkeypatch):
    """More planes than the always-tested table takes (64): the rest stay records of the scan that every ray keeps, here
    in a table that is in its own order -- forced for these 320 spheres, with the group records asked per tile in
    the camera rays' round and the binned scan afterwards."""
    import random
    rng = random.Random(7)
    objs = synth._room(2, 7)
    for i in range(70):                                # parallel planes behind the back wall: tested, never the closest hit
        objs.append({"shape": "plane", "is_light": False, "point": [0, 0, -100.0 - i], "normal": [0, 0, 1],
                     "diffuse1": [0.5, 0.5, 0.5], "diffuse2": [0.4, 0.4, 0.4], "specular1": [0, 0, 0], "specular2": [0, 0, 0],
                     "specular_exp": 1, "refraction_index": 0})
    objs += synth._lattice_objects(rng, 8, 8, 5, 320, 0.05, 0.0)
    header = {"num_threads": 4, "num_objs": len(objs), "exposure": 1, "num_blurs": 0, "paths_per_pixel": 1, "max_ray_depth": 3,
              "scale": 768 / 40.0, "width": 768, "height": 432, "lens_radius": 1.0, "focal_length": 60}
    sc = S.loads(S.dumps(header, objs))
    monkeypatch.setenv("RT_MFMA_MIN_BOUNDED", "64")
    monkeypatch.setenv("RT_MFMA_BINNED_MIN", "1")
    ren = api.Renderer(0)
    try:
        ren.set_option(api.OPT_OVERLAP, 1)            # one queue of 331 776 rays: more batches than workgroups, no slices
        g, _ = _compare(ren, sc, "320 spheres + 78 planes")
        t = ren.timing()
    finally:
        ren.close()
    # own order: the always-tested solids are not in the table (file order would give 416 records)
    assert t["broad_records"] == 352 and 0 < t["broad_tiles"] < g["rays"] // 32 * 11


def test_flat_sphere_layer_in_own_table_order_matches_oracle(renderer):
    """600 spheres in one plane z = const (no extent along one axis for the table's cuts), some of them concentric:
    enough for the scan's table to take its own order; pinned by the oracle."""
    objs = synth._room(2, 0)
    for i in range(600):
        x, y = -18.0 + 1.5 * (i % 25), -18.0 + 1.5 * (i // 25)
        if i % 50 == 49:
            x, y = -18.0, -18.0                       # a pile of concentric spheres: ties go to the lower file index
        objs.append({"shape": "sphere", "is_light": False, "center": [x, y, -20.0], "radius": 0.4 + 0.001 * (i % 7),
                     "diffuse": [0.5, 0.4, 0.3], "specular": [0.2, 0.2, 0.2], "specular_exp": 20, "refraction_index": 0,
                     "refraction_diffuse": [0, 0, 0], "refraction_specular": [0, 0, 0], "refraction_specular_exp": 0})
    header = {"num_threads": 4, "num_objs": len(objs), "exposure": 1, "num_blurs": 0, "paths_per_pixel": 1, "max_ray_depth": 3,
              "scale": 320 / 40.0, "width": 320, "height": 180, "lens_radius": 0.5, "focal_length": 60}
    sc = S.loads(S.dumps(header, objs))
    _compare(renderer, sc, "flat layer of 600 spheres")
    assert renderer.timing()["broad_records"] == 608          # 600 spheres in 19 groups, the room's solids not among them


def test_moved_camera_matches_oracle(renderer, golden_dir):
    """Header extension "camera": the device and the oracle read the same three parameters (the reference's constant
    (0,0,40) is their default); no reference output exists for a moved camera, so this case is pinned by the oracle only."""
    text = open(os.path.join(golden_dir, "small_room.scene.json")).read()
    sc = S.loads(text.replace('{"num_threads"', '{"camera": [3, -2.5, 45], "num_threads"', 1))
    g, o = _compare(renderer, sc, "small_room[camera]")
    base = S.loads(text)
    renderer.upload(base)
    assert not np.array_equal(renderer.render()["rgb"], g["rgb"])       # the key has an effect


@pytest.mark.parametrize("k", [4096.0, 0.25])
def test_scaled_scene_matches_oracle(renderer, golden_dir, k):
    """The same room k times as large (and as far from the origin: the matrix-core scan's f16 table works on lengths
    scaled by a power of two chosen from the scene, 2^-11 and 2^3 here).  Pinned by the oracle only."""
    sc = S.read_json_scene(os.path.join(golden_dir, "small_mixed.scene.json"))
    sc.prims[:, S.PR_A:S.PR_A + 3] *= k
    sc.prims[:, S.PR_RADIUS] *= k
    sc.verts *= k
    h = sc.header
    h["scale"] = h["scale"] / k
    h["lens_radius"] = h["lens_radius"] * k
    h["focal_length"] = h["focal_length"] * k
    h["camera"] = [c * k for c in S.DEFAULT_CAMERA]
    g, _ = _compare(renderer, sc, f"small_mixed x{k}")
    assert g["rays"] > 10_000


def test_camera_beyond_the_box_test_range(renderer, golden_dir):
    """Origins of 1e200: beyond the range for which the always-tested walls' coordinate-range test is proven, so every
    such ray takes the general test (whole-wave scan in the pipeline, the loop in the single-kernel renderer).  Pinned
    by the oracle only (no reference output exists for a moved camera)."""
    text = open(os.path.join(golden_dir, "small_room.scene.json")).read()
    sc = S.loads(text.replace('{"num_threads"', '{"camera": [0.5, -0.25, 1e200], "num_threads"', 1))
    for mode in (api.MODE_PIPELINE, api.MODE_SINGLE_KERNEL):
        try:
            renderer.set_option(api.OPT_MODE, mode)
            _compare(renderer, sc, f"small_room[camera 1e200, mode {mode}]")
        finally:
            renderer.set_option(api.OPT_MODE, api.MODE_PIPELINE)
