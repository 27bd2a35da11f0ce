"""Host side of the boundary without a GPU: scene stream reader (Python mirror and C++ loader), expose,
PNG writer, the C-ABI library's exports and its GPU-free entry points."""
import ctypes as C
import os
import re
import struct
import subprocess
import sys

import numpy as np
import pytest

import pyoracle as po
from raytracer_amd import api, scene as S, synth

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
PKG = os.path.join(ROOT, "raytracer_amd")


def _host_lib():
    p = os.path.join(PKG, "librt_host.so")
    if not os.path.exists(p):
        subprocess.check_call(["make", "-C", os.path.join(PKG, "csrc"), "../librt_host.so"])
    return C.CDLL(p)


def _dump_tables(path):
    exe = os.path.join(PKG, "rt_main")
    out = path + ".tables"
    subprocess.check_call([exe, path, "/dev/null", "--no-render", "--dump-tables", out])
    b = open(out, "rb").read()
    npar, npr, nmt, nv = struct.unpack("<4i", b[:16])
    off = 16
    res = []
    for n in (npar, npr * S.PR_STRIDE, nmt * S.MT_STRIDE, nv * 3):
        res.append(np.frombuffer(b, dtype="<f8", count=n, offset=off)); off += 8 * n
    return res


def test_scene_json_tables(golden_dir):
    sc = S.read_json_scene(os.path.join(golden_dir, "scene.json"))
    assert sc.counts() == {"spheres": 2, "planes": 1, "faces": 5}
    assert (sc.width, sc.height, sc.spp) == (400, 400, 10)
    assert sc.mats.shape == (9, S.MT_STRIDE) and sc.verts.shape == (20, 3)
    # gloss = spec/(diff+spec): mirror sphere 3/3, walls 0.9/(1.5+0.9)
    assert sc.mats[int(sc.prims[6, S.PR_MAT0]), S.MT_GLOSS] == 1.0
    assert sc.mats[int(sc.prims[2, S.PR_MAT0]), S.MT_GLOSS] == (0.3 + 0.3 + 0.3) / ((0.1 + 0.7 + 0.1) + (0.3 + 0.3 + 0.3))
    assert sc.prims[5, S.PR_LIGHT] == 1.0 and sc.mats[int(sc.prims[5, S.PR_MAT0]), S.MT_LIGHT] == 1.0


@pytest.mark.parametrize("name", ["scene.json", "small_room.scene.json", "small_mixed.scene.json"])
def test_cpp_loader_equals_python_loader(golden_dir, name, tmp_path):
    src = os.path.join(golden_dir, name)
    dst = str(tmp_path / name)
    open(dst, "w").write(open(src).read())
    params, prims, mats, verts = _dump_tables(dst)
    sc = S.read_json_scene(src)
    assert np.array_equal(params.view(np.uint64), sc.params().view(np.uint64))
    assert np.array_equal(prims.view(np.uint64), sc.prims.ravel().view(np.uint64))
    assert np.array_equal(mats.view(np.uint64), sc.mats.ravel().view(np.uint64))
    assert np.array_equal(verts.view(np.uint64), sc.verts.ravel().view(np.uint64))


def test_camera_header_key_is_optional_and_read_by_both_loaders(golden_dir, tmp_path):
    """Extension of the scene format: "camera": [x, y, z] in the header; default = the reference's (0,0,40)."""
    text = open(os.path.join(golden_dir, "small_room.scene.json")).read()
    assert S.loads(text).params()[S.PA_CAM_X:S.PA_CAM_Z + 1].tolist() == [0.0, 0.0, 40.0]
    moved = text.replace('{"num_threads"', '{"camera": [3, -2.5, 45], "num_threads"', 1)
    sc = S.loads(moved)
    assert sc.params()[S.PA_CAM_X:S.PA_CAM_Z + 1].tolist() == [3.0, -2.5, 45.0]
    dst = str(tmp_path / "moved.json")
    open(dst, "w").write(moved)
    params = _dump_tables(dst)[0]
    assert np.array_equal(params.view(np.uint64), sc.params().view(np.uint64))


def test_plane_normal_is_normalised_twice_and_face_normal_once(tmp_path):
    """readShape.cpp:73 + geometry.cpp:97 normalise a plane's normal twice, readShape.cpp:95 a face's once.  For
    (1, 20, -1) the second pass moves every component by an ulp, so the two solids' normals differ; both loaders
    must reproduce exactly that."""
    import math
    from edge_scenes import SCENES, text_of

    def nrm(n):                       # Vector::normalize: sqrt of the left-to-right sum of squares, one division each
        t = 0.0
        for c in n:
            t += c * c
        return [c / math.sqrt(t) for c in n]

    once = nrm([1.0, 20.0, -1.0])
    twice = nrm(once)
    assert once != twice
    _, header, objs = [s for s in SCENES if s[0] == "tilted_plane_normal_normalised_twice"][0]
    text = text_of(header, objs)
    sc = S.loads(text)
    assert sc.prims[0, S.PR_KIND] == S.KIND_PLANE and sc.prims[3, S.PR_KIND] == S.KIND_FACE
    assert sc.prims[0, S.PR_N:S.PR_N + 3].tolist() == twice
    assert sc.prims[3, S.PR_N:S.PR_N + 3].tolist() == once
    dst = str(tmp_path / "tilted.json")
    open(dst, "w").write(text)
    prims = _dump_tables(dst)[1]
    assert np.array_equal(prims.view(np.uint64), sc.prims.ravel().view(np.uint64))


def test_scene_errors():
    with pytest.raises(S.SceneError):
        S.loads("")
    with pytest.raises(S.SceneError):
        S.loads('{"num_objs": 1}')
    head = '{"num_threads":1,"num_objs":1,"exposure":1,"num_blurs":0,"paths_per_pixel":1,"max_ray_depth":1,"scale":1,"width":2,"height":2,"lens_radius":0,"focal_length":1}'
    with pytest.raises(S.SceneError):
        S.loads(head)                                        # object announced, none present
    with pytest.raises(S.SceneError):
        S.loads(head + '{"shape":"sphere","is_light":false,"center":[0,0,0]}')   # radius missing
    with pytest.raises(S.SceneError):
        S.loads(head + '{"shape": "sphere", ')               # malformed
    sc = S.loads(head + '{"shape":"sphere","is_light":true,"center":[0,0,0],"radius":1,"color":[1,2,3]} {"ignored": 1}')
    assert len(sc.prims) == 1 and sc.mats[0, S.MT_REF:S.MT_REF + 3].tolist() == [1, 2, 3]
    exe = os.path.join(PKG, "rt_main")
    r = subprocess.run([exe, "/nonexistent/scene.json", "/dev/null", "--no-render"], capture_output=True)
    assert r.returncode == 1 and b"cannot open" in r.stderr


def test_synth_is_deterministic_and_well_formed():
    t1, s1 = synth.make("c3", width=64)
    t2, s2 = synth.make("c3", width=64)
    assert t1 == t2
    assert s1.counts() == {"spheres": 1024, "planes": 1, "faces": 6}
    _, s5 = synth.make("c5", width=64)
    assert len(s5.prims) == 16384
    # solids stay inside the room and do not overlap (sampled check on sphere pairs of neighbouring cells)
    sp = s1.prims[s1.prims[:, S.PR_KIND] == S.KIND_SPHERE]
    c, r = sp[:, S.PR_A:S.PR_A + 3], sp[:, S.PR_RADIUS]
    assert (np.abs(c[:, 0]) + r < 20).all() and (c[:, 1] - r > -20).all() and (c[:, 1] + r < 20).all()
    d = np.linalg.norm(c[:, None, :] - c[None, :, :], axis=-1) + np.eye(len(c)) * 1e9
    assert (d > r[:, None] + r[None, :]).all()


def test_expose_host_implementations_agree():
    rng = np.random.default_rng(0)
    rgb = rng.random((50, 70, 3)) ** 3 * 3
    want, _ = po.expose(rgb, 1.0, 2)
    assert np.array_equal(api.expose(rgb, 1.0, 2), want)
    lib = _host_lib()
    buf = np.ascontiguousarray(rgb).copy()
    img = np.zeros((50, 70), dtype=np.int32)
    lib.rt_host_expose(img.ctypes.data_as(C.POINTER(C.c_int)), buf.ctypes.data_as(C.POINTER(C.c_double)),
                       70, 50, C.c_double(1.0), 2, 3)
    assert np.array_equal(img, want)
    want2, _ = po.expose(rgb * 0.2, 2.5, 0)
    assert np.array_equal(api.expose(rgb * 0.2, 2.5, 0), want2)


def test_write_image_round_trip(tmp_path):
    from PIL import Image
    lib = _host_lib()
    rng = np.random.default_rng(1)
    img = rng.integers(0, 1 << 24, (37, 53)).astype(np.int32)
    path = str(tmp_path / "o.png").encode()
    assert lib.writeImage(C.c_char_p(path), 53, 37, img.ctypes.data_as(C.POINTER(C.c_int))) == 0
    got = np.array(Image.open(path.decode()).convert("RGB"))
    assert np.array_equal(got, po.unpack_rgb(img))
    assert lib.writeImage(C.c_char_p(b"/nonexistent/dir/o.png"), 53, 37, img.ctypes.data_as(C.POINTER(C.c_int))) == -1


def test_c_abi_exports_every_declared_symbol():
    # the render path's calls (rt_hip.h) and the self-test / inspection calls (rt_hip_selftest.h)
    hdr = open(os.path.join(ROOT, "include", "rt_hip.h")).read() + open(os.path.join(ROOT, "include", "rt_hip_selftest.h")).read()
    assert "rt_selftest" not in open(os.path.join(ROOT, "include", "rt_hip.h")).read()
    declared = set(re.findall(r"^(?:int|void|double|const char\*)\s+(rt_[a-z_]+)\s*\(", hdr, flags=re.M))
    assert declared == set(api.EXPORTS), declared ^ set(api.EXPORTS)
    lib = api.load_library()
    for name in declared:
        assert hasattr(lib, name), name


def test_c_abi_entry_points_that_need_no_gpu():
    lib = api.load_library()
    assert api.rows_for_rank(2160, 8, 0, 8) == 272 and api.rows_for_rank(2160, 8, 7, 8) == 264
    assert sum(api.rows_for_rank(1081, 8, r, 3) for r in range(3)) == 1081
    assert api.rows_for_rank(10, 0, 0, 1) == -1 and api.rows_for_rank(10, 1, 2, 2) == -1
    assert lib.rt_last_error(None) == b"null handle"
    assert lib.rt_upload_scene(None, None, 0, None, 0, None, 0, None) == -1
    assert lib.rt_finish(None, None) == -1
    h = C.c_void_p()
    rc = lib.rt_create(C.byref(h), 0)
    if rc == 0:                       # a GPU is present: exercise the error paths that need a handle
        assert lib.rt_render(h, None, None, None) == -3
        assert lib.rt_set_option(h, 99, 0) == -1
        lib.rt_destroy(h)
    else:
        assert rc == -6 and not h.value


# ---------------------------------------------------------------- bounding sphere of a face's accepted region
def _face_accepts(verts, pts):
    """Face::intersect's vertex loop (geometry.cpp:162-166) on points of the plane: to_p * to_next < 0 rejects."""
    nv = len(verts)
    ok = np.ones(len(pts), dtype=bool)
    for i in range(nv):
        e = verts[i] - verts[(i + 1) % nv]
        to_next = e @ (verts[i] - verts[(i + 2) % nv])
        to_p = (verts[i] - pts) @ e
        ok &= ~(to_p * to_next < 0)
    return ok


FACES = {
    # name: (verts, expected bounded?)
    "rectangle": ([(-3, -2, -5), (4, -2, -5), (4, 1, -5), (-3, 1, -5)], True),
    "tilted_square": ([(0, 0, 0), (1, 0, 1), (1, 1.5, 1), (0, 1.5, 0)], True),
    "acute_triangle": ([(-6, -9, -10), (8, -9, -10), (1, 7, -10)], True),
    "right_triangle": ([(-8, -8, -14), (4, -8, -14), (-8, 6, -14)], False),
    "obtuse_triangle": ([(0, 0, 2), (10, 0, 2), (14, 3, 2)], None),
    "parallelogram": ([(-10, -8, -16), (2, -8, -16), (8, 4, -16), (-4, 4, -16)], None),
    "sheared_parallelogram": ([(0, 0, 1), (4, 0, 1), (24, 5, 1), (20, 5, 1)], None),
    "pentagon": ([(12, -10, -8), (18, -6, -8), (16, 2, -8), (9, 3, -8), (7, -5, -8)], True),
    "far_small_square": ([(1e6, 1e6, 3), (1e6 + 1, 1e6, 3), (1e6 + 1, 1e6 + 1, 3), (1e6, 1e6 + 1, 3)], True),
    "two_vertices": ([(0, 0, 0), (1, 0, 0)], False),
}


@pytest.mark.parametrize("name", sorted(FACES))
def test_face_bound_holds_every_accepted_point(name):
    """No point the reference's Face::intersect accepts may lie outside the sphere the broad phase and the grid cull
    with; shapes whose accepted region is unbounded must come back unbounded."""
    verts, bounded = FACES[name]
    v = np.array(verts, dtype=np.float64)
    if len(v) >= 3:
        n = np.cross(v[1] - v[0], v[2] - v[0])
    else:
        n = np.array([0.0, 0.0, 1.0])
    n = n / np.linalg.norm(n)
    b = api.face_bound(v, n)
    if bounded is not None:
        assert (b is not None) == bounded, f"{name}: bounded={b is not None}"
    # points of the plane: dense around the face, and sparse out to 200 extents
    rng = np.random.default_rng(5)
    cen = v.mean(0)
    ext = np.linalg.norm(v - cen, axis=1).max()
    u = np.cross(n, [1.0, 0, 0] if abs(n[0]) < 0.9 else [0, 1.0, 0]); u /= np.linalg.norm(u)
    w = np.cross(n, u)
    ab = np.concatenate([rng.uniform(-4, 4, (200000, 2)), rng.uniform(-200, 200, (50000, 2))]) * ext
    pts = cen + ab[:, :1] * u + ab[:, 1:] * w
    pts = pts - ((pts - v[0]) @ n)[:, None] * n
    acc = _face_accepts(v, pts)
    assert acc.any()
    if b is None:
        assert bounded is not True
        return
    c, r = b
    far = np.linalg.norm(pts[acc] - c, axis=1).max()
    assert far <= r, f"{name}: accepted point {far} from the centre, radius {r}"
    if name == "rectangle":
        assert abs(r - 0.5 * np.hypot(7, 3)) < 1e-4
    if name == "acute_triangle":     # the region is larger than the circle through the vertices allows
        assert far > np.linalg.norm(v - cen, axis=1).max()


def test_rng_uniform_reciprocal_division_is_the_ieee_quotient(tmp_path):
    """The device takes (double)g / 2147483646 (main.cpp:255) as g*y + (g - b*(g*y))*y with y = fl(1/b).  tools/check_rng_div.c
    compares that with the true division for every possible draw g = 1 .. 2^31 - 2 (a few seconds)."""
    exe = str(tmp_path / "check_rng_div")
    subprocess.check_call(["gcc", "-O2", "-mfma", "-ffp-contract=off", "-o", exe, os.path.join(ROOT, "tools", "check_rng_div.c"), "-lm"])
    r = subprocess.run([exe], capture_output=True, text=True, timeout=300)
    assert r.returncode == 0 and "bad = 0" in r.stdout, r.stdout
    # and the constant in the kernel source is that reciprocal
    assert "0x1.00000004p-31" in r.stdout and "0x1.00000004p-31" in open(os.path.join(PKG, "csrc", "rt_geometry.h")).read()
    assert float.fromhex("0x1.00000004p-31") == 1.0 / 2147483646.0


def test_numpy_expose_mirror_converts_like_x86_at_the_edges():
    """api.expose against the oracle's expose (which follows the reference's cvttsd2si conversion) on the values
    Color::to_int meets at its edges: out of int range in both directions, infinities, NaN, negative, exact steps."""
    special = np.array([0.0, -0.0, -0.3, 1.0, np.nextafter(1.0, 0), np.nextafter(1.0, 9), 1e300, -1e300, np.inf, -np.inf, np.nan,
                        3e9, -3e9, 8421504.5, 8421505.0, -8421505.0, 1e-320, 254.999999 / 255, 2147483647.9 / 255, 2147483648.0 / 255])
    rgb = np.zeros((5, 8, 3))
    rgb.reshape(-1)[: special.size] = special
    rgb.reshape(-1)[special.size: 2 * special.size] = special[::-1]
    for exposure in (1.0, 0.5, 2.0):
        want, _ = po.expose(rgb, exposure, 0)
        assert np.array_equal(api.expose(rgb, exposure, 0), want), exposure


def _is_f16(v):
    """v (a double) is a value of IEEE binary16, subnormals included"""
    if v == 0.0 or not np.isfinite(v):
        return True
    m, e = np.frexp(abs(v))                      # |v| = m 2^e, m in [0.5, 1)
    if e > 16:
        return False
    step = 2.0 ** (max(e, -13) - 11)              # 11 significant bits; a fixed step of 2^-24 below 2^-14
    return (abs(v) / step) == np.floor(abs(v) / step)


def test_matrix_core_record_pieces_are_f16_and_round_towards_keep():
    """The f16 table of the scan on the matrix cores: every piece is a binary16 value, a coordinate's two pieces give it
    back to 2^-21 of its magnitude, w's pieces never sum to less than w (a larger w keeps more), the scale is the power of
    two that puts the scene's extent in (32, 64], and never / always kept records carry -/+ 65504."""
    rng = np.random.default_rng(4)
    for cmax in (0.7, 33.3, 47.0, 64.0, 65.0, 900.0, 3.0e5, 1.0e-3):
        scale, _ = api.mfma_record_pieces([0, 0, 0, 0], cmax)
        m, e = np.frexp(scale)
        assert m == 0.5 and 32.0 < cmax * scale * (1 + 1e-12) <= 64.0 * (1 + 1e-6)
        for _ in range(200):
            c = rng.uniform(-1, 1, 3) * cmax * rng.choice([1.0, 1e-2, 1e-5])
            R = rng.uniform(0, 0.3) * cmax
            w = float(np.float32(R * R - c @ c))
            c = np.float32(c).astype(np.float64)
            s, p = api.mfma_record_pieces([*c, w], cmax)
            assert all(_is_f16(v) for v in p.ravel())
            for i in range(3):
                x = c[i] * s
                assert abs(p[i, 0] + p[i, 1] - x) <= abs(x) * 2.0 ** -21 + 2.0 ** -24
                assert abs(p[i, 0]) <= abs(x) and p[i, 0] * x >= 0       # cut towards zero
            ws = float(np.float32(w * s * s))
            assert p[3, 0] + p[3, 1] >= ws
            assert p[3, 0] + p[3, 1] - ws <= abs(ws) * 2.0 ** -20 + 2.0 ** -23
    _, p = api.mfma_record_pieces([0, 0, 0, -np.inf], 10.0)
    assert tuple(p[3]) == (-65504.0, 0.0)
    _, p = api.mfma_record_pieces([0, 0, 0, np.inf], 10.0)
    assert tuple(p[3]) == (65504.0, 0.0)


def test_matrix_core_table_order_and_group_spheres():
    """The scan's own record order: a permutation of the solids that can be kept (never-kept ones left out, unbounded ones
    last), neighbours in space side by side, and every group's sphere holds its members' spheres."""
    rng = np.random.default_rng(9)
    n = 1000
    c = rng.uniform(-30, 30, (n, 3))
    r = rng.uniform(0.1, 1.5, n)
    never = rng.random(n) < 0.03
    unb = (rng.random(n) < 0.01) & ~never
    r[never] = -1.0
    r[unb] = -2.0
    perm, gs = api.mfma_table_order(c, r, reorder=True)
    NONE = 0x7fffffff
    used = perm[perm != NONE]
    assert sorted(used.tolist()) == sorted(np.flatnonzero(~never).tolist())
    assert len(perm) % 32 == 0 and len(perm) - len(used) < 32
    n_b = int((~never & ~unb).sum())
    assert set(used[n_b:].tolist()) == set(np.flatnonzero(unb).tolist())          # unbounded ones at the end
    spread_sorted, spread_file = [], []
    for g in range(len(gs)):
        m = perm[32 * g: 32 * g + 32]
        m = m[m != NONE]
        if len(m) == 0:
            assert gs[g, 3] == -1.0
            continue
        if unb[m].any():
            assert np.isinf(gs[g, 3])
            continue
        d = np.linalg.norm(c[m] - gs[g, :3], axis=1) + r[m]
        assert (d <= gs[g, 3] * (1 + 1e-9)).all()
        spread_sorted.append(gs[g, 3])
    # file order of random points: every group spans the scene; the table's own order: a fraction of it
    _, gs_file = api.mfma_table_order(c, np.where(never | unb, 0.5, r), reorder=False)
    assert np.mean(spread_sorted) < 0.6 * np.mean(gs_file[:, 3])
    perm_f, _ = api.mfma_table_order(c, r, reorder=False)
    assert (perm_f[:n] == np.arange(n)).all() and (perm_f[n:] == NONE).all()


def test_gather_source_row_matches_the_partition():
    """rt_gather_source_row (host arithmetic of the in-library gather) against the packing order of the partition: rank r's
    buffer holds its rows in ascending order, buffers are stacked by rank with pad_rows rows each."""
    from raytracer_amd.dist import max_rows, rows_of_rank
    lib = api.load_library()
    for (h, t, w) in ((90, 8, 3), (37, 1, 4), (40, 8, 8), (2160, 8, 8), (23, 23, 2), (5, 2, 7)):
        pad = max_rows(h, t, w)
        want = np.full(h, -1)
        for r in range(w):
            rows = rows_of_rank(h, t, r, w).numpy()
            want[rows] = r * pad + np.arange(len(rows))
        got = np.array([lib.rt_gather_source_row(h, t, w, pad, y) for y in range(h)])
        assert np.array_equal(got, want), (h, t, w)
        assert sum(lib.rt_rows_for_rank(h, t, r, w) for r in range(w)) == h
    assert lib.rt_gather_source_row(10, 0, 2, 5, 0) < 0 and lib.rt_gather_source_row(10, 2, 2, 5, 10) < 0


def test_pixel_map_divisors_reproduce_the_quotient():
    """The kernels take item / width and row / tile_rows through a multiplier and a shift the library works out
    (rt_pixel_map_divisor; the reference walks the same pixels with a counter, main.cpp:250-251): equal to the integer
    quotient at every multiple of the divisor, one below it, and a random sample."""
    rng = np.random.default_rng(5)
    for d, n_max in [(2, 4096), (3, 10 ** 6), (7, 2160), (8, 4320), (37, 37 * 41 - 1), (192, 192 * 108), (1920, 1920 * 1080),
                     (3840, 3840 * 2160), (7680, 7680 * 4320), (65535, 2 ** 31 - 1), (2 ** 16, 2 ** 31 - 1), (46341, 2 ** 31 - 1)]:
        got = api.pixel_map_divisor(d, n_max)
        assert got is not None, (d, n_max)
        mul, sh = got
        k = np.arange(0, n_max // d + 1, dtype=np.uint64)
        n = np.unique(np.clip(np.concatenate([k * d, k * d + (d - 1), np.maximum(k * d, 1) - 1,
                                              rng.integers(0, n_max + 1, 20000).astype(np.uint64)]), 0, n_max))
        q = ((n * np.uint64(mul)) >> np.uint64(32)) >> np.uint64(sh)
        assert np.array_equal(q, n // np.uint64(d)), (d, n_max)
    assert api.pixel_map_divisor(1, 100) is None            # the kernels divide (by one)
    assert api.pixel_map_divisor(5, 2 ** 31) is None        # beyond the range the multiplier is proven for
