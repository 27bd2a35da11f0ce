"""GPU parity of expose() on the device (rt_expose_device: bloom + Color::to_int pack,
/root/reference/src/main.cpp:186-242, src/geometry.cpp:15-23) against the CPU oracle's expose, which the
golden tests pin to the compiled reference.  Bar: every packed pixel equal (integer output)."""
import os

import numpy as np
import pytest

import pyoracle as po
from raytracer_amd import api, scene as S

torch = pytest.importorskip("torch")
pytestmark = pytest.mark.gpu


def _expose_gpu(renderer, rgb, exposure, blurs):
    H, W, _ = rgb.shape
    d_rgb = torch.from_numpy(np.ascontiguousarray(rgb)).to("cuda:0")
    d_img = torch.full((H, W), -7, dtype=torch.int32, device="cuda:0")
    torch.cuda.synchronize()
    renderer.expose_device(d_rgb.data_ptr(), W, H, exposure, blurs, d_img.data_ptr(), torch.cuda.current_stream().cuda_stream)
    torch.cuda.synchronize()
    assert torch.equal(d_rgb.cpu(), torch.from_numpy(np.ascontiguousarray(rgb))) or np.isnan(rgb).any()   # input untouched
    return d_img.cpu().numpy()


@pytest.mark.parametrize("blurs", [0, 1, 3])
def test_expose_scene_json_frame(renderer, golden_dir, blurs):
    """A real frame (scene.json rendered on the GPU; 41.6 % of its pixels saturate, so the bloom is busy)."""
    sc = S.read_json_scene(os.path.join(golden_dir, "scene.json"))
    renderer.upload(sc)
    rgb = renderer.render()["rgb"]
    want, _ = po.expose(rgb, sc.header["exposure"], blurs)
    got = _expose_gpu(renderer, rgb, sc.header["exposure"], blurs)
    assert np.array_equal(got, want), f"{int((got != want).sum())} packed pixels differ"
    assert np.array_equal(got, api.expose(rgb, sc.header["exposure"], blurs))     # and the numpy host mirror


@pytest.mark.parametrize("shape,exposure,blurs", [((1, 1), 1.0, 2), ((7, 301), 0.5, 1), ((45, 16), 2.0, 2), ((33, 47), 1.3, 0),
                                                  ((130, 70), 0.7, 3)])
def test_expose_ragged_frames_and_edge_values(renderer, shape, exposure, blurs):
    """Frames smaller than the bloom radius, not multiples of the tile, with the values Color::to_int meets at its
    edges: negative, just below/above an integer step, huge, infinite, NaN (converted as x86 does: INT_MIN)."""
    H, W = shape
    rng = np.random.default_rng(H * 1000 + W)
    rgb = rng.uniform(0, 2.0 / exposure, (H, W, 3))
    flat = rgb.reshape(-1)
    special = np.array([0.0, -0.0, -0.3, 1.0 / exposure, np.nextafter(1.0 / exposure, 0), np.nextafter(1.0 / exposure, 9),
                        1e300, -1e300, np.inf, -np.inf, np.nan, 3e9, -3e9, 1e-320, 254.999999 / 255 / exposure])
    idx = rng.choice(flat.size, size=min(flat.size, special.size), replace=False)
    flat[idx] = special[: idx.size]
    want, _ = po.expose(rgb, exposure, blurs)
    got = _expose_gpu(renderer, rgb, exposure, blurs)
    assert np.array_equal(got, want), f"{int((got != want).sum())} of {got.size} packed pixels differ"


def test_expose_rows_before_gather(renderer):
    """num_blurs = 0: exposing a rank's packed rows gives the same pixels as exposing the whole frame."""
    rng = np.random.default_rng(9)
    rgb = rng.uniform(0, 3, (64, 50, 3))
    whole = _expose_gpu(renderer, rgb, 0.8, 0)
    rows = np.r_[8:16, 40:48]
    part = _expose_gpu(renderer, rgb[rows], 0.8, 0)
    assert np.array_equal(part, whole[rows])


def _expose_by_ranks(renderer, rgb, exposure, blurs, tile_rows, world):
    """The frame exposed rank by rank as FramePipeline's halo mode does it on `world` GPUs: bits of every rank's rows,
    concatenated as the all-gather delivers them, then rt_expose_rows_device per rank; rows put back in place."""
    from raytracer_amd.dist import max_rows, rows_of_rank
    H, W, _ = rgb.shape
    st = torch.cuda.current_stream().cuda_stream
    pad = max_rows(H, tile_rows, world)
    rows = [rows_of_rank(H, tile_rows, r, world).numpy() for r in range(world)]
    local = [torch.from_numpy(np.ascontiguousarray(rgb[rr])).to("cuda:0") if len(rr) else torch.zeros((1, W, 3), dtype=torch.float64, device="cuda:0")
             for rr in rows]
    bits_all = torch.full((world, pad, W), 0xff, dtype=torch.uint8, device="cuda:0")     # padding rows hold junk
    for r in range(world):
        bits = torch.empty((max(1, len(rows[r])), W), dtype=torch.uint8, device="cuda:0")
        renderer.expose_bits_device(local[r].data_ptr(), W, len(rows[r]), exposure, bits.data_ptr(), st)
        bits_all[r, : len(rows[r])] = bits[: len(rows[r])]
    out = np.full((H, W), -7, dtype=np.int32)
    for r in range(world):
        img = torch.full((max(1, len(rows[r])), W), -7, dtype=torch.int32, device="cuda:0")
        renderer.expose_rows_device(local[r].data_ptr(), bits_all.data_ptr(), W, H, tile_rows, r, world, pad, exposure, blurs,
                                    img.data_ptr(), st)
        torch.cuda.synchronize()
        out[rows[r]] = img.cpu().numpy()[: len(rows[r])]
    return out


@pytest.mark.parametrize("shape,tile_rows,world,blurs", [((130, 70), 8, 2, 3), ((130, 70), 8, 8, 2), ((97, 33), 5, 3, 1),
                                                         ((64, 40), 1, 4, 2), ((40, 50), 8, 8, 1), ((200, 48), 25, 8, 3),
                                                         ((33, 47), 8, 2, 0)])
def test_expose_by_ranks_with_halo_equals_whole_frame(renderer, shape, tile_rows, world, blurs):
    """expose() before the gather: every rank blooms the block rows its own rows depend on, from all ranks'
    over-exposure bits, and packs its rows.  Against the oracle's expose of the whole frame, every pixel equal;
    ragged frames, ranks without rows (40 rows in tiles of 8 over 8 ranks), one-row tiles, band partitions."""
    H, W = shape
    rng = np.random.default_rng(H * 131 + W + world)
    exposure = 0.9
    rgb = rng.uniform(0, 1.15 / exposure, (H, W, 3))            # about an eighth of the channels saturate
    rgb[H // 3, W // 2] = [np.nan, np.inf, -1.0]
    want, _ = po.expose(rgb, exposure, blurs)
    got = _expose_by_ranks(renderer, rgb, exposure, blurs, tile_rows, world)
    assert np.array_equal(got, want), f"{int((got != want).sum())} of {got.size} packed pixels differ"
    assert np.array_equal(got, _expose_gpu(renderer, rgb, exposure, blurs))


def test_expose_by_ranks_real_frame(renderer, golden_dir):
    """scene.json (3 blur passes, 41.6 % of the pixels saturated) split over 4 ranks in 8-row tiles."""
    sc = S.read_json_scene(os.path.join(golden_dir, "scene.json"))
    renderer.upload(sc)
    rgb = renderer.render()["rgb"]
    want, _ = po.expose(rgb, sc.header["exposure"], 3)
    got = _expose_by_ranks(renderer, rgb, sc.header["exposure"], 3, 8, 4)
    assert np.array_equal(got, want), f"{int((got != want).sum())} packed pixels differ"


def test_render_image_equals_render_then_expose(renderer, golden_dir):
    """rt_render_image (render + device expose, packed frame to the host) against the two-step path, bloom included."""
    sc = S.read_json_scene(os.path.join(golden_dir, "small_room.scene.json"))
    renderer.upload(sc)
    g = renderer.render()
    want, _ = po.expose(g["rgb"], sc.header["exposure"], sc.header["num_blurs"])
    got = renderer.render_image()
    assert got["rays"] == g["rays"]
    assert np.array_equal(got["img"], want)


@pytest.mark.parametrize("flags", [[], ["--host-expose"]])
def test_rt_main_writes_the_reference_image(golden_dir, tmp_path, flags):
    """The C++ host program end to end: scene.json -> PNG in the reference's RNG order (num_threads = 4, 3 bloom
    passes) must decode to the golden pixels of the compiled reference, with expose() on the device and on the host."""
    import json
    import subprocess
    from PIL import Image
    exe = os.path.join(os.path.dirname(api.HERE), "raytracer_amd", "rt_main")
    out = str(tmp_path / "test.png")
    r = subprocess.run([exe, os.path.join(golden_dir, "scene.json"), out, "--reference-order"] + flags,
                       capture_output=True, text=True, timeout=300)
    assert r.returncode == 0, r.stderr
    assert "Time: " in r.stdout and "Rays: " in r.stdout
    gold = [e for e in json.load(open(os.path.join(golden_dir, "golden.json")))
            if e["scene"] == "scene.json" and e["threads"] == 4 and e["blurs"] == 3][0]
    ref8 = np.load(os.path.join(golden_dir, gold["pixels"]))["rgb8"]
    got8 = np.array(Image.open(out).convert("RGB"))
    assert np.array_equal(got8, ref8)
