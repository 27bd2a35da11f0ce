"""The N > 1 path on CPU: two gloo ranks partition a frame into cyclic row tiles, each fills its packed
local buffer (from the oracle, standing in for the GPU renderer), and the gather + de-interleave on rank 0
must give the single-rank frame bit for bit."""
import os
import sys

import numpy as np
import pytest
import torch
import torch.distributed as dist
import torch.multiprocessing as mp

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _worker(rank, world, port, tile_rows, out_path):
    sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "oracle"))
    import pyoracle as po
    from raytracer_amd import synth
    from raytracer_amd.dist import FrameGatherer, rows_of_rank, sum_over_ranks
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    _, sc = synth.make("c2", width=64)           # 64 x 36: 36 rows, not a multiple of tile_rows * world
    H, W = sc.height, sc.width
    gat = FrameGatherer(H, W, 3, tile_rows, rank, world, torch.device("cpu"))
    mine = rows_of_rank(H, tile_rows, rank, world)
    full = po.render(sc, policy=po.POLICY_DEV, rng_mode=po.RNG_PIXEL, workers=2)
    gat.local.zero_()
    gat.local[: len(mine)] = torch.from_numpy(full["rgb"][mine.numpy()].reshape(len(mine), W * 3))
    frame = gat.gather()
    rays = sum_over_ranks(100 + rank, torch.device("cpu"), world)
    if rank == 0:
        ok = np.array_equal(frame.numpy().reshape(H, W, 3).view(np.uint64), full["rgb"].view(np.uint64))
        open(out_path, "w").write(f"{int(ok)} {rays}")
    dist.barrier()
    dist.destroy_process_group()


def _pipeline_worker(rank, world, port, tile_rows, blurs, use_halo, pipelined, out_path):
    """render -> expose -> gather (FramePipeline) with the oracle standing in for both device calls."""
    sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "oracle"))
    import pyoracle as po
    from raytracer_amd import synth
    from raytracer_amd.dist import FramePipeline, rows_of_rank
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    _, sc = synth.make("c2", width=64)
    H, W = sc.height, sc.width
    exposure = float(sc.header["exposure"]) * 3.0     # brighter: part of the frame saturates, so the bloom has input
    mine = rows_of_rank(H, tile_rows, rank, world).numpy()
    full = po.render(sc, policy=po.POLICY_DEV, rng_mode=po.RNG_PIXEL, workers=2)

    def render_rows(dst):
        dst.zero_()
        dst[: len(mine)] = torch.from_numpy(full["rgb"][mine].reshape(len(mine), W * 3))

    def expose(src, rows, dst, nb):
        img, _ = po.expose(src[:rows].numpy().reshape(rows, W, 3), exposure, nb, workers=2)
        dst[:rows] = torch.from_numpy(img)

    def expose_bits(src, rows, dst):
        v = src[:rows].numpy().reshape(rows, W, 3) * exposure > 1.0
        dst[:rows] = torch.from_numpy((v[..., 0] * 1 + v[..., 1] * 2 + v[..., 2] * 4).astype(np.uint8))

    def expose_rows(src, bits_all, pad_rows, dst):
        # stand-in for rt_expose_rows_device: a frame whose foreign rows merely reproduce their over-exposure bits
        # (2/exposure where a bit is set) has the same B0, hence the same bloom, as the real frame
        frame = np.zeros((H, W, 3))
        ba = bits_all.numpy().reshape(world, pad_rows, W)
        for r in range(world):
            rr = rows_of_rank(H, tile_rows, r, world).numpy()
            b = ba[r, : len(rr)]
            frame[rr] = np.stack([(b >> k) & 1 for k in range(3)], -1) * (2.0 / exposure)
        frame[mine] = src[: len(mine)].numpy().reshape(len(mine), W, 3)
        img, _ = po.expose(frame, exposure, blurs, workers=2)
        dst[: len(mine)] = torch.from_numpy(img[mine])

    halo = dict(expose_bits=expose_bits, expose_rows=expose_rows) if use_halo else {}
    pipe = FramePipeline(H, W, blurs, tile_rows, rank, world, torch.device("cpu"), render_rows, expose, pipelined=pipelined, **halo)
    assert pipe.halo == (use_halo and blurs > 0)
    if pipelined:
        # the gather of a frame runs while the next one is rendered: step() hands back the previous frame.  The second
        # frame is rendered brighter so that a mixed-up buffer shows.
        first = pipe.step()
        assert first is None
        want0, _ = po.expose(full["rgb"], exposure, blurs, workers=2)
        full["rgb"] = full["rgb"] * 1.5
        second = pipe.step()
        if rank == 0:
            assert np.array_equal(second.numpy().reshape(H, W), want0)
        got = pipe.flush()
        assert pipe.flush() is None
    else:
        got = pipe.step()
        got = pipe.step()                              # buffers are reused frame after frame
    if rank == 0:
        want, _ = po.expose(full["rgb"], exposure, blurs, workers=2)
        ok = np.array_equal(got.numpy().reshape(H, W), want) and len(np.unique(want)) > 10
        open(out_path, "w").write(str(int(ok)))
    dist.barrier()
    dist.destroy_process_group()


@pytest.mark.parametrize("tile_rows,blurs,halo,pipelined", [(8, 0, False, False), (5, 0, False, False), (8, 1, False, False),
                                                           (8, 1, True, False), (5, 2, True, False), (8, 0, False, True),
                                                           (5, 1, True, True)])
def test_two_rank_render_expose_gather(tmp_path, tile_rows, blurs, halo, pipelined):
    """Packed 8-bit rows gathered (num_blurs = 0), radiance rows gathered + rank-0 bloom, and over-exposure bits
    all-gathered + every rank blooming its own rows (num_blurs > 0) all give the single-process frame; so does the
    pipelined form, in which a frame's gather overlaps the next frame's render."""
    out = str(tmp_path / "res.txt")
    port = 31500 + (os.getpid() % 2000) + tile_rows + 16 * blurs + 64 * int(halo) + 128 * int(pipelined)
    mp.spawn(_pipeline_worker, args=(2, port, tile_rows, blurs, halo, pipelined, out), nprocs=2, join=True)
    assert open(out).read() == "1"


@pytest.mark.parametrize("tile_rows", [1, 8])
def test_two_rank_gather_equals_single_frame(tmp_path, tile_rows):
    out = str(tmp_path / "res.txt")
    port = 29500 + (os.getpid() % 2000) + tile_rows
    mp.spawn(_worker, args=(2, port, tile_rows, out), nprocs=2, join=True)
    ok, rays = open(out).read().split()
    assert ok == "1" and rays == "201"


def test_row_partition_covers_every_row_once():
    from raytracer_amd.dist import rows_of_rank, max_rows
    from raytracer_amd import api
    for H, t, world in ((2160, 8, 8), (1081, 8, 3), (5, 4, 4), (36, 1, 2)):
        rows = torch.cat([rows_of_rank(H, t, r, world) for r in range(world)])
        assert sorted(rows.tolist()) == list(range(H))
        assert max_rows(H, t, world) == len(rows_of_rank(H, t, 0, world))
        for r in range(world):
            assert api.rows_for_rank(H, t, r, world) == len(rows_of_rank(H, t, r, world))
