"""Row-tile partition of a frame over the GPUs of one node and the gather to rank 0.

The reference splits a frame over CPU threads by cyclic rows (thread t renders rows t, t+T,
..., /root/reference/src/main.cpp:250) and all threads write into one shared Color array
(main.cpp:303,308).  Here the frame is split into cyclic row TILES (tile k = rows
[k*tile_rows, (k+1)*tile_rows) belongs to rank k % world -- cyclic for the same load-balance
reason), every rank renders its tiles into a packed local buffer, and the shared array is
replaced by one gather to rank 0 (RCCL over xGMI when the process group is "nccl") followed by
a de-interleave.  Pixels are independent and seeded from their frame index, so the gathered
frame is bit-identical to a single-GPU frame.

One process per GPU; torch.distributed supplies the collective, nothing else.
"""
from __future__ import annotations

import torch
import torch.distributed as dist


def rows_of_rank(height: int, tile_rows: int, rank: int, world: int) -> torch.Tensor:
    """Global row indices owned by `rank`, in the order they are packed in its local buffer."""
    rows = []
    k = rank
    while k * tile_rows < height:
        r0 = k * tile_rows
        rows.extend(range(r0, min(r0 + tile_rows, height)))
        k += world
    return torch.tensor(rows, dtype=torch.long)


def max_rows(height: int, tile_rows: int, world: int) -> int:
    return max(len(rows_of_rank(height, tile_rows, r, world)) for r in range(world))


class FrameGatherer:
    """Gathers per-rank packed row buffers into the full frame on rank 0.

    local buffers: tensor [max_rows, width*channels] (rows beyond the rank's own count are padding).
    """

    def __init__(self, height: int, width: int, channels: int, tile_rows: int, rank: int, world: int,
                 device, dtype=torch.float64):
        self.h, self.w, self.c = height, width, channels
        self.tile_rows, self.rank, self.world = tile_rows, rank, world
        self.device, self.dtype = device, dtype
        self.pad_rows = max_rows(height, tile_rows, world)
        self.my_rows = rows_of_rank(height, tile_rows, rank, world)
        self.local = torch.empty((self.pad_rows, width * channels), dtype=dtype, device=device)
        if rank == 0:
            self.frame = torch.empty((height, width * channels), dtype=dtype, device=device)
            self.index = [rows_of_rank(height, tile_rows, r, world).to(device) for r in range(world)]
            self.recv = [torch.empty_like(self.local) for _ in range(world)] if world > 1 else None
        else:
            self.frame, self.index, self.recv = None, None, None
        # a gloo group moves host memory: device buffers are staged through the host around the collective
        # (CPU tests, and rehearsals of the multi-rank flow on one GPU); RCCL ("nccl") takes them as they are
        self.stage = world > 1 and torch.device(device).type != "cpu" and dist.get_backend() == "gloo"

    def gather(self):
        """Collective.  Returns the frame tensor [height, width*channels] on rank 0, None elsewhere."""
        if self.world == 1:
            self.frame.index_copy_(0, self.index[0], self.local[: len(self.index[0])])
            return self.frame
        if self.stage:
            recv = [torch.empty(self.local.shape, dtype=self.dtype) for _ in range(self.world)] if self.rank == 0 else None
            dist.gather(self.local.cpu(), recv, dst=0)
            if self.rank == 0:
                for r in range(self.world):
                    self.recv[r].copy_(recv[r])
        else:
            dist.gather(self.local, self.recv if self.rank == 0 else None, dst=0)
        if self.rank != 0:
            return None
        for r in range(self.world):
            idx = self.index[r]
            self.frame.index_copy_(0, idx, self.recv[r][: len(idx)])
        return self.frame


class FramePipeline:
    """render -> expose -> gather, the sequence main() times (main.cpp:305-315), over the ranks of one node.

    num_blurs == 0 (every timing configuration): Color::to_int works pixel by pixel, so each rank packs the rows it
    rendered and the gather moves 4 bytes per pixel instead of 24.  num_blurs > 0: the bloom reads 20 rows either
    side of a pixel per pass, so the radiance rows are gathered and rank 0 exposes the whole frame.

    render_rows(dst): fill dst[:my_rows] (float64 [rows, W*3]) with this rank's rows, in stream order.
    expose(src, rows, dst, num_blurs): pack src (float64 [rows, W*3]) into dst (int32 [rows, W]), in stream order.
    On a GPU both are thin wrappers of rt_render_rows_device / rt_expose_device; the CPU tests pass oracle stand-ins.
    """

    def __init__(self, height, width, num_blurs, tile_rows, rank, world, device, render_rows, expose):
        self.h, self.w, self.blurs = height, width, int(num_blurs)
        self.rank, self.world = rank, world
        self.render_rows, self.expose = render_rows, expose
        self.n_mine = len(rows_of_rank(height, tile_rows, rank, world))
        if self.blurs == 0:
            self.rad = torch.empty((max_rows(height, tile_rows, world), width * 3), dtype=torch.float64, device=device)
            self.gat = FrameGatherer(height, width, 1, tile_rows, rank, world, device, dtype=torch.int32)
        else:
            self.gat = FrameGatherer(height, width, 3, tile_rows, rank, world, device, dtype=torch.float64)
            self.rad = self.gat.local
            self.img = torch.empty((height, width), dtype=torch.int32, device=device) if rank == 0 else None

    def step(self):
        """One frame.  Collective.  Returns the packed frame int32[H, W] on rank 0, None elsewhere."""
        self.render_rows(self.rad)
        if self.blurs == 0:
            if self.n_mine:
                self.expose(self.rad, self.n_mine, self.gat.local, 0)
            return self.gat.gather()
        frame = self.gat.gather()
        if self.rank != 0:
            return None
        self.expose(frame, self.h, self.img, self.blurs)
        return self.img


def sum_over_ranks(value: int, device, world: int) -> int:
    """Ray counters: the reference sums long num_rays[T] (main.cpp:321-324); here one all-reduce."""
    if world == 1:
        return int(value)
    t = torch.tensor([int(value)], dtype=torch.int64, device="cpu" if dist.get_backend() == "gloo" else device)
    dist.all_reduce(t, op=dist.ReduceOp.SUM)
    return int(t.item())
