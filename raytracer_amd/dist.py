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


def default_tile_rows(height: int, world: int, num_blurs: int) -> int:
    """Tile height of the partition.  Without blur passes: 8-row tiles dealt out cyclically (load balance: rows through
    dense geometry cost more rays).  With blur passes every rank blooms what its rows depend on, 20 rows either side per
    pass: tiles that thin would make it bloom nearly the whole frame, so each rank gets ONE band of rows (4K, 3 passes,
    8 ranks: 2.3 ms per rank against 10.9 ms with 8-row tiles)."""
    if num_blurs > 0 and world > 1:
        return (height + world - 1) // world
    return 8


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
        # two send buffers: `local` is the one to fill next (see gather_begin)
        self._bufs = [torch.empty((self.pad_rows, width * channels), dtype=dtype, device=device) for _ in range(2)]
        self._cur, self._work, self._sent = 0, None, None
        self.local = self._bufs[0]
        if rank == 0:
            self.frame = torch.empty((height, width * channels), dtype=dtype, device=device)
            # every rank's buffer lands in one tensor [world, pad_rows, W*c]; global row y is row src_row[y] of its
            # flattened view, so the de-interleave is ONE index_select whatever the number of ranks.  Two such tensors,
            # used alternately like the send buffers: the gather of frame i+1 (started before frame i is put in row
            # order when the caller pipelines) never writes what frame i's index_select still reads, whatever the
            # backend's stream ordering.
            self._recv_all = [torch.empty((world, self.pad_rows, width * channels), dtype=dtype, device=device) for _ in range(2)] if world > 1 else None
            self._recv_cur, self._recv_sent = 0, 0
            self.recv_all = self._recv_all[0] if world > 1 else None
            self.recv = [self.recv_all[r] for r in range(world)] if world > 1 else None
            y = torch.arange(height, dtype=torch.long)
            tile = y // tile_rows
            self.src_row = ((tile % world) * self.pad_rows + (tile // world) * tile_rows + y % tile_rows).to(device)
        else:
            self.frame, self.recv_all, self.recv, self.src_row = None, None, None, None
        # a gloo group moves host memory: device buffers are staged through the host around the collective
        # (CPU tests, and rehearsals of the multi-rank flow on one GPU); RCCL ("nccl") takes them as they are
        self.stage = world > 1 and torch.device(device).type != "cpu" and dist.get_backend() == "gloo"

    def gather(self):
        """Collective.  Returns the frame tensor [height, width*channels] on rank 0, None elsewhere."""
        self.gather_begin()
        return self.gather_end()

    # The gather in two halves, for callers that render the next frame while this one travels: gather_begin() starts the
    # collective on the buffer `local` names and switches `local` to the other buffer; gather_end() waits for it (the
    # current stream does, not the host, under RCCL) and puts the rows in frame order.  One gather in flight at a time.
    def gather_begin(self):
        buf = self.local
        self._sent = buf
        self._work = None
        if self.world > 1 and self.rank == 0:
            self._recv_sent = self._recv_cur
            self.recv_all = self._recv_all[self._recv_cur]
            self.recv = [self.recv_all[r] for r in range(self.world)]
            self._recv_cur ^= 1
        if self.world > 1:
            if self.stage:
                recv = [torch.empty(buf.shape, dtype=self.dtype) for _ in range(self.world)] if self.rank == 0 else None
                dist.gather(buf.cpu(), recv, dst=0)
                if self.rank == 0:
                    for r in range(self.world):
                        self.recv[r].copy_(recv[r])
            else:
                self._work = dist.gather(buf, self.recv if self.rank == 0 else None, dst=0, async_op=True)
        self._cur ^= 1
        self.local = self._bufs[self._cur]

    def gather_end(self):
        if self._work is not None:
            self._work.wait()
            self._work = None
        if self.rank != 0:
            return None
        if self.world == 1:
            # one rank owns every tile, in order: its buffer IS the frame (a view of the send buffer just filled; the
            # next step fills the other one)
            return self._sent[: self.h]
        else:
            torch.index_select(self._recv_all[self._recv_sent].view(self.world * self.pad_rows, -1), 0, self.src_row, out=self.frame)
        return self.frame


class FramePipeline:
    """render -> expose -> gather, the sequence main() times (main.cpp:305-315), over the ranks of one node.

    num_blurs == 0 (every timing configuration): Color::to_int works pixel by pixel, so each rank packs the rows it
    rendered and the gather moves 4 bytes per pixel instead of 24.
    num_blurs > 0: the bloom reads 20 rows either side of a pixel per pass.  Its input is three bits per pixel (is the
    channel over-exposed, main.cpp:198-205), so with `expose_bits` / `expose_rows` given the ranks all-gather those
    bits (1 byte per pixel), every rank blooms the block rows its own rows depend on and packs its rows, and the gather
    again moves 4 bytes per pixel; the blur passes, the bulk of expose(), are shared out over the ranks.  Without the
    two callables the radiance rows are gathered and rank 0 exposes the whole frame.

    render_rows(dst): fill dst[:my_rows] (float64 [rows, W*3]) with this rank's rows, in stream order.
    expose(src, rows, dst, num_blurs): pack src (float64 [rows, W*3]) into dst (int32 [rows, W]), in stream order.
    expose_bits(src, rows, dst): dst (uint8 [rows, W]) = over-exposure bits of src (float64 [rows, W*3]).
    expose_rows(src, bits_all, pad_rows, dst): bits_all (uint8 [world*pad_rows, W]) holds every rank's bits; pack this
        rank's rows src (float64 [rows, W*3]) + bloom into dst (int32 [rows, W]).
    On a GPU they are thin wrappers of rt_render_rows_device / rt_expose_device / rt_expose_bits_device /
    rt_expose_rows_device; the CPU tests pass oracle stand-ins.
    """

    def __init__(self, height, width, num_blurs, tile_rows, rank, world, device, render_rows, expose,
                 expose_bits=None, expose_rows=None, pipelined=False):
        self.h, self.w, self.blurs = height, width, int(num_blurs)
        self.rank, self.world = rank, world
        self.render_rows, self.expose = render_rows, expose
        self.expose_bits, self.expose_rows = expose_bits, expose_rows
        self.n_mine = len(rows_of_rank(height, tile_rows, rank, world))
        self.halo = self.blurs > 0 and world > 1 and expose_bits is not None and expose_rows is not None
        # pipelined: step() starts the gather of its frame and returns the PREVIOUS frame, whose gather ran while this
        # one was rendered (flush() returns the last); only where packed rows are gathered
        self.pipelined = bool(pipelined) and (self.blurs == 0 or self.halo)
        self.pending = False
        if self.blurs == 0 or self.halo:
            self.rad = torch.empty((max_rows(height, tile_rows, world), width * 3), dtype=torch.float64, device=device)
            self.gat = FrameGatherer(height, width, 1, tile_rows, rank, world, device, dtype=torch.int32)
        else:
            self.gat = FrameGatherer(height, width, 3, tile_rows, rank, world, device, dtype=torch.float64)
            self.rad = None                                  # rendered straight into the gatherer's send buffer
            self.img = torch.empty((height, width), dtype=torch.int32, device=device) if rank == 0 else None
        if self.halo:
            pad = self.gat.pad_rows
            self.bits = torch.zeros((pad, width), dtype=torch.uint8, device=device)
            self.bits_all = torch.empty((world * pad, width), dtype=torch.uint8, device=device)

    def _all_gather_bits(self):
        if dist.get_backend() == "gloo":                     # moves host memory (CPU tests, one-GPU rehearsals)
            host = [torch.empty(self.bits.shape, dtype=torch.uint8) for _ in range(self.world)]
            dist.all_gather(host, self.bits.cpu())
            self.bits_all.copy_(torch.cat(host, 0))
        else:
            dist.all_gather_into_tensor(self.bits_all, self.bits)

    def _gather_packed(self):
        if not self.pipelined:
            return self.gat.gather()
        prev = self.gat.gather_end() if self.pending else None
        self.gat.gather_begin()
        self.pending = True
        return prev

    def flush(self):
        """Pipelined mode: wait for the last frame's gather.  Returns it on rank 0 (None elsewhere, or nothing pending)."""
        if not self.pending:
            return None
        self.pending = False
        return self.gat.gather_end()

    def step(self):
        """One frame.  Collective.  Returns the packed frame int32[H, W] on rank 0, None elsewhere (pipelined: the
        previous step's frame, None on the first step)."""
        self.render_rows(self.rad if self.rad is not None else self.gat.local)
        if self.blurs == 0:
            if self.n_mine:
                self.expose(self.rad, self.n_mine, self.gat.local, 0)
            return self._gather_packed()
        if self.halo:
            if self.n_mine:
                self.expose_bits(self.rad, self.n_mine, self.bits)
            self._all_gather_bits()
            if self.n_mine:
                self.expose_rows(self.rad, self.bits_all, self.gat.pad_rows, self.gat.local)
            return self._gather_packed()
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
