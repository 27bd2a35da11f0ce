#!/usr/bin/env python
"""Frame interval with one and with two frames in flight on one GPU (two contexts, two streams, frames alternate):
tools/two_frames_probe.py [workload] [frames]"""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from raytracer_amd import api, synth
wl = sys.argv[1] if len(sys.argv) > 1 else "c3"
K = int(sys.argv[2]) if len(sys.argv) > 2 else 60
_, sc = synth.make(wl)
H, W = sc.height, sc.width
for nctx in (1, 2, 3):
    rens = [api.Renderer(0) for _ in range(nctx)]
    for r in rens:
        r.upload(sc)
    streams = [torch.cuda.Stream() for _ in range(nctx)]
    rad = [torch.empty((H, W * 3), dtype=torch.float64, device="cuda") for _ in range(nctx)]
    img = [torch.empty((H, W), dtype=torch.int32, device="cuda") for _ in range(nctx)]
    def step(i):
        k = i % nctx
        rens[k].render_rows_device(rad[k].data_ptr(), 8, 0, 1, stream=streams[k].cuda_stream)
        rens[k].expose_device(rad[k].data_ptr(), W, H, 1.0, 0, img[k].data_ptr(), streams[k].cuda_stream)
    for i in range(2 * nctx):
        step(i)
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for i in range(K):
        step(i)
    torch.cuda.synchronize()
    ms = (time.perf_counter() - t0) * 1e3 / K
    same = all(torch.equal(img[0], im) for im in img[1:])
    print(f"{wl}: {nctx} frame(s) in flight: {ms:.3f} ms per frame (frames equal: {same})", flush=True)
    for r in rens:
        r.finish(); r.close()
