#!/usr/bin/env python
"""What one rank of an N-GPU run does, timed on one GPU: render + pack of rank 0's row tiles for N = 1, 2, 4, 8
(no gather).  tools/scaling_probe.py [workload] -- prints ms per frame and the speed-up over N = 1."""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from raytracer_amd import api, synth
from raytracer_amd.dist import max_rows
wl = sys.argv[1] if len(sys.argv) > 1 else "c3"
_, sc = synth.make(wl)
ren = api.Renderer(0)
ren.upload(sc)
H, W = sc.height, sc.width
st = torch.cuda.current_stream()
base = None
for world in (1, 2, 4, 8):
    rows = max_rows(H, 8, world)
    rad = torch.empty((rows, W * 3), dtype=torch.float64, device="cuda")
    img = torch.empty((rows, W), dtype=torch.int32, device="cuda")
    n = api.rows_for_rank(H, 8, 0, world)
    def step():
        ren.render_rows_device(rad.data_ptr(), 8, 0, world, stream=st.cuda_stream)
        ren.expose_device(rad.data_ptr(), W, n, 1.0, 0, img.data_ptr(), st.cuda_stream)
    for _ in range(3):
        step()
    torch.cuda.synchronize()
    K = 20
    t0 = time.perf_counter()
    for _ in range(K):
        step()
    torch.cuda.synchronize()
    ms = (time.perf_counter() - t0) * 1e3 / K
    base = base or ms
    print(f"{wl} N={world}: rank 0 renders {n} rows in {ms:.3f} ms/frame, speed-up {base / ms:.2f} (ideal {world})", flush=True)
