#!/usr/bin/env python
"""Render rank 0's rows of an N-rank partition a few times (for a kernel trace): tools/probe_once.py N [workload]"""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from raytracer_amd import api, synth
from raytracer_amd.dist import max_rows
world = int(sys.argv[1]); wl = sys.argv[2] if len(sys.argv) > 2 else "c3"
_, sc = synth.make(wl)
ren = api.Renderer(0); ren.upload(sc)
rad = torch.empty((max_rows(sc.height, 8, world), sc.width * 3), dtype=torch.float64, device="cuda")
for _ in range(3):
    ren.render_rows_device(rad.data_ptr(), 8, 0, world, stream=torch.cuda.current_stream().cuda_stream)
    ren.finish()
print(ren.timing())
