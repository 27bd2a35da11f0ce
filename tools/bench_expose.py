#!/usr/bin/env python
"""Time rt_expose_device (bloom + pack) on a 4K frame: tools/bench_expose.py [width] [blurs]"""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from raytracer_amd import api
W = int(sys.argv[1]) if len(sys.argv) > 1 else 3840
blurs = int(sys.argv[2]) if len(sys.argv) > 2 else 3
H = W * 9 // 16
ren = api.Renderer(0)
g = torch.Generator(device="cuda").manual_seed(1)
rgb = torch.rand((H, W, 3), dtype=torch.float64, device="cuda", generator=g) * 1.25       # 20 % of the channels saturate
if len(sys.argv) > 3 and sys.argv[3] == "sparse":                                        # only a ceiling light saturates
    rgb *= 0.5
    rgb[: H // 20, W // 3: 2 * W // 3] = 5.0
img = torch.empty((H, W), dtype=torch.int32, device="cuda")
st = torch.cuda.current_stream()
for nb in (0, blurs):
    ren.expose_device(rgb.data_ptr(), W, H, 1.0, nb, img.data_ptr(), st.cuda_stream)
    torch.cuda.synchronize()
    a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    a.record(st)
    for _ in range(3):
        ren.expose_device(rgb.data_ptr(), W, H, 1.0, nb, img.data_ptr(), st.cuda_stream)
    b.record(st)
    torch.cuda.synchronize()
    ms = a.elapsed_time(b) / 3
    taps = H * W * 3 * nb * 1681
    print(f"{W}x{H} blurs={nb}: {ms:.3f} ms" + (f", {2 * taps / ms / 1e9:.2f} FP64 TFLOP/s (mul+add per tap)" if nb else ""))
