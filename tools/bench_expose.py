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

# one rank's share when `world` ranks expose before the gather (bits -> [all-gather, not timed here] -> bloom with
# halo block rows -> pack of the rank's rows): tools/bench_expose.py W blurs dense|sparse world [tile_rows]
if len(sys.argv) > 4:
    from raytracer_amd.dist import max_rows, rows_of_rank
    world = int(sys.argv[4])
    for tile in ([int(sys.argv[5])] if len(sys.argv) > 5 else [8, (H + world - 1) // world]):
        pad = max_rows(H, tile, world)
        bits_all = torch.zeros((world, pad, W), dtype=torch.uint8, device="cuda")
        for r in range(world):
            rr = rows_of_rank(H, tile, r, world).cuda()
            loc = rgb[rr].contiguous()
            ren.expose_bits_device(loc.data_ptr(), W, len(rr), 1.0, bits_all[r].data_ptr(), st.cuda_stream)
        rr = rows_of_rank(H, tile, 0, world).cuda()
        loc = rgb[rr].contiguous()
        out = torch.empty((len(rr), W), dtype=torch.int32, device="cuda")
        bits = torch.empty((len(rr), W), dtype=torch.uint8, device="cuda")

        def one():
            ren.expose_bits_device(loc.data_ptr(), W, len(rr), 1.0, bits.data_ptr(), st.cuda_stream)
            ren.expose_rows_device(loc.data_ptr(), bits_all.data_ptr(), W, H, tile, 0, world, pad, 1.0, blurs, out.data_ptr(), st.cuda_stream)
        one()
        torch.cuda.synchronize()
        a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        a.record(st)
        for _ in range(3):
            one()
        b.record(st)
        torch.cuda.synchronize()
        same = torch.equal(out, img[rr])
        print(f"rank 0 of {world}, tiles of {tile} rows: {a.elapsed_time(b) / 3:.3f} ms per frame, rows equal to the whole-frame expose: {same}")
