#!/usr/bin/env python3
"""Grid walk (RT_OPT_ACCEL = 1) against the scan of every record (0) on one workload: same bits?  how much faster?

    python tools/accel_probe.py c3 [width] [frames]
Prints, per mode, frame kernel time (overlapped sub-pipelines), serial kernel time and the broad-phase share
(per-launch HIP events), candidates per ray and overflow rays; then whether the two frames are bit-identical."""
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)

import numpy as np  # noqa: E402
import torch  # noqa: E402

from raytracer_amd import api, synth  # noqa: E402


def main():
    wl = sys.argv[1] if len(sys.argv) > 1 else "c3"
    width = int(sys.argv[2]) if len(sys.argv) > 2 and int(sys.argv[2]) > 0 else None
    frames = int(sys.argv[3]) if len(sys.argv) > 3 else 3
    _, sc = synth.make(wl, width=width)
    ren = api.Renderer(0)
    H, W = sc.height, sc.width
    rgb = torch.empty((H, W, 3), dtype=torch.float64, device="cuda:0")
    hits = torch.empty((H, W, sc.spp), dtype=torch.int32, device="cuda:0")
    st = torch.cuda.current_stream().cuda_stream
    out = {}
    for accel in (0, 1):
        ren.set_option(api.OPT_ACCEL, accel)
        ren.upload(sc)
        res = []
        for detail in (0, 1):
            ren.set_option(api.OPT_TIMING_DETAIL, detail)
            ms = []
            for _ in range(frames + 1):
                ren.render_rows_device(rgb.data_ptr(), H, 0, 1, hits.data_ptr(), st)
                rays = ren.finish()
                ms.append(ren.timing())
            res.append(ms[1:])
        ren.set_option(api.OPT_TIMING_DETAIL, 0)
        t0, t1 = res
        k0 = float(np.mean([t["kernel_ms"] for t in t0]))
        k1 = float(np.mean([t["kernel_ms"] for t in t1]))
        b1 = float(np.mean([t["broad_ms"] for t in t1]))
        t = t1[-1]
        print(f"{wl} {W}x{H} accel={accel}: frame {k0:.3f} ms ({rays / k0 / 1e3:.0f} Mrays/s), serial {k1:.3f} ms, broad phase {b1:.3f} ms, "
              f"rays {rays}, candidates/ray {t['candidates'] / max(1, t['rays']):.2f}, overflow rays {t['overflow_rays']}", flush=True)
        out[accel] = (rgb.cpu().numpy().copy(), hits.cpu().numpy().copy(), rays)
    a, b = out[0], out[1]
    same = (a[0].view(np.uint64) == b[0].view(np.uint64)) | (np.isnan(a[0]) & np.isnan(b[0]))
    print(f"bit-identical radiance: {bool(same.all())} ({int((~same).sum())} words differ), hit ids equal: {bool((a[1] == b[1]).all())}, "
          f"rays equal: {a[2] == b[2]}", flush=True)
    if not (same.all() and (a[1] == b[1]).all() and a[2] == b[2]):
        bad = np.argwhere(~same)
        print("first differing pixels (y, x, channel):", bad[:8].tolist())
        sys.exit(1)


if __name__ == "__main__":
    main()
