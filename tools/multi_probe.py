#!/usr/bin/env python
"""rt_multi_* on the ranks of one machine: frames one at a time (rt_multi_render_image) and two in flight
(rt_multi_render_begin / _end), with the host-side split of a frame's wall-clock into enqueue and wait.
    RT_MULTI_LOOPBACK=1 python tools/multi_probe.py 8 [workload] [frames]      (8 ranks on one GPU)"""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from raytracer_amd import api, synth
n = int(sys.argv[1]); wl = sys.argv[2] if len(sys.argv) > 2 else "c3"; frames = int(sys.argv[3]) if len(sys.argv) > 3 else 20
_, sc = synth.make(wl)
m = api.MultiRenderer(n)
m.upload(sc)
for _ in range(3):
    m.render_image()
t0 = time.perf_counter()
one = [m.render_image() for _ in range(frames)]
t_one = (time.perf_counter() - t0) / frames * 1e3
m.begin()
t0 = time.perf_counter()
enq, wait = [], []
for _ in range(frames):
    m.begin()
    f = m.end()
    enq.append(f["enqueue_ms"]); wait.append(f["wait_ms"])
t_two = (time.perf_counter() - t0) / frames * 1e3
last = m.end()
assert (last["img"] == one[0]["img"]).all() and last["rays"] == one[0]["rays"]
print(f"{wl} ranks {n} rccl {m.uses_rccl()}: one frame at a time {t_one:.3f} ms per frame; two in flight {t_two:.3f} ms per frame "
      f"(enqueue {sum(enq) / len(enq):.3f} ms, wait {sum(wait) / len(wait):.3f} ms of the frame before); rays {last['rays']}")
m.close()
