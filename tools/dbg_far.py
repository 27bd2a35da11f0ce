#!/usr/bin/env python
"""Compare GPU and oracle on the small room seen from a camera 1e200 away (debug aid)."""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "oracle"))
import numpy as np
import pyoracle as po
from raytracer_amd import api, scene as S
cam = sys.argv[1] if len(sys.argv) > 1 else "[0.5, -0.25, 1e200]"
text = open(os.path.join(ROOT, "tests/golden/small_room.scene.json")).read()
sc = S.loads(text.replace('{"num_threads"', '{"camera": %s, "num_threads"' % cam, 1))
o = po.render(sc, policy=po.POLICY_DEV, rng_mode=po.RNG_PIXEL, want_hits=True)
ren = api.Renderer(0)
for mode in (api.MODE_PIPELINE, api.MODE_SINGLE_KERNEL):
    ren.set_option(api.OPT_MODE, mode)
    ren.upload(sc)
    g = ren.render(want_hits=True)
    bad = np.argwhere(g["hits"] != o["hits"])
    a = g["rgb"].view(np.uint64); b = o["rgb"].view(np.uint64)
    badc = np.argwhere((a != b) & ~(np.isnan(g["rgb"]) & np.isnan(o["rgb"])))
    print("mode", mode, "rays", g["rays"], int(o["rays"].sum()), "hit diffs", len(bad), "colour diffs", len(badc))
    for y, x, s_ in bad[:6]:
        print("  hit", y, x, s_, "gpu", g["hits"][y, x, s_], "oracle", o["hits"][y, x, s_])
    for y, x, c in badc[:6]:
        print("  rgb", y, x, c, g["rgb"][y, x, c], o["rgb"][y, x, c])
print("hit id histogram (oracle):", np.unique(o["hits"], return_counts=True))
