"""Kept bits per ray of the matrix-core broad phase on the bench scene's own spheres (translated like the upload does):
camera rays and bounce rays.  RT_HIP_LIB selects the library build."""
import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
from raytracer_amd import api, synth, scene as S
ren = api.Renderer(0)
_, sc = synth.make("c3")
k = sc.prims[:, S.PR_KIND].astype(int)
sp = sc.prims[k == S.KIND_SPHERE]
c = sp[:, S.PR_A:S.PR_A + 3].copy(); r = sp[:, S.PR_RADIUS].copy()
T = np.array([0.0, -0.5, -20.0])
c -= T
rng = np.random.default_rng(3)
n = 8192
# camera rays
o1 = np.tile(np.array([0.0, 0.0, 40.0]) - T, (n // 2, 1)) + rng.uniform(-1, 1, (n // 2, 3)) * (1, 1, 0)
px = np.stack([rng.uniform(-48, 48, n // 2), rng.uniform(-27, 27, n // 2), np.full(n // 2, -20.0)], 1) - T
d1 = px - o1
# bounce rays from sphere surfaces
pick = rng.integers(0, len(c), n // 2)
nrm = rng.standard_normal((n // 2, 3)); nrm /= np.linalg.norm(nrm, axis=1, keepdims=True)
o2 = c[pick] + nrm * r[pick][:, None]
d2 = nrm + rng.standard_normal((n // 2, 3)) * 0.7
o = np.concatenate([o1, o2]); d = np.concatenate([d1, d2]); d /= np.linalg.norm(d, axis=1, keepdims=True)
keep = ren.selftest_broad_mfma(o, d, c, r)
oc = o[:, None, :] - c[None, :, :]
hb = (d[:, None, :] * oc).sum(2)
disc = hb * hb - ((oc * oc).sum(2) - (r * r)[None, :])
per = keep.sum(1)
print("kept", int(keep.sum()), "true", int((disc >= 0).sum()), "dropped", int(((disc >= 0) & ~keep).sum()),
      "bits/ray cam %.2f bounce %.2f max %d rays>48: %d all-kept rows: %d" % (per[: n // 2].mean(), per[n // 2:].mean(), per.max(), int((per > 48).sum()), int(keep.all(1).sum())))
fp = keep & (disc < 0)
print("false positives: -disc percentiles 50/90/99/max", np.percentile(-disc[fp], [50, 90, 99, 100]))
