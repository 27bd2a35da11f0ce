"""How loose is the matrix-core broad phase?  Rays against small spheres at the bench scene's magnitudes: among the pairs
it keeps, the distribution of the exact discriminant below zero.  RT_HIP_LIB selects the library build."""
import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
from raytracer_amd import api
ren = api.Renderer(0)
rng = np.random.default_rng(5)
n_rec, n_rays = 1024, 8192
c = rng.uniform(-20, 20, (n_rec, 3)); c[:, 2] -= 0.0
r = rng.uniform(0.36, 0.72, n_rec)
o = rng.uniform(-20, 20, (n_rays, 3)); o[: n_rays // 4] = (0.0, 0.0, 60.0)
pick = rng.integers(0, n_rec, n_rays)
tgt = rng.standard_normal((n_rays, 3)); tgt /= np.linalg.norm(tgt, axis=1, keepdims=True)
aim = c[pick] + tgt * (r[pick] * rng.uniform(0.5, 1.5, n_rays))[:, None]
d = aim - o; d /= np.linalg.norm(d, axis=1, keepdims=True)
keep = ren.selftest_broad_mfma(o, d, c, r)
oc = o[:, None, :] - c[None, :, :]
hb = (d[:, None, :] * oc).sum(2)
disc = hb * hb - ((oc * oc).sum(2) - (r * r)[None, :])
fp = keep & (disc < 0)
print("kept", int(keep.sum()), "true", int((disc >= 0).sum()), "dropped", int(((disc >= 0) & ~keep).sum()))
print("false positives: -disc percentiles 50/90/99/max", np.percentile(-disc[fp], [50, 90, 99, 100]))
# the smallest -disc that is NOT kept
nk = (~keep) & (disc < 0)
print("not kept: smallest -disc percentiles 0/1/10", np.percentile(-disc[nk], [0, 1, 10]))
