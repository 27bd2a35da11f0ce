"""GPU parity on the small scenes of tests/edge_scenes.py (edges of the scene format and of the path logic).  Same bar
as the other parity tests: radiance, hit ids and ray counts bit-exact against the oracle, both renderers."""
import numpy as np
import pytest

import pyoracle as po
from helpers import bits_equal
from raytracer_amd import api, scene as S

pytestmark = pytest.mark.gpu

from edge_scenes import SCENES, text_of


@pytest.mark.parametrize("name,header,objs", SCENES, ids=[s[0] for s in SCENES])
def test_edge_scene_matches_oracle(renderer, name, header, objs):
    sc = S.loads(text_of(header, objs))
    o = po.render(sc, policy=po.POLICY_DEV, rng_mode=po.RNG_PIXEL, want_hits=True)
    for mode in (api.MODE_PIPELINE, api.MODE_SINGLE_KERNEL):
        try:
            renderer.set_option(api.OPT_MODE, mode)
            renderer.upload(sc)
            g = renderer.render(want_hits=True)
        finally:
            renderer.set_option(api.OPT_MODE, api.MODE_PIPELINE)
        assert g["rays"] == int(o["rays"].sum()), f"{name}/{mode}: rays {g['rays']} vs {int(o['rays'].sum())}"
        assert (g["hits"] == o["hits"]).all(), f"{name}/{mode}: hit ids differ"
        eq = bits_equal(g["rgb"], o["rgb"])
        assert eq.all(), f"{name}/{mode}: {int((~eq).sum())} radiance words differ"
    if name == "empty":
        assert g["rays"] == sc.width * sc.height * sc.spp and (g["hits"] == -1).all() and not g["rgb"].any()
    if name == "plane_flagged_emitter":
        assert sc.prims[0, S.PR_LIGHT] == 0.0            # readShape.cpp:75 never sets is_light for a plane


def test_reference_order_on_an_edge_scene(renderer):
    """The reference's RNG order (one stream per reference thread) on a frame with fewer rows than threads."""
    name, header, objs = SCENES[2]
    header = dict(header, width=16, height=3, num_threads=4)
    sc = S.loads(text_of(header, objs))
    renderer.upload(sc)
    g = renderer.render(want_hits=True, streams=True)
    o = po.render(sc, policy=po.POLICY_DEV, rng_mode=po.RNG_STREAM, want_hits=True)
    assert np.array_equal(g["stream_rays"].astype(np.int64), o["rays"])
    assert (g["hits"] == o["hits"]).all() and bits_equal(g["rgb"], o["rgb"]).all()
