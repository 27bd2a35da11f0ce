"""Error paths of the C ABI on the device: a refused upload leaves the handle without a scene and leaks nothing that a
later upload would trip over; a valid upload afterwards renders as if nothing had happened."""
import ctypes as C

import numpy as np
import pytest

import pyoracle as po
from helpers import bits_equal
from raytracer_amd import api, scene as S, synth

pytestmark = pytest.mark.gpu


def test_refused_uploads_leave_no_scene_and_do_not_disturb_the_next_one():
    ren = api.Renderer(0)
    try:
        _, good = synth.make("c3", width=96, spp=1)
        ren.upload(good)
        want = ren.render()
        params = good.params().copy()
        for k in range(300):                       # each of these used to leave two timing events behind
            p = params.copy()
            if k % 3 == 0:
                p[3] = 17.0                        # max_ray_depth above RT_MAX_DEPTH
                code = -4
            elif k % 3 == 1:
                p[0] = 0.0                         # width 0
                code = -1
            else:
                p[0], p[1] = 65536.0, 65536.0      # 2^32 pixels
                code = -4
            prims = np.ascontiguousarray(good.prims); mats = np.ascontiguousarray(good.mats); verts = np.ascontiguousarray(good.verts)
            dp = lambda a: a.ctypes.data_as(C.POINTER(C.c_double))
            rc = ren.lib.rt_upload_scene(ren.h, dp(p), len(prims), dp(prims), len(mats), dp(mats), len(verts), dp(verts))
            assert rc == code, (k, rc, ren.lib.rt_last_error(ren.h))
        # a bad material index is found after validation of the header, while the tables are being flattened
        prims = np.ascontiguousarray(good.prims).copy()
        prims[5, S.PR_MAT0] = 1e6
        dp = lambda a: a.ctypes.data_as(C.POINTER(C.c_double))
        mats = np.ascontiguousarray(good.mats); verts = np.ascontiguousarray(good.verts)
        rc = ren.lib.rt_upload_scene(ren.h, dp(params), len(prims), dp(prims), len(mats), dp(mats), len(verts), dp(verts))
        assert rc == -1
        # the earlier scene is still the one uploaded (a refused upload touches nothing)
        again = ren.render()
        assert again["rays"] == want["rays"] and bits_equal(again["rgb"], want["rgb"]).all()
        ren.upload(good)
        again = ren.render()
        assert again["rays"] == want["rays"] and bits_equal(again["rgb"], want["rgb"]).all()
    finally:
        ren.close()


def test_render_before_upload_and_bad_arguments_are_refused():
    ren = api.Renderer(0)
    try:
        with pytest.raises(api.RtError) as e:
            ren.lib.rt_render.restype = C.c_int
            rgb = np.zeros(3)
            rays = C.c_uint64(0)
            ren._check(ren.lib.rt_render(ren.h, rgb.ctypes.data_as(C.POINTER(C.c_double)), C.byref(rays), None))
        assert e.value.code == -3
        with pytest.raises(api.RtError):
            ren.set_option(api.OPT_OVERLAP, 9)
        with pytest.raises(api.RtError):
            ren.set_option(99, 1)
    finally:
        ren.close()
