"""The library's own RCCL calls (rt_multi_*: dlopen of librccl, ncclCommInitAll, ncclAllGather of the over-exposure bits,
grouped ncclSend / ncclRecv of the packed rows) executed on the one GPU of the box: RT_MULTI_FORCE_RCCL=1 makes a context
of ONE rank build a communicator and send its rows and bits to itself through those calls instead of device copies.
Bar: the frame equals rt_render_image's on a plain context, every packed pixel and the ray count (integer output).
Also: two overlapped frames (rt_multi_render_begin / _end) on 8 ranks that share the GPU equal the single-context frames."""
import os

import numpy as np
import pytest

from raytracer_amd import api, synth

pytestmark = pytest.mark.gpu


def _with_env(name, value):
    old = os.environ.get(name)
    os.environ[name] = value
    return old


def _restore(name, old):
    if old is None:
        os.environ.pop(name, None)
    else:
        os.environ[name] = old


@pytest.mark.parametrize("blurs,width", [(0, 160), (2, 128)])
def test_one_rank_through_rccl_equals_render_image(renderer, blurs, width):
    _, sc = synth.make("c3", width=width, spp=1, num_blurs=blurs)
    renderer.upload(sc)
    want = renderer.render_image()
    old_lb = os.environ.pop("RT_MULTI_LOOPBACK", None)
    old = _with_env("RT_MULTI_FORCE_RCCL", "1")
    try:
        m = api.MultiRenderer(1)
        try:
            assert m.uses_rccl()
            m.upload(sc)
            got = m.render_image()
            again = m.render_image(3)            # another partition: buffers re-made, communicator kept
        finally:
            m.close()
    finally:
        _restore("RT_MULTI_FORCE_RCCL", old)
        _restore("RT_MULTI_LOOPBACK", old_lb)
    assert any("librccl" in line for line in open("/proc/self/maps")), "RCCL is not mapped into this process"
    assert got["rays"] == want["rays"] and np.array_equal(got["img"], want["img"])
    assert again["rays"] == want["rays"] and np.array_equal(again["img"], want["img"])


def test_plain_one_rank_context_does_not_touch_rccl():
    old = os.environ.pop("RT_MULTI_FORCE_RCCL", None)
    try:
        m = api.MultiRenderer(1)
        try:
            assert not m.uses_rccl()
        finally:
            m.close()
    finally:
        _restore("RT_MULTI_FORCE_RCCL", old)


@pytest.mark.parametrize("ranks,blurs,width", [(8, 0, 96), (4, 2, 128)])
def test_two_frames_in_flight_equal_single_context_frames(renderer, ranks, blurs, width):
    _, sc = synth.make("c3", width=width, spp=1, num_blurs=blurs)
    renderer.upload(sc)
    want = renderer.render_image()
    old = _with_env("RT_MULTI_LOOPBACK", "1")
    try:
        m = api.MultiRenderer(ranks)
        try:
            m.upload(sc)
            m.begin()
            m.begin()
            with pytest.raises(api.RtError):     # a third frame has no buffers
                m.begin()
            a = m.end()
            m.begin()                            # frame 3 goes into frame 1's buffers while frame 2 is in flight
            b = m.end()
            c = m.end()
            with pytest.raises(api.RtError):
                m.end()
            d = m.render_image()
        finally:
            m.close()
    finally:
        _restore("RT_MULTI_LOOPBACK", old)
    for f in (a, b, c, d):
        assert f["rays"] == want["rays"] and np.array_equal(f["img"], want["img"])
    assert a["enqueue_ms"] > 0 and a["wait_ms"] >= 0 and a["ms"] >= a["enqueue_ms"]
