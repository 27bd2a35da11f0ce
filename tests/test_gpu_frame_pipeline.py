"""FramePipeline (raytracer_amd/dist.py) on the device with the library's own calls, the way bench.py drives it:
render -> expose -> gather of one rank, with and without blur passes, plain and pipelined (a frame's gather overlapping
the next frame's render, two send buffers).  The frame that comes out must be the oracle's: render with the device
arithmetic policy and per-pixel seeds, then expose()  (main.cpp:305-315 is the sequence).  Integer output: every packed
pixel equal."""
import numpy as np
import pytest

import pyoracle as po
from raytracer_amd import synth
from raytracer_amd.dist import FramePipeline

torch = pytest.importorskip("torch")
pytestmark = pytest.mark.gpu


@pytest.mark.parametrize("blurs,pipelined", [(0, False), (0, True), (2, False), (2, True)])
def test_one_rank_frame_pipeline_on_the_device(renderer, blurs, pipelined):
    _, sc = synth.make("c2", width=160)
    renderer.upload(sc)
    H, W = sc.height, sc.width
    exposure = float(sc.header["exposure"]) * 3.0            # part of the frame saturates: the bloom has input
    dev = torch.device("cuda", 0)
    st = torch.cuda.current_stream(dev)

    def render_rows(dst):
        renderer.render_rows_device(dst.data_ptr(), 8, 0, 1, stream=st.cuda_stream)

    def expose(src, rows, dst, nb):
        renderer.expose_device(src.data_ptr(), W, rows, exposure, nb, dst.data_ptr(), stream=st.cuda_stream)

    pipe = FramePipeline(H, W, blurs, 8, 0, 1, dev, render_rows, expose, pipelined=pipelined)
    assert pipe.pipelined == (pipelined and blurs == 0)     # with blur passes one rank exposes the whole frame: nothing to overlap
    frames = []
    for _ in range(3):                                       # buffers are reused, and alternate when pipelined
        f = pipe.step()
        renderer.finish()
        if f is not None:
            frames.append(f.cpu().numpy().reshape(H, W).copy())
    last = pipe.flush()
    if last is not None:
        frames.append(last.cpu().numpy().reshape(H, W).copy())
    assert len(frames) == 3
    o = po.render(sc, policy=po.POLICY_DEV, rng_mode=po.RNG_PIXEL)
    want, _ = po.expose(o["rgb"], exposure, blurs)
    assert len(np.unique(want)) > 10
    for f in frames:
        assert np.array_equal(f, want), f"{int((f != want).sum())} packed pixels differ"
