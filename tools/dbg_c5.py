"""Render a full-size synthetic configuration and report the guard pixel, if any: tools/dbg_c5.py [config]"""
import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from raytracer_amd import api, synth
cfg = sys.argv[1] if len(sys.argv) > 1 else "c5"
_, sc = synth.make(cfg)
r = api.Renderer(0)
r.upload(sc)
try:
    g = r.render()
    print("ok rays", g["rays"], r.timing())
except api.RtError as e:
    print("ERR", e)
