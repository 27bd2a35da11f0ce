"""ctypes wrapper of the CPU oracle (oracle/liboracle.so) and of the compiled
reference harness (oracle/_ref/ref_harness).

TEST INFRASTRUCTURE ONLY.  Imported by tests/, __graft_entry__.smoke() and
bench.py's cpu_baseline leg; the product package never imports this module.
"""
from __future__ import annotations

import ctypes as C
import os
import struct
import subprocess
import tempfile

import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))
LIB_PATH = os.path.join(HERE, "liboracle.so")
REF_HARNESS = os.path.join(HERE, "_ref", "ref_harness")
REF_KAT = os.path.join(HERE, "_ref", "ref_kat")

POLICY_REF, POLICY_DEV = 0, 1
RNG_STREAM, RNG_PIXEL = 0, 1

_lib = None


def build():
    subprocess.check_call(["make", "-C", HERE, "oracle"], stdout=subprocess.DEVNULL)


def lib():
    global _lib
    if _lib is None:
        if not os.path.exists(LIB_PATH):
            build()
        l = C.CDLL(LIB_PATH)
        dp = C.POINTER(C.c_double)
        l.oracle_render.restype = C.c_int
        l.oracle_render.argtypes = [dp, C.c_int, dp, C.c_int, dp, C.c_int, dp, C.c_int, C.c_int, C.c_int,
                                    C.c_int, C.c_int, dp, C.POINTER(C.c_int64), C.POINTER(C.c_int32), C.POINTER(C.c_int32)]
        l.oracle_expose.restype = None
        l.oracle_expose.argtypes = [C.c_int, C.c_int, C.c_double, C.c_int, C.c_int, dp, C.POINTER(C.c_int32)]
        l.oracle_pixel_seed.restype = C.c_uint32
        l.oracle_pixel_seed.argtypes = [C.c_uint32, C.c_uint32]
        l.oracle_rng_stream.restype = None
        l.oracle_rng_stream.argtypes = [C.c_uint32, C.c_int, C.POINTER(C.c_uint32)]
        l.oracle_math.restype = None
        l.oracle_math.argtypes = [C.c_int, C.c_int, dp, dp, dp]
        l.oracle_kat.restype = None
        l.oracle_kat.argtypes = [dp, C.c_int, dp, C.c_int, dp, C.c_int, dp, C.c_int, dp, dp]
        _lib = l
    return _lib


def _dp(a):
    return a.ctypes.data_as(C.POINTER(C.c_double))


def render(scene, policy=POLICY_DEV, rng_mode=RNG_PIXEL, workers=None, rows=None, want_hits=False, want_pix_rays=False):
    """Render ``scene`` (raytracer_amd.scene.Scene) on the CPU oracle.

    Returns dict(rgb=float64[H,W,3], rays=int64[...], hits=int32[H,W,spp]|None, pix_rays=int32[H,W]|None).
    """
    l = lib()
    W, H, spp = scene.width, scene.height, scene.spp
    params = scene.params()
    prims = np.ascontiguousarray(scene.prims, dtype=np.float64)
    mats = np.ascontiguousarray(scene.mats, dtype=np.float64)
    verts = np.ascontiguousarray(scene.verts, dtype=np.float64)
    if verts.size == 0:
        verts = np.zeros((1, 3))
    if workers is None:
        workers = min(os.cpu_count() or 1, 16)
    r0, r1 = rows if rows is not None else (0, H)
    rgb = np.zeros((H, W, 3), dtype=np.float64)
    nrays = int(scene.header["num_threads"]) if rng_mode == RNG_STREAM else 1
    rays = np.zeros(max(nrays, 1), dtype=np.int64)
    hits = np.full((H, W, spp), -2, dtype=np.int32) if want_hits else None
    pix = np.zeros((H, W), dtype=np.int32) if want_pix_rays else None
    rc = l.oracle_render(_dp(params), len(prims), _dp(prims), len(mats), _dp(mats), len(scene.verts), _dp(verts),
                         policy, rng_mode, workers, r0, r1, _dp(rgb), rays.ctypes.data_as(C.POINTER(C.c_int64)),
                         hits.ctypes.data_as(C.POINTER(C.c_int32)) if want_hits else None,
                         pix.ctypes.data_as(C.POINTER(C.c_int32)) if want_pix_rays else None)
    if rc != 0:
        raise RuntimeError(f"oracle_render failed rc={rc}")
    return dict(rgb=rgb, rays=rays, hits=hits, pix_rays=pix)


def trace_pixel(scene, x, y):
    """Debugging aid: renders one pixel (device policy, per-pixel seed) and prints every segment to stderr."""
    l = lib()
    params = scene.params()
    prims = np.ascontiguousarray(scene.prims, dtype=np.float64)
    mats = np.ascontiguousarray(scene.mats, dtype=np.float64)
    verts = np.ascontiguousarray(scene.verts, dtype=np.float64)
    if verts.size == 0:
        verts = np.zeros((1, 3))
    rgb = np.zeros(3)
    l.oracle_trace_pixel.restype = C.c_int
    rc = l.oracle_trace_pixel(_dp(params), len(prims), _dp(prims), len(mats), _dp(mats), len(scene.verts), _dp(verts),
                              C.c_int(x), C.c_int(y), _dp(rgb))
    return rc, rgb


def expose(rgb, exposure, num_blurs, workers=None):
    """Bloom + 8-bit pack; returns (int32[H,W] packed 0xRRGGBB, float64 rgb with bloom added)."""
    l = lib()
    H, W, _ = rgb.shape
    buf = np.ascontiguousarray(rgb, dtype=np.float64).copy()
    img = np.zeros((H, W), dtype=np.int32)
    if workers is None:
        workers = min(os.cpu_count() or 1, 16)
    l.oracle_expose(W, H, float(exposure), int(num_blurs), workers, _dp(buf), img.ctypes.data_as(C.POINTER(C.c_int32)))
    return img, buf


def pixel_seed(idx, frame_seed=0):
    return int(lib().oracle_pixel_seed(int(idx) & 0xFFFFFFFF, int(frame_seed) & 0xFFFFFFFF))


def rng_stream(state, n):
    out = np.zeros(n, dtype=np.uint32)
    lib().oracle_rng_stream(int(state), n, out.ctypes.data_as(C.POINTER(C.c_uint32)))
    return out


def math_fn(fn, x, y=None):
    x = np.ascontiguousarray(x, dtype=np.float64)
    y = np.ascontiguousarray(y if y is not None else np.zeros_like(x), dtype=np.float64)
    out = np.zeros_like(x)
    lib().oracle_math({"sin": 0, "cos": 1, "acos": 2, "pow": 3}[fn], x.size, _dp(x), _dp(y), _dp(out))
    return out


def unpack_rgb(img):
    """int32 0xRRGGBB -> uint8[H,W,3]."""
    return np.stack([(img >> 16) & 255, (img >> 8) & 255, img & 255], axis=-1).astype(np.uint8)


# ----------------------------------------------------------------- reference
def have_reference():
    return os.path.exists(REF_HARNESS)


def run_reference(scene_text, num_threads=-1, num_blurs=-1, timeout=600):
    """Run the compiled, unmodified reference on a scene stream (oracle/ref_harness.cpp)."""
    with tempfile.TemporaryDirectory() as td:
        sp, op = os.path.join(td, "scene.json"), os.path.join(td, "out.bin")
        with open(sp, "w") as f:
            f.write(scene_text)
        subprocess.run([REF_HARNESS, sp, op, str(num_threads), str(num_blurs)], check=True, timeout=timeout,
                       stdout=subprocess.DEVNULL, stderr=subprocess.DEVNULL)
        b = open(op, "rb").read()
    W, H, T, NB = struct.unpack("<4i", b[:16])
    off = 16
    rays = np.frombuffer(b, dtype="<i8", count=T, offset=off).copy(); off += 8 * T
    rgb = np.frombuffer(b, dtype="<f8", count=W * H * 3, offset=off).reshape(H, W, 3).copy(); off += 8 * W * H * 3
    img = np.frombuffer(b, dtype="<i4", count=W * H, offset=off).reshape(H, W).copy(); off += 4 * W * H
    secs = struct.unpack("<d", b[off:off + 8])[0]
    return dict(rgb=rgb, rays=rays, img=img, seconds=secs, threads=T, blurs=NB)


# ------------------------------------------------------------------ function-level known answers
def kat(scene, rays_in):
    """oracle_kat: rays_in float64[n,7] (o, d, RNG state) -> records float64[n, 35 + 3*n_prims] in ref_kat's layout."""
    l = lib()
    rays_in = np.ascontiguousarray(rays_in, dtype=np.float64)
    params = scene.params()
    prims = np.ascontiguousarray(scene.prims, dtype=np.float64)
    mats = np.ascontiguousarray(scene.mats, dtype=np.float64)
    verts = np.ascontiguousarray(scene.verts, dtype=np.float64)
    if verts.size == 0:
        verts = np.zeros((1, 3))
    out = np.zeros((len(rays_in), 35 + 3 * len(prims)), dtype=np.float64)
    l.oracle_kat(_dp(params), len(prims), _dp(prims), len(mats), _dp(mats), len(scene.verts), _dp(verts),
                 len(rays_in), _dp(rays_in), _dp(out))
    return out


def run_reference_kat(scene_path, n_rays, seed):
    """Runs oracle/_ref/ref_kat (the reference's own functions).  Returns (records float64[n, 35+3*n_objs],
    colors float64[512, 5] = r g b max packed)."""
    with tempfile.TemporaryDirectory() as td:
        out = os.path.join(td, "kat.bin")
        subprocess.run([REF_KAT, scene_path, str(int(n_rays)), str(int(seed)), out], check=True,
                       stdout=subprocess.DEVNULL, stderr=subprocess.DEVNULL, cwd=td)
        raw = open(out, "rb").read()
    n, nobj = struct.unpack_from("<ii", raw, 0)
    stride = 35 + 3 * nobj
    rec = np.frombuffer(raw, dtype="<f8", count=n * stride, offset=8).reshape(n, stride).copy()
    off = 8 + 8 * n * stride
    (nc,) = struct.unpack_from("<i", raw, off)
    col = np.frombuffer(raw, dtype="<f8", count=nc * 5, offset=off + 4).reshape(nc, 5).copy()
    return rec, col
