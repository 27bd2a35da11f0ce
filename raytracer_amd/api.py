"""ctypes binding of librt_hip.so (include/rt_hip.h) and the host-side render call.

Mirrors the host sequence of the reference's main()
(/root/reference/src/main.cpp:291-333): load scene -> render every pixel ->
expose -> write image, with the thread fan-out around do_rays_i replaced by one
call into the HIP library.  There is no CPU fallback: without the library or a
GPU every render call raises.
"""
from __future__ import annotations

import ctypes as C
import os

import numpy as np

from . import scene as S

HERE = os.path.dirname(os.path.abspath(__file__))
LIB_PATH = os.environ.get("RT_HIP_LIB", os.path.join(HERE, "librt_hip.so"))   # RT_HIP_LIB: A/B builds of the library

RT_OK = 0
# rt_set_option keys / values (include/rt_hip.h)
OPT_MODE, OPT_TIMING_DETAIL, OPT_BLOCK, OPT_BROAD_BLOCK, OPT_WG_PER_CU, OPT_CHUNK_RECORDS, OPT_MAX_SLOTS, OPT_FUSE_NARROW, OPT_OVERLAP, OPT_ACCEL = range(10)
ACCEL_SCAN, ACCEL_WALK, ACCEL_AUTO = 0, 1, 2
MODE_PIPELINE, MODE_SINGLE_KERNEL = 0, 1
NOALLOC_MAX_PIXELS = 262144         # RT_NOALLOC_MAX_PIXELS
STATUS = {0: "RT_OK", -1: "RT_ERR_ARG", -2: "RT_ERR_HIP", -3: "RT_ERR_NO_SCENE", -4: "RT_ERR_SCENE",
          -5: "RT_ERR_GUARD", -6: "RT_ERR_NO_DEVICE"}

EXPORTS = ("rt_create", "rt_destroy", "rt_set_option", "rt_upload_scene", "rt_rows_for_rank", "rt_render_rows_device",
           "rt_render_streams_device", "rt_finish", "rt_stream_rays", "rt_render", "rt_render_streams", "rt_expose_device", "rt_render_image",
           "rt_expose_bits_device", "rt_expose_rows_device", "rt_expose_device_noalloc",
           "rt_last_timing", "rt_last_error", "rt_selftest_math", "rt_selftest_broad", "rt_selftest_broad_mfma", "rt_face_bound", "rt_mfma_record_pieces", "rt_mfma_table_order", "rt_pixel_map_divisor",
           "rt_multi_create", "rt_multi_destroy", "rt_multi_gpus", "rt_multi_set_option", "rt_multi_upload_scene",
           "rt_multi_render_image", "rt_multi_render_begin", "rt_multi_render_end", "rt_multi_last_ms", "rt_multi_last_times",
           "rt_multi_uses_rccl", "rt_multi_last_error", "rt_gather_source_row")


class RtError(RuntimeError):
    def __init__(self, code, text):
        super().__init__(f"{STATUS.get(code, code)}: {text}")
        self.code = code


class RtTiming(C.Structure):
    _fields_ = [("upload_ms", C.c_double), ("kernel_ms", C.c_double), ("broad_ms", C.c_double), ("total_ms", C.c_double),
                ("rays", C.c_uint64), ("candidates", C.c_uint64), ("sphere_tests", C.c_uint64), ("plane_tests", C.c_uint64),
                ("face_tests", C.c_uint64), ("face_vert_tests", C.c_uint64), ("overflow_rays", C.c_uint64), ("broad_tiles", C.c_uint64), ("broad_records", C.c_uint64), ("accel_used", C.c_uint64), ("shade_ms", C.c_double),
                ("sub_pipelines", C.c_uint64)]


_lib = None


def load_library(path: str = LIB_PATH):
    """Load librt_hip.so and declare its prototypes.  Raises if it is missing: the HIP
    library is the only implementation of the render path."""
    global _lib
    if _lib is not None:
        return _lib
    if not os.path.exists(path):
        raise RtError(-6, f"{path} not found: build it with `make -C raytracer_amd/csrc` (python -c 'import __graft_entry__ as g; g.build()')")
    l = C.CDLL(path)
    dp = C.POINTER(C.c_double)
    l.rt_create.restype = C.c_int
    l.rt_create.argtypes = [C.POINTER(C.c_void_p), C.c_int]
    l.rt_destroy.restype = None
    l.rt_destroy.argtypes = [C.c_void_p]
    l.rt_set_option.restype = C.c_int
    l.rt_set_option.argtypes = [C.c_void_p, C.c_int, C.c_int]
    l.rt_upload_scene.restype = C.c_int
    l.rt_upload_scene.argtypes = [C.c_void_p, dp, C.c_int, dp, C.c_int, dp, C.c_int, dp]
    l.rt_rows_for_rank.restype = C.c_int
    l.rt_rows_for_rank.argtypes = [C.c_int, C.c_int, C.c_int, C.c_int]
    l.rt_render_rows_device.restype = C.c_int
    l.rt_render_rows_device.argtypes = [C.c_void_p, C.c_int, C.c_int, C.c_int, C.c_void_p, C.c_void_p, C.c_void_p]
    l.rt_render_streams_device.restype = C.c_int
    l.rt_render_streams_device.argtypes = [C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p]
    l.rt_finish.restype = C.c_int
    l.rt_finish.argtypes = [C.c_void_p, C.POINTER(C.c_uint64)]
    l.rt_stream_rays.restype = C.c_int
    l.rt_stream_rays.argtypes = [C.c_void_p, C.POINTER(C.c_uint64), C.c_int]
    for f in (l.rt_render, l.rt_render_streams):
        f.restype = C.c_int
        f.argtypes = [C.c_void_p, dp, C.POINTER(C.c_uint64), C.POINTER(C.c_int32)]
    l.rt_expose_device.restype = C.c_int
    l.rt_expose_device.argtypes = [C.c_void_p, C.c_void_p, C.c_int, C.c_int, C.c_double, C.c_int, C.c_void_p, C.c_void_p]
    l.rt_expose_device_noalloc.restype = C.c_int
    l.rt_expose_device_noalloc.argtypes = l.rt_expose_device.argtypes
    l.rt_expose_bits_device.restype = C.c_int
    l.rt_expose_bits_device.argtypes = [C.c_void_p, C.c_void_p, C.c_int, C.c_int, C.c_double, C.c_void_p, C.c_void_p]
    l.rt_expose_rows_device.restype = C.c_int
    l.rt_expose_rows_device.argtypes = [C.c_void_p, C.c_void_p, C.c_void_p, C.c_int, C.c_int, C.c_int, C.c_int, C.c_int, C.c_int,
                                        C.c_double, C.c_int, C.c_void_p, C.c_void_p]
    l.rt_render_image.restype = C.c_int
    l.rt_render_image.argtypes = [C.c_void_p, C.c_int, C.POINTER(C.c_int32), C.POINTER(C.c_uint64)]
    l.rt_last_timing.restype = C.c_int
    l.rt_last_timing.argtypes = [C.c_void_p, C.POINTER(RtTiming)]
    l.rt_last_error.restype = C.c_char_p
    l.rt_last_error.argtypes = [C.c_void_p]
    l.rt_selftest_math.restype = C.c_int
    l.rt_selftest_math.argtypes = [C.c_void_p, C.c_int, C.c_int, dp, dp, dp]
    l.rt_selftest_broad.restype = C.c_int
    l.rt_selftest_broad.argtypes = [C.c_void_p, C.c_int, dp, dp, dp, dp, C.POINTER(C.c_int32)]
    l.rt_selftest_broad_mfma.restype = C.c_int
    l.rt_selftest_broad_mfma.argtypes = [C.c_void_p, C.c_int, dp, dp, C.c_int, dp, dp, C.POINTER(C.c_uint32)]
    l.rt_multi_create.restype = C.c_int
    l.rt_multi_create.argtypes = [C.POINTER(C.c_void_p), C.c_int]
    l.rt_multi_destroy.restype = None
    l.rt_multi_destroy.argtypes = [C.c_void_p]
    l.rt_multi_gpus.restype = C.c_int
    l.rt_multi_gpus.argtypes = [C.c_void_p]
    l.rt_multi_set_option.restype = C.c_int
    l.rt_multi_set_option.argtypes = [C.c_void_p, C.c_int, C.c_int]
    l.rt_multi_upload_scene.restype = C.c_int
    l.rt_multi_upload_scene.argtypes = [C.c_void_p, dp, C.c_int, dp, C.c_int, dp, C.c_int, dp]
    l.rt_multi_render_image.restype = C.c_int
    l.rt_multi_render_image.argtypes = [C.c_void_p, C.c_int, C.POINTER(C.c_int32), C.POINTER(C.c_uint64)]
    l.rt_multi_render_begin.restype = C.c_int
    l.rt_multi_render_begin.argtypes = [C.c_void_p, C.c_int]
    l.rt_multi_render_end.restype = C.c_int
    l.rt_multi_render_end.argtypes = [C.c_void_p, C.POINTER(C.c_int32), C.POINTER(C.c_uint64)]
    l.rt_multi_last_times.restype = C.c_int
    l.rt_multi_last_times.argtypes = [C.c_void_p, C.POINTER(C.c_double), C.POINTER(C.c_double)]
    l.rt_multi_uses_rccl.restype = C.c_int
    l.rt_multi_uses_rccl.argtypes = [C.c_void_p]
    l.rt_multi_last_ms.restype = C.c_double
    l.rt_multi_last_ms.argtypes = [C.c_void_p]
    l.rt_multi_last_error.restype = C.c_char_p
    l.rt_multi_last_error.argtypes = [C.c_void_p]
    l.rt_gather_source_row.restype = C.c_int
    l.rt_gather_source_row.argtypes = [C.c_int] * 5
    l.rt_mfma_table_order.restype = C.c_int
    l.rt_mfma_table_order.argtypes = [C.c_int, dp, dp, C.c_int, C.POINTER(C.c_uint32), C.c_int, dp, C.c_int]
    l.rt_mfma_record_pieces.restype = C.c_int
    l.rt_mfma_record_pieces.argtypes = [dp, C.c_double, C.POINTER(C.c_double), dp]
    l.rt_pixel_map_divisor.restype = C.c_int
    l.rt_pixel_map_divisor.argtypes = [C.c_uint32, C.c_uint64, C.POINTER(C.c_uint32), C.POINTER(C.c_uint32)]
    l.rt_face_bound.restype = C.c_int
    l.rt_face_bound.argtypes = [C.c_int, dp, dp, dp, dp]
    _lib = l
    return l


def pixel_map_divisor(d, n_max):
    """(mul, shift) with n // d == ((n * mul) >> 32) >> shift for 0 <= n <= n_max, or None when the kernels divide.  Host only."""
    mul, sh = C.c_uint32(0), C.c_uint32(0)
    rc = load_library().rt_pixel_map_divisor(int(d), int(n_max), C.byref(mul), C.byref(sh))
    if rc < 0:
        raise RtError(rc, "rt_pixel_map_divisor: bad argument")
    return (int(mul.value), int(sh.value)) if rc == 1 else None


def face_bound(verts, normal):
    """Bounding sphere of the region Face::intersect accepts for this face: (centre[3], radius), or None when the region
    is unbounded (the face is then tested for every ray).  Host arithmetic only."""
    v = np.ascontiguousarray(verts, dtype=np.float64).reshape(-1, 3)
    n = np.ascontiguousarray(normal, dtype=np.float64)
    c = np.zeros(3)
    r = C.c_double(0.0)
    rc = load_library().rt_face_bound(len(v), _dp(v), _dp(n), _dp(c), C.byref(r))
    if rc < 0:
        raise RtError(rc, "rt_face_bound: bad argument")
    return (c, float(r.value)) if rc == 1 else None


def mfma_table_order(centres, radii, reorder=True):
    """(perm[32 * groups], group_spheres[groups][4]) of the matrix-core scan's table for these bounding spheres.  Host only."""
    c = np.ascontiguousarray(centres, dtype=np.float64).reshape(-1, 3)
    r = np.ascontiguousarray(radii, dtype=np.float64)
    cap = (len(c) // 32 + 2)
    perm = np.zeros(cap * 32, dtype=np.uint32)
    gs = np.zeros((cap, 4))
    n = load_library().rt_mfma_table_order(len(c), _dp(c), _dp(r), 1 if reorder else 0,
                                           perm.ctypes.data_as(C.POINTER(C.c_uint32)), len(perm), _dp(gs), cap)
    if n < 0:
        raise RtError(n, "rt_mfma_table_order: bad argument")
    return perm[: n * 32].copy(), gs[:n].copy()


def mfma_record_pieces(record, cmax):
    """(scale, pieces[4][2]) of one broad-phase record (cx, cy, cz, w) in the matrix-core scan's f16 table.  Host only."""
    rec = np.ascontiguousarray(record, dtype=np.float64).reshape(4)
    sc = C.c_double(0.0)
    out = np.zeros(8)
    rc = load_library().rt_mfma_record_pieces(_dp(rec), float(cmax), C.byref(sc), _dp(out))
    if rc < 0:
        raise RtError(rc, "rt_mfma_record_pieces: bad argument")
    return float(sc.value), out.reshape(4, 2)


def rows_for_rank(height, tile_rows, rank, world):
    return load_library().rt_rows_for_rank(height, tile_rows, rank, world)


def _dp(a):
    return a.ctypes.data_as(C.POINTER(C.c_double))


class Renderer:
    """One rt_handle bound to one GPU."""

    def __init__(self, device: int = 0):
        self.lib = load_library()
        self.h = C.c_void_p()
        rc = self.lib.rt_create(C.byref(self.h), int(device))
        if rc != RT_OK:
            raise RtError(rc, "rt_create failed (no usable gfx950 device?)")
        self.scene = None

    def _check(self, rc):
        if rc != RT_OK:
            raise RtError(rc, self.lib.rt_last_error(self.h).decode())

    def close(self):
        if self.h:
            self.lib.rt_destroy(self.h)
            self.h = C.c_void_p()

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass

    def set_option(self, key: int, value: int):
        self._check(self.lib.rt_set_option(self.h, int(key), int(value)))

    def upload(self, sc: S.Scene):
        params = sc.params()
        prims = np.ascontiguousarray(sc.prims, dtype=np.float64)
        mats = np.ascontiguousarray(sc.mats, dtype=np.float64)
        verts = np.ascontiguousarray(sc.verts, dtype=np.float64)
        self._check(self.lib.rt_upload_scene(self.h, _dp(params), len(prims), _dp(prims) if len(prims) else None,
                                             len(mats), _dp(mats) if len(mats) else None,
                                             len(verts), _dp(verts) if len(verts) else None))
        self.scene = sc

    def render(self, want_hits=False, streams=False):
        """Whole frame, outputs on the host: dict(rgb=float64[H,W,3], rays=int, hits=int32[H,W,spp]|None).
        streams=True renders in the reference's RNG order (one lane per reference thread; validation mode)
        and adds stream_rays=uint64[num_threads]."""
        sc = self.scene
        if sc is None:
            raise RtError(-3, "no scene uploaded")
        rgb = np.zeros((sc.height, sc.width, 3), dtype=np.float64)
        hits = np.full((sc.height, sc.width, sc.spp), -2, dtype=np.int32) if want_hits else None
        rays = C.c_uint64(0)
        fn = self.lib.rt_render_streams if streams else self.lib.rt_render
        self._check(fn(self.h, _dp(rgb), C.byref(rays), hits.ctypes.data_as(C.POINTER(C.c_int32)) if want_hits else None))
        out = dict(rgb=rgb, rays=int(rays.value), hits=hits)
        if streams:
            T = int(sc.header["num_threads"])
            sr = np.zeros(T, dtype=np.uint64)
            self._check(self.lib.rt_stream_rays(self.h, sr.ctypes.data_as(C.POINTER(C.c_uint64)), T))
            out["stream_rays"] = sr
        return out

    def render_image(self, streams=False):
        """Render + expose on the device, packed frame on the host: dict(img=int32[H,W], rays=int)."""
        sc = self.scene
        if sc is None:
            raise RtError(-3, "no scene uploaded")
        img = np.zeros((sc.height, sc.width), dtype=np.int32)
        rays = C.c_uint64(0)
        self._check(self.lib.rt_render_image(self.h, 1 if streams else 0, img.ctypes.data_as(C.POINTER(C.c_int32)), C.byref(rays)))
        return dict(img=img, rays=int(rays.value))

    def render_rows_device(self, d_rgb_ptr, tile_rows, rank, world, d_hits_ptr=None, stream=None):
        """Asynchronous render of this rank's row tiles into device memory (raw pointers)."""
        self._check(self.lib.rt_render_rows_device(self.h, int(tile_rows), int(rank), int(world), C.c_void_p(d_rgb_ptr),
                                                   C.c_void_p(d_hits_ptr) if d_hits_ptr else None,
                                                   C.c_void_p(stream) if stream else None))

    def expose_device(self, d_rgb_ptr, width, height, exposure, num_blurs, d_img_ptr, stream=None):
        """expose() on the device (bloom + 8-bit pack): double[height*width*3] -> int32[height*width], asynchronous."""
        self._check(self.lib.rt_expose_device(self.h, C.c_void_p(d_rgb_ptr), int(width), int(height), float(exposure),
                                              int(num_blurs), C.c_void_p(d_img_ptr), C.c_void_p(stream) if stream else None))

    def expose_device_noalloc(self, d_rgb_ptr, width, height, exposure, num_blurs, d_img_ptr, stream=None):
        """expose_device on the library's fixed per-GPU buffer (no device allocation); frames up to NOALLOC_MAX_PIXELS."""
        self._check(self.lib.rt_expose_device_noalloc(self.h, C.c_void_p(d_rgb_ptr), int(width), int(height), float(exposure),
                                                      int(num_blurs), C.c_void_p(d_img_ptr), C.c_void_p(stream) if stream else None))

    def expose_bits_device(self, d_rgb_ptr, width, rows, exposure, d_bits_ptr, stream=None):
        """Step 1 of expose() before the gather: the rank's rows -> one byte per pixel (bit k: channel k over-exposed)."""
        self._check(self.lib.rt_expose_bits_device(self.h, C.c_void_p(d_rgb_ptr), int(width), int(rows), float(exposure),
                                                   C.c_void_p(d_bits_ptr), C.c_void_p(stream) if stream else None))

    def expose_rows_device(self, d_rgb_ptr, d_bits_all_ptr, width, height, tile_rows, rank, world, pad_rows, exposure, num_blurs,
                           d_img_ptr, stream=None):
        """Step 3: bloom (with halo block rows, from every rank's gathered bits) + 8-bit pack of the rank's own rows."""
        self._check(self.lib.rt_expose_rows_device(self.h, C.c_void_p(d_rgb_ptr), C.c_void_p(d_bits_all_ptr) if d_bits_all_ptr else None,
                                                   int(width), int(height), int(tile_rows), int(rank), int(world), int(pad_rows),
                                                   float(exposure), int(num_blurs), C.c_void_p(d_img_ptr),
                                                   C.c_void_p(stream) if stream else None))

    def finish(self):
        rays = C.c_uint64(0)
        self._check(self.lib.rt_finish(self.h, C.byref(rays)))
        return int(rays.value)

    def timing(self):
        t = RtTiming()
        self._check(self.lib.rt_last_timing(self.h, C.byref(t)))
        return {k: getattr(t, k) for k, _ in RtTiming._fields_}

    def selftest_math(self, fn, x, y=None):
        x = np.ascontiguousarray(x, dtype=np.float64)
        y = np.ascontiguousarray(y if y is not None else np.zeros_like(x), dtype=np.float64)
        out = np.zeros_like(x)
        code = {"sin": 0, "cos": 1, "acos": 2, "pow": 3, "div": 4, "sqrt": 5, "rng": 6}[fn]
        self._check(self.lib.rt_selftest_math(self.h, code, x.size, _dp(x), _dp(y), _dp(out)))
        return out

    def selftest_broad_mfma(self, origins, dirs, centres, radii):
        """bool[n_rays, n_rec]: does the matrix-core broad phase keep record j for ray i."""
        o = np.ascontiguousarray(origins, dtype=np.float64); d = np.ascontiguousarray(dirs, dtype=np.float64)
        c = np.ascontiguousarray(centres, dtype=np.float64); r = np.ascontiguousarray(radii, dtype=np.float64)
        groups = (len(r) + 31) // 32
        masks = np.zeros((len(o), groups), dtype=np.uint32)
        self._check(self.lib.rt_selftest_broad_mfma(self.h, len(o), _dp(o), _dp(d), len(r), _dp(c), _dp(r), masks.ctypes.data_as(C.POINTER(C.c_uint32))))
        bits = (masks[:, :, None] >> (31 - np.arange(32, dtype=np.uint32))[None, None, :]) & 1
        return bits.reshape(len(o), groups * 32)[:, : len(r)].astype(bool)

    def selftest_broad(self, origins, dirs, centres, radii):
        o = np.ascontiguousarray(origins, dtype=np.float64); d = np.ascontiguousarray(dirs, dtype=np.float64)
        c = np.ascontiguousarray(centres, dtype=np.float64); r = np.ascontiguousarray(radii, dtype=np.float64)
        flags = np.zeros(len(r), dtype=np.int32)
        self._check(self.lib.rt_selftest_broad(self.h, len(r), _dp(o), _dp(d), _dp(c), _dp(r), flags.ctypes.data_as(C.POINTER(C.c_int32))))
        return flags


class MultiRenderer:
    """One rt_multi_handle: the GPUs 0..n_gpus-1 of this node in one process, rows gathered to GPU 0 inside the library
    (RCCL).  With RT_MULTI_LOOPBACK=1 in the environment the ranks all live on GPU 0 (tests on a one-GPU machine)."""

    def __init__(self, n_gpus: int):
        self.lib = load_library()
        self.h = C.c_void_p()
        rc = self.lib.rt_multi_create(C.byref(self.h), int(n_gpus))
        if rc != RT_OK:
            raise RtError(rc, f"rt_multi_create({n_gpus}) failed: " + self.lib.rt_multi_last_error(None).decode())
        self.scene = None

    def _check(self, rc):
        if rc != RT_OK:
            raise RtError(rc, self.lib.rt_multi_last_error(self.h).decode())

    def close(self):
        if self.h:
            self.lib.rt_multi_destroy(self.h)
            self.h = C.c_void_p()

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass

    def set_option(self, key: int, value: int):
        self._check(self.lib.rt_multi_set_option(self.h, int(key), int(value)))

    def upload(self, sc: S.Scene):
        params = sc.params()
        prims = np.ascontiguousarray(sc.prims, dtype=np.float64)
        mats = np.ascontiguousarray(sc.mats, dtype=np.float64)
        verts = np.ascontiguousarray(sc.verts, dtype=np.float64)
        self._check(self.lib.rt_multi_upload_scene(self.h, _dp(params), len(prims), _dp(prims) if len(prims) else None,
                                                   len(mats), _dp(mats) if len(mats) else None,
                                                   len(verts), _dp(verts) if len(verts) else None))
        self.scene = sc

    def render_image(self, tile_rows: int = 0):
        """Render + expose + gather: dict(img=int32[H,W], rays=int, ms=wall-clock of the call)."""
        sc = self.scene
        if sc is None:
            raise RtError(-3, "no scene uploaded")
        img = np.zeros((sc.height, sc.width), dtype=np.int32)
        rays = C.c_uint64(0)
        self._check(self.lib.rt_multi_render_image(self.h, int(tile_rows), img.ctypes.data_as(C.POINTER(C.c_int32)), C.byref(rays)))
        return dict(img=img, rays=int(rays.value), ms=float(self.lib.rt_multi_last_ms(self.h)))

    def uses_rccl(self) -> bool:
        return self.lib.rt_multi_uses_rccl(self.h) == 1

    def begin(self, tile_rows: int = 0):
        """Enqueue a frame on every rank without waiting (at most two frames in flight)."""
        if self.scene is None:
            raise RtError(-3, "no scene uploaded")
        self._check(self.lib.rt_multi_render_begin(self.h, int(tile_rows)))

    def end(self):
        """Wait for the oldest frame begun: dict(img, rays, ms, enqueue_ms, wait_ms)."""
        sc = self.scene
        img = np.zeros((sc.height, sc.width), dtype=np.int32)
        rays = C.c_uint64(0)
        self._check(self.lib.rt_multi_render_end(self.h, img.ctypes.data_as(C.POINTER(C.c_int32)), C.byref(rays)))
        enq, wait = C.c_double(0), C.c_double(0)
        self.lib.rt_multi_last_times(self.h, C.byref(enq), C.byref(wait))
        return dict(img=img, rays=int(rays.value), ms=float(self.lib.rt_multi_last_ms(self.h)), enqueue_ms=enq.value, wait_ms=wait.value)


# ------------------------------------------------------------------ host post-process
def expose(rgb: np.ndarray, exposure: float, num_blurs: int) -> np.ndarray:
    """expose() + blur() + Color::to_int of the reference (main.cpp:186-242,
    geometry.cpp:15-23) on the host in float64: bloom of over-exposed pixels through
    num_blurs passes of the 41x41 Gaussian (radius 20, sigma 3, zero outside the frame,
    taps summed ky-outer / kx-inner) and truncating 8-bit pack 0xRRGGBB.
    Returns int32[H,W]."""
    H, W, _ = rgb.shape
    col = rgb.astype(np.float64)
    radius, stddev = 20, 3.0
    denom = 2 * stddev * stddev
    import math
    kernel = np.array([[math.exp(-(x * x + y * y) / denom) / denom / math.pi for x in range(-radius, radius + 1)]
                       for y in range(-radius, radius + 1)], dtype=np.float64)
    cur = np.where(col * exposure > 1, float(exposure), 0.0)
    for _ in range(int(num_blurs)):
        pad = np.zeros((H + 2 * radius, W + 2 * radius, 3), dtype=np.float64)
        pad[radius:radius + H, radius:radius + W] = cur
        nxt = np.zeros_like(cur)
        # same tap order as the reference: ky outer, kx inner, one rounded product and one rounded sum per tap
        for ky in range(2 * radius + 1):
            for kx in range(2 * radius + 1):
                nxt = nxt + pad[ky:ky + H, kx:kx + W] * kernel[ky, kx]
        cur = nxt
    col = col + cur
    mx = 1.0 / exposure
    with np.errstate(invalid="ignore", over="ignore"):
        v = (255 * col / mx)
        # (int)v as the reference's x86 build converts it (cvttsd2si): INT_MIN for NaN and for anything outside int range
        in_range = np.isfinite(v) & (v > -2147483649.0) & (v < 2147483648.0)
        iv = np.where(in_range, np.trunc(np.where(in_range, v, 0.0)), -2147483648.0).astype(np.int64)
    iv = np.minimum(iv, 255)
    return ((iv[..., 0] << 16) + (iv[..., 1] << 8) + iv[..., 2]).astype(np.int32)
