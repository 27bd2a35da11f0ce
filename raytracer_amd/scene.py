"""Scene stream reader and flat scene tables.

Python mirror of the reference loader ``read_json_scene``
(/root/reference/src/readShape.cpp:26-122): the file is a stream of
``1 + num_objs`` concatenated top-level JSON objects (a header, then one object
per solid), not a JSON array.  The loader's derived values are reproduced with
the same double-precision operations: normals are normalised on load
(readShape.cpp:73,95 -> /root/reference/src/Vector.cpp:50-56) and
``gloss = spec/(diff+spec)`` with left-to-right component sums
(readShape.cpp:62-64).

The result is three flat float64 tables plus a parameter vector; the C ABI
(include/rt_hip.h) takes exactly these tables.
"""
from __future__ import annotations

import json
import math
from dataclasses import dataclass, field

import numpy as np

KIND_SPHERE, KIND_PLANE, KIND_FACE = 0, 1, 2

# prim record layout (float64 columns) -- keep in step with include/rt_hip.h
PR_KIND, PR_LIGHT, PR_MAT0, PR_MAT1, PR_VOFF, PR_VCNT, PR_A, PR_N, PR_RADIUS, PR_STRIDE = 0, 1, 2, 3, 4, 5, 6, 9, 12, 13
# material record layout
MT_REF, MT_LIGHT, MT_GLOSS, MT_DIFF, MT_SPEC, MT_SEXP, MT_RDIFF, MT_RSPEC, MT_RSEXP, MT_RINDEX, MT_STRIDE = 0, 3, 4, 5, 8, 11, 12, 15, 18, 19, 20
# parameter vector layout
(PA_WIDTH, PA_HEIGHT, PA_SPP, PA_DEPTH, PA_THREADS, PA_BLURS, PA_SCALE, PA_LENS, PA_FOCAL, PA_EXPOSURE, PA_SEED,
 PA_CAM_X, PA_CAM_Y, PA_CAM_Z, PA_COUNT) = range(15)
DEFAULT_CAMERA = (0.0, 0.0, 40.0)      # Point camera(0,0,40), /root/reference/src/main.cpp:16

HEADER_KEYS = ("exposure", "num_blurs", "num_objs", "paths_per_pixel", "width", "height", "scale",
               "lens_radius", "focal_length", "max_ray_depth", "num_threads")


class SceneError(ValueError):
    pass


@dataclass
class Scene:
    header: dict
    prims: np.ndarray = field(default_factory=lambda: np.zeros((0, PR_STRIDE)))
    mats: np.ndarray = field(default_factory=lambda: np.zeros((0, MT_STRIDE)))
    verts: np.ndarray = field(default_factory=lambda: np.zeros((0, 3)))
    frame_seed: int = 0

    @property
    def width(self) -> int:
        return int(self.header["width"])

    @property
    def height(self) -> int:
        return int(self.header["height"])

    @property
    def spp(self) -> int:
        return int(self.header["paths_per_pixel"])

    def params(self) -> np.ndarray:
        h = self.header
        p = np.zeros(PA_COUNT, dtype=np.float64)
        p[PA_WIDTH] = h["width"]; p[PA_HEIGHT] = h["height"]; p[PA_SPP] = h["paths_per_pixel"]
        p[PA_DEPTH] = h["max_ray_depth"]; p[PA_THREADS] = h["num_threads"]; p[PA_BLURS] = h["num_blurs"]
        p[PA_SCALE] = h["scale"]; p[PA_LENS] = h["lens_radius"]; p[PA_FOCAL] = h["focal_length"]
        p[PA_EXPOSURE] = h["exposure"]; p[PA_SEED] = self.frame_seed
        # optional header key "camera": [x, y, z] (an extension; the reference ignores unknown keys and uses its constant)
        p[PA_CAM_X:PA_CAM_Z + 1] = h.get("camera", DEFAULT_CAMERA)
        return p

    def counts(self):
        k = self.prims[:, PR_KIND].astype(int) if len(self.prims) else np.zeros(0, int)
        return {"spheres": int((k == KIND_SPHERE).sum()), "planes": int((k == KIND_PLANE).sum()),
                "faces": int((k == KIND_FACE).sum())}


def _normalize(n):
    # Vector::normalize: l = sqrt(((0 + x*x) + y*y) + z*z); each component / l
    t = 0.0
    for c in n:
        t += c * c
    l = math.sqrt(t)
    return [_div(c, l) for c in n]


def _div(a, b):
    try:
        return a / b
    except ZeroDivisionError:
        if a == 0 or a != a:
            return float("nan")
        return math.copysign(float("inf"), a) * math.copysign(1.0, b)


def _gloss(diffuse, specular):
    diff = diffuse[0] + diffuse[1] + diffuse[2]
    spec = specular[0] + specular[1] + specular[2]
    return _div(spec, diff + spec)


def _vec3(obj, key):
    try:
        v = obj[key]
    except KeyError:
        raise SceneError(f"missing key {key!r}") from None
    if not isinstance(v, list) or len(v) < 3:
        raise SceneError(f"key {key!r} must be an array of 3 numbers")
    return [float(v[0]), float(v[1]), float(v[2])]


def _num(obj, key):
    try:
        return float(obj[key])
    except KeyError:
        raise SceneError(f"missing key {key!r}") from None


def _material(ref=(0, 0, 0), is_light=False, gloss=0.0, diffuse=(0, 0, 0), specular=(0, 0, 0), sexp=0.0,
              rdiffuse=(0, 0, 0), rspecular=(0, 0, 0), rsexp=0.0, rindex=0.0):
    m = np.zeros(MT_STRIDE)
    m[MT_REF:MT_REF + 3] = ref; m[MT_LIGHT] = 1.0 if is_light else 0.0; m[MT_GLOSS] = gloss
    m[MT_DIFF:MT_DIFF + 3] = diffuse; m[MT_SPEC:MT_SPEC + 3] = specular; m[MT_SEXP] = sexp
    m[MT_RDIFF:MT_RDIFF + 3] = rdiffuse; m[MT_RSPEC:MT_RSPEC + 3] = rspecular; m[MT_RSEXP] = rsexp
    m[MT_RINDEX] = rindex
    return m


def parse_stream(text: str):
    """Split a stream of concatenated JSON values into Python objects."""
    dec = json.JSONDecoder()
    i, n, out = 0, len(text), []
    while True:
        while i < n and text[i].isspace():
            i += 1
        if i >= n:
            break
        try:
            obj, i = dec.raw_decode(text, i)
        except json.JSONDecodeError as e:
            raise SceneError(f"malformed scene stream: {e}") from None
        out.append(obj)
    return out


def from_objects(header: dict, objects: list, frame_seed: int = 0) -> Scene:
    for k in HEADER_KEYS:
        if k not in header:
            raise SceneError(f"header is missing {k!r}")
    num_objs = int(header["num_objs"])
    if len(objects) < num_objs:
        raise SceneError(f"header announces {num_objs} objects, stream holds {len(objects)}")
    prims, mats, verts = [], [], []
    for obj in objects[:num_objs]:
        shape = obj.get("shape")
        if shape is None or "is_light" not in obj:
            raise SceneError("object needs 'shape' and 'is_light'")
        light = bool(obj["is_light"])
        p = np.zeros(PR_STRIDE)
        if shape == "sphere":                                   # readShape.cpp:47-67
            p[PR_KIND] = KIND_SPHERE
            p[PR_A:PR_A + 3] = _vec3(obj, "center")
            p[PR_RADIUS] = _num(obj, "radius")
            p[PR_LIGHT] = 1.0 if light else 0.0
            if light:
                m = _material(ref=_vec3(obj, "color"), is_light=True)
            else:
                d, s = _vec3(obj, "diffuse"), _vec3(obj, "specular")
                m = _material(gloss=_gloss(d, s), diffuse=d, specular=s, sexp=_num(obj, "specular_exp"),
                              rdiffuse=_vec3(obj, "refraction_diffuse"), rspecular=_vec3(obj, "refraction_specular"),
                              rsexp=_num(obj, "refraction_specular_exp"), rindex=_num(obj, "refraction_index"))
            p[PR_MAT0] = p[PR_MAT1] = len(mats)
            mats.append(m)
        elif shape == "plane":                                  # readShape.cpp:68-90
            p[PR_KIND] = KIND_PLANE
            p[PR_A:PR_A + 3] = _vec3(obj, "point")
            # normalised twice, by the loader (readShape.cpp:73) and by Plane::Plane (geometry.cpp:97); the second pass can
            # move a component by an ulp.  A face's normal (below) is normalised once (readShape.cpp:95).
            p[PR_N:PR_N + 3] = _normalize(_normalize(_vec3(obj, "normal")))
            # Solid::is_light is never written for an emitting plane (readShape.cpp:75): treated as false
            p[PR_LIGHT] = 0.0
            if light:
                m1 = m2 = _material(is_light=True)
            else:
                d, s = _vec3(obj, "diffuse1"), _vec3(obj, "specular1")
                m1 = _material(gloss=_gloss(d, s), diffuse=d, specular=s, sexp=_num(obj, "specular_exp"))
                d, s = _vec3(obj, "diffuse2"), _vec3(obj, "specular2")
                m2 = _material(gloss=_gloss(d, s), diffuse=d, specular=s, sexp=_num(obj, "specular_exp"))
            p[PR_MAT0] = len(mats); mats.append(m1)
            p[PR_MAT1] = len(mats); mats.append(m2)
        else:                                                   # face, readShape.cpp:91-119
            p[PR_KIND] = KIND_FACE
            p[PR_N:PR_N + 3] = _normalize(_vec3(obj, "normal"))
            vs = obj.get("verts")
            if not isinstance(vs, list) or len(vs) < 1:
                raise SceneError("face needs 'verts'")
            p[PR_VOFF] = len(verts); p[PR_VCNT] = len(vs)
            for v in vs:
                verts.append([float(v[0]), float(v[1]), float(v[2])])
            p[PR_A:PR_A + 3] = verts[int(p[PR_VOFF])]
            p[PR_LIGHT] = 1.0 if light else 0.0
            if light:
                m = _material(ref=_vec3(obj, "color"), is_light=True)
            else:
                d, s = _vec3(obj, "diffuse"), _vec3(obj, "specular")
                m = _material(gloss=_gloss(d, s), diffuse=d, specular=s, sexp=_num(obj, "specular_exp"))
            p[PR_MAT0] = p[PR_MAT1] = len(mats)
            mats.append(m)
        prims.append(p)
    sc = Scene(header={k: header[k] for k in HEADER_KEYS}, frame_seed=frame_seed)
    if "camera" in header:
        sc.header["camera"] = _vec3(header, "camera")
    sc.prims = np.ascontiguousarray(np.array(prims, dtype=np.float64).reshape(-1, PR_STRIDE))
    sc.mats = np.ascontiguousarray(np.array(mats, dtype=np.float64).reshape(-1, MT_STRIDE))
    sc.verts = np.ascontiguousarray(np.array(verts, dtype=np.float64).reshape(-1, 3))
    return sc


def loads(text: str, frame_seed: int = 0) -> Scene:
    objs = parse_stream(text)
    if not objs:
        raise SceneError("empty scene stream")
    return from_objects(objs[0], objs[1:], frame_seed)


def read_json_scene(filename: str, frame_seed: int = 0) -> Scene:
    """Same name and role as the reference's loader (readShape.cpp:26)."""
    with open(filename, "r") as f:
        return loads(f.read(), frame_seed)


def dumps(header: dict, objects: list) -> str:
    """Serialise to the reference's stream format (header object, then one object per solid)."""
    h = dict(header)
    h["num_objs"] = len(objects)
    return "\n".join(json.dumps(o) for o in [h] + list(objects)) + "\n"
