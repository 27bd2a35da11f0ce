"""Synthetic benchmark scenes in the reference's stream format.

The reference ships one scene only (/root/reference/scene.json); the larger
workloads are generated here.  Every scene obeys the rules under which the
reference's behaviour is defined and terminates:

* the closed room of scene.json (checker floor plane y=-20, back / left / right
  wall faces, ceiling face, ceiling face emitters) so that paths find light;
* solids on a jittered lattice, each strictly inside its own cell: nothing
  overlaps, nothing pierces a wall, the camera (0,0,40) is outside everything;
  a glossy surface can therefore not be hit from behind, which is the case in
  which the reference's rejection loop (/root/reference/src/main.cpp:61-76)
  never accepts;
* free-floating quads are purely diffuse (specular 0 => gloss 0 => the branch at
  main.cpp:78-103, which has no loop);
* integer specular exponents, no emitting planes, no face normal (0,0,-1).
"""
from __future__ import annotations

import math
import random

from . import scene as S

ROOM_X = (-20.0, 20.0)
ROOM_Y = (-20.0, 20.0)
ROOM_Z = (-40.0, 0.0)
BOX_LO = (-19.0, -19.0, -39.0)
BOX_HI = (19.0, 18.0, -1.0)


def _room(n_emitters: int, extra_planes: int = 0):
    objs = []
    # With planes behind the walls a ray that slips through a corner of the room (a hit nearer than the 0.001
    # threshold is ignored, geometry.cpp:102) bounces around outside and can meet the floor from below: the floor
    # must then not be glossy, or the reference's rejection loop never accepts.  Without those planes such a ray
    # never comes back and the floor keeps the mirror-like finish of scene.json.
    floor_spec = ([0, 0, 0], [0, 0, 0], 1) if extra_planes else ([0.3, 0.3, 0.3], [0.1, 0.1, 0.1], 1000)
    objs.append({"shape": "plane", "is_light": False, "point": [0, -20, 0], "normal": [0, 1, 0],
                 "diffuse1": [0.7, 0.7, 0.7], "diffuse2": [0.1, 0.1, 0.1], "specular1": floor_spec[0],
                 "specular2": floor_spec[1], "specular_exp": floor_spec[2], "refraction_index": 0})
    # planes behind the walls: tested by every ray, the closest hit only for the rare ray that left the room
    behind = [([0, 0, -60], [0, 0, 1]), ([40, 0, 0], [-1, 0, 0]), ([-40, 0, 0], [1, 0, 0]), ([0, 40, 0], [0, -1, 0]),
              ([0, 0, -80], [0, 0, 1]), ([60, 0, 0], [-1, 0, 0]), ([-60, 0, 0], [1, 0, 0])]
    for pt, n in behind[:extra_planes]:
        objs.append({"shape": "plane", "is_light": False, "point": pt, "normal": n,
                     "diffuse1": [0.5, 0.5, 0.5], "diffuse2": [0.5, 0.5, 0.5], "specular1": [0, 0, 0],
                     "specular2": [0, 0, 0], "specular_exp": 1, "refraction_index": 0})

    def wall(normal, verts, diffuse):
        return {"shape": "face", "is_light": False, "normal": normal, "verts": verts, "diffuse": diffuse,
                "specular": [0.3, 0.3, 0.3], "specular_exp": 1, "refraction_index": 0}
    objs.append(wall([0, 0, 1], [[-20, -20, -40], [-20, 20, -40], [20, 20, -40], [20, -20, -40]], [0.7, 0.7, 0.7]))
    objs.append(wall([-1, 0, 0], [[20, -20, -40], [20, 20, -40], [20, 20, 0], [20, -20, 0]], [0.1, 0.7, 0.1]))
    objs.append(wall([1, 0, 0], [[-20, -20, -40], [-20, 20, -40], [-20, 20, 0], [-20, -20, 0]], [0.7, 0.1, 0.1]))
    objs.append(wall([0, -1, 0], [[-20, 20.01, -40], [-20, 20.01, 0], [20, 20.01, 0], [20, 20.01, -40]], [0.7, 0.7, 0.7]))
    # ceiling emitters: n strips along x
    cols = [[5, 5, 5], [5, 4, 3], [4, 5, 5], [5, 5, 4], [5, 4, 5], [4, 4, 5], [5, 5, 3], [3, 5, 5]]
    for k in range(n_emitters):
        x0 = -20 + 40.0 * k / n_emitters
        x1 = -20 + 40.0 * (k + 1) / n_emitters
        objs.append({"shape": "face", "is_light": True, "normal": [0, -1, 0],
                     "verts": [[x0, 20, -40], [x1, 20, -40], [x1, 20, 0], [x0, 20, 0]], "color": cols[k % len(cols)]})
    return objs


def _lattice_objects(rng, nx, ny, nz, count, glass_frac, quad_frac):
    cell = [(BOX_HI[i] - BOX_LO[i]) / n for i, n in enumerate((nx, ny, nz))]
    cmin = min(cell)
    cells = [(i, j, k) for i in range(nx) for j in range(ny) for k in range(nz)]
    if count < len(cells):
        rng.shuffle(cells)
        cells = sorted(cells[:count])
    objs = []
    for (i, j, k) in cells:
        lo = [BOX_LO[0] + i * cell[0], BOX_LO[1] + j * cell[1], BOX_LO[2] + k * cell[2]]
        r = cmin * rng.uniform(0.15, 0.3)
        c = [lo[a] + r + rng.random() * (cell[a] - 2 * r) for a in range(3)]
        kind = rng.random()
        if kind < quad_frac:
            # square of half-diagonal r, random orientation, inside the sphere of radius r about c
            while True:
                u = [rng.gauss(0, 1) for _ in range(3)]
                lu = math.sqrt(sum(x * x for x in u))
                w = [rng.gauss(0, 1) for _ in range(3)]
                if lu < 1e-3:
                    continue
                u = [x / lu for x in u]
                d = sum(a * b for a, b in zip(u, w))
                v = [a - d * b for a, b in zip(w, u)]
                lv = math.sqrt(sum(x * x for x in v))
                if lv < 1e-3:
                    continue
                v = [x / lv for x in v]
                n = [u[1] * v[2] - u[2] * v[1], u[2] * v[0] - u[0] * v[2], u[0] * v[1] - u[1] * v[0]]
                if abs(n[2] + 1) > 1e-2 and abs(n[2] - 1) > 1e-2:
                    break
            verts = [[c[a] + r * (su * u[a] + sv * v[a]) for a in range(3)]
                     for su, sv in ((1, 0), (0, 1), (-1, 0), (0, -1))]
            objs.append({"shape": "face", "is_light": False, "normal": n, "verts": verts,
                         "diffuse": [rng.uniform(0, 0.8) for _ in range(3)], "specular": [0, 0, 0],
                         "specular_exp": 1, "refraction_index": 0})
        else:
            glass = rng.random() < glass_frac
            o = {"shape": "sphere", "is_light": False, "center": c, "radius": r}
            if glass:
                # a transparent solid can be met from inside with the path believing it is outside (an entry whose
                # chord is shorter than the 0.001 hit threshold is never left): like every surface that can be hit
                # from behind it must not be glossy, or the reference's rejection loop (main.cpp:61-76) never ends
                o.update({"diffuse": [0.4, 0.4, 0.4], "specular": [0, 0, 0], "specular_exp": 1, "refraction_index": 1.5,
                          "refraction_diffuse": [0, 0, 0], "refraction_specular": [1.5, 1.5, 1.5], "refraction_specular_exp": 100})
            else:
                o.update({"diffuse": [rng.uniform(0, 0.8) for _ in range(3)], "specular": [0.3, 0.3, 0.3], "specular_exp": 50,
                          "refraction_index": 0, "refraction_diffuse": [0, 0, 0], "refraction_specular": [0, 0, 0],
                          "refraction_specular_exp": 0})
            objs.append(o)
    return objs


# name -> (width, height, lattice, n_lattice_objs|None, emitters, extra planes, depth, spp, lens, glass, quads)
CONFIGS = {
    # 1920x1080, 64 spheres, 2 lights, camera rays + one more segment
    "c2": dict(width=1920, height=1080, lattice=(4, 4, 4), count=None, emitters=2, planes=0, depth=1, spp=1, lens=0.0, glass=0.0, quads=0.0),
    # 3840x2160, 1024 spheres + room, depth 4
    "c3": dict(width=3840, height=2160, lattice=(16, 8, 8), count=None, emitters=2, planes=0, depth=4, spp=1, lens=1.0, glass=0.05, quads=0.0),
    # 7680x4320, 4096 spheres + room, depth 8
    "c4": dict(width=7680, height=4320, lattice=(16, 16, 16), count=None, emitters=2, planes=0, depth=8, spp=1, lens=1.0, glass=0.05, quads=0.0),
    # 3840x2160, 16384 mixed primitives, 8 lights, 16 samples per pixel
    "c5": dict(width=3840, height=2160, lattice=(32, 16, 32), count=16384 - 20, emitters=8, planes=7, depth=4, spp=16, lens=1.0, glass=0.05, quads=0.25),
}


def make(config="c3", width=None, height=None, spp=None, depth=None, lattice=None, seed=1234, num_threads=8,
         num_blurs=0, frame_seed=0, lens=None):
    """Return (scene_text, Scene).  ``width``/``height`` rescale the frame with identical framing
    (the pixel->world map is (x - W/2)/scale, /root/reference/src/main.cpp:257, so scale = width/40)."""
    cfg = dict(CONFIGS[config])
    if width is not None:
        cfg["width"] = int(width)
        cfg["height"] = int(height if height is not None else round(width * 9 / 16))
    if spp is not None:
        cfg["spp"] = int(spp)
    if depth is not None:
        cfg["depth"] = int(depth)
    if lattice is not None:
        cfg["lattice"] = tuple(lattice)
        cfg["count"] = None
    if lens is not None:
        cfg["lens"] = float(lens)
    rng = random.Random(seed)
    objs = _room(cfg["emitters"], cfg["planes"])
    nx, ny, nz = cfg["lattice"]
    count = cfg["count"] if cfg["count"] is not None else nx * ny * nz
    objs += _lattice_objects(rng, nx, ny, nz, count, cfg["glass"], cfg["quads"])
    header = {"num_threads": num_threads, "num_objs": len(objs), "exposure": 1, "num_blurs": num_blurs,
              "paths_per_pixel": cfg["spp"], "max_ray_depth": cfg["depth"], "scale": cfg["width"] / 40.0,
              "width": cfg["width"], "height": cfg["height"], "lens_radius": cfg["lens"], "focal_length": 60}
    text = S.dumps(header, objs)
    return text, S.loads(text, frame_seed=frame_seed)
