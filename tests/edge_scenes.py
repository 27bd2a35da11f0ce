"""Small scenes around the edges of the scene format and of the path logic
(/root/reference/src/readShape.cpp:26-122, src/main.cpp:126-183): empty list, no emitter, every emitter kind,
triangles and pentagons, faces whose accepted region is not their polygon, a face that is neither planar nor convex, a solid larger than the room, glass of index 1,
depth 0 and the maximum depth, 1x1 and odd frame sizes.  Surfaces that can be met from behind are purely diffuse
(a glossy one makes the reference's rejection loop spin, main.cpp:61-76).  Used by the GPU parity test, by the test
that runs the compiled reference on them, and by tests/golden/make_golden.py."""
import json

DIFF = dict(diffuse=[0.7, 0.6, 0.5], specular=[0, 0, 0], specular_exp=1)


def _header(n, width=48, height=27, spp=2, depth=3, lens=0.5, **kw):
    h = dict(num_threads=4, num_objs=n, exposure=1, num_blurs=0, paths_per_pixel=spp, max_ray_depth=depth,
             scale=width / 40.0, width=width, height=height, lens_radius=lens, focal_length=60)
    h.update(kw)
    return h


def _sphere(c, r, light=False, glass=None, **mat):
    o = dict(shape="sphere", is_light=light, center=list(c), radius=r)
    if light:
        o["color"] = [4, 3, 2]
    else:
        m = dict(DIFF, **mat)
        o.update(diffuse=m["diffuse"], specular=m["specular"], specular_exp=m["specular_exp"],
                 refraction_diffuse=[0.9, 0.9, 0.9], refraction_specular=[0, 0, 0], refraction_specular_exp=1,
                 refraction_index=0 if glass is None else glass)
    return o


def _face(verts, normal, light=False, **mat):
    o = dict(shape="face", is_light=light, normal=list(normal), verts=[list(v) for v in verts])
    if light:
        o["color"] = [5, 5, 4]
    else:
        o.update(dict(DIFF, **mat))
    return o


def _plane(point, normal, light=False):
    return dict(shape="plane", is_light=light, point=list(point), normal=list(normal), diffuse1=[0.8, 0.8, 0.8],
                diffuse2=[0.2, 0.3, 0.2], specular1=[0, 0, 0], specular2=[0, 0, 0], specular_exp=1)


FLOOR = _plane((0, -12, 0), (0, 1, 0))
CEIL_LIGHT = _face([(-15, 14, -30), (15, 14, -30), (15, 14, 5), (-15, 14, 5)], (0, -1, 0), light=True)


def _scenes():
    yield "empty", _header(0), []
    yield "no_emitter_open", _header(3, depth=2), [FLOOR, _sphere((0, -6, -10), 5), _sphere((9, -8, -6), 3)]
    yield "sphere_emitter", _header(3), [FLOOR, _sphere((0, 12, -10), 4, light=True), _sphere((0, -6, -10), 5)]
    yield "plane_flagged_emitter", _header(3), [_plane((0, -12, 0), (0, 1, 0), light=True), CEIL_LIGHT, _sphere((0, -6, -10), 5)]
    tri = _face([(-10, -11, -18), (10, -11, -18), (0, 8, -14)], (0, 0.2, 1))
    pent = _face([(12, -10, -8), (18, -6, -8), (16, 2, -8), (9, 3, -8), (7, -5, -8)], (0, 0, 1))
    yield "triangle_and_pentagon", _header(4), [FLOOR, CEIL_LIGHT, tri, pent]
    bent = _face([(-14, -10, -12), (-4, -10, -16), (-9, -2, -12), (-4, 6, -16), (-14, 6, -12)], (0, 0, 1))   # not planar, not convex
    yield "bent_face", _header(4), [FLOOR, CEIL_LIGHT, bent, _sphere((8, -7, -8), 4)]
    yield "huge_sphere", _header(4), [FLOOR, CEIL_LIGHT, _sphere((0, -80, -20), 70), _sphere((6, -5, -8), 3)]
    yield "glass_index_one", _header(4, depth=4), [FLOOR, CEIL_LIGHT, _sphere((0, -5, -10), 6, glass=1.0), _sphere((10, -8, -4), 3, glass=1.5)]
    yield "depth_zero", _header(3, depth=0), [FLOOR, CEIL_LIGHT, _sphere((0, -6, -10), 5)]
    yield "depth_max", _header(3, depth=16, spp=1), [FLOOR, CEIL_LIGHT, _sphere((0, -6, -10), 5, glass=1.5)]
    yield "one_pixel", _header(3, width=1, height=1, spp=5), [FLOOR, CEIL_LIGHT, _sphere((0, -6, -10), 5)]
    yield "odd_frame_no_lens", _header(3, width=37, height=23, spp=3, lens=0.0), [FLOOR, CEIL_LIGHT, _sphere((0, -6, -10), 5)]
    # Planar faces with exact normals whose accepted region (Face::intersect tests, per vertex, a half-plane bordered
    # by the line through v_i perpendicular to edge i, geometry.cpp:162-166) reaches far outside the polygon: camera and
    # bounce rays cross their planes outside the polygon and still hit.
    rtri = _face([(-8, -8, -14), (4, -8, -14), (-8, 6, -14)], (0, 0, 1))                      # right triangle: unbounded region
    yield "right_triangle_region", _header(4), [FLOOR, CEIL_LIGHT, rtri, _sphere((9, -8, -6), 3)]
    atri = _face([(-6, -9, -10), (8, -9, -10), (1, 7, -10)], (0, 0, 1))                       # acute triangle: a larger triangle
    para = _face([(-16, -2, -20), (-8, -2, -20), (-4, 6, -20), (-12, 6, -20)], (0, 0, 1))     # parallelogram
    yield "acute_triangle_and_parallelogram", _header(5, width=64, height=36), [FLOOR, CEIL_LIGHT, atri, para, _sphere((12, -8, -4), 3)]
    small = [_face([(x, y, -12), (x + 1.5, y, -12), (x + 0.75, y + 1.2, -12)], (0, 0, 1)) for x in (-12, -6, 0, 6, 12) for y in (-9, -4, 1)]
    yield "small_acute_triangles", _header(2 + len(small), width=64, height=36, depth=2), [FLOOR, CEIL_LIGHT] + small
    # Room walls as the always-tested "box" solids see them (axis-aligned normal, edges along the axes: the device tests
    # them by coordinate ranges) and walls that must NOT qualify: an edge shorter than 1, vertices in the other sense
    # of rotation, a normal along -z with a negative-x wall, a wall whose normal is not its vertices' plane.
    strip = _face([(-18, -10, -25), (-17.5, -10, -25), (-17.5, 12, -25), (-18, 12, -25)], (0, 0, 1))        # 0.5 wide: general test
    back_cw = _face([(-20, -12, -30), (-20, 14, -30), (20, 14, -30), (20, -12, -30)], (0, 0, 1))            # clockwise order
    left = _face([(-20, -12, -30), (-20, 14, -30), (-20, 14, 10), (-20, -12, 10)], (1, 0, 0))
    right_neg = _face([(20, -12, 10), (20, 14, 10), (20, 14, -30), (20, -12, -30)], (-1, 0, 0))
    tilted_normal = _face([(5, -12, -29), (12, -12, -29), (12, 2, -29), (5, 2, -29)], (0.1, 0, 1))          # normal off the axis
    yield "box_walls_and_non_box_walls", _header(9, width=64, height=36), [FLOOR, CEIL_LIGHT, back_cw, left, right_neg, strip,
                                                                            tilted_normal, _sphere((0, -7, -12), 5), _sphere((-9, -8, -6), 3, glass=1.5)]
    # A tilted plane whose normal changes under a second normalisation: the reference normalises a plane's normal in its
    # loader and again in Plane::Plane (readShape.cpp:73, geometry.cpp:97); for (1, 20, -1) the second pass moves every
    # component by an ulp.  The tilted face beside it carries the same normal, which is normalised once (readShape.cpp:95).
    # (Lit from close by so that paths off the plane reach an emitter: with the once-normalised normal a dozen radiance
    # doubles of this frame differ from the reference's.)
    tilted_floor = _plane((0, -8, 0), (1, 20, -1))
    tilted_face = _face([(-9, -3, -16), (9, -2.1, -16), (9.4, 5.9, -16), (-8.6, 5, -16)], (1, 20, -1))
    yield "tilted_plane_normal_normalised_twice", _header(5, width=64, height=36), [tilted_floor, CEIL_LIGHT, _sphere((0, 16, -10), 9, light=True),
                                                                                     tilted_face, _sphere((2, -3, -9), 4)]
    many = [FLOOR, CEIL_LIGHT] + [_sphere((-18 + 1.2 * (i % 31), -10 + 1.2 * (i // 31), -12 - 0.1 * i), 0.5) for i in range(90)]
    yield "row_of_spheres_overflowing_lists", _header(len(many), width=64, height=36, depth=2, lens=0.0), many


SCENES = list(_scenes())




def text_of(header, objs):
    """The scene stream the reference reads: header object followed by the solids."""
    return " ".join(json.dumps(o) for o in [header] + objs)
