// The scene as objects: the host-side value types a caller of the reference knows -- Vector, Point, Color, Material and
// the Solid family Sphere / Plane / Face (/root/reference/src/Vector.hpp:6-20, src/geometry.hpp:5-101) -- with the same
// names and data members, so that code written against the reference's loader (a loop over vector<Solid*> that
// dynamic_casts each solid and reads its members) compiles against this host unchanged.
//
// They carry DATA only.  The reference's methods on them (intersect, normal, get_material, rand_point, Line, Path) are
// the render loop, and that runs on the GPU from the flat tables rt::flatten() makes of these objects.
#pragma once
#include <string>
#include <vector>

struct Vector {
    double x = 0, y = 0, z = 0;
    Vector() = default;
    Vector(double x1, double y1, double z1) : x(x1), y(y1), z(z1) {}
    double& operator[](int i) { return i == 0 ? x : (i == 1 ? y : z); }
    double operator[](int i) const { return i == 0 ? x : (i == 1 ? y : z); }
    // in place, as Vector::normalize (src/Vector.cpp:50-56): every component divided by sqrt of the left-to-right dot
    Vector& normalize();
};

struct Point {
    double x = 0, y = 0, z = 0;
    Point() = default;
    Point(double x1, double y1, double z1) : x(x1), y(y1), z(z1) {}
};

struct Color {
    double r = 0, g = 0, b = 0;
    Color() = default;
    Color(double r1, double g1, double b1) : r(r1), g(g1), b(b1) {}
};

struct Material {                 // field for field struct Material, src/geometry.hpp:18-29
    Color ref;
    bool is_light = false;
    double gloss = 0;
    Color diffuse, specular;
    double specular_exp = 0;
    Color refraction_diffuse, refraction_specular;
    double refraction_specular_exp = 0;
    double refraction_index = 0;
};

struct Solid {
    bool is_light = false;        // the reference leaves it unset for an emitting plane (src/readShape.cpp:75); here: false
    Color color;
    virtual ~Solid() = default;
};

struct Sphere : Solid {
    double radius = 0;
    Point center;
    Material material;
    Sphere(const Point& p, double r) : radius(r), center(p) {}
};

struct Plane : Solid {
    Point point;
    Vector norm;                  // unit length
    Material material1, material2;
    Plane() = default;
    Plane(const Point& p, const Vector& n) : point(p), norm(n) { norm.normalize(); }
};

struct Face : Plane {             // point = verts[0]
    std::vector<Point> verts;     // (the reference holds a Point*: verts[i] reads the same)
    int num = 0;
    Material material;
    // As Face::Face (src/geometry.cpp:152-159): the normal is stored as given, NOT normalised here -- the loader has
    // normalised it once already (src/readShape.cpp:95), and a second pass could move a component by an ulp.
    // The vertices are copied; the caller keeps ownership of `v`.
    Face(const Vector& n, int num_verts, Point* v) : verts(v, v + (num_verts > 0 ? num_verts : 0)), num(num_verts > 0 ? num_verts : 0) {
        point = num ? verts[0] : Point();
        norm = n;
    }
    Face(const Vector& n, const std::vector<Point>& v) : verts(v), num((int)v.size()) {
        point = num ? verts[0] : Point();
        norm = n;
    }
};

// What the reference's loader leaves in globals (src/readShape.cpp:13-24, src/main.cpp:16-32), under the same names.
namespace rt {
namespace scene_globals {
extern int num_lights, num_objs, num_blurs, paths_per_pixel, width, height, max_ray_depth, num_threads;
extern double exposure, scale, lens_radius, focal_length;
extern Point camera;              // the reference's constant (0,0,40) unless the header has the optional key "camera"
}  // namespace scene_globals
}  // namespace rt

// The reference's loader, name and signature (src/readShape.cpp:26, src/rayTracer.hpp:13): reads the scene stream, fills
// rt::scene_globals, returns the solids in file order.  The caller owns the list (rt::free_scene; the reference leaks it).
// Unlike the reference a malformed stream or a missing key throws rt::SceneError instead of aborting.
std::vector<Solid*>* read_json_scene(std::string filename);
