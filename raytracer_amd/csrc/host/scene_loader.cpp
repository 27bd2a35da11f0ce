// Scene stream reader: a small JSON value parser plus the table builder that mirrors
// /root/reference/src/readShape.cpp:26-122 (same keys, same derived values, same object order).
#include <cctype>
#include <cmath>
#include <cstdlib>
#include <fstream>
#include <map>
#include <memory>
#include <sstream>

#include "../../../include/rt_hip.h"
#include "rt_host.hpp"

namespace rt {
namespace {

struct JVal;
using JPtr = std::shared_ptr<JVal>;
struct JVal {
    enum Kind { NUL, BOOL, NUM, STR, ARR, OBJ } kind = NUL;
    bool b = false;
    double num = 0;
    std::string str;
    std::vector<JPtr> arr;
    std::map<std::string, JPtr> obj;
};

struct Parser {
    const std::string& s;
    size_t i = 0;
    explicit Parser(const std::string& t) : s(t) {}
    [[noreturn]] void fail(const std::string& what) const {
        throw SceneError("malformed scene stream at byte " + std::to_string(i) + ": " + what);
    }
    void ws() { while (i < s.size() && std::isspace((unsigned char)s[i])) i++; }
    bool at_end() { ws(); return i >= s.size(); }
    JPtr value() {
        ws();
        if (i >= s.size()) fail("unexpected end");
        char c = s[i];
        auto v = std::make_shared<JVal>();
        if (c == '{') {
            v->kind = JVal::OBJ; i++; ws();
            if (i < s.size() && s[i] == '}') { i++; return v; }
            for (;;) {
                ws();
                if (i >= s.size() || s[i] != '"') fail("expected a key");
                std::string k = string();
                ws();
                if (i >= s.size() || s[i] != ':') fail("expected ':'");
                i++;
                v->obj[k] = value();
                ws();
                if (i < s.size() && s[i] == ',') { i++; continue; }
                if (i < s.size() && s[i] == '}') { i++; return v; }
                fail("expected ',' or '}'");
            }
        }
        if (c == '[') {
            v->kind = JVal::ARR; i++; ws();
            if (i < s.size() && s[i] == ']') { i++; return v; }
            for (;;) {
                v->arr.push_back(value());
                ws();
                if (i < s.size() && s[i] == ',') { i++; continue; }
                if (i < s.size() && s[i] == ']') { i++; return v; }
                fail("expected ',' or ']'");
            }
        }
        if (c == '"') { v->kind = JVal::STR; v->str = string(); return v; }
        if (s.compare(i, 4, "true") == 0) { v->kind = JVal::BOOL; v->b = true; i += 4; return v; }
        if (s.compare(i, 5, "false") == 0) { v->kind = JVal::BOOL; v->b = false; i += 5; return v; }
        if (s.compare(i, 4, "null") == 0) { i += 4; return v; }
        // number: strtod is correctly rounded, as is the reference's parser for these inputs
        const char* b = s.c_str() + i;
        char* e = nullptr;
        double d = std::strtod(b, &e);
        if (e == b) fail("unexpected character");
        v->kind = JVal::NUM; v->num = d; i += (size_t)(e - b);
        return v;
    }
    std::string string() {
        std::string out;
        i++;  // opening quote
        while (i < s.size() && s[i] != '"') {
            if (s[i] == '\\' && i + 1 < s.size()) {
                char n = s[i + 1];
                switch (n) {
                    case 'n': out += '\n'; break;
                    case 't': out += '\t'; break;
                    case 'r': out += '\r'; break;
                    case 'b': out += '\b'; break;
                    case 'f': out += '\f'; break;
                    case 'u': if (i + 5 < s.size()) { out += '?'; i += 4; } break;
                    default: out += n;
                }
                i += 2;
            } else out += s[i++];
        }
        if (i >= s.size()) fail("unterminated string");
        i++;
        return out;
    }
};

const JVal& key(const JVal& o, const char* k) {
    auto it = o.obj.find(k);
    if (o.kind != JVal::OBJ || it == o.obj.end()) throw SceneError(std::string("missing key '") + k + "'");
    return *it->second;
}
double num(const JVal& o, const char* k) {
    const JVal& v = key(o, k);
    if (v.kind == JVal::NUM) return v.num;
    if (v.kind == JVal::BOOL) return v.b ? 1.0 : 0.0;
    throw SceneError(std::string("key '") + k + "' must be a number");
}
bool truth(const JVal& o, const char* k) {
    const JVal& v = key(o, k);
    if (v.kind == JVal::BOOL) return v.b;
    if (v.kind == JVal::NUM) return v.num != 0;
    throw SceneError(std::string("key '") + k + "' must be a boolean");
}
void vec3(const JVal& o, const char* k, double out[3]) {
    const JVal& v = key(o, k);
    if (v.kind != JVal::ARR || v.arr.size() < 3) throw SceneError(std::string("key '") + k + "' must be an array of 3 numbers");
    for (int i = 0; i < 3; i++) {
        if (v.arr[i]->kind != JVal::NUM) throw SceneError(std::string("key '") + k + "' must hold numbers");
        out[i] = v.arr[i]->num;
    }
}
double gloss(const Color& d, const Color& s) {   // readShape.cpp:62-64
    double diff = d.r + d.g + d.b;
    double spec = s.r + s.g + s.b;
    return spec / (diff + spec);
}
Color color_of(const JVal& o, const char* k) { double c[3]; vec3(o, k, c); return Color(c[0], c[1], c[2]); }

}  // namespace

namespace scene_globals {
int num_lights = 0, num_objs = 0, num_blurs = 0, paths_per_pixel = 1, width = 0, height = 0, max_ray_depth = 0, num_threads = 1;
double exposure = 1, scale = 1, lens_radius = 0, focal_length = 0;
Point camera(0, 0, 40);
}  // namespace scene_globals

int SceneTables::n_prims() const { return (int)(prims.size() / RT_PR_STRIDE); }
int SceneTables::n_mats() const { return (int)(mats.size() / RT_MT_STRIDE); }
int SceneTables::n_verts() const { return (int)(verts.size() / 3); }
int SceneTables::width() const { return (int)params[RT_PA_WIDTH]; }
int SceneTables::height() const { return (int)params[RT_PA_HEIGHT]; }
double SceneTables::exposure() const { return params[RT_PA_EXPOSURE]; }
int SceneTables::num_blurs() const { return (int)params[RT_PA_BLURS]; }

void free_scene(std::vector<Solid*>* objs) {
    if (!objs) return;
    for (Solid* s : *objs) delete s;
    delete objs;
}

std::vector<Solid*>* parse_scene_text(const std::string& text) {
    namespace G = scene_globals;
    Parser p(text);
    if (p.at_end()) throw SceneError("empty scene stream");
    JPtr head = p.value();
    G::exposure = num(*head, "exposure");
    G::num_blurs = (int)num(*head, "num_blurs");
    G::num_lights = 0;
    G::num_objs = (int)num(*head, "num_objs");
    G::paths_per_pixel = (int)num(*head, "paths_per_pixel");
    G::width = (int)num(*head, "width");
    G::height = (int)num(*head, "height");
    G::scale = num(*head, "scale");
    G::lens_radius = num(*head, "lens_radius");
    G::focal_length = num(*head, "focal_length");
    G::max_ray_depth = (int)num(*head, "max_ray_depth");
    G::num_threads = (int)num(*head, "num_threads");
    // optional extension: "camera": [x, y, z]; the reference's constant (main.cpp:16) otherwise
    G::camera = Point(0, 0, 40);
    if (head->obj.find("camera") != head->obj.end()) {
        double cam[3];
        vec3(*head, "camera", cam);
        G::camera = Point(cam[0], cam[1], cam[2]);
    }
    std::vector<Solid*>* objs = new std::vector<Solid*>;
    try {
        for (int i = 0; i < G::num_objs; i++) {
            if (p.at_end()) throw SceneError("header announces " + std::to_string(G::num_objs) + " objects, stream holds " + std::to_string(i));
            JPtr op = p.value();
            const JVal& o = *op;
            const JVal& shape = key(o, "shape");
            if (shape.kind != JVal::STR) throw SceneError("'shape' must be a string");
            const bool light = truth(o, "is_light");
            if (light) G::num_lights++;
            double a[3], n[3];
            if (shape.str == "sphere") {                                    // readShape.cpp:47-67
                vec3(o, "center", a);
                Sphere* s = new Sphere(Point(a[0], a[1], a[2]), num(o, "radius"));
                objs->push_back(s);
                s->is_light = light;
                if (light) {
                    s->color = color_of(o, "color");
                    s->material.ref = s->color; s->material.is_light = true;
                } else {
                    Material& m = s->material;
                    m.diffuse = color_of(o, "diffuse"); m.specular = color_of(o, "specular");
                    m.refraction_diffuse = color_of(o, "refraction_diffuse"); m.refraction_specular = color_of(o, "refraction_specular");
                    m.gloss = gloss(m.diffuse, m.specular);
                    m.specular_exp = num(o, "specular_exp");
                    m.refraction_specular_exp = num(o, "refraction_specular_exp");
                    m.refraction_index = num(o, "refraction_index");
                }
            } else if (shape.str == "plane") {                              // readShape.cpp:68-90
                vec3(o, "point", a); vec3(o, "normal", n);
                // a plane's normal is normalised twice: by the loader (readShape.cpp:73) and again by Plane::Plane
                // (geometry.cpp:97).  The second pass is not always the identity: where the first result's squared
                // length rounds to 1 - 2^-53 it moves components by an ulp.  (A face's normal is normalised once, :95.)
                Vector nv(n[0], n[1], n[2]);
                nv.normalize();
                Plane* pl = new Plane(Point(a[0], a[1], a[2]), nv);
                objs->push_back(pl);
                pl->is_light = false;   // Solid::is_light is never written for an emitting plane (readShape.cpp:75)
                if (light) {
                    pl->material1.is_light = pl->material2.is_light = true;
                } else {
                    Material* ms[2] = {&pl->material1, &pl->material2};
                    const char* dk[2] = {"diffuse1", "diffuse2"};
                    const char* sk[2] = {"specular1", "specular2"};
                    for (int k = 0; k < 2; k++) {
                        ms[k]->diffuse = color_of(o, dk[k]); ms[k]->specular = color_of(o, sk[k]);
                        ms[k]->gloss = gloss(ms[k]->diffuse, ms[k]->specular);
                        ms[k]->specular_exp = num(o, "specular_exp");
                    }
                }
            } else {                                                        // face, readShape.cpp:91-119
                vec3(o, "normal", n);
                const JVal& vs = key(o, "verts");
                if (vs.kind != JVal::ARR || vs.arr.empty()) throw SceneError("face needs 'verts'");
                std::vector<Point> pts;
                for (auto& v : vs.arr) {
                    if (v->kind != JVal::ARR || v->arr.size() < 3) throw SceneError("a vertex must be an array of 3 numbers");
                    for (int k = 0; k < 3; k++) if (v->arr[k]->kind != JVal::NUM) throw SceneError("a vertex must hold numbers");
                    pts.push_back(Point(v->arr[0]->num, v->arr[1]->num, v->arr[2]->num));
                }
                Vector nv(n[0], n[1], n[2]);
                nv.normalize();                                             // readShape.cpp:95; Face::Face keeps it as given
                Face* f = new Face(nv, (int)pts.size(), pts.data());
                objs->push_back(f);
                f->is_light = light;
                if (light) {
                    f->color = color_of(o, "color");
                    f->material.ref = f->color; f->material.is_light = true;
                } else {
                    Material& m = f->material;
                    m.diffuse = color_of(o, "diffuse"); m.specular = color_of(o, "specular");
                    m.gloss = gloss(m.diffuse, m.specular);
                    m.specular_exp = num(o, "specular_exp");
                }
            }
        }
    } catch (...) {
        free_scene(objs);
        throw;
    }
    return objs;
}

SceneTables flatten(const std::vector<Solid*>& objs) {
    namespace G = scene_globals;
    SceneTables sc;
    sc.params.assign(RT_PA_COUNT, 0.0);
    sc.params[RT_PA_WIDTH] = G::width; sc.params[RT_PA_HEIGHT] = G::height; sc.params[RT_PA_SPP] = G::paths_per_pixel;
    sc.params[RT_PA_DEPTH] = G::max_ray_depth; sc.params[RT_PA_THREADS] = G::num_threads; sc.params[RT_PA_BLURS] = G::num_blurs;
    sc.params[RT_PA_SCALE] = G::scale; sc.params[RT_PA_LENS] = G::lens_radius; sc.params[RT_PA_FOCAL] = G::focal_length;
    sc.params[RT_PA_EXPOSURE] = G::exposure;
    sc.params[RT_PA_CAM_X] = G::camera.x; sc.params[RT_PA_CAM_Y] = G::camera.y; sc.params[RT_PA_CAM_Z] = G::camera.z;
    auto push_mat = [&](const Material& m) {
        const double r[RT_MT_STRIDE] = {m.ref.r, m.ref.g, m.ref.b, m.is_light ? 1.0 : 0.0, m.gloss,
                                        m.diffuse.r, m.diffuse.g, m.diffuse.b, m.specular.r, m.specular.g, m.specular.b, m.specular_exp,
                                        m.refraction_diffuse.r, m.refraction_diffuse.g, m.refraction_diffuse.b,
                                        m.refraction_specular.r, m.refraction_specular.g, m.refraction_specular.b,
                                        m.refraction_specular_exp, m.refraction_index};
        sc.mats.insert(sc.mats.end(), r, r + RT_MT_STRIDE);
        return (double)(sc.n_mats() - 1);
    };
    for (const Solid* s : objs) {
        double p[RT_PR_STRIDE] = {0};
        if (const Sphere* sp = dynamic_cast<const Sphere*>(s)) {
            p[RT_PR_KIND] = RT_KIND_SPHERE; p[RT_PR_LIGHT] = sp->is_light ? 1.0 : 0.0;
            p[RT_PR_A] = sp->center.x; p[RT_PR_A + 1] = sp->center.y; p[RT_PR_A + 2] = sp->center.z;
            p[RT_PR_RADIUS] = sp->radius;
            p[RT_PR_MAT0] = p[RT_PR_MAT1] = push_mat(sp->material);
        } else if (const Face* f = dynamic_cast<const Face*>(s)) {          // before Plane: a Face is a Plane
            p[RT_PR_KIND] = RT_KIND_FACE; p[RT_PR_LIGHT] = f->is_light ? 1.0 : 0.0;
            p[RT_PR_VOFF] = sc.n_verts(); p[RT_PR_VCNT] = f->num;
            for (const Point& v : f->verts) { sc.verts.push_back(v.x); sc.verts.push_back(v.y); sc.verts.push_back(v.z); }
            p[RT_PR_A] = f->point.x; p[RT_PR_A + 1] = f->point.y; p[RT_PR_A + 2] = f->point.z;
            for (int i = 0; i < 3; i++) p[RT_PR_N + i] = f->norm[i];
            p[RT_PR_MAT0] = p[RT_PR_MAT1] = push_mat(f->material);
        } else if (const Plane* pl = dynamic_cast<const Plane*>(s)) {
            p[RT_PR_KIND] = RT_KIND_PLANE;                                  // RT_PR_LIGHT stays 0
            p[RT_PR_A] = pl->point.x; p[RT_PR_A + 1] = pl->point.y; p[RT_PR_A + 2] = pl->point.z;
            for (int i = 0; i < 3; i++) p[RT_PR_N + i] = pl->norm[i];
            p[RT_PR_MAT0] = push_mat(pl->material1);
            p[RT_PR_MAT1] = push_mat(pl->material2);
        }
        sc.prims.insert(sc.prims.end(), p, p + RT_PR_STRIDE);
    }
    return sc;
}

SceneTables load_scene_tables(const std::string& filename) {
    std::vector<Solid*>* objs = ::read_json_scene(filename);
    SceneTables sc = flatten(*objs);
    free_scene(objs);
    return sc;
}

}  // namespace rt

// Vector::normalize, src/Vector.cpp:50-56 with Vector::dot :33-39 (sum from 0, left to right)
Vector& Vector::normalize() {
    double t = 0;
    t += x * x; t += y * y; t += z * z;
    const double l = std::sqrt(t);
    x /= l; y /= l; z /= l;
    return *this;
}

std::vector<Solid*>* read_json_scene(std::string filename) {
    std::ifstream f(filename);
    if (!f) throw rt::SceneError("cannot open " + filename);
    std::stringstream ss;
    ss << f.rdbuf();
    return rt::parse_scene_text(ss.str());
}
