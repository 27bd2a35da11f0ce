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
// Vector::normalize, src/Vector.cpp:50-56 with Vector::dot :33-39
void normalize(double n[3]) {
    double t = 0;
    for (int i = 0; i < 3; i++) t += n[i] * n[i];
    double l = std::sqrt(t);
    for (int i = 0; i < 3; i++) n[i] /= l;
}
double gloss(const double d[3], const double s[3]) {   // readShape.cpp:62-64
    double diff = d[0] + d[1] + d[2];
    double spec = s[0] + s[1] + s[2];
    return spec / (diff + spec);
}
void put3(std::vector<double>& rec, int at, const double v[3]) { for (int i = 0; i < 3; i++) rec[at + i] = v[i]; }

}  // namespace

int SceneTables::n_prims() const { return (int)(prims.size() / RT_PR_STRIDE); }
int SceneTables::n_mats() const { return (int)(mats.size() / RT_MT_STRIDE); }
int SceneTables::n_verts() const { return (int)(verts.size() / 3); }
int SceneTables::width() const { return (int)params[RT_PA_WIDTH]; }
int SceneTables::height() const { return (int)params[RT_PA_HEIGHT]; }
double SceneTables::exposure() const { return params[RT_PA_EXPOSURE]; }
int SceneTables::num_blurs() const { return (int)params[RT_PA_BLURS]; }

SceneTables parse_scene_text(const std::string& text) {
    Parser p(text);
    if (p.at_end()) throw SceneError("empty scene stream");
    JPtr head = p.value();
    SceneTables sc;
    sc.params.assign(RT_PA_COUNT, 0.0);
    sc.params[RT_PA_EXPOSURE] = num(*head, "exposure");
    sc.params[RT_PA_BLURS] = num(*head, "num_blurs");
    const int num_objs = (int)num(*head, "num_objs");
    sc.params[RT_PA_SPP] = num(*head, "paths_per_pixel");
    sc.params[RT_PA_WIDTH] = num(*head, "width");
    sc.params[RT_PA_HEIGHT] = num(*head, "height");
    sc.params[RT_PA_SCALE] = num(*head, "scale");
    sc.params[RT_PA_LENS] = num(*head, "lens_radius");
    sc.params[RT_PA_FOCAL] = num(*head, "focal_length");
    sc.params[RT_PA_DEPTH] = num(*head, "max_ray_depth");
    sc.params[RT_PA_THREADS] = num(*head, "num_threads");
    // optional extension: "camera": [x, y, z]; the reference's constant (main.cpp:16) otherwise
    sc.params[RT_PA_CAM_X] = 0.0; sc.params[RT_PA_CAM_Y] = 0.0; sc.params[RT_PA_CAM_Z] = 40.0;
    if (head->obj.find("camera") != head->obj.end()) {
        double cam[3];
        vec3(*head, "camera", cam);
        for (int k = 0; k < 3; k++) sc.params[RT_PA_CAM_X + k] = cam[k];
    }
    for (int i = 0; i < num_objs; i++) {
        if (p.at_end()) throw SceneError("header announces " + std::to_string(num_objs) + " objects, stream holds " + std::to_string(i));
        JPtr op = p.value();
        const JVal& o = *op;
        const JVal& shape = key(o, "shape");
        if (shape.kind != JVal::STR) throw SceneError("'shape' must be a string");
        const bool light = truth(o, "is_light");
        std::vector<double> pr(RT_PR_STRIDE, 0.0);
        auto material = [&](const double* ref, bool is_light, double g, const double* d, const double* s, double sexp,
                            const double* rd, const double* rs, double rsexp, double rindex) {
            std::vector<double> m(RT_MT_STRIDE, 0.0);
            const double z[3] = {0, 0, 0};
            put3(m, RT_MT_REF, ref ? ref : z); m[RT_MT_LIGHT] = is_light ? 1.0 : 0.0; m[RT_MT_GLOSS] = g;
            put3(m, RT_MT_DIFF, d ? d : z); put3(m, RT_MT_SPEC, s ? s : z); m[RT_MT_SEXP] = sexp;
            put3(m, RT_MT_RDIFF, rd ? rd : z); put3(m, RT_MT_RSPEC, rs ? rs : z); m[RT_MT_RSEXP] = rsexp; m[RT_MT_RINDEX] = rindex;
            const int idx = sc.n_mats();
            sc.mats.insert(sc.mats.end(), m.begin(), m.end());
            return idx;
        };
        double a[3], n[3], d[3], s[3], rd[3], rs[3], c[3];
        if (shape.str == "sphere") {                                    // readShape.cpp:47-67
            pr[RT_PR_KIND] = RT_KIND_SPHERE;
            vec3(o, "center", a); put3(pr, RT_PR_A, a);
            pr[RT_PR_RADIUS] = num(o, "radius");
            pr[RT_PR_LIGHT] = light ? 1.0 : 0.0;
            int m;
            if (light) { vec3(o, "color", c); m = material(c, true, 0, nullptr, nullptr, 0, nullptr, nullptr, 0, 0); }
            else {
                vec3(o, "diffuse", d); vec3(o, "specular", s); vec3(o, "refraction_diffuse", rd); vec3(o, "refraction_specular", rs);
                m = material(nullptr, false, gloss(d, s), d, s, num(o, "specular_exp"), rd, rs, num(o, "refraction_specular_exp"), num(o, "refraction_index"));
            }
            pr[RT_PR_MAT0] = pr[RT_PR_MAT1] = m;
        } else if (shape.str == "plane") {                              // readShape.cpp:68-90
            pr[RT_PR_KIND] = RT_KIND_PLANE;
            vec3(o, "point", a); put3(pr, RT_PR_A, a);
            vec3(o, "normal", n); normalize(n); put3(pr, RT_PR_N, n);
            pr[RT_PR_LIGHT] = 0.0;   // Solid::is_light is never written for an emitting plane (readShape.cpp:75)
            if (light) {
                pr[RT_PR_MAT0] = material(nullptr, true, 0, nullptr, nullptr, 0, nullptr, nullptr, 0, 0);
                pr[RT_PR_MAT1] = material(nullptr, true, 0, nullptr, nullptr, 0, nullptr, nullptr, 0, 0);
            } else {
                vec3(o, "diffuse1", d); vec3(o, "specular1", s);
                pr[RT_PR_MAT0] = material(nullptr, false, gloss(d, s), d, s, num(o, "specular_exp"), nullptr, nullptr, 0, 0);
                vec3(o, "diffuse2", d); vec3(o, "specular2", s);
                pr[RT_PR_MAT1] = material(nullptr, false, gloss(d, s), d, s, num(o, "specular_exp"), nullptr, nullptr, 0, 0);
            }
        } else {                                                        // face, readShape.cpp:91-119
            pr[RT_PR_KIND] = RT_KIND_FACE;
            vec3(o, "normal", n); normalize(n); put3(pr, RT_PR_N, n);
            const JVal& vs = key(o, "verts");
            if (vs.kind != JVal::ARR || vs.arr.empty()) throw SceneError("face needs 'verts'");
            pr[RT_PR_VOFF] = sc.n_verts();
            pr[RT_PR_VCNT] = (double)vs.arr.size();
            for (auto& v : vs.arr) {
                if (v->kind != JVal::ARR || v->arr.size() < 3) throw SceneError("a vertex must be an array of 3 numbers");
                for (int k = 0; k < 3; k++) {
                    if (v->arr[k]->kind != JVal::NUM) throw SceneError("a vertex must hold numbers");
                    sc.verts.push_back(v->arr[k]->num);
                }
            }
            for (int k = 0; k < 3; k++) pr[RT_PR_A + k] = sc.verts[(size_t)pr[RT_PR_VOFF] * 3 + k];
            pr[RT_PR_LIGHT] = light ? 1.0 : 0.0;
            int m;
            if (light) { vec3(o, "color", c); m = material(c, true, 0, nullptr, nullptr, 0, nullptr, nullptr, 0, 0); }
            else { vec3(o, "diffuse", d); vec3(o, "specular", s); m = material(nullptr, false, gloss(d, s), d, s, num(o, "specular_exp"), nullptr, nullptr, 0, 0); }
            pr[RT_PR_MAT0] = pr[RT_PR_MAT1] = m;
        }
        sc.prims.insert(sc.prims.end(), pr.begin(), pr.end());
    }
    return sc;
}

SceneTables read_json_scene(const std::string& filename) {
    std::ifstream f(filename);
    if (!f) throw SceneError("cannot open " + filename);
    std::stringstream ss;
    ss << f.rdbuf();
    return parse_scene_text(ss.str());
}

}  // namespace rt
