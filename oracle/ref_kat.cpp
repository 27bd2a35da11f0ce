// TEST INFRASTRUCTURE ONLY -- never linked into the product library.
//
// Function-level known answers from the UNMODIFIED reference sources (compiled where they lie under
// /root/reference/src, oracle/Makefile target _ref/ref_kat): for seeded rays through a scene file it calls
//   Sphere/Plane/Face::intersect            (/root/reference/src/geometry.cpp:99-105,129-138,160-168)
//   get_intersection                        (/root/reference/src/main.cpp:34-43; normal and material of the hit)
//   get_reflection_ray, get_refraction_ray  (main.cpp:47-104,107-122, with Line::reflect/refract, geometry.cpp:78-92)
//   add_to_path + Path::trace               (main.cpp:126-183, geometry.cpp:186-217)
//   Color::to_int                           (geometry.cpp:15-23)
// and writes inputs and outputs as raw doubles.  tests/golden/make_golden.py turns the file into a fixture, the
// oracle's oracle_kat() (oracle_capi.cpp) must reproduce every word.
//
// usage: ref_kat <scene.json> <n_rays> <seed> <out.bin>
// out.bin: int32 n_rays, n_objs, record doubles; then n_rays records of
//   o[3] d[3] state                                   inputs (d as Line's constructor normalises it; state seeds the RNG)
//   p_i[3] for every object i                         Solid::intersect
//   p[3] normal[3] gloss rindex sexp diffuse_r        get_intersection
//   rp[3] rd[3] state_after                           get_reflection_ray from `state` (zeros when nothing was hit)
//   ed[3]                                             get_refraction_ray entering from index 1
//   ld[3]                                             get_refraction_ray with the current index equal to the material's (leaving)
//   c[3] rays state_after                             add_to_path(max_ray_depth) + Path::trace from `state`
// then int32 n_colors and n_colors x (r g b max packed-as-double).
#include <iostream>
#include <sstream>
#include <random>
#include <vector>
#include <string>
#include <cstdio>
#include <cstdlib>
#include <cstdint>

using namespace std;

#include "Vector.hpp"
#include "geometry.hpp"
#include "rayTracer.hpp"

extern vector<Solid*> objs;
extern int num_objs;
extern default_random_engine* generators;
extern long* num_rays;
extern int max_ray_depth;
extern int num_threads;

Line get_reflection_ray(const Line& ray, const Point& p, const Vector& normal, const Material& mat, int thread_num);
Line get_refraction_ray(const Line& ray, const Point& p, const Vector& normal, const Material& mat, double refraction_index, int thread_num);
void get_intersection(const Line& ray, Material* mat, Vector* normal, Point* p);
void add_to_path(const Line& ray, int remaining, double refraction_index, int thread_num, Path* path);

static uint64_t lcg = 1;
static double urand() {      // own generator for the inputs (not the reference's)
    lcg = lcg * 6364136223846793005ULL + 1442695040888963407ULL;
    return (double)(lcg >> 11) / 9007199254740992.0;
}
static double rng_state() {
    ostringstream os;
    os << generators[0];
    return atof(os.str().c_str());
}

int main(int argc, char** argv) {
    if (argc < 5) { fprintf(stderr, "usage: %s scene.json n_rays seed out.bin\n", argv[0]); return 2; }
    objs = *read_json_scene(argv[1]);
    const int n = atoi(argv[2]);
    lcg = (uint64_t)atoll(argv[3]) * 2654435761ULL + 12345;
    num_threads = 1;
    num_rays = new long[1]();
    generators = new default_random_engine[1];
    FILE* f = fopen(argv[4], "wb");
    if (!f) { perror("out.bin"); return 1; }
    int32_t hdr[2] = {n, num_objs};
    fwrite(hdr, sizeof(hdr), 1, f);
    vector<double> rec;
    for (int r = 0; r < n; r++) {
        rec.clear();
        // origins in and around the room of the scenes (x, y in [-25,25], z in [-45,45]), aimed at a point inside it
        Point o(-25 + 50 * urand(), -25 + 50 * urand(), -45 + 90 * urand());
        Point t(-20 + 40 * urand(), -20 + 40 * urand(), -40 + 40 * urand());
        Line ray(o, t);
        const double state = 1 + floor(urand() * 2147483645.0);
        rec.push_back(ray.point.x); rec.push_back(ray.point.y); rec.push_back(ray.point.z);
        for (int k = 0; k < 3; k++) rec.push_back(ray.direction[k]);
        rec.push_back(state);
        for (int i = 0; i < num_objs; i++) {
            Point p = objs[i]->intersect(ray);
            rec.push_back(p.x); rec.push_back(p.y); rec.push_back(p.z);
        }
        Material mat = {};
        Vector normal;
        for (int k = 0; k < 3; k++) normal[k] = 0;
        Point p(INFINITY, INFINITY, INFINITY);
        get_intersection(ray, &mat, &normal, &p);
        rec.push_back(p.x); rec.push_back(p.y); rec.push_back(p.z);
        for (int k = 0; k < 3; k++) rec.push_back(normal[k]);
        rec.push_back(mat.gloss); rec.push_back(mat.refraction_index); rec.push_back(mat.specular_exp); rec.push_back(mat.diffuse.r);
        // a glossy lobe about a mirror direction that points into the surface never passes the reference's rejection
        // loop (main.cpp:61-76): sample only surfaces met from their front
        const bool front = p.is_valid() && normal.dot(ray.direction) < 0;
        if (front && !mat.is_light) {
            generators[0].seed((unsigned long)state);
            Line rr = get_reflection_ray(ray, p, normal, mat, 0);
            rec.push_back(rr.point.x); rec.push_back(rr.point.y); rec.push_back(rr.point.z);
            for (int k = 0; k < 3; k++) rec.push_back(rr.direction[k]);
            rec.push_back(rng_state());
            Line re = get_refraction_ray(ray, p, normal, mat, 1.0, 0);
            for (int k = 0; k < 3; k++) rec.push_back(re.direction[k]);
            Line rl = get_refraction_ray(ray, p, normal, mat, mat.refraction_index, 0);
            for (int k = 0; k < 3; k++) rec.push_back(rl.direction[k]);
        } else {
            for (int k = 0; k < 13; k++) rec.push_back(0.0);
        }
        if (front || !p.is_valid()) {
            generators[0].seed((unsigned long)state);
            num_rays[0] = 0;
            Path path(ray.point);
            add_to_path(ray, max_ray_depth, 1, 0, &path);
            Color c = path.trace();
            rec.push_back(c.r); rec.push_back(c.g); rec.push_back(c.b);
            rec.push_back((double)num_rays[0]);
            rec.push_back(rng_state());
        } else {
            for (int k = 0; k < 5; k++) rec.push_back(0.0);
        }
        fwrite(rec.data(), 8, rec.size(), f);
    }
    const int32_t nc = 512;
    fwrite(&nc, 4, 1, f);
    for (int i = 0; i < nc; i++) {
        const double mx = i % 3 == 0 ? 1.0 : (i % 3 == 1 ? 1.0 / 0.35 : 1.0 / 2.5);
        Color c(urand() * 3 * mx, i % 7 == 0 ? 0.0 : urand() * mx, urand() * 1.01 * mx);
        const double v[5] = {c.r, c.g, c.b, mx, (double)c.to_int(mx)};
        fwrite(v, 8, 5, f);
    }
    fclose(f);
    return 0;
}
