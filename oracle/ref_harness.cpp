// TEST INFRASTRUCTURE ONLY -- never linked into the product library.
//
// Driver for the UNMODIFIED reference sources, compiled where they lie under
// /root/reference/src (see oracle/Makefile, target _ref/ref_harness).  The
// reference's main.cpp is compiled with -Dmain=reference_main so that this file
// can supply its own main() and repeat the orchestration of
// /root/reference/src/main.cpp:291-333 around the reference's own
// read_json_scene / do_rays_i / expose, with three additions the reference lacks:
//   * the scene path comes from argv instead of the hard-coded "scene.json";
//   * num_threads / num_blurs may be overridden after loading;
//   * the un-exposed Color array, the packed image and the per-thread ray
//     counters are dumped to a binary file instead of a PNG.
//
// usage: ref_harness <scene.json> <out.bin> [num_threads|-1] [num_blurs|-1]
//
// out.bin layout (little endian):
//   int32 width, height, num_threads, num_blurs
//   int64 rays[num_threads]
//   double colors[width*height*3]   (r,g,b per pixel, before expose())
//   int32  img[width*height]        (0xRRGGBB after expose())
//   double seconds                  (render + expose, as main.cpp:305-318 times it)
#include <iostream>
#include <thread>
#include <chrono>
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

// globals defined in /root/reference/src/main.cpp:16-32
extern vector<Solid*> objs;
extern int num_objs;
extern default_random_engine* generators;
extern long* num_rays;
extern double exposure;
extern int num_blurs;
extern int paths_per_pixel;
extern int width;
extern int height;
extern double scale;
extern double lens_radius;
extern double focal_length;
extern int max_ray_depth;
extern int num_threads;

int main(int argc, char** argv) {
    if (argc < 3) {
        fprintf(stderr, "usage: %s scene.json out.bin [num_threads] [num_blurs]\n", argv[0]);
        return 2;
    }
    objs = *read_json_scene(argv[1]);
    if (argc > 3 && atoi(argv[3]) > 0) num_threads = atoi(argv[3]);
    if (argc > 4 && atoi(argv[4]) >= 0) num_blurs = atoi(argv[4]);

    num_rays = new long[num_threads]();
    generators = new default_random_engine[num_threads];
    int* img = new int[width * height];
    Color* colors = new Color[width * height];
    vector<thread> threads;
    auto start = chrono::system_clock::now();
    for (int i = 1; i < num_threads; i++) {
        generators[i].seed(0);
        threads.emplace_back(do_rays_i, colors, i);
    }
    do_rays_i(colors, 0);
    for (auto& t : threads) t.join();
    auto mid = chrono::system_clock::now();

    vector<double> raw((size_t)width * height * 3);
    for (int i = 0; i < width * height; i++) {
        raw[3 * i + 0] = colors[i].r;
        raw[3 * i + 1] = colors[i].g;
        raw[3 * i + 2] = colors[i].b;
    }
    auto mid2 = chrono::system_clock::now();
    expose(img, colors);
    auto end = chrono::system_clock::now();
    chrono::duration<double> e1 = mid - start, e2 = end - mid2;
    double seconds = e1.count() + e2.count();

    FILE* f = fopen(argv[2], "wb");
    if (!f) { perror("out.bin"); return 1; }
    int32_t hdr[4] = {width, height, num_threads, num_blurs};
    fwrite(hdr, sizeof(hdr), 1, f);
    for (int i = 0; i < num_threads; i++) {
        int64_t r = num_rays[i];
        fwrite(&r, 8, 1, f);
    }
    fwrite(raw.data(), 8, raw.size(), f);
    fwrite(img, 4, (size_t)width * height, f);
    fwrite(&seconds, 8, 1, f);
    fclose(f);
    long rays = 0;
    for (int i = 0; i < num_threads; i++) rays += num_rays[i];
    fprintf(stderr, "\nref_harness: %dx%d T=%d blurs=%d rays=%ld render_s=%.3f total_s=%.3f\n",
            width, height, num_threads, num_blurs, rays, e1.count(), seconds);
    return 0;
}
