// rt_main: the reference's main() (/root/reference/src/main.cpp:291-333) with the thread fan-out around
// do_rays_i replaced by the HIP library.  Same sequence -- load scene, echo parameters, render, expose,
// print "Time:" and "Rays:", write the PNG -- and the same defaults (./scene.json -> ./test.png); the
// paths and a few switches come from argv, which the reference does not read.
//
//   rt_main [scene.json] [out.png] [--device N] [--gpus N] [--tile-rows R] [--reference-order] [--host-expose]
//           [--dump-tables file] [--no-render]
//
// --reference-order renders with one GPU lane per reference thread in the reference's RNG order
// (validation mode: the 8-bit image then equals the reference binary's).  expose() runs on the device and only the
// packed image comes back; --host-expose copies the radiance array back and runs the C++ expose() instead.
// --gpus N (N > 1): the frame is split into cyclic row tiles over HIP devices 0..N-1, every device renders, exposes and
// packs its rows, and the rows are gathered to device 0 with RCCL before the PNG is written (rt_multi_*).
#include <chrono>
#include <cstdio>
#include <cstring>
#include <iostream>
#include <thread>

#include "../../../include/rt_hip.h"
#include "rt_host.hpp"

int main(int argc, char** argv) {
    std::string scene_path = "scene.json", out_path = "test.png", dump_path;
    int device = 0, gpus = 1, tile_rows = 0;
    bool ref_order = false, no_render = false, host_expose = false;
    int pos = 0;
    for (int i = 1; i < argc; i++) {
        std::string a = argv[i];
        if (a == "--device" && i + 1 < argc) device = atoi(argv[++i]);
        else if (a == "--gpus" && i + 1 < argc) gpus = atoi(argv[++i]);
        else if (a == "--tile-rows" && i + 1 < argc) tile_rows = atoi(argv[++i]);
        else if (a == "--reference-order") ref_order = true;
        else if (a == "--no-render") no_render = true;
        else if (a == "--host-expose") host_expose = true;
        else if (a == "--dump-tables" && i + 1 < argc) dump_path = argv[++i];
        else if (pos == 0) { scene_path = a; pos++; }
        else if (pos == 1) { out_path = a; pos++; }
        else { fprintf(stderr, "rt_main: unexpected argument %s\n", a.c_str()); return 2; }
    }
    // objs = *read_json_scene("scene.json") (main.cpp:292), then the object list flattened into the C ABI's tables
    rt::SceneTables sc;
    try {
        std::vector<Solid*>* objs = read_json_scene(scene_path);
        sc = rt::flatten(*objs);
        rt::free_scene(objs);
    } catch (const rt::SceneError& e) {
        fprintf(stderr, "rt_main: %s\n", e.what());
        return 1;
    }
    if (!dump_path.empty()) {   // the flat tables, for the loader's parity test
        FILE* f = fopen(dump_path.c_str(), "wb");
        if (!f) { perror("dump"); return 1; }
        int32_t hdr[4] = {(int32_t)sc.params.size(), sc.n_prims(), sc.n_mats(), sc.n_verts()};
        fwrite(hdr, sizeof(hdr), 1, f);
        fwrite(sc.params.data(), 8, sc.params.size(), f);
        fwrite(sc.prims.data(), 8, sc.prims.size(), f);
        fwrite(sc.mats.data(), 8, sc.mats.size(), f);
        fwrite(sc.verts.data(), 8, sc.verts.size(), f);
        fclose(f);
    }
    if (no_render) return 0;
    const int W = sc.width(), H = sc.height();
    std::cout << "Scale:           " << sc.params[RT_PA_SCALE] << std::endl;
    std::cout << "Width:           " << W << std::endl;
    std::cout << "Height:          " << H << std::endl;
    std::cout << "Paths per pixel: " << (int)sc.params[RT_PA_SPP] << std::endl;
    std::cout << "Threads:         " << (int)sc.params[RT_PA_THREADS] << std::endl;
    std::cout << "Max ray depth:   " << (int)sc.params[RT_PA_DEPTH] << std::endl;

    if (gpus > 1 || getenv("RT_MULTI_LOOPBACK")) {
        if (ref_order || host_expose) { fprintf(stderr, "rt_main: --gpus cannot be combined with --reference-order or --host-expose\n"); return 2; }
        rt_multi_handle m = nullptr;
        int rc = rt_multi_create(&m, gpus);
        if (rc != RT_OK) { fprintf(stderr, "rt_main: rt_multi_create(%d) failed (%d): %s\n", gpus, rc, rt_multi_last_error(nullptr)); return 1; }
        rc = rt_multi_upload_scene(m, sc.params.data(), sc.n_prims(), sc.prims.data(), sc.n_mats(), sc.mats.data(), sc.n_verts(), sc.verts.data());
        if (rc != RT_OK) { fprintf(stderr, "rt_main: %s\n", rt_multi_last_error(m)); rt_multi_destroy(m); return 1; }
        std::vector<int> img((size_t)W * H);
        uint64_t rays = 0;
        auto start = std::chrono::system_clock::now();
        rc = rt_multi_render_image(m, tile_rows, img.data(), &rays);
        if (rc != RT_OK) { fprintf(stderr, "rt_main: %s\n", rt_multi_last_error(m)); rt_multi_destroy(m); return 1; }
        std::chrono::duration<double> elapsed = std::chrono::system_clock::now() - start;
        double sec = elapsed.count();
        int min = (int)(sec / 60);
        sec -= min * 60;
        std::cout << "GPUs: " << rt_multi_gpus(m) << std::endl;
        std::cout << "Time: " << min << "m" << sec << "s" << std::endl;
        std::cout << "Rays: " << (double)rays << std::endl;
        rt_multi_destroy(m);
        if (writeImage((char*)out_path.c_str(), W, H, img.data()) != 0) { fprintf(stderr, "rt_main: cannot write %s\n", out_path.c_str()); return 1; }
        return 0;
    }
    rt_handle h = nullptr;
    int rc = rt_create(&h, device);
    if (rc != RT_OK) { fprintf(stderr, "rt_main: rt_create failed (%d): no usable GPU\n", rc); return 1; }
    rc = rt_upload_scene(h, sc.params.data(), sc.n_prims(), sc.prims.data(), sc.n_mats(), sc.mats.data(), sc.n_verts(), sc.verts.data());
    if (rc != RT_OK) { fprintf(stderr, "rt_main: %s\n", rt_last_error(h)); rt_destroy(h); return 1; }
    std::vector<int> img((size_t)W * H);
    uint64_t rays = 0;
    auto start = std::chrono::system_clock::now();
    if (host_expose) {
        std::vector<double> colors((size_t)W * H * 3);
        rc = ref_order ? rt_render_streams(h, colors.data(), &rays, nullptr) : rt_render(h, colors.data(), &rays, nullptr);
        if (rc != RT_OK) { fprintf(stderr, "rt_main: %s\n", rt_last_error(h)); rt_destroy(h); return 1; }
        unsigned hw = std::thread::hardware_concurrency();
        rt::expose(img.data(), colors.data(), W, H, sc.exposure(), sc.num_blurs(), hw ? (int)hw : 1);
    } else {
        rc = rt_render_image(h, ref_order ? 1 : 0, img.data(), &rays);
        if (rc != RT_OK) { fprintf(stderr, "rt_main: %s\n", rt_last_error(h)); rt_destroy(h); return 1; }
    }
    auto end = std::chrono::system_clock::now();
    std::chrono::duration<double> elapsed = end - start;
    double sec = elapsed.count();
    int min = (int)(sec / 60);
    sec -= min * 60;
    std::cout << "Time: " << min << "m" << sec << "s" << std::endl;
    std::cout << "Rays: " << (double)rays << std::endl;
    rt_timing tm;
    if (rt_last_timing(h, &tm) == RT_OK) std::cout << "GPU kernels: " << tm.kernel_ms << " ms" << std::endl;
    rt_destroy(h);
    if (writeImage((char*)out_path.c_str(), W, H, img.data()) != 0) { fprintf(stderr, "rt_main: cannot write %s\n", out_path.c_str()); return 1; }
    return 0;
}
