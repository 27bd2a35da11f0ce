// Host side of the renderer: what stays on the CPU around the render call.
//
// The reference's main() (/root/reference/src/main.cpp:291-333) is
//     objs = *read_json_scene("scene.json");  render on threads;  expose(img, colors);
//     writeImage("test.png", width, height, img);
// This header keeps those three host functions under their reference names -- read_json_scene
// (src/readShape.cpp:26; returns the reference's vector<Solid*>, types in rt_solids.hpp), expose
// (src/main.cpp:235) and the extern "C" writeImage (src/rayTracer.hpp:9-11, src/pngUtil.c:4); rt::flatten
// turns the object list into the flat tables the C ABI of include/rt_hip.h takes.
#pragma once
#include <cstdint>
#include <stdexcept>
#include <string>
#include <vector>

#include "rt_solids.hpp"

namespace rt {

// The state the reference's loader leaves in globals and in vector<Solid*>, as the flat tables of
// include/rt_hip.h (RT_PA_*, RT_PR_*, RT_MT_* layouts).
struct SceneTables {
    std::vector<double> params;   // RT_PA_COUNT
    std::vector<double> prims;    // n_prims * RT_PR_STRIDE
    std::vector<double> mats;     // n_mats * RT_MT_STRIDE
    std::vector<double> verts;    // n_verts * 3
    int n_prims() const;
    int n_mats() const;
    int n_verts() const;
    int width() const;
    int height() const;
    double exposure() const;
    int num_blurs() const;
};

struct SceneError : std::runtime_error {
    using std::runtime_error::runtime_error;
};

// Parses a scene stream held in memory -- 1 + num_objs concatenated JSON objects (src/readShape.cpp:26-122) -- into
// solids in file order and rt::scene_globals; ::read_json_scene is this on a file's contents.
std::vector<Solid*>* parse_scene_text(const std::string& text);
void free_scene(std::vector<Solid*>* objs);

// The object list and rt::scene_globals as the tables of rt_upload_scene: one record per solid in file order, one per
// Material in the order the loader made them, the face vertices.
SceneTables flatten(const std::vector<Solid*>& objs);

// read_json_scene + flatten + free_scene
SceneTables load_scene_tables(const std::string& filename);

// expose() + blur() + Color::to_int (src/main.cpp:186-242, src/geometry.cpp:15-23) in double:
// colors is width*height*3 (r,g,b), modified in place (bloom added); img receives 0xRRGGBB.
void expose(int* img, double* colors, int width, int height, double exposure, int num_blurs, int threads);

}  // namespace rt

// 8-bit RGB PNG from packed 0xRRGGBB ints; same signature and pixel order as the reference
// (src/pngUtil.c:4-29).  Returns 0, or -1 if the file cannot be written (the reference never checks).
extern "C" int writeImage(char* filename, int width, int height, int* image);
