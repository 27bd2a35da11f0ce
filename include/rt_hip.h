/* rt_hip.h -- C ABI of librt_hip.so, the MI355X (gfx950) implementation of the
 * per-pixel render loop of samkottler/rayTracer.
 *
 * The reference has no plugin or FFI interface; its render loop is the function
 * do_rays_i(Color* img, int num) that main() runs on num_threads std::threads
 * (/root/reference/src/main.cpp:247-289, spawned at :304-313), reading the scene
 * from globals that read_json_scene() filled (/root/reference/src/readShape.cpp:
 * 13-24,31-42) and writing Color colors[width*height] plus long num_rays[T].
 * This library replaces exactly that seam: the host loads a scene, hands it over
 * once as flat tables, and one rt_render* call fills the radiance array and the
 * ray counter.  Everything else of main() (expose/blur, writeImage) stays on the
 * host side of the boundary.
 *
 * Conventions: plain C types, caller-owned buffers, every function returns
 * RT_OK (0) or a negative rt_status; no exception crosses the boundary.  A handle
 * is bound to one HIP device and is not re-entrant (one caller at a time).
 */
#ifndef RT_HIP_H
#define RT_HIP_H

#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

typedef struct rt_context* rt_handle;

typedef enum {
    RT_OK = 0,
    RT_ERR_ARG = -1,        /* null pointer, bad size, bad enum                        */
    RT_ERR_HIP = -2,        /* a HIP runtime call failed (rt_last_error has the text)  */
    RT_ERR_NO_SCENE = -3,   /* render called before rt_upload_scene                    */
    RT_ERR_SCENE = -4,      /* scene violates a stated limit (depth, vertex count ...) */
    RT_ERR_GUARD = -5,      /* a device loop guard tripped: the scene is outside the   */
                            /* range in which the reference itself terminates          */
    RT_ERR_NO_DEVICE = -6   /* no gfx950 device / kernels not loadable                 */
} rt_status;

/* ---- scene tables: the state read_json_scene leaves behind, as flat doubles ----
 * One record per solid, in file order (order decides ties in the closest-hit scan,
 * /root/reference/src/main.cpp:34-43, and which emitter is "the first light",
 * :138-139).  Normals are already normalised, gloss already derived
 * (readShape.cpp:62-64,73,95). */
enum { RT_KIND_SPHERE = 0, RT_KIND_PLANE = 1, RT_KIND_FACE = 2 };

/* prim record, RT_PR_STRIDE doubles */
enum {
    RT_PR_KIND = 0,    /* RT_KIND_*                                                   */
    RT_PR_LIGHT = 1,   /* Solid::is_light (geometry.hpp:54); always 0 for planes      */
    RT_PR_MAT0 = 2,    /* material index (plane: material1)                           */
    RT_PR_MAT1 = 3,    /* plane: material2, else = MAT0                               */
    RT_PR_VOFF = 4,    /* face: first vertex in the vertex table                      */
    RT_PR_VCNT = 5,    /* face: vertex count                                          */
    RT_PR_A = 6,       /* [3] sphere centre | plane point | face verts[0]             */
    RT_PR_N = 9,       /* [3] plane / face unit normal                                */
    RT_PR_RADIUS = 12,
    RT_PR_STRIDE = 13
};

/* material record, RT_MT_STRIDE doubles: struct Material, geometry.hpp:18-29 */
enum {
    RT_MT_REF = 0,     /* [3] emission                                                */
    RT_MT_LIGHT = 3,
    RT_MT_GLOSS = 4,
    RT_MT_DIFF = 5,    /* [3]                                                         */
    RT_MT_SPEC = 8,    /* [3]                                                         */
    RT_MT_SEXP = 11,
    RT_MT_RDIFF = 12,  /* [3]                                                         */
    RT_MT_RSPEC = 15,  /* [3]                                                         */
    RT_MT_RSEXP = 18,
    RT_MT_RINDEX = 19,
    RT_MT_STRIDE = 20
};

/* parameter vector, RT_PA_COUNT doubles: the header object of the scene stream */
enum {
    RT_PA_WIDTH = 0, RT_PA_HEIGHT, RT_PA_SPP, RT_PA_DEPTH, RT_PA_THREADS, RT_PA_BLURS,
    RT_PA_SCALE, RT_PA_LENS, RT_PA_FOCAL, RT_PA_EXPOSURE,
    RT_PA_SEED,        /* frame seed mixed into every pixel's RNG seed (0 by default) */
    RT_PA_CAM_X, RT_PA_CAM_Y, RT_PA_CAM_Z,   /* camera; the reference's constant is (0,0,40), main.cpp:16 */
    RT_PA_COUNT
};

/* limits */
#define RT_MAX_DEPTH 16          /* max_ray_depth accepted by rt_upload_scene           */
#define RT_MAX_FACE_VERTS 64

typedef struct {
    double upload_ms;      /* last rt_upload_scene: table flattening + H2D              */
    double kernel_ms;      /* last render: device time of all its kernels (HIP events)  */
    double broad_ms;       /* last render: sum over its broad_kernel launches, measured  */
                           /* only while RT_OPT_TIMING_DETAIL is on (else 0)             */
    double total_ms;       /* last rt_render: kernel + D2H of the outputs               */
    uint64_t rays;         /* last render: ray segments traced (reference: num_rays)    */
    uint64_t candidates;   /* last render: (ray, solid) pairs the FP32 broad phase kept */
                           /* for the exact FP64 test                                    */
    uint64_t sphere_tests; /* last render: rays x spheres, etc. (roofline numerators)   */
    uint64_t plane_tests;
    uint64_t face_tests;
    uint64_t face_vert_tests;
    uint64_t overflow_rays; /* pipeline: rays with a long candidate list (more than 6 mask words), tested against every solid by their whole wave */
    uint64_t broad_tiles;   /* matrix-core scan: (32 rays x 32 records) products executed, group records included */
    uint64_t broad_records; /* records of the matrix-core scan's table (a multiple of 32): what every other launch tests each ray against */
    uint64_t accel_used;    /* last render: RT_ACCEL_SCAN or RT_ACCEL_WALK */
    double shade_ms;        /* last render: sum over its shade_kernel launches (exact closest hit + one path segment   */
                            /* per ray), measured like broad_ms only while RT_OPT_TIMING_DETAIL is on (else 0)          */
    uint64_t sub_pipelines; /* last render: parts of the rank's pixels that went through the pipeline side by side on   */
                            /* streams of their own (RT_OPT_OVERLAP above RT_OVERLAP_MIN_PIXELS pixels, else 1)          */
} rt_timing;

/* options for rt_set_option */
enum {
    RT_OPT_MODE = 0,           /* RT_MODE_PIPELINE (default) or RT_MODE_SINGLE_KERNEL           */
    RT_OPT_TIMING_DETAIL = 1,  /* 1: bracket every scan and shade_kernel launch with HIP events */
    RT_OPT_BLOCK = 2,          /* workgroup size of the single-kernel renderer: 256/512/1024    */
    RT_OPT_BROAD_BLOCK = 3,    /* workgroup size of broad_kernel: 256/512/1024                  */
    RT_OPT_WG_PER_CU = 4,      /* persistent workgroups per CU, 0 = occupancy query             */
    RT_OPT_CHUNK_RECORDS = 5,  /* LDS window in 16-byte records (next rt_upload_scene)          */
    RT_OPT_MAX_SLOTS = 6,      /* pixels whose path state is kept at once (memory cap)          */
    RT_OPT_FUSE_NARROW = 7,    /* retired (the narrow phase always runs inside shade_kernel); accepted, ignored */
    RT_OPT_OVERLAP = 8,        /* 1..3 sub-pipelines, each with an equal share of the pixels and a stream of its  */
                               /* own, so that one's kernels fill the tails of the other's (default 2; left alone, */
                               /* the library takes 3 for a grid-walk render of 4.7 M pixels or more)              */
    RT_OPT_ACCEL = 9           /* where a ray's candidate solids come from (RT_ACCEL_*); the closest hit and every   */
                               /* output bit are the same whichever is used                                        */
};
enum { RT_MODE_PIPELINE = 0, RT_MODE_SINGLE_KERNEL = 1 };
enum {
    RT_ACCEL_SCAN = 0,         /* every ray is tested against every solid (through bounding spheres of groups of   */
                               /* solids first), the work get_intersection defines (main.cpp:34-43)                */
    RT_ACCEL_WALK = 1,         /* candidate solids come from a front-to-back walk of a uniform grid: far fewer tests */
    RT_ACCEL_AUTO = 2          /* default: the library picks per scene by a measured threshold (the scan up to 1024  */
                               /* bounded solids, the walk beyond); rt_timing.accel_used says which one ran         */
};

/* Create a context on HIP device `device` (index as HIP enumerates it).
 * Replaces: the allocation block of main(), main.cpp:300-303.
 * Environment variables read here, once per context, for A/B runs and tests (none changes an output bit; the first seven
 * set the same values as rt_set_option): RT_MODE, RT_BLOCK, RT_BROAD_BLOCK, RT_WG_PER_CU, RT_CHUNK_RECORDS, RT_OVERLAP,
 * RT_ACCEL; RT_ACCEL_MIN_BOUNDED, RT_GRID_DENSITY (grid walk); RT_BROAD_MFMA, RT_MFMA_CULL, RT_MFMA_MIN_BOUNDED,
 * RT_MFMA_BINNED, RT_MFMA_BINNED_MIN, RT_MFMA_LIST, RT_MFMA_BEHIND (matrix-core scan); RT_NO_BOX, RT_CAND_LDS; and, at every render,
 * RT_SHADE_GRID; RT_OVERLAP_MIN_PIXELS (pixels of a render from which RT_OPT_OVERLAP applies).  rt_multi_create also reads RT_MULTI_LOOPBACK and RT_MULTI_FORCE_RCCL. */
int rt_create(rt_handle* out, int device);

/* Free device buffers and the context.  Replaces main.cpp:328-331. */
void rt_destroy(rt_handle h);

/* Tuning and instrumentation switches (RT_OPT_*); none of them changes a result bit.
 * No counterpart in the reference. */
int rt_set_option(rt_handle h, int key, int value);

/* Copy the scene into device tables (float structure-of-arrays, file order kept).
 * params: RT_PA_COUNT doubles; prims: n_prims*RT_PR_STRIDE; mats: n_mats*RT_MT_STRIDE;
 * verts: n_verts*3.  Replaces: objs = *read_json_scene(...) and the globals it sets,
 * main.cpp:292, readShape.cpp:31-42. */
int rt_upload_scene(rt_handle h, const double* params, int n_prims, const double* prims, int n_mats,
                    const double* mats, int n_verts, const double* verts);

/* Number of image rows rank `rank` of `world` owns under the cyclic row-tile
 * partition (tile k = rows [k*tile_rows, (k+1)*tile_rows) belongs to rank k % world).
 * The reference's own partition is the special case tile_rows = 1, world = num_threads
 * (main.cpp:250). */
int rt_rows_for_rank(int height, int tile_rows, int rank, int world);

/* Render this rank's rows into DEVICE memory, asynchronously on `stream`
 * (a hipStream_t, or NULL for the default stream).  One RNG stream per pixel, seeded
 * from the pixel's index in the full frame, so the result does not depend on the
 * partition.
 *   d_rgb     : double[rows*width*3], the rank's rows packed in ascending row order;
 *               un-exposed radiance, the reference's Color colors[] (main.cpp:275,303)
 *   d_hit_ids : optional int32[rows*width*spp]: file index of the solid each camera
 *               ray hit, -1 for none (the hit/miss mask); NULL to skip
 * Replaces: the thread fan-out main.cpp:304-313 around do_rays_i. */
int rt_render_rows_device(rt_handle h, int tile_rows, int rank, int world, double* d_rgb, int32_t* d_hit_ids,
                          void* stream);

/* Reference-order validation mode: one GPU lane per reference thread.  Lane t renders
 * rows t, t+T, ... (T = num_threads of the scene header) pixel after pixel from ONE
 * RNG stream started at state 1, exactly the order of main.cpp:250-253,301,307, so the
 * frame can be compared with the reference binary's own output.  Slow by design.
 *   d_rgb: double[height*width*3] (whole frame); d_hit_ids as above, whole frame. */
int rt_render_streams_device(rt_handle h, double* d_rgb, int32_t* d_hit_ids, void* stream);

/* Wait for the last render on this context, check the device guard word and return
 * the number of ray segments traced by it (reference: the sum main.cpp:321-324). */
int rt_finish(rt_handle h, uint64_t* rays);

/* Per-stream ray counters of the last stream-mode render (reference: long num_rays[T],
 * main.cpp:300,127).  Call after rt_finish; n <= num_threads. */
int rt_stream_rays(rt_handle h, uint64_t* out, int n);

/* Convenience: whole frame on this device, outputs copied to HOST buffers.
 *   rgb: double[height*width*3]; rays: out; hit_ids: optional int32[height*width*spp]. */
int rt_render(rt_handle h, double* rgb, uint64_t* rays, int32_t* hit_ids);
int rt_render_streams(rt_handle h, double* rgb, uint64_t* rays, int32_t* hit_ids);

/* expose() on the device: bloom of the over-exposed pixels (num_blurs passes of the 41x41 Gaussian, radius 20,
 * sigma 3, taps outside the frame skipped) added to the radiance, then Color::to_int: int(255*c/(1/exposure))
 * truncated, clamped to 255 from above only, packed r<<16 + g<<8 + b.  Double arithmetic in the reference's
 * operation order: the packed frame equals the reference's bit for bit.  Asynchronous on `stream`.
 *   d_rgb : double[height*width*3] un-exposed radiance (not modified; the reference adds the bloom in place)
 *   d_img : int32[height*width] packed pixels, what writeImage takes (pngUtil.c:4)
 * With num_blurs == 0 the rows are independent, so a rank may expose just the rows it rendered before the
 * gather (height = its row count).  Replaces expose() + blur(), main.cpp:186-242, and Color::to_int,
 * geometry.cpp:15-23. */
int rt_expose_device(rt_handle h, const double* d_rgb, int width, int height, double exposure, int num_blurs,
                     int32_t* d_img, void* stream);

/* rt_expose_device for callers that cannot allocate device memory: code under HIP graph capture, tight serving loops,
 * harnesses that forbid hipMalloc.  It has no counterpart in the reference.  Same arguments, same kernels and the same
 * packed frame bit for bit as rt_expose_device, but the blur passes' two frame-sized buffers and the Gaussian taps live in
 * a fixed array inside the library (12 MiB + 16 KiB on every GPU the library is loaded on, held until the process ends)
 * instead of buffers allocated on first use; the call makes no hipMalloc / hipFree.  Frames of more than
 * RT_NOALLOC_MAX_PIXELS pixels are refused with RT_ERR_ARG (rt_last_error says so).  Every handle on a GPU shares that
 * GPU's array.  Calls with blur passes therefore run one after the other on a GPU: each makes its stream wait for an event
 * the previous one recorded (whatever its stream, handle or thread).  Under graph capture that wait is not possible and is
 * left out: graphs that use this call must not run at the same time on one GPU, or they race on the array.
 * rt_create prepares the array for its device (if that fails, only this call is affected: it returns RT_ERR_HIP). */
#define RT_NOALLOC_MAX_PIXELS 262144      /* 512 x 512; the reference's own scene.json is 400 x 400 */
int rt_expose_device_noalloc(rt_handle h, const double* d_rgb, int width, int height, double exposure, int num_blurs,
                             int32_t* d_img, void* stream);

/* expose() BEFORE the gather, for frames rendered by several ranks with num_blurs > 0 (with num_blurs == 0
 * rt_expose_device on the rank's rows does).  The bloom's input B0 (main.cpp:198-205) is three bits per pixel: is the
 * channel over-exposed.  Step 1, every rank: rt_expose_bits_device reduces its rendered rows to those bits.  Step 2,
 * the caller: all-gather of the bits (1 byte per pixel; rank r's buffer at offset r*pad_rows*width).  Step 3, every
 * rank: rt_expose_rows_device runs the blur passes on the 16-row blocks its own rows depend on (two more block rows of
 * halo either side per pass), adds the bloom to its rows and packs them.  Same arithmetic per pixel as
 * rt_expose_device on the whole frame: the packed rows equal the frame's rows bit for bit.  Asynchronous on `stream`.
 *   d_rgb      : double[rows*width*3], the rank's rows as rt_render_rows_device packs them
 *   d_bits     : uint8[rows*width], bit k set iff channel k of the pixel is over-exposed (rgb[k]*exposure > 1)
 *   d_bits_all : uint8[world*pad_rows*width], every rank's bits; pad_rows >= the largest row count of a rank
 *   d_img      : int32[rows*width], the rank's rows of the packed frame
 * Replaces expose() + blur(), main.cpp:186-242 (which run on the whole frame after the threads have joined). */
int rt_expose_bits_device(rt_handle h, const double* d_rgb, int width, int rows, double exposure, uint8_t* d_bits,
                          void* stream);
int rt_expose_rows_device(rt_handle h, const double* d_rgb, const uint8_t* d_bits_all, int width, int height,
                          int tile_rows, int rank, int world, int pad_rows, double exposure, int num_blurs,
                          int32_t* d_img, void* stream);

/* Convenience: the whole of main()'s timed section on this device -- render every pixel, expose() with the exposure
 * and num_blurs of the uploaded scene header -- with only the packed image copied to the HOST (4 bytes per pixel
 * instead of the 24 of the radiance array).
 *   reference_order : 0 = one RNG stream per pixel; 1 = the reference's order (one lane per reference thread)
 *   img  : int32[height*width], what writeImage takes (pngUtil.c:4)
 *   rays : out, ray segments traced
 * Replaces main.cpp:304-314 (thread fan-out around do_rays_i, join, expose). */
int rt_render_image(rt_handle h, int reference_order, int32_t* img, uint64_t* rays);

/* Timing and work counters of the last upload/render (HIP-event device time). */
int rt_last_timing(rt_handle h, rt_timing* out);

/* Text of the last error on this context (never NULL). */
const char* rt_last_error(rt_handle h);

/* ---- one process, several GPUs ------------------------------------------------------------------------------------------
 * The reference renders one frame on num_threads threads that share the Color array, thread t taking rows t, t+T, ...
 * (main.cpp:250,303-313), then exposes and writes it (:314,327).  rt_multi_* is that for the GPUs of one node: rank r =
 * HIP device r renders the cyclic row tiles k*tile_rows.. with k % n_gpus == r (scene replicated, pixels seeded from their
 * frame index: the frame does not depend on the partition), exposes and packs its own rows -- with blur passes after an
 * all-gather of one byte of over-exposure bits per pixel --, and the packed rows (4 bytes per pixel in all) are gathered to
 * device 0 with RCCL (grouped ncclSend / ncclRecv, one xGMI link per sending device) and put in frame order there.
 * The RCCL library is loaded when the first context with more than one GPU is created.  One caller at a time per handle. */
typedef struct rt_multi_context* rt_multi_handle;

/* Contexts on HIP devices 0 .. n_gpus-1 and an RCCL communicator over them (n_gpus <= hipGetDeviceCount).
 * With RT_MULTI_LOOPBACK=1 in the environment the n_gpus ranks all live on device 0 and exchange rows by device copies:
 * the whole multi-rank path except the RCCL calls, for tests on a one-GPU machine.
 * Replaces: the allocation block and thread objects of main(), main.cpp:300-306. */
int rt_multi_create(rt_multi_handle* out, int n_gpus);
void rt_multi_destroy(rt_multi_handle m);
int rt_multi_gpus(rt_multi_handle m);
/* rt_set_option on every rank's context. */
int rt_multi_set_option(rt_multi_handle m, int key, int value);
/* rt_upload_scene on every rank (same arguments). */
int rt_multi_upload_scene(rt_multi_handle m, const double* params, int n_prims, const double* prims, int n_mats,
                          const double* mats, int n_verts, const double* verts);
/* The timed section of main() (main.cpp:304-314) on all ranks: render, expose with the scene header's exposure and
 * num_blurs, gather, frame order; the packed frame is copied to the host.
 *   tile_rows : rows per tile of the cyclic partition; 0 = default (8, or one band per rank when there are blur passes)
 *   img       : int32[height*width], what writeImage takes (pngUtil.c:4);  rays: out, ray segments traced on all ranks */
int rt_multi_render_image(rt_multi_handle m, int tile_rows, int32_t* img, uint64_t* rays);
/* The same frame in two calls, so that frames overlap: rt_multi_render_begin enqueues everything on every rank (the ranks
 * beyond the first through persistent worker threads) and returns without waiting; rt_multi_render_end waits for the
 * OLDEST frame begun, copies it out and returns its ray count.  At most two frames may be in flight: the gather, the
 * re-ordering and the copy to the host of frame i run on streams of their own while the ranks render frame i + 1.
 * Replaces main.cpp:304-315 like rt_multi_render_image, which is begin followed by end. */
int rt_multi_render_begin(rt_multi_handle m, int tile_rows);
int rt_multi_render_end(rt_multi_handle m, int32_t* img, uint64_t* rays);
/* Wall-clock milliseconds of the last frame ended (from the start of its begin call to the frame on the host), and the
 * two host-side parts of it: the time the begin call took to enqueue the frame and the time the end call waited. */
double rt_multi_last_ms(rt_multi_handle m);
int rt_multi_last_times(rt_multi_handle m, double* enqueue_ms, double* wait_ms);
/* 1 when the context exchanges rows through RCCL (more than one GPU; or one rank with RT_MULTI_FORCE_RCCL=1 in the
 * environment, which makes that rank send its rows and over-exposure bits to itself through the same RCCL calls), else 0. */
int rt_multi_uses_rccl(rt_multi_handle m);
/* Text of the last error on this context; with a NULL handle: why the last rt_multi_create failed. */
const char* rt_multi_last_error(rt_multi_handle m);
/* Host arithmetic of the gather: the row of the gathered buffer [n_gpus][pad_rows][width] that holds frame row y
 * (pad_rows >= the largest row count of a rank).  No GPU call. */
int rt_gather_source_row(int height, int tile_rows, int world, int pad_rows, int y);

#ifdef __cplusplus
}
#endif
#endif /* RT_HIP_H */
