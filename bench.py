#!/usr/bin/env python3
"""Benchmark of the render hot path on MI355X.

    python bench.py --gpus N --steps K --warmup W            (N = 1)
    python -m torch.distributed.run --nproc-per-node N ... bench.py --gpus N ...   (N > 1)

One step = one frame of the workload as the reference's main() times it (render + expose):
every rank renders its cyclic row tiles of the SAME frame (scene tables already resident in
HBM), packs them to 8-bit pixels (Color::to_int; the timing scenes have num_blurs = 0), and
the packed rows are gathered to rank 0 (RCCL) and de-interleaved there.  The frame is fixed,
so scaling over N is strong scaling.

Prints ONE JSON line on rank 0: Mrays/s (whole job), ms per frame, the FP32-VALU roofline
figures (algorithmic FLOP of the closest-hit scan over the time of all kernels of the frame --
the headline -- and over the broad-phase kernel's own time; executed FLOP; HIP events on the
launch stream) and, at N = 1, the CPU baseline: the reference's own binary
(oracle/_ref, kind "reference") or the oracle port timed on the host cores on a bounded sample.
"""
from __future__ import annotations

import argparse
import json
import os
import sys
import time

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)

import numpy as np  # noqa: E402
import torch  # noqa: E402
import torch.distributed as dist  # noqa: E402

from raytracer_amd import api, synth  # noqa: E402
from raytracer_amd.dist import FramePipeline, default_tile_rows, sum_over_ranks  # noqa: E402

FP32_VALU_PEAK_TFLOPS = 157.3   # MI355X vector FP32, FMA = 2 FLOP


def algorithmic_flops(t, sc):
    """FLOP of the closest-hit scan as the reference defines the work: every ray is tested against
    every solid; 19 per sphere, 12 per plane, 18 + 7*verts per face (FMA = 2)."""
    return 19 * t["sphere_tests"] + 12 * t["plane_tests"] + 18 * t["face_tests"] + 7 * t["face_vert_tests"]


def cpu_baseline(workload, cores):
    """Time the CPU path on a bounded sample (10-30 s of CPU work): the same scene at 1/2 linear resolution
    (identical framing: scale = width/40).  Prefers the compiled reference binary; falls back to the oracle port."""
    sys.path.insert(0, os.path.join(ROOT, "oracle"))
    import pyoracle as po
    cfg = synth.CONFIGS[workload]
    w = max(64, cfg["width"] // 2)
    text, sc = synth.make(workload, width=w, num_threads=cores)
    sample = f"{workload} at {sc.width}x{sc.height} (1/2 linear resolution, same framing), {cores} threads, render+expose as main.cpp:305-315 times it"
    try:
        if po.have_reference():
            r = po.run_reference(text, num_threads=cores, num_blurs=0, timeout=900)
            rays = int(r["rays"].sum())
            return {"value": rays / r["seconds"] / 1e6, "unit": "Mrays/s", "cores": cores, "kind": "reference",
                    "sample": sample, "seconds": r["seconds"], "rays": rays}
    except Exception as e:  # the binary may not run on this box (missing shared library)
        sys.stderr.write(f"bench: reference binary not usable here ({e}); timing the oracle port\n")
    t0 = time.time()
    o = po.render(sc, policy=po.POLICY_REF, rng_mode=po.RNG_STREAM)
    po.expose(o["rgb"], 1.0, 0, workers=cores)
    secs = time.time() - t0
    rays = int(o["rays"].sum())
    return {"value": rays / secs / 1e6, "unit": "Mrays/s", "cores": cores, "kind": "port", "sample": sample,
            "seconds": secs, "rays": rays}


PROFILE_DIR = os.path.join(ROOT, "profiles", "r2")


def measured_traffic(kernel_prefix, key="bytes_per_frame"):
    """Per-frame figure of the kernel's launches from the committed rocprofv3 --pmc passes of this workload
    (profiles/r2/traffic.json, written by tools/summarize_profile.py): HBM bytes, or with key="valu_insts" the
    SQ_INSTS_VALU count; None when no such profile is present."""
    try:
        k = json.load(open(os.path.join(PROFILE_DIR, "traffic.json")))["kernels"]
        vals = [v.get(key) for name, v in k.items() if name.startswith(kernel_prefix)]
        vals = [v for v in vals if v is not None]
        if vals:
            return float(sum(vals))          # every instantiation of the kernel (template arguments in the name)
    except (OSError, KeyError, ValueError):
        pass
    return None


EXECUTED_FLOP_PER_TEST = 15     # what broad_kernel executes per (ray, record): 7 fma + 1 sub (+ 1 bit operation, not counted)
# broad_mfma_kernel (the default): two v_mfma_f32_32x32x16_f16 (32*32*16*2 FLOP each) per 32 rays x 32 records, and on
# the vector ALU one fma per pair (+ 1 bit operation, not counted)
MFMA_FLOP_PER_TEST = 2 * 32 * 32 * 16 * 2 // 1024
MFMA_VALU_FLOP_PER_TEST = 2
F16_MFMA_PEAK_TFLOPS = 2500.0   # MI355X dense f16 / bf16


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=200)     # 1.2 s of frames on C3: long enough for an outside sampler to see the GPU busy
    ap.add_argument("--warmup", type=int, default=5)
    ap.add_argument("--workload", default="c3", choices=sorted(synth.CONFIGS))
    ap.add_argument("--width", type=int, default=None, help="override the frame width (height follows 16:9)")
    ap.add_argument("--tile-rows", type=int, default=None,
                    help="rows per tile of the cyclic partition (default: 8; with blur passes one band of rows per rank)")
    ap.add_argument("--blurs", type=int, default=None, help="override num_blurs of the workload (the timing scenes have 0)")
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--no-pipeline", action="store_true", help="N > 1: finish a frame's gather before rendering the next")
    ap.add_argument("--roofline-frames", type=int, default=3, help="untimed frames with per-launch kernel timing")
    ap.add_argument("--no-walk", action="store_true", help="skip the second measurement (candidate solids from the grid walk)")
    args = ap.parse_args()

    world = int(os.environ.get("WORLD_SIZE", "1"))
    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    if world != args.gpus:
        sys.stderr.write(f"bench: WORLD_SIZE={world} but --gpus {args.gpus}; using WORLD_SIZE\n")
    if not torch.cuda.is_available():
        raise SystemExit("bench.py needs a GPU (the render path has no CPU fallback)")
    # Rehearsal of the N > 1 control flow on a one-GPU box: BENCH_REHEARSE=1 puts every rank on GPU 0 and gathers over
    # gloo (RCCL refuses two ranks on one device).  Its numbers mean nothing; the driver's runs never set it.
    rehearse = os.environ.get("BENCH_REHEARSE") == "1"
    if rehearse:
        local_rank = 0
    torch.cuda.set_device(local_rank)
    device = torch.device("cuda", local_rank)
    if world > 1:
        if rehearse:
            dist.init_process_group(backend="gloo")
        else:
            dist.init_process_group(backend="nccl", device_id=device)

    _, sc = synth.make(args.workload, width=args.width)
    ren = api.Renderer(local_rank)
    ren.upload(sc)
    H, W = sc.height, sc.width
    stream = torch.cuda.current_stream(device)
    exposure, blurs = float(sc.header["exposure"]), int(sc.header["num_blurs"])
    if args.blurs is not None:
        blurs = args.blurs
    if args.tile_rows is None:
        args.tile_rows = default_tile_rows(H, world, blurs)
    ev = []          # (start, end) HIP events around the render kernels of every timed frame, on the launch stream
    timed = [False]

    def render_rows(dst):
        if timed[0]:
            ev.append((torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)))
            ev[-1][0].record(stream)
        ren.render_rows_device(dst.data_ptr(), args.tile_rows, rank, world, stream=stream.cuda_stream)
        if timed[0]:
            ev[-1][1].record(stream)

    def expose(src, rows, dst, nb):
        ren.expose_device(src.data_ptr(), W, rows, exposure, nb, dst.data_ptr(), stream=stream.cuda_stream)

    # one step = one frame as main() times it (main.cpp:305-315): render, expose (8-bit pack; the timing scenes have
    # num_blurs = 0), and the gather of the packed rows to rank 0
    # with blur passes and several ranks: over-exposure bits all-gathered, every rank blooms and packs its own rows
    def expose_bits(src, rows, dst):
        ren.expose_bits_device(src.data_ptr(), W, rows, exposure, dst.data_ptr(), stream=stream.cuda_stream)

    def expose_rows(src, bits_all, pad_rows, dst):
        ren.expose_rows_device(src.data_ptr(), bits_all.data_ptr(), W, H, args.tile_rows, rank, world, pad_rows, exposure, blurs,
                               dst.data_ptr(), stream=stream.cuda_stream)

    # Several ranks: a frame's gather (RCCL's own stream) runs while the next frame is rendered; the frames of the timed
    # region are all gathered and put in row order inside it (flush).  --no-pipeline: gather, then the next render.
    pipe = FramePipeline(H, W, blurs, args.tile_rows, rank, world, device, render_rows, expose, expose_bits, expose_rows,
                         pipelined=world > 1 and not args.no_pipeline)

    for _ in range(args.warmup):
        pipe.step()
        ren.finish()
    pipe.flush()
    torch.cuda.synchronize(device)
    if world > 1:
        dist.barrier()
    torch.cuda.synchronize(device)
    timed[0] = True
    t0 = time.perf_counter()
    for i in range(args.steps):
        pipe.step()
    pipe.flush()
    torch.cuda.synchronize(device)
    if world > 1:
        dist.barrier()
    torch.cuda.synchronize(device)
    elapsed = time.perf_counter() - t0
    rays_local = ren.finish()                     # counters of the last frame; every step renders the same frame
    frame_kernels_ms = float(np.mean([a.elapsed_time(b) for a, b in ev]))   # all kernels of one frame on this GPU
    # Roofline of the dominant kernel, outside the timed region: a few more frames with HIP events around every
    # broad_kernel launch on its launch stream.  Per-launch timing makes the library run the kernels one after the
    # other (in the timed frames the broad phase of one half-frame overlaps the shading of the other, and a launch's
    # duration would include the time it shares its CUs).
    timed[0] = False
    ren.set_option(api.OPT_TIMING_DETAIL, 1)
    broad, serial = [], []
    for _ in range(args.roofline_frames):
        pipe.step()
        pipe.flush()
        ren.finish()
        tm = ren.timing()
        broad.append(tm["broad_ms"] if tm["broad_ms"] > 0 else tm["kernel_ms"])
        serial.append(tm["kernel_ms"])
    if not broad:                                 # --roofline-frames 0 (counter passes): no per-launch timing
        tm = ren.timing()
        broad, serial = [tm["kernel_ms"]], [tm["kernel_ms"]]
    broad_ms, serial_ms = float(np.mean(broad)), float(np.mean(serial))

    def max_over_ranks(x):
        t = torch.tensor([x], dtype=torch.float64, device="cpu" if rehearse else device)
        if world > 1:
            dist.all_reduce(t, op=dist.ReduceOp.MAX)
        return float(t.item())

    elapsed = max_over_ranks(elapsed)
    # Second measurement, same frame, same bracket: candidate solids from the grid walk (RT_OPT_ACCEL) instead of the
    # scan of every record.  Output bits are identical (tests/test_gpu_accel.py); the scan stays the headline because
    # it is the work the reference's closest-hit loop defines and the roofline figure is stated on.
    walk = None
    if not args.no_walk:
        ren.set_option(api.OPT_TIMING_DETAIL, 0)
        ren.set_option(api.OPT_ACCEL, 1)
        for _ in range(max(1, args.warmup)):
            pipe.step()
            ren.finish()
        pipe.flush()
        torch.cuda.synchronize(device)
        if world > 1:
            dist.barrier()
        torch.cuda.synchronize(device)
        t0 = time.perf_counter()
        for i in range(args.steps):
            pipe.step()
        pipe.flush()
        torch.cuda.synchronize(device)
        if world > 1:
            dist.barrier()
        torch.cuda.synchronize(device)
        walk_elapsed = max_over_ranks(time.perf_counter() - t0)
        walk_rays = ren.finish()
        ren.set_option(api.OPT_ACCEL, 0)
        walk = (walk_elapsed, walk_rays)
    rays_frame = sum_over_ranks(rays_local, device, world)
    flops_local = algorithmic_flops(tm, sc)

    if rank == 0:
        ms_per_step = elapsed * 1e3 / args.steps
        value = rays_frame / (ms_per_step * 1e-3) / 1e6
        # Two readings of the same frame.  Headline (frac): the closest-hit scan's ALGORITHMIC FLOP -- every ray against every
        # solid, 19 per sphere, 12 per plane, 18 + 7 per vertex for a face, as the reference's O(N) loop defines the work --
        # divided by the time of ALL kernels of the frame.  kernel_*: the same FLOP over the broad-phase kernel's own time.
        # executed_*: what that kernel really executes per (ray, padded record) whatever the solid's kind (it evaluates a
        # conservative bounding-sphere discriminant -- on the matrix cores by default, 15 FP32 FLOP on the vector ALU with
        # RT_BROAD_MFMA=0; root, epsilon test, closest-hit compare and polygon test run in shade_kernel in FP64 on the few
        # records it keeps).
        frame_s = frame_kernels_ms * 1e-3
        achieved = flops_local / frame_s / 1e12
        kernel_achieved = flops_local / (broad_ms * 1e-3) / 1e12
        padded = (len(sc.prims) + 31) // 32 * 32
        on_mfma = os.environ.get("RT_BROAD_MFMA", "1") != "0"       # the library's default: the scan on the matrix cores
        kname = "broad_mfma_kernel + broad_mfma_binned_kernel" if on_mfma else "broad_kernel"
        kprefix = "broad_mfma" if on_mfma else "broad_kernel"          # the kernels' names in the committed counter passes
        pairs = tm["rays"] * padded
        if on_mfma:
            # the table of the scan (without the always-tested solids when it is in its own order); with the table in its own
            # order the kernels ask the groups' bounding spheres first and compute only the products those leave: the
            # device counts the (32 rays x 32 records) products it puts through the MFMAs
            padded = int(tm["broad_records"])
            pairs = int(tm["broad_tiles"]) * 1024
            executed = MFMA_FLOP_PER_TEST * pairs / (broad_ms * 1e-3) / 1e12
            executed_peak = F16_MFMA_PEAK_TFLOPS
            executed_basis = ("%d f16 MFMA FLOP x (ray, record) pairs put through the MFMAs (%d records in the table%s) / %s time, "
                              "against the dense f16 MFMA peak; beside it %d FP32 FLOP per pair on the vector ALU"
                              % (MFMA_FLOP_PER_TEST, padded, "; products counted on the device: %.1f %% of rays x records"
                                 % (100.0 * pairs / max(1, tm["rays"] * padded)), kname, MFMA_VALU_FLOP_PER_TEST))
        else:
            executed = EXECUTED_FLOP_PER_TEST * pairs / (broad_ms * 1e-3) / 1e12
            executed_peak = FP32_VALU_PEAK_TFLOPS
            executed_basis = "15 FLOP x rays x padded records (%d) / broad_kernel time" % padded
        profiled = args.workload == "c3" and args.width is None and world == 1
        valu = measured_traffic(kprefix, "valu_insts") if profiled else None
        cnt = sc.counts()
        out = {
            "metric": "Mrays/s", "value": value, "unit": "Mrays/s", "n_gpus": world, "steps": args.steps,
            "warmup": args.warmup, "ms_per_step": ms_per_step, "higher_is_better": True, "scaling": "strong",
            "vs_baseline": None, "dtype": ("f16x2 MFMA + f32 broad phase" if on_mfma else "f32 broad phase") + " + f64 exact phase", "data": "synthetic",
            "config": {"workload": f"{args.workload}: {W}x{H}, {cnt['spheres']} spheres + {cnt['planes']} planes + {cnt['faces']} faces, "
                                   f"depth {sc.header['max_ray_depth']}, {sc.spp} spp",
                       "rays_per_frame": rays_frame, "tile_rows": args.tile_rows,
                       "flop_credit": "19 per ray-sphere, 12 per ray-plane, 18 + 7 per vertex per ray-face (46 for a quad), although the FP32 "
                                      "broad phase runs the same bounding-sphere test for every kind of solid",
                       "partition": f"cyclic row tiles over {world} GPU(s), packed 8-bit rows gathered to rank 0",
                       "step": "render + expose (num_blurs %d%s) + gather" % (blurs, ", bloom on every rank from all-gathered over-exposure bits" if pipe.halo else "")
                               + ("; a frame's gather overlaps the next frame's render" if pipe.pipelined else "")},
            "roofline": {"bound": "fp32_valu", "achieved": achieved, "peak": FP32_VALU_PEAK_TFLOPS, "unit": "TFLOP/s",
                         "frac": achieved / FP32_VALU_PEAK_TFLOPS,
                         "basis": "algorithmic FLOP of the closest-hit scan (every ray x every solid) / time of all kernels of the frame",
                         "kernel_achieved": kernel_achieved, "kernel_frac": kernel_achieved / FP32_VALU_PEAK_TFLOPS,
                         "executed": executed, "executed_peak": executed_peak, "executed_frac": executed / executed_peak,
                         "executed_basis": executed_basis,
                         # issue slots used: one VALU wave-instruction per SIMD every 2 cycles at 2.4 GHz is the peak
                         # (SQ_INSTS_VALU of the committed counter pass; it includes the MFMA instructions)
                         "valu_issue_frac": (valu / (1024 * 1.2e9 * broad_ms * 1e-3)) if valu else None,
                         "traffic": measured_traffic(kprefix) if profiled else None,
                         "traffic_unit": "HBM bytes per frame (all %s launches), rocprofv3 FETCH_SIZE x2 + WRITE_SIZE" % kname,
                         "kernel": "%s (all launches of one frame)" % kname, "kernel_ms": broad_ms,
                         "frame_kernels_ms": frame_kernels_ms, "frame_kernels_serial_ms": serial_ms,
                         "rays_this_gpu": rays_local, "candidates_per_ray": tm["candidates"] / max(1, tm["rays"]),
                         "overflow_rays": tm.get("overflow_rays", 0)},
        }
        if walk is not None:
            wms = walk[0] * 1e3 / args.steps
            out["grid_walk"] = {"ms_per_step": wms, "value": rays_frame / (wms * 1e-3) / 1e6, "unit": "Mrays/s",
                                "same_output_bits": True, "rays_this_gpu": walk[1],
                                "note": "RT_OPT_ACCEL=1: candidate solids from a front-to-back walk of a uniform grid instead of the scan of "
                                        "every record; not the headline because it skips work the reference's O(N) closest-hit loop defines"}
        if world == 1 and not args.no_cpu_baseline:
            cores = min(os.cpu_count() or 1, 16)
            out["cpu_baseline"] = cpu_baseline(args.workload, cores)
        print(json.dumps(out), flush=True)
    if world > 1:
        dist.barrier()
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
