#!/usr/bin/env python3
"""Benchmark of the render hot path on MI355X.

    python bench.py --gpus N --steps K --warmup W            (N = 1)
    python -m torch.distributed.run --nproc-per-node N ... bench.py --gpus N ...   (N > 1)

One step = one frame of the workload as the reference's main() times it (render + expose):
every rank renders its cyclic row tiles of the SAME frame (scene tables already resident in
HBM), packs them to 8-bit pixels (Color::to_int; the timing scenes have num_blurs = 0), and
the packed rows are gathered to rank 0 (RCCL) and de-interleaved there.  The frame is fixed,
so scaling over N is strong scaling.

Prints ONE JSON line on rank 0: Mrays/s (whole job) and ms per frame of the library as a caller gets it (it picks
the candidate source per scene); "roofline" = the frame's dominant kernel (shade_kernel: FP64 vector work; executed FLOP
from the committed counter pass over its time measured here with HIP events around every launch, against the FP64
vector peak, datasheet and measured; its HBM bytes against 8 TB/s), the second kernel (the matrix-core scan) beside it,
and two whole-frame figures by name: frac_algorithmic_equiv (the reference's every-ray-times-every-solid FLOP over the
default frame's time: a speed-up figure, the default frame skips most pairs) and frac_in_contract (the same FLOP over a
third timed loop in which every pair IS evaluated).  Also the candidate source the library did not pick, with the
frames' hashes compared, and, at N = 1, the CPU baseline: the reference's own binary (oracle/_ref, kind "reference")
or the oracle port timed on the host cores on a bounded sample.
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


PROFILE_DIRS = [os.path.join(ROOT, "profiles", d) for d in ("r3", "r2")]   # the newest committed counter passes first


def committed_counters(kernel_prefix):
    """Per-frame counters of a kernel (all its instantiations and launches) from the committed rocprofv3 --pmc passes of
    this workload (profiles/rN/traffic.json, written by tools/summarize_profile.py from runs of tools/profile_run.sh):
    dict with bytes_per_frame, valu_insts, fp64_add / fp64_mul / fp64_fma, thread_cycles_valu, active_inst_valu,
    salu_insts, insts -- whichever the profile holds -- and "source"; {} when no profile is present."""
    for d in PROFILE_DIRS:
        try:
            t = json.load(open(os.path.join(d, "traffic.json")))
        except (OSError, ValueError):
            continue
        out = {}
        for name, v in t.get("kernels", {}).items():
            if not name.startswith(kernel_prefix):
                continue
            for key, val in v.items():
                if isinstance(val, (int, float)) and key != "launches":
                    out[key] = out.get(key, 0.0) + float(val)
        if out:
            out["source"] = os.path.relpath(os.path.join(d, "traffic.json"), ROOT) + " (" + str(t.get("source")) + ")"
            return out
    return {}


EXECUTED_FLOP_PER_TEST = 15     # what broad_kernel executes per (ray, record): 7 fma + 1 sub (+ 1 bit operation, not counted)
# the matrix-core scan (the default): two v_mfma_f32_32x32x16_f16 (32*32*16*2 FLOP each) per 32 rays x 32 records, and on
# the vector ALU one fma per pair (+ 1 bit operation, not counted)
MFMA_FLOP_PER_TEST = 2 * 32 * 32 * 16 * 2 // 1024
MFMA_VALU_FLOP_PER_TEST = 2
F16_MFMA_PEAK_TFLOPS = 2500.0   # MI355X dense f16 / bf16
FP64_VALU_PEAK_TFLOPS = 78.6    # MI355X vector FP64, datasheet (FMA = 2 FLOP)
# what tools/ubench_fp64.hip measures on this chip: v_fma_f64 issues at 2.1 ns per wave-instruction and SIMD
FP64_VALU_MEASURED_TFLOPS = 64 * 2 / 2.1e-9 * 1024 / 1e12
HBM_PEAK_BYTES_PER_S = 8.0e12


def timed_frames(pipe, ren, steps, warmup, device, world, max_over_ranks):
    """warmup untimed frames, then `steps` frames between barrier + synchronize on both sides; returns
    (seconds, max over ranks; rays of the last frame on this rank; the last frame, packed, on rank 0)."""
    for _ in range(warmup):
        pipe.step()
        ren.finish()
    pipe.flush()
    torch.cuda.synchronize(device)
    if world > 1:
        dist.barrier()
    torch.cuda.synchronize(device)
    t0 = time.perf_counter()
    frame = None
    for _ in range(steps):
        f = pipe.step()
        frame = f if f is not None else frame
    f = pipe.flush()
    frame = f if f is not None else frame
    torch.cuda.synchronize(device)
    if world > 1:
        dist.barrier()
    torch.cuda.synchronize(device)
    elapsed = max_over_ranks(time.perf_counter() - t0)
    return elapsed, ren.finish(), frame


def frame_hash(frame):
    """sha256 of the packed 8-bit frame (int32 0xRRGGBB per pixel) -- computed outside every timed region."""
    import hashlib
    return hashlib.sha256(frame.cpu().numpy().tobytes()).hexdigest()[:16] if frame is not None else None


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
    ap.add_argument("--no-walk", action="store_true", help="skip the second measurement (the other candidate source)")
    ap.add_argument("--no-in-contract", action="store_true", help="skip the third measurement (every ray x record pair evaluated)")
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

    def make_pipe(r):
        def render_rows(dst):
            if timed[0]:
                ev.append((torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)))
                ev[-1][0].record(stream)
            r.render_rows_device(dst.data_ptr(), args.tile_rows, rank, world, stream=stream.cuda_stream)
            if timed[0]:
                ev[-1][1].record(stream)

        def expose(src, rows, dst, nb):
            r.expose_device(src.data_ptr(), W, rows, exposure, nb, dst.data_ptr(), stream=stream.cuda_stream)

        # with blur passes and several ranks: over-exposure bits all-gathered, every rank blooms and packs its own rows
        def expose_bits(src, rows, dst):
            r.expose_bits_device(src.data_ptr(), W, rows, exposure, dst.data_ptr(), stream=stream.cuda_stream)

        def expose_rows(src, bits_all, pad_rows, dst):
            r.expose_rows_device(src.data_ptr(), bits_all.data_ptr(), W, H, args.tile_rows, rank, world, pad_rows, exposure, blurs,
                                 dst.data_ptr(), stream=stream.cuda_stream)

        # one step = one frame as main() times it (main.cpp:305-315): render, expose (8-bit pack; the timing scenes have
        # num_blurs = 0), and the gather of the packed rows to rank 0.  Several ranks: a frame's gather (RCCL's own stream)
        # runs while the next frame is rendered; the frames of the timed region are all gathered and put in row order inside
        # it (flush).  --no-pipeline: gather, then the next render.
        return FramePipeline(H, W, blurs, args.tile_rows, rank, world, device, render_rows, expose, expose_bits, expose_rows,
                             pipelined=world > 1 and not args.no_pipeline)

    def max_over_ranks(x):
        t = torch.tensor([x], dtype=torch.float64, device="cpu" if rehearse else device)
        if world > 1:
            dist.all_reduce(t, op=dist.ReduceOp.MAX)
        return float(t.item())

    pipe = make_pipe(ren)
    # ---- the headline: the library as a caller gets it (candidate source chosen per scene), `steps` timed frames
    for _ in range(args.warmup):
        pipe.step()
        ren.finish()
    pipe.flush()
    timed[0] = True
    elapsed, rays_local, frame = timed_frames(pipe, ren, args.steps, 0, device, world, max_over_ranks)
    timed[0] = False
    frame_kernels_ms = float(np.mean([a.elapsed_time(b) for a, b in ev]))   # all kernels of one frame on this GPU
    head_hash = frame_hash(frame)
    source_used = int(ren.timing()["accel_used"])
    sub_pipelines = int(ren.timing()["sub_pipelines"])      # parts of this rank's pixels that ran side by side in the timed frames
    source_name = {api.ACCEL_SCAN: "scan", api.ACCEL_WALK: "grid_walk"}[source_used]

    # ---- per-kernel times, outside the timed region: a few more frames with HIP events around every scan (or walk) and
    # every shade_kernel launch on its launch stream.  Per-launch timing makes the library run the kernels one after the
    # other (in the timed frames the two half-frames' kernels overlap, and a launch's duration would include the time it
    # shares its CUs).
    ren.set_option(api.OPT_TIMING_DETAIL, 1)
    broad, shade, serial = [], [], []
    tm = None
    for _ in range(args.roofline_frames):
        pipe.step()
        pipe.flush()
        ren.finish()
        tm = ren.timing()
        broad.append(tm["broad_ms"])
        shade.append(tm["shade_ms"])
        serial.append(tm["kernel_ms"])
    ren.set_option(api.OPT_TIMING_DETAIL, 0)
    if tm is None:                                # --roofline-frames 0 (counter passes): no per-launch timing
        tm = ren.timing()
    broad_ms = float(np.mean(broad)) if broad else None
    shade_ms = float(np.mean(shade)) if shade else None
    serial_ms = float(np.mean(serial)) if serial else float(tm["kernel_ms"])

    # ---- second measurement, same frame, same bracket: the OTHER candidate source (grid walk where the library chose the
    # scan and the other way round); the frames' hashes are compared
    other = None
    if not args.no_walk:
        other_mode = api.ACCEL_WALK if source_used == api.ACCEL_SCAN else api.ACCEL_SCAN
        ren.set_option(api.OPT_ACCEL, other_mode)
        o_elapsed, o_rays, o_frame = timed_frames(pipe, ren, args.steps, max(1, args.warmup), device, world, max_over_ranks)
        other = (o_elapsed, o_rays, frame_hash(o_frame), int(ren.timing()["accel_used"]))
        ren.set_option(api.OPT_ACCEL, api.ACCEL_AUTO)

    # ---- third measurement: the work the reference's closest-hit loop defines, done in full -- every ray against every
    # record (table in file order, no group records, no binning, no grid): a second context created with those switches
    in_contract = None
    if not args.no_in_contract:
        saved = {k: os.environ.get(k) for k in ("RT_MFMA_CULL", "RT_MFMA_BINNED", "RT_ACCEL")}
        os.environ.update(RT_MFMA_CULL="0", RT_MFMA_BINNED="0", RT_ACCEL="0")
        try:
            ren_ic = api.Renderer(local_rank)
        finally:
            for k, v in saved.items():
                if v is None:
                    os.environ.pop(k, None)
                else:
                    os.environ[k] = v
        ren_ic.upload(sc)
        pipe_ic = make_pipe(ren_ic)
        ic_elapsed, ic_rays, ic_frame = timed_frames(pipe_ic, ren_ic, args.steps, max(1, args.warmup), device, world, max_over_ranks)
        tm_ic = ren_ic.timing()
        in_contract = (ic_elapsed, ic_rays, frame_hash(ic_frame), tm_ic)
        del pipe_ic
        ren_ic.close()

    rays_frame = sum_over_ranks(rays_local, device, world)
    flops_local = algorithmic_flops(tm, sc)

    if rank == 0:
        ms_per_step = elapsed * 1e3 / args.steps
        value = rays_frame / (ms_per_step * 1e-3) / 1e6
        on_mfma = os.environ.get("RT_BROAD_MFMA", "1") != "0"       # the library's default: the scan on the matrix cores
        profiled = args.workload == "c3" and args.width is None and world == 1
        cnt = sc.counts()

        # ---- the frame's dominant kernel: shade_kernel (exact closest hit among the candidates + one segment of
        # add_to_path per ray, main.cpp:126-183, geometry.cpp:94-179), FP64 on the vector ALU.  Executed FLOP from the
        # committed counter pass of this workload: 64 lanes x (ADD_F64 + MUL_F64 + 2 FMA_F64) wave-instructions, idle lanes
        # of a wave-instruction included (issue-slot accounting, the way the peak is stated).
        sh = committed_counters("shade_kernel") if profiled else {}
        shade = {"kernel": "shade_kernel (all launches of one frame)", "bound": "fp64_valu", "kernel_ms": shade_ms,
                 "share_of_frame_kernels": (shade_ms / serial_ms) if shade_ms else None}
        if sh and shade_ms and all(k in sh for k in ("fp64_add", "fp64_mul", "fp64_fma")):
            flop = 64.0 * (sh["fp64_add"] + sh["fp64_mul"] + 2.0 * sh["fp64_fma"])
            ach = flop / (shade_ms * 1e-3) / 1e12
            shade.update({"achieved": ach, "peak": FP64_VALU_PEAK_TFLOPS, "unit": "TFLOP/s", "frac": ach / FP64_VALU_PEAK_TFLOPS,
                          "peak_measured": FP64_VALU_MEASURED_TFLOPS, "frac_of_measured_peak": ach / FP64_VALU_MEASURED_TFLOPS,
                          "peak_note": "78.6 = datasheet FP64 vector; %.1f = what tools/ubench_fp64.hip measures on this chip (v_fma_f64 at 2.1 ns "
                                       "per wave-instruction and SIMD)" % FP64_VALU_MEASURED_TFLOPS,
                          "flop_per_frame": flop,
                          "basis": "64 x (SQ_INSTS_VALU_ADD_F64 + MUL_F64 + 2 FMA_F64) of %s / shade_kernel time measured here "
                                   "(HIP events around every launch)" % sh["source"]})
            if sh.get("active_inst_valu"):
                shade["lanes_active_of_64"] = sh.get("thread_cycles_valu", 0.0) / sh["active_inst_valu"]
                # how much of the time the vector ALUs are issuing at all: the counter is in units of four cycles per SIMD
                shade["valu_busy_frac"] = 4.0 * sh["active_inst_valu"] / (1024 * 2.4e9 * shade_ms * 1e-3)
                shade["valu_busy_basis"] = "4 x SQ_ACTIVE_INST_VALU of %s / (1024 SIMDs x 2.4 GHz x shade_kernel time measured here)" % sh["source"]
            if sh.get("insts"):
                shade["salu_share_of_instructions"] = sh.get("salu_insts", 0.0) / sh["insts"]
        if sh.get("bytes_per_frame") and shade_ms:
            shade.update({"traffic": sh["bytes_per_frame"], "traffic_unit": "HBM bytes per frame (FETCH_SIZE x2 + WRITE_SIZE of the committed pass)",
                          "traffic_frac_of_hbm_peak": sh["bytes_per_frame"] / (shade_ms * 1e-3) / HBM_PEAK_BYTES_PER_S,
                          "algorithmic_bytes": 12.0 * W * H + 4.0 * W * H,
                          "algorithmic_bytes_note": "the frame it produces: 24 B of radiance is what main() keeps per pixel; here 12 B (3 x f32 would do) + 4 B packed"})

        # ---- the candidate source (scan on the matrix cores, or the grid walk)
        padded = (len(sc.prims) + 31) // 32 * 32
        scan = {"kernel_ms": broad_ms}
        if source_used == api.ACCEL_WALK:
            scan.update({"kernel": "walk_kernel (all launches of one frame)", "bound": "latency (one ray per lane walks the grid)"})
        elif on_mfma:
            sk = committed_counters("broad_mfma") if profiled else {}
            padded = int(tm["broad_records"])
            pairs = int(tm["broad_tiles"]) * 1024
            scan.update({"kernel": "broad_mfma_kernel + broad_mfma_binned_kernel (all launches of one frame)", "bound": "mfma_f16"})
            if broad_ms:
                executed = MFMA_FLOP_PER_TEST * pairs / (broad_ms * 1e-3) / 1e12
                scan.update({"achieved": executed, "peak": F16_MFMA_PEAK_TFLOPS, "unit": "TFLOP/s", "frac": executed / F16_MFMA_PEAK_TFLOPS,
                             "basis": "%d f16 MFMA FLOP x (ray, record) pairs put through the MFMAs, counted on the device, / time of the scan kernels"
                                      % MFMA_FLOP_PER_TEST,
                             "pairs_evaluated_frac": pairs / max(1, tm["rays"] * padded), "records_in_table": padded,
                             "valu_issue_frac": (sk["valu_insts"] / (1024 * 1.2e9 * broad_ms * 1e-3)) if sk.get("valu_insts") else None,
                             "traffic": sk.get("bytes_per_frame")})
        else:
            scan.update({"kernel": "broad_kernel (all launches of one frame)", "bound": "fp32_valu"})
            if broad_ms:
                executed = EXECUTED_FLOP_PER_TEST * tm["rays"] * padded / (broad_ms * 1e-3) / 1e12
                scan.update({"achieved": executed, "peak": FP32_VALU_PEAK_TFLOPS, "unit": "TFLOP/s", "frac": executed / FP32_VALU_PEAK_TFLOPS})

        dominant, second = (shade, scan) if (shade_ms or 0) >= (broad_ms or 0) else (scan, shade)
        roofline = dict(dominant)
        roofline["second_kernel"] = second
        # the figure the workload's pass criterion names: ALGORITHMIC FLOP of the closest-hit scan -- every ray against every
        # solid, 19 per sphere, 12 per plane, 18 + 7 per vertex for a face, as the reference's O(N) loop defines the work --
        # over the time of all kernels of a frame, against the FP32 vector peak.  The default frame does NOT execute that
        # work (group bounding spheres / the grid skip most pairs): it is a speed-up figure, not a utilisation.  The
        # in-contract figure below divides the same FLOP by the frame time with every pair evaluated.
        frame_s = frame_kernels_ms * 1e-3
        roofline.update({"frame_kernels_ms": frame_kernels_ms, "frame_kernels_serial_ms": serial_ms, "candidate_source": source_name,
                         "algorithmic_flop_per_frame": flops_local,
                         "frac_algorithmic_equiv": flops_local / frame_s / 1e12 / FP32_VALU_PEAK_TFLOPS,
                         "frac_algorithmic_equiv_basis": "algorithmic FLOP of the closest-hit scan (every ray x every solid) / time of all kernels of the "
                                                         "default frame / 157.3 TFLOP/s FP32 vector peak; exceeds what is executed wherever pairs are skipped",
                         "rays_this_gpu": rays_local, "candidates_per_ray": tm["candidates"] / max(1, tm["rays"]),
                         "overflow_rays": tm.get("overflow_rays", 0)})
        if in_contract is not None:
            ic_ms = in_contract[0] * 1e3 / args.steps
            roofline.update({"frac_in_contract": flops_local / (ic_ms * 1e-3) / 1e12 / FP32_VALU_PEAK_TFLOPS,
                             "in_contract": {"ms_per_step": ic_ms, "value": rays_frame / (ic_ms * 1e-3) / 1e6, "unit": "Mrays/s",
                                             "pairs_evaluated_frac": int(in_contract[3]["broad_tiles"]) * 1024 / max(1, in_contract[3]["rays"] * int(in_contract[3]["broad_records"])),
                                             "same_output_bits": in_contract[2] == head_hash, "rays_this_gpu": in_contract[1],
                                             "note": "RT_MFMA_CULL=0 RT_MFMA_BINNED=0 RT_ACCEL=0: table in file order, every (ray, record) pair through the "
                                                     "matrix-core test, same timed bracket (wall clock per frame, kernels of two half-frames overlapped)"}})
        out = {
            "metric": "Mrays/s", "value": value, "unit": "Mrays/s", "n_gpus": world, "steps": args.steps,
            "warmup": args.warmup, "ms_per_step": ms_per_step, "higher_is_better": True, "scaling": "strong",
            "vs_baseline": None, "dtype": "f64 exact phase + " + ("f16x2 MFMA / f32 candidate filter" if on_mfma else "f32 candidate filter"), "data": "synthetic",
            "config": {"workload": f"{args.workload}: {W}x{H}, {cnt['spheres']} spheres + {cnt['planes']} planes + {cnt['faces']} faces, "
                                   f"depth {sc.header['max_ray_depth']}, {sc.spp} spp",
                       "rays_per_frame": rays_frame, "tile_rows": args.tile_rows, "candidate_source": source_name, "sub_pipelines": sub_pipelines,
                       "frame_sha256_16": head_hash,
                       "flop_credit": "algorithmic figures: 19 per ray-sphere, 12 per ray-plane, 18 + 7 per vertex per ray-face (46 for a quad)",
                       "partition": f"cyclic row tiles over {world} GPU(s), packed 8-bit rows gathered to rank 0",
                       "step": "render + expose (num_blurs %d%s) + gather" % (blurs, ", bloom on every rank from all-gathered over-exposure bits" if pipe.halo else "")
                               + ("; a frame's gather overlaps the next frame's render" if pipe.pipelined else "")},
            "roofline": roofline,
        }
        if other is not None:
            oms = other[0] * 1e3 / args.steps
            key = "grid_walk" if other[3] == api.ACCEL_WALK else "scan"
            out[key] = {"ms_per_step": oms, "value": rays_frame / (oms * 1e-3) / 1e6, "unit": "Mrays/s",
                        "same_output_bits": other[2] == head_hash, "frame_sha256_16": other[2], "rays_this_gpu": other[1],
                        "note": "the candidate source the library did not pick for this scene, forced through RT_OPT_ACCEL; same timed bracket"}
        if world == 1 and not args.no_cpu_baseline:
            # threads of the CPU run = the box's CPU share for one GPU (OMP_NUM_THREADS on the box; 16), not the machine's count
            host_cpus = os.cpu_count() or 1
            share = int(os.environ.get("OMP_NUM_THREADS", "16") or 16)
            cores = max(1, min(host_cpus, share))
            out["cpu_baseline"] = cpu_baseline(args.workload, cores)
            out["cpu_baseline"].update({"host_cpus": host_cpus,
                                        "cores_note": "threads used = min(os.cpu_count() = %d, this job's CPU share OMP_NUM_THREADS = %d)" % (host_cpus, share)})
        print(json.dumps(out), flush=True)
    if world > 1:
        dist.barrier()
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
