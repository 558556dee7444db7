#!/usr/bin/env python3
"""Benchmark of the path-tracing hot path on MI355X.

    python bench.py --gpus N --steps K --warmup W [--workload c2_spheres|c3_mesh|c4_mesh_4k|c5_stress]

A step is one pass of the hot path over one batch: every GPU renders `--frames` frames (default
256, 1 sample per pixel per frame, depth 8) of the workload's image in ONE kernel launch, inputs
(scene, accumulation buffer) already resident in HBM; with N > 1 each rank owns a different
contiguous frame range and the float4 accumulation images are summed by one RCCL all-reduce
inside the step (weak scaling: frames per GPU fixed, total samples grow with N).
Metric: Mpaths/s = width * height * frames * rays_per_pixel * N / seconds / 1e6 (one path = one
Trace() call of Raytracing.comp:408-439).

Rank 0 prints ONE JSON line.  `roofline` prices the render kernel against the FP32 vector peak
using algorithmic FLOPs from exact event counters (an instrumented launch of the same config,
outside the timed region) and the kernel's own HIP-event time over the timed region.
`cpu_baseline` times the CPU oracle (test infrastructure, OpenMP) on a bounded sample of the same
workload on this machine's host cores.
"""
import argparse
import json
import os
import sys
import time

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)

# 256 CU x 4 SIMD x 32 lanes per cycle x 2 flops (one v_fma_f32 per wave64 every 2 cycles) x 2.4 GHz.
# The arithmetic contract rounds every multiply and add separately, so an instruction carries one flop:
# the ceiling of this contract is half the peak (78.6), 69.9 at the 2.25 cycles per instruction that
# v_mul_f32 / v_add_f32 were measured to issue at on the box (tools/ubench).
FP32_VECTOR_PEAK_TFLOPS = 157.3
HBM_PEAK_GBS = 8000.0
FP32_UNFUSED_PEAK_TFLOPS = 78.6
FP32_UNFUSED_MEASURED_TFLOPS = 69.9

WORKLOADS = {
    # name: (scene, width, height, max_bounce)
    "c2_spheres": ("spheres", 1920, 1080, 8),
    "c3_mesh": ("reference", 1920, 1080, 8),
    "c4_mesh_4k": ("reference", 3840, 2160, 8),
    "c1_plumbing": ("reference", 640, 360, 4),
    # 10^6 spheres + 10^6 triangles (generated, seed 12345), traced through the hierarchy
    "c5_stress": ("stress", 1920, 1080, 16),
    # the same scene with every fourth sphere and mesh strip transmitting (index of refraction 1.5): an EXTENSION,
    # the reference has no such material (Raytracing.comp:26-34); no parity target, reported separately from c5_stress
    "c5_stress_glass": ("stress_glass", 1920, 1080, 16),
}


def build_scene(R, scene_name):
    """(product scene, camera pose) of a workload."""
    if scene_name == "stress":
        return R.scenes.stress_scene()
    if scene_name == "stress_glass":
        return R.scenes.stress_scene(glass=True)
    return (R.sphere_scene() if scene_name == "spheres" else R.reference_scene()), R.BENCH_POSE


def cpu_baseline(workload, frames_hint, budget_s, profile="strict"):
    """Time the oracle on a bounded sample of the workload (same image, fewer frames; for the stress
    scene a coarser image of the same view)."""
    sys.path.insert(0, os.path.join(ROOT, "tests"))
    import cases
    import oracle_lib as O
    scene_name, w, h, mb = WORKLOADS[workload]
    try:
        affinity = len(os.sched_getaffinity(0))
    except AttributeError:
        affinity = os.cpu_count() or 1
    # the threads this process may use: its CPU affinity, or less when the job's share of the machine is
    # stated through OMP_NUM_THREADS (a GPU box shows every CPU of the host to each of its jobs)
    omp = os.environ.get("OMP_NUM_THREADS", "")
    cores = max(1, min(affinity, int(omp)) if omp.isdigit() and int(omp) > 0 else affinity)
    if scene_name.startswith("stress"):
        import raytracer_amd as R
        psc, pose = build_scene(R, scene_name)
        sc = O.Scene.from_product(psc)
        O.lib(profile).orc_set_transmission(1 if scene_name == "stress_glass" else 0)
        # the exhaustive loops cost about 0.1 s per pixel here, and the oracle's threads split image rows:
        # the sample is the same view rendered at 64x36 (one path per 30x30 block of the benchmark's pixels)
        sw, sh = 64, 36
        fd = O.frame_data(sw, sh, pos=pose[0], yaw=pose[1], pitch=pose[2], rpp=1, max_bounce=mb, scene=sc)
        t0 = time.perf_counter()
        O.render(fd, sc, 1, 1, n_threads=cores, want_debug=False, profile=profile)
        dt = time.perf_counter() - t0
        return {
            "value": sw * sh / dt / 1e6, "unit": "Mpaths/s", "cores": cores, "kind": "port",
            "sample": "%s scene and view at %dx%d instead of %dx%d, depth %d, frame 1, exhaustive loops over all "
                      "primitives as the reference makes them (%.1f s of oracle time, OpenMP over rows)" % (
                          workload, sw, sh, w, h, mb, dt),
        }
    case = dict(scene=scene_name, pose=cases.BENCH_POSE, size=(w, h), rpp=1, max_bounce=mb)
    sc = cases.oracle_scene(case)
    fd = cases.oracle_frame_data(case, sc)
    t0 = time.perf_counter()
    O.render(fd, sc, 1, 1, n_threads=cores, want_debug=False, profile=profile)
    t1 = time.perf_counter() - t0
    n = max(1, min(frames_hint, int(budget_s / max(t1, 1e-3))))
    t0 = time.perf_counter()
    O.render(fd, sc, 1, n, n_threads=cores, want_debug=False, profile=profile)
    dt = time.perf_counter() - t0
    return {
        "value": w * h * n / dt / 1e6, "unit": "Mpaths/s", "cores": cores, "kind": "port",
        "sample": "%s %dx%d depth %d, frames 1..%d of the benchmark's (%.1f s of oracle time, OpenMP over rows)" % (
            workload, w, h, mb, n, dt),
    }


def self_launch(n_gpus):
    """`python bench.py --gpus N` without a launcher: start the N ranks as fresh child processes
    (torch.distributed.run, one per GPU) and pass on rank 0's JSON line and their exit code.  This
    process never touches a GPU (torch.cuda.device_count() does not initialise one)."""
    import socket
    import subprocess
    import torch
    have = torch.cuda.device_count()
    if have < n_gpus and os.environ.get("RTMI_BENCH_REHEARSE") != "1":
        sys.stderr.write("bench.py: --gpus %d asked for, but only %d device%s visible on this machine\n" % (
            n_gpus, have, " is" if have == 1 else "s are"))
        return 2
    with socket.socket() as sk:
        sk.bind(("127.0.0.1", 0))
        port = sk.getsockname()[1]
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", str(n_gpus),
           "--master-addr", "127.0.0.1", "--master-port", str(port), os.path.abspath(__file__)] + sys.argv[1:]
    env = dict(os.environ)
    env.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")
    return subprocess.call(cmd, env=env)


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=100,
                    help="timed steps (default 100: a timed region of 0.8 s on the default workload, long enough for a utilisation sampler to see it)")
    ap.add_argument("--warmup", type=int, default=5)
    ap.add_argument("--workload", default="c2_spheres", choices=sorted(WORKLOADS))
    ap.add_argument("--frames", type=int, default=None, help="frames per GPU per step (default 256; 512 for c4, 64 for c1, 16 for c5)")
    ap.add_argument("--variant", type=int, default=None, choices=[0, 1, 2, 4],
                    help="0 nested megakernel, 1 regenerating megakernel, 2 classify + wave-cooperative paths (default), "
                         "4 pixel-streaming kernel (hierarchy scenes; default for c5)")
    ap.add_argument("--profile", default="strict", choices=["strict", "fma"],
                    help="arithmetic profile: strict (default; every multiply and add rounded separately, the profile the "
                         "headline number is quoted in) or fma (fused multiply-adds at a fixed list of sites)")
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--cpu-seconds", type=float, default=12.0)
    ap.add_argument("--no-pipeline", action="store_true",
                    help="N > 1: one context per rank, every step's collective finishes before the next step's render starts "
                         "(default: two contexts take alternate steps, so the collective overlaps the next render)")
    ap.add_argument("--reduce", default="reduce", choices=["reduce", "allreduce"],
                    help="N > 1: sum the images onto rank 0 only (default; the only rank that resolves and would "
                         "dump the picture) or onto every rank")
    args = ap.parse_args()
    if args.gpus > 1 and "WORLD_SIZE" not in os.environ:
        sys.exit(self_launch(args.gpus))

    import torch
    import torch.distributed as dist

    import raytracer_amd as R
    from raytracer_amd.distributed import ShardedRenderer

    rank = int(os.environ.get("RANK", "0"))
    world = int(os.environ.get("WORLD_SIZE", "1"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    args.gpus = world
    if not torch.cuda.is_available():
        sys.exit("bench.py: no GPU visible; the library has no CPU fallback")
    # Rehearsal aid for a one-GPU box: RTMI_BENCH_REHEARSE=1 puts every rank on GPU 0 and swaps RCCL
    # for gloo (RCCL refuses two ranks on one device).  Never set by the driver; numbers are meaningless.
    rehearse = os.environ.get("RTMI_BENCH_REHEARSE") == "1"
    if rehearse:
        local_rank = 0
    torch.cuda.set_device(local_rank)
    device = torch.device("cuda", local_rank)
    if world > 1:
        if rehearse:
            dist.init_process_group("gloo", rank=rank, world_size=world)
        else:
            dist.init_process_group("nccl", rank=rank, world_size=world, device_id=device)

    scene_name, width, height, max_bounce = WORKLOADS[args.workload]
    if args.variant is None:
        args.variant = 4 if args.workload.startswith("c5") else 2
    frames = args.frames or {"c4_mesh_4k": 512, "c1_plumbing": 64, "c5_stress": 16, "c5_stress_glass": 16}.get(args.workload, 256)
    scene, pose = build_scene(R, scene_name)
    fd = R.make_frame_data(width, height, scene, pose, rays_per_pixel=1, max_bounce=max_bounce)
    total_frames = frames * world

    def make_renderer():
        r = R.Renderer(width, height, device=local_rank, variant=args.variant, profile=args.profile)
        if scene_name == "stress_glass":
            r.enable_transmission(True)
        r.upload_scene(scene)
        return r

    renderer = make_renderer()
    sr = ShardedRenderer(renderer, device)
    # N > 1: consecutive steps (images) alternate between two contexts, each with an accumulation image and a
    # stream of its own, so that the sum of one image over the ranks runs while the next image is being rendered
    # (what a host producing a sequence of images would do); every step is still a full render of this rank's
    # frames plus the collective plus, on the rank holding the sum, the resolve.  --no-pipeline keeps one context.
    pipelined = world > 1 and not args.no_pipeline
    renderers = [renderer, make_renderer()] if pipelined else [renderer]
    srs = [sr, ShardedRenderer(renderers[1], device)] if pipelined else [sr]
    step_no = [0]

    dst = 0 if args.reduce == "reduce" else None

    def step():
        s = srs[step_no[0] % len(srs)]
        step_no[0] += 1
        s.render(fd, total_frames, dst=dst, join=not pipelined)
        if world > 1 and (dst is None or rank == dst):
            s.resolve(total_frames, join=not pipelined)

    # algorithmic work of this rank's launch, from exact event counters (instrumented kernel, untimed)
    first = 1 + rank * frames
    renderer.enable_validation(True)
    renderer.reset_counters()
    renderer.render_frames(fd, first, frames)
    counters = renderer.counters()
    flops_per_launch = R.algorithmic_flops(counters)
    # What the kernels execute is less than what the algorithm specifies: the shader casts the same primary
    # ray for every sample of a pixel, and variants 2 and 4 evaluate that segment once per pixel and launch
    # (variant 2 also the ray set-up, and the sky term of pixels that see only sky).  The events of one
    # primary segment per pixel are those of a one-frame, one-sample, one-bounce launch.
    fd1 = R.make_frame_data(width, height, scene, pose, rays_per_pixel=1, max_bounce=1)
    renderer.reset_counters()
    renderer.render_frames(fd1, first, 1)
    primary = renderer.counters()
    renderer.enable_validation(False)
    hoisted_per_repeat = R.algorithmic_flops({k: (primary[k] if k in ("n_sphere_tests", "n_sphere_disc_ge0", "n_box_tests", "n_tri_tests")
                                                  else 0) for k in primary})
    if args.variant == 2:
        hoisted_per_repeat += 74 * width * height + 63 * primary["n_env_evals"]
    repeats_saved = frames * fd.rays_per_pixel - 1 if args.variant in (2, 4) else 0
    flops_executed = flops_per_launch - repeats_saved * hoisted_per_repeat

    for _ in range(args.warmup):
        step()
    torch.cuda.synchronize(device)
    if world > 1:
        dist.barrier()
    torch.cuda.synchronize(device)
    for r_ in renderers:
        r_.timing_reset(True)
    if world > 1:
        for s_ in srs:
            s_.time_collectives(True)
    t0 = time.perf_counter()
    for _ in range(args.steps):
        step()
    torch.cuda.synchronize(device)
    if world > 1:
        dist.barrier()
    torch.cuda.synchronize(device)
    elapsed = time.perf_counter() - t0
    kernel_ms_total, launches = 0.0, 0
    for r_ in renderers:
        ms_, n_ = r_.timing_read()
        kernel_ms_total += ms_
        launches += n_
        r_.timing_reset(False)
    reduce_ms = kernel_ms_max = kernel_ms_min = None
    if world > 1:
        coll_ms_total, n_coll = 0.0, 0
        for s_ in srs:
            ms_, n_ = s_.collective_ms()
            coll_ms_total += ms_
            n_coll += n_
            s_.time_collectives(False)
        # per rank: wall seconds, render-kernel ms per step, collective ms per step (transfer + wait for slower ranks)
        mine = torch.tensor([elapsed, kernel_ms_total / max(1, launches), coll_ms_total / max(1, n_coll)],
                            dtype=torch.float64, device=device)
        hi, lo = mine.clone(), mine.clone()
        dist.all_reduce(hi, op=dist.ReduceOp.MAX)
        dist.all_reduce(lo, op=dist.ReduceOp.MIN)
        elapsed = float(hi[0].item())
        kernel_ms_max, kernel_ms_min = float(hi[1].item()), float(lo[1].item())
        # the rank that renders longest waits least: its figure is closest to the transfer itself
        reduce_ms = {"min_over_ranks": float(lo[2].item()), "max_over_ranks": float(hi[2].item()),
                     "rank0": float(mine[2].item()),
                     "is": "HIP events on the rank's stream around the collective, per step: transfer + wait for slower ranks"}

    paths_per_step = width * height * frames * world * fd.rays_per_pixel
    value = paths_per_step * args.steps / elapsed / 1e6
    kernel_ms = kernel_ms_total / max(1, launches)
    achieved_tflops = flops_per_launch / (kernel_ms * 1e-3) / 1e12
    executed_tflops = flops_executed / (kernel_ms * 1e-3) / 1e12
    algorithmic_bytes = width * height * (16 + 4 + (0 if first == 1 else 16))   # accum write (+read) + rgba8
    # HBM bytes per launch from the PMC passes kept under profiles/ (only valid for the configuration
    # they were collected on)
    traffic = None
    tpath = os.path.join(ROOT, "profiles", "traffic_%s.json" % args.workload)
    if os.path.exists(tpath):
        with open(tpath) as f:
            tj = json.load(f)
        if tj.get("frames", 256) == frames and tj.get("variant", 2) == args.variant:
            traffic = tj.get("hbm_bytes_per_launch")

    result = {
        "metric": "Mpaths/s at 1080p depth-8" if (height, max_bounce) == (1080, 8) else "Mpaths/s",
        "value": value, "unit": "Mpaths/s", "n_gpus": world, "steps": args.steps, "warmup": args.warmup,
        "ms_per_step": elapsed / args.steps * 1e3, "higher_is_better": True, "scaling": "weak",
        "vs_baseline": None, "dtype": "f32", "data": "synthetic",
        "config": {
            "workload": "%s: %s scene (%d spheres, %d triangles in %d meshes), %dx%d, %d frames x 1 spp per GPU per step, "
                        "max depth %d, bench pose%s" % (args.workload, scene_name, scene.n_spheres, scene.n_triangles,
                                                      scene.n_meshes, width, height, frames, max_bounce,
                                                      "; EXTENSION: every fourth sphere and mesh strip transmits (ior 1.5), no counterpart in "
                                                      "the reference, no parity target" if scene_name == "stress_glass" else ""),
            "closest_hit": "hierarchy" if renderer.acceleration() == R.abi.ACCEL_BVH else "exhaustive loops",
            "variant": {0: "megakernel", 1: "regenerate", 2: "compact", 4: "stream"}[args.variant],
            "arithmetic_profile": renderer.profile(),
            "frames_per_gpu": frames, "total_frames": total_frames,
            "parallelism": "frame-range split x%d + RCCL %s(sum) of float4 accumulation%s" % (
                world, "reduce to rank 0" if dst is not None else "all-reduce",
                "; consecutive steps alternate between two contexts so that a step's collective overlaps the next step's render"
                if pipelined else "") if world > 1 else "single GPU",
        },
        "roofline": {
            "bound": "fp32_valu", "achieved": achieved_tflops, "peak": FP32_VECTOR_PEAK_TFLOPS, "unit": "TFLOP/s",
            "frac": achieved_tflops / FP32_VECTOR_PEAK_TFLOPS, "traffic": traffic,
            "kernel": {0: "k_megakernel", 1: "k_regenerate", 2: "k_classify+k_paths (k_paths dominates)", 4: "k_stream"}[args.variant],
            "kernel_ms": kernel_ms, "launches_timed": launches,
            "algorithmic_flops_per_launch": flops_per_launch,
            "algorithmic_hbm_bytes_per_launch": algorithmic_bytes,
            "hbm_gbs_algorithmic": algorithmic_bytes / (kernel_ms * 1e-3) / 1e9, "hbm_peak_gbs": HBM_PEAK_GBS,
            # algorithmic = the reference algorithm's events (every path walks its primary segment);
            # executed = the same minus the primary segments, ray set-ups and sky terms that are evaluated once per
            # pixel and launch instead of once per path
            "executed_flops_per_launch": flops_executed,
            "achieved_executed": executed_tflops,
            "frac_executed": executed_tflops / FP32_VECTOR_PEAK_TFLOPS,
            # one flop per instruction (no FMA) halves the peak; 69.9 with the measured 2.25 cycles per v_mul / v_add
            "peak_unfused": FP32_UNFUSED_PEAK_TFLOPS,
            "peak_unfused_measured_issue": FP32_UNFUSED_MEASURED_TFLOPS,
            "frac_executed_of_unfused_peak": executed_tflops / FP32_UNFUSED_PEAK_TFLOPS,
            "note": "peak = one v_fma_f32 per wave64 per 2 cycles per SIMD; the arithmetic contract rounds every multiply and "
                    "add separately (no FMA), so its ceiling is peak_unfused; frac prices the reference algorithm's flops, "
                    "frac_executed the flops the kernels actually perform",
        },
        "kernel_mpaths_per_s": width * height * frames / (kernel_ms * 1e-3) / 1e6,
        "msegments_per_s": counters["n_segments"] / (kernel_ms * 1e-3) / 1e6,
        "counters_rank0_launch": counters,
    }
    if args.profile == "fma":
        # a fused multiply-add counts as two flops in the algorithmic figure, as the separate multiply and add do;
        # the sites that are not multiply-add shaped (compares, selects, divisions, the RNG) still issue one
        # operation per instruction
        result["roofline"]["note"] = ("fused profile: multiply-add shaped sites issue one v_fma_f32 (2 flops); peak = one "
                                      "v_fma_f32 per wave64 per 2 cycles per SIMD; frac prices the reference algorithm's "
                                      "flops, frac_executed the flops the kernels actually perform")
        for k in ("peak_unfused", "peak_unfused_measured_issue", "frac_executed_of_unfused_peak"):
            result["roofline"].pop(k)
    if traffic is not None:
        # what the counters saw reach the memory side, at this run's launch time
        result["roofline"]["hbm_gbs_from_traffic"] = traffic / (kernel_ms * 1e-3) / 1e9
    if renderer.acceleration() == R.abi.ACCEL_BVH:
        # The hierarchy walk is a chain of dependent gathers, bound by the memory system and not by
        # arithmetic: price it by the bytes the walk has to fetch (one node record, i.e. two child boxes, per
        # node visit, a 20-byte record per sphere tested, a 68-byte record per triangle tested).
        mesh_gates = counters["n_segments"] * fd.mesh_number
        node_visits = (counters["n_box_tests"] - mesh_gates) // 2
        node_bytes = R.abi.load().rtmi_bvh_node_bytes()   # what one visit fetches (both child boxes)
        gather_bytes = node_visits * node_bytes + counters["n_sphere_tests"] * 20 + counters["n_tri_tests"] * 68 + algorithmic_bytes
        gather_gbs = gather_bytes / (kernel_ms * 1e-3) / 1e9
        # headline: the bytes the PMC counters saw reach the memory side (profiles/traffic_*.json, collected on
        # this configuration) over this run's launch time; the gather bytes, most of which L2 and Infinity Cache
        # serve, are a secondary figure.  Without a counter file for this configuration only the latter exists.
        measured_gbs = traffic / (kernel_ms * 1e-3) / 1e9 if traffic is not None else None
        gbs = measured_gbs if measured_gbs is not None else gather_gbs
        result["roofline"].update({
            "bound": "hbm", "achieved": gbs, "peak": HBM_PEAK_GBS, "unit": "GB/s", "frac": gbs / HBM_PEAK_GBS,
            "achieved_is": "measured HBM traffic (PMC) / kernel time" if measured_gbs is not None else
                           "algorithmic gather bytes / kernel time (no PMC traffic file for this configuration)",
            "algorithmic_gather_bytes_per_launch": gather_bytes, "gather_gbs": gather_gbs,
            "gather_frac_of_hbm_peak": gather_gbs / HBM_PEAK_GBS, "node_visits_per_launch": node_visits,
            "node_bytes": node_bytes,
            "fp32_tflops_algorithmic": achieved_tflops,
            "note": "hierarchy traversal: a chain of dependent gathers (one node record per visit), bound by memory latency and instruction "
                    "issue rather than by HBM bandwidth; frac is the measured share of the HBM peak, gather_* price the "
                    "node and primitive records the walk fetches wherever they are served from",
        })
    if world > 1:
        result["reduce_ms"] = reduce_ms
        result["kernel_ms_over_ranks"] = {"min": kernel_ms_min, "max": kernel_ms_max}
        result["reduce_payload_bytes"] = width * height * 16
    if rank == 0 and world == 1 and not args.no_cpu_baseline:
        result["cpu_baseline"] = cpu_baseline(args.workload, frames, args.cpu_seconds, args.profile)
    if rank == 0:
        print(json.dumps(result))
    sr = None
    srs = None
    for r_ in renderers:
        r_.close()
    if world > 1:
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
