#!/usr/bin/env python3
"""Headline benchmark: Mrays/s (primary + shadow + reflection) of the trace path on the Stanford Dragon scene.

  python bench.py --gpus 1 --steps K --warmup W          Dragon 3840x2160, maxdepth 5, all features, one MI355X
  torchrun ... bench.py --gpus N ...                       the same frame cut into interleaved tiles over N GPUs,
                                                           RCCL gather of the tile buffers + reassembly on rank 0
  --workload dragon8k | cornell | spheres | grid           the other benchmark configurations (extra measurements)

A step is one frame: every kernel of the path over the whole image, scene and framebuffer resident in HBM; frames are
issued round-robin on several streams (frames in flight).  After the timed loop the last frame of every stream is compared
byte for byte with a frame rendered alone, and the synchronous host call's image with that; no line is printed otherwise.
Rank 0 prints ONE JSON line.  Rays are counted the way the reference's control flow spawns them
(RayTracer/main.cpp:176-223): primary = pixels, shadow = lights x hits at every depth, reflection = hits at depth < maxdepth.

Beside `value` the line carries `steady_state` (the same pipeline over 80 frames, outside the timed region),
`shader_clock_hz` (measured beside those frames), `frame_alone_ms` (one frame, nothing else on the GPU), `sync_host_ms`
(mirt_render: the call that replaces main.cpp:137-163, host buffer out), `roofline` (see roofline()) and `cpu_baseline` (rank 0, N = 1: the
reference's own compiled classes, oracle/_ref, kind "reference" -- or the C restatement, kind "port" -- on a bounded row
sample of the same image, one thread, plus what 16 processes of it reach together).
"""
import argparse
import json
import os
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)

# HIP spreads a process's streams over GPU_MAX_HW_QUEUES hardware queues (4 unless told otherwise) and launches of streams
# that share a queue run one after the other, so with more frames in flight than queues part of the pipeline is serial
# (4 queues: 9.7 Grays/s with 6 or 8 frames in flight; 16 queues: 10.7 with 6, 11.0 with 8).  Read when HIP initialises:
# set before anything touches the GPU.  A larger value already in the environment is kept.
FRAMES_IN_FLIGHT = 8
if int(os.environ.get("GPU_MAX_HW_QUEUES", "0") or 0) < 16:
    os.environ["GPU_MAX_HW_QUEUES"] = "16"

HBM_PEAK_GBS = 8000.0
RANDOM_READ_REQUESTS_G = 56.0  # memory-side read requests per ns at best (random 64- or 128-byte records over 1 GiB, measured)
NODE_BYTES, TRI_BYTES, SPHERE_BYTES, RAY_STATE_BYTES = 32, 40, 116, 32
FP32_PEAK_TFLOPS = 157.3
NODE_FLOPS, TRI_FLOPS, SPHERE_FLOPS = 24, 60, 120

# name -> (scene text generator in raytracer_amd.workloads, N=1 size, description).  "dragon" is the configuration the
# metric is quoted on; the others are the remaining benchmark configurations, selectable for extra measurements.
WORKLOADS = {
    "dragon": ("dragon", (3840, 2160), "final_scenes/scene7 (Stanford Dragon, 100000 triangles, 1 point light) %dx%d maxdepth 5"),
    "dragon8k": ("dragon", (7680, 4320), "final_scenes/scene7 (Stanford Dragon, 100000 triangles, 1 point light) %dx%d maxdepth 5"),
    "cornell": ("cornell", (640, 480), "final_scenes/scene6 (Cornell box, 34 triangles + 4 spheres, 2 lights) %dx%d maxdepth 5"),
    "spheres": ("spheres", (1920, 1080), "final_scenes/scene5 + 200 ellipsoids (1200 spheres, 2 lights) %dx%d maxdepth 5"),
    "grid": ("dragon_grid", (3840, 2160), "10x10 Stanford dragons (10000000 triangles, 16 point lights) %dx%d maxdepth 8"),
}


def cpu_baseline_and_visits(flat, scene_text, width, height, row_step, use_reference):
    """Oracle leg: (cpu_baseline dict, visits[3][3], sample ray counts[3]).  Only the checker is timed here."""
    from oracle.bind import CPort, RefOracle, REF_LIB
    port = CPort(flat)
    t0 = time.perf_counter()
    _, st = port.render_rows(width, height, 0, height, row_step=row_step)
    t_port = time.perf_counter() - t0
    rays = [st["primary_rays"], st["shadow_rays"], st["reflection_rays"]]
    sample = "every %dth row of %dx%d (%d rays), 1 thread" % (row_step, width, height, sum(rays))
    base = {"value": sum(rays) / t_port / 1e6, "unit": "Mrays/s", "cores": 1, "kind": "port", "sample": sample + ", C restatement with visit counters on"}
    if use_reference and os.path.exists(REF_LIB):
        import tempfile
        with tempfile.NamedTemporaryFile("wb", suffix=".test", delete=False) as f:
            f.write(scene_text)
        try:
            ref = RefOracle(f.name)  # the reference's own parser + Partition build
            _, rs = ref.render(width, height, row_step=row_step)
            ref.close()
        finally:
            os.unlink(f.name)
        n = rs["primary"] + rs["shadow"] + rs["reflection"]
        assert [rs["primary"], rs["shadow"], rs["reflection"]] == rays, "reference and restatement disagree on ray counts"
        base = {"value": n / rs["seconds"] / 1e6, "unit": "Mrays/s", "cores": 1, "kind": "reference",
                "sample": sample + "; reference classes and the shading functions of main.cpp compiled unmodified (g++ -O2), pixel loop only"}
        # the reference has no threads; what all host cores could do: P processes, each with the reference's own scene
        # and its own rows (every 4th row of the image, dealt round-robin), rays summed over the slowest worker's loop time
        try:
            import subprocess
            P = min(16, os.cpu_count() or 1)
            with tempfile.NamedTemporaryFile("wb", suffix=".test", delete=False) as f:
                f.write(scene_text)
            try:
                procs = [subprocess.Popen([sys.executable, "-m", "oracle.cpu_rows", f.name, str(width), str(height), str(4 * p), str(4 * P)],
                                          cwd=ROOT, stdout=subprocess.PIPE, stderr=subprocess.DEVNULL) for p in range(P)]
                outs = [json.loads(pr.communicate(timeout=300)[0]) for pr in procs]
            finally:
                os.unlink(f.name)
            base["all_cores"] = {"value": sum(o["rays"] for o in outs) / max(o["seconds"] for o in outs) / 1e6, "unit": "Mrays/s", "cores": P,
                                 "sample": "every 4th row, %d processes" % P}
        except Exception as exc:  # the one-core figure above is the baseline; this one is an extra
            base["all_cores"] = {"error": str(exc)[:200]}
    return base, st["visits"], rays


def device_walk_counts(flat, dev_index, width, height, trace_mode):
    """One frame through the statistics build of the library (raytracer_amd/libmirt_stats.so: same kernels with event
    counters, far slower, never timed): what the device walk really does."""
    import ctypes as C
    from raytracer_amd import _capi
    from raytracer_amd.device import DeviceScene
    path = os.path.join(ROOT, "raytracer_amd", "libmirt_stats.so")
    if not os.path.exists(path):
        return None
    lib = _capi.load_library(path)

    def q(s, k):
        v = C.c_uint64()
        return int(v.value) if lib.mirt_scene_query(s.h, k, C.byref(v)) == 0 else 0
    with DeviceScene(flat, dev_index, lib=lib) as s:
        for k in (101, 103, 105, 116, 118, 119):
            q(s, k)  # read and clear
        _, st = s.render(width, height, trace_mode=trace_mode)
        c = {"node_steps": q(s, 101), "primitive_tests": q(s, 103), "membership_checks": q(s, 105), "triangle_fallbacks": q(s, 116),
             "stack_pushes": q(s, 117), "stack_pushes_overflow": q(s, 118), "membership_boxes": q(s, 119)}
        c["hits"] = (st["shadow_rays"] // max(1, flat.lights.shape[0])) if flat.lights.shape[0] else st["primary_hits"]
    return c


def newest_profile(name):
    """profiles/rNN/<name> of the latest round that holds it, or None."""
    import glob
    found = sorted(glob.glob(os.path.join(ROOT, "profiles", "r[0-9][0-9]", name)))
    return found[-1] if found else None


def roofline(scene, flat, args, width, height, trace_mode, rays, vis, srays, kern, kern_alone, step_s, clock_hz, n_simds):
    """Roofline of the dominant kernel (the launches that walk the scene; with k_frame: one launch = one frame).

    Which roof: the scenes that live in the caches (everything but the 10 M-triangle grid) fetch a small fraction of the HBM
    peak, so their bound is VALU ISSUE: achieved = wave-level VALU instructions of a frame (SQ_INSTS_VALU, rocprofv3 --pmc
    pass over the same frame, profiles/rNN/pmc_<workload>.json) x 2 issue cycles (a wave64 instruction on a 32-lane SIMD)
    / ms_per_step, against peak = SIMDs x the shader clock MEASURED under this run's load (mirt_clock_probe: shader-cycle
    counter against the 100 MHz counter while the frames are in flight).  Two cycles is the floor per instruction (packed,
    transcendental and double-precision ones take more), so `frac` is a lower bound of the issue time.
    The grid reads memory: bound hbm, achieved = HBM bytes of the frame's launches from the counters (FETCH_SIZE x 2 for
    gfx950 + WRITE_SIZE) / their summed duration for a frame alone (HIP events on the launch stream).
    Side figures, labelled: `hbm` = counter bytes per frame over ms_per_step against the 8 TB/s peak; `hbm_demand` = the
    bytes the lanes ASK the memory system for (statistics build: 64 B per node step, 64 B per triangle test, 144 B per sphere
    test, 32 B per fallback to the reference's inside test, 16 + 32 B per leaf box of a membership check, 8 B per stack entry
    beyond LDS; per hit 48 B of record per set-up and light, 16-B light sums, 64 B of material; 3 B per pixel cleared and
    written) -- served mostly by L1/L2/MALL, NOT a bound; `reference_walk` = what the reference's exhaustive walk would need."""
    nl = int(flat.lights.shape[0])
    tri = flat.tri_verts.shape[0] >= flat.spheres.shape[0]
    walk_kinds = ("primary", "trace", "shade", "frame")
    ms_alone = sum(kern_alone[k]["ms_per_frame"] for k in walk_kinds if k in kern_alone)
    ms_pipe = sum(kern[k]["ms_per_frame"] for k in walk_kinds if k in kern)
    n_launch = sum(kern_alone[k]["launches_per_frame"] for k in walk_kinds if k in kern_alone)
    pmc = None
    ppath = newest_profile("pmc_%s.json" % args.workload)
    if ppath and (width, height) == WORKLOADS[args.workload][1]:
        pmc = json.load(open(ppath))
    hbm_bound = args.workload == "grid"
    out = {"kernel": "%d launch(es) per frame that walk the scene (%s)" % (n_launch, ", ".join(k for k in walk_kinds if k in kern_alone)),
           "launch_ms_alone": ms_alone, "launch_ms_in_pipeline": ms_pipe,
           "shader_clock_hz": clock_hz, "simds": n_simds,
           "clock_note": "shader clock measured in this run while frames were in flight (mirt_clock_probe: s_memtime ticks / s_memrealtime ticks x 100 MHz)"}
    traffic = pmc.get("hbm_bytes_per_frame") if pmc else None
    out["traffic"] = traffic
    if pmc:
        out["counters_from"] = os.path.relpath(ppath, ROOT)
        out["traffic_note"] = "HBM bytes of one frame: rocprofv3 --pmc FETCH_SIZE (x2 on gfx950: it tallies 128-B requests at 64 B) + WRITE_SIZE, KB x 1024, separate passes"
    if hbm_bound:
        out.update({"bound": "hbm", "peak": HBM_PEAK_GBS, "unit": "GB/s"})
        if traffic:
            out["achieved"] = traffic / (ms_alone * 1e-3) / 1e9
            out["frac"] = out["achieved"] / HBM_PEAK_GBS
        else:
            out.update({"achieved": None, "frac": None, "note": "no counter pass for this workload under profiles/"})
    else:
        peak = n_simds * clock_hz / 1e9 if clock_hz else None
        out.update({"bound": "valu issue", "peak": peak, "unit": "Gcycle/s"})
        if pmc and pmc.get("valu_insts_per_frame") and peak:
            cyc = 2.0
            out["achieved"] = pmc["valu_insts_per_frame"] * cyc / step_s / 1e9
            out["frac"] = out["achieved"] / peak
            out["frac_alone"] = pmc["valu_insts_per_frame"] * cyc / (ms_alone * 1e-3) / 1e9 / peak
            out["valu_insts_per_frame"] = pmc["valu_insts_per_frame"]
            out["cycles_per_inst"] = cyc
            if pmc.get("valu_active_quad_cycles"):
                # how long an instruction occupies ITS wave (SQ_ACTIVE_INST_VALU counts per wave, in quad-cycles): about 4
                # cycles, while the SIMD starts another wave's instruction every 2 -- waves overlap, so this is no busy figure
                out["valu_wave_cycles_per_inst"] = pmc["valu_active_quad_cycles"] * 4.0 / pmc["valu_insts_per_frame"]
            out["lane_utilisation"] = pmc.get("valu_lane_util")
            out["wait_frac"] = pmc.get("wait_frac")
        else:
            out.update({"achieved": None, "frac": None, "note": "no counter pass for this workload under profiles/ (or no clock reading)"})
    if traffic:
        out["hbm"] = {"bytes_per_frame": traffic, "GBs_per_step": traffic / step_s / 1e9, "utilisation_per_step": traffic / step_s / 1e9 / HBM_PEAK_GBS,
                      "GBs_alone": traffic / (ms_alone * 1e-3) / 1e9, "l2_hit": pmc.get("l2_hit")}
        if pmc.get("fetch_bytes_per_frame_x2_corrected"):
            # Beyond the caches the memory system is bound by read REQUESTS, not bytes: independent random reads of 64-byte and
            # of 128-byte records both run at one request per record and 56 G requests/s (exp/ubench/random_gather.hip, counters
            # beside the timings in profiles/r03/random_gather.txt), and FETCH_SIZE tallies every request at 64 B.
            req = pmc["fetch_bytes_per_frame_x2_corrected"] / 128.0
            out["hbm"]["read_requests"] = {"per_frame": req, "G_per_s_per_step": req / step_s / 1e9, "G_per_s_alone": req / (ms_alone * 1e-3) / 1e9,
                                           "ceiling_G_per_s": RANDOM_READ_REQUESTS_G, "frac_per_step": req / step_s / 1e9 / RANDOM_READ_REQUESTS_G,
                                           "note": "memory-side read requests (FETCH_SIZE / 64 B) against the measured rate of independent random reads"}
    c = device_walk_counts(flat, scene.device, width, height, trace_mode)
    if c:
        hits = c["hits"]
        b = (64.0 * c["node_steps"] + (64.0 if tri else 144.0) * c["primitive_tests"] + 32.0 * c["triangle_fallbacks"]
             + 16.0 * c["membership_checks"] + 32.0 * c["membership_boxes"] + 8.0 * c["stack_pushes_overflow"]
             + hits * (48.0 + 16.0 + 64.0 + nl * (48.0 + 48.0 + 32.0) + 48.0 + 16.0 + 16.0)
             + 3.0 * width * height + 3.0 * float(rays[0]))
        out["hbm_demand"] = {"note": "bytes the lanes ask the memory system for (served mostly by the caches): a demand figure, not a bound",
                             "bytes_per_frame": b, "GBs_alone": b / (ms_alone * 1e-3) / 1e9, "vs_hbm_peak_alone": b / (ms_alone * 1e-3) / 1e9 / HBM_PEAK_GBS,
                             "traffic_over_demand": (traffic / b) if traffic else None, "device_walk_counts": c,
                             "per_ray": {"node_steps": c["node_steps"] / float(rays[0] + rays[3] + rays[4]), "primitive_tests": c["primitive_tests"] / float(rays[0] + rays[3] + rays[4])}}
    # the reference's exhaustive walk priced in bytes and flops, from the oracle's visit counters on the CPU sample
    per_ray = []
    for k in range(3):
        n = max(int(srays[k]), 1)
        per_ray.append((float(vis[k][0]) * NODE_BYTES + float(vis[k][2]) * (TRI_BYTES if tri else SPHERE_BYTES)) / n + RAY_STATE_BYTES)
    ref_bytes = float(rays[0]) * per_ray[0] + float(rays[1]) * per_ray[1] + float(rays[2]) * per_ray[2] + width * height * 3
    prim_flops = TRI_FLOPS if tri else SPHERE_FLOPS
    ref_flops = sum(float(rays[k]) * (float(vis[k][0]) * NODE_FLOPS + float(vis[k][2]) * prim_flops) / max(int(srays[k]), 1) for k in range(3))
    out["reference_walk"] = {"note": "side figure: the bytes and flops the reference's exhaustive tree walk would need for this frame; the device walk is ordered, culled and de-duplicated, so a rate against these exceeds what the device does",
                             "bytes_per_ray": {"primary": per_ray[0], "shadow": per_ray[1], "reflection": per_ray[2]}, "bytes_per_frame": ref_bytes, "flops_per_frame": ref_flops,
                             "GBs_per_step": ref_bytes / step_s / 1e9, "TFLOPs_per_step": ref_flops / step_s / 1e12, "fp32_vector_peak_TFLOPs": FP32_PEAK_TFLOPS}
    return out


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=20)
    ap.add_argument("--warmup", type=int, default=3)
    ap.add_argument("--width", type=int, default=0)
    ap.add_argument("--height", type=int, default=0)
    ap.add_argument("--trace-mode", choices=["exact", "fast"], default="exact")
    ap.add_argument("--backend", default="nccl", help="torch.distributed backend (gloo = rehearsal with ranks sharing a GPU)")
    ap.add_argument("--pipeline", type=int, default=0, help="frames in flight (each on its own HIP stream and workspace)")
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--cpu-row-step", type=int, default=0, help="CPU baseline sample: every n-th row (0 = 16, or 128 for the grid)")
    ap.add_argument("--workload", choices=sorted(WORKLOADS), default="dragon")
    args = ap.parse_args()

    import torch
    from raytracer_amd import _capi, multigpu, workloads
    from raytracer_amd.device import DeviceScene
    from raytracer_amd.scene import FlatScene

    rank = int(os.environ.get("RANK", "0"))
    world = int(os.environ.get("WORLD_SIZE", "1"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    if world != args.gpus:
        raise SystemExit("--gpus %d but WORLD_SIZE=%d (launch N>1 with torch.distributed.run)" % (args.gpus, world))
    ndev = torch.cuda.device_count()
    if ndev == 0:
        raise SystemExit("no GPU visible: the trace path has no CPU fallback")
    dev_index = local_rank % ndev
    torch.cuda.set_device(dev_index)
    dist = None
    if world > 1:
        import torch.distributed as dist
        dist.init_process_group(backend=args.backend, rank=rank, world_size=world)

    gen, (w1, h1), describe = WORKLOADS[args.workload]
    # the same frame for every N (strong scaling of the configuration the metric is quoted on); the 7680x4320 frame of
    # multi-GPU configuration is --workload dragon8k, with an N = 1 run of its own
    width = args.width or w1
    height = args.height or h1
    trace_mode = _capi.TRACE_FAST if args.trace_mode == "fast" else _capi.TRACE_EXACT
    # scene text -> the library's own parser and Partition build (mirt_host_scene_parse) -> flattened scene
    scene_text = getattr(workloads, gen)(width, height)
    flat = FlatScene.from_text(scene_text)
    scene = DeviceScene(flat, dev_index)
    params = multigpu.rank_params(width, height, rank, world, trace_mode=trace_mode)
    nbytes = multigpu.per_rank_bytes(width, height, world) if world > 1 else width * height * 3
    # frames are issued round-robin on `pipeline` streams; the library keeps one workspace per stream, so the short
    # latency-bound deep-bounce launches of one frame overlap the bulk of the next
    # (several GPUs: a rank's share of the frame is small and its launches are short; 16 in flight -- one per hardware queue --
    # keep a GPU busy where 8 leave a quarter of it idle: a rank of eight 0.139 against 0.165 ms per frame)
    depth = args.pipeline if args.pipeline > 0 else (FRAMES_IN_FLIGHT if world == 1 else 16)
    streams = [torch.cuda.Stream() for _ in range(depth)]
    locals_ = [torch.empty(nbytes, dtype=torch.uint8, device="cuda") for _ in range(depth)]
    images = [torch.empty(width * height * 3, dtype=torch.uint8, device="cuda") for _ in range(depth)] if (world > 1 and rank == 0) else None
    gloo = world > 1 and args.backend == "gloo"
    # rank 0's landing area of the gather, one per stream, allocated once
    gathered = [torch.empty((world, nbytes), dtype=torch.uint8, device="cpu" if gloo else "cuda") for _ in range(depth)] if (world > 1 and rank == 0) else None
    frame_no = [0]

    def step():
        k = frame_no[0] % depth
        frame_no[0] += 1
        st, local = streams[k], locals_[k]
        with torch.cuda.stream(st):
            scene.render_device(params, local.data_ptr(), nbytes, st.cuda_stream)
            if world > 1:
                src = local.cpu() if gloo else local
                got = multigpu.gather_to_root(src, world, rank, dist, out=gathered[k] if gathered else None)
                if rank == 0:
                    got = got.cuda() if gloo else got
                    scene.assemble_tiles(params, world, got.data_ptr(), nbytes, images[k].data_ptr(), st.cuda_stream)

    def fence():
        torch.cuda.synchronize()
        if world > 1:
            dist.barrier()
        torch.cuda.synchronize()

    # the caller keeps `depth` frames in flight and says so: without it the first frames after every fence are launched as
    # if they had the GPU to themselves (they find nothing in flight), and the frames behind them wait
    scene.set_frames_in_flight(depth)
    # set-up, like the scene upload: the library allocates a stream's workspace (2.6 GB at 4K) at the stream's first frame
    for _ in range(depth):
        step()
    fence()
    for _ in range(args.warmup):
        step()
    fence()
    scene.set_profiling(True)  # HIP events around every launch, recorded on the launch stream itself
    t0 = time.perf_counter()
    for _ in range(args.steps):
        step()
    fence()
    elapsed = time.perf_counter() - t0
    # Outside the timed region, the same pipeline once more and longer: the steady-state rate (the K timed steps include
    # the start and the end of their burst), with a one-wave probe beside it that reads the shader clock the GPU holds
    # under this very load (the roofline is priced in cycles of that clock).
    n_steady = 80 if elapsed / args.steps < 3e-3 else 20
    clock_hz = None
    steady_s = None
    if world == 1:
        try:
            scene.clock_probe_start(max(1000, int(0.7 * n_steady * elapsed / args.steps * 1e6)))
        except Exception:
            pass
        t1 = time.perf_counter()
        for _ in range(n_steady):
            step()
        fence()
        steady_s = time.perf_counter() - t1
        try:
            clock_hz = scene.clock_probe_read()
        except Exception:
            clock_hz = None
    stream = streams[(frame_no[0] - 1) % depth].cuda_stream
    local = locals_[(frame_no[0] - 1) % depth]
    st = scene.stats(stream)
    scene.set_frames_in_flight(1)  # from here on frames are rendered one at a time
    # Self-check of the timed mode: the last frame of EVERY stream (rendered with the other streams' frames in flight)
    # must equal a frame rendered alone on a stream of its own, byte for byte; no figure is printed otherwise.
    check_stream = torch.cuda.Stream()
    check = torch.empty(nbytes, dtype=torch.uint8, device="cuda")
    with torch.cuda.stream(check_stream):
        scene.render_device(params, check.data_ptr(), nbytes, check_stream.cuda_stream)
    torch.cuda.synchronize()
    bad = [k for k in range(depth) if not torch.equal(locals_[k], check)]
    if bad:
        raise SystemExit("bench self-check FAILED: frames of streams %s differ from the frame rendered alone" % bad)
    if not bool(check.any()):
        raise SystemExit("bench self-check FAILED: the frame is all black")
    if world > 1 and rank == 0:
        # several GPUs: the image assembled from every rank's tiles must equal the whole image rendered by one GPU
        whole = torch.empty(width * height * 3, dtype=torch.uint8, device="cuda")
        with torch.cuda.stream(check_stream):
            scene.render_device(multigpu.rank_params(width, height, 0, 1, trace_mode=trace_mode), whole.data_ptr(), whole.numel(), check_stream.cuda_stream)
        torch.cuda.synchronize()
        if not torch.equal(images[(frame_no[0] - 1) % depth], whole):
            raise SystemExit("bench self-check FAILED: the image assembled from %d ranks differs from the image rendered by one GPU" % world)
    scene.release_stream(check_stream.cuda_stream)
    # counted as the reference spawns them; the last two entries: shadow and reflection rays the device really walked
    # (a shadow query whose light term is multiplied by an exact zero is answered without a walk)
    rays_local = np.array([st["primary_rays"], st["shadow_rays"], st["reflection_rays"], st["shadow_rays_traced"], st["reflection_rays_traced"]], dtype=np.int64)

    def collect(frames):
        acc = {}
        for lt in frames:
            for name, ms in lt:
                acc.setdefault(name, []).append(ms)
        nf = max(1, len(frames))
        return {k: {"launches_per_frame": len(v) // nf, "ms_per_frame": sum(v) / nf, "avg_launch_ms": sum(v) / len(v)} for k, v in acc.items()}

    # (a) launch durations INSIDE the timed region: the last frame of every stream (frames overlap, so a launch shares the GPU)
    used = [streams[(frame_no[0] - 1 - i) % depth] for i in range(min(depth, args.steps))]  # streams of the last timed frames
    kern = collect([scene.launch_times(s_.cuda_stream) for s_ in used])
    # (b) frames ALONE, outside the timed region: one at a time, nothing else on the GPU.  Launch durations from the same HIP
    # events; frame_alone_ms = first clear to last launch of the frame (the library's own pair of events on the stream)
    alone_lt, alone_ms = [], []
    for _ in range(7):
        scene.render_device(params, local.data_ptr(), nbytes, stream)
        alone_lt.append(scene.launch_times(stream))
        alone_ms.append(scene.stats(stream)["device_ms"])
    kern_alone = collect(alone_lt)
    frame_alone_ms = float(np.median(alone_ms))
    scene.set_profiling(False)
    # (c) the boundary's own synchronous call: mirt_render into host memory, what replaces main.cpp:137-163 as it stands
    sync_host_ms = None
    if world == 1:
        host_img = np.zeros((height, width, 3), np.uint8)
        scene.render(width, height, trace_mode=trace_mode, out=host_img)
        ts = []
        for _ in range(5):
            t1 = time.perf_counter()
            scene.render(width, height, trace_mode=trace_mode, out=host_img)
            ts.append((time.perf_counter() - t1) * 1e3)
        sync_host_ms = float(np.median(ts))
        if not np.array_equal(host_img.reshape(-1), check.cpu().numpy()):
            raise SystemExit("bench self-check FAILED: mirt_render's host image differs from the device frame")

    if world > 1:
        t = torch.tensor([elapsed], dtype=torch.float64, device="cpu" if gloo else "cuda")
        dist.all_reduce(t, op=dist.ReduceOp.MAX)
        elapsed = float(t.item())
        r = torch.tensor(rays_local, device="cpu" if gloo else "cuda")
        dist.all_reduce(r, op=dist.ReduceOp.SUM)
        rays = r.cpu().numpy()
    else:
        rays = rays_local

    if rank == 0:
        total_rays = int(rays[:3].sum())
        ms_per_step = elapsed / args.steps * 1e3
        value = total_rays * args.steps / elapsed / 1e6
        out = {
            "metric": "Mrays/s (primary+shadow+reflection), %s %dx%d maxdepth %d" % (args.workload.capitalize(), width, height, flat.max_depth),
            "value": value, "unit": "Mrays/s", "n_gpus": world, "steps": args.steps, "warmup": args.warmup,
            "ms_per_step": ms_per_step, "higher_is_better": True, "scaling": "strong" if world > 1 else "weak",
            "vs_baseline": None, "dtype": "f32", "data": "synthetic",
            "config": {
                "workload": (describe % (width, height)) + " diffuse+specular+shadows+reflections",
                "trace_mode": args.trace_mode, "frames_in_flight": depth, "hw_queues": int(os.environ["GPU_MAX_HW_QUEUES"]),
                "parallelism": "single GPU, whole image" if world == 1 else "interleaved %dx%d tiles over %d GPUs, scene replicated, %s gather of packed tiles to rank 0" % (multigpu.TILE_W, multigpu.TILE_H, world, "RCCL" if not gloo else "gloo (rehearsal)"),
                "scene_source": "scene text parsed and Partition tree built by the library's host loader in %.2f s (%d nodes, %d leaf entries)" % (flat.parse_seconds + flat.build_seconds, flat.nodes.shape[0], flat.leaf_prims.shape[0]),
            },
            "rays_per_frame": {"primary": int(rays[0]), "shadow": int(rays[1]), "reflection": int(rays[2])},
            "rays_walked_per_frame": {"primary": int(rays[0]), "shadow": int(rays[3]), "reflection": int(rays[4])},
            "self_check": "last frame of each of the %d streams byte-equal to a frame rendered alone" % depth + ("; image assembled from %d ranks byte-equal to the whole image rendered by one GPU" % world if world > 1 else ""),
            "steady_state": None if steady_s is None else {"frames": n_steady, "value": total_rays * n_steady / steady_s / 1e6, "ms_per_step": steady_s / n_steady * 1e3,
                                                           "note": "the same pipeline over more frames, outside the timed region: start and end of the burst weigh less"},
            "shader_clock_hz": clock_hz,        # measured beside the steady-state frames (mirt_clock_probe)
            "frame_alone_ms": frame_alone_ms,   # one frame with nothing else on the GPU: clears + every launch
            "sync_host_ms": sync_host_ms,       # mirt_render: the same plus the copy into the caller's host buffer, host clock
            "kernels": kern,                    # launch durations inside the timed region (launches of frames in flight share the GPU)
            "kernels_alone": kern_alone,        # the same launches of a frame rendered alone
        }
        if world == 1 and not args.no_cpu_baseline:
            row_step = args.cpu_row_step or (128 if args.workload == "grid" else 16)
            # the reference needs minutes to load the 10 M triangle grid: that workload is timed on the C restatement
            base, vis, srays = cpu_baseline_and_visits(flat, scene_text, width, height, row_step, use_reference=args.workload != "grid")
            out["cpu_baseline"] = base
            n_simds = 4 * torch.cuda.get_device_properties(dev_index).multi_processor_count
            out["roofline"] = roofline(scene, flat, args, width, height, trace_mode, rays, vis, srays, kern, kern_alone, elapsed / args.steps, clock_hz, n_simds)
        print(json.dumps(out), flush=True)
    scene.close()
    if world > 1:
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
