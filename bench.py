#!/usr/bin/env python3
"""bench.py — frames/sec of the MI355X rasterizer on BASELINE.json's configs.

One step = one FRAME of the hot path for a model already resident in HBM (SURVEY.md
section 8d): initialise the framebuffers (z = 1e6, colour = normal = 0), project all
triangles, rasterize — `AdvancedPixelBufferFiller.render_model` on cleared buffers.
Default workload: T-Rex.obj at 1024x1024, fov 45 (configs[1], the README benchmark).

  python bench.py [--gpus N] [--steps K] [--warmup W] [--workload NAME]

N > 1 (launched by torch.distributed.run, one rank per GPU), two modes:
  --mode strips (default)  north_star's layout: ONE frame per step, sharded into row strips, the
                           finished strips exchanged with RCCL: "strong" scaling.  Default workload
                           for N > 1 is T-Rex 8192x8192 (configs[3]).  --exchange picks what is
                           gathered (planes 28 B/pixel, color 12, present 3), --chunks N gathers
                           sub-strips on a second stream while the next one is rasterized.  The same
                           JSON line carries the step time without the exchange;
  --mode frames            every rank renders its own full frames, no collective in the data
                           path: "weak" scaling, value = frames of all ranks per second.

Rank 0 prints ONE JSON line.  `roofline` prices the kernel that rasterizes the timed region's
frames — `k_frame` (raster pass + the next frame's binning wavefronts in one launch) when the
swap chain runs with look-ahead, else `k_raster` — against HBM bandwidth with the algorithmic
bytes of SURVEY.md section 8d (108 B per triangle read once + 28 B per pixel written once); its
launch duration is measured with HIP events on the frame's own stream in a pass of its own
(frames one after another on ONE stream, so that a launch's duration is its own), and is never
taken shorter than the back-to-back frame time of that pass or the committed rocprofv3 average
(profiles/kernel_avg.json).  `cpu_baseline` times the CPU oracle in its Version-C shape (OpenMP
dynamic loop + per-pixel locks) on this host — a reported baseline, not the target; `api_call`
times the reference's own calls (`render_model(model)` from numpy arrays, the getters,
`Renderer.render`) through the drop-in classes, PCIe included, beside the oracle's same calls.
"""
import argparse
import json
import os
import sys
import time

# The CPU baseline's OpenMP threads stay where they start, one per physical core, neighbours first
# (a 256-CPU host otherwise migrates sixteen threads across its CCDs between frames: 16 threads were
# 1.9x one thread in round 3, and the figure moved 30 % from box to box).  libgomp reads OMP_PROC_BIND /
# OMP_PLACES once, when it is loaded, and then binds the thread that loaded it to the first place — so
# the baseline runs in a CHILD process of its own that gets the two variables (the oracle needs no
# GPU), and this process, whose main thread submits every frame, is never bound (under
# torch.distributed.run every rank would otherwise sit on the same core).
OMP_BINDING = {"OMP_PROC_BIND": "close", "OMP_PLACES": "cores"}
try:
    _ALLOWED_CPUS = sorted(os.sched_getaffinity(0))
except AttributeError:
    _ALLOWED_CPUS = list(range(os.cpu_count() or 1))

import numpy as np

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)

HBM_PEAK_GBPS = 8000.0      # MI355X_MICROARCH.md: HBM3E 8.0 TB/s spec (6.29 TB/s measured copy)


def algorithmic_bytes(T, rows, W):
    return 108 * T + 28 * rows * W


def csrc_sha16():
    """Fingerprint of what the kernels are built from (cython3dmodelrenderer_amd/_build.py)."""
    from cython3dmodelrenderer_amd import _build
    return _build.source_sha16()


def _profiled(name, workload):
    """The workload's entry of a committed profile summary (profiles/<name>) — only if that summary was
    made from the kernel sources this run is built from (its "csrc_sha16", stamped at profiling time
    by scripts/make_*_json.py): a figure measured on other kernels says nothing about this run."""
    try:
        with open(os.path.join(ROOT, "profiles", name)) as fh:
            doc = json.load(fh)
    except Exception:
        return {}
    if doc.get("csrc_sha16") != csrc_sha16():
        return {}
    return doc.get(workload, {})


def profile_key(kernel):
    """'k_frame<32,true,1>' -> 'k_frame_owners': how scripts/summarize_prof.py names the raster kernels."""
    base = "k_frame" if kernel.startswith("k_frame") else "k_raster"
    return base + ("_owners" if kernel.endswith(",1>") else "")


def load_traffic(workload, kernel):
    """HBM bytes per launch of THIS kernel from the committed PMC profile, if one exists for these sources."""
    return _profiled("traffic.json", workload).get(profile_key(kernel), {}).get("hbm_bytes_per_launch")


def load_rocprof_avg_ms(workload, kernel):
    """Average launch duration (ms) of `kernel` in the committed rocprofv3 kernel trace of this
    workload's bench command (profiles/kernel_avg.json, made by scripts/summarize_prof.py), if any."""
    ns = _profiled("kernel_avg.json", workload).get(kernel)
    return None if ns is None else float(ns) * 1e-6


def cpu_model():
    try:
        with open("/proc/cpuinfo") as fh:
            for line in fh:
                if line.startswith("model name"):
                    return line.split(":", 1)[1].strip()
    except OSError:
        pass
    return "unknown"


def host_topology():
    """CPUs this process may run on and the physical cores behind them (/proc/cpuinfo)."""
    allowed = _ALLOWED_CPUS
    cores = set()
    try:
        cpu = phys = core = None
        with open("/proc/cpuinfo") as fh:
            for line in fh:
                k, _, v = line.partition(":")
                k = k.strip()
                if k == "processor":
                    cpu, phys, core = int(v), None, None
                elif k == "physical id":
                    phys = int(v)
                elif k == "core id":
                    core = int(v)
                    if cpu in allowed:
                        cores.add((phys, core))
    except (OSError, ValueError):
        pass
    return len(allowed), (len(cores) or None)


def cgroup_cpu_quota():
    """CPUs' worth of time this process's cgroup may use (cpu.max), or None if unlimited / unknown."""
    try:
        quota, period = open("/sys/fs/cgroup/cpu.max").read().split()[:2]           # cgroup v2
        if quota != "max":
            return max(1, int(float(quota) / float(period)))
    except (OSError, ValueError):
        pass
    try:
        quota = int(open("/sys/fs/cgroup/cpu/cpu.cfs_quota_us").read())             # cgroup v1
        period = int(open("/sys/fs/cgroup/cpu/cpu.cfs_period_us").read())
        if quota > 0 and period > 0:
            return max(1, quota // period)
    except (OSError, ValueError):
        pass
    return None


def _cpu_child(spec):
    """Runs in the CPU-only child process (bench.py --cpu-child): the Version-C-shaped CPU oracle on this
    host's cores, same frame definition — thread counts 1, 8, 16 (the README's
    columns, /root/reference/README.md:76) and one thread per physical core the process may use
    (capped by the cgroup's CPU quota: more threads than that only measure the throttle); best of
    `passes` passes each after a warm-up frame, threads bound to cores.  With `api` also the oracle's
    side of the api_call block.  Prints one JSON object."""
    from cython3dmodelrenderer_amd import scenes
    from oracle import oracle as O
    tri, col, nrm, (H, W), fov = scenes.scene(spec["workload"], synth_T=spec["synth_T"])
    if spec["max_triangles"] >= 0:
        tri, col, nrm = (a[:spec["max_triangles"]] for a in (tri, col, nrm))
    budget_s, passes = spec["budget_s"], spec["passes"]
    ncpu, ncores = host_topology()
    quota = cgroup_cpu_quota()

    def measure(threads, budget):
        f = O.OracleFiller(H, W, fov=fov, n_threads=threads, mode="omp")

        def frame():
            f.clear()
            f.render_arrays(tri, col, nrm)

        t0 = time.perf_counter()
        frame()                                  # warm-up, also sizes the sample
        one = time.perf_counter() - t0
        n = int(max(1, min(100, budget / passes / max(one, 1e-4))))
        best = None
        for _ in range(passes):
            t0 = time.perf_counter()
            for _ in range(n):
                frame()
            dt = (time.perf_counter() - t0) / n
            best = dt if best is None or dt < best else best
        return best, n

    counts = [t for t in (1, 8, 16) if t <= ncpu]
    every = min(ncores or ncpu, quota or ncpu)   # one thread per physical core, within the quota
    if every not in counts and every > 1:
        counts.append(every)
    head = 16 if 16 in counts else counts[-1]
    by_threads, frames = {}, {}
    for t in counts:
        dt, n = measure(t, budget_s / len(counts))
        by_threads[str(t)] = 1.0 / dt
        frames[str(t)] = n
    dt = 1.0 / by_threads[str(head)]
    best_t = max(by_threads, key=lambda k: by_threads[k])
    out = {"value": 1.0 / dt, "unit": "frames/s", "cores": head, "kind": "port",
           "ms_per_frame": dt * 1e3,
           "best": {"threads": int(best_t), "frames_per_s": by_threads[best_t]},
           "host_cpus": ncpu, "host_physical_cores": ncores, "cgroup_cpu_quota": quota, "cpu_model": cpu_model(),
           "frames_per_s_by_threads": by_threads,
           "speedup_over_one_thread": {k: v / by_threads["1"] for k, v in by_threads.items()} if "1" in by_threads else None,
           "omp": {k: os.environ.get(k) for k in OMP_BINDING},
           "process": "CPU-only child of bench.py (the OpenMP binding is confined to it)",
           "sample": f"best of {passes} passes of {frames[str(head)]} full frames of the same workload (clear + project + "
                     f"raster) per thread count, OpenMP dynamic schedule + per-pixel locks, threads bound to cores"}
    res = {"cpu_baseline": out}
    if spec.get("api"):
        # the oracle's side of the api_call block: the reference's own calls on min(16, ncpu) threads
        from cython3dmodelrenderer_amd.illumination import GuroIllumination
        m = _Model(tri, col, nrm)
        light = GuroIllumination([0, 0, 1])
        threads = min(16, ncpu)
        of = O.OracleFiller(H, W, fov=fov, n_threads=threads, mode="omp")

        def cpu_render():
            of.render_model(m)
            light.draw_illumination(of.color_buffer, of.normals_buffer)
            return of.color_buffer
        res["api_cpu_oracle"] = {"threads": threads,
                                 "render_model_ms": _timed(lambda: of.render_model(m), lambda: None, 1.5),
                                 "renderer_render_ms": _timed(cpu_render, lambda: None, 1.5)}
    print(json.dumps(res), flush=True)


def cpu_baseline(workload, synth_T, max_triangles, api=False, budget_s=16.0, passes=3):
    """cpu_baseline (and the oracle's half of api_call) from the CPU-only child process."""
    import subprocess
    spec = {"workload": workload, "synth_T": synth_T, "max_triangles": max_triangles, "api": bool(api),
            "budget_s": budget_s, "passes": passes}
    env = dict(os.environ)
    for k, v in OMP_BINDING.items():
        env.setdefault(k, v)
    r = subprocess.run([sys.executable, os.path.abspath(__file__), "--cpu-child", json.dumps(spec)],
                       env=env, stdout=subprocess.PIPE, check=True, text=True)
    return json.loads(r.stdout.strip().splitlines()[-1])


def _timed(call, sync, budget):
    """ms per call: warm-up, then the calls in ten chunks, each closed by `sync`; the MEDIAN chunk.  (One
    total over all calls carried one-off stalls of the process — 60-130 ms once, somewhere in a loop of
    200 calls, seen only in this script's fourth Renderer loop and not when every call is timed by itself,
    scripts/fused_probe.py — as +0.3-0.6 ms "per call".)"""
    call(); call(); sync()                  # warm-up: plans, pinned buffers, staging
    t0 = time.perf_counter()
    call(); sync()
    one = time.perf_counter() - t0
    n = int(max(3, min(200, budget / max(one, 1e-5))))
    chunks = min(10, n)
    per = max(1, n // chunks)
    times = []
    for _ in range(chunks):
        t0 = time.perf_counter()
        for _ in range(per):
            call()
        sync()
        times.append((time.perf_counter() - t0) / per * 1e3)
    times.sort()
    return times[len(times) // 2]


class _Model:
    """What the reference's filler reads off a Model (.pyx:94-96)."""

    def __init__(self, tri, col, nrm):
        self._vertices_by_triangles, self._colors_by_triangles, self._normals_by_triangles = tri, col, nrm


def api_calls(tri, col, nrm, H, W, fov, device, budget_s=3.0):
    """The reference's OWN calls through the drop-in classes, model arrays in host numpy memory,
    results in host numpy memory where the reference returns them there — PCIe included:
      render_model(model)                       .pyx:92-104 (upload + K1 + K2, buffers composite)
      render_model(model); get_color_buffer()   .pyx:249 on top
      Renderer.render(model)                    cy/renderer.py:47-49 with GuroIllumination
    (the CPU oracle's same call sequences on min(16, ncpu) threads come from the CPU child process)."""
    import torch
    from cython3dmodelrenderer_amd import Renderer
    from cython3dmodelrenderer_amd.illumination import GuroIllumination
    from cython3dmodelrenderer_amd.pixel_buffer_filler import AdvancedPixelBufferFiller
    m = _Model(tri, col, nrm)
    light = GuroIllumination([0, 0, 1])

    def timed(call, sync, budget=budget_s):
        return _timed(call, sync, budget)

    def dsync():
        torch.cuda.synchronize(device)

    out = {"unit": "ms per call (median of ten chunks of calls, each chunk closed by a device synchronisation)",
           "model_arrays": "host numpy, uploaded by every call (as the reference copies them)"}
    f = AdvancedPixelBufferFiller(H, W, fov=fov, device=device)
    out["render_model_ms"] = timed(lambda: f.render_model(m), dsync)
    # the same call when the caller waits for the GPU after each one (what a caller that reads the
    # buffers right away sees; the figure above is the rate of calls that only enqueue)
    out["render_model_then_wait_ms"] = timed(lambda: (f.render_model(m), dsync()), lambda: None)
    out["render_model_plus_color_ms"] = timed(lambda: (f.render_model(m), f.get_color_buffer()), dsync)
    f3 = AdvancedPixelBufferFiller(H, W, fov=fov, device=device)
    out["render_model_plus_three_buffers_ms"] = timed(
        lambda: (f3.render_model(m), f3.get_color_buffer(), f3.get_normals_buffer(), f3.get_z_buffer()), dsync)
    for name, mode in (("numpy_illumination", False), ("default", None), ("on_device", True), ("fused", "fused")):
        r = Renderer(AdvancedPixelBufferFiller(H, W, fov=fov, device=device), light, None, H, W, on_device=mode)
        out[f"renderer_render_{name}_ms"] = timed(lambda: r.render(m), dsync)
    return out


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=200)
    ap.add_argument("--warmup", type=int, default=20)
    ap.add_argument("--workload", default=None,
                    choices=["cube256", "trex1024", "bunny4096", "trex8192", "synth10m"],
                    help="default: trex1024 on one GPU (BASELINE.json's metric config), trex8192 "
                         "(configs[3], the sharded frame) on several")
    ap.add_argument("--synth-triangles", type=int, default=10_000_000)
    ap.add_argument("--tile", type=int, default=0)
    ap.add_argument("--pipeline-tile", type=int, default=0,
                    help="A/B knob: tile size of the swap chain's plans (default: the chain's own choice, 32)")
    ap.add_argument("--max-triangles", type=int, default=-1,
                    help="experiment knob: keep only the first N triangles (0 = pure clear)")
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--no-api-calls", action="store_true",
                    help="skip the api_call block (the reference's own calls, PCIe included)")
    ap.add_argument("--no-pipeline", action="store_true",
                    help="do not overlap the next frame's bin pass with this frame's raster pass")
    ap.add_argument("--no-gather", action="store_true", help="N>1: skip the RCCL all-gather")
    ap.add_argument("--raster-path", default="auto", choices=["auto", "0", "1"],
                    help="which raster kernel 32-pixel plans use: auto = each plan's own choice from the size classes "
                         "its previous frames counted (the product's default), 0 general, 1 pixel owners only — "
                         "both exact on every tile (A/B knob)")
    ap.add_argument("--pipeline-depth", type=int, default=0,
                    help="frames in flight (swap-chain depth); 0 = the filler's choice")
    ap.add_argument("--lookahead", default="auto", choices=["auto", "on", "off"],
                    help="swap chain: one launch per frame, the raster pass together with the binning pass of "
                         "the slot's next frame (auto: scenes that fit the direct bins)")
    ap.add_argument("--backend", default="nccl", choices=["nccl", "gloo"],
                    help="N>1: 'nccl' (= RCCL, one rank per GPU).  'gloo' is a rehearsal of the "
                         "launch contract on a box with fewer GPUs than ranks: ranks share devices "
                         "(LOCAL_RANK modulo the device count) and host-side collectives carry the "
                         "timings; --mode strips then needs --no-gather")
    ap.add_argument("--mode", default="strips", choices=["frames", "strips"],
                    help="N>1: 'strips' (default) = ONE frame per step, row strips + RCCL exchange of "
                         "the finished strips (north_star's layout; strong scaling); 'frames' = every "
                         "rank renders its own full frame per step, no collective in the data path "
                         "(weak scaling)")
    ap.add_argument("--exchange", default="planes", choices=["planes", "color", "present"],
                    help="strips: what every rank receives — the three planes (28 B/pixel), the colour "
                         "plane (12) or the presented uint8 image (3)")
    ap.add_argument("--project", default="local", choices=["local", "broadcast"],
                    help="strips: who runs K1 — every rank for itself (model replicated, nothing on the "
                         "links) or rank 0, which broadcasts the projected vertices (36 B/triangle to "
                         "every rank; north_star's wording)")
    ap.add_argument("--chunks", type=int, default=1,
                    help="strips: sub-strips per rank, each exchanged on a second stream while the "
                         "next one is rasterized")
    ap.add_argument("--cpu-child", default=None, help=argparse.SUPPRESS)
    args = ap.parse_args()
    if args.cpu_child:
        return _cpu_child(json.loads(args.cpu_child))

    # more hardware queues than the runtime's default of 4, so that four streams of the swap
    # chain plus torch's own do not share one (read by the HIP runtime when it starts)
    os.environ.setdefault("GPU_MAX_HW_QUEUES", "8")
    import torch
    import torch.distributed as dist
    from cython3dmodelrenderer_amd import distributed as D
    from cython3dmodelrenderer_amd import scenes
    from cython3dmodelrenderer_amd.pixel_buffer_filler import AdvancedPixelBufferFiller

    world = int(os.environ.get("WORLD_SIZE", "1"))
    rank = int(os.environ.get("RANK", "0"))
    local = int(os.environ.get("LOCAL_RANK", "0"))
    if world != args.gpus:
        if world == 1 and args.gpus > 1:
            raise SystemExit("launch N>1 with: python -m torch.distributed.run --nproc-per-node N "
                             "--master-addr 127.0.0.1 bench.py --gpus N ...")
        raise SystemExit(f"--gpus {args.gpus} but WORLD_SIZE={world}")
    if args.workload is None:
        args.workload = "trex1024" if world == 1 else "trex8192"
    if args.backend == "gloo":
        local %= max(torch.cuda.device_count(), 1)
    torch.cuda.set_device(local)
    device = torch.device("cuda", local)
    if world > 1:
        os.environ.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")
        if args.backend == "nccl":
            dist.init_process_group("nccl", device_id=device)
        else:
            dist.init_process_group("gloo")      # (strips are then staged through the host)

    tri, col, nrm, (H, W), fov = scenes.scene(args.workload, synth_T=args.synth_triangles)
    if args.max_triangles >= 0:
        tri, col, nrm = tri[:args.max_triangles], col[:args.max_triangles], nrm[:args.max_triangles]
    T = int(tri.shape[0])
    strips = world > 1 and args.mode == "strips"
    rpath = None if args.raster_path == "auto" else int(args.raster_path)
    y0, y1 = D.strip_rows(H, world, rank) if strips else (0, H)
    if strips:
        sr = D.StripRenderer(H, W, rank, world, fov=fov, device=device, tile=args.tile,
                             exchange=args.exchange, chunks=args.chunks,
                             pipeline=not args.no_pipeline, project=args.project)
        filler = sr.filler
    else:
        sr = None
        filler = AdvancedPixelBufferFiller(H, W, fov=fov, device=device, tile=args.tile,
                                           pipeline=not args.no_pipeline,
                                           pipeline_depth=args.pipeline_depth,
                                           lookahead={"auto": None, "on": True, "off": False}[args.lookahead],
                                           raster_path=rpath)

    if args.pipeline_tile:
        filler._pipeline_tile = args.pipeline_tile       # (A/B knob: the chain's own choice otherwise)

    def step(pipelined=True, gather=True):
        if sr is not None:
            sr.render_frame(gather=gather and not args.no_gather)
        else:
            filler.render_frame(pipelined=pipelined)

    # set-up, untimed and not a step: upload the model, size the bin lists, and create the swap
    # chain's plans / streams / framebuffer sets (allocation must not land in the timed region
    # when the driver asks for --warmup 0)
    if sr is not None:
        sr.set_model_arrays(tri, col, nrm)
    else:
        filler.render_arrays(tri, col, nrm, clear=True)
        filler.synchronize()
    if not args.no_pipeline and not (sr is not None and sr.empty):
        step()
        filler.synchronize()
    for _ in range(args.warmup):
        step()
    filler.synchronize()

    def barrier():
        if world > 1:
            dist.barrier()
        torch.cuda.synchronize(device)

    # ---- the timed region: exactly K steps, nothing but the steps -------------------------
    barrier()
    t0 = time.perf_counter()
    for _ in range(args.steps):
        step()
    barrier()
    elapsed = time.perf_counter() - t0
    assert not (filler._pipe is not None and filler._pipe.overflowed(filler)), \
        "bin lists overflowed inside the timed region"
    if filler._pipe is not None:
        filler._pipe.n = 0

    # ---- per-kernel durations: the same K steps again with HIP events on the frame's stream
    # around the binning passes and around the raster kernel (single stream, not pipelined, so
    # each kernel's duration is its own).  Kept out of the timed region
    # above because the three event records cost ~11 us per frame on a ~35 us frame (measured:
    # scripts/hostoverhead.py vs this loop); the events loop's own frame time is reported too.
    elapsed_render_only = None
    if strips:
        # the same K steps without the exchange: what the rasterization of the sharded frame costs
        barrier()
        t3 = time.perf_counter()
        for _ in range(args.steps):
            step(gather=False)
        barrier()
        elapsed_render_only = time.perf_counter() - t3
    if sr is not None and sr.empty:
        raise SystemExit("more ranks than rows")
    step(pipelined=False, gather=False)            # also makes sure the single-stream plan exists
    barrier()
    t2 = time.perf_counter()         # the same K frames one at a time on one stream: frame latency
    for _ in range(args.steps):
        filler.render_frame(pipelined=False)
    barrier()
    elapsed_single = time.perf_counter() - t2
    filler.timing_begin(args.steps)
    barrier()
    t1 = time.perf_counter()
    for _ in range(args.steps):
        filler.render_frame(pipelined=False)
    barrier()
    elapsed_events = time.perf_counter() - t1
    n_timed, bin_ms, raster_ms = filler.timing_end()
    need, cap = filler.bin_usage()
    assert need <= cap, "bin lists overflowed inside the timing pass"

    # ---- the kernel of the timed region.  With look-ahead the swap chain's frames are ONE launch
    # each, k_frame (this frame's raster workgroups + the next frame's binning wavefronts), not the
    # k_setup_wave -> k_raster pair timed above.  Its duration is its own only when launches do not
    # overlap: a chain of depth 1 (same kernel, same arguments, frames one after another on one
    # stream), K frames back to back on the wall clock and K frames with HIP events around each launch.
    lookahead = filler._pipe is not None and filler._pipe.lookahead
    kframe_events_ms = kframe_b2b_ms = None
    if lookahead:
        probe = AdvancedPixelBufferFiller(H, W, fov=fov, device=device, tile=filler._pipe.tile, pipeline=True,
                                          pipeline_depth=1, lookahead=True, raster_path=rpath,
                                          row_strip=(y0, y1) if strips else None)
        probe.render_arrays(tri, col, nrm, clear=True)
        probe.synchronize()
        for _ in range(max(3, args.warmup)):
            probe.render_frame()
        probe.synchronize()
        assert probe._pipe.lookahead
        barrier()
        t4 = time.perf_counter()
        for _ in range(args.steps):
            probe.render_frame()
        probe.join()
        barrier()
        kframe_b2b_ms = (time.perf_counter() - t4) / args.steps * 1e3
        probe._pipe.timing_begin(args.steps)
        for _ in range(args.steps):
            probe.render_frame()
        n_k, kframe_events_ms = probe._pipe.timing_end()
        assert n_k == args.steps and not probe._pipe.overflowed(probe)
        probe_paths = probe.last_raster_paths()
        del probe

    # ---- strips: the SAME frame on ONE GPU, in the same run (rank 0; the others wait at the next
    # collective): the denominator of this line's strong-scaling speedup.  The N = 1 line of bench.py
    # is the headline workload (T-Rex 1024x1024), not this one, so a speedup computed from the two
    # lines' values would compare different frames.
    one_gpu = None
    if strips and rank == 0:
        solo = AdvancedPixelBufferFiller(H, W, fov=fov, device=device, tile=args.tile, pipeline=not args.no_pipeline)
        solo.render_arrays(tri, col, nrm, clear=True)
        solo.synchronize()
        for _ in range(max(2, args.warmup)):
            solo.render_frame()
        solo.synchronize()
        t5 = time.perf_counter()
        for _ in range(args.steps):
            solo.render_frame()
        solo.join()
        torch.cuda.synchronize(device)
        dt1 = (time.perf_counter() - t5) / args.steps
        assert not (solo._pipe is not None and solo._pipe.overflowed(solo))
        one_gpu = {"n_gpus": 1, "value": 1.0 / dt1, "unit": "frames/s", "ms_per_step": dt1 * 1e3,
                   "what": "the whole frame of this workload on rank 0's GPU alone, same process, same K "
                           "(no exchange: the frame is complete in its memory)"}
        del solo

    # ---- strips: and the WEAK-scaling curve of BASELINE.json's metric ("frames/sec, T-Rex.obj 1024x1024 at
    # 1/2/4/8 GPUs") from the same run: every rank renders K full frames of the headline workload on its own GPU,
    # no collective (what --mode frames times), between the same barriers; value = world x K / slowest rank.
    weak = None
    if strips:
        w_tri, w_col, w_nrm, (wH, wW), w_fov = scenes.scene("trex1024")
        wf = AdvancedPixelBufferFiller(wH, wW, fov=w_fov, device=device, pipeline=not args.no_pipeline, raster_path=rpath)
        wf.render_arrays(w_tri, w_col, w_nrm, clear=True)
        wf.synchronize()
        for _ in range(max(3, args.warmup)):
            wf.render_frame()
        wf.synchronize()
        barrier()
        t6 = time.perf_counter()
        for _ in range(args.steps):
            wf.render_frame()
        wf.join()
        barrier()
        weak_elapsed = time.perf_counter() - t6
        assert not (wf._pipe is not None and wf._pipe.overflowed(wf))
        del wf
        weak = weak_elapsed

    # CPUs the thread that submits the frames may run on, NOW (after torch / libgomp are loaded): a
    # runtime that bound it would show here — smallest over the ranks
    try:
        affinity = len(os.sched_getaffinity(0))
    except AttributeError:
        affinity = os.cpu_count() or 1
    if world > 1:
        t = torch.tensor([elapsed, raster_ms, bin_ms, elapsed_render_only or 0.0, kframe_events_ms or 0.0,
                          kframe_b2b_ms or 0.0, -float(affinity), weak or 0.0], dtype=torch.float64,
                         device=device if args.backend == "nccl" else "cpu")
        dist.all_reduce(t, op=dist.ReduceOp.MAX)
        elapsed, raster_ms, bin_ms, ero, ke, kb, naff, weak_max = (float(v) for v in t.cpu())
        if weak is not None:
            weak = weak_max
        affinity = int(-naff)
        if elapsed_render_only is not None:
            elapsed_render_only = ero
        if lookahead:
            kframe_events_ms, kframe_b2b_ms = ke, kb

    if rank == 0:
        frames_per_step = 1 if (world == 1 or strips) else world
        fps = frames_per_step * args.steps / elapsed
        rows = filler.y1 - filler.y0                 # rows the measured launches rasterize: this rank's strip
        abytes = algorithmic_bytes(T, rows, W)       # (ranks are symmetric), its first sub-strip with --chunks
        # Two views of the raster kernel's launch duration:
        #   raster_ms      HIP events right around the launch (second pass).  The event records
        #                  open idle bubbles in which the previous frame's dirty lines drain, so a
        #                  write-heavy launch looks up to ~10 % shorter than it is back to back;
        #   raster_b2b_ms  single-stream frame time (no events inside the loop) minus the event-
        #                  measured bin passes: the launch as it runs in a continuous stream.  This
        #                  is what rocprofv3's AverageNs of the same command shows (profiles/).
        # The roofline uses the LONGER of the two, i.e. never the flattering one.
        single_ms = elapsed_single / args.steps * 1e3
        raster_b2b_ms = max(single_ms - bin_ms, 0.0)
        ts = (filler._pipe.tile if lookahead else filler.tile) or (16 if H * W <= 1024 * 1024 else 32)
        # (32-pixel plans have two raster kernels, both exact; which one these launches were: the plans say)
        paths = filler.last_raster_paths()
        kpath = (probe_paths[-1] if lookahead else paths[0]) if ts == 32 else 0
        if lookahead:
            kernel = f"k_frame<{ts},true,{kpath}>"
            views = {"hip_events_around_each_launch": kframe_events_ms, "frames_back_to_back_on_one_stream": kframe_b2b_ms}
            prof = load_rocprof_avg_ms(args.workload, profile_key(kernel) + "_one_stream") if rows == H else None
        else:
            kernel = f"k_raster<{ts},true,{kpath}>"
            views = {"hip_events_around_each_launch": raster_ms,
                     "single_stream_frame_minus_event_measured_bin_passes": raster_b2b_ms}
            prof = load_rocprof_avg_ms(args.workload, profile_key(kernel)) if rows == H else None
        if prof is not None:
            views["committed_rocprofv3_average"] = prof
        launch_ms = max(v for v in views.values() if v is not None)
        achieved = abytes / (launch_ms * 1e-3) / 1e9 if launch_ms > 0 else 0.0
        out = {
            "metric": "frames/sec", "value": fps, "unit": "frames/s",
            "n_gpus": world, "steps": args.steps, "warmup": args.warmup,
            "ms_per_step": elapsed / args.steps * 1e3,
            "ms_per_step_without_exchange": (elapsed_render_only / args.steps * 1e3
                                             if elapsed_render_only is not None else None),
            "higher_is_better": True,
            # strips: the job is ONE frame however many GPUs share its rows; frames: one per rank
            "scaling": "strong" if strips else "weak",
            "vs_baseline": None,
            "dtype": "f32", "data": "synthetic" if args.workload == "synth10m" else
            "T-Rex/bunny/cube input arrays committed under tests/golden (made from the reference's "
            ".obj assets); no dataset download",
            "config": {"workload": args.workload, "triangles": T, "height": H, "width": W,
                       "fov": fov, "row_strips": world if strips else 1,
                       "multi_gpu": ("single GPU" if world == 1 else
                                     f"row strips + {'gloo (host-staged)' if args.backend == 'gloo' else 'RCCL'} "
                                     f"all-gather, exchange = {args.exchange}, {args.chunks} sub-strip(s) per rank, "
                                     f"projection = {'rank 0, broadcast of the projected vertices' if args.project == 'broadcast' else 'every rank its own (model replicated)'}"
                                     if strips else "independent full frames per rank, no collective"),
                       "tile": filler.tile or "auto",
                       "pipeline_tile": filler._pipe.tile if filler._pipe is not None else None,
                       "raster_path": {"asked": args.raster_path, "last_launch_of_each_plan": paths,
                                       "what": "0 general kernel, 1 pixel owners only; "
                                               "auto: each plan picks by the tile size classes its previous frames counted"},
                       "frame": "clear + project + rasterize, model resident in HBM",
                       "pipelined": (False if args.no_pipeline else
                                     f"swap chain of {filler._pipeline_depth} (GPU_MAX_HW_QUEUES="
                                     f"{os.environ.get('GPU_MAX_HW_QUEUES')}): frames in flight render into "
                                     "separate framebuffer sets on separate streams, each frame complete"
                                     + ("; one launch per frame: its raster pass and, in the same launch, the "
                                        "binning pass of the stream's next frame (every frame is binned exactly "
                                        "once; the first frame of a stream after a synchronisation bins in a "
                                        "launch of its own)"
                                        if filler._pipe is not None and filler._pipe.lookahead else "")),
                       "all_gather": bool(strips and not args.no_gather)},
            "mtris_per_sec": T * fps / 1e6,
            "frame_algorithmic_bytes": algorithmic_bytes(T, H, W),
            "whole_frame_gbps": algorithmic_bytes(T, H, W) * fps / 1e9,
            "kernel_ms": {"binning_passes": bin_ms, "raster": raster_ms,
                          "raster_back_to_back": raster_b2b_ms, "timed_frames": n_timed,
                          "how": "HIP events on the frame's stream, second pass of K steps",
                          "ms_per_step_with_events": elapsed_events / args.steps * 1e3},
            "ms_per_frame_single_stream": elapsed_single / args.steps * 1e3,
            "roofline": {"kernel": kernel, "bound": "hbm", "achieved": achieved,
                         "peak": HBM_PEAK_GBPS, "unit": "GB/s", "frac": achieved / HBM_PEAK_GBPS,
                         "algorithmic_bytes_per_launch": abytes,
                         "avg_launch_ms": launch_ms,
                         "avg_launch_ms_views": views,
                         "avg_launch_ms_how": "the LONGEST of the views: the kernel the timed region's frames "
                                              "run, launches one after another on one stream",
                         # what the overlapped stream of frames achieves with the same launches
                         "overlapped_frames_gbps": abytes * (args.steps / elapsed) / 1e9,
                         # PMC bytes of the committed profile are per WHOLE-frame launch: a strip's
                         # launch was not profiled
                         "traffic": load_traffic(args.workload, kernel) if rows == H else None},
            "bin_entries": {"needed": need, "capacity": cap},
            # the kernels' sources; profiles/*.json figures are quoted only if they were measured on these
            "csrc_sha16": csrc_sha16(),
            "submit_thread_affinity_cpus_min_over_ranks": affinity,
        }
        if strips:
            # what the exchange moves and what it costs: the line explains its own scaling
            step_ms = elapsed / args.steps * 1e3
            bare_ms = elapsed_render_only / args.steps * 1e3
            gathered = not args.no_gather
            out["exchange"] = {
                "kind": args.exchange if gathered else None,
                "exchange_bytes_received_per_rank": D.exchange_bytes_received(args.exchange, H, W, world, 0) if gathered else 0,
                "exchange_ms": step_ms - bare_ms,
                "ms_per_step": step_ms, "ms_per_step_without_exchange": bare_ms,
                "bytes_received_per_rank_by_choice": {k: D.exchange_bytes_received(k, H, W, world, 0) for k in D.EXCHANGES},
                "projection_broadcast_bytes_per_rank": 36 * T if args.project == "broadcast" else 0,
                "expectation": ("DESIGN.md section 5: with exchange = planes every rank receives 28 B/pixel x (N-1)/N "
                                "of the frame over xGMI, several times the time ONE GPU needs to render the whole "
                                "frame — a speed-up below 1x is the predicted outcome of north_star's layout; "
                                "'present' moves 9x less"),
            }
            out["config"]["multi_gpu"] += ("; predicted by DESIGN.md section 5 to be SLOWER than one GPU with exchange = "
                                           "planes (the exchange, not the rasterization, bounds the sharded frame)"
                                           if args.exchange == "planes" and gathered else "")
        if weak is not None:
            out["weak_scaling_frames"] = {
                "metric": "frames/sec", "workload": "trex1024", "n_gpus": world, "scaling": "weak",
                "value": world * args.steps / weak, "ms_per_step": weak / args.steps * 1e3, "steps": args.steps,
                "what": "measured in this same run after the strips: every rank K full frames of BASELINE.json's "
                        "headline workload on its own GPU, no collective (bench.py --mode frames), barrier to barrier, "
                        "slowest rank; divide by the N = 1 line's value for the weak-scaling efficiency"}
        if one_gpu is not None:
            one_gpu["speedup_of_this_line"] = fps / one_gpu["value"]
            out["strong_scaling_reference"] = one_gpu
        if args.workload == "synth10m":
            out["config"]["resident_model"] = (
                "sorted ONCE at upload into tile-coherent order (Morton code of the 4-pixel cell of each "
                "triangle's projected centroid: key kernel + device radix sort + three gathers, about 1 GB of "
                "traffic, ~2 ms, outside the timed region); depth ties and the winner plane keep the caller's indices")
        want_api = not args.no_api_calls and world == 1 and H * W <= 4096 * 4096 and T <= 1_000_000
        child = None
        if not args.no_cpu_baseline:
            # (rank 0, beside every N: the other ranks wait at the closing barrier meanwhile)
            child = cpu_baseline(args.workload, args.synth_triangles, args.max_triangles, api=want_api)
            cb = out["cpu_baseline"] = child["cpu_baseline"]
            # north_star's denominator is the 16-thread column; the best column is beside it
            out["speedup_vs_cpu_baseline"] = fps / cb["value"]
            out["speedup_vs_best_cpu_column"] = fps / cb["best"]["frames_per_s"]
            lone = 1e3 / (elapsed_single / args.steps * 1e3)       # frames/s of frames one at a time on one stream
            out["speedup_lone_frame"] = {"frames_per_s": lone, "vs_16_threads": lone / cb["value"],
                                         "vs_best_cpu_column": lone / cb["best"]["frames_per_s"],
                                         "what": "ms_per_frame_single_stream (no frames in flight together) "
                                                 "against the CPU columns"}
        if want_api:
            out["api_call"] = api_calls(tri, col, nrm, H, W, fov, device)
            if child is not None and "api_cpu_oracle" in child:
                out["api_call"]["cpu_oracle"] = child["api_cpu_oracle"]
        print(json.dumps(out), flush=True)

    if world > 1:
        dist.barrier()
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
