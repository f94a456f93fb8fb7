#!/usr/bin/env python3
"""bench.py - throughput of the IRaytracer hot path on MI355X, in BASELINE.json's metric.

    python bench.py [--gpus N] [--steps K] [--warmup W] [--workload cfg3|cfg2|cfg4|cfg4crop|cfg5base|cfg5]

A step = one frame: every primary ray of the workload through hit-test -> shade -> reflection loop, scene and
rays resident in HBM before the timed region (rays are regenerated in-kernel from (W,H,z); the framebuffer is
written to HBM; for N > 1 the step includes the RCCL gather of row-tiles to rank 0).

Default workload, for every N = BASELINE.json configs[3], the configuration the metric ("... at 4096x4096,
1/2/4/8 GPU") and the north_star target are quoted on: synthetic 100k spheres + 32 lights, 4096x4096,
shade_and_reflect, depth 3 - it fits one GPU (32 MB of scene) and is the same frame at every N, so the per-N
values are comparable (strong scaling). At N = 1 the line also carries `extra.cfg3`: BASELINE configs[2]
(simpleScene 4096x4096 depth 3, the HBM-bound small-scene kernel) with its own HBM roofline.

Prints ONE JSON line (rank 0). `value` is Mrays/s over the rays this backend really TRACED for the frame (primary +
shadow + reflection after its exact eliminations = R_act, SURVEY.md 8d's primary figure); the rays the reference
semantics trace for the same frame (R_ref: every shadow ray of every light, 15x more for the default workload) are
reported beside it as `mrays_reference_equivalent_per_s` - algebra saved, not throughput.
"""
from __future__ import annotations

import argparse
import json
import os
import sys
import time
from pathlib import Path

ROOT = Path(__file__).resolve().parent
sys.path.insert(0, str(ROOT))

import numpy as np  # noqa: E402

import _pkg  # noqa: E402

_pkg.load()
from opencl_raytracer_amd import camera, scene_loader, sharding, synthetic, tessellate  # noqa: E402

HBM_PEAK_GBS = 8000.0  # MI355X_MICROARCH.md: HBM3E 8 TB/s spec
VALU_PEAK_LANE_OPS = 256 * 4 * 32 * 2.4e9  # 256 CUs x 4 SIMD-32 x 2.4 GHz = 7.86e13 fp32 lane-instructions/s (157.3 TFLOP/s FMA)
SPHERE_TEST_LANE_OPS = 35  # minimal exact ray-sphere rejection test: 19 fma + 12 mul + 4 add

WORKLOADS = {
    # name: (description, scene, W, H, kernel, depth)
    "cfg2": ("multipleSpheres.txt 1920x1080 shade (BASELINE configs[1])", "multipleSpheres", 1920, 1080, "shade", 0),
    "cfg3": ("simpleScene.txt 4096x4096 shade_and_reflect depth=3 (BASELINE configs[2])", "simpleScene", 4096, 4096, "shade_and_reflect", 3),
    "cfg5base": ("roundedCube.txt 8192x8192 shade_and_reflect depth=5 (analytic base of BASELINE configs[4])", "roundedCube", 8192, 8192, "shade_and_reflect", 5),
    # EXTENSION: the reference has no triangle type; own spec, self-parity only
    "cfg5": ("roundedCube.txt tessellated to ~1M triangles, 8192x8192 shade_and_reflect depth=5 (BASELINE configs[4]; triangle "
             "semantics are this repo's own extension)", "roundedCube", 8192, 8192, "shade_and_reflect", 5),
    "cfg4": ("synthetic 100k spheres + 32 lights 4096x4096 shade_and_reflect depth=3 (BASELINE configs[3])", None, 4096, 4096, "shade_and_reflect", 3),
    "cfg4crop": ("synthetic 100k spheres + 32 lights, centred 512x512 window of the 4096x4096 grid, depth=3", None, 4096, 4096, "shade_and_reflect", 3),
}


def library_sha16():
    """sha-256 (first 16 hex digits) of the C-ABI library this process loads - the stamp tools/pmc_frame.sh puts on the
    counter files it writes."""
    import hashlib
    from opencl_raytracer_amd import hip_raytracer
    path = os.environ.get("RT_LIB_OVERRIDE", str(hip_raytracer.LIB_PATH))
    return hashlib.sha256(open(path, "rb").read()).hexdigest()[:16]


def profile_key(workload, flags=()):
    """Name of the counter file of a run: the workload plus every flag that changes what the kernels do or read
    (`--ray-buffer`, `--literal`, `--no-grid`), e.g. cfg4+ray-buffer."""
    return workload + "".join("+" + f.lstrip("-") for f in sorted(flags))


def frame_profile(workload, flags=()):
    """profiles/frame_<key>.json (rocprofv3 PMC + kernel trace of one frame, tools/pmc_frame.sh) - but only when it was
    measured on THIS build of the library with THESE flags; otherwise (None, reason): a counter file of another build, or of
    a run that read a ray buffer / tested every object, says nothing about the kernels that are running now."""
    workload = profile_key(workload, flags)
    path = ROOT / "profiles" / f"frame_{workload}.json"
    if not path.exists():
        return None, f"profiles/frame_{workload}.json not collected (tools/pmc_frame.sh {' --'.join(workload.split('+'))})"
    try:
        d = json.loads(path.read_text())
    except Exception as ex:  # noqa: BLE001
        return None, f"profiles/frame_{workload}.json unreadable: {ex}"
    have = library_sha16()
    if sorted(d.get("flags", [])) != sorted(f.lstrip("-") for f in flags):
        return None, f"profiles/frame_{workload}.json was measured with flags {d.get('flags', [])}: counters not quoted"
    if d.get("lib_sha16") != have:
        return None, (f"profiles/frame_{workload}.json was measured on library {d.get('lib_sha16')}, this run uses {have}: "
                      "counters not quoted (re-run tools/pmc_frame.sh)")
    return d, None


def load_workload(name):
    desc, scene, W, H, kernel, depth = WORKLOADS[name]
    if scene is not None:
        objs, lights = scene_loader.load_scene(str(ROOT / "scenes" / f"{scene}.txt"))
        if name == "cfg5":
            lat, lon, k = tessellate.subdivision_for(objs, 1_000_000)
            objs = tessellate.tessellate(objs, lat, lon, k)
            desc = desc.replace("~1M triangles", f"{len(objs)} triangles (spheres {lat}x{lon} lat-long, box faces {k}x{k})")
    else:
        objs, lights = synthetic.spheres_and_lights(100_000, 32)
    return desc, objs, lights, W, H, kernel, depth


def host_cores():
    """CPU share this process really has: min(logical CPUs, affinity mask, cgroup cpu.max quota)."""
    n = os.cpu_count() or 1
    try:
        n = min(n, len(os.sched_getaffinity(0)))
    except (AttributeError, OSError):
        pass
    for path in ("/sys/fs/cgroup/cpu.max", "/sys/fs/cgroup/cpu/cpu.cfs_quota_us"):
        try:
            txt = open(path).read().split()
            if path.endswith("cpu.max"):
                if txt[0] != "max":
                    n = min(n, max(1, int(int(txt[0]) / int(txt[1]))))
            else:
                quota = int(txt[0])
                period = int(open("/sys/fs/cgroup/cpu/cpu.cfs_period_us").read())
                if quota > 0:
                    n = min(n, max(1, quota // period))
        except (OSError, ValueError, IndexError):
            pass
    return n


def cpu_baseline(objs, lights, rays, kernel, depth, sample_desc):
    """A bounded sample of the same workload on the host's cores: the REFERENCE's own kernels where oracle/_ref exists (the
    three .cl files compiled verbatim for the host in the build container - kind 'reference'; the library travels to the GPU
    box with the snapshot, the sources do not), else the oracle's CPU restatement (kind 'port'). Both trace every ray the
    reference traces; the restatement's time is reported beside the reference's."""
    from oracle import oracle
    rs = oracle.Restatement(True)
    threads = host_cores()
    best = None
    t_total = 0.0
    for _ in range(3):
        t0 = time.perf_counter()
        res = rs.render(kernel, objs, lights, rays, depth, threads=threads, want_aux=False)
        dt = time.perf_counter() - t0
        t_total += dt
        best = dt if best is None else min(best, dt)
        if t_total > 12.0:
            break
    port = {"value": res["rays_ref"] / best / 1e6, "unit": "Mrays/s", "cores": int(res["threads"]), "kind": "port",
            "sample": sample_desc, "seconds": best}
    try:
        if not oracle.reference_available():
            return port
        ref = oracle.Reference(kernel, True)
        best_ref, t_total, used = None, 0.0, 0
        for _ in range(3):
            t0 = time.perf_counter()
            used = ref.render(objs, lights, rays, depth, threads=threads)["threads"]
            dt = time.perf_counter() - t0
            t_total += dt
            best_ref = dt if best_ref is None else min(best_ref, dt)
            if t_total > 12.0:
                break
        return {"value": res["rays_ref"] / best_ref / 1e6, "unit": "Mrays/s", "cores": int(used), "kind": "reference",
                "sample": sample_desc, "seconds": best_ref,
                "what": "the reference's shade_and_reflect / shade / hittest kernels compiled verbatim for the host (oracle/_ref, OpenMP over "
                        "work-items; OpenCL builtins from oracle/ref_shim.cl) - no OpenCL CPU device exists on this box",
                "port": port}
    except Exception as ex:  # noqa: BLE001  (an optional leg must never cost the line)
        port["reference_unavailable"] = f"{type(ex).__name__}: {ex}"
        return port


def cpu_backend_baseline(objs, lights, rays, kernel, depth, sample_desc):
    """The same sample on this repo's CPU backend behind IRaytracer (host/CPURaytracer.cpp; std::thread,
    every core) - the product's own no-GPU backend, not test infrastructure."""
    from opencl_raytracer_amd.cpu_raytracer import CPURaytracer
    rt = CPURaytracer(objs, lights, rays, depth, kernel=kernel, threads=host_cores())
    rt.Render()
    return {"value": rt.rays_traced / rt.seconds / 1e6, "unit": "Mrays/s", "cores": rt.threads_used, "kind": "backend",
            "sample": sample_desc, "seconds": rt.seconds,
            "what": "CPURaytracer : IRaytracer (host/CPURaytracer.cpp), every ray against every object like the reference's kernels"}


def measure_brute_force_window(device_index, edge, default_frame=None):
    """The brute-force traversal (every object for every ray: packed-pair stream, RT_FLAG_NO_GRID) on the centred
    edge x edge window of the cfg4 grid - the kernel that is measured against the FP32 VALU roofline."""
    import torch
    from opencl_raytracer_amd.hip_raytracer import HIPRaytracer
    desc, objs, lights, W, H, kernel, depth = load_workload("cfg4")
    if edge >= W:
        rt = HIPRaytracer(objs, lights, None, depth, kernel=kernel, device=device_index, grid=False,
                          camera=(W, H, float(camera.camera_z(H))))
        n_out = W * H
    else:
        rays = camera.crop_rays(W, H, W // 2 - edge // 2, H // 2 - edge // 2, edge, edge)
        rt = HIPRaytracer(objs, lights, rays, depth, kernel=kernel, device=device_index, grid=False)
        n_out = len(rays)
    out = torch.empty((n_out, 4), dtype=torch.float32, device=torch.device("cuda", device_index))
    st = rt.count_rays()  # doubles as the warm-up frame
    torch.cuda.synchronize()
    rt.timing_reset()
    t0 = time.perf_counter()
    steps = 1 if edge >= W else 2
    for _ in range(steps):
        rt.render_device(out.data_ptr())
    torch.cuda.synchronize()
    dt = (time.perf_counter() - t0) / steps
    ms_sum, n = rt.timing_summary()
    kernel_ms = ms_sum / max(n, 1)
    valu = st.object_tests * SPHERE_TEST_LANE_OPS / (kernel_ms * 1e-3)
    res = {"workload": ("same frame" if edge >= W else f"same scene, centred {edge}x{edge} window of the 4096x4096 grid") +
                       ", every object tested for every ray (RT_FLAG_NO_GRID)",
           "kernel": "rt::wf_trace_closest + rt::wf_trace_any_slice (+ wf_resume)",
           "value": st.rays_traced / dt / 1e6, "unit": "Mrays/s", "ms_per_step": dt * 1e3,
           "mrays_reference_equivalent_per_s": st.rays_reference / dt / 1e6,
           "bound": "valu", "achieved": valu / 1e12, "peak": VALU_PEAK_LANE_OPS / 1e12, "roofline_unit": "T lane-instr/s",
           "frac": valu / VALU_PEAK_LANE_OPS, "object_tests": int(st.object_tests),
           "tests_per_s": st.object_tests / (kernel_ms * 1e-3)}
    if default_frame is not None and edge >= W and tuple(default_frame.shape) == tuple(out.shape):
        # the same frame by two routes - every object tested for every ray here, screen tiles + block walk + light tiles in the timed
        # run: all 4 x 16.8 M words must agree bit for bit (the grid is an exact elimination)
        same = torch.eq(out.view(torch.int32), default_frame.view(torch.int32))
        res["frame_bit_identical_to_default_path"] = bool(same.all().item())
        res["pixels_compared"] = int(out.shape[0])
        res["pixels_differing"] = int((~same.all(dim=1)).sum().item())
    rt.close()
    return res


def measure_cfg3(device_index, fast_phong=False):
    """BASELINE configs[2] (simpleScene 4096x4096 shade_and_reflect depth 3): the small-scene kernel, HBM-write bound.
    fast_phong: the same frame under RT_FLAG_FAST_PHONG (opt-in; colours within 1e-5 of the reference, rays bit-exact)."""
    import torch
    from opencl_raytracer_amd.hip_raytracer import HIPRaytracer
    desc, objs, lights, W, H, kernel, depth = load_workload("cfg3")
    z = float(camera.camera_z(H))
    rt = HIPRaytracer(objs, lights, None, depth, kernel=kernel, camera=(W, H, z), device=device_index, fast_phong=fast_phong)
    out = torch.empty((W * H, 4), dtype=torch.float32, device=torch.device("cuda", device_index))
    st = rt.count_rays()
    for _ in range(3):
        rt.render_device(out.data_ptr())
    torch.cuda.synchronize()
    rt.timing_reset()
    steps = 20
    t0 = time.perf_counter()
    for _ in range(steps):
        rt.render_device(out.data_ptr())
    torch.cuda.synchronize()
    dt = (time.perf_counter() - t0) / steps
    ms_sum, n = rt.timing_summary()
    kernel_ms = ms_sum / max(n, 1)
    alg = 16 * W * H + 320 * len(objs) + 64 * len(lights)
    prof, why = frame_profile("cfg3", ["--fast-phong"] if fast_phong else [])
    traffic = prof["hbm_bytes_per_launch"] if prof else None
    roof = {"bound": "hbm", "achieved": alg / (kernel_ms * 1e-3) / 1e9, "peak": HBM_PEAK_GBS, "unit": "GB/s",
            "frac": alg / (kernel_ms * 1e-3) / 1e9 / HBM_PEAK_GBS, "traffic": traffic, "kernel": "rt::render_pixels<2,true,false>",
            "kernel_ms": kernel_ms, "algorithmic_bytes_per_launch": alg}
    if prof:
        roof["traffic_over_algorithmic"] = traffic / alg
        roof["traffic_source"] = f"profiles/frame_cfg3.json (rocprofv3 PMC, library {prof['lib_sha16']})"
    else:
        roof["traffic_unavailable"] = why
    res = {"workload": desc, "value": st.rays_traced / dt / 1e6, "unit": "Mrays/s", "ms_per_step": dt * 1e3, "steps": steps,
           "mrays_reference_equivalent_per_s": st.rays_reference / dt / 1e6,
           "render_wall_ms_incl_d2h": rt.render_host_ms(3), "setup_ms": rt.setup_times(),
           "rays_reference": int(st.rays_reference), "rays_traced": int(st.rays_traced), "hit_pixels": int(st.hit_pixels),
           "mrays_traced_per_s": st.rays_traced / dt / 1e6, "roofline": roof}
    rt.close()
    return res


def free_port():
    import socket
    with socket.socket() as so:
        so.bind(("127.0.0.1", 0))
        return so.getsockname()[1]


def launch_ranks(n, argv, script=None, run=None):
    """`bench.py --gpus N` started WITHOUT a launcher (no WORLD_SIZE in the environment): start the N ranks ourselves -
    `python -m torch.distributed.run --nproc-per-node N bench.py ...` as a CHILD process (never exec: a process that has
    touched the GPU must not be replaced, and this one must not touch it at all) - relay rank 0's JSON line and return
    the child's exit code. Refuses when the node has fewer than N GPUs, unless RT_BENCH_ONE_GPU=1 asks for the rehearsal
    in which every rank shares GPU 0 (then over gloo unless RT_BENCH_BACKEND says otherwise)."""
    import subprocess
    import torch  # (import only: torch.cuda.device_count() does not initialise the GPU on this image)
    env = dict(os.environ)
    env.setdefault("MASTER_ADDR", "127.0.0.1")
    env.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")
    if env.get("RT_BENCH_ONE_GPU"):
        env.setdefault("RT_BENCH_BACKEND", "gloo")
    else:
        have = torch.cuda.device_count()
        if n > have:
            print(f"bench.py: --gpus {n} but this node shows {have} GPU(s); RT_BENCH_ONE_GPU=1 rehearses the {n}-rank path on one",
                  file=sys.stderr)
            return 2
    env["RT_BENCH_LAUNCHER"] = "self"
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", f"--nproc-per-node={n}", "--master-addr", "127.0.0.1",
           "--master-port", str(free_port()), str(script or Path(__file__).resolve())] + list(argv)
    res = (run or subprocess.run)(cmd, env=env, stdout=subprocess.PIPE, text=True)
    line = None
    for cand in (res.stdout or "").splitlines():
        cand = cand.strip()
        if cand.startswith("{") and '"metric"' in cand:
            line = cand
    if line is not None:
        print(line, flush=True)
    elif res.returncode == 0:
        print("bench.py: the ranks exited cleanly but printed no result line", file=sys.stderr)
        return 3
    assert not torch.cuda.is_initialized(), "the launching process must not touch the GPU"
    return res.returncode


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=None)
    ap.add_argument("--warmup", type=int, default=None)
    ap.add_argument("--workload", default="cfg4", choices=sorted(WORKLOADS))
    ap.add_argument("--tile-rows", type=int, default=16)
    ap.add_argument("--crop", type=int, default=512, help="window edge for --workload cfg4crop")
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--no-extra", action="store_true", help="skip the secondary large-scene measurement")
    ap.add_argument("--literal", action="store_true", help="trace every ray the reference traces (no exact eliminations)")
    ap.add_argument("--ray-buffer", action="store_true", help="read primary rays from an uploaded buffer instead of in-kernel generation")
    ap.add_argument("--fast-phong", action="store_true", help="RT_FLAG_FAST_PHONG: colour-only normalisations / specular power on the fast hardware paths (opt-in)")
    ap.add_argument("--no-grid", action="store_true", help="large scenes: test every object for every ray (brute-force traversal, the VALU-roofline kernel)")
    args = ap.parse_args()

    if args.gpus > 1 and "WORLD_SIZE" not in os.environ:
        sys.exit(launch_ranks(args.gpus, sys.argv[1:]))

    import torch
    import torch.distributed as dist

    world = int(os.environ.get("WORLD_SIZE", "1"))
    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    if world > 1:
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        os.environ.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")
        # RT_BENCH_BACKEND=gloo + RT_BENCH_ONE_GPU=1 rehearses the N-rank path on a single-GPU box
        backend = os.environ.get("RT_BENCH_BACKEND", "nccl")
        if os.environ.get("RT_BENCH_ONE_GPU"):
            local_rank = 0
        torch.cuda.set_device(local_rank)
        if backend == "nccl":
            dist.init_process_group("nccl", device_id=torch.device("cuda", local_rank))
        else:
            dist.init_process_group(backend)
    if args.gpus != world:  # a launcher started a different number of ranks than --gpus says: the line would lie about N
        if rank == 0:
            print(f"bench.py: --gpus {args.gpus} but WORLD_SIZE {world}", file=sys.stderr)
        sys.exit(2)
    device = torch.device("cuda", local_rank)
    torch.cuda.set_device(device)

    heavy = args.workload in ("cfg4", "cfg5")
    steps = args.steps if args.steps is not None else (2 if heavy else 20)
    warmup = args.warmup if args.warmup is not None else (1 if heavy else 3)

    desc, objs, lights, W, H, kernel, depth = load_workload(args.workload)
    z = float(camera.camera_z(H))
    crop = None
    if args.workload == "cfg4crop":
        crop = (W // 2 - args.crop // 2, H // 2 - args.crop // 2, args.crop, args.crop)
        desc = desc.replace("512x512", f"{args.crop}x{args.crop}")

    from opencl_raytracer_amd.distributed import ShardedHIPRaytracer
    from opencl_raytracer_amd.hip_raytracer import HIPRaytracer  # noqa: F401
    # N > 1 over RCCL: frame k's tiles travel to rank 0 and are put in place while frame k + 1 is being rendered
    # (distributed.FrameGather, pipelined mode; all of it inside the timed region, which ends with a device-wide sync).
    # RT_BENCH_PIPELINE=0: render, exchange, assemble one after the other. The gloo rehearsal stages through host memory anyway.
    # RT_BENCH_PIPELINE=force: also in the rehearsal (the same code path end to end on a one-GPU box).
    # OPT-IN (RT_BENCH_PIPELINE=1) until the pipelined exchange has run once over RCCL on two real GPUs: the one-GPU gloo
    # rehearsal blocks the host in .cpu() / req.wait() and so cannot show a stream-ordering stall.
    _pl = os.environ.get("RT_BENCH_PIPELINE", "0")
    pipeline = world > 1 and ((dist.get_backend() == "nccl" and _pl == "1") or _pl == "force")
    pipeline_why = ("RT_BENCH_PIPELINE=%s" % _pl) if pipeline else (
        "synchronous exchange: the pipelined one is opt-in (RT_BENCH_PIPELINE=1) until it has been validated over RCCL on >= 2 GPUs")

    if crop is not None:
        rays = camera.crop_rays(W, H, *crop)
        rt = ShardedHIPRaytracer(objs, lights, rays, depth, kernel=kernel, tile_rows=args.tile_rows, width=crop[2],
                                 device_index=local_rank, pipeline=pipeline, literal=args.literal, grid=not args.no_grid, fast_phong=args.fast_phong)
        frame_w, frame_h = crop[2], crop[3]
        ray_source = "buffer"
    elif args.ray_buffer:
        rays = camera.primary_rays(W, H)
        rt = ShardedHIPRaytracer(objs, lights, rays, depth, kernel=kernel, tile_rows=args.tile_rows, width=W,
                                 device_index=local_rank, pipeline=pipeline, literal=args.literal, raygen=False, grid=not args.no_grid, fast_phong=args.fast_phong)
        frame_w, frame_h = W, H
        ray_source = "buffer"
    else:
        rt = ShardedHIPRaytracer(objs, lights, None, depth, camera=(W, H, z), kernel=kernel, tile_rows=args.tile_rows,
                                 device_index=local_rank, pipeline=pipeline, literal=args.literal, grid=not args.no_grid, fast_phong=args.fast_phong)
        frame_w, frame_h = W, H
        ray_source = "in-kernel pinhole"
    n_rays = rt.n_rays

    # untimed instrumentation pass: ray counts of this rank's tiles
    st = rt.rt.count_rays()
    red_dev = device if (world == 1 or dist.get_backend() == "nccl") else torch.device("cpu")
    counts = torch.tensor([st.rays_reference, st.rays_traced, st.hit_pixels, st.object_tests], dtype=torch.int64, device=red_dev)
    if world > 1:
        dist.all_reduce(counts)
    rays_ref, rays_act, hit_pixels, tests_total = (int(x) for x in counts.tolist())

    def step():
        return rt.Render()

    pipeline_note = None
    if pipeline:
        # self-check on THIS node before anything is timed: the pipelined exchange must deliver the frame the synchronous
        # exchange delivers, bit for bit (both slots exercised); otherwise time the synchronous path and say so
        ref = rt.render_synchronous()
        ok = 1
        for _ in range(3):
            f = step()
            if rank == 0 and not torch.equal(f, ref):
                ok = 0
        torch.cuda.synchronize(device)
        flag = torch.tensor([ok], dtype=torch.int64, device=red_dev)
        dist.all_reduce(flag, op=dist.ReduceOp.MIN)
        if int(flag.item()) != 1:
            rt.gatherer.drain()
            rt.gatherer.pipeline = False
            rt.gatherer.k = 0
            pipeline = False
            pipeline_note = "the pipelined exchange disagreed with the synchronous one in the self-check: timed synchronously"
        del ref
    for _ in range(warmup):
        step()
    torch.cuda.synchronize(device)
    if world > 1:
        dist.barrier()
    torch.cuda.synchronize(device)
    rt.rt.timing_reset()
    t0 = time.perf_counter()
    for _ in range(steps):
        frame = step()
    torch.cuda.synchronize(device)
    if world > 1:
        dist.barrier()
    torch.cuda.synchronize(device)
    elapsed = time.perf_counter() - t0
    kernel_ms_sum, launches = rt.rt.timing_summary()

    extra = {}
    if world == 1 and args.workload == "cfg4" and not args.no_extra:
        extra["cfg3"] = measure_cfg3(local_rank)
        extra["cfg3_fast_phong"] = measure_cfg3(local_rank, fast_phong=True)
        extra["cfg3_fast_phong"]["flag"] = ("RT_FLAG_FAST_PHONG (opt-in): shading normal / view vector / reflected light vector by v_rsq + "
                                            "one Newton step, specular power by exp2(e log2 x); every ray bit-exact, colours within 1e-5 of the reference "
                                            "(tests/test_fast_phong_gpu.py)")

    t = torch.tensor([elapsed], dtype=torch.float64, device=red_dev)
    if world > 1:
        dist.all_reduce(t, op=dist.ReduceOp.MAX)
    elapsed = float(t.item())
    # every rank's own share: device time of its kernels per frame (HIP events on its render stream)
    share = torch.zeros(world, dtype=torch.float64, device=red_dev)
    share[rank] = kernel_ms_sum / max(launches, 1)
    if world > 1:
        dist.all_reduce(share)
    share_ms = [float(x) for x in share.tolist()]
    # what the boundary's synchronous Render() costs a caller: kernels + blocking read-back into pinned host memory
    # (OpenCLRaytracer.cpp:94), and the one-time set-up that sits outside every timer (its constructor's work, :13-74)
    render_wall_ms = rt.rt.render_host_ms(3) if world == 1 else None
    setup_ms = rt.rt.setup_times()

    if rank == 0:
        assert frame is not None and frame.shape[0] == n_rays
        ms_per_step = elapsed / steps * 1e3
        value = rays_act * steps / elapsed / 1e6  # rays really traced (R_act); R_ref/t is mrays_reference_equivalent_per_s
        kernel_ms = kernel_ms_sum / max(launches, 1)
        # algorithmic (compulsory) HBM bytes per launch of this rank's kernel:
        #   16 B/ray framebuffer write (+ 32 B/ray ray-buffer read when rays come from HBM) + scene as uploaded
        local = rt.rt.local_rays
        alg_bytes = 16 * local + (32 * local if ray_source == "buffer" else 0) + 320 * len(objs) + 64 * len(lights)
        achieved = alg_bytes / (kernel_ms * 1e-3) / 1e9 if kernel_ms > 0 else 0.0
        run_flags = [f for f, on in (("--ray-buffer", args.ray_buffer), ("--literal", args.literal), ("--no-grid", args.no_grid), ("--fast-phong", args.fast_phong)) if on]
        prof, prof_why = (frame_profile(args.workload, run_flags) if world == 1 else (None, "counters are collected on one GPU"))
        traffic = prof["hbm_bytes_per_launch"] if prof else None
        out = {
            "metric": "Mrays/s (primary+reflect+shadow) at 4096x4096, 1/2/4/8 GPU; max RGB diff vs ref",
            "value": value, "unit": "Mrays/s", "n_gpus": world, "steps": steps, "warmup": warmup,
            "ms_per_step": ms_per_step, "higher_is_better": True, "scaling": "strong", "vs_baseline": None,
            "dtype": "f32", "data": "synthetic",
            "config": {"workload": desc, "width": frame_w, "height": frame_h, "objects": int(len(objs)),
                       "lights": int(len(lights)), "kernel": kernel, "depth": depth, "primary_rays": ray_source,
                       "partition": (f"row-tiles of {args.tile_rows} rows, interleaved over {world} rank(s), gather to rank 0"
                                     + (", frame k's exchange overlapped with frame k+1's render (self-checked against the synchronous exchange)" if pipeline else "")
                                     + (f" [{pipeline_note}]" if pipeline_note else ""))
                       if world > 1 else "single GPU", "arithmetic": "fused (fma where the OpenCL front-end marks fmuladd)",
                       "literal": bool(args.literal), "fast_phong": bool(args.fast_phong)},
            # `value` counts the rays this backend really traced (R_act: the figure to hold against other
            # raytracers); the rays the REFERENCE semantics trace for the same frame (R_ref) give the reference-equivalent rate
            "rays_reference": rays_ref, "rays_traced": rays_act, "hit_pixels": hit_pixels,
            "mrays_traced_per_s": rays_act * steps / elapsed / 1e6,
            "mrays_reference_equivalent_per_s": rays_ref * steps / elapsed / 1e6,
            "library_sha16": library_sha16(),
            "render_wall_ms_incl_d2h": render_wall_ms,
            "setup_ms": setup_ms,
        }
        if world > 1:
            out["ranks_seen"] = dist.get_world_size()
            out["launcher"] = os.environ.get("RT_BENCH_LAUNCHER", "external (torch.distributed.run)")
            out["backend"] = dist.get_backend()
            out["share_ms"] = {"max": max(share_ms), "min": min(share_ms), "per_rank": share_ms}
            out["exchange_ms"] = max(0.0, ms_per_step - max(share_ms))  # what the gather adds to the slowest share
            out["pipeline"] = {"on": bool(pipeline), "why": pipeline_note or pipeline_why}
        hbm = {"bound": "hbm", "achieved": achieved, "peak": HBM_PEAK_GBS, "unit": "GB/s", "frac": achieved / HBM_PEAK_GBS,
               "traffic": traffic, "kernel_ms": kernel_ms, "algorithmic_bytes_per_launch": alg_bytes}
        if prof:
            # what the counters saw leave the L2s for the fabric (Infinity Cache hits included), against the compulsory bytes
            hbm["traffic_over_algorithmic"] = traffic / alg_bytes
            hbm["fabric_GBps"] = traffic / (kernel_ms * 1e-3) / 1e9 if kernel_ms > 0 else None
            hbm["fabric_frac_of_hbm_peak"] = traffic / (kernel_ms * 1e-3) / 1e9 / HBM_PEAK_GBS if kernel_ms > 0 else None
            hbm["traffic_source"] = f"profiles/frame_{args.workload}.json (rocprofv3 PMC, library {prof['lib_sha16']})"
        else:
            hbm["traffic_unavailable"] = prof_why
        if args.workload.startswith("cfg4"):
            # SURVEY.md 8d's figure for 100k-object frames: counted ray-object tests x 35 lane-instructions (the minimal exact
            # ray-sphere rejection test) / device time of all kernels of the frame, against 256 CU x 4 SIMD x 32 lanes x 2.4 GHz.
            tests = int(tests_total)
            valu = tests * SPHERE_TEST_LANE_OPS / (kernel_ms * 1e-3) if kernel_ms > 0 else 0.0
            culled = not (args.no_grid or args.literal)
            out["roofline"] = {"bound": "valu", "achieved": valu / 1e12, "peak": VALU_PEAK_LANE_OPS * world / 1e12,
                               "unit": "T lane-instr/s", "frac": valu / (VALU_PEAK_LANE_OPS * world), "traffic": traffic,
                               "kernel": ("rt::wf_walk_blocks (closest-hit rays) + rt::wf_walk<any> (shadow rays) + rt::wf_trace_primary_tiles" if culled else
                                          "rt::wf_trace_closest + rt::wf_trace_any_slice") + " (+ wf_resume)",
                               "kernel_ms": kernel_ms, "object_tests": tests,
                               "tests_per_s": tests / (kernel_ms * 1e-3) if kernel_ms > 0 else 0.0, "hbm": hbm}
            if culled and prof:
                # the issue-slot view and the per-kernel memory view of the same frame, from the stamped counter file
                rate = prof["valu_wave_instructions"] / (kernel_ms * 1e-3)
                out["roofline"]["valu_issue"] = {
                    "bound": "valu_issue", "achieved": rate / 1e9, "peak": prof["peak_valu_wave_instructions_per_s"] / 1e9,
                    "unit": "G wave-instr/s", "frac": rate / prof["peak_valu_wave_instructions_per_s"],
                    "peak_note": prof.get("peak_note"),
                    "lanes_per_instruction": prof["lanes_per_valu_instruction"],
                    "waves_waiting_frac": prof["wait_quad_cycles"] / prof["wave_quad_cycles"],
                    "salu_per_valu": prof["salu_wave_instructions"] / max(prof["valu_wave_instructions"], 1.0),
                    "source": f"profiles/frame_{args.workload}.json: PMC instruction count of one frame / live kernel time"}
                out["roofline"]["per_kernel"] = {
                    k: {"launches": v["launches"], "ms": v["ms"], "traffic_bytes": v["traffic_bytes"],
                        "traffic_frac_of_hbm_peak": v["traffic_frac_of_hbm_peak"], "valu_issue_frac": v["valu_issue_frac"],
                        "lanes_per_valu_instruction": v["lanes_per_valu_instruction"]}
                    for k, v in prof["kernels"].items()}
            elif culled:
                out["roofline"]["counters_unavailable"] = prof_why
            if culled:
                out["roofline"]["note"] = (
                    "default path = conservative grid culling: %.1f exact ray-object tests per traced ray instead of %d, so by "
                    "SURVEY 8d's definition (35 lane-instructions x executed tests) the frame sits at a small fraction of the VALU "
                    "roofline; what the kernels are really limited by is in valu_issue (issue slots, lanes per instruction) and in "
                    "hbm / per_kernel (bytes the L2s pulled through the fabric vs the compulsory bytes). The brute-force traversal "
                    "kernels the grid replaces - which ARE bound by those tests - are measured in brute_force") % (
                        tests / max(rays_act, 1), len(objs))
                if world == 1 and not args.no_extra:
                    out["roofline"]["brute_force"] = measure_brute_force_window(local_rank, 4096, default_frame=frame if torch.is_tensor(frame) else None)
        elif args.workload == "cfg5":
            hbm["kernel"] = "rt::wf_trace_grid_persistent<closest|any, triangles> + rt::wf_trace_primary_tiles (+ wf_resume)"
            hbm["note"] = ("triangles are this repo's extension (no reference semantics; SURVEY.md 8f5 defines no per-test work "
                           "figure for them): the mandated HBM figure only - the frame is bound by the grid walk's instruction "
                           "issue, like cfg4")
            out["roofline"] = hbm
        else:
            hbm["kernel"] = "rt::render_pixels"
            out["roofline"] = hbm
        if extra:
            out["extra"] = extra
        if not args.no_cpu_baseline and world == 1:
            if crop is not None or args.workload in ("cfg4", "cfg5"):
                e = 16 if args.workload == "cfg5" else 64   # the CPU statement tests every object for every ray
                cx, cy, cw, ch = (W // 2 - e // 2, H // 2 - e // 2, e, e)
                sample = camera.crop_rays(W, H, cx, cy, cw, ch)
                sdesc = f"centred {cw}x{ch} window of the {W}x{H} ray grid, same scene"
            else:
                rows = min(H, 1024)
                sample = camera.primary_rays(W, H, row_begin=H // 2 - rows // 2, row_end=H // 2 + rows // 2)
                sdesc = f"centre {rows} rows of the {W}x{H} ray grid, same scene"
            out["cpu_baseline"] = cpu_baseline(objs, lights, sample, kernel, depth, sdesc)
            try:  # (an optional baseline must never cost the result line)
                out["cpu_backend"] = cpu_backend_baseline(objs, lights, sample, kernel, depth, sdesc)
            except Exception as ex:  # noqa: BLE001
                out["cpu_backend"] = None
                out["cpu_backend_unavailable"] = f"{type(ex).__name__}: {ex}"
        else:
            out["cpu_baseline"] = None
        print(json.dumps(out))
    rt.close()
    if world > 1:
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
