#!/usr/bin/env python3
"""B-ref: the reference's OWN kernels (the three .cl files compiled verbatim for the host, oracle/_ref - build container
only, /root/reference must be present) timed on this container's CPU threads, beside this repo's CPU statements of the
same algorithm (oracle port, CPURaytracer backend) on the same rays. BASELINE.md section 3 quotes the result.

    python tools/bref_timing.py > profiles/r02_bref_timing.json
"""
import json, os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import numpy as np
import bench
from opencl_raytracer_amd import camera
from opencl_raytracer_amd.cpu_raytracer import CPURaytracer
from oracle import oracle

if not oracle.reference_available():
    sys.exit("oracle/_ref is not built: this script only runs where /root/reference exists")
threads = len(os.sched_getaffinity(0))
rows = []
for wl, sample in (("cfg2", None), ("cfg3", None), ("cfg4", 64)):
    desc, objs, lights, W, H, kernel, depth = bench.load_workload(wl)
    if sample:
        rays = camera.crop_rays(W, H, W // 2 - sample // 2, H // 2 - sample // 2, sample, sample)
        what = f"centred {sample}x{sample} window of the {W}x{H} ray grid"
    else:
        rays = camera.primary_rays(W, H)
        what = f"full {W}x{H} frame"
    rs = oracle.Restatement(True)
    t0 = time.perf_counter(); port = rs.render(kernel, objs, lights, rays, depth, threads=threads, want_aux=False); t_port = time.perf_counter() - t0
    ref = oracle.Reference(kernel, True)
    t0 = time.perf_counter(); r = ref.render(objs, lights, rays, depth, threads=threads); t_ref = time.perf_counter() - t0
    be = CPURaytracer(objs, lights, rays, depth, kernel=kernel, threads=threads)
    t0 = time.perf_counter(); b = be.Render(); t_be = time.perf_counter() - t0
    n_ref = port["rays_ref"]
    same_port = bool(np.array_equal(np.ascontiguousarray(r["out"][:, :3]).view(np.uint32), np.ascontiguousarray(port["out"][:, :3]).view(np.uint32)))
    same_be = bool(np.array_equal(np.ascontiguousarray(r["out"][:, :3]).view(np.uint32), np.ascontiguousarray(b[:, :3]).view(np.uint32)))
    rows.append({"workload": desc, "sample": what, "rays_reference": n_ref, "threads": threads,
                 "reference_kernels_s": t_ref, "reference_kernels_Mrays_s": n_ref / t_ref / 1e6,
                 "oracle_port_s": t_port, "oracle_port_Mrays_s": n_ref / t_port / 1e6,
                 "cpu_backend_s": t_be, "cpu_backend_Mrays_s": n_ref / t_be / 1e6,
                 "port_equals_reference_bits": same_port, "backend_equals_reference_bits": same_be})
    print(json.dumps(rows[-1]), file=sys.stderr, flush=True)
cpu = [l.split(":", 1)[1].strip() for l in open("/proc/cpuinfo") if l.startswith("model name")]
print(json.dumps({"what": "reference kernels compiled verbatim for the host (fused flavour: -ffp-contract=on -mfma) vs this repo's CPU statements, same rays, same threads",
                  "cpu": cpu[0] if cpu else "?", "threads": threads, "rows": rows}, indent=1))
