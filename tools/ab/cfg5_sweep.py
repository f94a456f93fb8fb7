#!/usr/bin/env python3
"""cfg5 (triangle extension) frame time: skip cap variants and block cell factors. usage: python tools/ab/cfg5_sweep.py"""
import json, os, subprocess, sys
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
def run(tag, env_extra):
    env = dict(os.environ, **env_extra)
    r = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--workload", "cfg5", "--no-cpu-baseline", "--no-extra"], capture_output=True, text=True, env=env)
    try:
        d = json.loads(r.stdout.strip().splitlines()[-1])
        print(f"{tag:28s} {d['ms_per_step']:8.3f} ms  traced {d['rays_traced']}", flush=True)
    except Exception:
        print("FAILED", tag, r.stderr[-300:], flush=True)
V = os.path.join(ROOT, "opencl-raytracer_amd/csrc/variants")
run("cap 8 factor 1.67", {})
for n in (4, 16, 63):
    run(f"cap {n} factor 1.67", {"RT_LIB_OVERRIDE": f"{V}/libhip_raytracer_sk{n}.so"})
for f in (1.0, 1.3, 2.2, 3.0):
    run(f"cap 8 factor {f}", {"RT_WALK_BLOCK_FACTOR": str(f)})
