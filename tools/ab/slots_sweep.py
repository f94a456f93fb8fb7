#!/usr/bin/env python3
"""cfg4 frame time for block variants (slots per block x cell factor). usage: python tools/ab/slots_sweep.py"""
import json, os, subprocess, sys
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
for tag, factors in (("default", (1.67,)), ("e6", (1.55, 1.67)), ("e5", (1.4, 1.55, 1.67)), ("e4", (1.25, 1.4, 1.55))):
    for f in factors:
        env = dict(os.environ, RT_WALK_BLOCK_FACTOR=str(f))
        if tag != "default":
            env["RT_LIB_OVERRIDE"] = os.path.join(ROOT, f"opencl-raytracer_amd/csrc/variants/libhip_raytracer_{tag}.so")
        r = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--no-cpu-baseline", "--no-extra", "--steps", "5", "--warmup", "2"],
                           capture_output=True, text=True, env=env)
        try:
            d = json.loads(r.stdout.strip().splitlines()[-1])
            print(f"{tag:8s} factor {f:5.2f}  {d['ms_per_step']:8.3f} ms  traced {d['rays_traced']}", flush=True)
        except Exception:
            print("FAILED", tag, f, r.stderr[-300:], flush=True)
