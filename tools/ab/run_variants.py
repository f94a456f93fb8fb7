#!/usr/bin/env python3
"""Times bench.py's frame for every library under csrc/variants/ (and the default build), one child process each.
usage: python tools/ab/run_variants.py [--only tag,tag] [bench.py args ...]"""
import glob, json, os, subprocess, sys
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
args = sys.argv[1:]
only = None
if args and args[0] == "--only":
    only = set(args[1].split(",")); args = args[2:]
libs = sorted(glob.glob(os.path.join(ROOT, "opencl-raytracer_amd/csrc/variants/libhip_raytracer_*.so")))
for src in [None] + libs:
    tag = os.path.basename(src)[len("libhip_raytracer_"):-3] if src else "default"
    if only and tag not in only:
        continue
    env = dict(os.environ)
    if src:
        env["RT_LIB_OVERRIDE"] = src
    r = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--no-cpu-baseline", "--no-extra", "--steps", "5", "--warmup", "2"] + args,
                       capture_output=True, text=True, env=env)
    try:
        d = json.loads(r.stdout.strip().splitlines()[-1])
        print(f"{tag:28s} {d['ms_per_step']:8.3f} ms  kernel_ms {d['roofline'].get('kernel_ms', 0):8.3f}  ref {d['rays_reference']} traced {d['rays_traced']}", flush=True)
    except Exception:
        print("FAILED", tag, r.stdout[-300:], r.stderr[-600:], flush=True)
