#!/usr/bin/env python3
"""Wall clock of the synchronous Render() (kernels + read-back into pinned host memory) of the cfg4 frame under different
splits of rt_render's passes (RT_RENDER_SPLIT; RT_RENDER_PASSES=1: one pass).
usage: [SPLIT_WORKLOAD=cfg5] python tools/ab/render_split.py [split ...]"""
import json, os, subprocess, sys
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
CHILD = """
import sys; sys.path.insert(0, %r)
import bench
from opencl_raytracer_amd import camera
from opencl_raytracer_amd.hip_raytracer import HIPRaytracer
desc, objs, lights, W, H, kernel, depth = bench.load_workload(%r)
rt = HIPRaytracer(objs, lights, None, depth, kernel=kernel, camera=(W, H, float(camera.camera_z(H))))
rt.render_host_ms(3)
print(min(rt.render_host_ms(5) for _ in range(3)))
""" % (ROOT, os.environ.get("SPLIT_WORKLOAD", "cfg4"))
splits = sys.argv[1:] or ["one", "1,1", "3,1", "11,5", "2,1", "5,2,1", "4,2,1", "5,3", "9,4,2,1", "7,1"]
for s in splits:
    env = dict(os.environ)
    if s == "one": env["RT_RENDER_PASSES"] = "1"
    else: env["RT_RENDER_SPLIT"] = s
    r = subprocess.run([sys.executable, "-c", CHILD], capture_output=True, text=True, env=env)
    print(f"{s:10s} {r.stdout.strip() or r.stderr[-300:]}", flush=True)
