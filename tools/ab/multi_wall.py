#!/usr/bin/env python3
"""What a caller of the boundary waits for: wall clock of the synchronous Render() - kernels + read-back into pinned host memory
(OpenCLRaytracer.cpp:94) - through ONE context (rt_render) and through k contexts sharing the one GPU (rt_render_multi: every shard
copies its tiles straight into the pinned host frame, persistent host threads). VERDICT r3 item 5: k = 4 within 5 % of k = 1.
usage: python tools/ab/multi_wall.py [workload ...]   -> one JSON line"""
import json, os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
import bench
from opencl_raytracer_amd import camera
from opencl_raytracer_amd.hip_raytracer import HIPRaytracer, MultiHIPRaytracer

res = {"library_sha16": bench.library_sha16(), "what": "best of 5 synchronous Render() calls, ms (kernels + D2H into pinned host memory)", "workloads": {}}
for name in (sys.argv[1:] or ["cfg4", "cfg3"]):
    desc, objs, lights, W, H, kernel, depth = bench.load_workload(name)
    z = float(camera.camera_z(H))
    row = {}
    with HIPRaytracer(objs, lights, None, depth, kernel=kernel, camera=(W, H, z)) as rt:
        rt.Render()
        row["one_context"] = rt.render_host_ms(5)
    for k in (2, 4):
        with MultiHIPRaytracer(objs, lights, None, depth, devices=[0] * k, kernel=kernel, camera=(W, H, z)) as rt:
            rt.Render()
            row[f"{k}_contexts_one_gpu"] = rt.render_host_ms(5)
    res["workloads"][name] = row
print(json.dumps(res))
