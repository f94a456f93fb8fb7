#!/usr/bin/env python3
"""Where rt_create's one-time host work goes (RT_SETUP_TRACE laps on stderr + rt_get_setup_times), for a workload.
usage: python tools/ab/setup_trace.py [workload ...]"""
import json, os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
os.environ["RT_SETUP_TRACE"] = "1"
import torch
torch.zeros(1, device="cuda")  # the HIP runtime's own start-up stays out of the first rt_create
import bench
from opencl_raytracer_amd import camera
from opencl_raytracer_amd.hip_raytracer import HIPRaytracer
for wl in sys.argv[1:] or ["cfg4"]:
    desc, objs, lights, W, H, kernel, depth = bench.load_workload(wl)
    for rep in range(2):
        print(f"--- {wl}, create #{rep}", file=sys.stderr, flush=True)
        rt = HIPRaytracer(objs, lights, None, depth, kernel=kernel, camera=(W, H, float(camera.camera_z(H))))
        buf = torch.zeros((rt.local_rays, 4), dtype=torch.float32, device="cuda")
        rt.render_device(buf.data_ptr(), 0)
        torch.cuda.synchronize()
        print(json.dumps({"workload": wl, "setup_ms": rt.setup_times()}), flush=True)
        rt.close()
