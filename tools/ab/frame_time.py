#!/usr/bin/env python3
"""Device time of a cfg4 frame through rt_render_device on (a) the legacy default stream (handle 0) and (b) a torch side
stream. usage: python tools/ab/frame_time.py [workload] [frames]"""
import os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
import torch
import bench
from opencl_raytracer_amd import camera
from opencl_raytracer_amd.hip_raytracer import HIPRaytracer
wl = sys.argv[1] if len(sys.argv) > 1 else "cfg4"
frames = int(sys.argv[2]) if len(sys.argv) > 2 else 5
desc, objs, lights, W, H, kernel, depth = bench.load_workload(wl)
rt = HIPRaytracer(objs, lights, None, depth, kernel=kernel, camera=(W, H, float(camera.camera_z(H))))
out = torch.empty((W * H, 4), dtype=torch.float32, device="cuda")
for name, stream in (("default stream (0)", None), ("torch side stream", torch.cuda.Stream())):
    handle = stream.cuda_stream if stream is not None else 0
    for _ in range(2):
        rt.render_device(out.data_ptr(), handle)
    torch.cuda.synchronize()
    rt.timing_reset()
    t0 = time.perf_counter()
    for _ in range(frames):
        rt.render_device(out.data_ptr(), handle)
    torch.cuda.synchronize()
    wall = (time.perf_counter() - t0) / frames * 1e3
    ms, n = rt.timing_summary()
    print(f"{wl} {name:22s}: wall {wall:7.3f} ms/frame, events {ms / max(n, 1):7.3f} ms/frame, rounds {rt.stats().rounds}", flush=True)
rt.close()
