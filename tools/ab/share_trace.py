#!/usr/bin/env python3
"""Renders rank 0's share of the cfg4 frame at a given world size a few times (for rocprofv3 --kernel-trace --stats).
usage: python tools/ab/share_trace.py <world> [frames]"""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
import torch
import bench
from opencl_raytracer_amd import camera, sharding
from opencl_raytracer_amd.hip_raytracer import HIPRaytracer
world = int(sys.argv[1]); frames = int(sys.argv[2]) if len(sys.argv) > 2 else 4
desc, objs, lights, W, H, kernel, depth = bench.load_workload("cfg4")
rt = HIPRaytracer(objs, lights, None, depth, kernel=kernel, camera=(W, H, float(camera.camera_z(H))))
rt.set_shard(sharding.tile_rays_for_rows(W, 16), 0, world)
buf = torch.zeros((rt.local_rays, 4), dtype=torch.float32, device="cuda")
for _ in range(frames):
    rt.render_device(buf.data_ptr(), 0)
torch.cuda.synchronize()
rt.close()
