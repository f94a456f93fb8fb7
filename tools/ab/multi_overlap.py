#!/usr/bin/env python3
"""Does a frame gain from running parts of it side by side on ONE GPU? rt_render_multi_device with k shards, all on device 0
(one context, one host thread, its own streams per shard): the walks of one shard can overlap the resume steps of another.
usage: python tools/ab/multi_overlap.py [workload] [tile rows]"""
import os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
import torch
import bench
from opencl_raytracer_amd.hip_raytracer import MultiHIPRaytracer
from opencl_raytracer_amd import camera

name = sys.argv[1] if len(sys.argv) > 1 else "cfg4"
rows = int(sys.argv[2]) if len(sys.argv) > 2 else 0
desc, objs, lights, W, H, kernel, depth = bench.load_workload(name)
z = float(camera.camera_z(H))
for k in tuple(int(x) for x in os.environ.get("RT_OVERLAP_SHARDS", "1,2,3,4").split(",")):
    tile_rows = rows or (H // k)
    with MultiHIPRaytracer(objs, lights, None, depth, devices=[0] * k, kernel=kernel, camera=(W, H, z), tile_rays=tile_rows * W) as rt:
        frame = torch.empty((rt.frame_elems, 4), dtype=torch.float32, device="cuda")
        for _ in range(2):
            rt.render_device(frame.data_ptr())
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        n = 5
        for _ in range(n):
            rt.render_device(frame.data_ptr())
        torch.cuda.synchronize()
        ms = (time.perf_counter() - t0) / n * 1e3
        print(f"{name}: {k} shard(s) on one GPU, tiles of {tile_rows} rows: {ms:.3f} ms per frame, checksum {float(frame[: W * H, :3].sum()):.6g}", flush=True)
