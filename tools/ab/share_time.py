#!/usr/bin/env python3
"""Single-GPU rehearsal of the multi-GPU split: the time rank 0 needs for ITS share of the cfg4 frame at world =
1, 2, 4, 8 (interleaved 16-row tiles, rt_set_shard) - what one GPU can say about scaling before the gather.
usage: python tools/ab/share_time.py [tile_rows]   -> one JSON line"""
import json, os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
import torch
import bench
from opencl_raytracer_amd import camera, sharding
from opencl_raytracer_amd.hip_raytracer import HIPRaytracer
tile_rows = int(sys.argv[1]) if len(sys.argv) > 1 else 16
desc, objs, lights, W, H, kernel, depth = bench.load_workload("cfg4")
z = float(camera.camera_z(H))
res = {"workload": desc, "tile_rows": tile_rows, "library_sha16": bench.library_sha16(), "share_ms": {}}
for world in (1, 2, 4, 8):
    rt = HIPRaytracer(objs, lights, None, depth, kernel=kernel, camera=(W, H, z))
    rt.set_shard(sharding.tile_rays_for_rows(W, tile_rows), 0, world)
    buf = torch.zeros((rt.local_rays, 4), dtype=torch.float32, device="cuda")
    for _ in range(3):
        rt.render_device(buf.data_ptr(), 0)
    torch.cuda.synchronize()
    K = 10
    t0 = time.perf_counter()
    for _ in range(K):
        rt.render_device(buf.data_ptr(), 0)
    torch.cuda.synchronize()
    res["share_ms"][str(world)] = (time.perf_counter() - t0) / K * 1e3
    rt.close()
one = res["share_ms"]["1"]
res["speedup_before_gather"] = {w: one / t for w, t in res["share_ms"].items()}
print(json.dumps(res))
