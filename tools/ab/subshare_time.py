#!/usr/bin/env python3
"""One rank's share of the cfg4 frame at world W, rendered as P concurrent sub-shares (ranks r, r + W, ... of a world of
W * P), each on its own stream and host thread - does a small share hide its launch tails that way?
usage: python tools/ab/subshare_time.py [W] [frames]"""
import os, sys, threading, time
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
import torch
import bench
from opencl_raytracer_amd import camera, sharding
from opencl_raytracer_amd.hip_raytracer import HIPRaytracer
W_ = int(sys.argv[1]) if len(sys.argv) > 1 else 8
K = int(sys.argv[2]) if len(sys.argv) > 2 else 10
desc, objs, lights, W, H, kernel, depth = bench.load_workload("cfg4")
z = float(camera.camera_z(H))
for parts in (1, 2, 3):
    rts, bufs, streams = [], [], []
    for r in range(parts):
        rt = HIPRaytracer(objs, lights, None, depth, kernel=kernel, camera=(W, H, z))
        rt.set_shard(sharding.tile_rays_for_rows(W, 16), r * W_, parts * W_)
        rts.append(rt)
        bufs.append(torch.zeros((rt.local_rays, 4), dtype=torch.float32, device="cuda"))
        streams.append(torch.cuda.Stream())
    def frames(r, k):
        for _ in range(k):
            rts[r].render_device(bufs[r].data_ptr(), streams[r].cuda_stream)
    def run(k):
        th = [threading.Thread(target=frames, args=(r, k)) for r in range(parts)]
        for t in th: t.start()
        for t in th: t.join()
        torch.cuda.synchronize()
    run(3)
    t0 = time.perf_counter()
    run(K)
    print(f"world {W_}, {parts} concurrent sub-share(s): {(time.perf_counter() - t0) / K * 1e3:7.3f} ms/frame-share", flush=True)
    for rt in rts: rt.close()
