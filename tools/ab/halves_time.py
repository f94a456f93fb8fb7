#!/usr/bin/env python3
"""Does a frame finish sooner as P concurrent sub-frames? P contexts, each an interleaved 1/P share of the cfg4 frame
(rt_set_shard), each on its own stream and host thread: while one share sits in wf_resume or in the tail of a persistent
walk, the other shares' kernels can fill the machine. usage: python tools/ab/halves_time.py [workload] [frames]"""
import os, sys, threading, time
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
import torch
import bench
from opencl_raytracer_amd import camera, sharding
from opencl_raytracer_amd.hip_raytracer import HIPRaytracer
wl = sys.argv[1] if len(sys.argv) > 1 else "cfg4"
K = int(sys.argv[2]) if len(sys.argv) > 2 else 6
desc, objs, lights, W, H, kernel, depth = bench.load_workload(wl)
z = float(camera.camera_z(H))
for parts in tuple(int(x) for x in os.environ.get("HALVES_PARTS", "1,2,3,4").split(",")):
    rts, bufs, streams = [], [], []
    for r in range(parts):
        rt = HIPRaytracer(objs, lights, None, depth, kernel=kernel, camera=(W, H, z))
        rt.set_shard(sharding.tile_rays_for_rows(W, 16), r, parts)
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
    run(2)
    t0 = time.perf_counter()
    run(K)
    print(f"{wl} parts {parts}: {(time.perf_counter() - t0) / K * 1e3:7.3f} ms/frame", flush=True)
    for rt in rts: rt.close()
