#!/usr/bin/env python3
"""The cfg4 frame at full size (4096 x 4096, 100 000 spheres, 32 lights, depth 3) by three routes - the round machine, rounds of
{walk, wf_step} (RT_STEP_ROUNDS=1), the whole frame in one launch (RT_FRAME_KERNEL=1) - compared on the device, word for word,
plus the ray counters of each. usage: python tools/ab/frame_identity.py [workload]  -> one JSON line"""
import json, os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
import torch
import bench
from opencl_raytracer_amd import camera
from opencl_raytracer_amd.hip_raytracer import HIPRaytracer

name = sys.argv[1] if len(sys.argv) > 1 else "cfg4"
desc, objs, lights, W, H, kernel, depth = bench.load_workload(name)
z = float(camera.camera_z(H))
routes = {"rounds": {}, "step_rounds": {"RT_STEP_ROUNDS": "1"}, "frame_kernel": {"RT_FRAME_KERNEL": "1"}}
frames, res = {}, {"workload": desc, "library_sha16": bench.library_sha16(), "pixels": W * H, "routes": {}}
for route, env in routes.items():
    for k in ("RT_STEP_ROUNDS", "RT_FRAME_KERNEL"):
        os.environ.pop(k, None)
    os.environ.update(env)
    with HIPRaytracer(objs, lights, None, depth, kernel=kernel, camera=(W, H, z)) as rt:
        out = torch.full((W * H, 4), float("nan"), dtype=torch.float32, device="cuda")
        torch.cuda.synchronize()
        rt.render_device(out.data_ptr(), 0)
        torch.cuda.synchronize()
        st = rt.count_rays()
        frames[route] = out
        res["routes"][route] = {"rays_traced": int(st.rays_traced), "rays_reference": int(st.rays_reference), "hit_pixels": int(st.hit_pixels), "rounds": int(st.rounds)}
base = frames["rounds"].view(torch.int32)
for route in ("step_rounds", "frame_kernel"):
    same = torch.eq(frames[route].view(torch.int32), base).all(dim=1)
    res["routes"][route]["pixels_differing_from_rounds"] = int((~same).sum().item())
res["nan_pixels"] = int(torch.isnan(frames["rounds"]).any(dim=1).sum().item())
print(json.dumps(res))
