"""Fuzz: sharding (interleaved tiles, ragged ends, tile order) and literal-vs-optimised, bit for bit."""
import sys, time
ROOT = __import__('pathlib').Path(__file__).resolve().parents[2]; sys.path.insert(0, str(ROOT)); sys.path.insert(0, str(ROOT / 'tests'))
import numpy as np
import _pkg; _pkg.load()
from helpers import R, rotation, instance, same_floats, random_scene
from opencl_raytracer_amd import camera, sharding
from opencl_raytracer_amd.hip_raytracer import HIPRaytracer
bad = 0
n_seeds = int(sys.argv[1]) if len(sys.argv) > 1 else 100
t0 = time.time()
for seed in range(n_seeds):
    rng = np.random.default_rng(9000 + seed)
    n_s, n_b = [(2, 1), (20, 10), (150, 60), (900, 300)][int(rng.integers(0, 4))]
    objs, lights = random_scene(n_s, n_b, int(rng.integers(1, 5)), seed=int(rng.integers(0, 1 << 30)), spread=float(rng.choice([3.0, 12.0])),
                                zrange=(-40.0, -6.0), directional_lights=int(rng.integers(0, 2)))
    W, H = [(64, 48), (128, 80), (200, 120), (96, 96)][int(rng.integers(0, 4))]
    kernel = ["shade_and_reflect", "shade", "hittest"][int(rng.integers(0, 3))]
    depth = int(rng.integers(0, 4))
    fused = bool(rng.integers(0, 2))
    z = float(camera.camera_z(H))
    pin = bool(rng.integers(0, 2))
    rays = None if pin else camera.primary_rays(W, H)
    kw = dict(camera=(W, H, z)) if pin else dict(raygen=False)
    with HIPRaytracer(objs, lights, rays, depth, kernel=kernel, fused=fused, **kw) as rt:
        full = rt.Render().copy()
    # (a) shards
    world = int(rng.integers(2, 6))
    tile_rows = int(rng.choice([1, 3, 8, 16, 24]))
    tile_rays = W * tile_rows if rng.uniform() < 0.8 else int(rng.integers(17, 999))
    pieces = []
    for rank in range(world):
        with HIPRaytracer(objs, lights, rays, depth, kernel=kernel, fused=fused, **kw) as rt:
            rt.set_shard(tile_rays, rank, world)
            pieces.append(rt.Render().copy())
    import torch
    asm = sharding.assemble_frame([torch.from_numpy(p) for p in pieces], tile_rays, W * H).numpy()
    if not np.array_equal(asm.view(np.uint32).reshape(-1), full.view(np.uint32).reshape(-1)):
        bad += 1
        print('SHARD MISMATCH seed', seed, n_s + n_b, kernel, depth, (W, H), 'pin', pin, 'world', world, 'tile_rays', tile_rays, flush=True)
    # (b) literal
    if n_s + n_b <= 210:
        with HIPRaytracer(objs, lights, rays, depth, kernel=kernel, fused=fused, literal=True, **kw) as rt:
            lit = rt.Render().copy()
        if not np.array_equal(lit.view(np.uint32), full.view(np.uint32)):
            bad += 1
            print('LITERAL MISMATCH seed', seed, n_s + n_b, kernel, depth, (W, H), 'pin', pin, flush=True)
print('TOTAL mismatches', bad, f'{time.time()-t0:.0f}s')
