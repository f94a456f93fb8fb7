"""Differential fuzz: grid path vs brute-force path (RT_FLAG_NO_GRID), bit for bit, on random scenes / rays."""
import sys, time, os
ROOT = __import__('pathlib').Path(__file__).resolve().parents[2]; sys.path.insert(0, str(ROOT)); sys.path.insert(0, str(ROOT / 'tests'))
import numpy as np
import _pkg; _pkg.load()
from helpers import R, rotation, instance, same_floats
from opencl_raytracer_amd.hip_raytracer import HIPRaytracer

def scene(rng):
    n = int(rng.choice([100, 300, 1000, 3000]))
    spread = float(rng.choice([2.0, 10.0, 40.0, 300.0]))
    far = float(rng.choice([0.0, 50.0, 1000.0]))            # offset of the whole cloud from the origin
    base = rng.normal(size=3); base = base / np.linalg.norm(base) * far
    smin, smax = [(0.005, 0.05), (0.05, 1.0), (0.5, 5.0)][int(rng.integers(0, 3))]
    aniso = float(rng.choice([1.0, 1.0, 3.0, 30.0]))
    recs = []
    for i in range(n):
        pos = base + rng.uniform(-spread, spread, 3)
        s = rng.uniform(smin, smax)
        sc = (s, s * rng.uniform(1, aniso), s / rng.uniform(1, aniso)) if aniso > 1 else (s, s, s)
        rot = rotation(rng.normal(size=3), rng.uniform(0, 6.3)) if rng.uniform() < 0.7 else None
        mv, inv = instance(pos, rot, sc)
        mat = R.Material(tuple(rng.uniform(0, 1, 3)), tuple(rng.uniform(0, 1, 3)), tuple(rng.uniform(0, 1, 3)),
                         absorption=float(rng.choice([0.2, 0.6, 1.0])), shininess=float(rng.uniform(1, 40)))
        recs.append(R.make_object(R.BOX if rng.uniform() < 0.3 else R.SPHERE, mat, mv, inv))
    objs = R.objects_array(recs)
    nl = int(rng.integers(1, 4))
    lights = R.lights_array([R.make_light(R.LightProperties(tuple(rng.uniform(0, .3, 3)), tuple(rng.uniform(0, .5, 3)), tuple(rng.uniform(0, .5, 3))),
                                          position=(*(base + rng.uniform(-spread, spread, 3) * 1.5), 1.0)) for _ in range(nl)])
    m = 4096
    rays = np.zeros(m, dtype=R.RAY_DTYPE)
    mode = int(rng.integers(0, 3))
    if mode == 0:      # one eye point outside, rays through the cloud
        eye = base + rng.normal(size=3) * spread * 3
        tgt = base + rng.uniform(-spread, spread, (m, 3))
        rays["start"][:, :3] = eye; d = tgt - eye
    elif mode == 1:    # origins all over (inside too), random directions
        o = base + rng.uniform(-spread, spread, (m, 3)) * 1.2
        rays["start"][:, :3] = o; d = rng.normal(size=(m, 3))
    else:              # aimed at object centres with tiny offsets (grazing / through centre)
        k = rng.integers(0, n, m)
        c = objs["mv"].reshape(-1, 4, 4)[k, 3, :3].astype(np.float64)
        o = base + rng.normal(size=(m, 3)) * spread * 2
        rays["start"][:, :3] = o; d = (c - o) + rng.normal(size=(m, 3)) * smin * rng.choice([0.0, 0.5, 1.0, 2.0], size=(m, 1))
    d = d * rng.choice([1e-2, 1.0, 50.0], size=(m, 1))
    rays["direction"][:, :3] = d
    rays["start"][:, 3] = 1.0
    import os
    if os.environ.get("FUZZ_DW"): rays["direction"][:, 3] = float(os.environ["FUZZ_DW"])
    return objs, lights, rays

bad = 0
t0 = time.time()
n_seeds = int(sys.argv[1]) if len(sys.argv) > 1 else 100
first = int(sys.argv[2]) if len(sys.argv) > 2 else 0
for seed in range(first, first + n_seeds):
    rng = np.random.default_rng(1000 + seed)
    objs, lights, rays = scene(rng)
    for kernel, depth in (("hittest", 0), ("shade_and_reflect", 3)):
        res = []
        for grid in (True, False):
            try:
                with HIPRaytracer(objs, lights, rays, depth, kernel=kernel, path="wavefront", grid=grid, fused=not os.environ.get("FUZZ_UNFUSED")) as rt:
                    out = rt.Render(); t, i = rt.render_aux(); st = rt.count_rays()
                    res.append((out.copy(), t.copy(), i.copy(), st.rays_reference, rt.stats().object_tests))
            except Exception as ex:
                res.append(None); print('seed', seed, 'exception', ex)
        a, b = res
        if a is None or b is None: continue
        ok = np.array_equal(a[0].view(np.uint32), b[0].view(np.uint32)) and np.array_equal(a[2], b[2]) and same_floats(a[1], b[1]) and a[3] == b[3]
        if not ok:
            bad += 1
            print('MISMATCH seed', seed, kernel, 'pixels', int(np.any(a[0].reshape(len(rays), -1) != b[0].reshape(len(rays), -1), axis=1).sum()), flush=True)
    if seed % 10 == 9: print('seed', seed, 'done, mismatches so far', bad, f'{time.time()-t0:.0f}s', 'grid tests/brute tests', a[4], b[4], flush=True)
print('TOTAL mismatches', bad)
