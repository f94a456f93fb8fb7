"""Fuzz: pinhole frames. (a) small scenes: monolithic kernel with per-bundle screen culling vs wavefront brute force;
(b) large scenes: screen-tile primaries + grid vs brute force. Bit for bit."""
import sys, time
ROOT = __import__('pathlib').Path(__file__).resolve().parents[2]; sys.path.insert(0, str(ROOT)); sys.path.insert(0, str(ROOT / 'tests'))
import numpy as np
import _pkg; _pkg.load()
from helpers import R, rotation, instance, same_floats
from opencl_raytracer_amd import camera
from opencl_raytracer_amd.hip_raytracer import HIPRaytracer

def scene(rng, n):
    zc = float(rng.choice([-3.0, -10.0, -40.0, -200.0]))
    spread = float(rng.choice([1.0, 5.0, 30.0]))
    smin, smax = [(0.01, 0.1), (0.1, 1.0), (1.0, 8.0)][int(rng.integers(0, 3))]
    aniso = float(rng.choice([1.0, 1.0, 4.0]))
    recs = []
    for i in range(n):
        pos = np.array([rng.uniform(-spread, spread), rng.uniform(-spread, spread), zc + rng.uniform(-spread, spread)])
        if rng.uniform() < 0.1: pos[2] = rng.uniform(-1.0, 3.0)      # around / behind the camera plane
        s = rng.uniform(smin, smax)
        sc = (s, s * rng.uniform(1, aniso), s / rng.uniform(1, aniso)) if aniso > 1 else (s, s, s)
        rot = rotation(rng.normal(size=3), rng.uniform(0, 6.3)) if rng.uniform() < 0.7 else None
        mv, inv = instance(pos, rot, sc)
        mat = R.Material(tuple(rng.uniform(0, 1, 3)), tuple(rng.uniform(0, 1, 3)), tuple(rng.uniform(0, 1, 3)),
                         absorption=float(rng.choice([0.2, 0.6, 1.0])), shininess=float(rng.uniform(1, 40)))
        recs.append(R.make_object(R.BOX if rng.uniform() < 0.4 else R.SPHERE, mat, mv, inv))
    objs = R.objects_array(recs)
    lights = R.lights_array([R.make_light(R.LightProperties(tuple(rng.uniform(0, .3, 3)), tuple(rng.uniform(0, .5, 3)), tuple(rng.uniform(0, .5, 3))),
                                          position=(rng.uniform(-20, 20), rng.uniform(-20, 20), rng.uniform(-50, 20), 1.0)) for _ in range(int(rng.integers(1, 4)))])
    return objs, lights

bad = 0
n_seeds = int(sys.argv[1]) if len(sys.argv) > 1 else 100
t0 = time.time()
for seed in range(n_seeds):
    rng = np.random.default_rng(7000 + seed)
    small = seed % 2 == 0
    n = int(rng.choice([1, 3, 9, 30, 64])) if small else int(rng.choice([100, 400, 2000]))
    objs, lights = scene(rng, n)
    W, H = [(64, 64), (128, 72), (256, 128), (192, 200)][int(rng.integers(0, 4))]
    fov = float(rng.choice([20.0, 60.0, 120.0]))
    z = float(-(H / 2) / np.tan(np.radians(fov) / 2))
    kernel = ["shade_and_reflect", "shade", "hittest"][int(rng.integers(0, 3))]
    depth = int(rng.integers(0, 4))
    res = []
    for mode in (0, 1):
        kw = dict(path="monolithic") if (small and mode == 0) else (dict(path="wavefront", grid=True) if mode == 0 else dict(path="wavefront", grid=False))
        with HIPRaytracer(objs, lights, None, depth, camera=(W, H, z), kernel=kernel, **kw) as rt:
            out = rt.Render().copy(); t, i = rt.render_aux(); st = rt.count_rays()
            res.append((out, t.copy(), i.copy(), st.rays_reference))
    a, b = res
    ok = np.array_equal(a[0].view(np.uint32), b[0].view(np.uint32)) and np.array_equal(a[2], b[2]) and same_floats(a[1], b[1]) and a[3] == b[3]
    if not ok:
        bad += 1
        print('MISMATCH seed', seed, 'small' if small else 'large', n, kernel, depth, (W, H), fov, 'pixels', int(np.any(a[0].reshape(W * H, -1) != b[0].reshape(W * H, -1), axis=1).sum()), 'idx', int((a[2] != b[2]).sum()), flush=True)
print('TOTAL mismatches', bad, f'{time.time()-t0:.0f}s')
