"""GPU: the closest-hit walk's coarse grid of 32-byte blocks (rt_grid.h: BlockGrid; rt_api.cpp: build_walk_blocks;
rt_wavefront.hip: block_segment) at its edges - every case must give the brute-force loop's (t, index) bit for bit:
 * spheres far larger than a cell next to crowds of tiny ones (lattice scales 1-3 and the whole-cell fallback),
 * cells with many more than seven candidates (chained blocks),
 * caller-made rays the walk is not made for (origins far outside the grid box, directions of 1e-20 / 1e+20 / exactly 0):
   they take the test-everything path inside the same kernel,
 * the block walk against the record walk (RT_WALK3=0) and the round-2 walk (RT_NO_WALK2 + RT_WALK3=0)."""
import numpy as np
import pytest

from helpers import R, camera, instance, rotation, same_floats

pytestmark = pytest.mark.gpu


def hip(*a, **k):
    from opencl_raytracer_amd.hip_raytracer import HIPRaytracer
    return HIPRaytracer(*a, **k)


def _scene(rng, n_small, big, cluster, lights=2):
    objs = []
    mat = lambda: R.Material(ambient=rng.uniform(0, 1, 3), diffuse=rng.uniform(0, 1, 3), specular=rng.uniform(0, 1, 3),
                             absorption=float(rng.choice([1.0, 0.5, 0.2])), reflection=0.0, shininess=float(rng.choice([1.0, 20.0])))
    for k in range(n_small):
        pos = rng.uniform([-30, -30, -90], [30, 30, -25])
        sc = np.full(3, rng.uniform(0.05, 0.6)) if k % 4 else rng.uniform(0.05, 0.6, 3)
        mv, inv = instance(pos, rotation(rng.normal(size=3), rng.uniform(0, 6)), sc)
        objs.append(R.make_object(R.BOX if k % 6 == 0 else R.SPHERE, mat(), mv, inv))
    for k in range(big):  # several cells across
        pos = rng.uniform([-25, -25, -85], [25, 25, -30])
        mv, inv = instance(pos, rotation(rng.normal(size=3), rng.uniform(0, 6)), np.full(3, rng.uniform(3.0, 9.0)))
        objs.append(R.make_object(R.SPHERE if k % 2 else R.BOX, mat(), mv, inv))
    centre = np.array([5.0, -4.0, -50.0])
    for k in range(cluster):  # dozens of objects through one cell
        mv, inv = instance(centre + rng.normal(scale=0.25, size=3), None, np.full(3, rng.uniform(0.2, 0.5)))
        objs.append(R.make_object(R.SPHERE, mat(), mv, inv))
    rng.shuffle(objs)
    props = R.LightProperties((.1, .1, .1), (.5, .5, .5), (.5, .5, .5))
    ls = [R.make_light(props, position=(*rng.uniform([-40, -40, 0], [40, 40, 15]), 1.0)) for _ in range(lights)]
    return R.objects_array(objs), R.lights_array(ls)


def _rays(rng, n, kind):
    rays = np.zeros(n, dtype=R.RAY_DTYPE)
    rays["start"][:, 3] = 1.0
    if kind == "inside":       # origins all over the cloud, every direction
        rays["start"][:, :3] = rng.uniform([-30, -30, -90], [30, 30, -25], (n, 3))
        rays["direction"][:, :3] = rng.normal(size=(n, 3)) * rng.uniform(0.01, 50.0, (n, 1))
    elif kind in ("far", "outside"):  # origins thousands of units away - or ("outside") ten to a thousand cells - aimed at the cloud
        o = rng.normal(size=(n, 3))
        dist = rng.uniform(2.0e3, 4.0e4, (n, 1)) if kind == "far" else np.exp(rng.uniform(np.log(60.0), np.log(3.0e3), (n, 1)))
        o = o / np.linalg.norm(o, axis=1, keepdims=True) * dist + np.array([0, 0, -55.0])
        tgt = rng.uniform([-28, -28, -88], [28, 28, -27], (n, 3))
        rays["start"][:, :3] = o
        rays["direction"][:, :3] = (tgt - o) * rng.uniform(0.5, 2.0, (n, 1))
    elif kind == "scaled":     # sane origins, directions of absurd magnitude (and a few exactly zero)
        rays["start"][:, :3] = rng.uniform([-30, -30, -90], [30, 30, -25], (n, 3))
        d = rng.normal(size=(n, 3))
        mag = rng.choice([1e-20, 1e-17, 1e14, 1e17, 1.0], size=(n, 1))  # (|d|^2 stays finite: a direction whose square overflows, or one
        rays["direction"][:, :3] = d * mag                                 #  that is exactly 0, makes every test of the reference's loop
                                                                            #  produce NaN times - outside every path's domain but the literal one)
    return rays


@pytest.mark.parametrize("kind", ["inside", "far", "outside", "scaled"])
def test_block_walk_equals_brute_force_on_awkward_scenes_and_rays(kind):
    rng = np.random.default_rng({"inside": 5, "far": 6, "scaled": 7, "outside": 8}[kind])
    objs, lights = _scene(rng, 2500, 9, 60)
    rays = _rays(rng, 6000, kind)
    with np.errstate(all="ignore"):
        rays["direction"] = rays["direction"].astype(np.float32)
    got = {}
    for grid in (True, False):
        with hip(objs, lights, rays, 0, kernel="hittest", grid=grid) as rt:
            t, idx = rt.render_aux()
            assert rt.stats().wavefront == 1 and rt.stats().pinhole == 0
        got[grid] = (t, idx)
    assert np.array_equal(got[True][1], got[False][1]), f"{kind}: {(got[True][1] != got[False][1]).sum()} hit indices differ"
    assert same_floats(got[True][0], got[False][0])
    if kind != "scaled":
        assert (got[True][1] >= 0).sum() > 500


def test_three_closest_hit_walks_agree_on_a_reflecting_frame(monkeypatch):
    rng = np.random.default_rng(11)
    objs, lights = _scene(rng, 1800, 4, 40, lights=3)
    rays = camera.primary_rays(160, 120)
    frames = {}
    for name, env in (("blocks", {}), ("records", {"RT_WALK3": "0"}), ("round2", {"RT_WALK3": "0", "RT_WALK2": "none"}), ("brute", None)):
        for k in ("RT_WALK3", "RT_WALK2"):
            monkeypatch.delenv(k, raising=False)
        for k, v in (env or {}).items():
            monkeypatch.setenv(k, v)
        with hip(objs, lights, rays, 4, grid=env is not None) as rt:
            frames[name] = rt.Render()
            st = rt.count_rays()
            frames[name + "_rays"] = (st.rays_reference, st.rays_traced)
    for name in ("records", "round2", "brute"):
        assert np.array_equal(frames["blocks"].view(np.uint32), frames[name].view(np.uint32)), name
        assert frames["blocks_rays"] == frames[name + "_rays"], name


def test_light_tiles_as_records_and_serial_setup_give_the_same_frame(monkeypatch):
    """The last light's tiles exist in two forms: three-candidate blocks (default; the record form is then not even built) and
    records (RT_NO_LT_BLOCKS; also what a table beyond 2^24 blocks falls back to). rt_create builds the blocks on several
    threads (RT_SETUP_THREADS=1: on one). Same frame and same ray counts every way, and as brute force."""
    rng = np.random.default_rng(23)
    objs, lights = _scene(rng, 2500, 4, 40, lights=3)
    W, H = 192, 128
    z = float(camera.camera_z(H))
    frames = {}
    for name, env in (("blocks", {}), ("serial", {"RT_SETUP_THREADS": "1"}), ("threads3", {"RT_SETUP_THREADS": "3"}), ("records", {"RT_NO_LT_BLOCKS": "1"}),
                      ("brute", None)):
        for k in ("RT_SETUP_THREADS", "RT_NO_LT_BLOCKS"):
            monkeypatch.delenv(k, raising=False)
        for k, v in (env or {}).items():
            monkeypatch.setenv(k, v)
        with hip(objs, lights, None, 3, camera=(W, H, z), grid=env is not None) as rt:
            frames[name] = rt.Render()
            st = rt.count_rays()
            frames[name + "_rays"] = (st.rays_reference, st.rays_traced)
    for name in ("serial", "threads3", "records", "brute"):
        assert np.array_equal(frames["blocks"].view(np.uint32), frames[name].view(np.uint32)), name
        assert frames["blocks_rays"] == frames[name + "_rays"], name


def test_meshes_block_walk_equals_the_round2_walk(monkeypatch):
    """Triangles (the extension) have no brute-force path to compare with; their block walk - triangle branch in the exact tests,
    empty-space steps from the block headers, light tiles in block form - is held against round 2's walk (RT_NO_TRI_BLOCKS), which
    tests/test_triangles_gpu.py pins to the CPU statement: same frame, same ray counts, bit for bit."""
    from opencl_raytracer_amd import scene_loader, tessellate
    from helpers import SCENES
    objs, lights = scene_loader.load_scene(str(SCENES / "roundedCube.txt"))
    mesh = tessellate.tessellate(objs, 24, 48, 20)
    rng = np.random.default_rng(23)
    extra = []                                 # analytic spheres / boxes among the triangles
    for k in range(120):
        mv, inv = instance(rng.uniform([-4, -4, -14], [4, 4, -6]), rotation(rng.normal(size=3), rng.uniform(0, 6)), np.full(3, rng.uniform(0.1, 0.4)))
        mat = R.Material(ambient=rng.uniform(0, 1, 3), diffuse=rng.uniform(0, 1, 3), specular=rng.uniform(0, 1, 3),
                         absorption=float(rng.choice([1.0, 0.4])), reflection=0.0, shininess=8.0)
        extra.append(R.make_object(R.BOX if k % 3 == 0 else R.SPHERE, mat, mv, inv))
    extra = R.objects_array(extra)
    scene = np.concatenate([mesh, extra])
    W, H = 192, 144
    z = float(camera.camera_z(H))
    frames = {}
    for name, env in (("blocks", None), ("round2", "1")):
        monkeypatch.delenv("RT_NO_TRI_BLOCKS", raising=False)
        if env:
            monkeypatch.setenv("RT_NO_TRI_BLOCKS", env)
        with hip(scene, lights, None, 5, camera=(W, H, z)) as rt:
            frames[name] = rt.Render()
            st = rt.count_rays()
            assert st.wavefront == 1
            frames[name + "_rays"] = (st.rays_reference, st.rays_traced, st.hit_pixels)
    assert frames["blocks_rays"] == frames["round2_rays"]
    assert frames["blocks_rays"][2] > 2000
    assert np.array_equal(frames["blocks"].view(np.uint32), frames["round2"].view(np.uint32))
