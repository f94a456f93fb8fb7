"""GPU: the HIP path (through the C ABI) against the golden vectors of the reference's own kernels and
against the CPU oracle on seeded scenes.

Bars (BASELINE.json north_star): primary hit object index and nearest t exact (helpers.same_floats: exact
float equality, signed zero excepted); shaded RGB within 1e-5 absolute (in practice <= 2 ulp: the only
non-shared arithmetic is powf - OCML on the GPU, libm on the CPU); the reference-equivalent ray count must
equal the oracle's, which pins the hit/miss structure of every secondary ray.
"""
import numpy as np
import pytest

from helpers import (R, camera, compare_frames, count_float_mismatches, expected_full, fixture_names, load_fixture, rotation,
                     random_scene, same_floats)

pytestmark = pytest.mark.gpu

RGB_ATOL = 1e-5  # north_star tolerance for floating-point colour
NAMES = fixture_names()
KNAME = {0: "hittest", 1: "shade", 2: "shade_and_reflect"}


def hip(*a, **k):
    from opencl_raytracer_amd.hip_raytracer import HIPRaytracer
    return HIPRaytracer(*a, **k)


@pytest.mark.parametrize("fused", [True, False], ids=["fused", "unfused"])
@pytest.mark.parametrize("name", NAMES)
def test_golden_vectors(name, fused):
    fx = load_fixture(name)
    want = expected_full(fx, fused)
    # monolithic / wavefront with the conservative grid / wavefront brute force / literal (every reference ray)
    for literal, path, grid in ((False, "monolithic", True), (True, "monolithic", True), (False, "wavefront", True),
                                (False, "wavefront", False), (True, "wavefront", True)):
        with hip(fx["objs"], fx["lights"], fx["rays"], fx["max_bounces"], kernel=fx["kernel"], fused=fused,
                 literal=literal, path=path, grid=grid) as rt:
            got = rt.Render()
            assert rt.stats().wavefront == int(path == "wavefront")
        if fx["kernel"] == 0:
            assert same_floats(got, want), f"{name}: {count_float_mismatches(got, want)} nearest-t values differ"
        else:
            assert got.shape == want.shape
            assert compare_frames(got, want) <= RGB_ATOL
            # hit/miss mask must be identical: background pixels are exactly (0,0,0)
            assert np.array_equal(np.any(got[:, :3] != 0, axis=1), np.any(want[:, :3] != 0, axis=1))
            assert np.all(got[:, 3] == 1.0)


def test_index_recovery_matches_reference_index():
    """The reference's `shade` output in this fixture IS its hit index; compare with the HIP aux index."""
    fx = load_fixture("index_recovery_shade")
    ref_rgb = fx["out_fused"]
    ref_index = np.rint(ref_rgb[:, 0] * 256 + ref_rgb[:, 1] * 65536).astype(np.int64) - 1
    with hip(fx["objs"], fx["lights"], fx["rays"], 0, kernel="shade") as rt:
        _, idx = rt.render_aux()
    assert np.array_equal(idx.astype(np.int64), ref_index)


@pytest.mark.parametrize("fused", [True, False], ids=["fused", "unfused"])
@pytest.mark.parametrize("seed,n_s,n_b,n_l,res", [(201, 30, 20, 3, (64, 48)), (202, 0, 40, 2, (48, 48)),
                                                   (203, 200, 100, 6, (48, 32)), (204, 5, 5, 1, (128, 96))])
def test_random_scenes_vs_oracle(seed, n_s, n_b, n_l, res, fused, restatement):
    objs, lights = random_scene(n_s, n_b, n_l, seed=seed, directional_lights=seed % 2, spread=7.0)
    rays = camera.primary_rays(*res)
    for kernel in ("hittest", "shade", "shade_and_reflect"):
        want = restatement[fused].render(kernel, objs, lights, rays, 3)
        outs = {}
        for literal, raygen, path, grid in ((False, True, "monolithic", True), (False, False, "monolithic", True),
                                            (True, True, "monolithic", True), (True, False, "monolithic", True),
                                            (False, True, "wavefront", True), (False, False, "wavefront", False),
                                            (True, False, "wavefront", True)):
            if True:
                with hip(objs, lights, rays, 3, kernel=kernel, fused=fused, literal=literal, raygen=raygen, path=path,
                         grid=grid) as rt:
                    out = rt.Render()
                    t, idx = rt.render_aux()
                    st = rt.count_rays()
                assert st.pinhole == int(raygen)
                assert np.array_equal(idx, want["hit_index"]), f"{kernel}: primary hit index differs"
                assert same_floats(t, want["hit_t"]), f"{kernel}: primary t differs"
                assert st.rays_reference == want["rays_ref"], f"{kernel}: reference-equivalent ray count differs"
                if literal:
                    assert st.rays_traced == st.rays_reference
                else:
                    assert st.rays_traced <= st.rays_reference
                if kernel == "hittest":
                    assert same_floats(out, want["out"])
                else:
                    assert compare_frames(out, want["out"]) <= RGB_ATOL
                outs[(literal, raygen, path, grid)] = out
        # the exact eliminations, in-kernel ray generation and the wavefront path must not change a single bit
        base = outs[(True, False, "monolithic", True)]
        for key, o in outs.items():
            assert np.array_equal(o.view(np.uint32), base.view(np.uint32)), f"{kernel} {key} differs from literal"


def test_config2_multiple_spheres_1080p_properties(restatement):
    """BASELINE config 2 at full size: 1920x1080 `shade`. Checked against the oracle on sampled rows and
    through size-independent properties (idempotence; shard union == full frame)."""
    from helpers import SCENES
    from opencl_raytracer_amd import scene_loader
    objs, lights = scene_loader.load_scene(str(SCENES / "multipleSpheres.txt"))
    W, H = 1920, 1080
    z = float(camera.camera_z(H))
    with hip(objs, lights, None, 0, kernel="shade", camera=(W, H, z)) as rt:
        a = rt.Render()
        b = rt.Render()
        assert np.array_equal(a.view(np.uint32), b.view(np.uint32))
        full = a.reshape(H, W, 4)
    rows = [0, 1, 269, 270, 539, 540, 541, 700, 1079]
    for r in rows:
        rays = camera.primary_rays(W, H, row_begin=r, row_end=r + 1)
        want = restatement[True].render("shade", objs, lights, rays, 0)["out"]
        assert compare_frames(full[r], want) <= RGB_ATOL
        assert np.array_equal(np.any(full[r][:, :3] != 0, axis=1), np.any(want[:, :3] != 0, axis=1))
    # shards: 2-row tiles over 3 ranks, reassembled, must equal the full frame bit for bit
    tile = 2 * W
    pieces = []
    for rank in range(3):
        with hip(objs, lights, None, 0, kernel="shade", camera=(W, H, z)) as rt:
            rt.set_shard(tile, rank, 3)
            pieces.append(rt.Render())
    from opencl_raytracer_amd.sharding import assemble_frame
    frame = assemble_frame(pieces, tile, W * H)
    assert np.array_equal(frame.view(np.uint32), a.view(np.uint32))


def test_config3_simple_scene_4096_properties(restatement):
    """BASELINE config 3 at full size (4096x4096, shade_and_reflect, D=3): sampled rows against the oracle,
    checksum stability, and buffer-rays == generated-rays on a band."""
    from helpers import SCENES
    from opencl_raytracer_amd import scene_loader
    objs, lights = scene_loader.load_scene(str(SCENES / "simpleScene.txt"))
    W = H = 4096
    z = float(camera.camera_z(H))
    with hip(objs, lights, None, 3, kernel="shade_and_reflect", camera=(W, H, z)) as rt:
        full = rt.Render().reshape(H, W, 4)
        st = rt.count_rays()
    assert st.rays_reference >= W * H
    hit_rows = [1500, 2047, 2048, 2049, 2600]
    for r in [0, 4095] + hit_rows:
        rays = camera.primary_rays(W, H, row_begin=r, row_end=r + 1)
        want = restatement[True].render("shade_and_reflect", objs, lights, rays, 3)["out"]
        assert compare_frames(full[r], want) <= RGB_ATOL
        assert np.array_equal(np.any(full[r][:, :3] != 0, axis=1), np.any(want[:, :3] != 0, axis=1))
    assert np.any(full[2048][:, :3] != 0)
    # uploaded ray buffer (no in-kernel generation) on a 64-row band
    band = camera.primary_rays(W, H, row_begin=2016, row_end=2080)
    with hip(objs, lights, band, 3, kernel="shade_and_reflect", raygen=False) as rt:
        got = rt.Render().reshape(64, W, 4)
    assert np.array_equal(got.view(np.uint32), full[2016:2080].view(np.uint32))


def test_empty_and_ragged_inputs():
    objs, lights = random_scene(3, 3, 2, seed=5)
    # zero rays
    with hip(objs, lights, camera.primary_rays(8, 8)[:0], 2) as rt:
        assert rt.Render().shape == (0, 4)
    # no objects, no lights
    rays = camera.primary_rays(33, 7)  # 231 rays: not a multiple of the 256-thread workgroup
    with hip(R.objects_array([]), R.lights_array([]), rays, 2) as rt:
        out = rt.Render()
        assert out.shape == (231, 4) and np.all(out[:, :3] == 0) and np.all(out[:, 3] == 1)
    with hip(objs, R.lights_array([]), rays, 2) as rt:
        out = rt.Render()
        assert np.all(out[:, :3] == 0)
    # ragged shard: 231 rays in tiles of 50 over 2 ranks
    from opencl_raytracer_amd.sharding import assemble_frame
    with hip(objs, lights, rays, 2) as rt:
        full = rt.Render()
    pieces = []
    for rank in range(2):
        with hip(objs, lights, rays, 2) as rt:
            rt.set_shard(50, rank, 2)
            pieces.append(rt.Render())
    assert np.array_equal(assemble_frame(pieces, 50, 231).view(np.uint32), full.view(np.uint32))


def test_errors_are_reported_not_thrown():
    from opencl_raytracer_amd.hip_raytracer import RTError
    objs, lights = random_scene(1, 1, 1, seed=1)
    with pytest.raises(RTError):
        hip(objs, lights, camera.primary_rays(4, 4), 1, kernel=7)
    with pytest.raises(RTError):
        hip(objs, lights, None, 1, camera=(4, 4, -3.0)).set_shard(0, 0, 2)
    with pytest.raises(RTError):
        hip(objs, lights, camera.primary_rays(4, 4), 1, device=99)


def test_wavefront_equals_monolithic_on_a_larger_scene(restatement):
    """600 objects (above the automatic wavefront threshold), 4 lights, depth 3: auto path == wavefront ==
    monolithic bit for bit, shards included, and all match the oracle."""
    from opencl_raytracer_amd import synthetic
    from opencl_raytracer_amd.sharding import assemble_frame
    objs, lights = synthetic.spheres_and_lights(600, 4, absorption=0.5)
    objs["mvInverse"][:, 0] *= 0.25   # spheres 4x larger so that most rays hit and reflect
    objs["mvInverse"][:, 5] *= 0.25
    objs["mvInverse"][:, 10] *= 0.25
    objs["mvInverse"][:, 12:15] *= 0.25
    objs["mv"][:, 0] *= 4
    objs["mv"][:, 5] *= 4
    objs["mv"][:, 10] *= 4
    rays = camera.primary_rays(96, 64)
    want = restatement[True].render("shade_and_reflect", objs, lights, rays, 3)
    outs = {}
    for path in ("auto", "wavefront", "monolithic"):
        with hip(objs, lights, rays, 3, path=path) as rt:
            outs[path] = rt.Render()
            st = rt.count_rays()
            assert st.wavefront == int(path != "monolithic")
            assert st.rays_reference == want["rays_ref"]
            if path != "monolithic":
                assert 2 <= st.rounds <= 64
    assert np.array_equal(outs["auto"].view(np.uint32), outs["monolithic"].view(np.uint32))
    assert np.array_equal(outs["wavefront"].view(np.uint32), outs["monolithic"].view(np.uint32))
    assert compare_frames(outs["auto"], want["out"]) <= RGB_ATOL
    assert int((want["hit_index"] >= 0).sum()) > 2000
    pieces = []
    for rank in range(3):
        with hip(objs, lights, rays, 3, path="wavefront") as rt:
            rt.set_shard(96 * 4, rank, 3)
            pieces.append(rt.Render())
    assert np.array_equal(assemble_frame(pieces, 96 * 4, len(rays)).view(np.uint32), outs["monolithic"].view(np.uint32))


def test_large_scene_sliced_shadow_stream_vs_oracle(restatement):
    """20 000 spheres, 3 lights: the wavefront path with the size-sorted, sliced shadow stream (4 slices) and
    the L2-warmed pair traversal, against the oracle and against the monolithic kernel."""
    from opencl_raytracer_amd import synthetic
    objs, lights = synthetic.spheres_and_lights(20000, 3, absorption=0.6)
    rays = camera.crop_rays(512, 512, 240, 240, 32, 32)
    want = restatement[True].render("shade_and_reflect", objs, lights, rays, 2)
    assert int((want["hit_index"] >= 0).sum()) > 500
    outs = {}
    for path, literal, grid in (("auto", False, True), ("auto", False, False), ("auto", True, True), ("monolithic", False, True)):
        with hip(objs, lights, rays, 2, path=path, literal=literal, grid=grid) as rt:
            outs[(path, literal, grid)] = rt.Render()
            t, idx = rt.render_aux()
            st = rt.count_rays()
        assert st.wavefront == int(path == "auto")
        assert np.array_equal(idx, want["hit_index"]) and same_floats(t, want["hit_t"])
        assert st.rays_reference == want["rays_ref"]
        assert compare_frames(outs[(path, literal, grid)], want["out"]) <= RGB_ATOL
    base = outs[("monolithic", False, True)]
    for o in outs.values():
        assert np.array_equal(o.view(np.uint32), base.view(np.uint32))
    # the shade kernel (every light's shadow ray, summed) through the same machinery
    want1 = restatement[True].render("shade", objs, lights, rays, 0)
    with hip(objs, lights, rays, 0, kernel="shade") as rt:
        got1 = rt.Render()
        assert rt.count_rays().rays_reference == want1["rays_ref"]
    assert compare_frames(got1, want1["out"]) <= RGB_ATOL


def test_config4_scene_window_properties(restatement):
    """BASELINE config 4's scene (100 000 spheres, 32 lights, depth 3) on windows of its 4096x4096 ray grid:
    an 8x8 block against the oracle, window-in-window equality, idempotence and shard union."""
    from opencl_raytracer_amd import synthetic
    from opencl_raytracer_amd.sharding import assemble_frame
    objs, lights = synthetic.spheres_and_lights(100_000, 32)
    W = H = 4096
    big = camera.crop_rays(W, H, 2048 - 64, 2048 - 64, 128, 128)
    with hip(objs, lights, big, 3) as rt:
        a = rt.Render()
        b = rt.Render()
        st = rt.count_rays()
    assert st.wavefront == 1
    assert np.array_equal(a.view(np.uint32), b.view(np.uint32))
    # brute force over all 100 000 objects (no grid culling): the same bits, and it counts its ray-object tests
    with hip(objs, lights, big, 3, grid=False) as rt:
        brute = rt.Render()
        stb = rt.count_rays()
    assert np.array_equal(brute.view(np.uint32), a.view(np.uint32))
    assert stb.object_tests > 100_000 * 128 * 128 and stb.rays_reference == st.rays_reference and stb.rays_traced == st.rays_traced
    a = a.reshape(128, 128, 4)
    # the inner 32x32 window rendered on its own: same pixels, bit for bit
    small = camera.crop_rays(W, H, 2048 - 16, 2048 - 16, 32, 32)
    with hip(objs, lights, small, 3) as rt:
        s = rt.Render().reshape(32, 32, 4)
    assert np.array_equal(s.view(np.uint32), a[48:80, 48:80].view(np.uint32))
    # an 8x8 block against the oracle
    blk = camera.crop_rays(W, H, 2048 - 4, 2048 - 4, 8, 8)
    want = restatement[True].render("shade_and_reflect", objs, lights, blk, 3)
    got = a[60:68, 60:68].reshape(64, 4)
    assert compare_frames(got, want["out"]) <= RGB_ATOL
    assert int((want["hit_index"] >= 0).sum()) == 64
    # shards of the window (2 ranks, 16-row tiles) reassemble to the same frame
    pieces = []
    for rank in range(2):
        with hip(objs, lights, big, 3) as rt:
            rt.set_shard(128 * 16, rank, 2)
            pieces.append(rt.Render())
    assert np.array_equal(assemble_frame(pieces, 128 * 16, 128 * 128).view(np.uint32), a.reshape(-1, 4).view(np.uint32))


def _grid_equals_brute(objs, lights, rays, depth, kernels=("shade_and_reflect",)):
    for kernel in kernels:
        with hip(objs, lights, rays, depth, kernel=kernel, path="wavefront", grid=True) as rt:
            a = rt.Render()
            ta, ia = rt.render_aux()
            sa = rt.count_rays()
        with hip(objs, lights, rays, depth, kernel=kernel, path="wavefront", grid=False) as rt:
            b = rt.Render()
            tb, ib = rt.render_aux()
            sb = rt.count_rays()
        assert np.array_equal(a.view(np.uint32), b.view(np.uint32)), f"{kernel}: grid and brute-force frames differ"
        assert np.array_equal(ia, ib) and same_floats(ta, tb)
        assert sa.rays_reference == sb.rays_reference and sa.rays_traced == sb.rays_traced
    return a


def test_grid_mixed_boxes_and_spheres(restatement):
    """2 000 rotated, non-uniformly scaled boxes and spheres, positional + directional lights, reflective materials:
    grid == brute force bit for bit, and both match the oracle."""
    objs, lights = random_scene(1200, 800, 4, seed=909, directional_lights=1, spread=14.0, zrange=(-60.0, -12.0))
    rays = camera.primary_rays(96, 64)
    got = _grid_equals_brute(objs, lights, rays, 3, kernels=("hittest", "shade", "shade_and_reflect"))
    want = restatement[True].render("shade_and_reflect", objs, lights, rays, 3)
    assert compare_frames(got, want["out"]) <= RGB_ATOL
    assert int((want["hit_index"] >= 0).sum()) > 3000


def test_grid_tiny_far_objects_numerical_fuzz():
    """Radius-0.01 spheres 60-140 units away: |o| ~ 1e4 object units, where the reference's own discriminant
    accepts rays that miss the sphere by many radii. The grid margins must keep every such false hit."""
    rng = np.random.default_rng(4242)
    n = 3000
    from helpers import instance
    recs = []
    for i in range(n):
        pos = (rng.uniform(-40, 40), rng.uniform(-40, 40), rng.uniform(-140, -60))
        r = rng.choice([0.01, 0.02, 0.05])
        mv, inv = instance(pos, None, (r, r, r))
        recs.append(R.make_object(R.SPHERE, R.Material((rng.uniform(), rng.uniform(), rng.uniform()), (.5, .5, .5), (.5, .5, .5),
                                                       absorption=0.5), mv, inv))
    objs = R.objects_array(recs)
    lights = R.lights_array([R.make_light(R.LightProperties((.3, .3, .3), (.7, .7, .7), (1, 1, 1)), position=(10, 10, 0, 1)),
                             R.make_light(R.LightProperties((.1, .1, .1), (.4, .4, .4), (.5, .5, .5)), position=(-30, 5, -100, 1))])
    # aim a dense bundle of rays at the cloud so that plenty of them graze spheres
    rays = camera.crop_rays(4096, 4096, 1536, 1536, 256, 256)
    out = _grid_equals_brute(objs, lights, rays, 2)
    assert int(np.any(out[:, :3] != 0, axis=1).sum()) > 50


def _instanced(rng, n, scales, zrange, spread, box_every=3):
    from helpers import instance
    recs = []
    for i in range(n):
        pos = (rng.uniform(-spread, spread), rng.uniform(-spread, spread), rng.uniform(*zrange))
        rot = rotation(rng.normal(size=3), rng.uniform(0, 2 * np.pi))
        mv, inv = instance(pos, rot, scales(rng))
        mat = R.Material((rng.uniform(), rng.uniform(), rng.uniform()), (.5, .5, .5), (.5, .5, .5),
                         absorption=float(rng.choice([0.3, 0.6, 1.0])), shininess=float(rng.uniform(1, 30)))
        recs.append(R.make_object(R.BOX if box_every and i % box_every == 0 else R.SPHERE, mat, mv, inv))
    return R.objects_array(recs)


def test_grid_strongly_anisotropic_objects():
    """Needles and pancakes (axis ratios up to 150, rotated): the condition number enters the grid's error bound
    squared, objects with kappa^2 > 4 carry their full registration radius in the pre-test (negative w), the
    rest use the distance-dependent form - both must keep every hit the reference's fp32 test reports."""
    rng = np.random.default_rng(777)
    def scales(rng):
        kind = rng.integers(0, 4)
        if kind == 0: return (rng.uniform(0.01, 0.03), rng.uniform(1.0, 3.0), rng.uniform(0.2, 0.5))   # needle-ish plate
        if kind == 1: return (rng.uniform(0.5, 1.5), rng.uniform(0.5, 1.5), rng.uniform(0.01, 0.02))    # pancake
        if kind == 2: return (rng.uniform(0.3, 0.5),) * 3                                             # isotropic
        return (rng.uniform(0.2, 0.4), rng.uniform(0.4, 0.8), rng.uniform(0.3, 0.5))                   # mild (kappa^2 <= 4)
    objs = _instanced(rng, 1500, scales, (-70.0, -15.0), 18.0)
    lights = R.lights_array([R.make_light(R.LightProperties((.2, .2, .2), (.6, .6, .6), (.8, .8, .8)), position=(15, 20, 5, 1)),
                             R.make_light(R.LightProperties((.1, .1, .1), (.4, .4, .4), (.5, .5, .5)), position=(-25, -5, -40, 1))])
    rays = camera.primary_rays(128, 96)
    out = _grid_equals_brute(objs, lights, rays, 3, kernels=("hittest", "shade_and_reflect"))
    assert int(np.any(out[:, :3] != 0, axis=1).sum()) > 2000


def test_grid_grazing_rays():
    """Rays aimed at the silhouettes of spheres from far away, offset from the tangent by -4..+4 parts per million
    of the radius - the rays the reference's discriminant decides by rounding. Nearest t and index must be
    those of the brute-force loop for every one of them."""
    rng = np.random.default_rng(2718)
    objs = _instanced(rng, 600, lambda r: (r.choice([0.05, 0.2, 0.5]),) * 3, (-110.0, -20.0), 30.0, box_every=0)
    centres = objs["mv"].reshape(-1, 4, 4)[:, 3, :3].astype(np.float64)       # column-major: translation = column 3
    radii = np.linalg.norm(objs["mv"].reshape(-1, 4, 4)[:, 0, :3].astype(np.float64), axis=1)
    rays = np.zeros(600 * 64, dtype=R.RAY_DTYPE)
    k = 0
    for i in range(600):
        for j in range(64):
            origin = np.array([rng.uniform(-30, 30), rng.uniform(-30, 30), rng.uniform(-5, 5)]) if j % 2 else np.zeros(3)
            to_c = centres[i] - origin
            dist = np.linalg.norm(to_c)
            side = np.cross(to_c, rng.normal(size=3)); side /= np.linalg.norm(side)
            off = radii[i] * (1.0 + rng.integers(-4, 5) * 1e-6)
            # tangent point of a cone around to_c: the line passes the centre at distance `off`
            sin_a = off / dist
            d = to_c / dist * np.sqrt(max(1.0 - sin_a * sin_a, 0.0)) + side * sin_a
            rays["start"][k] = (*origin, 1.0)
            rays["direction"][k] = (*(d * rng.uniform(0.5, 300.0)), 0.0)
            k += 1
    lights = R.lights_array([R.make_light(R.LightProperties((.2, .2, .2), (.6, .6, .6), (.8, .8, .8)), position=(0, 30, 0, 1))])
    with hip(objs, lights, rays, 1, kernel="hittest", path="wavefront", grid=True) as rt:
        a = rt.Render()
    with hip(objs, lights, rays, 1, kernel="hittest", path="wavefront", grid=False) as rt:
        b = rt.Render()
    assert same_floats(a, b)
    hits = int((a < 3.0e38).sum())
    assert 0.2 * len(rays) < hits < len(rays)      # a healthy mix of both outcomes
    _grid_equals_brute(objs, lights, rays[: 600 * 16], 2)


def test_grid_ticketed_runs_on_a_long_queue():
    """More rays than the resident waves can take one run each (> 8192 x 128): the persistent waves draw runs from
    the ticket counter. Same frame as the brute-force path, bit for bit."""
    from opencl_raytracer_amd import synthetic
    objs, lights = synthetic.spheres_and_lights(1500, 3, absorption=0.5, seed=99)
    rays = camera.crop_rays(4096, 4096, 1400, 1500, 1280, 1024)   # 1 310 720 rays
    assert len(rays) > 8192 * 128
    _grid_equals_brute(objs, lights, rays, 2)


def test_grid_rays_starting_inside_objects_and_lights_inside_the_cloud():
    objs, lights = random_scene(700, 300, 3, seed=1234, spread=5.0, zrange=(-14.0, 4.0))  # the camera sits inside the cloud
    lights["position"][0] = (0.5, -0.5, -6.0, 1.0)                                          # a light in the middle of it
    rays = camera.primary_rays(80, 60)
    _grid_equals_brute(objs, lights, rays, 4, kernels=("shade", "shade_and_reflect"))


def test_device_resident_render_through_torch_plumbing():
    """rt_render_device into a torch tensor on torch's current stream (what bench.py and the multi-GPU path use)
    gives the same bytes as the host-buffer Render()."""
    import torch
    from opencl_raytracer_amd.distributed import ShardedHIPRaytracer
    objs, lights = random_scene(10, 6, 2, seed=66)
    W, H = 128, 80
    z = float(camera.camera_z(H))
    with hip(objs, lights, None, 3, camera=(W, H, z)) as rt:
        want = rt.Render()
    srt = ShardedHIPRaytracer(objs, lights, None, 3, camera=(W, H, z), tile_rows=16)
    frame = srt.Render()
    torch.cuda.synchronize()
    got = frame.cpu().numpy()
    srt.close()
    assert got.shape == want.shape and np.array_equal(got.view(np.uint32), want.view(np.uint32))


def test_grid_always_list_and_fallbacks(restatement):
    """Objects as large as the scene go to the grid's always-tested list; non-affine matrices or primary rays with
    start.w != 1 switch the large-scene path back to brute force. All of them must match the oracle."""
    from helpers import instance
    objs, lights = random_scene(500, 200, 3, seed=555, spread=10.0, zrange=(-50.0, -10.0))
    huge = []
    mv, inv = instance((0, -40, -30), None, (60, 60, 60))        # a sphere bigger than the whole cloud below it
    huge.append(R.make_object(R.SPHERE, R.Material((.2, .2, .2), (.5, .5, .5), (.3, .3, .3), absorption=0.6), mv, inv))
    mv, inv = instance((0, 0, -75), rotation((0, 1, 0), 0.3), (120, 120, 4))  # a wall behind everything
    huge.append(R.make_object(R.BOX, R.Material((.1, .2, .3), (.4, .4, .4), (.6, .6, .6), absorption=0.4), mv, inv))
    objs = np.concatenate([objs[:350], R.objects_array(huge), objs[350:]])
    rays = camera.primary_rays(96, 64)
    want = restatement[True].render("shade_and_reflect", objs, lights, rays, 3)
    got = _grid_equals_brute(objs, lights, rays, 3)
    assert compare_frames(got, want["out"]) <= RGB_ATOL
    assert int((want["hit_index"] >= 0).sum()) == len(rays)  # the wall catches every ray

    # non-affine: perturb the bottom row of one mvInverse -> grid must not be used, results still the oracle's
    odd = objs.copy()
    odd["mvInverse"][7][15] = np.float32(1.0000001)
    want = restatement[True].render("shade_and_reflect", odd, lights, rays, 2)
    with hip(odd, lights, rays, 2) as rt:
        out = rt.Render()
        assert rt.stats().wavefront == 1
        st = rt.count_rays()
    assert st.object_tests >= st.rays_traced * 100  # brute force: every ray saw hundreds of objects
    assert compare_frames(out, want["out"]) <= RGB_ATOL and st.rays_reference == want["rays_ref"]

    # primary rays with start.w != 1
    wrays = rays.copy()
    wrays["start"][::3, 3] = 0.75
    want = restatement[True].render("shade_and_reflect", objs, lights, wrays, 2)
    with hip(objs, lights, wrays, 2) as rt:
        out = rt.Render()
        t, idx = rt.render_aux()
    assert np.array_equal(idx, want["hit_index"]) and same_floats(t, want["hit_t"])
    assert compare_frames(out, want["out"]) <= RGB_ATOL


def test_config5_base_rounded_cube_8192_properties(restatement):
    """The analytic base of BASELINE config 5 at its full size: roundedCube.txt, 8192x8192, depth 5 (67 M pixels,
    1 GiB framebuffer, real multi-bounce paths through a mirror box). Sampled rows against the oracle, idempotent."""
    import torch
    from helpers import SCENES
    from opencl_raytracer_amd import scene_loader
    objs, lights = scene_loader.load_scene(str(SCENES / "roundedCube.txt"))
    W = H = 8192
    z = float(camera.camera_z(H))
    with hip(objs, lights, None, 5, camera=(W, H, z)) as rt:
        out = torch.empty((W * H, 4), dtype=torch.float32, device="cuda")
        rt.render_device(out.data_ptr())
        torch.cuda.synchronize()
        first = out.view(H, W, 4)[[0, 3000, 4096, 4097, 5200, 8191]].cpu().numpy()
        checksum1 = out.view(torch.int32).sum(dtype=torch.int64).item()
        rt.render_device(out.data_ptr())
        torch.cuda.synchronize()
        assert out.view(torch.int32).sum(dtype=torch.int64).item() == checksum1  # same bits again
        st = rt.count_rays()
    assert st.hit_pixels > 10_000_000 and st.rays_reference > st.hit_pixels
    for k, r in enumerate([0, 3000, 4096, 4097, 5200, 8191]):
        rays = camera.primary_rays(W, H, row_begin=r, row_end=r + 1)
        want = restatement[True].render("shade_and_reflect", objs, lights, rays, 5)["out"]
        assert compare_frames(first[k], want) <= RGB_ATOL
        assert np.array_equal(np.any(first[k][:, :3] != 0, axis=1), np.any(want[:, :3] != 0, axis=1))


@pytest.mark.parametrize("seed", list(range(300, 324)))
def test_randomised_configurations_vs_oracle(seed, restatement):
    """Seeded sweep over scene size, primitive mix, light count / kind, depth, kernel, arithmetic flavour, path and
    ray source; every combination against the oracle (index, t, ray accounting exact; RGB <= 1e-5)."""
    rng = np.random.default_rng(seed)
    n_s, n_b = int(rng.integers(0, 30)), int(rng.integers(0, 20))
    if n_s + n_b == 0:
        n_s = 1
    n_l = int(rng.integers(0, 5))
    objs, lights = random_scene(n_s, n_b, n_l, seed=seed, directional_lights=int(rng.integers(0, n_l + 1)) if n_l else 0,
                                spread=float(rng.uniform(3, 9)), nonuniform=bool(rng.integers(0, 2)))
    w, h = int(rng.integers(9, 70)), int(rng.integers(5, 40))
    depth = int(rng.integers(0, 6))
    kernel = ["hittest", "shade", "shade_and_reflect"][int(rng.integers(0, 3))]
    fused = bool(rng.integers(0, 2))
    path = ["monolithic", "wavefront"][int(rng.integers(0, 2))]
    rays = camera.primary_rays(w, h)
    if rng.integers(0, 3) == 0:  # arbitrary ray list instead of the pinhole grid
        rays = rays.copy()
        rays["start"][:, :3] = rng.uniform(-1, 1, (len(rays), 3)).astype(np.float32)
    want = restatement[fused].render(kernel, objs, lights, rays, depth)
    with hip(objs, lights, rays, depth, kernel=kernel, fused=fused, path=path, literal=bool(rng.integers(0, 4) == 0),
             grid=bool(rng.integers(0, 4) != 0)) as rt:
        out = rt.Render()
        t, idx = rt.render_aux()
        st = rt.count_rays()
    assert np.array_equal(idx, want["hit_index"]) and same_floats(t, want["hit_t"])
    assert st.rays_reference == want["rays_ref"]
    if kernel == "hittest":
        assert same_floats(out, want["out"])
    else:
        assert compare_frames(out, want["out"]) <= RGB_ATOL


def test_screen_tiles_fall_back_to_wide_tiles_for_objects_that_cover_the_screen():
    """Thirty spheres that each cover ~15 % of a 512 x 512 frame: binned into 8 x 8-pixel screen tiles they exceed the
    (object, tile) budget, binned into 64 x 8 tiles they do not - the first round must still go through tile lists (the
    wide ones) and give the brute-force picture; the same scene in a frame whose height is not a multiple of 8 walks the
    frame row by row (64 x 8 tiles from the start)."""
    from helpers import instance
    rng = np.random.default_rng(4242)
    objs = []
    for k in range(30):
        mat = R.Material((.2, .3, .4), (.5, .5, .5), (.4, .4, .4), absorption=float(rng.choice([0.3, 0.7])), reflection=0.5, shininess=5.0)
        pos = (float(rng.uniform(-14, 14)), float(rng.uniform(-14, 14)), float(rng.uniform(-60, -30)))
        mv, inv = instance(pos, None, (9.0, 9.0, 9.0))
        objs.append(R.make_object(R.SPHERE, mat, mv, inv))
    objs = R.objects_array(objs)
    lights = R.lights_array([R.make_light(R.LightProperties((.2, .2, .2), (.6, .6, .6), (.5, .5, .5)), position=(30.0, 40.0, 10.0, 1.0))])
    for W, H in ((512, 512), (512, 508)):
        z = float(camera.camera_z(H))
        res = {}
        for grid in (True, False):
            with hip(objs, lights, None, 3, camera=(W, H, z), grid=grid, path="wavefront") as rt:
                out = rt.Render()
                t, idx = rt.render_aux()
                st = rt.count_rays()
                res[grid] = (out.view(np.uint32).copy(), t.view(np.uint32).copy(), idx.copy(), st.rays_reference, st.hit_pixels, st.object_tests)
        a, b = res[True], res[False]
        assert np.array_equal(a[0], b[0]) and np.array_equal(a[1], b[1]) and np.array_equal(a[2], b[2]) and a[3:5] == b[3:5]
        assert a[4] > W * H // 4          # the spheres do cover a good part of the frame
        assert a[5] < b[5] // 2           # ... and the grid path went through lists, not through every object


def test_config4_full_frame_grid_equals_brute_force():
    """BASELINE configs[3] at full size (100 000 spheres, 32 lights, 4096 x 4096, depth 3): the default (grid) path
    and the brute-force path - which tests every object for every one of the 109 M rays - produce the same 16.7 M
    pixels bit for bit, trace the same number of rays and agree on every primary hit."""
    from opencl_raytracer_amd import synthetic
    objs, lights = synthetic.spheres_and_lights(100_000, 32)
    W = H = 4096
    z = float(camera.camera_z(H))
    res = {}
    for grid in (True, False):
        with hip(objs, lights, None, 3, camera=(W, H, z), grid=grid) as rt:
            out = rt.Render()
            t, idx = rt.render_aux()
            st = rt.count_rays()
            res[grid] = (out.view(np.uint32).copy(), t.view(np.uint32).copy(), idx.copy(), st.rays_reference, st.rays_traced, st.hit_pixels)
    a, b = res[True], res[False]
    assert np.array_equal(a[0], b[0]) and np.array_equal(a[1], b[1]) and np.array_equal(a[2], b[2])
    assert a[3:] == b[3:] and a[3] == 1690436982 and a[5] == 16694242
