"""CPU: the CPURaytracer backend (host/CPURaytracer.cpp) against the golden vectors of the reference's
own kernels (fused flavour: the backend contracts multiply-adds where the OpenCL front-end does).

Bar: exact float equality on every fixture - geometry and colour (both sides use the host's libm powf). The backend
shares no code with oracle/; it is compared with the reference's outputs, and with the oracle only for its ray count."""
import numpy as np
import pytest

from helpers import camera, count_float_mismatches, fixture_names, load_fixture, random_scene, same_floats
from opencl_raytracer_amd.cpu_raytracer import CPURaytracer

NAMES = fixture_names()


@pytest.mark.parametrize("name", NAMES)
def test_cpu_backend_matches_reference_golden(name):
    fx = load_fixture(name)
    rt = CPURaytracer(fx["objs"], fx["lights"], fx["rays"], fx["max_bounces"], kernel=fx["kernel"], threads=4)
    out = rt.Render()
    got = out if fx["kernel"] == 0 else np.ascontiguousarray(out[:, :3])
    want = fx["out_fused"]
    assert got.shape == want.shape
    assert same_floats(got, want), f"{name}: {count_float_mismatches(got, want)} differing values"
    if fx["kernel"] != 0:  # untouched pixels keep the upload-time value {0,0,0,1}; written ones carry w = 1 too
        assert np.all(out[:, 3] == 1.0)


def test_cpu_backend_counts_the_rays_the_reference_traces(restatement):
    objs, lights = random_scene(8, 6, 3, seed=5, directional_lights=1)
    rays = camera.primary_rays(48, 32)
    for kernel, depth in (("hittest", 0), ("shade", 0), ("shade_and_reflect", 4)):
        rt = CPURaytracer(objs, lights, rays, depth, kernel=kernel, threads=3)
        out = rt.Render()
        want = restatement[True].render(kernel, objs, lights, rays, depth)
        assert rt.rays_traced == want["rays_ref"] and rt.hit_pixels == int((want["hit_index"] >= 0).sum())
        assert rt.threads_used == 3
        assert same_floats(out if kernel == "hittest" else out[:, :3], want["out"] if kernel == "hittest" else want["out"][:, :3])


def test_cpu_backend_thread_count_does_not_change_the_frame():
    objs, lights = random_scene(5, 5, 2, seed=6)
    rays = camera.primary_rays(40, 30)
    a = CPURaytracer(objs, lights, rays, 3, threads=1).Render()
    b = CPURaytracer(objs, lights, rays, 3, threads=7).Render()
    assert np.array_equal(a.view(np.uint32), b.view(np.uint32))


def test_cpu_backend_empty_inputs():
    objs, lights = random_scene(2, 1, 1, seed=7)
    assert CPURaytracer(objs, lights, camera.primary_rays(4, 4)[:0], 2).Render().shape == (0, 4)
    out = CPURaytracer(objs[:0], lights, camera.primary_rays(4, 4), 2).Render()
    assert np.all(out == np.array([0, 0, 0, 1], dtype=np.float32))
