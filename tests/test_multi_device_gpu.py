"""GPU: the C ABI's several-GPU entry points (rt_create_multi / rt_render_multi / rt_render_multi_device) - one process,
one context and one host thread per shard, tiles copied device-to-device to their place in the frame. Rehearsed with
every shard on the test box's one GPU: frame equal to a single-context render bit for bit (SURVEY.md 8e, row e')."""
import numpy as np
import pytest
import torch

from helpers import camera, random_scene
from opencl_raytracer_amd.hip_raytracer import HIPRaytracer, MultiHIPRaytracer

pytestmark = pytest.mark.gpu


def _single(objs, lights, rays, depth, **kw):
    with HIPRaytracer(objs, lights, rays, depth, **kw) as rt:
        return rt.Render()


@pytest.mark.parametrize("n_shards", [2, 3])
def test_small_scene_frame_is_the_single_context_frame(n_shards):
    objs, lights = random_scene(10, 6, 3, seed=41, directional_lights=1)
    W, H = 96, 72  # 72 rows in tiles of 16: a ragged last tile
    rays = camera.primary_rays(W, H)
    want = _single(objs, lights, rays, 3)
    with MultiHIPRaytracer(objs, lights, rays, 3, devices=[0] * n_shards, tile_rays=16 * W) as rt:
        got = rt.Render()
        again = rt.Render()
    assert got.shape == want.shape
    assert np.array_equal(got.view(np.uint32), want.view(np.uint32))
    assert np.array_equal(again.view(np.uint32), want.view(np.uint32))


def test_grid_path_camera_mode_and_device_frame():
    objs, lights = random_scene(260, 40, 4, seed=43, spread=14.0, zrange=(-60.0, -12.0))
    W, H = 160, 120
    z = float(camera.camera_z(H))
    with HIPRaytracer(objs, lights, None, 3, camera=(W, H, z)) as rt:
        want = rt.Render()
        assert rt.stats().wavefront == 1
    with MultiHIPRaytracer(objs, lights, None, 3, devices=[0, 0, 0], camera=(W, H, z)) as rt:
        got = rt.Render()
        frame = torch.full((rt.frame_elems, 4), float("nan"), dtype=torch.float32, device="cuda")
        torch.cuda.synchronize()  # hip_raytracer.h: the frame must be idle on entry (the shards write it from their own streams)
        rt.render_device(frame.data_ptr())
        dev = frame[: W * H].cpu().numpy()
    assert np.array_equal(got.view(np.uint32), want.view(np.uint32))
    assert np.array_equal(dev.view(np.uint32), want.view(np.uint32))


def test_stats_cover_the_whole_frame_and_render_twice_reuses_the_workers():
    """Stats() of the several-GPU raytracer used to be shard 0's alone. rt_get_stats_multi sums the counters over the
    shards; the host threads live as long as the raytracer (many frames through the same workers)."""
    objs, lights = random_scene(260, 40, 4, seed=43, spread=14.0, zrange=(-60.0, -12.0))
    W, H = 160, 120
    z = float(camera.camera_z(H))
    with HIPRaytracer(objs, lights, None, 3, camera=(W, H, z)) as rt:
        want = rt.Render()
        one = rt.count_rays()
    with MultiHIPRaytracer(objs, lights, None, 3, devices=[0, 0, 0, 0], camera=(W, H, z)) as rt:
        st = rt.count_rays()
        assert (st.rays_traced, st.rays_reference, st.hit_pixels) == (one.rays_traced, one.rays_reference, one.hit_pixels)
        assert st.local_rays == rt.frame_elems >= W * H
        for _ in range(6):
            got = rt.Render()
            assert np.array_equal(got.view(np.uint32), want.view(np.uint32))
        assert rt.render_host_ms(2) > 0.0


def test_hittest_kernel_and_uneven_shares():
    objs, lights = random_scene(5, 3, 1, seed=47)
    W, H = 64, 40  # 40 rows / 8-row tiles = 5 tiles over 4 shards: shares of 2, 1, 1, 1
    rays = camera.primary_rays(W, H)
    want = _single(objs, lights, rays, 0, kernel="hittest")
    with MultiHIPRaytracer(objs, lights, rays, 0, devices=[0, 0, 0, 0], kernel="hittest", tile_rays=8 * W) as rt:
        got = rt.Render()
    assert np.array_equal(got.view(np.uint32), want.view(np.uint32))


def test_errors_are_reported():
    from opencl_raytracer_amd.hip_raytracer import RTError
    objs, lights = random_scene(2, 1, 1, seed=3)
    with pytest.raises(RTError):
        MultiHIPRaytracer(objs, lights, camera.primary_rays(8, 8), 1, devices=[0, 99])
