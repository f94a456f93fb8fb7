"""CPU: host-side mirrors of the reference interface - scene loader, camera rays, PPM sink, synthetic
generator, shard arithmetic, record layouts."""
import hashlib

import numpy as np
import pytest

from helpers import R, ROOT, SCENES, camera
from opencl_raytracer_amd import ppm, scene_loader, sharding, synthetic


def col_major(rows):
    return np.array(rows, dtype=np.float32).T.reshape(16)


def test_record_layouts():
    assert R.OBJECT_DTYPE.itemsize == 320 and R.OBJECT_DTYPE.fields["mvInverse"][1] == 128
    assert R.OBJECT_DTYPE.fields["type"][1] == 256 and R.OBJECT_DTYPE.fields["absorption"][1] == 48
    assert R.LIGHT_DTYPE.itemsize == 64 and R.LIGHT_DTYPE.fields["position"][1] == 48
    assert R.RAY_DTYPE.itemsize == 32


def test_simple_sphere_records():
    """SURVEY.md Appendix A known answers (exactly representable)."""
    objs, lights = scene_loader.load_scene(str(SCENES / "simpleSphere.txt"))
    assert len(objs) == 1 and len(lights) == 1 and objs["type"][0] == R.SPHERE
    assert np.array_equal(objs["mv"][0], col_major([[1, 0, 0, 0], [0, 1, 0, 0], [0, 0, 1, -10], [0, 0, 0, 1]]))
    assert np.array_equal(objs["mvInverse"][0], col_major([[1, 0, 0, 0], [0, 1, 0, 0], [0, 0, 1, 10], [0, 0, 0, 1]]))
    assert np.array_equal(lights["position"][0], np.array([10, 10, 0, 1], dtype=np.float32))
    assert np.allclose(objs["ambient"][0][:3], (1, 0, 0)) and objs["absorption"][0] == 1 and objs["shininess"][0] == 1
    assert np.allclose(lights["ambient"][0][:3], .3) and np.allclose(lights["specular"][0][:3], 1)


def test_multiple_spheres_records():
    objs, lights = scene_loader.load_scene(str(SCENES / "multipleSpheres.txt"))
    assert len(objs) == 3
    assert np.array_equal(objs["mv"][1], col_major([[2, 0, 0, 4], [0, 3, 0, 2], [0, 0, 1, -10], [0, 0, 0, 1]]))
    assert np.array_equal(objs["mv"][2], col_major([[2, 0, 0, -3], [0, 2, 0, -4], [0, 0, 2, -10], [0, 0, 0, 1]]))
    assert np.array_equal(lights["position"][0], np.array([10, 10, 0, 1], dtype=np.float32))
    # inverse really inverts
    for o in objs:
        m = o["mv"].reshape(4, 4).T.astype(np.float64)
        mi = o["mvInverse"].reshape(4, 4).T.astype(np.float64)
        assert np.allclose(m @ mi, np.eye(4), atol=1e-6)
        assert np.array_equal(o["mvInverseTranspose"].reshape(4, 4), o["mvInverse"].reshape(4, 4).T)


def test_simple_scene_and_rounded_cube():
    objs, lights = scene_loader.load_scene(str(SCENES / "simpleScene.txt"))
    assert list(objs["type"]) == [R.BOX, R.SPHERE]
    assert np.array_equal(lights["position"][0], np.array([20, 20, 10, 1], dtype=np.float32))  # light under scale 2
    assert np.array_equal(objs["mv"][1], col_major([[2, 0, 0, 0], [0, 2, 0, 0], [0, 0, 2, -10], [0, 0, 0, 1]]))
    box = objs["mv"][0].reshape(4, 4).T
    assert np.allclose(box[:3, 3], (2, 0, -10)) and np.allclose(np.linalg.norm(box[:3, :3], axis=0), 4, atol=1e-5)
    objs, lights = scene_loader.load_scene(str(SCENES / "roundedCube.txt"))
    assert len(objs) == 9 and objs["type"][0] == R.BOX and np.all(objs["type"][1:] == R.SPHERE)
    assert np.isclose(objs["absorption"][0], .2) and np.isclose(objs["absorption"][1], .7) and objs["shininess"][0] == 100


@pytest.mark.parametrize("text,needle", [
    ("material a\n ambient 1 0 0\n===\n", "proper indentation"),
    ("material a\n    ambient 1 0 0\n===\n", "indented too far"),
    ("material a\n  ambient 1 0\n===\n", "ambient expects 3 arguments, found 3"),
    ("material\n===\n", "material expects 1 argument, found 0"),
    ("foo a\n===\n", "unsupported command 'foo' in header"),
    ("material a\n  glow 1\n===\n", "unsupported command 'glow' while parsing material"),
    ("material a\n  light b\n===\n", "tried to declare a light in a nested scope"),
    ("material a\n===\nprimative cone a\n", "unsupported primative type 'cone'"),
    ("material a\n===\nprimative sphere\n", "primative expects 2 argument, found 1"),
    ("material a\n===\ntranslate 1 2\n", "translate expects 3 arguments, found 3"),
    ("material a\n===\nrotate 1 2 3\n", "rotate expects 4 arguments, found 4"),
    ("material a\n===\nshear 1 2 3\n", "unsupported command 'shear' in body"),
])
def test_loader_errors(text, needle):
    with pytest.raises(scene_loader.SceneError) as ei:
        scene_loader.SceneLoader().LoadString(text)
    assert needle in str(ei.value) and "Error parsing scene file at line" in str(ei.value)


def test_loader_misc():
    with pytest.raises(scene_loader.SceneError):
        scene_loader.load_scene("/nonexistent/scene.txt")
    with pytest.raises(KeyError):  # std::out_of_range from map::at
        scene_loader.SceneLoader().LoadString("material a\n===\nprimative sphere nope\n")
    # comments, blank lines, scoping by indentation, extra tokens ignored
    objs, lights = scene_loader.SceneLoader().LoadString(
        "# c\nmaterial m\n  ambient 1 1 1 junk\n\nlight l\n  diffuse 1 1 1\n===\ntranslate 1 0 0\n  primative box m\n"
        "  scale 2 2 2\n    primative sphere m\nprimative sphere m\nlight l\n")
    assert list(objs["type"]) == [R.BOX, R.SPHERE, R.SPHERE] and len(lights) == 1
    assert objs["mv"][0][12] == 1 and objs["mv"][1][0] == 2 and objs["mv"][2][12] == 0
    assert np.array_equal(lights["position"][0], np.array([0, 0, -10, 1], dtype=np.float32))


def test_camera_rays_q14():
    W, H = 8, 6
    rays = camera.primary_rays(W, H).reshape(H, W)
    assert np.all(rays["start"] == np.array([0, 0, 0, 1], dtype=np.float32))
    assert np.all(rays["direction"][..., 3] == 0)
    assert np.array_equal(rays["direction"][0, :, 0], np.arange(W, dtype=np.float32) - 4)
    assert np.array_equal(rays["direction"][:, 0, 1], (6 - np.arange(H, dtype=np.float32)) - 3)  # y runs H..1
    assert np.all(rays["direction"][..., 2] == camera.camera_z(H))
    assert abs(float(camera.camera_z(256)) - (-221.7025)) < 1e-3
    band = camera.primary_rays(W, H, row_begin=2, row_end=4)
    assert np.array_equal(band, rays[2:4].reshape(-1))
    crop = camera.crop_rays(W, H, 3, 1, 2, 2).reshape(2, 2)
    assert np.array_equal(crop, rays[1:3, 3:5])


def test_ppm_export(tmp_path):
    rgba = np.array([[0, .5, 1, 1], [1.5, -0.1, 0.999, 0]], dtype=np.float32)
    blob = ppm.format_p3(2, 1, ppm.rgba_to_rgb(rgba))
    assert blob == b"P3\n2 1\n255\n0 127 255\n255 -26 254\n"  # min(255, floor(v*255)), no lower clamp
    ppm.ExportP3(str(tmp_path / "x.ppm"), 2, 1, ppm.rgba_to_rgb(rgba))
    assert (tmp_path / "x.ppm").read_bytes() == blob


def test_synthetic_generator_is_deterministic():
    o1, l1 = synthetic.spheres_and_lights(500, 8)
    o2, l2 = synthetic.spheres_and_lights(500, 8)
    assert o1.tobytes() == o2.tobytes() and l1.tobytes() == l2.tobytes()
    assert hashlib.md5(o1.tobytes()).hexdigest() == hashlib.md5(o2.tobytes()).hexdigest()
    assert np.all(o1["type"] == 0) and np.all(o1["absorption"] == .5)
    s = o1["mv"][:, 0]
    assert s.min() >= .1 and s.max() <= .5
    assert np.allclose(o1["mvInverse"][:, 0] * s, 1, atol=1e-6)
    assert np.all(l1["position"][:, 3] == 1)
    # prefix property: the first 100 objects of the 500-object scene are the 100-object scene's objects
    o3, _ = synthetic.spheres_and_lights(100, 8)
    assert o3.tobytes() == o1[:100].tobytes()


def test_shard_arithmetic():
    n, tile = 231, 50  # 5 tiles, the last ragged
    assert sharding.n_tiles(n, tile) == 5
    assert sharding.local_tiles(n, tile, 0, 2) == [0, 2, 4] and sharding.local_tiles(n, tile, 1, 2) == [1, 3]
    assert sharding.local_rays(n, tile, 0, 2) == 150 and sharding.local_rays(n, tile, 1, 2) == 100
    assert sharding.max_local_rays(n, tile, 2) == 150
    frame = np.arange(5 * tile * 4, dtype=np.float32).reshape(5 * tile, 4)
    pieces = [np.concatenate([frame[t * tile:(t + 1) * tile] for t in sharding.local_tiles(n, tile, r, 2)]) for r in range(2)]
    pieces[1] = np.concatenate([pieces[1], np.zeros((50, 4), np.float32)])  # padded to max_local like the gather does
    assert np.array_equal(sharding.assemble_frame(pieces, tile, n), frame[:n])
    assert np.array_equal(sharding.assemble_frame([frame], tile, n), frame[:n])


def test_scene_files_are_what_the_generator_prints():
    """scenes/*.txt (the inputs BASELINE's configs name) are emitted by scenes/make_scenes.py from a structured
    description; the committed text must be exactly that."""
    import importlib.util
    spec = importlib.util.spec_from_file_location("make_scenes", ROOT / "scenes" / "make_scenes.py")
    mod = importlib.util.module_from_spec(spec)
    spec.loader.exec_module(mod)
    for name in mod.SCENES:
        assert (ROOT / "scenes" / f"{name}.txt").read_text() == mod.scene_text(name), name


def test_light_tile_block_indices_share_one_limit():
    """light-tile blocks are addressed through 24-bit indices - the walks keep the entry to resume a block at in
    bits 24+ of their cursor - while the host used to accept tables of up to 2^30 blocks. Host limit and device masks now come from
    ONE constant (rt_grid.h: kLtBlockIndexBits, with a compile-time check of the boundary), the builder counts the blocks exactly,
    and no walk masks a light-tile cursor with a literal of its own."""
    import re
    from pathlib import Path
    src = Path(__file__).resolve().parent.parent / "opencl-raytracer_amd" / "csrc"
    grid, api, wf = (src / "rt_grid.h").read_text(), (src / "rt_api.cpp").read_text(), (src / "rt_wavefront.hip").read_text()
    assert re.search(r"constexpr uint32_t kLtBlockIndexBits = 24;", grid)
    assert "static_assert(light_tile_blocks_fit((1ull << 24) - 1) && !light_tile_blocks_fit(1ull << 24)" in grid
    assert "rt::light_tile_blocks_fit(n_lt_blocks)" in api and "1ull << 30" not in api
    # every place that splits a light-tile cursor into (block, position) uses the shared mask / shift
    for m in re.finditer(r"const uint32_t b = (\w+) & (\w+), pos = \1 >> (\w+);", wf):
        if m.group(1) in ("e", "cursor"):   # the light-tile walks of trace_segment / walk_segment (the block GRID's cursors have their own 24-bit rule, checked by build_walk_blocks)
            assert (m.group(2), m.group(3)) == ("kLtBlockIndexMask", "kLtBlockIndexBits"), m.group(0)
    assert wf.count("kLtBlockIndexMask") >= 2
