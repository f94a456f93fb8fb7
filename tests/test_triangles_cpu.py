"""Triangles (type 2) - an extension with no reference semantics: the tessellator and the
oracle's statement of the spec, on the CPU. Parity of the HIP path is in tests/test_triangles_gpu.py."""
import numpy as np
import pytest

from helpers import R, camera
from opencl_raytracer_amd import scene_loader, tessellate as T


def _one_triangle(v0, v1, v2):
    tmpl = R.make_object(R.SPHERE, R.Material((.2, .3, .4), (.5, .5, .5), (.1, .1, .1), absorption=1.0), np.eye(4, dtype=np.float32))
    return T.triangle_records([v0], [v1], [v2], tmpl)


def _ray(s, d):
    r = np.zeros(1, dtype=R.RAY_DTYPE)
    r["start"][0] = (*s, 1.0)
    r["direction"][0] = (*d, 0.0)
    return r


def test_record_layout_and_guard_sphere():
    tri = _one_triangle((0, 0, -5), (2, 0, -5), (0, 2, -5))
    assert tri.dtype.itemsize == 320 and int(tri["type"][0]) == T.TRIANGLE
    mv = tri["mv"][0].reshape(4, 4)                      # [column][row]
    assert np.array_equal(mv[0], (0, 0, -5, 1)) and np.array_equal(mv[1], (2, 0, -5, 1)) and np.array_equal(mv[2], (0, 2, -5, 1))
    c = tri["mvInverse"][0][:3].astype(np.float64)
    rad = float(tri["mvInverse"][0][3])
    for v in ((0, 0, -5), (2, 0, -5), (0, 2, -5)):
        assert np.linalg.norm(np.asarray(v, dtype=np.float64) - c) <= rad    # every vertex inside the guard sphere


def test_oracle_known_answers(restatement):
    tri = _one_triangle((0, 0, -5), (2, 0, -5), (0, 2, -5))
    lights = R.lights_array([R.make_light(R.LightProperties((1, 1, 1), (0, 0, 0), (0, 0, 0)), position=(0, 0, 10, 1))])
    o = restatement[True]
    # straight through the interior: t = 5 / 1
    hit = o.render("hittest", tri, lights, _ray((0.5, 0.5, 0), (0, 0, -1)), 0)
    assert hit["out"][0] == np.float32(5.0) and hit["hit_index"][0] == 0
    # un-normalised direction: t scales like the reference's other primitives
    assert o.render("hittest", tri, lights, _ray((0.5, 0.5, 0), (0, 0, -10)), 0)["out"][0] == np.float32(0.5)
    # two-sided: from behind
    assert o.render("hittest", tri, lights, _ray((0.5, 0.5, -10), (0, 0, 1)), 0)["out"][0] == np.float32(5.0)
    # outside the triangle (u + v > 1), behind the origin (t < 0), parallel (det = 0): misses
    big = np.float32(3.402823466e+38)
    assert o.render("hittest", tri, lights, _ray((1.5, 1.5, 0), (0, 0, -1)), 0)["out"][0] == big
    assert o.render("hittest", tri, lights, _ray((0.5, 0.5, -6), (0, 0, -1)), 0)["out"][0] == big
    assert o.render("hittest", tri, lights, _ray((0.5, 0.5, 0), (1, 0, 0)), 0)["out"][0] == big
    # the edges belong to the triangle (u = 0, v = 0, u + v = 1)
    assert o.render("hittest", tri, lights, _ray((0.0, 0.5, 0), (0, 0, -1)), 0)["out"][0] == np.float32(5.0)
    assert o.render("hittest", tri, lights, _ray((1.0, 1.0, 0), (0, 0, -1)), 0)["out"][0] == np.float32(5.0)
    # shading: ambient only (absorption 1, light ambient 1) = material ambient
    col = o.render("shade_and_reflect", tri, lights, _ray((0.5, 0.5, 0), (0, 0, -1)), 3)["out"][0]
    assert np.allclose(col[:3], (.2, .3, .4), atol=1e-6)


def test_guard_sphere_clips_the_triangle(restatement):
    """The spec tests a triangle only for rays whose line passes the record's guard sphere."""
    tri = _one_triangle((0, 0, -5), (2, 0, -5), (0, 2, -5))
    tri["mvInverse"][0][3] = 0.25     # a guard far smaller than the triangle, around its centroid (2/3, 2/3, -5)
    lights = R.lights_array([R.make_light(R.LightProperties((1, 1, 1), (0, 0, 0), (0, 0, 0)), position=(0, 0, 10, 1))])
    o = restatement[True]
    big = np.float32(3.402823466e+38)
    assert o.render("hittest", tri, lights, _ray((0.7, 0.7, 0), (0, 0, -1)), 0)["out"][0] == np.float32(5.0)
    assert o.render("hittest", tri, lights, _ray((0.1, 0.1, 0), (0, 0, -1)), 0)["out"][0] == big


def test_tessellation_is_closed_and_outward():
    objs, _ = scene_loader.load_scene(str(pytest.importorskip("pathlib").Path(__file__).resolve().parents[1] / "scenes" / "roundedCube.txt"))
    lat, lon, k = 6, 12, 2
    tri = T.tessellate(objs, lat, lon, k)
    n_s, n_b = int((objs["type"] == R.SPHERE).sum()), int((objs["type"] == R.BOX).sum())
    assert len(tri) == n_s * 2 * lon * (lat - 1) + n_b * 12 * k * k
    mv = tri["mv"].reshape(-1, 4, 4).astype(np.float64)
    v0, v1, v2 = mv[:, 0, :3], mv[:, 1, :3], mv[:, 2, :3]
    # closed surfaces: every undirected edge is shared by exactly two triangles (per object, vertices are bit-equal)
    edges = {}
    for a, b in ((v0, v1), (v1, v2), (v2, v0)):
        for p, q in zip(a, b):
            key = tuple(sorted((tuple(p), tuple(q))))
            edges[key] = edges.get(key, 0) + 1
    assert set(edges.values()) == {2}
    # outward winding: the normal points away from the object's centre
    per = [2 * lon * (lat - 1) if int(t) == R.SPHERE else 12 * k * k for t in objs["type"]]
    start = 0
    for rec, cnt in zip(objs, per):
        centre = rec["mv"].reshape(4, 4)[3, :3].astype(np.float64)
        sl = slice(start, start + cnt)
        n = np.cross(v1[sl] - v0[sl], v2[sl] - v0[sl])
        mid = (v0[sl] + v1[sl] + v2[sl]) / 3.0
        assert np.all(((mid - centre) * n).sum(1) > 0)
        start += cnt
    assert T.subdivision_for(objs, 1_000_000)[0] > 100


def test_tessellated_scene_converges_to_the_analytic_one(restatement):
    """Self-consistency of the extension: a finely tessellated roundedCube renders like the analytic one (same hit /
    miss for almost every pixel, nearest t within the chord error)."""
    from pathlib import Path
    objs, lights = scene_loader.load_scene(str(Path(__file__).resolve().parents[1] / "scenes" / "roundedCube.txt"))
    tri = T.tessellate(objs, 24, 48, 4)
    rays = camera.primary_rays(64, 64)
    o = restatement[True]
    a = o.render("hittest", objs, lights, rays, 0)["out"]
    b = o.render("hittest", tri, lights, rays, 0)["out"]
    big = np.float32(3.0e38)
    both = (a < big) & (b < big)
    assert ((a < big) != (b < big)).mean() < 0.02          # silhouettes only
    diff = np.abs(a[both] - b[both])                       # t is in units of the un-normalised pixel direction (~55 long)
    assert np.median(diff) < 2e-4 and diff.max() < 5e-3    # chord error of a 24 x 48 sphere; largest at grazing angles
