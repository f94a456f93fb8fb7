"""CPU, build container only: the restatement against the reference kernels compiled verbatim (oracle/_ref)
on seeded random scenes beyond the committed fixtures. Skipped where /root/reference never existed
(the GPU box) - there the committed golden vectors carry the pin."""
import numpy as np
import pytest

from helpers import camera, random_scene, same_floats
from oracle import oracle

pytestmark = pytest.mark.skipif(not oracle.reference_available(), reason="oracle/_ref not built (no /root/reference)")


@pytest.mark.parametrize("fused", [True, False], ids=["fused", "unfused"])
@pytest.mark.parametrize("kernel", ["hittest", "shade", "shade_and_reflect"])
@pytest.mark.parametrize("seed", [101, 102, 103])
def test_random_scene_bit_exact(seed, kernel, fused, restatement):
    objs, lights = random_scene(10 + seed % 7, 8, 1 + seed % 4, seed=seed, directional_lights=seed % 2)
    rays = camera.primary_rays(40, 30)
    ref = oracle.Reference(kernel, fused).render(objs, lights, rays, 3)["out"]
    got = restatement[fused].render(kernel, objs, lights, rays, 3)["out"]
    if kernel != "hittest":
        ref, got = ref[:, :3], got[:, :3]
    assert same_floats(ref, got)


def test_known_answer_simple_sphere_ppm():
    """Config 1 known answer: 256x256 simpleSphere written as P3 has md5 28365bd1..."""
    import hashlib
    from helpers import SCENES
    from opencl_raytracer_amd import ppm, scene_loader
    objs, lights = scene_loader.load_scene(str(SCENES / "simpleSphere.txt"))
    rays = camera.primary_rays(256, 256)
    for kernel in ("shade", "shade_and_reflect"):
        for fused in (True, False):
            out = oracle.Reference(kernel, fused).render(objs, lights, rays, 3)["out"]
            assert int((out[:, :3].sum(1) != 0).sum()) == 1565
            blob = ppm.format_p3(256, 256, ppm.rgba_to_rgb(out))
            assert len(blob) == 397825
            assert hashlib.md5(blob).hexdigest() == "28365bd12a502710be0c9a9a1a8057a9"
