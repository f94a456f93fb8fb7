"""CPU: the C++ host mirrors that need no GPU - SceneLoader (scene text -> device records) through scene_tool,
compared with the Python loader and with the exactly representable known answers."""
import subprocess
from pathlib import Path

import numpy as np
import pytest

from helpers import R, SCENES
from opencl_raytracer_amd import scene_loader

ROOT = Path(__file__).resolve().parent.parent
TOOL = ROOT / "opencl-raytracer_amd" / "host" / "scene_tool"


@pytest.fixture(scope="module")
def tool():
    if not TOOL.exists():
        import __graft_entry__
        __graft_entry__.build()
    return TOOL


def cpp_records(tool, scene_path, tmp_path):
    out = tmp_path / "records.bin"
    res = subprocess.run([str(tool), "records", str(scene_path), str(out)], capture_output=True, text=True, timeout=60)
    assert res.returncode == 0, res.stdout + res.stderr
    raw = out.read_bytes()
    n_objs, n_lights = np.frombuffer(raw[:8], dtype=np.uint32)
    objs = np.frombuffer(raw[8:8 + 320 * n_objs], dtype=R.OBJECT_DTYPE)
    lights = np.frombuffer(raw[8 + 320 * n_objs:], dtype=R.LIGHT_DTYPE)
    assert len(lights) == n_lights
    return objs, lights


@pytest.mark.parametrize("scene,exact", [("simpleSphere", True), ("multipleSpheres", True), ("simpleScene", False), ("roundedCube", False)])
def test_cpp_loader_matches_python_loader(tool, tmp_path, scene, exact):
    objs, lights = cpp_records(tool, SCENES / f"{scene}.txt", tmp_path)
    po, pl = scene_loader.load_scene(str(SCENES / f"{scene}.txt"))
    assert len(objs) == len(po) and len(lights) == len(pl)
    for key in ("ambient", "diffuse", "specular", "absorption", "reflection", "transparency", "shininess", "type"):
        assert np.array_equal(objs[key], po[key])
    tol = 0.0 if exact else 1e-5  # rotated scenes go through libm sin/cos and a different (equally valid) inverse
    for key in ("mv", "mvInverse", "mvInverseTranspose"):
        assert np.abs(objs[key] - po[key]).max() <= tol
    assert np.abs(lights["position"] - pl["position"]).max() <= tol
    for key in ("ambient", "diffuse", "specular"):
        assert np.array_equal(lights[key], pl[key])


@pytest.mark.parametrize("text,needle,code", [
    ("material a\n ambient 1 0 0\n===\n", "proper indentation", 2),
    ("material a\n  ambient 1 0\n===\n", "ambient expects 3 arguments, found 3", 2),
    ("material a\n===\nprimative cone a\n", "unsupported primative type 'cone'", 2),
    ("material a\n===\nrotate 1 2 3\n", "rotate expects 4 arguments, found 4", 2),
    ("material a\n===\nprimative sphere nope\n", "out_of_range", 3),
])
def test_cpp_loader_errors(tool, tmp_path, text, needle, code):
    scene = tmp_path / "bad.txt"
    scene.write_text(text)
    res = subprocess.run([str(tool), "records", str(scene), str(tmp_path / "o.bin")], capture_output=True, text=True, timeout=60)
    assert res.returncode == code and needle in res.stdout


def test_cpp_loader_missing_file(tool, tmp_path):
    res = subprocess.run([str(tool), "records", "/nonexistent/scene.txt", str(tmp_path / "o.bin")], capture_output=True, text=True)
    assert res.returncode == 2 and "could not be found" in res.stdout


def test_config1_ppm_known_answer_on_the_cpu_backend(tool, tmp_path):
    """BASELINE configs[0] (simpleSphere.txt at 256x256 -> PPM, the no-GPU plumbing / diff harness) end to end in C++:
    SceneLoader -> primary rays -> `new CPURaytracer(...)` through the IRaytracer base -> PPMExporter. The P3 file must
    be the known answer of the reference's own kernels (397 825 bytes, md5 28365bd1...)."""
    import hashlib
    out = tmp_path / "cfg1.ppm"
    res = subprocess.run([str(tool), "render", str(SCENES / "simpleSphere.txt"), "256", "256", "3", str(out), "-", "cpu"],
                         capture_output=True, text=True, timeout=300)
    assert res.returncode == 0, res.stdout + res.stderr
    data = out.read_bytes()
    assert len(data) == 397825 and hashlib.md5(data).hexdigest() == "28365bd12a502710be0c9a9a1a8057a9"
    assert data.splitlines()[3 + 128 * 256 + 128] == b"76 95 136"
