"""CPU: the C-ABI library loads and exports every symbol include/hip_raytracer.h declares; without a GPU the
product fails loudly (no CPU fallback). No compute calls here."""
import ctypes
import re
from pathlib import Path

import pytest

import helpers  # noqa: F401  (registers the package under its importable name)

ROOT = Path(__file__).resolve().parent.parent
HEADER = ROOT / "include" / "hip_raytracer.h"


def declared_functions():
    text = re.sub(r"/\*.*?\*/", "", HEADER.read_text(), flags=re.S)
    return sorted(set(re.findall(r"\b(rt_[a-z_]+)\s*\(", text)))


def test_header_declares_the_boundary():
    fns = declared_functions()
    for must in ("rt_create", "rt_render", "rt_render_device", "rt_destroy", "rt_last_error", "rt_set_camera",
                 "rt_set_shard"):
        assert must in fns


def test_library_exports_every_declared_symbol():
    from opencl_raytracer_amd import hip_raytracer as hr
    if not hr.LIB_PATH.exists():
        import __graft_entry__
        __graft_entry__.build()
    lib = ctypes.CDLL(str(hr.LIB_PATH))
    for name in declared_functions():
        assert hasattr(lib, name), f"{name} declared in hip_raytracer.h but not exported"
    assert sorted(hr.EXPORTS) == declared_functions()
    lib.rt_abi_version.restype = ctypes.c_int
    assert lib.rt_abi_version() == 3  # 3: multi-device stats, per-device host placement, ray-domain guard (history: hip_raytracer.h)


def test_no_silent_cpu_fallback():
    """Without a usable GPU, constructing the raytracer must raise (RT_ERR_NO_DEVICE), not degrade."""
    import torch
    if torch.cuda.is_available():
        pytest.skip("a GPU is present")
    from helpers import camera, random_scene
    from opencl_raytracer_amd.hip_raytracer import HIPRaytracer, RTError
    objs, lights = random_scene(1, 1, 1, seed=3)
    with pytest.raises(RTError) as ei:
        HIPRaytracer(objs, lights, camera.primary_rays(4, 4), 1)
    assert ei.value.code == -2 and "no CPU fallback" in str(ei.value)


def test_product_never_touches_the_oracle():
    """Nothing under the package (Python or C++/HIP) may import, include or link oracle/."""
    pkg = ROOT / "opencl-raytracer_amd"
    for p in list(pkg.rglob("*.py")) + list(pkg.rglob("*.h")) + list(pkg.rglob("*.hpp")) + list(pkg.rglob("*.cpp")) \
            + list(pkg.rglob("*.hip")) + list(pkg.rglob("Makefile")):
        text = p.read_text()
        assert not re.search(r"^\s*(from|import)\s+oracle\b", text, flags=re.M), p
        assert "rt_oracle" not in text and "oracle/_" not in text, p
