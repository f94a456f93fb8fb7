"""TEST INFRASTRUCTURE - loaders for the checkers under oracle/.

Only tests/, __graft_entry__.smoke() and bench.py's cpu_baseline leg import this module.
The product path (opencl-raytracer_amd) never does.

Two checkers:
  * Restatement  - oracle/rt_oracle.c, this repo's own CPU restatement of the reference
                   algorithm (travels to the GPU box as oracle/_build/*.so).
  * Reference    - oracle/_ref/*.so, the reference's OpenCL-C kernels compiled verbatim for
                   the host (only exists where /root/reference was present at build time).
Both take the reference's device-layout record buffers as numpy arrays.
"""
from __future__ import annotations

import ctypes
import os
import subprocess
from pathlib import Path

import numpy as np

HERE = Path(__file__).resolve().parent
BUILD = HERE / "_build"
REFDIR = HERE / "_ref"

KERNELS = {"hittest": 0, "shade": 1, "shade_and_reflect": 2}
MAX_FLOAT = np.float32(3.402823466e+38)


def build(verbose: bool = False) -> None:
    """Compile the checkers (restatement always; oracle/_ref when /root/reference exists)."""
    res = subprocess.run(["make", "-C", str(HERE)], capture_output=True, text=True)
    if verbose or res.returncode != 0:
        print(res.stdout)
        print(res.stderr)
    if res.returncode != 0:
        raise RuntimeError("oracle build failed")


def _cpu_has_fma() -> bool:
    try:
        with open("/proc/cpuinfo") as f:
            for line in f:
                if line.startswith("flags"):
                    return " fma " in (line + " ")
    except OSError:
        pass
    return False


def _as_c(a: np.ndarray):
    assert a.flags["C_CONTIGUOUS"]
    return a.ctypes.data_as(ctypes.c_void_p)


def _kernel_id(kernel) -> int:
    return KERNELS[kernel] if isinstance(kernel, str) else int(kernel)


def _alloc_out(kernel_id: int, n: int, init_out):
    """Output buffers start in the state the reference host uploads: pixels {0,0,0,1}
    (OpenCLRaytracer.cpp:28-32); for hittest (never run by the reference host) MAX_FLOAT."""
    if init_out is not None:
        return np.ascontiguousarray(init_out, dtype=np.float32).copy()
    if kernel_id == 0:
        return np.full(n, MAX_FLOAT, dtype=np.float32)
    out = np.zeros((n, 4), dtype=np.float32)
    out[:, 3] = 1.0
    return out


class Restatement:
    """This repo's CPU restatement (oracle/rt_oracle.c)."""

    def __init__(self, fused: bool = True):
        self.fused = bool(fused)
        if fused:
            name = "librt_oracle_fused_fma.so" if _cpu_has_fma() else "librt_oracle_fused.so"
        else:
            name = "librt_oracle_unfused.so"
        path = BUILD / name
        if not path.exists():
            build()
        self.path = path
        self.lib = ctypes.CDLL(str(path))
        self.lib.rto_render.restype = ctypes.c_int
        self.lib.rto_render.argtypes = [
            ctypes.c_int, ctypes.c_uint32, ctypes.c_uint32, ctypes.c_void_p, ctypes.c_uint32, ctypes.c_void_p,
            ctypes.c_void_p, ctypes.c_void_p, ctypes.c_uint64, ctypes.c_uint64,
            ctypes.c_void_p, ctypes.c_void_p, ctypes.c_void_p, ctypes.c_int]
        assert self.lib.rto_fused() == int(self.fused)

    def render(self, kernel, objs: np.ndarray, lights: np.ndarray, rays: np.ndarray, max_bounces: int = 0,
               threads: int = 0, init_out=None, want_aux: bool = True):
        """Returns dict(out, hit_index, hit_t, rays_ref, threads)."""
        kid = _kernel_id(kernel)
        n = int(rays.shape[0])
        out = _alloc_out(kid, n, init_out)
        idx = np.full(n, -1, dtype=np.int32)
        t = np.full(n, MAX_FLOAT, dtype=np.float32)
        rr = ctypes.c_uint64(0)
        used = self.lib.rto_render(
            kid, int(max_bounces), int(objs.shape[0]), _as_c(objs) if objs.shape[0] else None,
            int(lights.shape[0]), _as_c(lights) if lights.shape[0] else None, _as_c(rays), _as_c(out), 0, n,
            _as_c(idx) if want_aux else None, _as_c(t) if want_aux else None, ctypes.byref(rr), int(threads))
        return {"out": out, "hit_index": idx, "hit_t": t, "rays_ref": int(rr.value), "threads": used}


def reference_available() -> bool:
    return (REFDIR / "libref_shade_and_reflect_unfused.so").exists() and _cpu_has_fma()


class Reference:
    """The reference's own kernels, compiled verbatim for the host (oracle/_ref)."""

    def __init__(self, kernel, fused: bool = True, shim_variant: int = 0):
        """shim_variant 1..3: the fused kernels linked against other conforming definitions of dot() / normalize()
        (ref_shim.cl) - shade / shade_and_reflect only."""
        self.kernel_id = _kernel_id(kernel)
        name = {0: "hittest", 1: "shade", 2: "shade_and_reflect"}[self.kernel_id]
        flavour = "fused" if fused else "unfused"
        if shim_variant:
            assert fused and self.kernel_id in (1, 2)
            flavour += f"_shim{int(shim_variant)}"
        path = REFDIR / f"libref_{name}_{flavour}.so"
        if not path.exists():
            raise FileNotFoundError(f"{path} (oracle/_ref only exists where /root/reference was present)")
        # RTLD_LOCAL: the kernel files define clashing globals
        self.lib = ctypes.CDLL(str(path), mode=os.RTLD_LOCAL | os.RTLD_NOW)
        self.lib.ref_run.restype = ctypes.c_int
        self.lib.ref_run.argtypes = [
            ctypes.c_uint32, ctypes.c_uint32, ctypes.c_void_p, ctypes.c_uint32, ctypes.c_void_p,
            ctypes.c_void_p, ctypes.c_void_p, ctypes.c_uint64, ctypes.c_uint64, ctypes.c_int]
        assert self.lib.ref_kernel_id() == self.kernel_id

    def render(self, objs: np.ndarray, lights: np.ndarray, rays: np.ndarray, max_bounces: int = 0,
               threads: int = 0, init_out=None):
        n = int(rays.shape[0])
        out = _alloc_out(self.kernel_id, n, init_out)
        # zero-length arrays still need a valid pointer for ctypes
        o = objs if objs.shape[0] else np.zeros(1, dtype=objs.dtype)
        li = lights if lights.shape[0] else np.zeros(1, dtype=lights.dtype)
        used = self.lib.ref_run(int(max_bounces), int(objs.shape[0]), _as_c(o), int(lights.shape[0]), _as_c(li),
                                _as_c(rays), _as_c(out), 0, n, int(threads))
        return {"out": out, "threads": used}
