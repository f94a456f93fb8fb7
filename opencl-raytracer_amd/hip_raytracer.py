"""HIPRaytracer - the Python flavour of the drop-in for the reference's OpenCLRaytracer.

Mirrors the reference interface (IRaytracer.hpp:10-21, OpenCLRaytracer.hpp:61-65):

    rt = HIPRaytracer(objects, lights, rays, MAX_BOUNCES)     # OpenCLRaytracer(objects, lights, rays, MAX_BOUNCES)
    pixels = rt.Render()                                      # cl_float4* Render(): R x 4 float32, host memory

`objects`, `lights`, `rays` are numpy record arrays in the reference's device layouts (records.py). Everything
goes through the C ABI of include/hip_raytracer.h (ctypes; no torch types cross the boundary). The library is
required: if csrc/libhip_raytracer.so is missing or no HIP device is usable, construction raises - there is no
CPU fallback.
"""
from __future__ import annotations

import ctypes
import os
from pathlib import Path

import numpy as np

from .records import LIGHT_DTYPE, OBJECT_DTYPE, RAY_DTYPE

LIB_PATH = Path(__file__).resolve().parent / "csrc" / "libhip_raytracer.so"

KERNEL_HITTEST, KERNEL_SHADE, KERNEL_SHADE_AND_REFLECT = 0, 1, 2
KERNELS = {"hittest": 0, "shade": 1, "shade_and_reflect": 2}
FLAG_UNFUSED, FLAG_LITERAL, FLAG_NO_RAYGEN, FLAG_WAVEFRONT, FLAG_MONOLITHIC, FLAG_NO_GRID, FLAG_FAST_PHONG = 0x1, 0x2, 0x4, 0x8, 0x10, 0x20, 0x40

EXPORTS = [
    "rt_abi_version", "rt_create", "rt_set_camera", "rt_set_shard", "rt_local_rays", "rt_render",
    "rt_render_device", "rt_set_aux_device", "rt_render_aux", "rt_count_rays", "rt_get_stats",
    "rt_timing_reset", "rt_timing_summary", "rt_destroy", "rt_last_error", "rt_get_setup_times",
    "rt_create_multi", "rt_set_camera_multi", "rt_multi_frame_elems", "rt_render_multi", "rt_render_multi_device",
    "rt_multi_context", "rt_multi_last_error", "rt_destroy_multi", "rt_count_rays_multi", "rt_get_stats_multi",
]


class RTError(RuntimeError):
    def __init__(self, code: int, msg: str):
        super().__init__(f"hip_raytracer error {code}: {msg}")
        self.code = code


class RTStats(ctypes.Structure):
    _fields_ = [
        ("rays_traced", ctypes.c_uint64), ("rays_reference", ctypes.c_uint64), ("hit_pixels", ctypes.c_uint64),
        ("last_kernel_ms", ctypes.c_float), ("pinhole", ctypes.c_uint32),
        ("width", ctypes.c_uint32), ("height", ctypes.c_uint32), ("local_rays", ctypes.c_uint64),
        ("wavefront", ctypes.c_uint32), ("rounds", ctypes.c_uint32), ("object_tests", ctypes.c_uint64),
    ]


class RTSetupTimes(ctypes.Structure):
    _fields_ = [(n, ctypes.c_double) for n in ("create_ms", "upload_ms", "grid_ms", "blocks_ms", "light_tiles_ms",
                                                "screen_tiles_ms", "buffers_ms")]

    def as_dict(self):
        return {n: float(getattr(self, n)) for n, _ in self._fields_}


_lib = None


def load_library(path: os.PathLike | None = None) -> ctypes.CDLL:
    """dlopen the C-ABI library and declare its signatures. Fails loudly when it is not built."""
    global _lib
    if _lib is not None and path is None:
        return _lib
    p = Path(path) if path else Path(os.environ.get("RT_LIB_OVERRIDE", LIB_PATH))  # override: A/B builds of the library
    if not p.exists():
        raise FileNotFoundError(
            f"{p} is missing - build it with `make -C opencl-raytracer_amd/csrc` (or __graft_entry__.build()); "
            "this backend has no CPU fallback")
    lib = ctypes.CDLL(str(p))
    vp, u32, u64, i32 = ctypes.c_void_p, ctypes.c_uint32, ctypes.c_uint64, ctypes.c_int
    lib.rt_abi_version.restype = i32
    lib.rt_create.restype = i32
    lib.rt_create.argtypes = [ctypes.POINTER(vp), vp, u32, vp, u32, vp, u64, u32, i32, i32, u32]
    lib.rt_set_camera.restype = i32
    lib.rt_set_camera.argtypes = [vp, u32, u32, ctypes.c_float]
    lib.rt_set_shard.restype = i32
    lib.rt_set_shard.argtypes = [vp, u64, u32, u32]
    lib.rt_local_rays.restype = u64
    lib.rt_local_rays.argtypes = [vp]
    lib.rt_render.restype = i32
    lib.rt_render.argtypes = [vp, ctypes.POINTER(ctypes.POINTER(ctypes.c_float))]
    lib.rt_render_device.restype = i32
    lib.rt_render_device.argtypes = [vp, vp, vp]
    lib.rt_set_aux_device.restype = i32
    lib.rt_set_aux_device.argtypes = [vp, vp, vp]
    lib.rt_render_aux.restype = i32
    lib.rt_render_aux.argtypes = [vp, vp, vp]
    lib.rt_count_rays.restype = i32
    lib.rt_count_rays.argtypes = [vp]
    lib.rt_get_stats.restype = i32
    lib.rt_get_stats.argtypes = [vp, ctypes.POINTER(RTStats)]
    lib.rt_timing_reset.restype = i32
    lib.rt_timing_reset.argtypes = [vp]
    lib.rt_timing_summary.restype = i32
    lib.rt_timing_summary.argtypes = [vp, ctypes.POINTER(ctypes.c_double), ctypes.POINTER(u32)]
    lib.rt_destroy.restype = None
    lib.rt_destroy.argtypes = [vp]
    lib.rt_last_error.restype = ctypes.c_char_p
    lib.rt_last_error.argtypes = [vp]
    lib.rt_get_setup_times.restype = i32
    lib.rt_get_setup_times.argtypes = [vp, ctypes.POINTER(RTSetupTimes)]
    lib.rt_create_multi.restype = i32
    lib.rt_create_multi.argtypes = [ctypes.POINTER(vp), vp, u32, vp, u32, vp, u64, u32, i32, ctypes.POINTER(i32), u32, u64, u32]
    lib.rt_set_camera_multi.restype = i32
    lib.rt_set_camera_multi.argtypes = [vp, u32, u32, ctypes.c_float]
    lib.rt_count_rays_multi.restype = i32
    lib.rt_count_rays_multi.argtypes = [vp]
    lib.rt_get_stats_multi.restype = i32
    lib.rt_get_stats_multi.argtypes = [vp, ctypes.POINTER(RTStats)]
    lib.rt_multi_frame_elems.restype = u64
    lib.rt_multi_frame_elems.argtypes = [vp]
    lib.rt_render_multi.restype = i32
    lib.rt_render_multi.argtypes = [vp, ctypes.POINTER(ctypes.POINTER(ctypes.c_float))]
    lib.rt_render_multi_device.restype = i32
    lib.rt_render_multi_device.argtypes = [vp, vp]
    lib.rt_multi_context.restype = vp
    lib.rt_multi_context.argtypes = [vp, u32]
    lib.rt_multi_last_error.restype = ctypes.c_char_p
    lib.rt_multi_last_error.argtypes = [vp]
    lib.rt_destroy_multi.restype = None
    lib.rt_destroy_multi.argtypes = [vp]
    if path is None:
        _lib = lib
    return lib


def _ptr(a: np.ndarray | None):
    if a is None or a.size == 0:
        return None
    assert a.flags["C_CONTIGUOUS"]
    return a.ctypes.data_as(ctypes.c_void_p)


class HIPRaytracer:
    """IRaytracer backend for one MI355X."""

    def __init__(self, objects: np.ndarray, lights: np.ndarray, rays: np.ndarray | None, MAX_BOUNCES: int = 0, *,
                 kernel="shade_and_reflect", device: int = 0, fused: bool = True, literal: bool = False,
                 raygen: bool = True, camera: tuple[int, int, float] | None = None, path: str = "auto",
                 grid: bool = True, fast_phong: bool = False):
        self._lib = load_library()
        self._ctx = ctypes.c_void_p()
        objects = np.ascontiguousarray(objects, dtype=OBJECT_DTYPE)
        lights = np.ascontiguousarray(lights, dtype=LIGHT_DTYPE)
        self.kernel = KERNELS[kernel] if isinstance(kernel, str) else int(kernel)
        flags = (0 if fused else FLAG_UNFUSED) | (FLAG_LITERAL if literal else 0) | (0 if raygen else FLAG_NO_RAYGEN)
        flags |= {"auto": 0, "wavefront": FLAG_WAVEFRONT, "monolithic": FLAG_MONOLITHIC}[path]
        flags |= 0 if grid else FLAG_NO_GRID
        flags |= FLAG_FAST_PHONG if fast_phong else 0
        if rays is not None:
            rays = np.ascontiguousarray(rays, dtype=RAY_DTYPE)
            n_rays = int(rays.shape[0])
        else:
            if camera is None:
                raise ValueError("either rays or camera=(width, height, z) is required")
            n_rays = int(camera[0]) * int(camera[1])
        rc = self._lib.rt_create(ctypes.byref(self._ctx), _ptr(objects), int(objects.shape[0]), _ptr(lights),
                                 int(lights.shape[0]), _ptr(rays), n_rays, int(MAX_BOUNCES), self.kernel,
                                 int(device), flags)
        if rc != 0:
            msg = self._lib.rt_last_error(None)
            self._ctx = ctypes.c_void_p()
            raise RTError(rc, msg.decode() if msg else "rt_create failed")
        if camera is not None:
            self._check(self._lib.rt_set_camera(self._ctx, int(camera[0]), int(camera[1]), float(camera[2])))
        self.n_rays = n_rays

    # -- plumbing ------------------------------------------------------------------------------
    def _check(self, rc: int):
        if rc != 0:
            msg = self._lib.rt_last_error(self._ctx)
            raise RTError(rc, msg.decode() if msg else "")

    @property
    def elem_floats(self) -> int:
        return 1 if self.kernel == KERNEL_HITTEST else 4

    @property
    def local_rays(self) -> int:
        return int(self._lib.rt_local_rays(self._ctx))

    def close(self):
        if getattr(self, "_ctx", None) and self._ctx.value:
            self._lib.rt_destroy(self._ctx)
            self._ctx = ctypes.c_void_p()

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass

    def __enter__(self):
        return self

    def __exit__(self, *exc):
        self.close()

    # -- the IRaytracer interface --------------------------------------------------------------
    def Render(self) -> np.ndarray:
        """Synchronous render; returns a copy of the context-owned host framebuffer:
        (local_rays, 4) float32 for shade / shade_and_reflect, (local_rays,) for hittest."""
        out = ctypes.POINTER(ctypes.c_float)()
        self._check(self._lib.rt_render(self._ctx, ctypes.byref(out)))
        n = self.local_rays
        if n == 0:
            return np.zeros((0, 4) if self.elem_floats == 4 else (0,), dtype=np.float32)
        arr = np.ctypeslib.as_array(out, shape=(n * self.elem_floats,)).copy()
        return arr.reshape(n, 4) if self.elem_floats == 4 else arr

    def render_host_ms(self, frames: int = 3) -> float:
        """Wall time of the synchronous Render() through the boundary - kernels + the blocking read-back into the
        context's pinned host buffer (OpenCLRaytracer.cpp:94) - without this wrapper's numpy copy: best of `frames`."""
        import time
        out = ctypes.POINTER(ctypes.c_float)()
        best = None
        for _ in range(max(1, frames)):
            t0 = time.perf_counter()
            self._check(self._lib.rt_render(self._ctx, ctypes.byref(out)))
            dt = (time.perf_counter() - t0) * 1e3
            best = dt if best is None else min(best, dt)
        return best

    def setup_times(self) -> dict:
        """One-time host-side work outside every render timer (rt_setup_times_t), milliseconds."""
        t = RTSetupTimes()
        self._check(self._lib.rt_get_setup_times(self._ctx, ctypes.byref(t)))
        return t.as_dict()

    # -- extensions over the reference interface -----------------------------------------------
    def set_shard(self, tile_rays: int, rank: int, world: int):
        self._check(self._lib.rt_set_shard(self._ctx, int(tile_rays), int(rank), int(world)))

    def render_device(self, d_out_ptr: int, stream_ptr: int = 0):
        """Asynchronous render into device memory (raw pointers, e.g. torch tensor.data_ptr())."""
        self._check(self._lib.rt_render_device(self._ctx, ctypes.c_void_p(d_out_ptr),
                                               ctypes.c_void_p(stream_ptr) if stream_ptr else None))

    def render_aux(self):
        """Primary-ray (t, winning object index) per work-item; index -1 on a miss."""
        n = self.local_rays
        t = np.empty(n, dtype=np.float32)
        idx = np.empty(n, dtype=np.int32)
        self._check(self._lib.rt_render_aux(self._ctx, _ptr(t) if n else None, _ptr(idx) if n else None))
        return t, idx

    def count_rays(self) -> RTStats:
        self._check(self._lib.rt_count_rays(self._ctx))
        return self.stats()

    def stats(self) -> RTStats:
        s = RTStats()
        self._check(self._lib.rt_get_stats(self._ctx, ctypes.byref(s)))
        return s

    def timing_reset(self):
        self._check(self._lib.rt_timing_reset(self._ctx))

    def timing_summary(self):
        total = ctypes.c_double(0)
        n = ctypes.c_uint32(0)
        self._check(self._lib.rt_timing_summary(self._ctx, ctypes.byref(total), ctypes.byref(n)))
        return float(total.value), int(n.value)


class MultiHIPRaytracer:
    """IRaytracer backend for several GPUs driven from ONE process through the C ABI (rt_create_multi): one context and one
    host thread per device, interleaved row-tiles; Render(): every device copies its tiles straight into the pinned host
    frame (render_device: device-to-device to their place in a frame on devices[0]). `devices` may repeat an ordinal (rehearsal on fewer GPUs). The torch.distributed flavour - one process
    per GPU, RCCL exchange - is distributed.ShardedHIPRaytracer."""

    def __init__(self, objects, lights, rays, MAX_BOUNCES: int = 0, *, devices=(0,), kernel="shade_and_reflect",
                 camera: tuple[int, int, float] | None = None, tile_rays: int = 0, fused: bool = True, literal: bool = False,
                 grid: bool = True):
        self._lib = load_library()
        self._m = ctypes.c_void_p()
        objects = np.ascontiguousarray(objects, dtype=OBJECT_DTYPE)
        lights = np.ascontiguousarray(lights, dtype=LIGHT_DTYPE)
        self.kernel = KERNELS[kernel] if isinstance(kernel, str) else int(kernel)
        flags = (0 if fused else FLAG_UNFUSED) | (FLAG_LITERAL if literal else 0) | (0 if grid else FLAG_NO_GRID)
        if rays is not None:
            rays = np.ascontiguousarray(rays, dtype=RAY_DTYPE)
            n_rays = int(rays.shape[0])
        else:
            if camera is None:
                raise ValueError("either rays or camera=(width, height, z) is required")
            n_rays = int(camera[0]) * int(camera[1])
            if tile_rays == 0:
                tile_rays = 16 * int(camera[0])
        devs = (ctypes.c_int * len(devices))(*[int(d) for d in devices])
        rc = self._lib.rt_create_multi(ctypes.byref(self._m), _ptr(objects), int(objects.shape[0]), _ptr(lights),
                                       int(lights.shape[0]), _ptr(rays), n_rays, int(MAX_BOUNCES), self.kernel, devs,
                                       len(devices), int(tile_rays), flags)
        if rc != 0:
            msg = self._lib.rt_multi_last_error(None)
            self._m = ctypes.c_void_p()
            raise RTError(rc, msg.decode() if msg else "rt_create_multi failed")
        if camera is not None:
            self._check(self._lib.rt_set_camera_multi(self._m, int(camera[0]), int(camera[1]), float(camera[2])))
        self.n_rays = n_rays
        self.n_devices = len(devices)

    def _check(self, rc: int):
        if rc != 0:
            msg = self._lib.rt_multi_last_error(self._m)
            raise RTError(rc, msg.decode() if msg else "")

    @property
    def elem_floats(self) -> int:
        return 1 if self.kernel == KERNEL_HITTEST else 4

    @property
    def frame_elems(self) -> int:
        return int(self._lib.rt_multi_frame_elems(self._m))

    def Render(self) -> np.ndarray:
        out = ctypes.POINTER(ctypes.c_float)()
        self._check(self._lib.rt_render_multi(self._m, ctypes.byref(out)))
        if self.n_rays == 0:
            return np.zeros((0, 4) if self.elem_floats == 4 else (0,), dtype=np.float32)
        arr = np.ctypeslib.as_array(out, shape=(self.n_rays * self.elem_floats,)).copy()
        return arr.reshape(self.n_rays, 4) if self.elem_floats == 4 else arr

    def count_rays(self) -> RTStats:
        """Untimed counted render on every shard; the counters summed over the shards (the whole frame's)."""
        self._check(self._lib.rt_count_rays_multi(self._m))
        return self.stats()

    def stats(self) -> RTStats:
        st = RTStats()
        self._check(self._lib.rt_get_stats_multi(self._m, ctypes.byref(st)))
        return st

    def render_host_ms(self, repeats: int = 3) -> float:
        """Wall clock of the synchronous Render() (kernels + every device's copy into the pinned host frame), best of `repeats`."""
        import time
        best = None
        out = ctypes.POINTER(ctypes.c_float)()
        for _ in range(max(1, repeats)):
            t0 = time.perf_counter()
            self._check(self._lib.rt_render_multi(self._m, ctypes.byref(out)))
            dt = (time.perf_counter() - t0) * 1e3
            best = dt if best is None else min(best, dt)
        return best

    def render_device(self, d_frame_ptr: int):
        """The whole frame into device memory on devices[0] (frame_elems elements); returns when it is complete. The buffer
        must be idle on entry (hip_raytracer.h)."""
        self._check(self._lib.rt_render_multi_device(self._m, ctypes.c_void_p(d_frame_ptr)))

    def close(self):
        if getattr(self, "_m", None) and self._m.value:
            self._lib.rt_destroy_multi(self._m)
            self._m = ctypes.c_void_p()

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass

    def __enter__(self):
        return self

    def __exit__(self, *exc):
        self.close()
