"""Primary-ray generation - the caller side of the IRaytracer boundary.

Restates the reference's pinhole camera (OpenCL-Raytracer.cpp:18-26 `screenSpaceToViewSpace`
and the ray loop :68-72): origin (0,0,0,1); for pixel column ii, row jj (row-major, jj outer)

    direction = ( ii - W/2 , (H - jj) - H/2 , -(H/2) / tan(fov/2) , 0 )        (not normalised)

Note y runs H..1. x and y are exact small (half-)integers in fp32, so the
HIP backend can regenerate these rays in-kernel bit-exactly from (W, H, z).
"""
from __future__ import annotations

import ctypes
import ctypes.util

import numpy as np

from .records import RAY_DTYPE, radians

F = np.float32
_libm = ctypes.CDLL(ctypes.util.find_library("m") or "libm.so.6")
_libm.tanf.restype = ctypes.c_float
_libm.tanf.argtypes = [ctypes.c_float]


def half_fov(fov_degrees: float = 60.0) -> np.float32:
    """main(): `float fov = glm::radians(60.f); fov *= 0.5f;` (OpenCL-Raytracer.cpp:32-33)."""
    return F(radians(fov_degrees) * F(0.5))


def camera_z(height: int, fov_degrees: float = 60.0) -> np.float32:
    """-(halfHeight / tan(angle)) in float32 with libm tanf, as a C++ host computes it."""
    half_h = F(F(height) / F(2.0))
    t = F(_libm.tanf(float(half_fov(fov_degrees))))
    return F(-(half_h / t))


def primary_rays(width: int, height: int, fov_degrees: float = 60.0,
                 row_begin: int = 0, row_end: int | None = None) -> np.ndarray:
    """Ray3D array in the reference's order (rows [row_begin,row_end) of the W x H grid)."""
    row_end = height if row_end is None else row_end
    z = camera_z(height, fov_degrees)
    half_w = F(F(width) / F(2.0))
    half_h = F(F(height) / F(2.0))
    ii = np.arange(width, dtype=F)
    jj = np.arange(row_begin, row_end, dtype=F)
    x = (ii - half_w).astype(F)
    y = ((F(height) - jj) - half_h).astype(F)
    rays = np.zeros((row_end - row_begin, width), dtype=RAY_DTYPE)
    rays["start"][..., 3] = 1.0
    rays["direction"][..., 0] = x[None, :]
    rays["direction"][..., 1] = y[:, None]
    rays["direction"][..., 2] = z
    return rays.reshape(-1)


def crop_rays(width: int, height: int, x0: int, y0: int, w: int, h: int, fov_degrees: float = 60.0) -> np.ndarray:
    """The rays of a w x h window of the full W x H grid (for bounded CPU baselines)."""
    full_rows = primary_rays(width, height, fov_degrees, y0, y0 + h).reshape(h, width)
    return np.ascontiguousarray(full_rows[:, x0:x0 + w]).reshape(-1)
