"""ASCII PPM sink. Mirrors `PPMExporter::ExportP3(path, width, height, vector<float> rgb)`
(PPMExporter.hpp:8, PPMExporter.cpp:7-30): header "P3\\n<W> <H>\\n255\\n", then one pixel per line,
each channel `min(255, (int)floorf(v * 255.f))` (no lower clamp), separated by single spaces.
"""
from __future__ import annotations

import numpy as np


def rgba_to_rgb(rgba: np.ndarray) -> np.ndarray:
    """float4 framebuffer (Render()'s return, IRaytracer.hpp:13) -> packed RGB floats, stride 3."""
    rgba = np.asarray(rgba, dtype=np.float32).reshape(-1, 4)
    return np.ascontiguousarray(rgba[:, :3]).reshape(-1)


def quantise(rgb: np.ndarray) -> np.ndarray:
    v = np.floor(np.asarray(rgb, dtype=np.float32) * np.float32(255.0))
    return np.minimum(255, v.astype(np.int64))


def format_p3(width: int, height: int, rgb: np.ndarray) -> bytes:
    q = quantise(rgb).reshape(width * height, 3)
    body = "".join(f"{r} {g} {b}\n" for r, g, b in q.tolist())
    return (f"P3\n{width} {height}\n255\n" + body).encode("ascii")


def ExportP3(out_file: str, width: int, height: int, pixel_data) -> None:
    with open(out_file, "wb") as f:
        f.write(format_p3(width, height, np.asarray(pixel_data, dtype=np.float32)))
