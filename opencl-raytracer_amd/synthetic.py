"""Synthetic scenes for the BASELINE.json configurations that have no scene file.

`spheres_and_lights(N, L)` is the config-4 workload: N unit spheres instanced
by translate(x,y,z) * scale(s), L positional lights, drawn from a xorshift64 stream with the
survey's seed so that every rank / run / the CPU baseline sees identical record bytes.

Draw order per object: x, y, z, s, ambient rgb, diffuse rgb, specular rgb, shininess;
then per light: x, y, z.  u = (state >> 40) / 2^24 (24 random bits -> exact float32).
"""
from __future__ import annotations

import functools

import numpy as np

from .records import LIGHT_DTYPE, OBJECT_DTYPE, SPHERE

F = np.float32
SEED = 88172645463325252
_MASK = (1 << 64) - 1


def _xorshift64_stream(n: int, seed: int = SEED) -> np.ndarray:
    out = np.empty(n, dtype=np.float64)
    x = seed & _MASK
    for i in range(n):
        x ^= (x << 13) & _MASK
        x ^= x >> 7
        x ^= (x << 17) & _MASK
        out[i] = (x >> 40) * (1.0 / 16777216.0)
    return out


@functools.lru_cache(maxsize=4)
def _cached(n_objects: int, n_lights: int, absorption: float, seed: int):
    per_obj, per_light = 14, 3
    u = _xorshift64_stream(n_objects * per_obj + n_lights * per_light, seed).astype(F)
    uo = u[: n_objects * per_obj].reshape(n_objects, per_obj)
    ul = u[n_objects * per_obj:].reshape(n_lights, per_light)

    def lerp(t, lo, hi):
        return (F(lo) + t * F(hi - lo)).astype(F)

    x = lerp(uo[:, 0], -40, 40)
    y = lerp(uo[:, 1], -40, 40)
    z = lerp(uo[:, 2], -80, -20)
    s = lerp(uo[:, 3], 0.1, 0.5)
    objs = np.zeros(n_objects, dtype=OBJECT_DTYPE)
    objs["ambient"][:, :3] = uo[:, 4:7]
    objs["diffuse"][:, :3] = uo[:, 7:10]
    objs["specular"][:, :3] = uo[:, 10:13]
    objs["shininess"] = lerp(uo[:, 13], 1, 51)
    objs["absorption"] = F(absorption)
    objs["reflection"] = F(1.0) - F(absorption)
    objs["type"] = SPHERE
    mv = np.zeros((n_objects, 4, 4), dtype=F)  # [obj][column][row]
    mv[:, 0, 0] = s
    mv[:, 1, 1] = s
    mv[:, 2, 2] = s
    mv[:, 3, 0], mv[:, 3, 1], mv[:, 3, 2], mv[:, 3, 3] = x, y, z, 1
    inv_s = (F(1) / s).astype(F)
    inv = np.zeros((n_objects, 4, 4), dtype=F)
    inv[:, 0, 0] = inv_s
    inv[:, 1, 1] = inv_s
    inv[:, 2, 2] = inv_s
    inv[:, 3, 0], inv[:, 3, 1], inv[:, 3, 2], inv[:, 3, 3] = -(x / s), -(y / s), -(z / s), 1
    objs["mv"] = mv.reshape(n_objects, 16)
    objs["mvInverse"] = inv.reshape(n_objects, 16)
    objs["mvInverseTranspose"] = np.ascontiguousarray(inv.transpose(0, 2, 1)).reshape(n_objects, 16)

    lights = np.zeros(n_lights, dtype=LIGHT_DTYPE)
    lights["ambient"][:, :3] = F(0.01)
    lights["diffuse"][:, :3] = F(0.05)
    lights["specular"][:, :3] = F(0.05)
    lights["position"][:, 0] = lerp(ul[:, 0], -30, 30)
    lights["position"][:, 1] = lerp(ul[:, 1], -30, 30)
    lights["position"][:, 2] = lerp(ul[:, 2], 0, 10)
    lights["position"][:, 3] = 1
    objs.setflags(write=False)
    lights.setflags(write=False)
    return objs, lights


def spheres_and_lights(n_objects: int, n_lights: int, absorption: float = 0.5, seed: int = SEED):
    """Returns (ObjectData[n_objects], Light[n_lights]) record arrays (fresh writable copies)."""
    objs, lights = _cached(int(n_objects), int(n_lights), float(absorption), int(seed))
    return objs.copy(), lights.copy()
