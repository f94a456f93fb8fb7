"""Tessellation of the reference's two primitives into triangles (BASELINE configs[4]: "roundedCube.txt
tessellated to ~1M triangles").

EXTENSION - the reference has no triangle type, no tessellator and therefore no semantics to match
(SURVEY.md 8d/8f5). The triangle record and its arithmetic are this repo's own spec;
parity for triangle scenes is self-parity: the HIP path against the tests' CPU statement of the same spec.

Record (the reference's 320-byte ObjectData, `type = 2`):
  mv        columns 0,1,2 = vertices v0,v1,v2 in VIEW space (w = 1); column 3 = (0,0,0,1)
  mvInverse column 0 = guard sphere (cx, cy, cz, R): centroid of the vertices and 1.01 x the largest vertex distance
            (float32, rounded up) - the spec only accepts hits whose line passes it AND whose point lies in it
  material  as for the other primitives

Winding: counter-clockwise seen from outside, so normalize((v1-v0) x (v2-v0)) is the outward normal.
"""
from __future__ import annotations

import numpy as np

from .records import BOX, OBJECT_DTYPE, SPHERE

F = np.float32
TRIANGLE = 2


def triangle_records(v0, v1, v2, template: np.ndarray) -> np.ndarray:
    """(n,3) float arrays of view-space vertices + one ObjectData record supplying the material -> n records."""
    v0 = np.asarray(v0, dtype=np.float64).reshape(-1, 3)
    v1 = np.asarray(v1, dtype=np.float64).reshape(-1, 3)
    v2 = np.asarray(v2, dtype=np.float64).reshape(-1, 3)
    n = len(v0)
    out = np.zeros(n, dtype=OBJECT_DTYPE)
    for name in ("ambient", "diffuse", "specular", "absorption", "reflection", "transparency", "shininess"):
        out[name] = template[name]
    f0, f1, f2 = v0.astype(F), v1.astype(F), v2.astype(F)
    mv = np.zeros((n, 4, 4), dtype=F)          # [column][row]
    mv[:, 0, :3], mv[:, 1, :3], mv[:, 2, :3] = f0, f1, f2
    mv[:, 0, 3] = mv[:, 1, 3] = mv[:, 2, 3] = 1.0
    mv[:, 3, 3] = 1.0
    out["mv"] = mv.reshape(n, 16)
    # guard sphere around the ROUNDED vertices (those are the triangle the kernels see)
    d0, d1, d2 = f0.astype(np.float64), f1.astype(np.float64), f2.astype(np.float64)
    c = ((d0 + d1 + d2) / 3.0).astype(F)
    cd = c.astype(np.float64)
    r = np.sqrt(np.maximum.reduce([((d0 - cd) ** 2).sum(1), ((d1 - cd) ** 2).sum(1), ((d2 - cd) ** 2).sum(1)]))
    rf = np.nextafter((r * 1.01).astype(F), F(np.inf))   # 1 % of slack: a hit on the farthest vertex must pass the point test
    inv = np.zeros((n, 4, 4), dtype=F)
    inv[:, 0, :3] = c
    inv[:, 0, 3] = rf
    out["mvInverse"] = inv.reshape(n, 16)
    out["type"] = TRIANGLE
    return out


def _unit_sphere(n_lat: int, n_lon: int):
    """Lat-long triangulation of the unit sphere: 2*n_lon*(n_lat-1) triangles, outward winding."""
    th = np.linspace(0.0, np.pi, n_lat + 1)           # polar angle, 0 = +y pole
    ph = np.linspace(0.0, 2.0 * np.pi, n_lon + 1)[:-1]

    def pt(i, j):
        j = j % n_lon
        return np.array([np.sin(th[i]) * np.cos(ph[j]), np.cos(th[i]), np.sin(th[i]) * np.sin(ph[j])])
    tris = []
    for i in range(n_lat):
        for j in range(n_lon):
            a, b, c, d = pt(i, j), pt(i + 1, j), pt(i + 1, j + 1), pt(i, j + 1)
            if i > 0: tris.append((a, d, b))          # upper triangle (degenerate at the top pole)
            if i < n_lat - 1: tris.append((b, d, c))  # lower triangle (degenerate at the bottom pole)
    return np.array(tris)                               # (n, 3 vertices, 3 coords)


def _unit_box(k: int):
    """The box [-0.5,0.5]^3: 6 faces x k x k quads x 2 triangles, outward winding."""
    g = np.linspace(-0.5, 0.5, k + 1)
    tris = []
    for axis in range(3):
        for sign in (-1.0, 1.0):
            u_ax, v_ax = (axis + 1) % 3, (axis + 2) % 3
            for i in range(k):
                for j in range(k):
                    def p(a, b):
                        q = np.zeros(3)
                        q[axis] = 0.5 * sign
                        q[u_ax], q[v_ax] = g[a], g[b]
                        return q
                    a, b, c, d = p(i, j), p(i + 1, j), p(i + 1, j + 1), p(i, j + 1)
                    if sign > 0: tris += [(a, b, c), (a, c, d)]
                    else: tris += [(a, c, b), (a, d, c)]
    return np.array(tris)


def _outward(tris_obj: np.ndarray, tris_view: np.ndarray, mv: np.ndarray, centre_obj=np.zeros(3)):
    """Flip triangles whose view-space normal points towards the (transformed) object centre - a mirrored
    instance (negative determinant) reverses the winding."""
    c = mv[:3, :3] @ centre_obj + mv[:3, 3]
    n = np.cross(tris_view[:, 1] - tris_view[:, 0], tris_view[:, 2] - tris_view[:, 0])
    mid = tris_view.mean(axis=1)
    flip = ((mid - c) * n).sum(1) < 0
    tris_view[flip] = tris_view[flip][:, [0, 2, 1]]
    return tris_view


def tessellate(objects: np.ndarray, sphere_lat: int = 16, sphere_lon: int = 32, box_k: int = 4) -> np.ndarray:
    """Every sphere / box record of `objects` -> triangle records (view space), materials carried over."""
    out = []
    sph = _unit_sphere(sphere_lat, sphere_lon)
    box = _unit_box(box_k)
    for rec in objects:
        mv = rec["mv"].reshape(4, 4).T.astype(np.float64)        # column-major storage -> matrix
        if int(rec["type"]) == SPHERE: tris = sph
        elif int(rec["type"]) == BOX: tris = box
        else: continue
        tv = tris @ mv[:3, :3].T + mv[:3, 3]
        tv = _outward(tris, tv.copy(), mv)
        out.append(triangle_records(tv[:, 0], tv[:, 1], tv[:, 2], rec))
    return np.concatenate(out) if out else np.zeros(0, dtype=OBJECT_DTYPE)


def subdivision_for(objects: np.ndarray, target_triangles: int):
    """(sphere_lat, sphere_lon, box_k) that brings the scene to about `target_triangles`."""
    n_s = int((objects["type"] == SPHERE).sum())
    n_b = int((objects["type"] == BOX).sum())
    best = None
    for lat in range(2, 2048):
        lon = 2 * lat
        per_s = 2 * lon * (lat - 1)
        k = max(1, int(round(np.sqrt(per_s / 12.0))))
        total = n_s * per_s + n_b * 12 * k * k
        if best is None or abs(total - target_triangles) < abs(best[3] - target_triangles):
            best = (lat, lon, k, total)
        if total > target_triangles: break
    return best[:3]
