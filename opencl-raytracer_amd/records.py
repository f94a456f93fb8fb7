"""Device-layout records shared by every backend of the raytracer, and the small
float32 matrix toolkit the scene side needs.

The byte layouts are the reference's device structs (shade_and_reflect_kernel.cl:1-29,
host mirrors OpenCLRaytracer.hpp:25-58):

    ObjectData 320 B : Material(64) | mv(64) | mvInverse(64) | mvInverseTranspose(64) | uint type | 60 B pad
    Light       64 B : ambient | diffuse | specular (float3 in 16 B each) | float4 position
    Ray         32 B : float4 start (w=1) | float4 direction (w=0, not normalised)
    pixel       16 B : float3 colour in a float4 slot

Matrices are column-major (glm::value_ptr order, OpenCLRaytracer.cpp:114-117).

The matrix helpers restate glm's published formulas (glm is an un-pinned vcpkg dependency
of the reference - vcpkg.json:8 - and is not vendored): translate/scale/rotate/lookAt as
used by SceneLoader.cpp:215,270,282,294 and inverse/transpose as used by
ObjectData.cpp:7-8, evaluated in float32.
"""
from __future__ import annotations

import numpy as np

F = np.float32

OBJECT_DTYPE = np.dtype([
    ("ambient", "<f4", 4), ("diffuse", "<f4", 4), ("specular", "<f4", 4),
    ("absorption", "<f4"), ("reflection", "<f4"), ("transparency", "<f4"), ("shininess", "<f4"),
    ("mv", "<f4", 16), ("mvInverse", "<f4", 16), ("mvInverseTranspose", "<f4", 16),
    ("type", "<u4"), ("pad", "u1", 60),
])
LIGHT_DTYPE = np.dtype([
    ("ambient", "<f4", 4), ("diffuse", "<f4", 4), ("specular", "<f4", 4), ("position", "<f4", 4),
])
RAY_DTYPE = np.dtype([("start", "<f4", 4), ("direction", "<f4", 4)])

assert OBJECT_DTYPE.itemsize == 320 and LIGHT_DTYPE.itemsize == 64 and RAY_DTYPE.itemsize == 32

SPHERE, BOX = 0, 1  # ObjectData::PrimativeType (ObjectData.hpp:9-12)
MAX_FLOAT = np.float32(3.402823466e+38)


# ----------------------------------------------------------------------------------------
# Material / Light property bags (Material.hpp:6-11, Light.hpp:5-7)
# ----------------------------------------------------------------------------------------
class Material:
    def __init__(self, ambient=(0, 0, 0), diffuse=(0, 0, 0), specular=(0, 0, 0),
                 absorption=1.0, reflection=0.0, transparency=0.0, shininess=1.0):
        self.ambient = tuple(float(x) for x in ambient)
        self.diffuse = tuple(float(x) for x in diffuse)
        self.specular = tuple(float(x) for x in specular)
        self.absorption = float(absorption)
        self.reflection = float(reflection)
        self.transparency = float(transparency)
        self.shininess = float(shininess)


class LightProperties:
    def __init__(self, ambient=(0, 0, 0), diffuse=(0, 0, 0), specular=(0, 0, 0)):
        self.ambient = tuple(float(x) for x in ambient)
        self.diffuse = tuple(float(x) for x in diffuse)
        self.specular = tuple(float(x) for x in specular)


# ----------------------------------------------------------------------------------------
# float32 column-major 4x4 toolkit; m[c][r] like glm (stored as ndarray shape (4,4), m[c] = column c)
# ----------------------------------------------------------------------------------------
def mat_identity() -> np.ndarray:
    return np.eye(4, dtype=F)


def mat_mul(a: np.ndarray, b: np.ndarray) -> np.ndarray:
    """glm operator*(mat4, mat4): result column j = a[0]*b[j][0] + a[1]*b[j][1] + a[2]*b[j][2] + a[3]*b[j][3]."""
    out = np.zeros((4, 4), dtype=F)
    for j in range(4):
        acc = a[0] * b[j][0]
        acc = acc + a[1] * b[j][1]
        acc = acc + a[2] * b[j][2]
        acc = acc + a[3] * b[j][3]
        out[j] = acc
    return out


def mat_vec(m: np.ndarray, v) -> np.ndarray:
    """glm operator*(mat4, vec4)."""
    v = np.asarray(v, dtype=F)
    # glm: Mov0*m[0] + Mov1*m[1] paired adds: (m0*v0 + m1*v1) + (m2*v2 + m3*v3)
    a0 = m[0] * v[0]
    a1 = m[1] * v[1]
    a2 = m[2] * v[2]
    a3 = m[3] * v[3]
    return ((a0 + a1) + (a2 + a3)).astype(F)


def translate(m: np.ndarray, v) -> np.ndarray:
    v = np.asarray(v, dtype=F)
    out = m.copy()
    out[3] = m[0] * v[0] + m[1] * v[1] + m[2] * v[2] + m[3]
    return out


def scale(m: np.ndarray, v) -> np.ndarray:
    v = np.asarray(v, dtype=F)
    out = m.copy()
    out[0] = m[0] * v[0]
    out[1] = m[1] * v[1]
    out[2] = m[2] * v[2]
    return out


def normalize3(v) -> np.ndarray:
    v = np.asarray(v, dtype=F)
    d = F(v[0] * v[0] + v[1] * v[1] + v[2] * v[2])
    return (v * (F(1) / np.sqrt(d, dtype=F))).astype(F)  # glm: v * inversesqrt(dot(v,v))


def rotate(m: np.ndarray, angle_rad, axis) -> np.ndarray:
    a = F(angle_rad)
    c = F(np.cos(a, dtype=F))
    s = F(np.sin(a, dtype=F))
    axis = normalize3(axis)
    temp = (F(1) - c) * axis
    rot = np.zeros((3, 3), dtype=F)
    rot[0][0] = c + temp[0] * axis[0]
    rot[0][1] = temp[0] * axis[1] + s * axis[2]
    rot[0][2] = temp[0] * axis[2] - s * axis[1]
    rot[1][0] = temp[1] * axis[0] - s * axis[2]
    rot[1][1] = c + temp[1] * axis[1]
    rot[1][2] = temp[1] * axis[2] + s * axis[0]
    rot[2][0] = temp[2] * axis[0] + s * axis[1]
    rot[2][1] = temp[2] * axis[1] - s * axis[0]
    rot[2][2] = c + temp[2] * axis[2]
    out = np.zeros((4, 4), dtype=F)
    out[0] = m[0] * rot[0][0] + m[1] * rot[0][1] + m[2] * rot[0][2]
    out[1] = m[0] * rot[1][0] + m[1] * rot[1][1] + m[2] * rot[1][2]
    out[2] = m[0] * rot[2][0] + m[1] * rot[2][1] + m[2] * rot[2][2]
    out[3] = m[3]
    return out


def radians(deg) -> np.float32:
    return F(F(deg) * F(0.01745329251994329576923690768489))


def look_at(eye, center, up) -> np.ndarray:
    """glm::lookAtRH."""
    eye = np.asarray(eye, dtype=F)
    center = np.asarray(center, dtype=F)
    up = np.asarray(up, dtype=F)
    f = normalize3(center - eye)
    s = normalize3(np.cross(f, up).astype(F))
    u = np.cross(s, f).astype(F)
    out = np.eye(4, dtype=F)
    out[0][0], out[1][0], out[2][0] = s[0], s[1], s[2]
    out[0][1], out[1][1], out[2][1] = u[0], u[1], u[2]
    out[0][2], out[1][2], out[2][2] = -f[0], -f[1], -f[2]
    out[3][0] = -F(np.dot(s, eye))
    out[3][1] = -F(np.dot(u, eye))
    out[3][2] = F(np.dot(f, eye))
    return out


def transpose(m: np.ndarray) -> np.ndarray:
    return np.ascontiguousarray(m.T).astype(F)


def inverse(m: np.ndarray) -> np.ndarray:
    """glm::inverse(mat4): cofactor expansion, then multiply by 1/det."""
    m = m.astype(F)
    c00 = m[2][2] * m[3][3] - m[3][2] * m[2][3]
    c02 = m[1][2] * m[3][3] - m[3][2] * m[1][3]
    c03 = m[1][2] * m[2][3] - m[2][2] * m[1][3]
    c04 = m[2][1] * m[3][3] - m[3][1] * m[2][3]
    c06 = m[1][1] * m[3][3] - m[3][1] * m[1][3]
    c07 = m[1][1] * m[2][3] - m[2][1] * m[1][3]
    c08 = m[2][1] * m[3][2] - m[3][1] * m[2][2]
    c10 = m[1][1] * m[3][2] - m[3][1] * m[1][2]
    c11 = m[1][1] * m[2][2] - m[2][1] * m[1][2]
    c12 = m[2][0] * m[3][3] - m[3][0] * m[2][3]
    c14 = m[1][0] * m[3][3] - m[3][0] * m[1][3]
    c15 = m[1][0] * m[2][3] - m[2][0] * m[1][3]
    c16 = m[2][0] * m[3][2] - m[3][0] * m[2][2]
    c18 = m[1][0] * m[3][2] - m[3][0] * m[1][2]
    c19 = m[1][0] * m[2][2] - m[2][0] * m[1][2]
    c20 = m[2][0] * m[3][1] - m[3][0] * m[2][1]
    c22 = m[1][0] * m[3][1] - m[3][0] * m[1][1]
    c23 = m[1][0] * m[2][1] - m[2][0] * m[1][1]
    fac0 = np.array([c00, c00, c02, c03], dtype=F)
    fac1 = np.array([c04, c04, c06, c07], dtype=F)
    fac2 = np.array([c08, c08, c10, c11], dtype=F)
    fac3 = np.array([c12, c12, c14, c15], dtype=F)
    fac4 = np.array([c16, c16, c18, c19], dtype=F)
    fac5 = np.array([c20, c20, c22, c23], dtype=F)
    v0 = np.array([m[1][0], m[0][0], m[0][0], m[0][0]], dtype=F)
    v1 = np.array([m[1][1], m[0][1], m[0][1], m[0][1]], dtype=F)
    v2 = np.array([m[1][2], m[0][2], m[0][2], m[0][2]], dtype=F)
    v3 = np.array([m[1][3], m[0][3], m[0][3], m[0][3]], dtype=F)
    inv0 = v1 * fac0 - v2 * fac1 + v3 * fac2
    inv1 = v0 * fac0 - v2 * fac3 + v3 * fac4
    inv2 = v0 * fac1 - v1 * fac3 + v3 * fac5
    inv3 = v0 * fac2 - v1 * fac4 + v2 * fac5
    sa = np.array([+1, -1, +1, -1], dtype=F)
    sb = np.array([-1, +1, -1, +1], dtype=F)
    inv = np.stack([inv0 * sa, inv1 * sb, inv2 * sa, inv3 * sb]).astype(F)
    row0 = np.array([inv[0][0], inv[1][0], inv[2][0], inv[3][0]], dtype=F)
    dot0 = m[0] * row0
    det = (dot0[0] + dot0[1]) + (dot0[2] + dot0[3])
    return (inv * (F(1) / F(det))).astype(F)


# ----------------------------------------------------------------------------------------
# record builders
# ----------------------------------------------------------------------------------------
def make_object(prim_type: int, mat: Material, mv: np.ndarray, mv_inverse: np.ndarray | None = None) -> np.ndarray:
    """ObjectData ctor (ObjectData.cpp:4-9) + device conversion (OpenCLRaytracer.cpp:128-134)."""
    rec = np.zeros((), dtype=OBJECT_DTYPE)
    rec["ambient"][:3] = mat.ambient
    rec["diffuse"][:3] = mat.diffuse
    rec["specular"][:3] = mat.specular
    rec["absorption"] = mat.absorption
    rec["reflection"] = mat.reflection
    rec["transparency"] = mat.transparency
    rec["shininess"] = mat.shininess
    mv = np.asarray(mv, dtype=F).reshape(4, 4)
    inv = inverse(mv) if mv_inverse is None else np.asarray(mv_inverse, dtype=F).reshape(4, 4)
    rec["mv"] = mv.reshape(16)
    rec["mvInverse"] = inv.reshape(16)
    rec["mvInverseTranspose"] = transpose(inv).reshape(16)
    rec["type"] = int(prim_type)
    return rec


def make_light(props: LightProperties, modelview: np.ndarray | None = None, position=None) -> np.ndarray:
    """Light ctor (Light.hpp:12-14): position = modelview * (0,0,0,1); or an explicit float4
    (w = 0 selects the directional branch, shade_and_reflect_kernel.cl:195-198)."""
    rec = np.zeros((), dtype=LIGHT_DTYPE)
    rec["ambient"][:3] = props.ambient
    rec["diffuse"][:3] = props.diffuse
    rec["specular"][:3] = props.specular
    if position is not None:
        rec["position"] = np.asarray(position, dtype=F)
    else:
        rec["position"] = mat_vec(modelview, (0, 0, 0, 1))
    return rec


def objects_array(recs) -> np.ndarray:
    recs = list(recs)
    out = np.zeros(len(recs), dtype=OBJECT_DTYPE)
    for i, r in enumerate(recs):
        out[i] = r
    return out


def lights_array(recs) -> np.ndarray:
    recs = list(recs)
    out = np.zeros(len(recs), dtype=LIGHT_DTYPE)
    for i, r in enumerate(recs):
        out[i] = r
    return out
