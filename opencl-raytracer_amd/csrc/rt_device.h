// rt_device.h - gfx950 device functions of the hot path (ray/primitive tests, hit materialisation,
// Phong + hard-shadow shading, reflection bookkeeping).
//
// What is computed is fixed by the reference kernels (shade_and_reflect_kernel.cl, shade_kernel.cl,
// hittest_kernel.cl); how it is computed is not a translation of them:
//   * objects are re-packed at upload into a 64-byte HOT record (rows x,y,z of mvInverse + type) that the
//     traversal loops read with wave-uniform (scalar, SGPR) loads, and a 128-byte COLD record (mv, row w of
//     mvInverse, material) that is touched once per finished ray;
//   * traversal keeps only (t, index); the hit record is materialised once after the loop (the reference
//     rebuilds it on every improvement, shade_and_reflect_kernel.cl:110-119,161-166) - same winner, same
//     t, same arithmetic, so the same bits;
//   * shadow rays stop at the first occluder with t < 1 (the reference's full closest-hit is only tested
//     with `time >= 1 || time < 0`, :229); shade_and_reflect's last-light-wins colour (:238) is evaluated by
//     scanning the lights backwards; the reflection raycast whose result nothing reads (:268) is skipped.
//     All three are exact for finite (non-NaN) hit times and can be disabled with RT_FLAG_LITERAL.
//
// Floating point contract (re-derived from the reference kernels' LLVM IR): this file is
// compiled with -ffp-contract=off and -fhip-fp32-correctly-rounded-divide-sqrt; the places where the OpenCL
// front-end forms llvm.fmuladd are written fma_<FUSED>(). FUSED=true is a single-rounding v_fma_f32,
// FUSED=false a v_mul_f32 followed by v_add_f32. In a sum of products the first product is fused onto the
// second: t = b*y; t = fma(a,x,t); t = fma(c,z,t); t = fma(d,w,t). dot()/normalize() are the unfused
// left-to-right forms the oracle defines for the OpenCL builtins (oracle/ref_shim.cl).
#pragma once

#include <hip/hip_runtime.h>
#include <stdint.h>

namespace rt {

constexpr float kMaxFloat = 3.402823466e+38F;  // shade_and_reflect_kernel.cl:31

// ---- HBM layouts built by the upload path (rt_api.cpp) -------------------------------------------------
struct HotObject {   // 64 B, 64-B aligned: one s_load_dwordx16 per object per wave
    float4 row0;     // mvInverse row 0: (m[0], m[4], m[8],  m[12])
    float4 row1;     // mvInverse row 1: (m[1], m[5], m[9],  m[13])
    float4 row2;     // mvInverse row 2: (m[2], m[6], m[10], m[14])
    uint32_t type;   // 0 sphere, 1 box, 2 triangle (extension), anything else: never hit (the reference's switch has no default)
    uint32_t pad[3];
    // type 2 (triangle, extension - DESIGN.md section 11): row0 = (v0, cx), row1 = (v1 - v0, cy), row2 = (v2 - v0, cz),
    // pad[0] = bits of R; (c, R) is the record's guard sphere
};
struct ColdObject {  // 128 B
    float4 mv_row[4];    // mv by ROWS: row r = (m[r], m[4 + r], m[8 + r], m[12 + r]) of the column-major upload - one 16-byte
                         // load per output component of `mv * p`, and the last row ((0,0,0,1) for an affine instance) is
                         // only fetched for scenes that have a non-affine one (Scene::affine)
    float4 inv_row3;     // mvInverse row 3: (m[3], m[7], m[11], m[15])
    float4 amb_absorb;   // ambient rgb, absorption
    float4 dif_shine;    // diffuse rgb, shininess
    float4 spec_type;    // specular rgb, type bits
};
// What materialise() needs of an object, in ONE 128-byte line (round 3): rows x,y,z of mvInverse (= the HotObject, which stays
// the walks' 64-byte record), rows x,y,z of mv, and the absorption the loop bookkeeping reads right after. wf_resume was gathering
// these from two records in two lines - by section 4.4's figures a line fill is ~5 CU-cycles of a pixel-step that has ~45.
struct ObjectRecord {
    float4 inv_row[3];   // as HotObject::row0..2 (triangles: (v0, cx), (e1, cy), (e2, cz))
    uint32_t type;
    uint32_t pad0;       // triangles: bits of the guard radius (HotObject::pad[0])
    float absorption;    // mat.absorption (ColdObject::amb_absorb.w)
    uint32_t spare0;
    float4 mv_row[3];    // as ColdObject::mv_row[0..2]
    float4 spare1;
};
struct LightRec {    // the reference's 64-byte Light, unchanged
    float4 ambient, diffuse, specular, position;
};
static_assert(sizeof(HotObject) == 64 && sizeof(ColdObject) == 128 && sizeof(LightRec) == 64 && sizeof(ObjectRecord) == 128, "layout");

// Read-only scene data is addressed through the constant address space: that tells the compiler the bytes
// cannot change during the kernel, so a wave-uniform address becomes a scalar load (s_load_dwordx4/8/16) even
// when the kernel has already stored pixels (persistent workgroups loop over many pixel blocks, and stores
// would otherwise demote every later uniform load to a 64-lane vector load of one address).
#define RT_CONST __attribute__((address_space(4)))
typedef float f4 __attribute__((ext_vector_type(4)));
struct HotObjectC {  // HotObject with native vectors (plain loads from the constant address space)
    f4 row0, row1, row2;
    uint32_t type;
    uint32_t pad[3];
};

struct Ray {
    float sx, sy, sz, sw;
    float dx, dy, dz, dw;
};

struct HitRec {
    float px, py, pz, pw;  // view-space intersection (float4; w is carried like the reference does)
    float nx, ny, nz;      // normalised view-space normal
    float rx, ry, rz;      // reflection of the incoming direction
    int index;
};

struct Counters {
    unsigned long long traced;     // rays actually issued
    unsigned long long reference;  // rays the reference semantics trace
    unsigned long long hits;       // primary hits
    unsigned long long tests;      // ray-object tests executed by the wavefront traversal kernels (2 per pair visited)
    // grid walk, counted pass only; [0] closest-hit rays, [1] shadow rays:
    // rays, wave trips, live lane-trips, cell fetches, pre-tests, exact tests, exact-test rounds, hand-out rounds
    unsigned long long walk[2][8];
};

// ---- arithmetic primitives -----------------------------------------------------------------------------
template <bool FUSED>
__device__ __forceinline__ float fma_(float a, float b, float c) {
    if constexpr (FUSED) return __builtin_fmaf(a, b, c);
    else return a * b + c;  // -ffp-contract=off: two roundings
}

__device__ __forceinline__ float dot3(float ax, float ay, float az, float bx, float by, float bz) {
    float s = ax * bx;
    s = s + ay * by;
    s = s + az * bz;
    return s;
}

__device__ __forceinline__ void normalize3(float& x, float& y, float& z) {
    float s = x * x;
    s = s + y * y;
    s = s + z * z;
    const float len = __builtin_sqrtf(s);
    x = x / len;
    y = y / len;
    z = z / len;
}

// RT_FLAG_FAST_PHONG (opt-in): normalisation of a vector that feeds COLOUR only - the shading normal, the view vector, the
// reflected light vector of the Phong term; never a vector a ray is built from. dot() as the reference's, then the
// hardware reciprocal square root (1 ulp) with one Newton step and three multiplies instead of an IEEE square root and
// three IEEE divisions (~8 instead of ~40 instructions); relative error ~1e-7, far inside the 1e-5 colour tolerance.
__device__ __forceinline__ void normalize3_shading(bool fast, float& x, float& y, float& z) {
    if (!fast) { normalize3(x, y, z); return; }
    float s = x * x;
    s = s + y * y;
    s = s + z * z;
    float r = __builtin_amdgcn_rsqf(s);
    r = r * __builtin_fmaf(-0.5f * s, r * r, 1.5f);
    x = x * r;
    y = y * r;
    z = z * r;
}

// one row of transform(): m0*x + m1*y + m2*z + m3*w in the reference's association
template <bool FUSED>
__device__ __forceinline__ float row4(float m0, float m1, float m2, float m3, float x, float y, float z, float w) {
    float t = m1 * y;
    t = fma_<FUSED>(m0, x, t);
    t = fma_<FUSED>(m2, z, t);
    t = fma_<FUSED>(m3, w, t);
    return t;
}
// the same row for a vector whose w is exactly 0 (shadow / reflection / pinhole directions): the dropped
// fma(m3, 0, t) can only change the sign of a zero, which no later operation observes
template <bool FUSED>
__device__ __forceinline__ float row3(float m0, float m1, float m2, float x, float y, float z) {
    float t = m1 * y;
    t = fma_<FUSED>(m0, x, t);
    t = fma_<FUSED>(m2, z, t);
    return t;
}

// ---- primitive tests in object space ---------------------------------------------------------------------
// unit sphere (shade_and_reflect_kernel.cl:82-106): true if the candidate passes the reference's
// `radical < 0` and `tMin < 0` rejections; t = the reference's tMin
template <bool FUSED>
__device__ __forceinline__ bool sphere_candidate(float sx, float sy, float sz, float dx, float dy, float dz, float& t) {
    float A = dy * dy;
    A = fma_<FUSED>(dx, dx, A);
    A = fma_<FUSED>(dz, dz, A);
    float B = sy * dy;
    B = fma_<FUSED>(dx, sx, B);
    B = fma_<FUSED>(dz, sz, B);
    B = B * 2.0f;
    float C = sy * sy;
    C = fma_<FUSED>(sx, sx, C);
    C = fma_<FUSED>(sz, sz, C);
    C = C + -1.0f;
    const float radical = fma_<FUSED>(B, B, (A * 4.0f) * (-C));
    if (radical < 0) return false;
    const float root = __builtin_sqrtf(radical);
    const float den = A * 2.0f;
    const float t1 = (-B - root) / den;
    const float t2 = (root - B) / den;
    const float tMin = (t1 >= 0 && t2 >= 0) ? __builtin_fminf(t1, t2) : __builtin_fmaxf(t1, t2);
    if (tMin < 0) return false;
    t = tMin;
    return true;
}

// one slab of the unit box (shade_and_reflect_kernel.cl:33-58)
__device__ __forceinline__ bool box_slab(float& tmin, float& tmax, float start, float dir) {
    float t1 = -0.5f - start;
    float t2 = 0.5f - start;
    if (dir == 0) {
        if (__builtin_copysignf(t1, t2) == t1) return false;
        tmin = -kMaxFloat;
        tmax = kMaxFloat;
        return true;
    }
    t1 = t1 / dir;
    t2 = t2 / dir;
    if (dir < 0) {
        tmin = __builtin_fminf(t1, t2);
        tmax = __builtin_fmaxf(t1, t2);
    } else {
        tmin = t1;
        tmax = t2;
    }
    return true;
}

// Cheap rejection before the six IEEE divides of the slab test: if the ray's LINE passes the box centre at
// more than 1 object unit it cannot touch the box, whose farthest point is sqrt(0.75) = 0.866 away. disc / A
// = 1 - b^2 for the unit sphere; a box hit needs 1 - b^2 >= 0.25, the rounding error of this expression is
// ~5e-7 * |s|^2, so with |s|^2 < 1e5 a negative value leaves a >= 5x margin. NaNs fall through to the full test.
__device__ __forceinline__ bool box_line_misses(float sx, float sy, float sz, float dx, float dy, float dz) {
    const float ss = sx * sx + sy * sy + sz * sz;
    const float A = dx * dx + dy * dy + dz * dz;
    const float B = sx * dx + sy * dy + sz * dz;
    const float disc = B * B - A * (ss - 1.0f);
    return ss < 1.0e5f && disc < 0.0f;
}

// EXTENSION, no reference semantics (spec: DESIGN.md section 11; the tests hold a CPU statement of it):
// triangle with vertices in view space. fp32, no contraction in either flavour, sums left to right, IEEE division.
// A ray is only tested if its line passes the record's guard sphere, which makes every accepted hit local.
__device__ __forceinline__ bool triangle_candidate(float v0x, float v0y, float v0z, float e1x, float e1y, float e1z, float e2x,
                                                   float e2y, float e2z, float cx, float cy, float cz, float R, const Ray& ray,
                                                   float& t) {
    const float ocx = cx - ray.sx, ocy = cy - ray.sy, ocz = cz - ray.sz;
    const float gx = ocy * ray.dz - ocz * ray.dy;
    const float gy = ocz * ray.dx - ocx * ray.dz;
    const float gz = ocx * ray.dy - ocy * ray.dx;
    const float dd = ray.dx * ray.dx + ray.dy * ray.dy + ray.dz * ray.dz;
    if (gx * gx + gy * gy + gz * gz > (R * R) * dd) return false;  // the line passes the guard sphere's centre further than R
    const float px = ray.dy * e2z - ray.dz * e2y;
    const float py = ray.dz * e2x - ray.dx * e2z;
    const float pz = ray.dx * e2y - ray.dy * e2x;
    const float det = e1x * px + e1y * py + e1z * pz;
    if (!(det != 0)) return false;
    const float inv = 1.0f / det;
    const float tx = ray.sx - v0x, ty = ray.sy - v0y, tz = ray.sz - v0z;
    const float u = (tx * px + ty * py + tz * pz) * inv;
    if (!(u >= 0 && u <= 1)) return false;
    const float qx = ty * e1z - tz * e1y;
    const float qy = tz * e1x - tx * e1z;
    const float qz = tx * e1y - ty * e1x;
    const float v = (ray.dx * qx + ray.dy * qy + ray.dz * qz) * inv;
    if (!(v >= 0 && u + v <= 1)) return false;
    const float tt = (e2x * qx + e2y * qy + e2z * qz) * inv;
    if (!(tt >= 0)) return false;
    // the hit point itself must lie in the guard sphere (for a ray almost in the triangle's plane t is rounding noise)
    const float hx = (ray.sx + tt * ray.dx) - cx, hy = (ray.sy + tt * ray.dy) - cy, hz = (ray.sz + tt * ray.dz) - cz;
    if (!(hx * hx + hy * hy + hz * hz <= R * R)) return false;
    t = tt;
    return true;
}

// unit box [-0.5,0.5]^3 (shade_and_reflect_kernel.cl:123-144)
__device__ __forceinline__ bool box_candidate(float sx, float sy, float sz, float dx, float dy, float dz, float& t) {
    if (box_line_misses(sx, sy, sz, dx, dy, dz)) return false;
    float txMin, txMax, tyMin, tyMax, tzMin, tzMax;
    if (!box_slab(txMin, txMax, sx, dx)) return false;
    if (!box_slab(tyMin, tyMax, sy, dy)) return false;
    if (!box_slab(tzMin, tzMax, sz, dz)) return false;
    const float tMin = __builtin_fmaxf(__builtin_fmaxf(txMin, tyMin), tzMin);
    const float tMax = __builtin_fminf(__builtin_fminf(txMax, tyMax), tzMax);
    if (tMax < tMin) return false;
    const float tHit = (tMin >= 0 && tMax >= 0) ? __builtin_fminf(tMin, tMax) : __builtin_fmaxf(tMin, tMax);
    if (tHit < 0) return false;
    t = tHit;
    return true;
}

// ---- traversal -------------------------------------------------------------------------------------------
// Objects are streamed as PAIRS (HotPair, 128 B: every mvInverse entry of objects 2p and 2p+1 side by side).
// The pair index is wave-uniform, so a pair arrives through two scalar loads and each entry pair sits in an
// aligned SGPR pair. One v_pk_fma_f32 / v_pk_mul_f32 then does the same multiply-add for BOTH objects:
// measured on gfx950 a VALU instruction with an SGPR source issues at ~4.4 cycles per wave against ~2.3 for an
// all-VGPR one, while the packed form with the same SGPR source also costs ~4.3 cycles but retires two
// objects (tools/ubench/valu_issue_rate.hip). Each packed lane is an ordinary IEEE fp32 mul / fma, so every bit of
// s_obj, d_obj and the discriminant is identical to the one-object-at-a-time evaluation.
//
// The packed part is only a FILTER: it computes the reference's `radical` for both spheres of a pair; lanes
// where `radical < 0` is false for either object (a candidate, or a NaN) take the rare path, which runs the
// reference's full sphere test (sqrt, two IEEE divides, root choice, tie rule) on the same object-space ray.
// Pairs that contain a box (or an unknown type) take the generic per-object path.
typedef float f2 __attribute__((ext_vector_type(2)));

struct HotPair {       // 128 B, 128-B aligned
    f2 m[12];          // m[4*r + c] = (A.mvInverse[r][c], B.mvInverse[r][c]) for rows r = 0..2
    uint32_t type_a, type_b;  // 0xffffffff pads an odd object count (never hit)
    uint32_t pad[6];
};
static_assert(sizeof(HotPair) == 128, "layout");

struct RaySplat {      // the ray with every component duplicated into a VGPR pair (built once per ray)
    f2 sx, sy, sz, sw, dx, dy, dz, dw;
};

__device__ __forceinline__ RaySplat splat(const Ray& r) {
    RaySplat q;
    q.sx = f2{r.sx, r.sx}; q.sy = f2{r.sy, r.sy}; q.sz = f2{r.sz, r.sz}; q.sw = f2{r.sw, r.sw};
    q.dx = f2{r.dx, r.dx}; q.dy = f2{r.dy, r.dy}; q.dz = f2{r.dz, r.dz}; q.dw = f2{r.dw, r.dw};
    return q;
}

template <bool FUSED>
__device__ __forceinline__ f2 fma2_(f2 a, f2 b, f2 c) {
    if constexpr (FUSED) return __builtin_elementwise_fma(a, b, c);
    else return a * b + c;
}

// both objects of a pair: ray -> object space (rows x,y,z), the reference's association (row4 / row3)
template <bool FUSED, bool DW0>
__device__ __forceinline__ void pair_object_space(const HotPair& h, const RaySplat& r, f2& sx, f2& sy, f2& sz, f2& dx,
                                                  f2& dy, f2& dz) {
    sx = h.m[1] * r.sy; sx = fma2_<FUSED>(h.m[0], r.sx, sx); sx = fma2_<FUSED>(h.m[2], r.sz, sx); sx = fma2_<FUSED>(h.m[3], r.sw, sx);
    sy = h.m[5] * r.sy; sy = fma2_<FUSED>(h.m[4], r.sx, sy); sy = fma2_<FUSED>(h.m[6], r.sz, sy); sy = fma2_<FUSED>(h.m[7], r.sw, sy);
    sz = h.m[9] * r.sy; sz = fma2_<FUSED>(h.m[8], r.sx, sz); sz = fma2_<FUSED>(h.m[10], r.sz, sz); sz = fma2_<FUSED>(h.m[11], r.sw, sz);
    dx = h.m[1] * r.dy; dx = fma2_<FUSED>(h.m[0], r.dx, dx); dx = fma2_<FUSED>(h.m[2], r.dz, dx);
    dy = h.m[5] * r.dy; dy = fma2_<FUSED>(h.m[4], r.dx, dy); dy = fma2_<FUSED>(h.m[6], r.dz, dy);
    dz = h.m[9] * r.dy; dz = fma2_<FUSED>(h.m[8], r.dx, dz); dz = fma2_<FUSED>(h.m[10], r.dz, dz);
    if constexpr (!DW0) {
        dx = fma2_<FUSED>(h.m[3], r.dw, dx);
        dy = fma2_<FUSED>(h.m[7], r.dw, dy);
        dz = fma2_<FUSED>(h.m[11], r.dw, dz);
    }
}

// the reference's `radical` (shade_and_reflect_kernel.cl:85-94) for both spheres of a pair
template <bool FUSED>
__device__ __forceinline__ f2 pair_radical(f2 sx, f2 sy, f2 sz, f2 dx, f2 dy, f2 dz) {
    f2 A = dy * dy; A = fma2_<FUSED>(dx, dx, A); A = fma2_<FUSED>(dz, dz, A);
    f2 B = sy * dy; B = fma2_<FUSED>(dx, sx, B); B = fma2_<FUSED>(dz, sz, B); B = B * 2.0f;
    f2 C = sy * sy; C = fma2_<FUSED>(sx, sx, C); C = fma2_<FUSED>(sz, sz, C); C = C + -1.0f;
    return fma2_<FUSED>(B, B, (A * 4.0f) * (-C));
}

// one object of a pair, generic type, on an object-space ray: closest-hit update with the reference's tie rules
template <bool FUSED>
__device__ __forceinline__ void closest_update(uint32_t type, float sx, float sy, float sz, float dx, float dy, float dz,
                                               int k, float& T, int& index) {
    float t;
    if (type == 0u) {
        if (sphere_candidate<FUSED>(sx, sy, sz, dx, dy, dz, t)) {
            if (!(T < t)) { T = t; index = k; }   // ties: the later sphere wins (Q3)
        }
    } else if (type == 1u) {
        if (box_candidate(sx, sy, sz, dx, dy, dz, t)) {
            if (!(T <= t)) { T = t; index = k; }  // ties: the earlier object wins (Q3)
        }
    }
}

template <bool FUSED, bool DW0>
__device__ __forceinline__ void closest_pair(const HotPair& h, int p, const RaySplat& r, float& T, int& index) {
    f2 sx, sy, sz, dx, dy, dz;
    pair_object_space<FUSED, DW0>(h, r, sx, sy, sz, dx, dy, dz);
    if ((h.type_a | h.type_b) == 0u) {  // two spheres: packed discriminant filter
        const f2 rad = pair_radical<FUSED>(sx, sy, sz, dx, dy, dz);
        const bool ca = !(rad.x < 0), cb = !(rad.y < 0);
        if (ca || cb) {
            if (ca) closest_update<FUSED>(0u, sx.x, sy.x, sz.x, dx.x, dy.x, dz.x, 2 * p, T, index);
            if (cb) closest_update<FUSED>(0u, sx.y, sy.y, sz.y, dx.y, dy.y, dz.y, 2 * p + 1, T, index);
        }
    } else {
        closest_update<FUSED>(h.type_a, sx.x, sy.x, sz.x, dx.x, dy.x, dz.x, 2 * p, T, index);
        closest_update<FUSED>(h.type_b, sx.y, sy.y, sz.y, dx.y, dy.y, dz.y, 2 * p + 1, T, index);
    }
}

// L2 warm-up for the pair stream. Scalar loads go K$ -> L2 -> Infinity Cache / HBM; a 100k-object stream
// (6.4 MB) does not fit one XCD's 4 MB L2 and every wave walks it at the same pace, so without help each pair
// costs one ~900-cycle miss per wave (measured: 20 ms for ONE wave over 50k pairs). Every 32 pairs each lane
// therefore touches one 64-B line of the stream kWarmAhead pairs ahead with an ordinary vector load (64 lanes
// x 64 B = 32 pairs); the value is only xor-ed into a sink one period later, so the wait lands long after the
// data has arrived. The scalar loads of the traversal then hit L2.
constexpr uint32_t kWarmPeriod = 32;   // pairs covered by one wave-wide touch
constexpr uint32_t kWarmAhead = 128;   // pairs of look-ahead

struct Warm {
    uint32_t pending = 0, sink = 0;
};

__device__ __forceinline__ void l2_warm(const HotPair* __restrict__ pairs, uint32_t n_pairs, uint32_t p, Warm& w) {
    w.sink ^= w.pending;  // consumes the touch issued one period ago (long since landed)
    uint32_t line = (p + kWarmAhead) * 2u + (__lane_id() & 63u);  // 64-B lines, two per pair
    const uint32_t last = n_pairs * 2u - 1u;
    line = line < last ? line : last;                              // clamp instead of branching
    w.pending = *reinterpret_cast<const uint32_t*>(reinterpret_cast<const char*>(pairs) + (uint64_t)line * 64u);
}
__device__ __forceinline__ void l2_warm_finish(Warm& w) {
    asm volatile("" ::"v"(w.sink ^ w.pending));  // keep the touches alive; nothing is computed from them
}

// Closest hit over all objects in ascending index order (the reference's raycast(), :72-177, keeping only
// (t, index)). Two pair records are kept in flight: the next one is requested before the current one is used.
template <bool FUSED, bool DW0>
__device__ __forceinline__ void closest_hit(const HotPair* __restrict__ pairs, uint32_t n_pairs, const Ray& ray, float& T,
                                            int& index) {
    if (n_pairs == 0) return;
    const RaySplat r = splat(ray);
    Warm warm;
    HotPair a = pairs[0];
    uint32_t p = 0;
    for (; p + 1 < n_pairs; p += 2) {
        if ((p & (kWarmPeriod - 1u)) == 0u) l2_warm(pairs, n_pairs, p, warm);
        const HotPair b = pairs[p + 1];
        closest_pair<FUSED, DW0>(a, (int)p, r, T, index);
        a = pairs[(p + 2 < n_pairs) ? p + 2 : p + 1];
        closest_pair<FUSED, DW0>(b, (int)p + 1, r, T, index);
    }
    if (p < n_pairs) closest_pair<FUSED, DW0>(a, (int)p, r, T, index);
    l2_warm_finish(warm);
}

template <bool FUSED>
__device__ __forceinline__ bool occludes(uint32_t type, float sx, float sy, float sz, float dx, float dy, float dz) {
    float t;
    bool cand = false;
    if (type == 0u) cand = sphere_candidate<FUSED>(sx, sy, sz, dx, dy, dz, t);
    else if (type == 1u) cand = box_candidate(sx, sy, sz, dx, dy, dz, t);
    // `!(t >= 1)`, not `t < 1`: a NaN time (a shadow ray with a NaN in it - a light exactly at the hit point, a zero
    // directional light) is accepted by every sphere / box of the reference's loop, the final time is NaN, and its
    // `time >= 1 || time < 0` (:229) then reports the light as blocked
    return cand && !(t >= 1.f);
}

template <bool FUSED>
__device__ __forceinline__ bool any_hit_pair(const HotPair& h, const RaySplat& r) {
    f2 sx, sy, sz, dx, dy, dz;
    pair_object_space<FUSED, true>(h, r, sx, sy, sz, dx, dy, dz);
    bool occ = false;
    if ((h.type_a | h.type_b) == 0u) {
        const f2 rad = pair_radical<FUSED>(sx, sy, sz, dx, dy, dz);
        const bool ca = !(rad.x < 0), cb = !(rad.y < 0);
        if (ca || cb) {
            if (ca) occ = occludes<FUSED>(0u, sx.x, sy.x, sz.x, dx.x, dy.x, dz.x);
            if (cb && !occ) occ = occludes<FUSED>(0u, sx.y, sy.y, sz.y, dx.y, dy.y, dz.y);
        }
    } else {
        occ = occludes<FUSED>(h.type_a, sx.x, sy.x, sz.x, dx.x, dy.x, dz.x);
        if (!occ) occ = occludes<FUSED>(h.type_b, sx.y, sy.y, sz.y, dx.y, dy.y, dz.y);
    }
    return occ;
}

// Any accepted candidate with t < 1 (shadow rays; direction = un-normalised light vector, so t in [0,1)
// means an occluder between the point and the light, shade_and_reflect_kernel.cl:201-209,229). The loop stays
// wave-uniform (no per-lane break: lanes that already found their occluder just ride along) and the wave
// leaves as soon as every active lane has one.
template <bool FUSED>
__device__ __forceinline__ bool any_hit_before_one(const HotPair* __restrict__ pairs, uint32_t n_pairs, const Ray& ray,
                                                   uint32_t* pairs_visited = nullptr) {
    if (pairs_visited) *pairs_visited = 0;
    if (n_pairs == 0) return false;
    const RaySplat r = splat(ray);
    bool occluded = false, done = false;
    Warm warm;
    HotPair a = pairs[0];
    uint32_t p = 0;
    for (; p + 1 < n_pairs; p += 2) {
        if ((p & (kWarmPeriod - 1u)) == 0u) l2_warm(pairs, n_pairs, p, warm);
        const HotPair b = pairs[p + 1];
        occluded |= any_hit_pair<FUSED>(a, r);
        a = pairs[(p + 2 < n_pairs) ? p + 2 : p + 1];
        occluded |= any_hit_pair<FUSED>(b, r);
        if (__ballot(!occluded) == 0ull) { done = true; p += 2; break; }  // wave-uniform exit
    }
    if (!done && p < n_pairs) {  // odd pair count: one record left
        if (__ballot(!occluded) != 0ull) { occluded |= any_hit_pair<FUSED>(a, r); p = n_pairs; }
    }
    if (pairs_visited) *pairs_visited = p;
    l2_warm_finish(warm);
    return occluded;
}

// ---- small scenes: one object at a time, optional per-wave candidate mask -----------------------------------
// With a handful of objects the pair stream buys nothing (most of the time goes into shading, and the wide
// loops cost registers there). The monolithic kernel therefore walks HotObject records one by one; `mask`
// (wave-uniform, bit k = object k may be hit by some ray of this wave) skips objects that the per-tile bounding
// test (rt_kernels.hip) has ruled out for the whole wave. Object order stays ascending, so the tie rules hold.
template <bool FUSED, bool DW0>
__device__ __forceinline__ void object_space_one(const RT_CONST HotObjectC* o, const Ray& ray, float& sx, float& sy,
                                                 float& sz, float& dx, float& dy, float& dz) {
    const f4 r0 = o->row0, r1 = o->row1, r2 = o->row2;
    sx = row4<FUSED>(r0.x, r0.y, r0.z, r0.w, ray.sx, ray.sy, ray.sz, ray.sw);
    sy = row4<FUSED>(r1.x, r1.y, r1.z, r1.w, ray.sx, ray.sy, ray.sz, ray.sw);
    sz = row4<FUSED>(r2.x, r2.y, r2.z, r2.w, ray.sx, ray.sy, ray.sz, ray.sw);
    if constexpr (DW0) {
        dx = row3<FUSED>(r0.x, r0.y, r0.z, ray.dx, ray.dy, ray.dz);
        dy = row3<FUSED>(r1.x, r1.y, r1.z, ray.dx, ray.dy, ray.dz);
        dz = row3<FUSED>(r2.x, r2.y, r2.z, ray.dx, ray.dy, ray.dz);
    } else {
        dx = row4<FUSED>(r0.x, r0.y, r0.z, r0.w, ray.dx, ray.dy, ray.dz, ray.dw);
        dy = row4<FUSED>(r1.x, r1.y, r1.z, r1.w, ray.dx, ray.dy, ray.dz, ray.dw);
        dz = row4<FUSED>(r2.x, r2.y, r2.z, r2.w, ray.dx, ray.dy, ray.dz, ray.dw);
    }
}

template <bool FUSED, bool DW0>
__device__ __forceinline__ void closest_hit_small(const HotObject* __restrict__ hot_, uint32_t n, bool use_mask, uint64_t mask,
                                                  const Ray& ray, float& T, int& index) {
    const RT_CONST HotObjectC* hot = (const RT_CONST HotObjectC*)(hot_);
    if (use_mask) {
        while (mask) {
            const int k = __builtin_ctzll(mask);
            mask &= mask - 1ull;
            float sx, sy, sz, dx, dy, dz;
            object_space_one<FUSED, DW0>(hot + k, ray, sx, sy, sz, dx, dy, dz);
            closest_update<FUSED>(hot[k].type, sx, sy, sz, dx, dy, dz, k, T, index);
        }
    } else {
        for (uint32_t k = 0; k < n; ++k) {
            float sx, sy, sz, dx, dy, dz;
            object_space_one<FUSED, DW0>(hot + k, ray, sx, sy, sz, dx, dy, dz);
            closest_update<FUSED>(hot[k].type, sx, sy, sz, dx, dy, dz, (int)k, T, index);
        }
    }
}

template <bool FUSED>
__device__ __forceinline__ bool any_hit_small(const HotObject* __restrict__ hot_, uint32_t n, const Ray& ray) {
    const RT_CONST HotObjectC* hot = (const RT_CONST HotObjectC*)(hot_);
    bool occluded = false;
    for (uint32_t k = 0; k < n; ++k) {
        float sx, sy, sz, dx, dy, dz;
        object_space_one<FUSED, true>(hot + k, ray, sx, sy, sz, dx, dy, dz);
        occluded |= occludes<FUSED>(hot[k].type, sx, sy, sz, dx, dy, dz);
        if (__ballot(!occluded) == 0ull) break;
    }
    return occluded;
}

// ---- hit materialisation (once per finished ray) ---------------------------------------------------------
// intersection = mv * p, normal = normalize((mv * (n,0)).xyz), reflection = reflect(dir, normal)
// (shade_and_reflect_kernel.cl:112-118 sphere, :149-165 box, :175)
// rows x,y,z of an object's mvInverse as materialise() loaded them (+ type and the triangle's guard radius): handed on to a
// caller that is about to run the reference's exact test against the SAME object (the reflection ray's own object,
// rt_wavefront.hip: begin_shade_lit), so that the 64-byte record is not gathered a second time
struct ObjRows {
    float4 r0, r1, r2;
    uint32_t type, pad0;
};

template <bool FUSED>
__device__ __forceinline__ void materialise(const ObjectRecord* __restrict__ objrec, const ColdObject* __restrict__ cold,
                                            int index, float t, const Ray& ray, HitRec& h, bool affine = false, ObjRows* rows = nullptr,
                                            float* absorption = nullptr) {
    const ObjectRecord* o = objrec + index;
    const ColdObject* c = cold + index;   // (only touched for the bottom rows of a non-affine instance)
    const uint32_t type = o->type;
    if (rows) { rows->r0 = o->inv_row[0]; rows->r1 = o->inv_row[1]; rows->r2 = o->inv_row[2]; rows->type = type; rows->pad0 = o->pad0; }
    if (absorption) *absorption = o->absorption;
    if (type == 2u) {  // triangle (extension): view-space point on the ray, normal = normalize(e1 x e2)
        const float4 e1 = o->inv_row[1], e2 = o->inv_row[2];
        h.px = fma_<FUSED>(t, ray.dx, ray.sx);
        h.py = fma_<FUSED>(t, ray.dy, ray.sy);
        h.pz = fma_<FUSED>(t, ray.dz, ray.sz);
        h.pw = fma_<FUSED>(t, ray.dw, ray.sw);
        float nx = e1.y * e2.z - e1.z * e2.y;
        float ny = e1.z * e2.x - e1.x * e2.z;
        float nz = e1.x * e2.y - e1.y * e2.x;
        normalize3(nx, ny, nz);
        h.nx = nx; h.ny = ny; h.nz = nz;
        const float k2 = dot3(ray.dx, ray.dy, ray.dz, nx, ny, nz) * -2.0f;
        h.rx = fma_<FUSED>(k2, nx, ray.dx);
        h.ry = fma_<FUSED>(k2, ny, ray.dy);
        h.rz = fma_<FUSED>(k2, nz, ray.dz);
        h.index = index;
        return;
    }
    const float4 r0 = o->inv_row[0], r1 = o->inv_row[1], r2 = o->inv_row[2];
    const float sx = row4<FUSED>(r0.x, r0.y, r0.z, r0.w, ray.sx, ray.sy, ray.sz, ray.sw);
    const float sy = row4<FUSED>(r1.x, r1.y, r1.z, r1.w, ray.sx, ray.sy, ray.sz, ray.sw);
    const float sz = row4<FUSED>(r2.x, r2.y, r2.z, r2.w, ray.sx, ray.sy, ray.sz, ray.sw);
    const float dx = row4<FUSED>(r0.x, r0.y, r0.z, r0.w, ray.dx, ray.dy, ray.dz, ray.dw);
    const float dy = row4<FUSED>(r1.x, r1.y, r1.z, r1.w, ray.dx, ray.dy, ray.dz, ray.dw);
    const float dz = row4<FUSED>(r2.x, r2.y, r2.z, r2.w, ray.dx, ray.dy, ray.dz, ray.dw);
    // Row w of the transforms. With bottom rows (0,0,0,1) - every instance of an `affine` scene, checked at upload - the
    // reference's own expression 0*y + 0*x + 0*z + 1*w returns w for finite x, y, z (the zeros can only change the sign of a
    // zero): the two 16-byte rows are not fetched and the identity is used instead.
    float sw, dw;
    if (affine) {
        sw = ray.sw;
        dw = ray.dw;
    } else {
        const float4 r3 = c->inv_row3;
        sw = row4<FUSED>(r3.x, r3.y, r3.z, r3.w, ray.sx, ray.sy, ray.sz, ray.sw);
        dw = row4<FUSED>(r3.x, r3.y, r3.z, r3.w, ray.dx, ray.dy, ray.dz, ray.dw);
    }
    const float px = fma_<FUSED>(t, dx, sx);
    const float py = fma_<FUSED>(t, dy, sy);
    const float pz = fma_<FUSED>(t, dz, sz);
    const float pw = fma_<FUSED>(t, dw, sw);
    float ox, oy, oz;  // object-space normal
    if (type == 0u) {
        ox = px; oy = py; oz = pz;
    } else {
        ox = 0.f; oy = 0.f; oz = 0.f;
        if (px > 0.4998f) ox += 1.f; else if (px < -0.4998f) ox -= 1.f;
        if (py > 0.4998f) oy += 1.f; else if (py < -0.4998f) oy -= 1.f;
        if (pz > 0.4998f) oz += 1.f; else if (pz < -0.4998f) oz -= 1.f;
    }
    const float4 m0 = o->mv_row[0], m1 = o->mv_row[1], m2 = o->mv_row[2];
    h.px = row4<FUSED>(m0.x, m0.y, m0.z, m0.w, px, py, pz, pw);
    h.py = row4<FUSED>(m1.x, m1.y, m1.z, m1.w, px, py, pz, pw);
    h.pz = row4<FUSED>(m2.x, m2.y, m2.z, m2.w, px, py, pz, pw);
    if (affine) {
        h.pw = pw;
    } else {
        const float4 m3 = c->mv_row[3];
        h.pw = row4<FUSED>(m3.x, m3.y, m3.z, m3.w, px, py, pz, pw);
    }
    float nx = row4<FUSED>(m0.x, m0.y, m0.z, m0.w, ox, oy, oz, 0.f);
    float ny = row4<FUSED>(m1.x, m1.y, m1.z, m1.w, ox, oy, oz, 0.f);
    float nz = row4<FUSED>(m2.x, m2.y, m2.z, m2.w, ox, oy, oz, 0.f);
    normalize3(nx, ny, nz);
    h.nx = nx; h.ny = ny; h.nz = nz;
    const float k2 = dot3(ray.dx, ray.dy, ray.dz, nx, ny, nz) * -2.0f;
    h.rx = fma_<FUSED>(k2, nx, ray.dx);
    h.ry = fma_<FUSED>(k2, ny, ray.dy);
    h.rz = fma_<FUSED>(k2, nz, ray.dz);
    h.index = index;
}

// ---- rays with a NaN in them ---------------------------------------------------------------------------------------
// Every rejection in the reference's tests is a comparison, and a NaN fails them all, so its loop does not skip
// objects for such a ray - what it ends with depends only on the LAST sphere / box of the scene (types >= 2 never hit):
//  * a NaN reaches the object-space DIRECTION of every object (any NaN in the view-space direction: 0 * NaN = NaN in
//    every row of transform()): every sphere and every box accepts with time = NaN -> the last one leaves NaN;
//  * the direction is exactly (0,0,0) and the start is NaN - the shadow ray of a light AT the hit point or of a zero
//    directional light, start = P + 0.01 * normalize(0) (.cl:195-205): a sphere accepts with NaN, a box takes the
//    `dir == 0` branch of all three slabs (:36-43; copysign(NaN, .) == NaN is false) -> tHit = MAX_FLOAT, accepted iff
//    the time so far is NaN -> a last sphere leaves NaN, a last box leaves MAX_FLOAT.
// Shadow rays (built from finite scene data) can only be of these two kinds; a NaN start with a finite non-zero
// direction (possible for caller-made primary rays) is outside this closed form and outside the grid path's domain.
// Rays with an infinite component are not classified (the outcome then depends on each object's matrix).
// The brute-force loops need none of this for closest-hit rays (they run the reference's loop in order); the order-free
// parts - any-hit shadow tests, the grid - ask here first.
enum : int { kNanRayNone = 0, kNanRayTimeNaN = 1, kNanRayTimeMax = 2 };
__device__ __forceinline__ int nan_ray_outcome(const Ray& ray, int nan_winner, uint32_t winner_is_sphere) {
    const float s = ((ray.sx + ray.sy) + ray.sz) + ((ray.dx + ray.dy) + ray.dz);
    if (s == s) return kNanRayNone;  // no NaN (and no inf - inf)
    const float inf = __builtin_inff();
    if (__builtin_fabsf(ray.sx) == inf || __builtin_fabsf(ray.sy) == inf || __builtin_fabsf(ray.sz) == inf ||
        __builtin_fabsf(ray.dx) == inf || __builtin_fabsf(ray.dy) == inf || __builtin_fabsf(ray.dz) == inf)
        return kNanRayNone;
    if (nan_winner < 0) return kNanRayTimeMax;  // nothing in the scene accepts anything
    const bool zero_dir = ray.dx == 0.f && ray.dy == 0.f && ray.dz == 0.f;
    return (zero_dir && !winner_is_sphere) ? kNanRayTimeMax : kNanRayTimeNaN;
}

// ---- shading -----------------------------------------------------------------------------------------------
struct Scene {
    const HotPair* __restrict__ pairs;   // traversal stream, ceil(n_objs / 2) records
    uint32_t n_pairs;
    const HotObject* __restrict__ hot;   // per-object rows: materialise(), and the small-scene traversal
    const float4* __restrict__ bounds;   // per object (pinhole grids, <= 64 objects): conservative screen rectangle
                                         // (xmin, xmax, ymin, ymax) in ray-direction units; empty = can never be hit
    const ColdObject* __restrict__ cold;
    const ObjectRecord* __restrict__ objrec;  // per object: what materialise() reads, in one line
    const LightRec* __restrict__ lights;
    uint32_t n_objs;
    uint32_t n_lights;
    uint32_t literal;  // RT_FLAG_LITERAL
    uint32_t affine;   // every mv / mvInverse has bottom row (0,0,0,1) exactly: materialise() skips the two rows it then knows
    uint32_t fast_phong;  // RT_FLAG_FAST_PHONG: colour-only normalisations and the specular power on the fast hardware paths
    int nan_winner;            // index of the LAST sphere / box of the scene (-1: none) and whether it is a sphere:
    uint32_t nan_winner_sphere;  // what the reference's loop ends with for a ray with a NaN in it (nan_ray_outcome)
};

// secondary rays (shadow / reflection; direction.w == 0): pair stream in the wavefront kernels, one object at a
// time in the monolithic kernel (SMALL)
template <bool FUSED, bool SMALL>
__device__ __forceinline__ void closest_secondary(const Scene& S, const Ray& ray, float& T, int& idx) {
    if constexpr (SMALL) closest_hit_small<FUSED, true>(S.hot, S.n_objs, false, 0ull, ray, T, idx);
    else closest_hit<FUSED, true>(S.pairs, S.n_pairs, ray, T, idx);
}
template <bool FUSED, bool SMALL>
__device__ __forceinline__ bool any_secondary(const Scene& S, const Ray& ray) {
    if constexpr (SMALL) return any_hit_small<FUSED>(S.hot, S.n_objs, ray);
    else return any_hit_before_one<FUSED>(S.pairs, S.n_pairs, ray);
}

struct LightGeom {
    float nlx, nly, nlz;  // normalised light vector
    float nDotL, rDotV;
    Ray shadow;
};

// pow(rDotV, fmax(shininess, 1)) (:232). An exponent of exactly 1 (the default material) returns the base:
// that is what libm's powf returns for every finite x >= 0 (its error is orders of magnitude below half an ulp
// there), and it saves the ~100-instruction OCML powf.
// RT_FLAG_FAST_PHONG: exp2(e log2 x) on the hardware's transcendental units (v_log_f32 / v_exp_f32, ~1 ulp each) instead of
// OCML's ~100-instruction powf: relative error ~ e x 2^-23 ln 2 |log2 x| x ... - below 1e-5 of a colour channel for the
// exponents a Phong material carries (measured over every golden vector and the shipped scenes: tests).
__device__ __forceinline__ float specular_power(float rDotV, float shininess, bool fast = false) {
    const float e = __builtin_fmaxf(shininess, 1.f);
    if (e == 1.f) return rDotV;
    if (fast) return rDotV > 0.f ? __builtin_amdgcn_exp2f(e * __builtin_amdgcn_logf(rDotV)) : 0.f;  // (rDotV = max(., 0): pow(0, e) = 0)
    return powf(rDotV, e);
}

// the shadow ray of light L from the point (px, py, pz), and the normalised light vector (shade_and_reflect_kernel.cl:195-205).
// One function for everybody who needs that ray: the light loop, and the trace kernels that REBUILD it from the stored
// hit point instead of reading a stored copy (rt_wavefront.hip) - same statements, same bits.
template <bool FUSED>
__device__ __forceinline__ void shadow_ray_to(const LightRec& L, float px, float py, float pz, Ray& shadow, float& nlx, float& nly, float& nlz) {
    float lx, ly, lz;
    if (L.position.w != 0) { lx = L.position.x - px; ly = L.position.y - py; lz = L.position.z - pz; }
    else { lx = -L.position.x; ly = -L.position.y; lz = -L.position.z; }
    nlx = lx; nly = ly; nlz = lz;
    normalize3(nlx, nly, nlz);
    shadow.sx = fma_<FUSED>(nlx, 0.01f, px);
    shadow.sy = fma_<FUSED>(nly, 0.01f, py);
    shadow.sz = fma_<FUSED>(nlz, 0.01f, pz);
    shadow.sw = fma_<FUSED>(0.0f, 0.01f, 1.0f);
    shadow.dx = lx; shadow.dy = ly; shadow.dz = lz; shadow.dw = 0.0f;
}

// everything of one light-loop iteration that does not depend on the shadow test
// (shade_and_reflect_kernel.cl:194-224)
template <bool FUSED>
__device__ __forceinline__ void light_geometry(const LightRec& L, const HitRec& h, float nvx, float nvy, float nvz,
                                               float vvx, float vvy, float vvz, LightGeom& g, bool fast = false) {
    float nlx, nly, nlz;
    shadow_ray_to<FUSED>(L, h.px, h.py, h.pz, g.shadow, nlx, nly, nlz);
    g.nlx = nlx; g.nly = nly; g.nlz = nlz;
    g.nDotL = dot3(nvx, nvy, nvz, nlx, nly, nlz);
    const float kk = dot3(-nlx, -nly, -nlz, nvx, nvy, nvz) * -2.0f;
    float rx = fma_<FUSED>(kk, nvx, -nlx);
    float ry = fma_<FUSED>(kk, nvy, -nly);
    float rz = fma_<FUSED>(kk, nvz, -nlz);
    normalize3_shading(fast, rx, ry, rz);
    g.rDotV = __builtin_fmaxf(dot3(rx, ry, rz, vvx, vvy, vvz), 0.0f);
}

// lit iff the reference's `shadowcastHit.time >= 1.f || shadowcastHit.time < 0` (:229)
template <bool FUSED, bool COUNT>
__device__ __forceinline__ bool light_visible(const Scene& S, const Ray& shadow, Counters& ctr) {
    if constexpr (COUNT) ctr.traced += 1;
    if (!S.literal) {  // the any-hit test is order-free: a NaN shadow ray's outcome is not (`time >= 1 || time < 0`, :229)
        const int k = nan_ray_outcome(shadow, S.nan_winner, S.nan_winner_sphere);
        if (k != kNanRayNone) return k == kNanRayTimeMax;
    }
    if (S.literal) {
        float T = kMaxFloat;
        int idx = -1;
        closest_secondary<FUSED, true>(S, shadow, T, idx);
        return (T >= 1.f || T < 0);
    }
    return !any_secondary<FUSED, true>(S, shadow);
}

// The light loop in the reference's order. ACCUMULATE: shade_kernel.cl:252 (sum over lights);
// otherwise shade_and_reflect_kernel.cl:238 (assignment - the last light's terms survive). Both carry the
// stale specular: it is re-assigned only when lit with nDotL > 0 and zeroed when shadowed (:229-237).
template <bool FUSED, bool ACCUMULATE, bool COUNT>
__device__ __forceinline__ void shade_forward(const Scene& S, const HitRec& h, float& cr, float& cg, float& cb,
                                              Counters& ctr) {
    const ColdObject* c = S.cold + h.index;
    const float4 amb = c->amb_absorb, dif = c->dif_shine, spec = c->spec_type;
    float nvx = h.nx, nvy = h.ny, nvz = h.nz;
    normalize3_shading(S.fast_phong != 0u, nvx, nvy, nvz);
    float vvx = -h.px, vvy = -h.py, vvz = -h.pz;
    normalize3_shading(S.fast_phong != 0u, vvx, vvy, vvz);
    cr = 0.f; cg = 0.f; cb = 0.f;
    float sr = 0.f, sg = 0.f, sb = 0.f;  // specular carried across iterations
    if constexpr (COUNT) ctr.reference += S.n_lights;
    for (uint32_t li = 0; li < S.n_lights; ++li) {
        const LightRec L = S.lights[li];
        LightGeom g;
        light_geometry<FUSED>(L, h, nvx, nvy, nvz, vvx, vvy, vvz, g, S.fast_phong != 0u);
        const bool lit = light_visible<FUSED, COUNT>(S, g.shadow, ctr);
        const float ar = amb.x * L.ambient.x, ag = amb.y * L.ambient.y, ab = amb.z * L.ambient.z;
        float dr, dg, db;
        if (lit) {
            const float nd = __builtin_fmaxf(g.nDotL, 0.f);
            dr = (dif.x * L.diffuse.x) * nd; dg = (dif.y * L.diffuse.y) * nd; db = (dif.z * L.diffuse.z) * nd;
            if (g.nDotL > 0) {
                const float pw = specular_power(g.rDotV, dif.w, S.fast_phong != 0u);
                sr = (spec.x * L.specular.x) * pw; sg = (spec.y * L.specular.y) * pw; sb = (spec.z * L.specular.z) * pw;
            }
        } else {
            dr = 0.f; dg = 0.f; db = 0.f;
            sr = 0.f; sg = 0.f; sb = 0.f;
        }
        if constexpr (ACCUMULATE) {
            cr = ((cr + ar) + dr) + sr; cg = ((cg + ag) + dg) + sg; cb = ((cb + ab) + db) + sb;
        } else {
            cr = (ar + dr) + sr; cg = (ag + dg) + sg; cb = (ab + db) + sb;
        }
    }
}

// shade_and_reflect's colour without tracing the shadow rays whose result cannot reach it: ambient and
// diffuse come from the last light only; the specular from the last light j that was either shadowed
// (-> 0) or lit with nDotL > 0. Scan backwards from the last light and stop at j.
template <bool FUSED, bool COUNT>
__device__ __forceinline__ void shade_last_light_wins(const Scene& S, const HitRec& h, float& cr, float& cg, float& cb,
                                                      Counters& ctr) {
    cr = 0.f; cg = 0.f; cb = 0.f;
    if constexpr (COUNT) ctr.reference += S.n_lights;
    if (S.n_lights == 0) return;
    const ColdObject* c = S.cold + h.index;
    const float4 amb = c->amb_absorb, dif = c->dif_shine, spec = c->spec_type;
    float nvx = h.nx, nvy = h.ny, nvz = h.nz;
    normalize3_shading(S.fast_phong != 0u, nvx, nvy, nvz);
    float vvx = -h.px, vvy = -h.py, vvz = -h.pz;
    normalize3_shading(S.fast_phong != 0u, vvx, vvy, vvz);
    float sr = 0.f, sg = 0.f, sb = 0.f;
    float dr = 0.f, dg = 0.f, db = 0.f;
    float ar = 0.f, ag = 0.f, ab = 0.f;
    bool need_specular = true;
    for (uint32_t li = S.n_lights; li-- > 0 && need_specular;) {
        const LightRec L = S.lights[li];
        LightGeom g;
        light_geometry<FUSED>(L, h, nvx, nvy, nvz, vvx, vvy, vvz, g, S.fast_phong != 0u);
        const bool lit = light_visible<FUSED, COUNT>(S, g.shadow, ctr);
        if (li == S.n_lights - 1) {
            ar = amb.x * L.ambient.x; ag = amb.y * L.ambient.y; ab = amb.z * L.ambient.z;
            if (lit) {
                const float nd = __builtin_fmaxf(g.nDotL, 0.f);
                dr = (dif.x * L.diffuse.x) * nd; dg = (dif.y * L.diffuse.y) * nd; db = (dif.z * L.diffuse.z) * nd;
            }
        }
        if (!lit) {
            need_specular = false;  // zeroed here, nothing later re-assigns it
        } else if (g.nDotL > 0) {
            const float pw = specular_power(g.rDotV, dif.w, S.fast_phong != 0u);
            sr = (spec.x * L.specular.x) * pw; sg = (spec.y * L.specular.y) * pw; sb = (spec.z * L.specular.z) * pw;
            need_specular = false;
        }
        // lit with nDotL <= 0: the specular of an earlier light is still live - keep scanning
    }
    cr = (ar + dr) + sr; cg = (ag + dg) + sg; cb = (ab + db) + sb;
}

template <bool FUSED, bool COUNT>
__device__ __forceinline__ void shade_assign(const Scene& S, const HitRec& h, float& cr, float& cg, float& cb,
                                             Counters& ctr) {
    if (S.literal) shade_forward<FUSED, false, COUNT>(S, h, cr, cg, cb, ctr);
    else shade_last_light_wins<FUSED, COUNT>(S, h, cr, cg, cb, ctr);
}

// next ray of the reflection chain: start = intersection + 0.001 * normalize(reflection), direction = reflection
// (shade_and_reflect_kernel.cl:260-263, 275-277)
template <bool FUSED>
__device__ __forceinline__ void reflection_ray(const HitRec& h, Ray& r) {
    float nx = h.rx, ny = h.ry, nz = h.rz;
    normalize3(nx, ny, nz);
    r.sx = fma_<FUSED>(nx, 0.001f, h.px);
    r.sy = fma_<FUSED>(ny, 0.001f, h.py);
    r.sz = fma_<FUSED>(nz, 0.001f, h.pz);
    r.sw = fma_<FUSED>(0.0f, 0.001f, h.pw);
    r.dx = h.rx; r.dy = h.ry; r.dz = h.rz; r.dw = 0.0f;
}

// __kernel shade_and_reflect for one work-item whose primary ray hit (shade_and_reflect_kernel.cl:253-284)
template <bool FUSED, bool COUNT>
__device__ __forceinline__ void shade_and_reflect_pixel(const Scene& S, uint32_t max_bounces, const HitRec& hit,
                                                        float& outr, float& outg, float& outb, Counters& ctr) {
    float cr, cg, cb;
    shade_assign<FUSED, COUNT>(S, hit, cr, cg, cb, ctr);
    float ap = S.cold[hit.index].amb_absorb.w;
    float abr = cr * ap, abg = cg * ap, abb = cb * ap;
    float rr = 0.f, rg = 0.f, rb = 0.f;  // reflectColor
    uint32_t bounces = max_bounces;
    HitRec from = hit;  // the hit whose reflection ray the next iteration casts (built only if it is cast)
    // while (bounces-- > 0 && raycast(...) && absorptionPercent <= 0.999f)      (:268)
    for (;;) {
        const uint32_t before = bounces;
        bounces = bounces - 1u;  // the unsigned post-decrement happens whether or not the test passes
        if (!(before > 0u)) break;
        if constexpr (COUNT) ctr.reference += 1;
        const bool absorbing = (ap <= 0.999f);
        if (!absorbing && !S.literal) break;  // the reference still casts this ray but never reads the result
        if constexpr (COUNT) ctr.traced += 1;
        Ray ray;
        reflection_ray<FUSED>(from, ray);
        float T = kMaxFloat;
        int idx = -1;
        closest_secondary<FUSED, true>(S, ray, T, idx);
        if (T == kMaxFloat) break;  // raycast() returned false (:173)
        if (!absorbing) break;
        HitRec rh;
        materialise<FUSED>(S.objrec, S.cold, idx, T, ray, rh, S.affine != 0u);
        shade_assign<FUSED, COUNT>(S, rh, rr, rg, rb, ctr);
        const float ra = (1.f - ap) * S.cold[rh.index].amb_absorb.w;
        abr = fma_<FUSED>(ra, rr, abr); abg = fma_<FUSED>(ra, rg, abg); abb = fma_<FUSED>(ra, rb, abb);
        ap = ap + ra;
        from = rh;
    }
    if (bounces == 0u && ap < 1.f) {  // (:281-282)
        const float w = 1.f - ap;
        abr = fma_<FUSED>(w, rr, abr); abg = fma_<FUSED>(w, rg, abg); abb = fma_<FUSED>(w, rb, abb);
    }
    outr = abr; outg = abg; outb = abb;
}

}  // namespace rt
