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
    uint32_t type;   // 0 sphere, 1 box, anything else: never hit (the reference's switch has no default)
    uint32_t pad[3];
};
struct ColdObject {  // 128 B
    float mv[16];        // column-major, as uploaded
    float4 inv_row3;     // mvInverse row 3: (m[3], m[7], m[11], m[15])
    float4 amb_absorb;   // ambient rgb, absorption
    float4 dif_shine;    // diffuse rgb, shininess
    float4 spec_type;    // specular rgb, type bits
};
struct LightRec {    // the reference's 64-byte Light, unchanged
    float4 ambient, diffuse, specular, position;
};
static_assert(sizeof(HotObject) == 64 && sizeof(ColdObject) == 128 && sizeof(LightRec) == 64, "layout");

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

// unit box [-0.5,0.5]^3 (shade_and_reflect_kernel.cl:123-144)
__device__ __forceinline__ bool box_candidate(float sx, float sy, float sz, float dx, float dy, float dz, float& t) {
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

// ray -> object space (x,y,z rows only; the w row is needed only by materialise())
template <bool FUSED, bool DW0>
__device__ __forceinline__ void to_object_space(const float4 r0, const float4 r1, const float4 r2, const Ray& ray,
                                                float& sx, float& sy, float& sz, float& dx, float& dy, float& dz) {
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

// ---- traversal -------------------------------------------------------------------------------------------
// Closest hit over objects [0,n) in ascending index order with the reference's tie rules (Q3): a sphere
// replaces the current hit unless `time < t` (later sphere wins ties), a box unless `time <= t` (earlier
// wins). `k` is wave-uniform, so the HotObject loads are scalar.
template <bool FUSED, bool DW0>
__device__ __forceinline__ void closest_hit(const HotObject* __restrict__ hot, uint32_t n, const Ray& ray, float& T,
                                            int& index) {
    for (uint32_t k = 0; k < n; ++k) {
        const HotObject* o = hot + k;
        const float4 r0 = o->row0, r1 = o->row1, r2 = o->row2;
        const uint32_t type = o->type;
        float sx, sy, sz, dx, dy, dz, t;
        to_object_space<FUSED, DW0>(r0, r1, r2, ray, sx, sy, sz, dx, dy, dz);
        if (type == 0u) {
            if (sphere_candidate<FUSED>(sx, sy, sz, dx, dy, dz, t)) {
                if (!(T < t)) { T = t; index = (int)k; }
            }
        } else if (type == 1u) {
            if (box_candidate(sx, sy, sz, dx, dy, dz, t)) {
                if (!(T <= t)) { T = t; index = (int)k; }
            }
        }
    }
}

// Any accepted candidate with t < 1 (shadow rays; direction = un-normalised light vector, so t in [0,1)
// means an occluder between the point and the light, shade_and_reflect_kernel.cl:201-209,229).
template <bool FUSED>
__device__ __forceinline__ bool any_hit_before_one(const HotObject* __restrict__ hot, uint32_t n, const Ray& ray) {
    bool occluded = false;
    for (uint32_t k = 0; k < n; ++k) {
        const HotObject* o = hot + k;
        const float4 r0 = o->row0, r1 = o->row1, r2 = o->row2;
        const uint32_t type = o->type;
        float sx, sy, sz, dx, dy, dz, t;
        to_object_space<FUSED, true>(r0, r1, r2, ray, sx, sy, sz, dx, dy, dz);
        bool cand = false;
        if (type == 0u) cand = sphere_candidate<FUSED>(sx, sy, sz, dx, dy, dz, t);
        else if (type == 1u) cand = box_candidate(sx, sy, sz, dx, dy, dz, t);
        if (cand && t < 1.f) { occluded = true; break; }
    }
    return occluded;
}

// ---- hit materialisation (once per finished ray) ---------------------------------------------------------
// intersection = mv * p, normal = normalize((mv * (n,0)).xyz), reflection = reflect(dir, normal)
// (shade_and_reflect_kernel.cl:112-118 sphere, :149-165 box, :175)
template <bool FUSED>
__device__ __forceinline__ void materialise(const HotObject* __restrict__ hot, const ColdObject* __restrict__ cold,
                                            int index, float t, const Ray& ray, HitRec& h) {
    const HotObject* o = hot + index;
    const ColdObject* c = cold + index;
    const float4 r0 = o->row0, r1 = o->row1, r2 = o->row2, r3 = c->inv_row3;
    const float sx = row4<FUSED>(r0.x, r0.y, r0.z, r0.w, ray.sx, ray.sy, ray.sz, ray.sw);
    const float sy = row4<FUSED>(r1.x, r1.y, r1.z, r1.w, ray.sx, ray.sy, ray.sz, ray.sw);
    const float sz = row4<FUSED>(r2.x, r2.y, r2.z, r2.w, ray.sx, ray.sy, ray.sz, ray.sw);
    const float sw = row4<FUSED>(r3.x, r3.y, r3.z, r3.w, ray.sx, ray.sy, ray.sz, ray.sw);
    const float dx = row4<FUSED>(r0.x, r0.y, r0.z, r0.w, ray.dx, ray.dy, ray.dz, ray.dw);
    const float dy = row4<FUSED>(r1.x, r1.y, r1.z, r1.w, ray.dx, ray.dy, ray.dz, ray.dw);
    const float dz = row4<FUSED>(r2.x, r2.y, r2.z, r2.w, ray.dx, ray.dy, ray.dz, ray.dw);
    const float dw = row4<FUSED>(r3.x, r3.y, r3.z, r3.w, ray.dx, ray.dy, ray.dz, ray.dw);
    const float px = fma_<FUSED>(t, dx, sx);
    const float py = fma_<FUSED>(t, dy, sy);
    const float pz = fma_<FUSED>(t, dz, sz);
    const float pw = fma_<FUSED>(t, dw, sw);
    float ox, oy, oz;  // object-space normal
    if (o->type == 0u) {
        ox = px; oy = py; oz = pz;
    } else {
        ox = 0.f; oy = 0.f; oz = 0.f;
        if (px > 0.4998f) ox += 1.f; else if (px < -0.4998f) ox -= 1.f;
        if (py > 0.4998f) oy += 1.f; else if (py < -0.4998f) oy -= 1.f;
        if (pz > 0.4998f) oz += 1.f; else if (pz < -0.4998f) oz -= 1.f;
    }
    const float* m = c->mv;
    h.px = row4<FUSED>(m[0], m[4], m[8], m[12], px, py, pz, pw);
    h.py = row4<FUSED>(m[1], m[5], m[9], m[13], px, py, pz, pw);
    h.pz = row4<FUSED>(m[2], m[6], m[10], m[14], px, py, pz, pw);
    h.pw = row4<FUSED>(m[3], m[7], m[11], m[15], px, py, pz, pw);
    float nx = row4<FUSED>(m[0], m[4], m[8], m[12], ox, oy, oz, 0.f);
    float ny = row4<FUSED>(m[1], m[5], m[9], m[13], ox, oy, oz, 0.f);
    float nz = row4<FUSED>(m[2], m[6], m[10], m[14], ox, oy, oz, 0.f);
    normalize3(nx, ny, nz);
    h.nx = nx; h.ny = ny; h.nz = nz;
    const float k2 = dot3(ray.dx, ray.dy, ray.dz, nx, ny, nz) * -2.0f;
    h.rx = fma_<FUSED>(k2, nx, ray.dx);
    h.ry = fma_<FUSED>(k2, ny, ray.dy);
    h.rz = fma_<FUSED>(k2, nz, ray.dz);
    h.index = index;
}

// ---- shading -----------------------------------------------------------------------------------------------
struct Scene {
    const HotObject* __restrict__ hot;
    const ColdObject* __restrict__ cold;
    const LightRec* __restrict__ lights;
    uint32_t n_objs;
    uint32_t n_lights;
    uint32_t literal;  // RT_FLAG_LITERAL
};

struct LightGeom {
    float nlx, nly, nlz;  // normalised light vector
    float nDotL, rDotV;
    Ray shadow;
};

// everything of one light-loop iteration that does not depend on the shadow test
// (shade_and_reflect_kernel.cl:194-224)
template <bool FUSED>
__device__ __forceinline__ void light_geometry(const LightRec& L, const HitRec& h, float nvx, float nvy, float nvz,
                                               float vvx, float vvy, float vvz, LightGeom& g) {
    float lx, ly, lz;
    if (L.position.w != 0) { lx = L.position.x - h.px; ly = L.position.y - h.py; lz = L.position.z - h.pz; }
    else { lx = -L.position.x; ly = -L.position.y; lz = -L.position.z; }
    float nlx = lx, nly = ly, nlz = lz;
    normalize3(nlx, nly, nlz);
    g.shadow.sx = fma_<FUSED>(nlx, 0.01f, h.px);
    g.shadow.sy = fma_<FUSED>(nly, 0.01f, h.py);
    g.shadow.sz = fma_<FUSED>(nlz, 0.01f, h.pz);
    g.shadow.sw = fma_<FUSED>(0.0f, 0.01f, 1.0f);
    g.shadow.dx = lx; g.shadow.dy = ly; g.shadow.dz = lz; g.shadow.dw = 0.0f;
    g.nlx = nlx; g.nly = nly; g.nlz = nlz;
    g.nDotL = dot3(nvx, nvy, nvz, nlx, nly, nlz);
    const float kk = dot3(-nlx, -nly, -nlz, nvx, nvy, nvz) * -2.0f;
    float rx = fma_<FUSED>(kk, nvx, -nlx);
    float ry = fma_<FUSED>(kk, nvy, -nly);
    float rz = fma_<FUSED>(kk, nvz, -nlz);
    normalize3(rx, ry, rz);
    g.rDotV = __builtin_fmaxf(dot3(rx, ry, rz, vvx, vvy, vvz), 0.0f);
}

// lit iff the reference's `shadowcastHit.time >= 1.f || shadowcastHit.time < 0` (:229)
template <bool FUSED, bool COUNT>
__device__ __forceinline__ bool light_visible(const Scene& S, const Ray& shadow, Counters& ctr) {
    if constexpr (COUNT) ctr.traced += 1;
    if (S.literal) {
        float T = kMaxFloat;
        int idx = -1;
        closest_hit<FUSED, true>(S.hot, S.n_objs, shadow, T, idx);
        return (T >= 1.f || T < 0);
    }
    return !any_hit_before_one<FUSED>(S.hot, S.n_objs, shadow);
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
    normalize3(nvx, nvy, nvz);
    float vvx = -h.px, vvy = -h.py, vvz = -h.pz;
    normalize3(vvx, vvy, vvz);
    cr = 0.f; cg = 0.f; cb = 0.f;
    float sr = 0.f, sg = 0.f, sb = 0.f;  // specular carried across iterations
    if constexpr (COUNT) ctr.reference += S.n_lights;
    for (uint32_t li = 0; li < S.n_lights; ++li) {
        const LightRec L = S.lights[li];
        LightGeom g;
        light_geometry<FUSED>(L, h, nvx, nvy, nvz, vvx, vvy, vvz, g);
        const bool lit = light_visible<FUSED, COUNT>(S, g.shadow, ctr);
        const float ar = amb.x * L.ambient.x, ag = amb.y * L.ambient.y, ab = amb.z * L.ambient.z;
        float dr, dg, db;
        if (lit) {
            const float nd = __builtin_fmaxf(g.nDotL, 0.f);
            dr = (dif.x * L.diffuse.x) * nd; dg = (dif.y * L.diffuse.y) * nd; db = (dif.z * L.diffuse.z) * nd;
            if (g.nDotL > 0) {
                const float pw = powf(g.rDotV, __builtin_fmaxf(dif.w, 1.f));
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
    normalize3(nvx, nvy, nvz);
    float vvx = -h.px, vvy = -h.py, vvz = -h.pz;
    normalize3(vvx, vvy, vvz);
    float sr = 0.f, sg = 0.f, sb = 0.f;
    float dr = 0.f, dg = 0.f, db = 0.f;
    float ar = 0.f, ag = 0.f, ab = 0.f;
    bool need_specular = true;
    for (uint32_t li = S.n_lights; li-- > 0 && need_specular;) {
        const LightRec L = S.lights[li];
        LightGeom g;
        light_geometry<FUSED>(L, h, nvx, nvy, nvz, vvx, vvy, vvz, g);
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
            const float pw = powf(g.rDotV, __builtin_fmaxf(dif.w, 1.f));
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
    Ray ray;
    reflection_ray<FUSED>(hit, ray);
    // while (bounces-- > 0 && raycast(...) && absorptionPercent <= 0.999f)      (:268)
    for (;;) {
        const uint32_t before = bounces;
        bounces = bounces - 1u;  // the unsigned post-decrement happens whether or not the test passes
        if (!(before > 0u)) break;
        if constexpr (COUNT) ctr.reference += 1;
        const bool absorbing = (ap <= 0.999f);
        if (!absorbing && !S.literal) break;  // the reference still casts this ray but never reads the result
        if constexpr (COUNT) ctr.traced += 1;
        float T = kMaxFloat;
        int idx = -1;
        closest_hit<FUSED, true>(S.hot, S.n_objs, ray, T, idx);
        if (T == kMaxFloat) break;  // raycast() returned false (:173)
        if (!absorbing) break;
        HitRec rh;
        materialise<FUSED>(S.hot, S.cold, idx, T, ray, rh);
        shade_assign<FUSED, COUNT>(S, rh, rr, rg, rb, ctr);
        const float ra = (1.f - ap) * S.cold[rh.index].amb_absorb.w;
        abr = fma_<FUSED>(ra, rr, abr); abg = fma_<FUSED>(ra, rg, abg); abb = fma_<FUSED>(ra, rb, abb);
        ap = ap + ra;
        reflection_ray<FUSED>(rh, ray);
    }
    if (bounces == 0u && ap < 1.f) {  // (:281-282)
        const float w = 1.f - ap;
        abr = fma_<FUSED>(w, rr, abr); abg = fma_<FUSED>(w, rg, abg); abb = fma_<FUSED>(w, rb, abb);
    }
    outr = abr; outg = abg; outb = abb;
}

}  // namespace rt
