// TEST INFRASTRUCTURE - not product code.
//
// OpenCL-C builtin shim for the host-compiled *verbatim* reference kernels
// (oracle/_ref, built by oracle/Makefile from /root/reference/*.cl, never copied).
// The reference kernels leave exactly eight undefined builtins:
//   get_global_id, sqrt, pow, fmin, fmax, copysign, dot(float3,float3), normalize(float3)
// get_global_id lives in ref_driver.c (thread-local); the other seven are here.
//
// These definitions ARE the oracle's definition of the OpenCL builtins (the reference
// pins no OpenCL implementation): IEEE sqrt and divide, libm powf, and
//   dot(a,b)     = (a.x*b.x + a.y*b.y) + a.z*b.z      no contraction, left to right
//   normalize(v) = v / sqrt(dot(v,v))                 three IEEE divides
// This file is ALWAYS compiled with -ffp-contract=off (both oracle flavours), so the
// builtins do not change with the kernel's contraction mode. The HIP product and the
// C restatement (oracle/rt_oracle.c) implement the same definitions.
//
// Compiled with -cl-no-stdinc so that these overloads do not clash with opencl-c.h.

typedef float float3 __attribute__((ext_vector_type(3)));

float __attribute__((overloadable)) sqrt(float x) { return __builtin_sqrtf(x); }
float __attribute__((overloadable)) pow(float x, float y) { return __builtin_powf(x, y); }
float __attribute__((overloadable)) fmin(float a, float b) { return __builtin_fminf(a, b); }
float __attribute__((overloadable)) fmax(float a, float b) { return __builtin_fmaxf(a, b); }
float __attribute__((overloadable)) copysign(float a, float b) { return __builtin_copysignf(a, b); }

// SHIM_VARIANT (default 0 = the definitions above, what the golden vectors and the restatement use) selects OTHER
// conforming definitions of the two geometric builtins, to measure how far the reference's output moves when its
// OpenCL runtime implements them differently (tools/shim_bounds.py):
//   1: dot() as an fma chain  fma(a.z, b.z, fma(a.y, b.y, a.x * b.x))      (what a GPU compiler emits)
//   2: normalize() as v * (1 / sqrt(dot(v, v)))                            (reciprocal-multiply: one divide, three products)
//   3: both
//   4: what AMD's OWN OpenCL builtin library computes on the device family this repo targets, restated from the LLVM IR of
//      /opt/rocm/amdgcn/bitcode/opencl.bc (part of this image; `llvm-dis opencl.bc`, functions @_Z3dotDv3_fS_ and
//      @_Z9normalizeDv3_f, and ocml.bc @__ocml_rsqrt_f32):
//        dot(a, b)    = fmuladd(a.z, b.z, fmuladd(a.y, b.y, a.x * b.x))                       (= variant 1's chain)
//        normalize(v) = v                          if v.x == v.y == v.z == 0   (a ZERO vector comes back unchanged, not NaN)
//                     = (v s) rsqrt(dot(v s, v s)) with s = 2^86 if dot(v, v) < 2^-126, s = 2^-66 if it is +inf (and the
//                       infinite components replaced by +-1, the others by +-0, if it still is), else s = 1
//        rsqrt(x)     = v_rsq_f32 (1 ulp; inputs below 2^-126 pre-scaled by 2^24) - a hardware instruction no CPU has, so it
//                       is MODELLED: the correctly rounded 1 / sqrt(x) here,
//   5: ... the same one ulp UP, 6: one ulp DOWN - the three runs bracket every value the instruction may return.
#ifndef SHIM_VARIANT
#define SHIM_VARIANT 0
#endif

float __attribute__((overloadable)) dot(float3 a, float3 b) {
#if SHIM_VARIANT == 1 || SHIM_VARIANT >= 3
    return __builtin_fmaf(a.z, b.z, __builtin_fmaf(a.y, b.y, a.x * b.x));
#else
    float s = a.x * b.x;
    s = s + a.y * b.y;
    s = s + a.z * b.z;
    return s;
#endif
}

#if SHIM_VARIANT >= 4
// model of v_rsq_f32: correctly rounded (4), one ulp above (5) / below (6) that
static float rsqrt_model(float x) {
    float r = (float)(1.0 / __builtin_sqrt((double)x));
    if (r != r || r == __builtin_inff() || r == 0.0f) return r;  // rsqrt(NaN / 0 / inf): nothing to nudge
    int bits = __builtin_astype(r, int);
#if SHIM_VARIANT == 5
    bits += 1;
#elif SHIM_VARIANT == 6
    bits -= 1;
#endif
    return __builtin_astype(bits, float);
}
static float3 scale3(float3 v, float s) { float3 r; r.x = v.x * s; r.y = v.y * s; r.z = v.z * s; return r; }
float3 __attribute__((overloadable)) normalize(float3 v) {
    if (v.x == 0.0f && v.y == 0.0f && v.z == 0.0f) return v;            // %2-%5: all(v == 0) -> the argument itself
    float l2 = dot(v, v);
    if (l2 < 0x1.0p-126f) {                                              // %8-%11
        v = scale3(v, 0x1.0p+86f);
        l2 = dot(v, v);
    } else if (l2 == __builtin_inff()) {                                 // %13-%22
        v = scale3(v, 0x1.0p-66f);
        l2 = dot(v, v);
        if (l2 == __builtin_inff()) {
            float3 u;
            u.x = __builtin_copysignf(__builtin_fabsf(v.x) == __builtin_inff() ? 1.0f : 0.0f, v.x);
            u.y = __builtin_copysignf(__builtin_fabsf(v.y) == __builtin_inff() ? 1.0f : 0.0f, v.y);
            u.z = __builtin_copysignf(__builtin_fabsf(v.z) == __builtin_inff() ? 1.0f : 0.0f, v.z);
            v = u;
            l2 = dot(v, v);
        }
    }
    return scale3(v, rsqrt_model(l2));                                   // %26-%29
}
#else
float3 __attribute__((overloadable)) normalize(float3 v) {
#if SHIM_VARIANT == 1 || SHIM_VARIANT == 3
    float s = __builtin_fmaf(v.z, v.z, __builtin_fmaf(v.y, v.y, v.x * v.x));
#else
    float s = v.x * v.x;
    s = s + v.y * v.y;
    s = s + v.z * v.z;
#endif
    float len = __builtin_sqrtf(s);
    float3 r;
#if SHIM_VARIANT == 2 || SHIM_VARIANT == 3
    float inv = 1.0f / len;
    r.x = v.x * inv;
    r.y = v.y * inv;
    r.z = v.z * inv;
#else
    r.x = v.x / len;
    r.y = v.y / len;
    r.z = v.z / len;
#endif
    return r;
}
#endif
