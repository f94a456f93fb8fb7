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
#ifndef SHIM_VARIANT
#define SHIM_VARIANT 0
#endif

float __attribute__((overloadable)) dot(float3 a, float3 b) {
#if SHIM_VARIANT == 1 || SHIM_VARIANT == 3
    return __builtin_fmaf(a.z, b.z, __builtin_fmaf(a.y, b.y, a.x * b.x));
#else
    float s = a.x * b.x;
    s = s + a.y * b.y;
    s = s + a.z * b.z;
    return s;
#endif
}

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
