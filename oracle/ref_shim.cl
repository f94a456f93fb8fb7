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

float __attribute__((overloadable)) dot(float3 a, float3 b) {
    float s = a.x * b.x;
    s = s + a.y * b.y;
    s = s + a.z * b.z;
    return s;
}

float3 __attribute__((overloadable)) normalize(float3 v) {
    float s = v.x * v.x;
    s = s + v.y * v.y;
    s = s + v.z * v.z;
    float len = __builtin_sqrtf(s);
    float3 r;
    r.x = v.x / len;
    r.y = v.y / len;
    r.z = v.z / len;
    return r;
}
