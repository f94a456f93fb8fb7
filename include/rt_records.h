/*
 * rt_records.h - the byte layouts that cross the C ABI (hip_raytracer.h).
 *
 * These are the reference's OpenCL device structs (shade_and_reflect_kernel.cl:1-29) and their host
 * mirrors cl_Ray / cl_Material / cl_ObjectData / cl_Light (OpenCLRaytracer.hpp:25-58). OpenCL float3 is
 * 16 bytes, float16 is 64 bytes / 64-aligned, so:
 *
 *   rt_material    64 B   +0 ambient(3)+pad  +16 diffuse(3)+pad  +32 specular(3)+pad
 *                         +48 absorption  +52 reflection  +56 transparency  +60 shininess
 *   rt_object_data 320 B  +0 material  +64 mv  +128 mvInverse  +192 mvInverseTranspose  +256 type  +260 pad[60]
 *                         matrices column-major (glm::value_ptr order, OpenCLRaytracer.cpp:114-117)
 *   rt_light       64 B   +0 ambient  +16 diffuse  +32 specular  +48 position (w=1 positional, w=0 directional)
 *   rt_ray         32 B   +0 start (w=1)  +16 direction (w=0, not normalised)
 *   rt_pixel       16 B   float3 colour in a float4 slot (cl_float3 == cl_float4)
 *
 * Alignment is the CALLER's business: the OpenCL types are 16- / 64-byte aligned (cl_float16 members), these plain C structs
 * ask for 4. The ABI only ever copies the arrays byte for byte (rt_create: "host buffers are copied"), so any alignment works;
 * a caller that casts a reference-side cl_ObjectData array to `const void*` passes exactly these bytes.
 */
#ifndef RT_RECORDS_H
#define RT_RECORDS_H

#include <stdint.h>

#ifdef __cplusplus
#define RT_STATIC_ASSERT(c, m) static_assert(c, m)
#else
#define RT_STATIC_ASSERT(c, m) _Static_assert(c, m)
#endif

typedef struct rt_material {
    float ambient[4];
    float diffuse[4];
    float specular[4];
    float absorption, reflection, transparency, shininess;
} rt_material;

typedef struct rt_object_data {
    rt_material mat;
    float mv[16];
    float mvInverse[16];
    float mvInverseTranspose[16]; /* uploaded by the reference, read by no kernel */
    uint32_t type;                /* 0 sphere, 1 box (ObjectData.hpp:9-12); 2 triangle - an EXTENSION of this backend with
                                   * no reference semantics: mv columns 0..2 = vertices in view space
                                   * (w = 1), mvInverse column 0 = guard sphere (cx, cy, cz, R); large-scene (grid) path only */
    uint8_t spacer[60];
} rt_object_data;

typedef struct rt_light {
    float ambient[4];
    float diffuse[4];
    float specular[4];
    float position[4];
} rt_light;

typedef struct rt_ray {
    float start[4];
    float direction[4];
} rt_ray;

typedef struct rt_pixel {
    float s[4];
} rt_pixel;

RT_STATIC_ASSERT(sizeof(rt_material) == 64, "Material must be 64 bytes");
RT_STATIC_ASSERT(sizeof(rt_object_data) == 320, "ObjectData must be 320 bytes");
RT_STATIC_ASSERT(sizeof(rt_light) == 64, "Light must be 64 bytes");
RT_STATIC_ASSERT(sizeof(rt_ray) == 32, "Ray must be 32 bytes");
RT_STATIC_ASSERT(sizeof(rt_pixel) == 16, "pixel must be 16 bytes");

#define RT_TYPE_SPHERE 0u
#define RT_TYPE_BOX 1u
#define RT_MAX_FLOAT 3.402823466e+38F /* shade_and_reflect_kernel.cl:31 */

#endif /* RT_RECORDS_H */
