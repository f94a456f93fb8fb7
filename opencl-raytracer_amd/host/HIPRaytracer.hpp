// HIPRaytracer.hpp - drop-in replacement for the reference's OpenCLRaytracer (OpenCLRaytracer.hpp:24-85):
//     IRaytracer* raytracer = new HIPRaytracer(objects, lights, rays, MAX_BOUNCES);
//     cl_float4* pixels = raytracer->Render();
// The host side sees only the C ABI (include/hip_raytracer.h); kernels live in libhip_raytracer.so.
#pragma once

#include <stdexcept>
#include <string>
#include <vector>

#include "IRaytracer.hpp"
#include "hip_raytracer.h"

class HIPRaytracer : public IRaytracer {
public:
    // Same argument meaning as OpenCLRaytracer's ctor (OpenCLRaytracer.hpp:61). Errors throw std::runtime_error
    // (the reference lets Boost.Compute exceptions escape). `kernel` selects hittest / shade / shade_and_reflect;
    // the reference host only ever loads shade_and_reflect (OpenCLRaytracer.cpp:53-59).
    HIPRaytracer(const std::vector<ObjectData>& objects, const std::vector<Light>& lights, const std::vector<Ray3D>& rays,
                 unsigned int MAX_BOUNCES, int device = 0, unsigned int flags = 0,
                 int kernel = RT_KERNEL_SHADE_AND_REFLECT);
    // The same raytracer over several GPUs of one node (north_star: row-tiles across the GPUs, gathered on the first):
    // one entry of `devices` per shard (an ordinal may repeat: a rehearsal on fewer GPUs), interleaved row-tiles,
    // Render() returns the whole frame as before. rt_create_multi / rt_render_multi of the C ABI.
    HIPRaytracer(const std::vector<ObjectData>& objects, const std::vector<Light>& lights, const std::vector<Ray3D>& rays,
                 unsigned int MAX_BOUNCES, const std::vector<int>& devices, unsigned int flags = 0,
                 int kernel = RT_KERNEL_SHADE_AND_REFLECT);
    ~HIPRaytracer() override;
    HIPRaytracer(const HIPRaytracer&) = delete;
    HIPRaytracer& operator=(const HIPRaytracer&) = delete;

    // Inherited via IRaytracer: synchronous; the returned buffer is owned by this object and overwritten by
    // the next call (OpenCLRaytracer.cpp:94,104).
    cl_float4* Render() override;

    rt_stats_t Stats();
    rt_context* Context() { return ctx; }

private:
    rt_context* ctx = nullptr;
    rt_multi* multi = nullptr;   // set instead of ctx by the several-GPU constructor
};
