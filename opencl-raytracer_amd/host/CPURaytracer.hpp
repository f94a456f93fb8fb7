// CPURaytracer.hpp - a host-CPU backend behind the same IRaytracer boundary.
//
// The reference once had one: `new CPURaytracer(...)` survives as a comment next to the OpenCL backend's construction
// (OpenCL-Raytracer.cpp:74), and ObjectData::Raycast (ObjectData.cpp:12-133) is what is left of its ray tests. This
// class resurrects it as an in-repo baseline: plain C++ on std::thread, no GPU, no dependency on libhip_raytracer.
//
// Semantics are those of the KERNELS (shade_and_reflect_kernel.cl / shade_kernel.cl / hittest_kernel.cl), not of the
// ObjectData.cpp remnant, which disagrees with them: normals go through `mv` (Q2), equal hit times go to
// the later sphere / the earlier box (Q3), the box slab for a zero direction uses copysign (Q5), shade_and_reflect keeps
// the last light's colour with a possibly stale specular (Q1/Q1b), the bounce loop post-decrements an unsigned (Q8).
// Arithmetic: fp32, multiply-adds contracted exactly where the OpenCL front-end marks them (fmaf), IEEE sqrt / divide -
// the same contract as the HIP backend's default flavour, so the two backends can be compared pixel for pixel.
#pragma once

#include <cstdint>
#include <vector>

#include "IRaytracer.hpp"

class CPURaytracer : public IRaytracer {
public:
    enum Kernel { kHittest = 0, kShade = 1, kShadeAndReflect = 2 };

    // Same argument meaning as OpenCLRaytracer's ctor (OpenCLRaytracer.hpp:61); `threads` = 0 uses every hardware thread.
    CPURaytracer(const std::vector<ObjectData>& objects, const std::vector<Light>& lights, const std::vector<Ray3D>& rays,
                 unsigned int MAX_BOUNCES, Kernel kernel = kShadeAndReflect, unsigned int threads = 0);

    // Inherited via IRaytracer. Synchronous; one float4 per ray, owned by this object, overwritten by the next call.
    // shade / shade_and_reflect: RGB in s[0..2]; pixels whose primary ray misses keep the reference's upload-time
    // value {0,0,0,1} (OpenCLRaytracer.cpp:32). hittest: nearest t in s[0] (MAX_FLOAT on a miss).
    cl_float4* Render() override;

    uint64_t RaysTraced() const { return rays_traced; }   // primary + shadow + reflection rays of the last Render()
    uint64_t HitPixels() const { return hit_pixels; }
    unsigned int Threads() const { return n_threads; }

    struct Instance;  // per object: what the object loop streams for every ray (built once by the ctor)
    struct Surface;   // per object: what a finished ray needs (matrices, material)

private:
    unsigned int max_bounces;
    Kernel kernel;
    unsigned int n_threads;
    std::vector<Instance> instances;
    std::vector<Surface> surfaces;
    std::vector<cl_float4> pixels;
    uint64_t rays_traced = 0, hit_pixels = 0;

public:
    ~CPURaytracer() override;
};
