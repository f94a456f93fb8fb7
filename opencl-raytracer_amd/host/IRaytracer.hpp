// IRaytracer.hpp - the plugin boundary, as the reference declares it (IRaytracer.hpp:10-21): a backend is
// constructed from const references to the caller-owned scene vectors and exposes one call, Render(),
// returning a backend-owned host array of one float4 per ray.
#pragma once

#include <vector>

#if defined(__has_include)
#if __has_include(<CL/cl.h>)
#ifndef CL_TARGET_OPENCL_VERSION
#define CL_TARGET_OPENCL_VERSION 300
#endif
#include <CL/cl.h>
#define RT_HAVE_CL_TYPES 1
#endif
#endif
#ifndef RT_HAVE_CL_TYPES
struct alignas(16) cl_float4 { float s[4]; };  // layout-identical stand-in when the OpenCL headers are absent
#endif

#include "SceneTypes.hpp"

class IRaytracer {
public:
    virtual cl_float4* Render() = 0;
    virtual ~IRaytracer() {}

protected:
    const std::vector<ObjectData>& objects;
    const std::vector<Light>& lights;
    const std::vector<Ray3D>& rays;

    IRaytracer(const std::vector<ObjectData>& objects_, const std::vector<Light>& lights_, const std::vector<Ray3D>& rays_)
        : objects(objects_), lights(lights_), rays(rays_) {}
};
