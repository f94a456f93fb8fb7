#include "HIPRaytracer.hpp"

#include <cstring>

#include "rt_records.h"

namespace {

// host records -> the device layouts of rt_records.h (what the reference's cl_* converters do,
// OpenCLRaytracer.cpp:108-146)
void put3(float* dst, const rtm::vec3& v) { dst[0] = v.x; dst[1] = v.y; dst[2] = v.z; dst[3] = 0.f; }
void put4(float* dst, const rtm::vec4& v) { dst[0] = v.x; dst[1] = v.y; dst[2] = v.z; dst[3] = v.w; }

rt_object_data to_device(const ObjectData& o) {
    rt_object_data d;
    std::memset(&d, 0, sizeof(d));
    put3(d.mat.ambient, o.mat.ambient);
    put3(d.mat.diffuse, o.mat.diffuse);
    put3(d.mat.specular, o.mat.specular);
    d.mat.absorption = o.mat.absorption;
    d.mat.reflection = o.mat.reflection;
    d.mat.transparency = o.mat.transparency;
    d.mat.shininess = o.mat.shininess;
    std::memcpy(d.mv, o.mv.data(), sizeof(d.mv));
    std::memcpy(d.mvInverse, o.mvInverse.data(), sizeof(d.mvInverse));
    std::memcpy(d.mvInverseTranspose, o.mvInverseTranspose.data(), sizeof(d.mvInverseTranspose));
    d.type = static_cast<uint32_t>(o.type);
    return d;
}

rt_light to_device(const Light& l) {
    rt_light d;
    put3(d.ambient, l.ambient);
    put3(d.diffuse, l.diffuse);
    put3(d.specular, l.specular);
    put4(d.position, l.lightPosition);
    return d;
}

}  // namespace

HIPRaytracer::HIPRaytracer(const std::vector<ObjectData>& objects_, const std::vector<Light>& lights_,
                           const std::vector<Ray3D>& rays_, unsigned int MAX_BOUNCES, int device, unsigned int flags,
                           int kernel)
    : IRaytracer(objects_, lights_, rays_) {
    std::vector<rt_object_data> objs;
    objs.reserve(objects.size());
    for (const ObjectData& o : objects) objs.push_back(to_device(o));
    std::vector<rt_light> ls;
    ls.reserve(lights.size());
    for (const Light& l : lights) ls.push_back(to_device(l));
    static_assert(sizeof(Ray3D) == sizeof(rt_ray), "Ray3D is already in device layout");
    const int rc = rt_create(&ctx, objs.data(), static_cast<uint32_t>(objs.size()), ls.data(),
                             static_cast<uint32_t>(ls.size()), rays.data(), rays.size(), MAX_BOUNCES, kernel, device, flags);
    if (rc != RT_OK) throw std::runtime_error(std::string("HIPRaytracer: ") + rt_last_error(nullptr));
}

HIPRaytracer::HIPRaytracer(const std::vector<ObjectData>& objects_, const std::vector<Light>& lights_,
                           const std::vector<Ray3D>& rays_, unsigned int MAX_BOUNCES, const std::vector<int>& devices,
                           unsigned int flags, int kernel)
    : IRaytracer(objects_, lights_, rays_) {
    std::vector<rt_object_data> objs;
    objs.reserve(objects.size());
    for (const ObjectData& o : objects) objs.push_back(to_device(o));
    std::vector<rt_light> ls;
    ls.reserve(lights.size());
    for (const Light& l : lights) ls.push_back(to_device(l));
    // tile_rays = 0: row-tiles of 16 rows when the rays are the pinhole grid, else 65 536 rays
    const int rc = rt_create_multi(&multi, objs.data(), static_cast<uint32_t>(objs.size()), ls.data(), static_cast<uint32_t>(ls.size()),
                                   rays.data(), rays.size(), MAX_BOUNCES, kernel, devices.data(), static_cast<uint32_t>(devices.size()), 0, flags);
    if (rc != RT_OK) throw std::runtime_error(std::string("HIPRaytracer: ") + rt_multi_last_error(nullptr));
}

HIPRaytracer::~HIPRaytracer() {
    rt_destroy(ctx);
    rt_destroy_multi(multi);
}

cl_float4* HIPRaytracer::Render() {
    const float* out = nullptr;
    if (multi) {
        if (rt_render_multi(multi, &out) != RT_OK) throw std::runtime_error(std::string("HIPRaytracer::Render: ") + rt_multi_last_error(multi));
        return reinterpret_cast<cl_float4*>(const_cast<float*>(out));
    }
    if (rt_render(ctx, &out) != RT_OK) throw std::runtime_error(std::string("HIPRaytracer::Render: ") + rt_last_error(ctx));
    return reinterpret_cast<cl_float4*>(const_cast<float*>(out));
}

rt_stats_t HIPRaytracer::Stats() {
    rt_stats_t s;
    if (multi) {  // several GPUs: the counters summed over the shards, the slowest shard's kernel time - the whole frame's figures
        if (rt_get_stats_multi(multi, &s) != RT_OK) throw std::runtime_error(std::string("HIPRaytracer::Stats: ") + rt_multi_last_error(multi));
        return s;
    }
    if (rt_get_stats(ctx, &s) != RT_OK) throw std::runtime_error(std::string("HIPRaytracer::Stats: ") + rt_last_error(ctx));
    return s;
}
