// cpu_backend_c.cpp - C entry point over CPURaytracer, for callers that hold the scene as the reference's DEVICE-layout
// records (include/rt_records.h): the Python tests and bench.py's `cpu_baseline` leg (ctypes). It rebuilds the host
// records (the inverse of HIPRaytracer.cpp's converters), constructs the backend through the IRaytracer boundary and
// copies the frame out. No GPU, no libhip_raytracer.
#include <chrono>
#include <cstring>
#include <memory>
#include <vector>

#include "CPURaytracer.hpp"
#include "rt_records.h"

extern "C" {

// out: n_rays x 4 floats (kernels 1, 2) or n_rays floats (kernel 0). Returns 0, or -1 for arguments it cannot serve
// (unknown kernel; type-2 triangle records, which are the HIP backend's own extension).
int cpu_rt_render(int kernel, uint32_t max_bounces, const void* objs_, uint32_t n_objs, const void* lights_, uint32_t n_lights,
                  const void* rays_, uint64_t n_rays, float* out, unsigned int threads, uint64_t* rays_traced, uint64_t* hit_pixels,
                  double* seconds, unsigned int* threads_used) {
    if (kernel < 0 || kernel > 2 || (!out && n_rays)) return -1;
    const rt_object_data* objs = static_cast<const rt_object_data*>(objs_);
    const rt_light* lights = static_cast<const rt_light*>(lights_);
    const rt_ray* rays = static_cast<const rt_ray*>(rays_);
    std::vector<ObjectData> objects;
    objects.reserve(n_objs);
    for (uint32_t i = 0; i < n_objs; ++i) {
        const rt_object_data& d = objs[i];
        if (d.type == 2u) return -1;
        Material m;
        m.ambient = rtm::vec3(d.mat.ambient[0], d.mat.ambient[1], d.mat.ambient[2]);
        m.diffuse = rtm::vec3(d.mat.diffuse[0], d.mat.diffuse[1], d.mat.diffuse[2]);
        m.specular = rtm::vec3(d.mat.specular[0], d.mat.specular[1], d.mat.specular[2]);
        m.absorption = d.mat.absorption; m.reflection = d.mat.reflection; m.transparency = d.mat.transparency; m.shininess = d.mat.shininess;
        rtm::mat4 mv(1.f);
        std::memcpy(mv.data(), d.mv, sizeof(d.mv));
        ObjectData o(static_cast<ObjectData::PrimativeType>(d.type > 255u ? 255u : d.type), m, rtm::mat4(1.f));
        o.mv = mv;  // the uploaded matrices, byte for byte (the ctor's own inverse is not what the caller uploaded)
        std::memcpy(o.mvInverse.data(), d.mvInverse, sizeof(d.mvInverse));
        std::memcpy(o.mvInverseTranspose.data(), d.mvInverseTranspose, sizeof(d.mvInverseTranspose));
        objects.push_back(o);
    }
    std::vector<Light> ls;
    ls.reserve(n_lights);
    for (uint32_t i = 0; i < n_lights; ++i) {
        const rt_light& d = lights[i];
        LightProperties p;
        p.ambient = rtm::vec3(d.ambient[0], d.ambient[1], d.ambient[2]);
        p.diffuse = rtm::vec3(d.diffuse[0], d.diffuse[1], d.diffuse[2]);
        p.specular = rtm::vec3(d.specular[0], d.specular[1], d.specular[2]);
        Light l(p, rtm::mat4(1.f));
        l.lightPosition = rtm::vec4(d.position[0], d.position[1], d.position[2], d.position[3]);
        ls.push_back(l);
    }
    std::vector<Ray3D> rs;
    rs.reserve(n_rays);
    for (uint64_t i = 0; i < n_rays; ++i) {
        Ray3D r(rtm::vec3(0.f, 0.f, 0.f), rtm::vec3(0.f, 0.f, 0.f));
        r.start = rtm::vec4(rays[i].start[0], rays[i].start[1], rays[i].start[2], rays[i].start[3]);
        r.direction = rtm::vec4(rays[i].direction[0], rays[i].direction[1], rays[i].direction[2], rays[i].direction[3]);
        rs.push_back(r);
    }
    std::unique_ptr<CPURaytracer> backend(new CPURaytracer(objects, ls, rs, max_bounces, static_cast<CPURaytracer::Kernel>(kernel), threads));
    IRaytracer* raytracer = backend.get();  // everything below goes through the reference's interface
    const auto t0 = std::chrono::steady_clock::now();
    const cl_float4* px = raytracer->Render();
    const auto t1 = std::chrono::steady_clock::now();
    for (uint64_t i = 0; i < n_rays; ++i) {
        if (kernel == 0) out[i] = px[i].s[0];
        else std::memcpy(out + 4 * i, px[i].s, 16);
    }
    if (rays_traced) *rays_traced = backend->RaysTraced();
    if (hit_pixels) *hit_pixels = backend->HitPixels();
    if (seconds) *seconds = std::chrono::duration<double>(t1 - t0).count();
    if (threads_used) *threads_used = backend->Threads();
    return 0;
}

}  // extern "C"
