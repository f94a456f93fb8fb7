// scene_tool.cpp - the reference's main() without the window (OpenCL-Raytracer.cpp:28-104):
//   scene_tool records <scene.txt> <out.bin>          parse the scene, dump the device records (no GPU needed)
//   scene_tool render  <scene.txt> <W> <H> <D> <out.ppm> [z-bits|-] [hip|cpu]
//                       parse, build rays, render, write a PPM. Backend: `hip` (default) = HIPRaytracer on the GPU;
//                       `cpu` = CPURaytracer, no GPU (the reference's main() has both lines, OpenCL-Raytracer.cpp:74-75)
#include <cmath>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <memory>

#include "CPURaytracer.hpp"
#include "HIPRaytracer.hpp"
#include "PPMExporter.hpp"
#include "SceneLoader.hpp"
#include "rt_records.h"

static void dump_records(const std::vector<ObjectData>& objects, const std::vector<Light>& lights, const char* path) {
    std::FILE* f = std::fopen(path, "wb");
    if (!f) throw std::runtime_error("cannot write records");
    const uint32_t n[2] = {(uint32_t)objects.size(), (uint32_t)lights.size()};
    std::fwrite(n, 4, 2, f);
    for (const ObjectData& o : objects) {
        rt_object_data d;
        std::memset(&d, 0, sizeof(d));
        const float amb[4] = {o.mat.ambient.x, o.mat.ambient.y, o.mat.ambient.z, 0}, dif[4] = {o.mat.diffuse.x, o.mat.diffuse.y, o.mat.diffuse.z, 0},
                    spe[4] = {o.mat.specular.x, o.mat.specular.y, o.mat.specular.z, 0};
        std::memcpy(d.mat.ambient, amb, 16); std::memcpy(d.mat.diffuse, dif, 16); std::memcpy(d.mat.specular, spe, 16);
        d.mat.absorption = o.mat.absorption; d.mat.reflection = o.mat.reflection; d.mat.transparency = o.mat.transparency; d.mat.shininess = o.mat.shininess;
        std::memcpy(d.mv, o.mv.data(), 64); std::memcpy(d.mvInverse, o.mvInverse.data(), 64); std::memcpy(d.mvInverseTranspose, o.mvInverseTranspose.data(), 64);
        d.type = (uint32_t)o.type;
        std::fwrite(&d, sizeof(d), 1, f);
    }
    for (const Light& l : lights) {
        const float rec[16] = {l.ambient.x, l.ambient.y, l.ambient.z, 0, l.diffuse.x, l.diffuse.y, l.diffuse.z, 0,
                               l.specular.x, l.specular.y, l.specular.z, 0, l.lightPosition.x, l.lightPosition.y, l.lightPosition.z, l.lightPosition.w};
        std::fwrite(rec, 64, 1, f);
    }
    std::fclose(f);
}

int main(int argc, char** argv) {
    try {
        if (argc < 4) { std::fprintf(stderr, "usage: scene_tool records|render ...\n"); return 1; }
        std::vector<ObjectData> objects;
        std::vector<Light> lights;
        SceneLoader loader;
        loader.Load(argv[2], objects, lights);
        std::printf("Scene file loaded without any errors.\n");
        if (std::strcmp(argv[1], "records") == 0) { dump_records(objects, lights, argv[3]); return 0; }
        if (argc < 7) return 1;
        const int width = std::atoi(argv[3]), height = std::atoi(argv[4]);
        const unsigned depth = (unsigned)std::atoi(argv[5]);
        float fov = rtm::radians(60.f);
        fov *= 0.5f;
        float z = -((height / 2.0f) / tanf(fov));
        if (argc > 7 && std::strcmp(argv[7], "-") != 0) { const uint32_t bits = (uint32_t)std::strtoul(argv[7], nullptr, 16); std::memcpy(&z, &bits, 4); }
        const bool cpu = argc > 8 && std::strcmp(argv[8], "cpu") == 0;
        std::vector<Ray3D> rays;
        rays.reserve((size_t)width * height);
        for (int jj = 0; jj < height; ++jj)
            for (int ii = 0; ii < width; ++ii)
                rays.emplace_back(rtm::vec3(0, 0, 0), rtm::vec3((float)ii - width / 2.0f, (float)(height - jj) - height / 2.0f, z));
        std::unique_ptr<IRaytracer> raytracer;
        if (cpu) raytracer.reset(new CPURaytracer(objects, lights, rays, depth));
        else raytracer.reset(new HIPRaytracer(objects, lights, rays, depth));
        cl_float4* pixels = raytracer->Render();
        PPMExporter::ExportP3(argv[6], (size_t)width, (size_t)height, PPMExporter::RGBAtoRGB(reinterpret_cast<const float*>(pixels), (size_t)width * height));
        std::printf("wrote %s\n", argv[6]);
        return 0;
    } catch (const std::out_of_range& e) {
        std::printf("out_of_range %s\n", e.what());
        return 3;
    } catch (const std::exception& e) {
        std::printf("%s\n", e.what());
        return 2;
    }
}
