// host_test.cpp - exercises the C++ flavour of the boundary the way the reference's main() does
// (OpenCL-Raytracer.cpp:62-87): build the scene vectors, build the primary rays, construct the backend through
// the IRaytracer base, Render(), look at the pixels. Prints one line per check for tests/test_host_cpp_gpu.py.
#include <cmath>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <memory>

#include "HIPRaytracer.hpp"

int main(int argc, char** argv) {
    const int width = 256, height = 256;
    float fov = 60.f * 0.01745329251994329576923690768489f;
    fov *= 0.5f;

    // the `simpleSphere` scene: one unit sphere under the loader's root lookAt = translate(0,0,-10), one light
    Material m;
    m.ambient = rtm::vec3(1, 0, 0);
    m.diffuse = rtm::vec3(0, 1, 0);
    m.specular = rtm::vec3(0, 0, 1);
    LightProperties lp;
    lp.ambient = rtm::vec3(.3f, .3f, .3f);
    lp.diffuse = rtm::vec3(.7f, .7f, .7f);
    lp.specular = rtm::vec3(1, 1, 1);
    const rtm::mat4 root = rtm::translate(rtm::mat4(1.f), rtm::vec3(0, 0, -10));
    std::vector<ObjectData> objects;
    objects.emplace_back(ObjectData::PrimativeType::sphere, m, root);
    std::vector<Light> lights;
    lights.emplace_back(lp, rtm::translate(root, rtm::vec3(10, 10, 10)));

    std::vector<Ray3D> rays;
    rays.reserve((size_t)width * height);
    // argv[2] (optional): camera z as a float bit pattern, so that a comparison run uses bit-identical rays
    float z = -((height / 2.0f) / tanf(fov));
    if (argc > 2) {
        const uint32_t bits = (uint32_t)std::strtoul(argv[2], nullptr, 16);
        std::memcpy(&z, &bits, 4);
    }
    for (int jj = 0; jj < height; ++jj)
        for (int ii = 0; ii < width; ++ii)
            rays.emplace_back(rtm::vec3(0, 0, 0), rtm::vec3((float)ii - width / 2.0f, (float)(height - jj) - height / 2.0f, z));

    try {
        std::unique_ptr<IRaytracer> raytracer(new HIPRaytracer(objects, lights, rays, 3));
        cl_float4* pixels = raytracer->Render();
        cl_float4* again = raytracer->Render();
        int nonblack = 0;
        double sum[3] = {0, 0, 0};
        for (int i = 0; i < width * height; ++i) {
            const float* p = reinterpret_cast<const float*>(&pixels[i]);
            if (p[0] != 0 || p[1] != 0 || p[2] != 0) ++nonblack;
            for (int c = 0; c < 3; ++c) sum[c] += p[c];
        }
        const float* c = reinterpret_cast<const float*>(&pixels[128 * width + 128]);
        uint32_t bits[3];
        std::memcpy(bits, c, 12);
        rt_stats_t st = static_cast<HIPRaytracer*>(raytracer.get())->Stats();
        std::printf("same_buffer %d\n", pixels == again);
        std::printf("nonblack %d\n", nonblack);
        std::printf("center %08x %08x %08x\n", bits[0], bits[1], bits[2]);
        std::printf("sum %.6f %.6f %.6f\n", sum[0], sum[1], sum[2]);
        std::printf("pinhole %u %u %u\n", st.pinhole, st.width, st.height);
        std::printf("kernel_ms %.4f\n", st.last_kernel_ms);
        if (argc > 3) {  // argv[3] = n: the same frame from n shards (contexts on device 0) through the several-GPU constructor
            const int n = std::atoi(argv[3]);
            std::unique_ptr<IRaytracer> sharded(new HIPRaytracer(objects, lights, rays, 3, std::vector<int>((size_t)n, 0)));
            cl_float4* frame = sharded->Render();
            std::printf("multi_equal %d\n", std::memcmp(frame, pixels, sizeof(cl_float4) * (size_t)width * height) == 0);
        }
        if (argc > 1) {  // dump raw RGBA for comparison
            FILE* f = std::fopen(argv[1], "wb");
            if (!f) return 3;
            std::fwrite(pixels, 16, (size_t)width * height, f);
            std::fclose(f);
        }
    } catch (const std::exception& e) {
        std::printf("error %s\n", e.what());
        return 2;
    }
    return 0;
}
