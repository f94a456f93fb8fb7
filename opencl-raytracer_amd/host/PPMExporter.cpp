#include "PPMExporter.hpp"

#include <algorithm>
#include <cmath>
#include <cstdio>
#include <stdexcept>

void PPMExporter::ExportP3(const std::string& outFileLoc, size_t width, size_t height, const std::vector<float>& pixelData) {
    std::FILE* f = std::fopen(outFileLoc.c_str(), "wb");
    if (!f) throw std::runtime_error("cannot open '" + outFileLoc + "' for writing");
    std::fprintf(f, "P3\n%zu %zu\n255\n", width, height);
    for (size_t i = 0; i < width * height; ++i) {
        int c[3];
        for (int k = 0; k < 3; ++k) c[k] = std::min(255, (int)floorf(pixelData[i * 3 + k] * 255.f));
        std::fprintf(f, "%d %d %d\n", c[0], c[1], c[2]);
    }
    std::fclose(f);
}

std::vector<float> PPMExporter::RGBAtoRGB(const float* rgba, size_t pixels) {
    std::vector<float> rgb(pixels * 3);
    for (size_t i = 0; i < pixels; ++i)
        for (int k = 0; k < 3; ++k) rgb[i * 3 + k] = rgba[i * 4 + k];
    return rgb;
}
