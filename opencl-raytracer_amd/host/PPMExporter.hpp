// PPMExporter.hpp - ASCII PPM sink with the reference's signature (PPMExporter.hpp:8) and byte-for-byte output
// (PPMExporter.cpp:13-23): "P3\n<w> <h>\n255\n", then per pixel `min(255, (int)floorf(v * 255.f))` for r, g, b
// separated by single spaces, one pixel per line. RGBAtoRGB repacks Render()'s float4 frame to the packed
// stride-3 vector ExportP3 expects.
#pragma once

#include <cstddef>
#include <string>
#include <vector>

class PPMExporter {
public:
    static void ExportP3(const std::string& outFileLoc, size_t width, size_t height, const std::vector<float>& pixelData);
    static std::vector<float> RGBAtoRGB(const float* rgba, size_t pixels);
};
