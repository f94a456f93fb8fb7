// SceneLoader.hpp - text scene -> objects / lights, same entry point as the reference's
// `SceneLoader::Load(path, objects&, lights&)` (SceneLoader.hpp:14). Grammar and error wording follow
// SceneLoader.cpp:50-331; the implementation is a small line/indent state machine of
// its own. Parse errors throw std::runtime_error with 1-based line numbers; an unknown material / light name throws
// std::out_of_range (map::at), as in the reference.
#pragma once

#include <map>
#include <stdexcept>
#include <string>
#include <vector>

#include "SceneTypes.hpp"

class SceneLoader {
public:
    void Load(const std::string& sceneFileLoc, std::vector<ObjectData>& objects, std::vector<Light>& lights);
    void LoadString(const std::string& text, std::vector<ObjectData>& objects, std::vector<Light>& lights);

private:
    struct Line { std::string text; size_t indent; };
    bool Next(Line& out);
    [[noreturn]] void Fail(const std::string& what) const;
    void ParseHeader();
    void ParseBody(std::vector<ObjectData>& objects, std::vector<Light>& lights);

    std::vector<std::string> lines;
    std::map<std::string, Material> materials;
    std::map<std::string, LightProperties> lightProperties;
    size_t lineNum = 0;
    size_t lastIndent = 0;
};
