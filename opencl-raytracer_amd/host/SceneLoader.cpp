#include "SceneLoader.hpp"

#include <fstream>
#include <sstream>

namespace {

std::vector<std::string> tokens(const std::string& line) {
    std::istringstream in(line);
    std::vector<std::string> out;
    std::string t;
    while (in >> t) out.push_back(t);
    return out;
}

// `count` floats after the command word; on failure reports how many the reference would have counted
bool floats(const std::vector<std::string>& tok, size_t count, float* dst, size_t& bad_at) {
    for (size_t i = 0; i < count; ++i) {
        bad_at = i;
        if (1 + i >= tok.size()) return false;
        std::istringstream in(tok[1 + i]);
        if (!(in >> dst[i])) return false;
    }
    return true;
}

}  // namespace

void SceneLoader::Fail(const std::string& what) const {
    throw std::runtime_error("Error parsing scene file at line " + std::to_string(lineNum) + ":\n\t" + what);
}

void SceneLoader::Load(const std::string& sceneFileLoc, std::vector<ObjectData>& objects, std::vector<Light>& lights) {
    std::ifstream in(sceneFileLoc);
    if (!in.is_open()) throw std::runtime_error("Scene file '" + sceneFileLoc + "' could not be found.");
    std::stringstream ss;
    ss << in.rdbuf();
    LoadString(ss.str(), objects, lights);
}

void SceneLoader::LoadString(const std::string& text, std::vector<ObjectData>& objects, std::vector<Light>& lights) {
    lines.clear();
    std::istringstream in(text);
    std::string l;
    while (std::getline(in, l)) {
        if (!l.empty() && l.back() == '\r') l.pop_back();
        lines.push_back(l);
    }
    ParseHeader();
    ParseBody(objects, lights);
}

// next significant line: skips blank and `#` lines, enforces even indentation not deeper than the open scope
bool SceneLoader::Next(Line& out) {
    while (lineNum < lines.size()) {
        const std::string& l = lines[lineNum++];
        const size_t first = l.find_first_not_of(' ');
        if (l.empty() || first == std::string::npos) continue;
        if (l[first] == '#') continue;
        if (first % 2 != 0) Fail("line does not have proper indentation, must be multiples of two");
        if (first > lastIndent) Fail("line is indented too far");
        out.text = l;
        out.indent = first;
        return true;
    }
    return false;
}

void SceneLoader::ParseHeader() {
    enum { NONE, MATERIAL, LIGHT } state = NONE;
    std::string name;
    Line ln;
    while (Next(ln)) {
        if (ln.text == "===") return;
        while (lastIndent > ln.indent) { lastIndent -= 2; state = NONE; }
        const std::vector<std::string> tok = tokens(ln.text);
        const std::string& cmd = tok[0];
        float v[4] = {0, 0, 0, 0};
        size_t bad = 0;
        if (state == NONE) {
            if (cmd != "material" && cmd != "light")
                Fail("unsupported command '" + cmd + "' in header\n\tif you are trying to specify properties, ensure the correct level of indentation");
            lastIndent += 2;
            if (tok.size() < 2) Fail(cmd + " expects 1 argument, found 0\n\t" + cmd + " <" + cmd + " name>");
            name = tok[1];
            if (cmd == "material") { materials.emplace(name, Material()); state = MATERIAL; }
            else { lightProperties.emplace(name, LightProperties()); state = LIGHT; }
            continue;
        }
        const bool colour = (cmd == "ambient" || cmd == "diffuse" || cmd == "specular");
        if (colour) {
            if (!floats(tok, 3, v, bad)) Fail(cmd + " expects 3 arguments, found " + std::to_string(bad + 1) + "\n\t" + cmd + " <r> <g> <b>");
            const rtm::vec3 c(v[0], v[1], v[2]);
            if (state == MATERIAL) {
                Material& m = materials[name];
                (cmd == "ambient" ? m.ambient : cmd == "diffuse" ? m.diffuse : m.specular) = c;
            } else {
                LightProperties& p = lightProperties[name];
                (cmd == "ambient" ? p.ambient : cmd == "diffuse" ? p.diffuse : p.specular) = c;
            }
        } else if (state == MATERIAL && (cmd == "absorption" || cmd == "reflection" || cmd == "transparency" || cmd == "shininess")) {
            const std::string what = cmd == "shininess" ? "shininess value" : cmd + " ratio";
            if (!floats(tok, 1, v, bad)) Fail(cmd + " expects 1 argument, found 0\n\t" + cmd + " <" + what + ">");
            Material& m = materials[name];
            (cmd == "absorption" ? m.absorption : cmd == "reflection" ? m.reflection : cmd == "transparency" ? m.transparency : m.shininess) = v[0];
        } else if (cmd == "material" || cmd == "light") {
            Fail("tried to declare a " + cmd + " in a nested scope, unindent to declare a new " + cmd);
        } else {
            Fail("unsupported command '" + cmd + "' while parsing " + (state == MATERIAL ? "material" : "light"));
        }
    }
}

void SceneLoader::ParseBody(std::vector<ObjectData>& objects, std::vector<Light>& lights) {
    // world -> view: the camera sits at (0,0,10) looking at the origin (SceneLoader.cpp:211-216)
    std::vector<rtm::mat4> stack;
    stack.push_back(rtm::mat4(1.f) * rtm::lookAt(rtm::vec3(0, 0, 10), rtm::vec3(0, 0, 0), rtm::vec3(0, 1, 0)));
    stack.push_back(stack.back());
    Line ln;
    while (Next(ln)) {
        while (lastIndent > ln.indent) { lastIndent -= 2; stack.pop_back(); }
        const std::vector<std::string> tok = tokens(ln.text);
        const std::string& cmd = tok[0];
        float v[4] = {0, 0, 0, 0};
        size_t bad = 0;
        if (cmd == "primative") {
            if (tok.size() < 2) Fail("primative expects 2 argument, found 0\n\tprimative <primative type> <material name>");
            if (tok.size() < 3) Fail("primative expects 2 argument, found 1\n\tprimative <primative type> <material name>");
            ObjectData::PrimativeType type;
            if (tok[1] == "sphere") type = ObjectData::PrimativeType::sphere;
            else if (tok[1] == "box") type = ObjectData::PrimativeType::box;
            else Fail("unsupported primative type '" + tok[1] + "'");
            objects.emplace_back(type, materials.at(tok[2]), stack.back());
        } else if (cmd == "light") {
            if (tok.size() < 2) Fail("light expects 1 argument, found 0\n\tlight <light name>");
            lights.emplace_back(lightProperties.at(tok[1]), stack.back());
        } else if (cmd == "translate" || cmd == "scale") {
            if (!floats(tok, 3, v, bad)) Fail(cmd + " expects 3 arguments, found " + std::to_string(bad + 1) + "\n\t" + cmd + " <x> <y> <z>");
            const rtm::vec3 a(v[0], v[1], v[2]);
            const rtm::mat4 op = cmd == "translate" ? rtm::translate(rtm::mat4(1.f), a) : rtm::scale(rtm::mat4(1.f), a);
            stack.push_back(stack.back() * op);
            lastIndent += 2;
        } else if (cmd == "rotate") {
            if (!floats(tok, 4, v, bad))
                Fail("rotate expects 4 arguments, found " + std::to_string(bad + 1) + "\n\trotate <angle in degrees> <axis x> <axis y> <axis z>");
            stack.push_back(stack.back() * rtm::rotate(rtm::mat4(1.f), rtm::radians(v[0]), rtm::normalize(rtm::vec3(v[1], v[2], v[3]))));
            lastIndent += 2;
        } else {
            Fail("unsupported command '" + cmd + "' in body");
        }
    }
}
