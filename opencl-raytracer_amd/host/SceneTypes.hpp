// SceneTypes.hpp - glm-free host records with the reference's names and field order, so that code written
// against the reference's headers (Material.hpp:6-11, Light.hpp:5-16, Ray3D.hpp:5-10, HitRecord.hpp:8-16,
// ObjectData.hpp:6-27) compiles against the HIP backend unchanged. A maintainer who already has glm keeps
// using the reference's own headers: the field layouts are identical (vec3 = 3 floats, vec4 = 4 floats,
// mat4 = 16 floats column-major), which is all HIPRaytracer relies on.
#pragma once

#include <cmath>
#include <cstdint>
#include <limits>

namespace rtm {

struct vec3 {
    float x, y, z;
    vec3() : x(0), y(0), z(0) {}
    vec3(float x_, float y_, float z_) : x(x_), y(y_), z(z_) {}
};

struct vec4 {
    float x, y, z, w;
    vec4() : x(0), y(0), z(0), w(0) {}
    vec4(float x_, float y_, float z_, float w_) : x(x_), y(y_), z(z_), w(w_) {}
    vec4(const vec3& v, float w_) : x(v.x), y(v.y), z(v.z), w(w_) {}
};

// column-major 4x4: c[col][row], contiguous like glm::mat4 / glm::value_ptr
struct mat4 {
    float c[4][4];
    explicit mat4(float d = 1.f) {
        for (int i = 0; i < 4; ++i)
            for (int j = 0; j < 4; ++j) c[i][j] = (i == j) ? d : 0.f;
    }
    const float* data() const { return &c[0][0]; }
    float* data() { return &c[0][0]; }
};

inline mat4 operator*(const mat4& a, const mat4& b) {
    mat4 r(0.f);
    for (int j = 0; j < 4; ++j)
        for (int i = 0; i < 4; ++i) {
            float s = a.c[0][i] * b.c[j][0];
            s += a.c[1][i] * b.c[j][1];
            s += a.c[2][i] * b.c[j][2];
            s += a.c[3][i] * b.c[j][3];
            r.c[j][i] = s;
        }
    return r;
}

inline vec4 operator*(const mat4& m, const vec4& v) {
    vec4 r;
    r.x = (m.c[0][0] * v.x + m.c[1][0] * v.y) + (m.c[2][0] * v.z + m.c[3][0] * v.w);
    r.y = (m.c[0][1] * v.x + m.c[1][1] * v.y) + (m.c[2][1] * v.z + m.c[3][1] * v.w);
    r.z = (m.c[0][2] * v.x + m.c[1][2] * v.y) + (m.c[2][2] * v.z + m.c[3][2] * v.w);
    r.w = (m.c[0][3] * v.x + m.c[1][3] * v.y) + (m.c[2][3] * v.z + m.c[3][3] * v.w);
    return r;
}

inline mat4 translate(const mat4& m, const vec3& v) {
    mat4 r = m;
    for (int i = 0; i < 4; ++i) r.c[3][i] = m.c[0][i] * v.x + m.c[1][i] * v.y + m.c[2][i] * v.z + m.c[3][i];
    return r;
}

inline mat4 scale(const mat4& m, const vec3& v) {
    mat4 r = m;
    for (int i = 0; i < 4; ++i) {
        r.c[0][i] = m.c[0][i] * v.x;
        r.c[1][i] = m.c[1][i] * v.y;
        r.c[2][i] = m.c[2][i] * v.z;
    }
    return r;
}

inline vec3 normalize(const vec3& v) {
    const float inv = 1.0f / std::sqrt(v.x * v.x + v.y * v.y + v.z * v.z);
    return vec3(v.x * inv, v.y * inv, v.z * inv);
}
inline vec3 cross(const vec3& a, const vec3& b) { return vec3(a.y * b.z - a.z * b.y, a.z * b.x - a.x * b.z, a.x * b.y - a.y * b.x); }
inline float dot(const vec3& a, const vec3& b) { return a.x * b.x + a.y * b.y + a.z * b.z; }
inline float radians(float deg) { return deg * 0.01745329251994329576923690768489f; }

// axis-angle rotation appended to m (axis must be normalised), glm's formulation
inline mat4 rotate(const mat4& m, float angle, const vec3& axis) {
    const float c = std::cos(angle), s = std::sin(angle);
    const vec3 t((1.f - c) * axis.x, (1.f - c) * axis.y, (1.f - c) * axis.z);
    float r[3][3];
    r[0][0] = c + t.x * axis.x;          r[0][1] = t.x * axis.y + s * axis.z; r[0][2] = t.x * axis.z - s * axis.y;
    r[1][0] = t.y * axis.x - s * axis.z; r[1][1] = c + t.y * axis.y;          r[1][2] = t.y * axis.z + s * axis.x;
    r[2][0] = t.z * axis.x + s * axis.y; r[2][1] = t.z * axis.y - s * axis.x; r[2][2] = c + t.z * axis.z;
    mat4 out(0.f);
    for (int i = 0; i < 4; ++i) {
        out.c[0][i] = m.c[0][i] * r[0][0] + m.c[1][i] * r[0][1] + m.c[2][i] * r[0][2];
        out.c[1][i] = m.c[0][i] * r[1][0] + m.c[1][i] * r[1][1] + m.c[2][i] * r[1][2];
        out.c[2][i] = m.c[0][i] * r[2][0] + m.c[1][i] * r[2][1] + m.c[2][i] * r[2][2];
        out.c[3][i] = m.c[3][i];
    }
    return out;
}

// right-handed look-at
inline mat4 lookAt(const vec3& eye, const vec3& center, const vec3& up) {
    const vec3 f = normalize(vec3(center.x - eye.x, center.y - eye.y, center.z - eye.z));
    const vec3 s = normalize(cross(f, up));
    const vec3 u = cross(s, f);
    mat4 r(1.f);
    r.c[0][0] = s.x; r.c[1][0] = s.y; r.c[2][0] = s.z;
    r.c[0][1] = u.x; r.c[1][1] = u.y; r.c[2][1] = u.z;
    r.c[0][2] = -f.x; r.c[1][2] = -f.y; r.c[2][2] = -f.z;
    r.c[3][0] = -dot(s, eye); r.c[3][1] = -dot(u, eye); r.c[3][2] = dot(f, eye);
    return r;
}

inline mat4 transpose(const mat4& m) {
    mat4 r(0.f);
    for (int i = 0; i < 4; ++i)
        for (int j = 0; j < 4; ++j) r.c[i][j] = m.c[j][i];
    return r;
}

// general inverse by cofactors (adjugate / determinant)
inline mat4 inverse(const mat4& m) {
    const float* a = m.data();
    float inv[16];
    inv[0] = a[5] * a[10] * a[15] - a[5] * a[11] * a[14] - a[9] * a[6] * a[15] + a[9] * a[7] * a[14] + a[13] * a[6] * a[11] - a[13] * a[7] * a[10];
    inv[4] = -a[4] * a[10] * a[15] + a[4] * a[11] * a[14] + a[8] * a[6] * a[15] - a[8] * a[7] * a[14] - a[12] * a[6] * a[11] + a[12] * a[7] * a[10];
    inv[8] = a[4] * a[9] * a[15] - a[4] * a[11] * a[13] - a[8] * a[5] * a[15] + a[8] * a[7] * a[13] + a[12] * a[5] * a[11] - a[12] * a[7] * a[9];
    inv[12] = -a[4] * a[9] * a[14] + a[4] * a[10] * a[13] + a[8] * a[5] * a[14] - a[8] * a[6] * a[13] - a[12] * a[5] * a[10] + a[12] * a[6] * a[9];
    inv[1] = -a[1] * a[10] * a[15] + a[1] * a[11] * a[14] + a[9] * a[2] * a[15] - a[9] * a[3] * a[14] - a[13] * a[2] * a[11] + a[13] * a[3] * a[10];
    inv[5] = a[0] * a[10] * a[15] - a[0] * a[11] * a[14] - a[8] * a[2] * a[15] + a[8] * a[3] * a[14] + a[12] * a[2] * a[11] - a[12] * a[3] * a[10];
    inv[9] = -a[0] * a[9] * a[15] + a[0] * a[11] * a[13] + a[8] * a[1] * a[15] - a[8] * a[3] * a[13] - a[12] * a[1] * a[11] + a[12] * a[3] * a[9];
    inv[13] = a[0] * a[9] * a[14] - a[0] * a[10] * a[13] - a[8] * a[1] * a[14] + a[8] * a[2] * a[13] + a[12] * a[1] * a[10] - a[12] * a[2] * a[9];
    inv[2] = a[1] * a[6] * a[15] - a[1] * a[7] * a[14] - a[5] * a[2] * a[15] + a[5] * a[3] * a[14] + a[13] * a[2] * a[7] - a[13] * a[3] * a[6];
    inv[6] = -a[0] * a[6] * a[15] + a[0] * a[7] * a[14] + a[4] * a[2] * a[15] - a[4] * a[3] * a[14] - a[12] * a[2] * a[7] + a[12] * a[3] * a[6];
    inv[10] = a[0] * a[5] * a[15] - a[0] * a[7] * a[13] - a[4] * a[1] * a[15] + a[4] * a[3] * a[13] + a[12] * a[1] * a[7] - a[12] * a[3] * a[5];
    inv[14] = -a[0] * a[5] * a[14] + a[0] * a[6] * a[13] + a[4] * a[1] * a[14] - a[4] * a[2] * a[13] - a[12] * a[1] * a[6] + a[12] * a[2] * a[5];
    inv[3] = -a[1] * a[6] * a[11] + a[1] * a[7] * a[10] + a[5] * a[2] * a[11] - a[5] * a[3] * a[10] - a[9] * a[2] * a[7] + a[9] * a[3] * a[6];
    inv[7] = a[0] * a[6] * a[11] - a[0] * a[7] * a[10] - a[4] * a[2] * a[11] + a[4] * a[3] * a[10] + a[8] * a[2] * a[7] - a[8] * a[3] * a[6];
    inv[11] = -a[0] * a[5] * a[11] + a[0] * a[7] * a[9] + a[4] * a[1] * a[11] - a[4] * a[3] * a[9] - a[8] * a[1] * a[7] + a[8] * a[3] * a[5];
    inv[15] = a[0] * a[5] * a[10] - a[0] * a[6] * a[9] - a[4] * a[1] * a[10] + a[4] * a[2] * a[9] + a[8] * a[1] * a[6] - a[8] * a[2] * a[5];
    const float det = a[0] * inv[0] + a[1] * inv[4] + a[2] * inv[8] + a[3] * inv[12];
    const float inv_det = 1.0f / det;
    mat4 r(0.f);
    for (int i = 0; i < 16; ++i) r.data()[i] = inv[i] * inv_det;
    return r;
}

}  // namespace rtm

const static float MAX_FLOAT = std::numeric_limits<float>::max();

struct Material {
    rtm::vec3 ambient, diffuse, specular;
    float absorption = 1, reflection = 0, transparency = 0;
    float shininess = 1;
};

struct LightProperties {
    rtm::vec3 ambient, diffuse, specular;
};

struct Light : LightProperties {
    rtm::vec4 lightPosition{0.f, 0.f, 0.f, 1.f};
    Light(const LightProperties& props, const rtm::mat4& modelview) : LightProperties(props) {
        lightPosition = modelview * lightPosition;
    }
};

struct Ray3D {
    rtm::vec4 start;
    rtm::vec4 direction;
    Ray3D(rtm::vec3 start_, rtm::vec3 direction_) : start(start_, 1.f), direction(direction_, 0.f) {}
};

struct HitRecord {
    Material mat;
    rtm::vec3 intersection;
    rtm::vec3 normal;
    rtm::vec3 reflection;
    float time = MAX_FLOAT;
};

class ObjectData {
public:
    enum class PrimativeType : uint8_t { sphere, box };  // (sic) the reference's spelling

    ObjectData(PrimativeType type_, const Material& mat_, const rtm::mat4& mv_)
        : mat(mat_), mv(mv_), mvInverse(rtm::inverse(mv_)), mvInverseTranspose(rtm::transpose(rtm::inverse(mv_))), type(type_) {}

    Material mat;
    rtm::mat4 mv, mvInverse, mvInverseTranspose;
    PrimativeType type;
};
