// CPURaytracer.cpp - see CPURaytracer.hpp. Own code: it shares nothing with oracle/ (the test-side checker) or with the
// HIP kernels; what it shares with both is the specification, i.e. the reference kernels' statements, cited inline as
// `.cl:<line>` = shade_and_reflect_kernel.cl (shade_kernel.cl / hittest_kernel.cl where they differ).
//
// Built with -ffp-contract=off: a multiply-add is fused exactly where fmuladd() is written (the sites the OpenCL
// front-end contracts, re-derived from the kernels' LLVM IR: in a*x + b*y + c*z + d*w the first product is fused onto
// the second), everything else rounds after every operation. dot() and normalize() are the plain left-to-right forms.
#include "CPURaytracer.hpp"

#include <algorithm>
#include <atomic>
#include <cmath>
#include <cstring>
#include <thread>

#define RT_INLINE inline __attribute__((always_inline))

// Per object, split by who reads it: the object loop streams `rows` + `type` for EVERY ray (52 of the record's 248 bytes -
// with 100 000 objects the loop is bound by how fast the cores can stream this array), everything else is touched once
// per finished ray.
struct CPURaytracer::Instance {
    float rows[3][4];  // rows x, y, z of mvInverse: rows[r] = (s_r, s_{4+r}, s_{8+r}, s_{12+r})
    int type;          // 0 sphere, 1 box, anything else: the kernels' switch has no case for it - never hit
};
struct CPURaytracer::Surface {
    float inv[16];     // mvInverse, column-major (s0..sF)
    float mv[16];
    float ambient[3], diffuse[3], specular[3];
    float absorption, shininess;
};

namespace {

struct F3 { float x, y, z; };
struct F4 { float x, y, z, w; };
struct Ray { F4 start, direction; };
struct Hit {
    float time;
    int index;
    F4 intersection;
    F3 normal, reflection;
};
struct Scene {
    const CPURaytracer::Instance* objs;
    const CPURaytracer::Surface* surf;
    size_t n_objs;
    const Light* lights;
    size_t n_lights;
};

RT_INLINE float fmuladd(float a, float b, float c) { return __builtin_fmaf(a, b, c); }

// transform() (.cl:60-65), one component: m[r]*v.x + m[4+r]*v.y + m[8+r]*v.z + m[12+r]*v.w
RT_INLINE float transform_row(const float* m, int r, const F4& v) {
    float t = m[4 + r] * v.y;
    t = fmuladd(m[r], v.x, t);
    t = fmuladd(m[8 + r], v.z, t);
    t = fmuladd(m[12 + r], v.w, t);
    return t;
}
RT_INLINE float row_of(const float* r, const F4& v) {  // the same component from a row stored contiguously
    float t = r[1] * v.y;
    t = fmuladd(r[0], v.x, t);
    t = fmuladd(r[2], v.z, t);
    t = fmuladd(r[3], v.w, t);
    return t;
}
RT_INLINE F4 transform(const float* m, const F4& v) {
    return F4{transform_row(m, 0, v), transform_row(m, 1, v), transform_row(m, 2, v), transform_row(m, 3, v)};
}
RT_INLINE float dot(const F3& a, const F3& b) {
    float s = a.x * b.x;
    s = s + a.y * b.y;
    s = s + a.z * b.z;
    return s;
}
RT_INLINE F3 normalize(const F3& v) {
    const float len = std::sqrt(dot(v, v));
    return F3{v.x / len, v.y / len, v.z / len};
}
// reflect() (.cl:68-70): incident - 2 dot(incident, normal) normal, the last step contracted
RT_INLINE F3 reflect(const F3& i, const F3& n) {
    const float k = dot(i, n) * -2.0f;
    return F3{fmuladd(k, n.x, i.x), fmuladd(k, n.y, i.y), fmuladd(k, n.z, i.z)};
}

// one slab of the unit box, intersectsWidthBoxSide() (.cl:33-58)
RT_INLINE bool box_side(float& tMin, float& tMax, float start, float dir) {
    float t1 = -0.5f - start;
    float t2 = 0.5f - start;
    if (dir == 0) {
        if (std::copysign(t1, t2) == t1) return false;  // both walls on the same side of the origin
        tMin = -MAX_FLOAT;
        tMax = MAX_FLOAT;
        return true;
    }
    t1 = t1 / dir;
    t2 = t2 / dir;
    if (dir < 0) { tMin = std::fmin(t1, t2); tMax = std::fmax(t1, t2); }
    else { tMin = t1; tMax = t2; }
    return true;
}

// The object loop of raycast() (.cl:72-171). The kernels rebuild the whole hit record on every improvement; only
// (time, object) decide the outcome, so the record is built once, after the loop, for the winner (finish_hit).
RT_INLINE void closest(const Scene& sc, const Ray& ray, float& time, int& index) {
    for (size_t k = 0; k < sc.n_objs; ++k) {
        const CPURaytracer::Instance& o = sc.objs[k];
        // transform(start), transform(direction) by mvInverse (.cl:78-79); the w components only matter to the hit record
        const F3 s{row_of(o.rows[0], ray.start), row_of(o.rows[1], ray.start), row_of(o.rows[2], ray.start)};
        const F3 d{row_of(o.rows[0], ray.direction), row_of(o.rows[1], ray.direction), row_of(o.rows[2], ray.direction)};
        if (o.type == 0) {  // unit sphere (.cl:82-121)
            float A = d.y * d.y; A = fmuladd(d.x, d.x, A); A = fmuladd(d.z, d.z, A);
            float B = s.y * d.y; B = fmuladd(d.x, s.x, B); B = fmuladd(d.z, s.z, B); B = B * 2.0f;
            float C = s.y * s.y; C = fmuladd(s.x, s.x, C); C = fmuladd(s.z, s.z, C); C = C + -1.0f;
            const float radical = fmuladd(B, B, (A * 4.0f) * (-C));
            if (radical < 0) continue;
            const float root = std::sqrt(radical);
            const float t1 = (-B - root) / (A * 2.0f);
            const float t2 = (-B + root) / (A * 2.0f);
            const float tMin = (t1 >= 0 && t2 >= 0) ? std::fmin(t1, t2) : std::fmax(t1, t2);
            if (tMin < 0) continue;
            if (time < tMin) continue;  // an equal time goes to the LATER sphere (.cl:108)
            time = tMin;
            index = (int)k;
        } else if (o.type == 1) {  // unit box (.cl:123-168)
            float txMin, txMax, tyMin, tyMax, tzMin, tzMax;
            if (!box_side(txMin, txMax, s.x, d.x)) continue;
            if (!box_side(tyMin, tyMax, s.y, d.y)) continue;
            if (!box_side(tzMin, tzMax, s.z, d.z)) continue;
            const float tMin = std::fmax(std::fmax(txMin, tyMin), tzMin);
            const float tMax = std::fmin(std::fmin(txMax, tyMax), tzMax);
            if (tMax < tMin) continue;
            const float tHit = (tMin >= 0 && tMax >= 0) ? std::fmin(tMin, tMax) : std::fmax(tMin, tMax);
            if (tHit < 0) continue;
            if (time <= tHit) continue;  // an equal time stays with the EARLIER object (.cl:147)
            time = tHit;
            index = (int)k;
        }
    }
}

// what raycast() leaves in the hit record for the winner (.cl:110-119 sphere, :149-166 box, :175 reflection)
RT_INLINE void finish_hit(const Scene& sc, const Ray& ray, Hit& h) {
    const CPURaytracer::Surface& o = sc.surf[h.index];
    const int type = sc.objs[h.index].type;
    const F4 s = transform(o.inv, ray.start), d = transform(o.inv, ray.direction);
    const F4 p{fmuladd(h.time, d.x, s.x), fmuladd(h.time, d.y, s.y), fmuladd(h.time, d.z, s.z), fmuladd(h.time, d.w, s.w)};
    F4 n{0.f, 0.f, 0.f, 0.f};
    if (type == 0) {
        n.x = p.x; n.y = p.y; n.z = p.z;
    } else {
        if (p.x > 0.4998f) n.x += 1.f; else if (p.x < -0.4998f) n.x -= 1.f;
        if (p.y > 0.4998f) n.y += 1.f; else if (p.y < -0.4998f) n.y -= 1.f;
        if (p.z > 0.4998f) n.z += 1.f; else if (p.z < -0.4998f) n.z -= 1.f;
    }
    h.intersection = transform(o.mv, p);
    const F4 nv = transform(o.mv, n);  // normals go through mv, not its inverse transpose (Q2)
    h.normal = normalize(F3{nv.x, nv.y, nv.z});
    h.reflection = reflect(F3{ray.direction.x, ray.direction.y, ray.direction.z}, h.normal);
}

// shade() / the body of __kernel shade (.cl:184-242, shade_kernel.cl:197-258). ACCUMULATE: shade_kernel.cl:252 sums the
// lights; shade_and_reflect's `fColor = ...` (.cl:238) keeps the last one. `specular` survives an iteration that is lit
// with nDotL <= 0 (Q1b).
template <bool ACCUMULATE>
RT_INLINE F3 shade(const Scene& sc, const Hit& hit, uint64_t& rays) {
    const CPURaytracer::Surface& mat = sc.surf[hit.index];
    const F3 P{hit.intersection.x, hit.intersection.y, hit.intersection.z};
    F3 color{0.f, 0.f, 0.f}, specular{0.f, 0.f, 0.f};
    for (size_t li = 0; li < sc.n_lights; ++li) {
        const Light& L = sc.lights[li];
        F3 lightVec;
        if (L.lightPosition.w != 0) lightVec = F3{L.lightPosition.x - P.x, L.lightPosition.y - P.y, L.lightPosition.z - P.z};
        else lightVec = F3{-L.lightPosition.x, -L.lightPosition.y, -L.lightPosition.z};
        // shadow ray: un-normalised direction, 0.01 of skin along it (.cl:201-205); a FULL closest-hit raycast (:209)
        const F3 nl = normalize(lightVec);
        Ray toLight;
        toLight.start = F4{fmuladd(nl.x, 0.01f, P.x), fmuladd(nl.y, 0.01f, P.y), fmuladd(nl.z, 0.01f, P.z), fmuladd(0.0f, 0.01f, 1.0f)};
        toLight.direction = F4{lightVec.x, lightVec.y, lightVec.z, 0.f};
        float st = MAX_FLOAT;
        int si = -1;
        closest(sc, toLight, st, si);
        rays += 1;
        const F3 normalView = normalize(hit.normal);  // (normalises an already normalised vector - the low bits move, Q12)
        const float nDotL = dot(normalView, nl);
        const F3 viewVec = normalize(F3{-P.x, -P.y, -P.z});
        const F3 reflectVec = normalize(reflect(F3{-nl.x, -nl.y, -nl.z}, normalView));
        const float rDotV = std::fmax(dot(reflectVec, viewVec), 0.0f);
        const F3 ambient{mat.ambient[0] * L.ambient.x, mat.ambient[1] * L.ambient.y, mat.ambient[2] * L.ambient.z};
        F3 diffuse;
        if (st >= 1.f || st < 0) {  // nothing between the point and the light (:229)
            const float nd = std::fmax(nDotL, 0.f);
            diffuse = F3{(mat.diffuse[0] * L.diffuse.x) * nd, (mat.diffuse[1] * L.diffuse.y) * nd, (mat.diffuse[2] * L.diffuse.z) * nd};
            if (nDotL > 0) {
                const float pw = std::pow(rDotV, std::fmax(mat.shininess, 1.f));
                specular = F3{(mat.specular[0] * L.specular.x) * pw, (mat.specular[1] * L.specular.y) * pw, (mat.specular[2] * L.specular.z) * pw};
            }
        } else {
            diffuse = F3{0.f, 0.f, 0.f};
            specular = F3{0.f, 0.f, 0.f};
        }
        if (ACCUMULATE) color = F3{((color.x + ambient.x) + diffuse.x) + specular.x, ((color.y + ambient.y) + diffuse.y) + specular.y, ((color.z + ambient.z) + diffuse.z) + specular.z};
        else color = F3{(ambient.x + diffuse.x) + specular.x, (ambient.y + diffuse.y) + specular.y, (ambient.z + diffuse.z) + specular.z};
    }
    return color;
}

// start = intersection + 0.001 normalize(reflection), direction = reflection (.cl:260-263, 275-277)
RT_INLINE Ray reflection_ray(const Hit& h) {
    const F3 n = normalize(h.reflection);
    Ray r;
    r.start = F4{fmuladd(n.x, 0.001f, h.intersection.x), fmuladd(n.y, 0.001f, h.intersection.y), fmuladd(n.z, 0.001f, h.intersection.z),
                 fmuladd(0.0f, 0.001f, h.intersection.w)};
    r.direction = F4{h.reflection.x, h.reflection.y, h.reflection.z, 0.f};
    return r;
}

// one work-item of the three kernels; false = the kernel returned without writing its output element
template <int KERNEL>
RT_INLINE bool work_item(const Scene& sc, const Ray3D& in, unsigned int max_bounces, float out[3], uint64_t& rays) {
    Ray ray;
    ray.start = F4{in.start.x, in.start.y, in.start.z, in.start.w};
    ray.direction = F4{in.direction.x, in.direction.y, in.direction.z, in.direction.w};
    Hit hit;
    hit.time = MAX_FLOAT;
    hit.index = -1;
    closest(sc, ray, hit.time, hit.index);
    rays += 1;
    // raycast()'s verdict: `time == MAX_FLOAT -> false` in shade_and_reflect (.cl:173), `time < MAX_FLOAT` in the two
    // older kernels (shade_kernel.cl:167, hittest_kernel.cl:149) - they part ways on a NaN time
    const bool is_hit = (KERNEL == 2) ? !(hit.time == MAX_FLOAT) : (hit.time < MAX_FLOAT);
    if (!is_hit) return false;
    if (KERNEL == 0) { out[0] = hit.time; return true; }
    finish_hit(sc, ray, hit);
    if (KERNEL == 1) {
        const F3 c = shade<true>(sc, hit, rays);
        out[0] = c.x; out[1] = c.y; out[2] = c.z;
        return true;
    }
    // __kernel shade_and_reflect (.cl:253-284)
    const F3 first = shade<false>(sc, hit, rays);
    float absorptionPercent = sc.surf[hit.index].absorption;
    F3 absorbColor{first.x * absorptionPercent, first.y * absorptionPercent, first.z * absorptionPercent};
    F3 reflectColor{0.f, 0.f, 0.f};
    unsigned int bounces = max_bounces;
    Hit from = hit;
    // while (bounces-- > 0 && raycast(...) && absorptionPercent <= 0.999f)   - the unsigned post-decrement happens whether
    // or not the test passes, and the ray is cast before the absorption is looked at (Q8)
    for (;;) {
        const unsigned int before = bounces;
        bounces = bounces - 1u;
        if (!(before > 0u)) break;
        const Ray rr = reflection_ray(from);
        Hit rh;
        rh.time = MAX_FLOAT;
        rh.index = -1;
        closest(sc, rr, rh.time, rh.index);
        rays += 1;
        if (rh.time == MAX_FLOAT) break;
        if (!(absorptionPercent <= 0.999f)) break;
        finish_hit(sc, rr, rh);
        reflectColor = shade<false>(sc, rh, rays);
        const float reflectedAbsorption = (1.f - absorptionPercent) * sc.surf[rh.index].absorption;
        absorbColor = F3{fmuladd(reflectedAbsorption, reflectColor.x, absorbColor.x), fmuladd(reflectedAbsorption, reflectColor.y, absorbColor.y),
                         fmuladd(reflectedAbsorption, reflectColor.z, absorbColor.z)};
        absorptionPercent = absorptionPercent + reflectedAbsorption;
        from = rh;
    }
    if (bounces == 0u && absorptionPercent < 1.f) {  // (.cl:281-282) - only when the loop ended on its last permitted iteration
        const float w = 1.f - absorptionPercent;
        absorbColor = F3{fmuladd(w, reflectColor.x, absorbColor.x), fmuladd(w, reflectColor.y, absorbColor.y), fmuladd(w, reflectColor.z, absorbColor.z)};
    }
    out[0] = absorbColor.x; out[1] = absorbColor.y; out[2] = absorbColor.z;
    return true;
}

// a tile of consecutive work-items; compiled twice, the FMA3 clone is picked at load time where the CPU has it (the
// generic clone calls libm's fmaf: same bits, slower)
template <int KERNEL>
__attribute__((target_clones("fma", "default")))
void render_tile(const Scene& sc, const Ray3D* rays, size_t first, size_t last, unsigned int max_bounces, cl_float4* pixels,
                 uint64_t& traced, uint64_t& hits) {
    for (size_t i = first; i < last; ++i) {
        float out[3] = {0.f, 0.f, 0.f};
        if (work_item<KERNEL>(sc, rays[i], max_bounces, out, traced)) {
            hits += 1;
            if (KERNEL == 0) pixels[i].s[0] = out[0];
            else { pixels[i].s[0] = out[0]; pixels[i].s[1] = out[1]; pixels[i].s[2] = out[2]; pixels[i].s[3] = 1.0f; }
        }
    }
}

}  // namespace

CPURaytracer::CPURaytracer(const std::vector<ObjectData>& objects_, const std::vector<Light>& lights_, const std::vector<Ray3D>& rays_,
                           unsigned int MAX_BOUNCES, Kernel kernel_, unsigned int threads)
    : IRaytracer(objects_, lights_, rays_), max_bounces(MAX_BOUNCES), kernel(kernel_) {
    n_threads = threads ? threads : std::max(1u, std::thread::hardware_concurrency());
    instances.resize(objects.size());
    surfaces.resize(objects.size());
    for (size_t i = 0; i < objects.size(); ++i) {  // what the reference's ctor does for the device (OpenCLRaytracer.cpp:16-24)
        const ObjectData& o = objects[i];
        Surface& d = surfaces[i];
        std::memcpy(d.inv, o.mvInverse.data(), sizeof(d.inv));
        std::memcpy(d.mv, o.mv.data(), sizeof(d.mv));
        d.ambient[0] = o.mat.ambient.x; d.ambient[1] = o.mat.ambient.y; d.ambient[2] = o.mat.ambient.z;
        d.diffuse[0] = o.mat.diffuse.x; d.diffuse[1] = o.mat.diffuse.y; d.diffuse[2] = o.mat.diffuse.z;
        d.specular[0] = o.mat.specular.x; d.specular[1] = o.mat.specular.y; d.specular[2] = o.mat.specular.z;
        d.absorption = o.mat.absorption;
        d.shininess = o.mat.shininess;
        Instance& h = instances[i];
        for (int r = 0; r < 3; ++r)
            for (int c = 0; c < 4; ++c) h.rows[r][c] = d.inv[4 * c + r];
        h.type = static_cast<int>(o.type);
    }
    pixels.resize(rays.size());
}

CPURaytracer::~CPURaytracer() {}

cl_float4* CPURaytracer::Render() {
    const size_t n = rays.size();
    for (size_t i = 0; i < n; ++i) {  // the buffer as the reference uploads it: {0,0,0,1} (OpenCLRaytracer.cpp:32); hittest: "no hit"
        if (kernel == kHittest) pixels[i] = cl_float4{{MAX_FLOAT, 0.f, 0.f, 0.f}};
        else pixels[i] = cl_float4{{0.f, 0.f, 0.f, 1.f}};
    }
    const Scene sc{instances.data(), surfaces.data(), instances.size(), lights.data(), lights.size()};
    // row-tiles handed out from one counter: cost per tile varies with what the rays hit
    const size_t tile = std::max<size_t>(1, std::min<size_t>(256, n / ((size_t)n_threads * 8u)));
    std::atomic<size_t> next{0};
    std::atomic<uint64_t> total_rays{0}, total_hits{0};
    auto worker = [&]() {
        uint64_t traced = 0, hits = 0;
        for (;;) {
            const size_t first = next.fetch_add(tile);
            if (first >= n) break;
            const size_t last = std::min(n, first + tile);
            switch (kernel) {
                case kHittest: render_tile<0>(sc, rays.data(), first, last, max_bounces, pixels.data(), traced, hits); break;
                case kShade: render_tile<1>(sc, rays.data(), first, last, max_bounces, pixels.data(), traced, hits); break;
                default: render_tile<2>(sc, rays.data(), first, last, max_bounces, pixels.data(), traced, hits); break;
            }
        }
        total_rays += traced;
        total_hits += hits;
    };
    const unsigned int nt = (unsigned int)std::min<size_t>(n_threads, (n + tile - 1) / tile ? (n + tile - 1) / tile : 1);
    std::vector<std::thread> pool;
    for (unsigned int t = 1; t < nt; ++t) pool.emplace_back(worker);
    worker();
    for (std::thread& t : pool) t.join();
    rays_traced = total_rays.load();
    hit_pixels = total_hits.load();
    return pixels.data();
}
