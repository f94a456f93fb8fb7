// rt_api.cpp - the C ABI of include/hip_raytracer.h: context life cycle, scene re-pack + upload, render.
//
// Plays the role of the reference's OpenCLRaytracer ctor/Render()/dtor (OpenCLRaytracer.cpp:13-105) for one
// MI355X. There is deliberately no CPU path here: every failure to reach the GPU is an error.
#include "hip_raytracer.h"
#include "rt_records.h"
#include "rt_kernels.h"

#include <hip/hip_runtime.h>

#include <algorithm>
#include <chrono>
#include <cmath>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <limits>
#include <new>
#include <string>
#include <atomic>
#include <condition_variable>
#include <mutex>
#include <thread>
#include <vector>

namespace {

thread_local std::string g_create_error;

struct StopWatch {
    std::chrono::steady_clock::time_point t0 = std::chrono::steady_clock::now();
    double lap_ms() {
        const auto t1 = std::chrono::steady_clock::now();
        const double ms = std::chrono::duration<double, std::milli>(t1 - t0).count();
        t0 = t1;
        return ms;
    }
};

// engineering aid (RT_SETUP_TRACE=1): where the one-time host work goes, lap by lap, on stderr
struct SetupTrace {
    const char* who;
    bool on = std::getenv("RT_SETUP_TRACE") != nullptr;
    StopWatch sw;
    explicit SetupTrace(const char* w) : who(w) {}
    void operator()(const char* what) { if (on) std::fprintf(stderr, "[%s] %-28s %8.2f ms\n", who, what, sw.lap_ms()); }
};

// One-time host work over independent items (per-tile sorts, per-tile block chains, ...) on several threads: f(begin, end) over
// [0, n) in contiguous chunks, the calling thread taking the first one. RT_SETUP_THREADS=1 keeps it serial; the results do not
// depend on the number of threads (every item writes its own outputs).
template <class F>
void parallel_for(size_t n, size_t grain, F&& f) {
    size_t threads = std::thread::hardware_concurrency();
    if (threads == 0) threads = 1;
    threads = std::min<size_t>(threads, 16);
    if (const char* env = std::getenv("RT_SETUP_THREADS")) threads = (size_t)std::max(1, std::atoi(env));
    threads = std::min(threads, n / std::max<size_t>(grain, 1));
    if (threads <= 1) { if (n) f((size_t)0, n); return; }
    const size_t chunk = (n + threads - 1) / threads;
    std::vector<std::thread> pool;
    pool.reserve(threads - 1);
    for (size_t t = 1; t < threads; ++t) {
        const size_t lo = std::min(n, t * chunk), hi = std::min(n, lo + chunk);
        if (lo == hi) continue;
        try { pool.emplace_back([&f, lo, hi] { f(lo, hi); }); }
        catch (...) { f(lo, hi); }  // (no thread to be had: this chunk on the calling thread)
    }
    f((size_t)0, std::min(n, chunk));
    for (std::thread& th : pool) th.join();
}

constexpr uint32_t kTimingSlots = 256;
constexpr uint32_t kMaxPasses = 4;  // of rt_render's frame (render_in_passes)

}  // namespace

struct rt_context {
    int device = 0;
    hipStream_t stream = nullptr;
    uint32_t flags = 0;
    int kernel = 2;
    uint32_t n_objs = 0, n_lights = 0, max_bounces = 0;
    uint64_t n_rays = 0;

    rt::HotPair* d_pairs = nullptr;
    rt::HotPair* d_shadow_pairs = nullptr;  // the same objects sorted by decreasing size (shadow rays are order-free)
    uint32_t n_pairs = 0;
    rt::HotObject* d_hot = nullptr;
    rt::ColdObject* d_cold = nullptr;
    rt::ObjectRecord* d_objrec = nullptr;   // what materialise() reads of an object, in one 128-byte line
    float4* d_bounds = nullptr;            // screen rectangles for the current camera
    std::vector<double> h_spheres;         // per object: bounding sphere cx, cy, cz, R (R = +inf never cull, -inf never hit)
    bool rects_dirty = true;

    rt::LightRec* d_lights = nullptr;
    float4* d_rays = nullptr;
    bool have_rays = false;  // ray buffer uploaded
    bool dir_w_zero = true;

    bool pinhole = false;
    uint32_t width = 0, height = 0;
    float z = 0.f;

    uint64_t tile_rays = 0;
    uint32_t rank = 0, world = 1;
    uint32_t span = 1;      // consecutive ranks one launch stands for (> 1 only inside render_in_passes)
    uint64_t n_local = 0;

    void* d_out = nullptr;  // context-owned device framebuffer
    size_t d_out_bytes = 0;
    void* h_out = nullptr;  // context-owned pinned host framebuffer (Render()'s return value)
    size_t h_out_bytes = 0;
    // rt_render in passes (render_in_passes): the stream the read-backs run on, an event per pass
    hipStream_t copy_stream = nullptr;
    hipEvent_t ev_pass[kMaxPasses] = {};

    float* aux_t = nullptr;  // caller-owned device buffers for the next render
    int32_t* aux_index = nullptr;

    rt::GridDesc grid = {};                 // device pointers owned by this context
    uint2* d_grid_cell_range = nullptr;
    float4* d_grid_cell_rec = nullptr;
    uint32_t* d_grid_entries = nullptr;
    uint32_t* d_grid_always = nullptr;
    float4* d_grid_entry_sphere = nullptr;
    std::vector<double> h_grid_spheres;     // per object: centre + grid radius (inf: always tested, < 0: never hit)
    std::vector<float> h_grid_pre;          // per object: pre-test radius as the grid's entry spheres carry it
    uint2* d_lt_range = nullptr;            // light tiles (rt_grid.h: LightTiles) for the last light's shadow rays
    float4* d_lt_records = nullptr;
    uint4* d_lt_blocks = nullptr;           // the light tiles' lists as blocks of three candidates (LightTiles::blocks)
    uint32_t* d_lt_block_ids = nullptr;
    rt::LightTiles light_tiles = {};
    rt::BlockGrid blocks = {};              // the closest-hit walk's coarse grid of 32-byte blocks (rt_grid.h: BlockGrid)
    uint4* d_walk_blocks = nullptr;
    uint32_t* d_walk_ids = nullptr;
    std::vector<float4> h_walk;             // the unified walk's records while they are being put together (rt_grid.h: GridDesc::walk_rec)
    float4* d_walk_rec = nullptr;
    uint32_t* d_tile_start = nullptr;       // screen tiles (64 x 8 pixels) -> objects a pinhole primary ray can reach
    uint32_t* d_tile_entries = nullptr;
    rt::ScreenTiles tiles = {};
    bool tiles_dirty = true;
    uint32_t tiles_built_for = 0;           // the tile width (as a shift) the last build was asked for
    bool has_triangles = false;             // type-2 records (extension): only the grid path knows them
    int nan_winner = -1;                    // the last sphere / box of the scene decides what a NaN ray ends with (rt_device.h)
    bool nan_winner_sphere = false;
    bool forced_literal = false;            // a degenerate instance switched the context to RT_FLAG_LITERAL (rt_create)
    // Primary directions the exact eliminations are not made for - |d|^2 == 0, below 1e-30 or above 1e30 (or not finite): the
    // reference's tests then produce NaN times for every object (.cl:85-108), which only the literal loops reproduce. Such a
    // frame is rendered the literal way as a whole (apply_ray_domain): `flags` = base_flags | LITERAL while the rays in use
    // (the uploaded buffer, or the pinhole camera that replaced it) hold such a direction.
    uint32_t base_flags = 0;                // `flags` after rt_create's instance checks
    bool rays_out_of_domain = false, camera_out_of_domain = false;
    bool affine_w = true;                   // every mv / mvInverse has bottom row (0,0,0,1) exactly
    bool primary_w_one = true;              // every uploaded primary ray has start.w == 1
    double origin_lo[3] = {0, 0, 0}, origin_hi[3] = {0, 0, 0};  // box of the primary ray origins
    rt::WavefrontBuffers wf;
    bool last_wavefront = false;
    uint32_t last_rounds = 0;

    rt::Counters* d_counters = nullptr;
    rt::Counters counters = {};

    rt_setup_times_t setup = {};

    hipEvent_t ev_begin[kTimingSlots];
    hipEvent_t ev_end[kTimingSlots];
    uint32_t ev_count = 0;   // launches recorded since the last rt_timing_reset
    uint32_t ev_begin_made = 0, ev_end_made = 0;  // events created so far (rt_destroy frees a partial set too)
    float last_ms = 0.f;

    std::string error;
};

namespace {

int fail(rt_context* ctx, int code, const std::string& msg) {
    if (ctx) ctx->error = msg;
    else g_create_error = msg;
    return code;
}

int fail_hip(rt_context* ctx, hipError_t e, const char* what) {
    return fail(ctx, e == hipErrorOutOfMemory ? RT_ERR_OUT_OF_MEMORY : RT_ERR_HIP,
                std::string(what) + ": " + hipGetErrorString(e));
}

// The C ABI must not change the calling thread's current HIP device (the caller is usually a host framework with
// its own idea of it): every entry point that needs the context's device switches to it through this guard, which
// restores the caller's device on every exit path.
struct DeviceGuard {
    int saved = -1;
    bool ok = true;
    hipError_t err = hipSuccess;
    explicit DeviceGuard(int device) {
        if (hipGetDevice(&saved) != hipSuccess) saved = -1;
        if (saved != device) {
            err = hipSetDevice(device);
            ok = (err == hipSuccess);
        }
    }
    ~DeviceGuard() {
        int now = -1;
        if (saved >= 0 && hipGetDevice(&now) == hipSuccess && now != saved) (void)hipSetDevice(saved);
    }
    DeviceGuard(const DeviceGuard&) = delete;
    DeviceGuard& operator=(const DeviceGuard&) = delete;
};
#define RT_DEVICE(ctx)                                                       \
    DeviceGuard device_guard_((ctx)->device);                                \
    if (!device_guard_.ok) return fail_hip((ctx), device_guard_.err, "hipSetDevice")

#define RT_HIP(ctx, call)                                           \
    do {                                                            \
        hipError_t e_ = (call);                                     \
        if (e_ != hipSuccess) return fail_hip((ctx), e_, #call);    \
    } while (0)

size_t elem_bytes(const rt_context* c) { return c->kernel == RT_KERNEL_HITTEST ? sizeof(float) : 4 * sizeof(float); }

// tiles of the frame that ranks rank .. rank + span - 1 of `world` own (tile t belongs to rank t % world)
uint64_t local_tiles(uint64_t tiles, uint32_t rank, uint32_t world, uint32_t span) {
    const uint64_t rest = tiles % world;
    return (tiles / world) * span + (rest > rank ? std::min<uint64_t>(rest - rank, span) : 0);
}

uint64_t local_count(uint64_t n_rays, uint64_t tile_rays, uint32_t rank, uint32_t world, uint32_t span = 1) {
    if (world <= 1) return n_rays;
    const uint64_t tiles = (n_rays + tile_rays - 1) / tile_rays;
    return local_tiles(tiles, rank, world, span) * tile_rays;  // the last tile may be ragged: its padding work-items write background
}

// traversal stream: objects order[2p] and order[2p+1] (no order: 2p, 2p+1) interleaved entry by entry (rows x,y,z of mvInverse)
void pack_pairs(const rt_object_data* objs, const uint32_t* order, uint32_t n, std::vector<rt::HotPair>& pairs) {
    pairs.assign((n + 1) / 2, rt::HotPair{});
    parallel_for(pairs.size(), 8192, [&](size_t p0, size_t p1) {
        for (size_t p = p0; p < p1; ++p)
            for (uint32_t i = (uint32_t)(2 * p); i < n && i < 2 * p + 2; ++i) {
                rt::HotPair& hp = pairs[p];
                const rt_object_data& o = objs[order ? order[i] : i];
                const float* m = o.mvInverse;
                const int half = (int)(i & 1u);
                for (int r = 0; r < 3; ++r)
                    for (int c = 0; c < 4; ++c) hp.m[4 * r + c][half] = m[4 * c + r];
                (half ? hp.type_b : hp.type_a) = o.type;
            }
    });
    if (n & 1u) pairs[n / 2].type_b = 0xffffffffu;  // odd count: the missing partner can never be hit
}

// ObjectData[] (320 B AoS, as uploaded by the reference) -> hot traversal stream + cold shading records
void repack_objects(const rt_object_data* objs, uint32_t n, std::vector<rt::HotPair>& pairs,
                    std::vector<rt::HotObject>& hot, std::vector<rt::ColdObject>& cold) {
    hot.resize(n);
    cold.resize(n);
    pack_pairs(objs, nullptr, n, pairs);
    parallel_for(n, 16384, [&](size_t i0, size_t i1) {
    for (size_t i = i0; i < i1; ++i) {
        const rt_object_data& o = objs[i];
        const float* m = o.mvInverse;
        rt::HotObject& h = hot[i];
        h.row0 = make_float4(m[0], m[4], m[8], m[12]);
        h.row1 = make_float4(m[1], m[5], m[9], m[13]);
        h.row2 = make_float4(m[2], m[6], m[10], m[14]);
        h.type = o.type;
        h.pad[0] = h.pad[1] = h.pad[2] = 0;
        if (o.type == 2u) {
            // triangle (extension): mv columns 0..2 = vertices, mvInverse column 0 = guard
            // sphere; the edges are single fp32 subtractions, exactly what the CPU statement computes per ray
            const float* v = o.mv;
            const volatile float e1x = v[4] - v[0], e1y = v[5] - v[1], e1z = v[6] - v[2];
            const volatile float e2x = v[8] - v[0], e2y = v[9] - v[1], e2z = v[10] - v[2];
            h.row0 = make_float4(v[0], v[1], v[2], m[0]);
            h.row1 = make_float4(e1x, e1y, e1z, m[1]);
            h.row2 = make_float4(e2x, e2y, e2z, m[2]);
            std::memcpy(&h.pad[0], &m[3], 4);
        }
        rt::ColdObject& c = cold[i];
        for (int r = 0; r < 4; ++r) c.mv_row[r] = make_float4(o.mv[r], o.mv[4 + r], o.mv[8 + r], o.mv[12 + r]);
        c.inv_row3 = make_float4(m[3], m[7], m[11], m[15]);
        c.amb_absorb = make_float4(o.mat.ambient[0], o.mat.ambient[1], o.mat.ambient[2], o.mat.absorption);
        c.dif_shine = make_float4(o.mat.diffuse[0], o.mat.diffuse[1], o.mat.diffuse[2], o.mat.shininess);
        float type_bits;
        std::memcpy(&type_bits, &o.type, 4);
        c.spec_type = make_float4(o.mat.specular[0], o.mat.specular[1], o.mat.specular[2], type_bits);
    }
    });
}

// View-space bounding sphere of what the traversal tests for object o: { x : |A x + b| <= r0 } with A, b the
// rows x,y,z of mvInverse (the kernels never consult mv for intersection) -> centre -A^-1 b, radius
// r0 * sigma_max(A^-1) <= r0 * |A^-1|_F. Computed in double, then inflated:
//   R_eff = R * (1 + 2^-9) + |c| * 2^-9
// which covers (a) the reference's own rounding: its discriminant accepts rays that pass a sphere at up to
// sqrt(1 + ~1e-6 (|c|/R)^2) radii, (b) the fp32 rounding of the bundle test. Anything doubtful (singular or
// non-finite matrices) gets +inf = never culled; unknown primitive types can never be hit = -inf.
struct Sphere { double x, y, z, r; };
// Bounding sphere of an instanced unit sphere / unit box in view space, in double precision and WITHOUT safety
// margins (callers add the ones their use needs): centre -A^-1 b, radius r0 * sigma_max(A^-1) (r0 = 1 or
// sqrt(0.75)), plus an upper bound of the squared condition number kappa^2 = (sigma_max / sigma_min)^2.
// r = +inf: no usable bound (test it for every ray); r = -inf: unknown type, can never be hit.
struct Bound { double x, y, z, r, kappa2; };
Bound object_bound(const rt_object_data& o) {
    const double inf = std::numeric_limits<double>::infinity();
#define make_float4(X, Y, Z, R) Bound{(double)(X), (double)(Y), (double)(Z), (double)(R), 1.0}
    if (o.type == 2u) {  // triangle: the record's guard sphere bounds every hit (the test itself demands it)
        const float* gs = o.mvInverse;
        if (!std::isfinite(gs[0] + gs[1] + gs[2]) || !(gs[3] >= 0.f)) return make_float4(0.f, 0.f, 0.f, -inf);  // never passes its guard
        if (!std::isfinite(gs[3])) return make_float4(0.f, 0.f, 0.f, inf);
        return make_float4(gs[0], gs[1], gs[2], gs[3]);
    }
    if (o.type > 2u) return make_float4(0.f, 0.f, 0.f, -inf);
    const float* m = o.mvInverse;
    double A[3][3] = {{m[0], m[4], m[8]}, {m[1], m[5], m[9]}, {m[2], m[6], m[10]}};
    const double b[3] = {m[12], m[13], m[14]};
    for (int i = 0; i < 3; ++i)
        for (int j = 0; j < 3; ++j)
            if (!std::isfinite(A[i][j]) || !std::isfinite(b[i])) return make_float4(0.f, 0.f, 0.f, inf);
    const double det = A[0][0] * (A[1][1] * A[2][2] - A[1][2] * A[2][1]) - A[0][1] * (A[1][0] * A[2][2] - A[1][2] * A[2][0]) +
                       A[0][2] * (A[1][0] * A[2][1] - A[1][1] * A[2][0]);
    double norm2 = 0;
    for (int i = 0; i < 3; ++i)
        for (int j = 0; j < 3; ++j) norm2 += A[i][j] * A[i][j];
    if (!(std::fabs(det) > 1e-12 * std::pow(norm2, 1.5))) return make_float4(0.f, 0.f, 0.f, inf);
    double inv[3][3];
    inv[0][0] = (A[1][1] * A[2][2] - A[1][2] * A[2][1]) / det;
    inv[0][1] = (A[0][2] * A[2][1] - A[0][1] * A[2][2]) / det;
    inv[0][2] = (A[0][1] * A[1][2] - A[0][2] * A[1][1]) / det;
    inv[1][0] = (A[1][2] * A[2][0] - A[1][0] * A[2][2]) / det;
    inv[1][1] = (A[0][0] * A[2][2] - A[0][2] * A[2][0]) / det;
    inv[1][2] = (A[0][2] * A[1][0] - A[0][0] * A[1][2]) / det;
    inv[2][0] = (A[1][0] * A[2][1] - A[1][1] * A[2][0]) / det;
    inv[2][1] = (A[0][1] * A[2][0] - A[0][0] * A[2][1]) / det;
    inv[2][2] = (A[0][0] * A[1][1] - A[0][1] * A[1][0]) / det;
    double c[3], fro2 = 0;
    for (int i = 0; i < 3; ++i) {
        c[i] = -(inv[i][0] * b[0] + inv[i][1] * b[1] + inv[i][2] * b[2]);
        for (int j = 0; j < 3; ++j) fro2 += inv[i][j] * inv[i][j];
    }
    // sigma_max(A^-1)^2 = largest eigenvalue of S = A^-1 A^-T (closed form for a symmetric 3x3; double precision,
    // then padded by 1e-6 relative and never allowed above the Frobenius bound or below a third of it)
    double S[3][3];
    for (int i = 0; i < 3; ++i)
        for (int j = 0; j < 3; ++j) S[i][j] = inv[i][0] * inv[j][0] + inv[i][1] * inv[j][1] + inv[i][2] * inv[j][2];
    double lam_max = fro2;
    {
        const double q = (S[0][0] + S[1][1] + S[2][2]) / 3.0;
        const double p1 = S[0][1] * S[0][1] + S[0][2] * S[0][2] + S[1][2] * S[1][2];
        const double p2 = (S[0][0] - q) * (S[0][0] - q) + (S[1][1] - q) * (S[1][1] - q) + (S[2][2] - q) * (S[2][2] - q) + 2.0 * p1;
        const double pp = std::sqrt(p2 / 6.0);
        if (pp > 0 && std::isfinite(pp)) {
            double Bm[3][3];
            for (int i = 0; i < 3; ++i)
                for (int j = 0; j < 3; ++j) Bm[i][j] = (S[i][j] - (i == j ? q : 0.0)) / pp;
            double r = (Bm[0][0] * (Bm[1][1] * Bm[2][2] - Bm[1][2] * Bm[2][1]) - Bm[0][1] * (Bm[1][0] * Bm[2][2] - Bm[1][2] * Bm[2][0]) +
                        Bm[0][2] * (Bm[1][0] * Bm[2][1] - Bm[1][1] * Bm[2][0])) / 2.0;
            r = r < -1.0 ? -1.0 : (r > 1.0 ? 1.0 : r);
            const double lam = q + 2.0 * pp * std::cos(std::acos(r) / 3.0);
            if (std::isfinite(lam) && lam > 0) lam_max = lam * (1.0 + 1e-6);
        } else if (pp == 0) {
            lam_max = q * (1.0 + 1e-6);  // S is a multiple of the identity
        }
        if (lam_max > fro2) lam_max = fro2;
        if (lam_max < fro2 / 3.0) lam_max = fro2 / 3.0;  // lambda_max >= trace / 3 always holds
    }
    const double r0 = (o.type == 0u) ? 1.0 : std::sqrt(0.75);
    const double R = r0 * std::sqrt(lam_max);
    // lambda_min >= det(S) / lambda_max^2 (the other two eigenvalues are <= lambda_max), det(S) = 1 / det(A)^2
    const double kappa2 = lam_max * lam_max * lam_max * det * det * (1.0 + 1e-6);
    if (!std::isfinite(R) || !std::isfinite(kappa2) || !(kappa2 >= 0.999) || !std::isfinite(c[0] + c[1] + c[2]))
        return make_float4(0.f, 0.f, 0.f, inf);
    Bound bd{c[0], c[1], c[2], R, kappa2 < 1.0 ? 1.0 : kappa2};
    return bd;
#undef make_float4
}

// the small-scene screen culling keeps its original, generous margins: 2^-9 of the radius and of the distance
Sphere bounding_sphere(const rt_object_data& o) {
    const Bound b = object_bound(o);
    if (!std::isfinite(b.r)) return Sphere{0.0, 0.0, 0.0, b.r};
    const double cl = std::sqrt(b.x * b.x + b.y * b.y + b.z * b.z);
    const double Reff = b.r * (1.0 + 1.0 / 512.0) + cl / 512.0;
    if (!std::isfinite(Reff)) return Sphere{0.0, 0.0, 0.0, std::numeric_limits<double>::infinity()};
    return Sphere{b.x, b.y, b.z, Reff};
}

// Conservative projection of a bounding sphere onto the pinhole image plane, in ray-direction units
// (direction = (x, y, z), z < 0 fixed): [xmin, xmax] from the two tangent planes that contain the camera's y
// axis, [ymin, ymax] likewise. Unbounded when the sphere reaches the plane z = 0 through the camera; empty when
// it lies entirely behind it. Padded by one pixel plus 1e-6 relative before rounding outwards to float.
float4 screen_rect(const Sphere& s, double z) {
    const float inf = std::numeric_limits<float>::infinity();
    const float4 all = make_float4(-inf, inf, -inf, inf), none = make_float4(inf, -inf, inf, -inf);
    if (s.r == -std::numeric_limits<double>::infinity()) return none;
    if (!std::isfinite(s.r) || !(z < 0)) return all;
    if (s.z - s.r >= 0) return none;          // entirely behind the camera: every root is negative
    if (s.z + s.r >= 0) return all;           // reaches the camera plane: silhouette unbounded
    auto extent = [&](double cu, float& lo, float& hi) {
        // tangent planes through the origin containing the other image axis: (z cu - u cz)^2 = R^2 (u^2 + z^2)
        const double a = s.z * s.z - s.r * s.r, b = -2.0 * z * cu * s.z, c = z * z * (cu * cu - s.r * s.r);
        const double disc = b * b - 4.0 * a * c;
        if (!(a > 0) || !(disc >= 0)) { lo = -inf; hi = inf; return; }
        const double sq = std::sqrt(disc);
        double u0 = (-b - sq) / (2.0 * a), u1 = (-b + sq) / (2.0 * a);
        if (u0 > u1) std::swap(u0, u1);
        u0 -= 1.0 + 1e-6 * std::fabs(u0);
        u1 += 1.0 + 1e-6 * std::fabs(u1);
        lo = std::nextafter((float)u0, -inf);
        hi = std::nextafter((float)u1, inf);
    };
    float4 r;
    extent(s.x, r.x, r.y);
    extent(s.y, r.z, r.w);
    return r;
}

// Is the uploaded ray list bit-for-bit the reference's pinhole grid (OpenCL-Raytracer.cpp:18-26,68-72)?
bool detect_pinhole(const rt_ray* rays, uint64_t n, uint32_t& W, uint32_t& H, float& z) {
    if (n == 0 || n > 0xffffffffull) return false;
    const float y0 = rays[0].direction[1];
    uint64_t w = n;
    for (uint64_t i = 1; i < n; ++i) {
        if (rays[i].direction[1] != y0) { w = i; break; }
    }
    if (w == 0 || n % w != 0) return false;
    const uint64_t h = n / w;
    if (w > 0x1000000ull || h > 0x1000000ull) return false;  // exact integer -> float conversion range
    const float zz = rays[0].direction[2];
    const float half_w = (float)w / 2.0f, half_h = (float)h / 2.0f, hf = (float)h;
    for (uint64_t j = 0; j < h; ++j) {
        const float dy = (hf - (float)j) - half_h;
        const rt_ray* row = rays + j * w;
        for (uint64_t i = 0; i < w; ++i) {
            const rt_ray& r = row[i];
            const float expect[8] = {0.f, 0.f, 0.f, 1.f, (float)i - half_w, dy, zz, 0.f};
            if (std::memcmp(&r, expect, sizeof(expect)) != 0) return false;
        }
    }
    W = (uint32_t)w;
    H = (uint32_t)h;
    z = zz;
    return true;
}

int ensure_out(rt_context* c) {
    const size_t need = (size_t)c->n_local * elem_bytes(c);
    if (need > c->d_out_bytes) {
        if (c->d_out) (void)hipFree(c->d_out);
        c->d_out = nullptr;
        c->d_out_bytes = 0;
        RT_HIP(c, hipMalloc(&c->d_out, need ? need : 16));
        c->d_out_bytes = need;
    }
    return RT_OK;
}

int ensure_host_out(rt_context* c) {
    const size_t need = (size_t)c->n_local * elem_bytes(c);
    if (need > c->h_out_bytes) {
        if (c->h_out) (void)hipHostFree(c->h_out);
        c->h_out = nullptr;
        c->h_out_bytes = 0;
        RT_HIP(c, hipHostMalloc(&c->h_out, need ? need : 16, hipHostMallocDefault));
        c->h_out_bytes = need;
    }
    return RT_OK;
}

// Path choice unless a flag says otherwise. Measured at 2048^2 (scratch sweep, depth 3, 4 lights): the small-scene
// kernel wins up to 64 objects (per-bundle culling), the wavefront path with the grid from ~100 objects on
// (N=128: 1.4 vs 0.9 ms, N=512: 7.4 vs 1.3 ms). Without a usable grid the wavefront path only pays once the
// traversal loop dwarfs its per-round state traffic.
constexpr uint32_t kWavefrontMinObjects = 512;      // brute-force wavefront
constexpr uint32_t kWavefrontGridMinObjects = 96;   // wavefront when the conservative grid is available

// |d|^2 exactly as the walks compute it (fp32, unfused, left to right) against their `tame` window
bool direction_in_domain(float dx, float dy, float dz) {
    const volatile float xx = dx * dx, yy = dy * dy, zz = dz * dz;
    const volatile float s1 = xx + yy;
    const float dd = s1 + zz;
    return dd > 1.0e-30f && dd < 1.0e30f;
}
// a pinhole grid's directions are (col - W/2, (H - row) - H/2, z): the shortest belongs to the centre pixel, the longest to a corner
bool camera_in_domain(uint32_t W, uint32_t H, float z) {
    const double zz = (double)z * (double)z;
    const double lo = zz + ((W & 1u) ? 0.25 : 0.0) + ((H & 1u) ? 0.25 : 0.0);
    const double hi = zz + 0.25 * (double)W * (double)W + 0.25 * (double)H * (double)H;
    return std::isfinite(zz) && lo > 1.0e-29 && hi < 1.0e29;  // (a decade inside the walks' window: fp32 rounding of the sum)
}
void apply_ray_domain(rt_context* c) {
    const bool out = c->pinhole ? c->camera_out_of_domain : c->rays_out_of_domain;
    c->flags = c->base_flags | (out ? RT_FLAG_LITERAL : 0u);
}

bool use_wavefront(const rt_context* c) {
    if (c->has_triangles) return true;
    if (c->flags & RT_FLAG_WAVEFRONT) return true;
    if (c->flags & RT_FLAG_MONOLITHIC) return false;
    if (c->n_objs >= kWavefrontMinObjects) return true;
    return c->n_objs >= kWavefrontGridMinObjects && c->grid.enabled && !(c->flags & RT_FLAG_LITERAL);
}

void free_wavefront(rt_context* c) {
    rt::WavefrontBuffers& b = c->wf;
    if (b.side_stream) (void)hipStreamDestroy(b.side_stream);
    if (b.ev_fork) (void)hipEventDestroy(b.ev_fork);
    if (b.ev_join) (void)hipEventDestroy(b.ev_join);
    b.side_stream = nullptr; b.ev_fork = b.ev_join = nullptr;
    if (b.state) (void)hipFree(b.state);
    for (int i = 0; i < 2; ++i) {
        if (b.q_closest[i]) (void)hipFree(b.q_closest[i]);
        if (b.q_any[i]) (void)hipFree(b.q_any[i]);
        if (b.q_slice[i]) (void)hipFree(b.q_slice[i]);
    }
    if (b.counts) (void)hipFree(b.counts);
    if (b.h_counts) (void)hipHostFree(b.h_counts);
    b = rt::WavefrontBuffers{};
}

// size proxy for ordering the shadow stream: r0 * |A^-1|_F (an upper bound of the bounding radius)
double size_proxy(const rt_object_data& o) {
    if (o.type > 1u) return -1.0;
    const float* m = o.mvInverse;
    const double A[3][3] = {{m[0], m[4], m[8]}, {m[1], m[5], m[9]}, {m[2], m[6], m[10]}};
    const double det = A[0][0] * (A[1][1] * A[2][2] - A[1][2] * A[2][1]) - A[0][1] * (A[1][0] * A[2][2] - A[1][2] * A[2][0]) +
                       A[0][2] * (A[1][0] * A[2][1] - A[1][1] * A[2][0]);
    if (!(std::fabs(det) > 0) || !std::isfinite(det)) return 1e300;  // degenerate: test it first
    double cof2 = 0;  // |adj(A)|_F^2 ; A^-1 = adj / det
    const int nx[3] = {1, 2, 0}, pv[3] = {2, 0, 1};
    for (int i = 0; i < 3; ++i)
        for (int j = 0; j < 3; ++j) {
            const double cf = A[nx[i]][nx[j]] * A[pv[i]][pv[j]] - A[nx[i]][pv[j]] * A[pv[i]][nx[j]];
            cof2 += cf * cf;
        }
    const double r = std::sqrt(cof2) / std::fabs(det) * (o.type == 0u ? 1.0 : 0.8660254037844386);
    return std::isfinite(r) ? r : 1e300;
}

int ensure_wavefront(rt_context* c) {
    rt::WavefrontBuffers& b = c->wf;
    if (b.capacity >= c->n_local && b.state) return RT_OK;
    free_wavefront(c);
    const uint64_t n = c->n_local ? c->n_local : 1;
    RT_HIP(c, hipMalloc((void**)&b.state, rt::wavefront_state_bytes(n)));
    for (int i = 0; i < 2; ++i) {
        RT_HIP(c, hipMalloc((void**)&b.q_closest[i], rt::wavefront_queue_bytes(n)));
        RT_HIP(c, hipMalloc((void**)&b.q_any[i], rt::wavefront_queue_bytes(n)));
        RT_HIP(c, hipMalloc((void**)&b.q_slice[i], rt::wavefront_queue_bytes(n)));
    }
    RT_HIP(c, hipMalloc((void**)&b.counts, rt::wavefront_counter_bytes()));  // device-side round state + run-ticket counters
    RT_HIP(c, hipHostMalloc((void**)&b.h_counts, 16 * sizeof(uint32_t), hipHostMallocDefault));
    RT_HIP(c, hipStreamCreateWithFlags(&b.side_stream, hipStreamNonBlocking));
    RT_HIP(c, hipEventCreateWithFlags(&b.ev_fork, hipEventDisableTiming));
    RT_HIP(c, hipEventCreateWithFlags(&b.ev_join, hipEventDisableTiming));
    b.shadow_pairs = c->d_shadow_pairs;
    b.grid = c->grid;
    b.light_tiles = c->light_tiles;
    b.blocks = c->blocks;
    b.capacity = n;
    return RT_OK;
}

int build_screen_tiles(rt_context* c, hipStream_t stream, uint32_t col_shift);

int do_launch(rt_context* c, void* d_out, hipStream_t stream, bool count) {
    if (c->n_local == 0) {  // empty launch: nothing to render, nothing to time
        if (count) c->counters = rt::Counters{};
        c->aux_t = nullptr;
        c->aux_index = nullptr;
        return RT_OK;
    }
    if (!c->pinhole && !c->have_rays)
        return fail(c, RT_ERR_STATE, "no primary rays: rt_create got rays == NULL and rt_set_camera was not called");
    rt::RenderParams p;
    std::memset(&p, 0, sizeof(p));
    p.scene.pairs = c->d_pairs;
    p.scene.n_pairs = c->n_pairs;
    p.scene.hot = c->d_hot;
    p.scene.bounds = c->d_bounds;
    p.scene.cold = c->d_cold;
    p.scene.objrec = c->d_objrec;
    p.scene.lights = c->d_lights;
    p.scene.n_objs = c->n_objs;
    p.scene.n_lights = c->n_lights;
    p.scene.literal = (c->flags & RT_FLAG_LITERAL) ? 1u : 0u;
    p.scene.fast_phong = (c->flags & RT_FLAG_FAST_PHONG) ? 1u : 0u;
    p.scene.affine = (c->affine_w && !c->has_triangles) ? 1u : 0u;
    p.scene.nan_winner = c->nan_winner;
    p.scene.nan_winner_sphere = c->nan_winner_sphere ? 1u : 0u;
    p.rays = c->pinhole ? nullptr : c->d_rays;
    p.n_rays = c->n_rays;
    p.n_local = c->n_local;
    p.tile_rays = c->tile_rays ? c->tile_rays : 1;
    p.run_rays = p.tile_rays * c->span;
    p.rank = c->rank;
    p.world = c->world;
    p.pinhole = c->pinhole ? 1u : 0u;
    p.width = c->width ? c->width : 1;
    p.half_w = (float)c->width / 2.0f;
    p.half_h = (float)c->height / 2.0f;
    p.height_f = (float)c->height;
    p.z = c->z;
    p.dir_w_zero = (c->pinhole || c->dir_w_zero) ? 1u : 0u;
    // 2-D pixel bundles need the grid shape (pinhole mode) and shard tiles made of whole 8-row bands
    const bool row_tiles = c->world <= 1 || (c->width && c->tile_rays % c->width == 0 && (c->tile_rays / c->width) % 8 == 0);
    if (c->pinhole && row_tiles && c->n_local % c->width == 0 && c->n_local < 0x7fffffffull) {
        p.tile2d = 1u;
        p.bundles_x = (c->width + 7u) / 8u;
        p.local_rows = (uint32_t)(c->n_local / c->width);
        p.tile_rows = c->world > 1 ? (uint32_t)(c->tile_rays / c->width) : p.local_rows;
        if (p.tile_rows == 0) p.tile_rows = 1;
        p.n_bundles = p.bundles_x * ((p.local_rows + 7u) / 8u);
        p.wf_tile_order = (c->width % 8u == 0 && p.local_rows % 8u == 0) ? 1u : 0u;
        p.tile_cull = (c->n_objs > 0 && c->n_objs <= 64 && c->z < 0.0f && !(c->flags & RT_FLAG_LITERAL)) ? 1u : 0u;
    } else {
        if (c->n_local > 0xffffffffull - 64) return fail(c, RT_ERR_INVALID_ARGUMENT, "too many rays for one launch");
        p.n_bundles = (uint32_t)((c->n_local + 63u) / 64u);
    }
    p.max_bounces = c->max_bounces;
    p.out = d_out;
    p.aux_t = c->aux_t;
    p.aux_index = c->aux_index;
    p.counters = c->d_counters;

    RT_DEVICE(c);
    if (p.tile_cull && c->rects_dirty) {  // screen rectangles of the bounding spheres for this camera
        std::vector<float4> rects(c->n_objs);
        for (uint32_t i = 0; i < c->n_objs; ++i)
            rects[i] = screen_rect(Sphere{c->h_spheres[4 * i], c->h_spheres[4 * i + 1], c->h_spheres[4 * i + 2], c->h_spheres[4 * i + 3]},
                                   (double)c->z);
        RT_HIP(c, hipMemcpyAsync(c->d_bounds, rects.data(), sizeof(float4) * c->n_objs, hipMemcpyHostToDevice, stream));
        RT_HIP(c, hipStreamSynchronize(stream));  // `rects` is pageable host memory
        c->rects_dirty = false;
    }
    c->last_wavefront = use_wavefront(c);
    c->last_rounds = 0;
    if (c->last_wavefront) {  // one-time host-side set-up (buffers, per-camera screen tiles) stays outside the timed region
        StopWatch sw;
        const bool had_buffers = c->wf.capacity >= c->n_local && c->wf.state;
        int rc = ensure_wavefront(c);
        if (rc) return rc;
        if (!had_buffers) c->setup.buffers_ms += sw.lap_ms();
        // a first-round wave is an 8 x 8 block of pixels (work-items in tile order) or 64 pixels of one row: the tile lists follow
        const uint32_t col_shift = p.wf_tile_order ? 3u : 6u;
        if (c->tiles_dirty || c->tiles_built_for != col_shift) {
            rc = build_screen_tiles(c, stream, col_shift);
            if (rc) return rc;
            // objects that cover much of the screen can exceed the pair budget at 8 x 8: the 64 x 8 tiles of round 1 serve an
            // 8 x 8 wave as well (its block lies inside one of them)
            if (!c->tiles.enabled && col_shift == 3u) {
                rc = build_screen_tiles(c, stream, 6u);
                if (rc) return rc;
            }
            c->tiles_built_for = col_shift;
            RT_HIP(c, hipStreamSynchronize(stream));
            c->setup.screen_tiles_ms += sw.lap_ms();
        }
        c->wf.tiles = c->tiles;
    }
    if (count) RT_HIP(c, hipMemsetAsync(c->d_counters, 0, sizeof(rt::Counters), stream));
    const uint32_t slot = c->ev_count % kTimingSlots;
    RT_HIP(c, hipEventRecord(c->ev_begin[slot], stream));
    hipError_t e;
    if (c->last_wavefront) e = rt::launch_wavefront(p, c->kernel, !(c->flags & RT_FLAG_UNFUSED), count, c->wf, stream, &c->last_rounds);
    else e = rt::launch_render(p, c->kernel, !(c->flags & RT_FLAG_UNFUSED), count, stream);
    if (e != hipSuccess) return fail_hip(c, e, "kernel launch");
    RT_HIP(c, hipEventRecord(c->ev_end[slot], stream));
    c->ev_count += 1;
    // aux buffers apply to one render only
    c->aux_t = nullptr;
    c->aux_index = nullptr;
    return RT_OK;
}

// Primary rays of a pinhole grid: per screen tile the objects whose conservative screen rectangle (projection of the grid
// sphere, i.e. with the same error-bound inflation) overlaps the tile, ascending index. A wave of the first trace round
// holds the 64 pixels of ONE tile - an 8 x 8 block (col_shift 3) when the work-items walk the frame in such blocks, else 64
// consecutive pixels of a row inside a 64 x 8 tile (col_shift 6) - so it walks that list with wave-uniform scalar loads
// instead of 64 separate grid walks. (Round 2: 8 x 8 tiles instead of 64 x 8 wherever the order allows - a wave no longer
// tests what only the seven other blocks of its 64 x 8 tile can see.)
int build_screen_tiles(rt_context* c, hipStream_t stream, uint32_t col_shift) {
    c->tiles = rt::ScreenTiles{};
    c->tiles_dirty = false;
    const uint32_t tile_w = 1u << col_shift;
    if (!c->grid.enabled || !c->pinhole || !(c->z < 0.f) || c->width % tile_w != 0 || c->h_grid_spheres.empty()) return RT_OK;
    const uint32_t tx = c->width / tile_w, ty = (c->height + 7u) / 8u;
    const size_t n_tiles = (size_t)tx * ty;
    const uint32_t n = c->n_objs;
    const double half_w = (double)((float)c->width / 2.0f), half_h = (double)((float)c->height / 2.0f), H = (double)c->height;
    const double inf = std::numeric_limits<double>::infinity();
    std::vector<uint32_t> start(n_tiles + 1, 0), entries, fill, global;
    struct Range { int x0, x1, y0, y1; };
    std::vector<Range> rng(n);
    // (object, tile) pairs are counted in 64 bits against the budget BEFORE any per-tile loop runs: an object whose
    // sphere reaches the camera plane projects onto the whole screen (131 072 tiles at 8192^2), and a few ten
    // thousand of those would wrap a 32-bit prefix sum. Such objects go to a per-camera global list that every tile
    // wave tests (at most kMaxGlobal of them; beyond that the grid walk is the better tool for primary rays too).
    constexpr size_t kMaxGlobal = 64;
    const uint64_t budget = 256ull * n + 4096ull;
    uint64_t total = 0;
    // every object's tile rectangle (the expensive part: screen_rect), on several threads; what depends on the order - the budget,
    // the global list, the counts - in a second, serial sweep
    parallel_for(n, 8192, [&](size_t i0, size_t i1) {
        for (size_t i = i0; i < i1; ++i) {
            const double r = c->h_grid_spheres[4 * i + 3];
            Range& q = rng[i];
            q.x0 = 0; q.x1 = -1; q.y0 = 0; q.y1 = -1;
            if (!(r >= 0) || r == inf) continue;  // never hit / always-list (handled by the kernel)
            const float4 rect = screen_rect(Sphere{c->h_grid_spheres[4 * i], c->h_grid_spheres[4 * i + 1], c->h_grid_spheres[4 * i + 2], r}, (double)c->z);
            if (!(rect.x <= rect.y) || !(rect.z <= rect.w)) continue;  // empty: behind the camera
            // direction x = col - W/2  ->  col range; direction y = (H - row) - H/2  ->  row range
            const double c0 = (double)rect.x + half_w, c1 = (double)rect.y + half_w;
            const double r0 = H - half_h - (double)rect.w, r1 = H - half_h - (double)rect.z;
            const double cx0 = std::max(0.0, std::floor(c0)), cx1 = std::min((double)c->width - 1, std::ceil(c1));
            const double ry0 = std::max(0.0, std::floor(r0)), ry1 = std::min((double)c->height - 1, std::ceil(r1));
            if (cx0 > cx1 || ry0 > ry1) continue;
            q.x0 = (int)(cx0 / tile_w); q.x1 = (int)(cx1 / tile_w); q.y0 = (int)(ry0 / 8); q.y1 = (int)(ry1 / 8);
        }
    });
    for (uint32_t i = 0; i < n; ++i) {
        Range& q = rng[i];
        if (q.x1 < q.x0 || q.y1 < q.y0) continue;
        const uint64_t covered = (uint64_t)(q.x1 - q.x0 + 1) * (uint64_t)(q.y1 - q.y0 + 1);
        if (covered == (uint64_t)n_tiles && n_tiles > 1) {  // the whole screen
            if (global.size() >= kMaxGlobal) return RT_OK;
            global.push_back(i);
            q.x0 = 0; q.x1 = -1; q.y0 = 0; q.y1 = -1;
            continue;
        }
        total += covered;
        if (total > budget) return RT_OK;  // objects cover most of the screen: the grid walk is the better tool
        for (int y = q.y0; y <= q.y1; ++y)
            for (int x = q.x0; x <= q.x1; ++x) start[(size_t)y * tx + x + 1] += 1;
    }
    for (size_t k = 0; k < n_tiles; ++k) start[k + 1] += start[k];  // total <= budget < 2^32 (n_objs is a uint32, budget clamps below)
    if (total > 0xfffffff0ull) return RT_OK;
    entries.assign((size_t)total + global.size(), 0);
    fill.assign(start.begin(), start.end() - 1);
    for (uint32_t i = 0; i < n; ++i) {
        const Range& q = rng[i];
        for (int y = q.y0; y <= q.y1; ++y)
            for (int x = q.x0; x <= q.x1; ++x) entries[fill[(size_t)y * tx + x]++] = i;
    }
    for (size_t k = 0; k < n_tiles; ++k)
        if (fill[k] != start[k + 1]) return fail(c, RT_ERR_STATE, "internal: screen-tile fill does not match its count");
    for (size_t k = 0; k < global.size(); ++k) entries[(size_t)total + k] = global[k];  // the global list sits behind the last tile's
    if (c->d_tile_start) (void)hipFree(c->d_tile_start);
    if (c->d_tile_entries) (void)hipFree(c->d_tile_entries);
    c->d_tile_start = c->d_tile_entries = nullptr;
    RT_HIP(c, hipMalloc((void**)&c->d_tile_start, sizeof(uint32_t) * (n_tiles + 1)));
    RT_HIP(c, hipMalloc((void**)&c->d_tile_entries, sizeof(uint32_t) * (entries.size() + 1)));
    RT_HIP(c, hipMemcpyAsync(c->d_tile_start, start.data(), sizeof(uint32_t) * (n_tiles + 1), hipMemcpyHostToDevice, stream));
    if (!entries.empty()) RT_HIP(c, hipMemcpyAsync(c->d_tile_entries, entries.data(), sizeof(uint32_t) * entries.size(), hipMemcpyHostToDevice, stream));
    RT_HIP(c, hipStreamSynchronize(stream));
    c->tiles.tile_start = c->d_tile_start;
    c->tiles.entries = c->d_tile_entries;
    c->tiles.tiles_x = tx;
    c->tiles.col_shift = col_shift;
    c->tiles.global_begin = (uint32_t)total;
    c->tiles.n_global = (uint32_t)global.size();
    c->tiles.enabled = 1u;
    return RT_OK;
}

// The unified walk's record table (rt_grid.h: GridDesc::walk_rec), host side. build_grid puts down one head per cell of
// the grid padded by two empty cells on every side, then the 2nd, 3rd ... entries of every cell; build_light_tiles appends
// the light tiles' entries; upload_walk_records ships the table (or drops it when it would not fit 32-bit byte offsets).
constexpr uint32_t kWalkBorder = 2;
inline float4 walk_sphere(const float4& es) {
    const volatile float w = es.w;
    const volatile float w2 = w * w;  // the fp32 product the pre-test used to form per trip
    return make_float4(es.x, es.y, es.z, w2);
}
inline float4 walk_link(uint32_t object, uint32_t next, float key) {
    float4 r;
    std::memcpy(&r.x, &object, 4);
    std::memcpy(&r.y, &next, 4);
    r.z = key;
    r.w = 0.f;
    return r;
}
void build_walk_records(rt_context* c, const int dim[3], const std::vector<uint2>& ranges, const std::vector<float4>& es,
                        const std::vector<uint32_t>& entries) {
    c->h_walk.clear();
    c->grid.walk_rec = nullptr;
    c->grid.walk_cells = 0;
    if (std::getenv("RT_NO_WALK2")) return;  // measurement knob: the round-2 walk
    const uint64_t wnx = (uint64_t)dim[0] + 2 * kWalkBorder, wny = (uint64_t)dim[1] + 2 * kWalkBorder, wnz = (uint64_t)dim[2] + 2 * kWalkBorder;
    const uint64_t cells = wnx * wny * wnz;
    uint64_t overflow = 0;
    for (const uint2& r : ranges) overflow += r.y > 1u ? r.y - 1u : 0u;
    if ((cells + overflow) * 32ull >= 0xf0000000ull) return;
    const float ninf = -std::numeric_limits<float>::infinity();
    const uint32_t none = c->n_objs;
    std::vector<float4>& rec = c->h_walk;
    rec.assign(2 * (size_t)(cells + overflow), make_float4(0.f, 0.f, 0.f, 0.f));
    for (size_t k = 0; k < (size_t)cells; ++k) {
        rec[2 * k] = make_float4(0.f, 0.f, 0.f, ninf);
        rec[2 * k + 1] = walk_link(none, 0u, ninf);
    }
    uint64_t next_free = cells;
    for (int z = 0; z < dim[2]; ++z)
        for (int y = 0; y < dim[1]; ++y)
            for (int x = 0; x < dim[0]; ++x) {
                const uint2 r = ranges[((size_t)z * dim[1] + y) * dim[0] + x];
                if (r.y == 0u) continue;
                const size_t head = (size_t)(((uint64_t)(z + kWalkBorder) * wny + (y + kWalkBorder)) * wnx + (x + kWalkBorder));
                rec[2 * head] = walk_sphere(es[r.x]);
                rec[2 * head + 1] = walk_link(entries[r.x], r.y > 1u ? (uint32_t)next_free : 0u, ninf);
                for (uint32_t j = 1; j < r.y; ++j) {
                    const size_t at = (size_t)next_free++;
                    rec[2 * at] = walk_sphere(es[r.x + j]);
                    rec[2 * at + 1] = walk_link(entries[r.x + j], j + 1u < r.y ? (uint32_t)next_free : 0u, ninf);
                }
            }
    c->grid.walk_cells = (uint32_t)cells;
    c->grid.walk_nx = (uint32_t)wnx;
    c->grid.walk_nxy = (uint32_t)(wnx * wny);
    c->grid.walk_none = none;
}

int upload_walk_records(rt_context* c) {
    if (c->h_walk.empty() || (uint64_t)c->h_walk.size() * 16ull >= 0xfffffff0ull) {
        std::vector<float4>().swap(c->h_walk);
        c->grid.walk_rec = nullptr;
        c->grid.walk_cells = 0;
        c->light_tiles.walk_base = 0;
        return RT_OK;
    }
    RT_HIP(c, hipMalloc((void**)&c->d_walk_rec, sizeof(float4) * c->h_walk.size()));
    RT_HIP(c, hipMemcpy(c->d_walk_rec, c->h_walk.data(), sizeof(float4) * c->h_walk.size(), hipMemcpyHostToDevice));
    c->grid.walk_rec = c->d_walk_rec;
    std::vector<float4>().swap(c->h_walk);
    return RT_OK;
}

// The closest-hit walk's coarse grid of 32-byte blocks (rt_grid.h: BlockGrid). Same box and same objects as the fine grid;
// an object sits in every block-cell its registration sphere reaches (the fine grid's radius with the walk-arithmetic slack
// re-sized to this grid's cell). Every entry's sphere is rounded OUTWARDS onto its block's lattice - see the struct's comment;
// the derivation of the radius: with the centre off by delta (quantisation error, known exactly, + the rounding of the
// ray's transform into lattice coordinates), "the line passes c within sqrt(w^2 + a |oc|^2)" (a = 6e-6 K^2, the pre-test's
// distance term) implies "it passes c' within sqrt(w'^2 + a |oc'|^2)" for
//     w'^2 = (w + delta)^2 + 2 delta sqrt(a) D + a (2 delta D + delta^2),     D = the largest possible |oc|,
// and a sphere that contains the original can only be "entirely behind the origin" if the original is.
int build_walk_blocks(rt_context* c, uint32_t n, const std::vector<Bound>& sph, const std::vector<double>& rg, double cell_fine,
                      const double glo[3], const double ghi[3], double K2) {
    c->blocks = rt::BlockGrid{};
    SetupTrace lap("block grid");
    if (std::getenv("RT_NO_WALK3") || c->h_grid_pre.size() != n) return RT_OK;  // (measurement knob: the record walk)
    double factor = 1.67;
    if (const char* env = std::getenv("RT_WALK_BLOCK_FACTOR")) {  // tuning knob (results do not depend on it)
        const double v = std::atof(env);
        if (v >= 0.25 && v <= 16.0) factor = v;
    }
    const double cell = cell_fine * factor;
    const float cellf = (float)cell;
    const float lof[3] = {(float)glo[0], (float)glo[1], (float)glo[2]};
    const int B = (int)rt::kBlockBorder;
    int dim[3];
    uint64_t wn[3];
    for (int a = 0; a < 3; ++a) {
        dim[a] = (int)std::ceil((ghi[a] - (double)lof[a]) / (double)cellf) + 1;
        if (dim[a] < 1) dim[a] = 1;
        if (dim[a] > 1040) return RT_OK;
        wn[a] = (uint64_t)dim[a] + 2 * B;
    }
    const uint64_t n_cells = wn[0] * wn[1] * wn[2];
    if (n_cells >= (1ull << 24)) return RT_OK;  // block indices are 24 bits in a header
    if (n_cells > 64ull * n + (1ull << 18)) return RT_OK;  // a few objects in a huge box: 64 bytes per (mostly empty) cell would be all table
    float c0[3];
    for (int a = 0; a < 3; ++a) c0[a] = (float)((double)lof[a] + (0.5 - B) * (double)cellf);
    const volatile float inv_step_v = 128.0f / cellf;
    const float inv_step = inv_step_v;
    // D: no ray of a frame starts outside the grid box - build_grid's box holds every uploaded primary origin (origin_lo / hi;
    // the pinhole camera's (0,0,0) is their initial value) and secondary rays start on object surfaces - so |oc| never exceeds
    // the box's diagonal (+ the two border cells the walks may look at). The walks' `t_enter <= 4096 dmin` admission is therefore
    // always met (t_enter = 0); what their `tame` test really screens is |d|^2 (the bound and the admission rule agree)
    double Dmax = 0;
    for (int a = 0; a < 3; ++a) Dmax += (ghi[a] - glo[a] + 2 * cell) * (ghi[a] - glo[a] + 2 * cell);
    Dmax = std::sqrt(Dmax);
    const double alpha0 = 6e-6 * K2, sq_alpha0 = std::sqrt(alpha0);
    // registration (as build_grid: the cells the registration SPHERE reaches)
    std::vector<uint32_t> start(n_cells + 1, 0), entries, fill;
    auto cell_index = [&](int x, int y, int z) { return (size_t)((((uint64_t)(z + B)) * wn[1] + (uint64_t)(y + B)) * wn[0] + (uint64_t)(x + B)); };
    uint64_t total = 0;
    for (int pass = 0; pass < 2; ++pass) {
        if (pass == 1) {
            for (size_t k = 0; k < (size_t)n_cells; ++k) start[k + 1] += start[k];
            total = start[n_cells];
            if (total > 0x7fffffffull) return RT_OK;
            entries.assign((size_t)total, 0);
            fill.assign(start.begin(), start.end() - 1);
        }
        for (uint32_t i = 0; i < n; ++i) {
            if (!(rg[i] >= 0) || !std::isfinite(rg[i])) continue;
            const double rb = rg[i] - 0.01 * cell_fine + 0.01 * cell;
            const double cc[3] = {sph[i].x, sph[i].y, sph[i].z};
            int lo_i[3], hi_i[3];
            for (int a = 0; a < 3; ++a) {
                lo_i[a] = std::max((int)std::floor((cc[a] - rb - (double)lof[a]) / (double)cellf), 0);
                hi_i[a] = std::min((int)std::floor((cc[a] + rb - (double)lof[a]) / (double)cellf), dim[a] - 1);
            }
            const double r2 = rb * rb;
            auto gap2 = [&](int a, int k) {
                const double w0 = (double)lof[a] + (double)cellf * k, w1 = w0 + (double)cellf;
                const double d = cc[a] < w0 ? w0 - cc[a] : (cc[a] > w1 ? cc[a] - w1 : 0.0);
                return d * d;
            };
            for (int z = lo_i[2]; z <= hi_i[2]; ++z) {
                const double dz2 = gap2(2, z);
                for (int y = lo_i[1]; y <= hi_i[1]; ++y) {
                    const double dyz2 = dz2 + gap2(1, y);
                    if (dyz2 > r2) continue;
                    for (int x = lo_i[0]; x <= hi_i[0]; ++x) {
                        if (dyz2 + gap2(0, x) > r2) continue;
                        const size_t k = cell_index(x, y, z);
                        if (pass == 0) { start[k + 1] += 1; if (++total > 64ull * n + 1024ull) return RT_OK; }
                        else entries[fill[k]++] = i;
                    }
                }
            }
        }
    }
    lap("registration");
    // blocks
    struct Enc { uint32_t word; uint32_t id; int s; };
    // Two sweeps over the cells, each on several threads: (1) every entry of a cell encoded on the finest of the four lattices that
    // holds it, the cell's entries ordered by lattice (stable), the blocks its chain needs counted; (2) - after a prefix sum that
    // gives every chain the indices it would get if the cells were laid out one after the other - the blocks written. The table is
    // the serial builder's, byte for byte.
    std::vector<Enc> encs((size_t)total);
    std::vector<uint32_t> chain_at((size_t)n_cells + 1, 0u);  // blocks behind the head, per cell; then their prefix sum
    std::atomic<uint64_t> hist_sum[5];
    for (auto& h : hist_sum) h.store(0);
    parallel_for((size_t)n_cells, 4096, [&](size_t k0, size_t k1) {
        uint64_t hist_local[5] = {0, 0, 0, 0, 0};
        for (size_t k = k0; k < k1; ++k) {
            const uint32_t cnt = start[k + 1] - start[k];
            if (cnt == 0) continue;
            const uint64_t wx = (uint64_t)k % wn[0], wy = ((uint64_t)k / wn[0]) % wn[1], wz = (uint64_t)k / (wn[0] * wn[1]);
            const float Cdev[3] = {std::fmaf((float)wx, cellf, c0[0]), std::fmaf((float)wy, cellf, c0[1]), std::fmaf((float)wz, cellf, c0[2])};
            Enc* enc = encs.data() + start[k];
            for (uint32_t j = 0; j < cnt; ++j) {
                const uint32_t i = entries[start[k] + j];
                const double w = std::fabs((double)c->h_grid_pre[i]);
                const double cc[3] = {sph[i].x, sph[i].y, sph[i].z};
                Enc e{(255u << 24) | (128u << 16) | (128u << 8) | 128u, i, 3};  // the whole cell and then some: 255 steps of cell / 16 around its centre
                bool found = false;
                for (int sc = 0; sc < 4 && !found; ++sc) {
                    const double inv = (double)inv_step * std::ldexp(1.0, -sc), step = 1.0 / inv;
                    double d2 = 0;
                    int q[3];
                    bool ok = true;
                    for (int a = 0; a < 3; ++a) {
                        const double u = (cc[a] - (double)Cdev[a]) * inv + 128.0;
                        q[a] = (int)std::floor(u + 0.5);
                        if (q[a] < 0 || q[a] > 255) ok = false;
                        d2 += (u - q[a]) * (u - q[a]);
                    }
                    if (!ok) continue;
                    // centre error in view-space units + the rounding of the device's transform of the ray origin
                    const double delta = std::sqrt(d2) * step + 1.5 * 5.97e-8 * (2.0 * Dmax + 128.0 * step);
                    const double w2 = (w + delta) * (w + delta) + 2.0 * delta * sq_alpha0 * Dmax + alpha0 * (2.0 * delta * Dmax + delta * delta);
                    const double r_lat = std::sqrt(w2) * (1.0 + 1e-6) * inv + 1e-3;
                    const int r = (int)std::ceil(r_lat);
                    if (r > 255) continue;
                    e = Enc{((uint32_t)r << 24) | ((uint32_t)q[2] << 16) | ((uint32_t)q[1] << 8) | (uint32_t)q[0], i, sc};
                    found = true;
                }
                hist_local[found ? e.s : 4] += 1;
                enc[j] = e;
            }
            std::stable_sort(enc, enc + cnt, [](const Enc& a, const Enc& b) { return a.s < b.s; });
            uint32_t n_blocks = 0;
            for (uint32_t pos = 0; pos < cnt; ++n_blocks) {
                const int sc = enc[pos].s;
                uint32_t m = 0;
                while (m < rt::kBlockEntries && pos < cnt && enc[pos].s == sc) { ++m; ++pos; }
            }
            chain_at[k + 1] = n_blocks - 1u;
        }
        for (int h = 0; h < 5; ++h) hist_sum[h].fetch_add(hist_local[h], std::memory_order_relaxed);
    });
    uint64_t hist[5], chain_blocks = 0;
    for (int h = 0; h < 5; ++h) hist[h] = hist_sum[h].load();
    for (size_t k = 0; k < (size_t)n_cells; ++k) { chain_blocks += chain_at[k + 1]; chain_at[k + 1] = (uint32_t)chain_blocks; }  // (< total <= 2^31)
    if (chain_blocks != 0 && n_cells + chain_blocks > (1ull << 24)) return RT_OK;  // block indices are 24 bits in a header
    const size_t n_blocks_all = (size_t)(n_cells + chain_blocks);
    std::vector<uint32_t> blocks(n_blocks_all * 8, 0u), ids(n_blocks_all * 8, c->n_objs);
    parallel_for((size_t)n_cells, 4096, [&](size_t k0, size_t k1) {
        for (size_t k = k0; k < k1; ++k) {
            const uint32_t cnt = start[k + 1] - start[k];
            if (cnt == 0) continue;
            const Enc* enc = encs.data() + start[k];
            size_t at = k, next_free = (size_t)n_cells + chain_at[k];
            uint32_t pos = 0;
            while (pos < cnt) {
                const int sc = enc[pos].s;
                uint32_t m = 0;
                while (m < rt::kBlockEntries && pos < cnt && enc[pos].s == sc) {
                    blocks[at * 8 + 1 + m] = enc[pos].word;
                    ids[at * 8 + m] = enc[pos].id;
                    ++m; ++pos;
                }
                const uint32_t next = pos < cnt ? (uint32_t)next_free++ : 0u;
                blocks[at * 8] = next | ((uint32_t)sc << 27);
                at = next;
            }
        }
    });
    if ((uint64_t)blocks.size() * 4ull >= 0xf0000000ull) return RT_OK;
    lap("encoding");
    {   // Empty cells say how many FURTHER steps of a walk are sure to stay in empty cells: the Chebyshev distance to the nearest
        // occupied cell minus one (a walk moves by one face per step), two-pass chamfer over the 26-neighbourhood of the padded
        // array, capped at 63 - in the six header bits the chain pointer and the scale leave free (24-26 and 29-31). The walk takes
        // those steps without fetching their blocks (a mesh leaves most of its grid empty: cfg5 walks 107 cells per ray).
        const int nx = (int)wn[0], ny = (int)wn[1], nz = (int)wn[2];
        std::vector<uint8_t> dist((size_t)n_cells);
        for (size_t k = 0; k < (size_t)n_cells; ++k) dist[k] = (start[k + 1] - start[k]) ? 0 : 64;
        auto at = [&](int x, int y, int z) -> uint8_t& { return dist[((size_t)z * ny + y) * nx + x]; };
        for (int pass = 0; pass < 2; ++pass) {
            const int dz = pass ? -1 : 1;
            for (int z = pass ? nz - 1 : 0; z != (pass ? -1 : nz); z += dz)
                for (int y = pass ? ny - 1 : 0; y != (pass ? -1 : ny); y += dz)
                    for (int x = pass ? nx - 1 : 0; x != (pass ? -1 : nx); x += dz) {
                        uint8_t& d = at(x, y, z);
                        if (d == 0) continue;
                        int best = d;
                        for (int oz = -1; oz <= 0; ++oz)  // the 13 neighbours already visited in this scan direction
                            for (int oy = -1; oy <= (oz ? 1 : 0); ++oy)
                                for (int ox = -1; ox <= ((oz || oy) ? 1 : -1); ++ox) {
                                    const int X = x + ox * dz, Y = y + oy * dz, Z = z + oz * dz;
                                    if ((unsigned)X >= (unsigned)nx || (unsigned)Y >= (unsigned)ny || (unsigned)Z >= (unsigned)nz) continue;
                                    best = std::min(best, (int)at(X, Y, Z) + 1);
                                }
                        d = (uint8_t)best;
                    }
        }
        c->blocks.take_skips = 1u;  // (cfg4 - a quarter of the cells empty, by ones and twos - still gains 0.25 ms: 15.0 -> 12.4 trips per ray)
        for (size_t k = 0; k < (size_t)n_cells; ++k) {
            if (dist[k] < 2) continue;
            const uint32_t skip = std::min<uint32_t>(63u, (uint32_t)dist[k] - 1u);
            blocks[8 * k] |= ((skip & 7u) << 24) | ((skip >> 3) << 29);
        }
    }
    lap("empty-space distances");
    RT_HIP(c, hipMalloc((void**)&c->d_walk_blocks, sizeof(uint32_t) * blocks.size()));
    RT_HIP(c, hipMalloc((void**)&c->d_walk_ids, sizeof(uint32_t) * ids.size()));
    RT_HIP(c, hipMemcpy(c->d_walk_blocks, blocks.data(), sizeof(uint32_t) * blocks.size(), hipMemcpyHostToDevice));
    RT_HIP(c, hipMemcpy(c->d_walk_ids, ids.data(), sizeof(uint32_t) * ids.size(), hipMemcpyHostToDevice));
    lap("uploads");
    rt::BlockGrid& b = c->blocks;
    if (const char* env = std::getenv("RT_BLOCK_SKIPS")) b.take_skips = env[0] != '0' ? 1u : 0u;  // measurement knob
    b.lox = lof[0]; b.loy = lof[1]; b.loz = lof[2];
    b.cell = cellf;
    b.inv_cell = 1.0f / cellf;
    b.nx = dim[0]; b.ny = dim[1]; b.nz = dim[2];
    b.c0x = c0[0]; b.c0y = c0[1]; b.c0z = c0[2];
    b.inv_step = inv_step;
    b.wnx = (uint32_t)wn[0]; b.wny = (uint32_t)wn[1];
    b.n_cells = (uint32_t)n_cells;
    b.blocks = c->d_walk_blocks;
    b.ids = c->d_walk_ids;
    b.none = c->n_objs;
    b.enabled = 1u;
    if (std::getenv("RT_WALK_STATS"))
        std::fprintf(stderr, "[blocks] %d x %d x %d cells, edge %g, %llu entries, %zu blocks (%llu chained), scale 0/1/2/3/whole-cell: %llu %llu %llu %llu %llu\n",
                     dim[0], dim[1], dim[2], (double)cellf, (unsigned long long)total, blocks.size() / 8, (unsigned long long)chain_blocks,
                     (unsigned long long)hist[0], (unsigned long long)hist[1], (unsigned long long)hist[2], (unsigned long long)hist[3], (unsigned long long)hist[4]);
    return RT_OK;
}

// Conservative uniform grid for the large-scene trace kernels (rt_grid.h explains the margins).
int build_grid(rt_context* c, const rt_object_data* objs, uint32_t n) {
    SetupTrace lap("grid");
    c->grid = rt::GridDesc{};
    // The grid reasons about ONE line in view space per ray. That needs w = 1 starts, affine instances (checked at
    // upload) and direction.w = 0: with a non-zero direction.w the reference adds every object's own translation
    // column to the object-space direction, i.e. each object sees a different line (found by the differential fuzz).
    if (n == 0 || (c->flags & RT_FLAG_NO_GRID) || !c->affine_w || !c->primary_w_one || !(c->pinhole || c->dir_w_zero)) return RT_OK;
    std::vector<Bound> sph(n);
    double lo[3] = {c->origin_lo[0], c->origin_lo[1], c->origin_lo[2]};
    double hi[3] = {c->origin_hi[0], c->origin_hi[1], c->origin_hi[2]};
    const double inf = std::numeric_limits<double>::infinity();
    parallel_for(n, 8192, [&](size_t i0, size_t i1) { for (size_t i = i0; i < i1; ++i) sph[i] = object_bound(objs[i]); });
    for (uint32_t i = 0; i < n; ++i) {
        if (!std::isfinite(sph[i].r)) continue;  // +inf: always-list, -inf: can never be hit
        const double cc[3] = {sph[i].x, sph[i].y, sph[i].z};
        const double pad = sph[i].r * 1.01;
        for (int a = 0; a < 3; ++a) {
            lo[a] = std::min(lo[a], cc[a] - pad);
            hi[a] = std::max(hi[a], cc[a] + pad);
        }
    }
    const double ext[3] = {hi[0] - lo[0], hi[1] - lo[1], hi[2] - lo[2]};
    const double diag = std::sqrt(ext[0] * ext[0] + ext[1] * ext[1] + ext[2] * ext[2]);
    if (!(diag > 0) || !std::isfinite(diag)) return RT_OK;
    // cell edge: about two cells per object in volume, at most 256 cells per axis
    double cells_per_object = 3.0;  // (2: 22.3 ms per cfg4 frame, 3: 22.0, 4: 22.0 - since the shadow rays of the last light left the grid for the light tiles)
    if (const char* env = std::getenv("RT_GRID_CELLS_PER_OBJECT")) {  // tuning knob (results do not depend on it)
        const double v = std::atof(env);
        if (v > 0.01 && v < 1000.0) cells_per_object = v;
    }
    double cell = std::cbrt(std::max(ext[0], 1e-6) * std::max(ext[1], 1e-6) * std::max(ext[2], 1e-6) / (cells_per_object * n));
    cell = std::max(cell, std::max(ext[0], std::max(ext[1], ext[2])) / 256.0);
    if (!(cell > 0) || !std::isfinite(cell)) return RT_OK;
    // Objects that crowd on surfaces (tessellated meshes) leave most of such a grid empty and pile up in the rest:
    // refine while an occupied cell holds more than four object centres on average (random clouds: ~1.3, no change)
    {
        const double max_ext = std::max(ext[0], std::max(ext[1], ext[2]));
        std::vector<uint64_t> keys;
        const double vol = std::max(ext[0], 1e-6) * std::max(ext[1], 1e-6) * std::max(ext[2], 1e-6);
        double max_dim = c->has_triangles ? 560.0 : 280.0;  // meshes (cfg5, 1 M triangles): 280: 76.5 ms per frame, 400: 70.6, 560: 66.4, 800: 75.5
        if (const char* env = std::getenv("RT_GRID_MAX_DIM")) max_dim = std::min(1000.0, std::max(16.0, std::atof(env)));
        for (int attempt = 0; attempt < 10 && max_ext / (cell * 0.7) < max_dim && vol / std::pow(cell * 0.7, 3) < 128.0 * n; ++attempt) {
            keys.clear();
            keys.reserve(n);
            for (uint32_t i = 0; i < n; ++i) {
                if (!std::isfinite(sph[i].r)) continue;
                const uint64_t kx = (uint64_t)std::max(0.0, std::floor((sph[i].x - lo[0]) / cell));
                const uint64_t ky = (uint64_t)std::max(0.0, std::floor((sph[i].y - lo[1]) / cell));
                const uint64_t kz = (uint64_t)std::max(0.0, std::floor((sph[i].z - lo[2]) / cell));
                keys.push_back((kz << 42) | (ky << 21) | kx);
            }
            std::sort(keys.begin(), keys.end());
            const size_t distinct = (size_t)(std::unique(keys.begin(), keys.end()) - keys.begin());
            if (distinct == 0 || (double)keys.size() / (double)distinct <= 4.0) break;
            cell *= 0.7;
        }
    }
    lap("bounds + cell size");
    // Radii (rt_grid.h derives the bound): with u = 2^-24, a ray that starts `dist` from the centre c of an object
    // with bounding radius R and condition number kappa can only be accepted by the reference's fp32 test if its
    // line passes c within
    //     sqrt(R^2 (1 + 8u) + 14 u kappa^2 dist^2) + 10.4 u kappa (|start| + |c|) + 9 u kappa^2 dist.
    // Everything below uses u_eff = 2e-7 (3.3 u):  a = R sqrt(1 + 2e-6),  L = 2.2e-6 kappa (S + |c|) + 2e-6 kappa^2 D,
    //   registration radius  sqrt(a^2 + 3e-6 kappa^2 D^2) + L   (D = farthest possible ray origin, S = largest |origin|)
    //   pre-test radius      a + 2L + 1e-6 |c|, used with the ACTUAL distance: r^2 = w^2 + alpha dist^2, alpha = 6e-6 K^2
    // (K^2 = largest kappa^2 among the objects that use the distance-dependent form, at most 4; more anisotropic
    // objects carry their full registration radius instead, flagged by a negative w).
    double S_max = 0;
    for (int k = 0; k < 8; ++k) {
        const double px = (k & 1) ? hi[0] : lo[0], py = (k & 2) ? hi[1] : lo[1], pz = (k & 4) ? hi[2] : lo[2];
        S_max = std::max(S_max, std::sqrt(px * px + py * py + pz * pz));
    }
    constexpr double kKappa2Tight = 4.0;
    double K2 = 1.0;
    std::vector<double> rg(n), rpre(n);
    std::vector<uint32_t> always;
    for (uint32_t i = 0; i < n; ++i) {
        rpre[i] = 0;
        if (sph[i].r == -inf) { rg[i] = -1.0; continue; }
        if (!std::isfinite(sph[i].r)) { rg[i] = inf; always.push_back(i); continue; }
        double D2 = 0;
        const double cc[3] = {sph[i].x, sph[i].y, sph[i].z};
        for (int a = 0; a < 3; ++a) {
            const double d = std::max(std::fabs(cc[a] - lo[a]), std::fabs(hi[a] - cc[a]));
            D2 += d * d;
        }
        const double k2 = sph[i].kappa2, kap = std::sqrt(k2);
        const double cl = std::sqrt(cc[0] * cc[0] + cc[1] * cc[1] + cc[2] * cc[2]);
        const double a2 = sph[i].r * sph[i].r * (1.0 + 2e-6);
        const double L = 2.2e-6 * kap * (S_max + cl) + 2e-6 * k2 * std::sqrt(D2);
        double reg = std::sqrt(a2 + 3e-6 * k2 * D2) + L;
        if (objs[i].type == 2u) {
            // triangle: the guard |(c - start) x d|^2 <= R^2 |d|^2 is computed with an error of ~10 u R |c - start| |d|^2
            // (cross-product form), i.e. it can pass lines up to R + ~5 u D away; + the rounding of c - start itself
            reg = sph[i].r * (1.0 + 1e-6) + 2e-6 * std::sqrt(D2) + 1e-6 * (S_max + cl);
        }
        rg[i] = reg * (1.0 + 1e-6) + 0.01 * cell;  // + slack for the kernels' fp32 cell arithmetic
        if (objs[i].type == 2u) rpre[i] = -(reg * (1.0 + 1e-6) + 1e-6 * cl);  // no distance term: its guard has none to speak of
        else if (k2 <= kKappa2Tight) { rpre[i] = std::sqrt(a2) + 2.0 * L + 1e-6 * cl; K2 = std::max(K2, k2); }
        else rpre[i] = -(reg * (1.0 + 1e-6) + 1e-6 * cl);
        if (!std::isfinite(rg[i]) || rg[i] > 0.25 * diag) { rg[i] = inf; always.push_back(i); }  // as big as the scene: test it for every ray
    }
    // many scene-sized objects: a grid would not pay, stay with the brute-force stream - unless the scene holds
    // triangles, which only this path can trace (then every ray simply tests the whole always-list)
    if (always.size() > 64 && !c->has_triangles) return RT_OK;
    // the grid box: everything registered plus the ray origins, padded by one cell
    double glo[3], ghi[3];
    for (int a = 0; a < 3; ++a) { glo[a] = lo[a] - cell; ghi[a] = hi[a] + cell; }
    for (uint32_t i = 0; i < n; ++i) {
        if (!(rg[i] >= 0) || !std::isfinite(rg[i])) continue;
        const double cc[3] = {sph[i].x, sph[i].y, sph[i].z};
        for (int a = 0; a < 3; ++a) { glo[a] = std::min(glo[a], cc[a] - rg[i] - cell); ghi[a] = std::max(ghi[a], cc[a] + rg[i] + cell); }
    }
    int dim[3];
    for (int a = 0; a < 3; ++a) {
        dim[a] = (int)std::ceil((ghi[a] - glo[a]) / cell);
        if (dim[a] < 1) dim[a] = 1;
        if (dim[a] > 1040) return RT_OK;
    }
    const float cellf = (float)cell;
    const float lof[3] = {(float)glo[0], (float)glo[1], (float)glo[2]};
    // cell range of a sphere's box, computed with the SAME float origin / cell edge the kernels use (rg already
    // contains 0.01 cell of slack for the kernels' fp32 cell arithmetic)
    auto range = [&](uint32_t i, int a, int& i0, int& i1) {
        const double cc = (a == 0 ? sph[i].x : a == 1 ? sph[i].y : sph[i].z);
        i0 = (int)std::floor((cc - rg[i] - (double)lof[a]) / (double)cellf);
        i1 = (int)std::floor((cc + rg[i] - (double)lof[a]) / (double)cellf);
        i0 = std::max(i0, 0);
        i1 = std::min(i1, dim[a] - 1);
    };
    const size_t n_cells = (size_t)dim[0] * dim[1] * dim[2];
    std::vector<uint32_t> start(n_cells + 1, 0);
    size_t total = 0;
    lap("radii + box");
    // (object, cell) pairs: counted in 64 bits against the budget as the first pass goes, and an object's BOX of cells
    // is checked before its cells are visited - many large overlapping objects must neither wrap the 32-bit prefix
    // sums nor cost billions of iterations before the grid is given up
    const uint64_t entry_budget = std::min<uint64_t>((c->has_triangles ? 1024ull : 64ull) * n + 1024ull, 0x7fffffffull);
    uint64_t counted = 0;
    std::vector<uint32_t> entries, fill;
    for (int pass = 0; pass < 2; ++pass) {
        if (pass == 1) {
            fill.assign(start.begin(), start.end() - 1);
            entries.assign(total, 0);
        }
        for (uint32_t i = 0; i < n; ++i) {
            if (!(rg[i] >= 0) || !std::isfinite(rg[i])) continue;
            int x0, x1, y0, y1, z0, z1;
            range(i, 0, x0, x1); range(i, 1, y0, y1); range(i, 2, z0, z1);
            if (pass == 0) {
                if (x1 < x0 || y1 < y0 || z1 < z0) continue;
                const uint64_t box_cells = (uint64_t)(x1 - x0 + 1) * (uint64_t)(y1 - y0 + 1) * (uint64_t)(z1 - z0 + 1);
                if (box_cells > 8ull * entry_budget) return RT_OK;  // the sphere fills >= ~half of its box: far over budget on its own
            }
            // of the cells its box touches, only those the registration SPHERE reaches (distance from the centre to
            // the cell's box <= radius, cell walls taken from the kernels' float origin / edge, radius already
            // holding 0.01 cell of slack): a quarter fewer entries for spheres that are small against a cell
            const double cc[3] = {sph[i].x, sph[i].y, sph[i].z};
            const double r2 = rg[i] * rg[i];
            auto gap2 = [&](int a, int k) {  // squared distance from the centre to slab k of axis a
                const double w0 = (double)lof[a] + (double)cellf * k, w1 = w0 + (double)cellf;
                const double d = cc[a] < w0 ? w0 - cc[a] : (cc[a] > w1 ? cc[a] - w1 : 0.0);
                return d * d;
            };
            for (int z = z0; z <= z1; ++z) {
                const double dz2 = gap2(2, z);
                for (int y = y0; y <= y1; ++y) {
                    const double dyz2 = dz2 + gap2(1, y);
                    if (dyz2 > r2) continue;
                    for (int x = x0; x <= x1; ++x) {
                        if (dyz2 + gap2(0, x) > r2) continue;
                        const size_t cidx = ((size_t)z * dim[1] + y) * dim[0] + x;
                        if (pass == 0) { start[cidx + 1] += 1; ++counted; }
                        else entries[fill[cidx]++] = i;
                    }
                }
            }
            if (pass == 0 && counted > entry_budget) return RT_OK;  // objects too large for this cell size: not worth it
        }
        if (pass == 0) {
            lap("cell lists: count");
            for (size_t k = 0; k < n_cells; ++k) start[k + 1] += start[k];
            total = start[n_cells];
            // the kernels address these tables with 32-bit byte offsets (table_at)
            if ((uint64_t)total * 16ull >= 0xffffffffull || (uint64_t)n_cells * 32ull >= 0xffffffffull) return RT_OK;
        } else {
            lap("cell lists: fill");
            RT_HIP(c, hipMalloc((void**)&c->d_grid_entries, sizeof(uint32_t) * (total + 1)));
            RT_HIP(c, hipMalloc((void**)&c->d_grid_always, sizeof(uint32_t) * (always.size() + 1)));
            {   // per entry: the sphere the object was registered with (rounded outwards), for the kernels' pre-test
                // pre-test radius per object, rounded away from zero; negative = "already holds the worst-case distance term"
                c->h_grid_pre.resize(n);
                for (uint32_t i = 0; i < n; ++i)
                    c->h_grid_pre[i] = rpre[i] >= 0 ? std::nextafter((float)rpre[i], std::numeric_limits<float>::infinity())
                                                    : std::nextafter((float)rpre[i], -std::numeric_limits<float>::infinity());
                std::vector<float4> es(total);
                parallel_for(total, 1u << 16, [&](size_t k0, size_t k1) {
                    for (size_t k = k0; k < k1; ++k) {
                        const uint32_t i = entries[k];
                        es[k] = make_float4((float)sph[i].x, (float)sph[i].y, (float)sph[i].z, c->h_grid_pre[i]);
                    }
                });
                RT_HIP(c, hipMalloc((void**)&c->d_grid_entry_sphere, sizeof(float4) * (total + 1)));
                if (total) RT_HIP(c, hipMemcpy(c->d_grid_entry_sphere, es.data(), sizeof(float4) * total, hipMemcpyHostToDevice));
                std::vector<uint2> ranges(n_cells);
                parallel_for(n_cells, 1u << 18, [&](size_t k0, size_t k1) {
                    for (size_t k = k0; k < k1; ++k) ranges[k] = make_uint2(start[k], start[k + 1] - start[k]);
                });
                // Empty cells carry, in the unused offset word, how many FURTHER steps of a walk are sure to land in
                // empty cells too: the Chebyshev distance to the nearest occupied cell minus one (a walk moves by one
                // face per step), from a two-pass chamfer over the 26-neighbourhood, capped at 255. The grid walk
                // takes those steps without fetching their records (meshes leave most of a grid empty).
                if (n_cells <= (64ull << 20)) {
                    std::vector<uint8_t> dist(n_cells);
                    for (size_t k = 0; k < n_cells; ++k) dist[k] = ranges[k].y ? 0 : 255;
                    const int nx = dim[0], ny = dim[1], nz = dim[2];
                    auto at = [&](int x, int y, int z) -> uint8_t& { return dist[((size_t)z * ny + y) * nx + x]; };
                    for (int pass = 0; pass < 2; ++pass) {
                        const int z0 = pass ? nz - 1 : 0, z1 = pass ? -1 : nz, dz = pass ? -1 : 1;
                        for (int z = z0; z != z1; z += dz)
                            for (int y = pass ? ny - 1 : 0; y != (pass ? -1 : ny); y += dz)
                                for (int x = pass ? nx - 1 : 0; x != (pass ? -1 : nx); x += dz) {
                                    uint8_t& d = at(x, y, z);
                                    if (d == 0) continue;
                                    int best = d;
                                    // the 13 neighbours already visited in this scan direction
                                    for (int oz = -1; oz <= 0; ++oz)
                                        for (int oy = -1; oy <= (oz ? 1 : 0); ++oy)
                                            for (int ox = -1; ox <= ((oz || oy) ? 1 : -1); ++ox) {
                                                const int X = x + ox * dz, Y = y + oy * dz, Z = z + oz * dz;
                                                if ((unsigned)X >= (unsigned)nx || (unsigned)Y >= (unsigned)ny || (unsigned)Z >= (unsigned)nz) continue;
                                                best = std::min(best, (int)at(X, Y, Z) + 1);
                                            }
                                    d = (uint8_t)best;
                                }
                    }
                    for (size_t k = 0; k < n_cells; ++k)
                        if (ranges[k].y == 0) ranges[k].x = dist[k] > 1 ? (uint32_t)dist[k] - 1u : 0u;
                } else {
                    parallel_for(n_cells, 1u << 18, [&](size_t k0, size_t k1) {
                        for (size_t k = k0; k < k1; ++k)
                            if (ranges[k].y == 0) ranges[k].x = 0u;
                    });
                }
                RT_HIP(c, hipMalloc((void**)&c->d_grid_cell_range, sizeof(uint2) * n_cells));
                RT_HIP(c, hipMemcpy(c->d_grid_cell_range, ranges.data(), sizeof(uint2) * n_cells, hipMemcpyHostToDevice));
                // the same ranges with the first entry inline (the persistent walk's 32-byte cell records; the kernel variants
                // for scenes without triangles read these, the mesh variants the 8-byte ranges: a mesh leaves its grid mostly
                // empty, and an empty cell has nothing to inline)
                if (!c->has_triangles) {
                    std::vector<float4> rec(2 * n_cells);
                    for (size_t k = 0; k < n_cells; ++k) {
                        const bool any = ranges[k].y != 0u;
                        rec[2 * k] = any ? es[ranges[k].x] : make_float4(0.f, 0.f, 0.f, 0.f);
                        uint32_t w[4] = {ranges[k].x, ranges[k].y, any ? entries[ranges[k].x] : 0u, 0u};
                        std::memcpy(&rec[2 * k + 1], w, sizeof(w));
                    }
                    RT_HIP(c, hipMalloc((void**)&c->d_grid_cell_rec, sizeof(float4) * 2 * n_cells));
                    RT_HIP(c, hipMemcpy(c->d_grid_cell_rec, rec.data(), sizeof(float4) * 2 * n_cells, hipMemcpyHostToDevice));
                    build_walk_records(c, dim, ranges, es, entries);
                }
            }
            if (total) RT_HIP(c, hipMemcpy(c->d_grid_entries, entries.data(), sizeof(uint32_t) * total, hipMemcpyHostToDevice));
            if (!always.empty())
                RT_HIP(c, hipMemcpy(c->d_grid_always, always.data(), sizeof(uint32_t) * always.size(), hipMemcpyHostToDevice));
        }
    }
    rt::GridDesc& g = c->grid;
    g.lox = lof[0]; g.loy = lof[1]; g.loz = lof[2];
    g.cell = cellf;
    g.inv_cell = 1.0f / cellf;
    g.nx = dim[0]; g.ny = dim[1]; g.nz = dim[2];
    g.cell_range = c->d_grid_cell_range;
    g.cell_rec = c->d_grid_cell_rec;
    g.entries = c->d_grid_entries;
    g.entry_sphere = c->d_grid_entry_sphere;
    g.always = c->d_grid_always;
    g.n_always = (uint32_t)always.size();
    g.has_triangles = c->has_triangles ? 1u : 0u;
    lap("cell lists + uploads");
    {   // first-cell rule (rt_grid.h: entered_inside; RT_WALK_FIRST_CELL): the ball it uses must stay inside every object's
        // registration radius by 1e-3 cell (DDA) + the rule's own fp32 rounding (~1e-6 of the largest coordinate)
        double worst = 0.0;  // max over objects of |pre-test radius| - registration radius (normally about -0.01 cell)
        bool any = false;
        for (uint32_t i = 0; i < n; ++i) {
            if (!(rg[i] >= 0) || !std::isfinite(rg[i])) continue;
            const double d = std::fabs((double)c->h_grid_pre[i]) - rg[i];
            worst = any ? std::max(worst, d) : d;
            any = true;
        }
        const double shrink = std::max(0.0, worst + 1e-3 * cell + 2e-6 * (S_max + diag));
        g.own_shrink = std::nextafter((float)shrink, std::numeric_limits<float>::infinity());
    }
    g.pretest_alpha = std::nextafter((float)(6e-6 * K2 + 8e-6), std::numeric_limits<float>::infinity());
    g.enabled = 1u;
    if (!c->has_triangles || !std::getenv("RT_NO_TRI_BLOCKS")) {  // (meshes too since round 3: cfg5 56 -> 40 ms; the knob restores round 2's walk for them)
        StopWatch sw;
        const int rc = build_walk_blocks(c, n, sph, rg, cell, glo, ghi, K2);
        if (rc != RT_OK) return rc;
        c->setup.blocks_ms = sw.lap_ms();
    }
    c->h_grid_spheres.resize(4 * (size_t)n);
    for (uint32_t i = 0; i < n; ++i) {
        c->h_grid_spheres[4 * i] = sph[i].x; c->h_grid_spheres[4 * i + 1] = sph[i].y; c->h_grid_spheres[4 * i + 2] = sph[i].z;
        c->h_grid_spheres[4 * i + 3] = rg[i];
    }
    return RT_OK;
}

// Light tiles (rt_grid.h: LightTiles): the objects a shadow ray towards the LAST positional light can meet, binned by
// direction as seen from that light. shade_and_reflect's colour comes from the last light (Q1), so outside literal
// mode nearly every shadow ray goes there; rays towards other lights (stale-specular scans) keep using the grid walk.
// Needs: the conservative grid (its registration radii are the ones used here, + 1e-3 for the ray's own rounding: the
// line of a shadow ray passes the light within ~1e-5), no always-tested objects, and an axis-aligned plane through the
// light with every object strictly (by its radius + 0.05) on one side - else nothing is built and the grid walk serves.
int build_light_tiles(rt_context* c, const rt_light* lights) {
    c->light_tiles = rt::LightTiles{};
    if (!c->grid.enabled || c->grid.n_always != 0 || c->kernel != RT_KERNEL_SHADE_AND_REFLECT || (c->flags & RT_FLAG_LITERAL) ||
        c->n_lights == 0 || c->h_grid_spheres.empty() || c->h_grid_pre.size() != c->n_objs)
        return RT_OK;
    if (std::getenv("RT_NO_LIGHT_TILES")) return RT_OK;  // measurement knob
    const uint32_t li = c->n_lights - 1u;
    const float* lp = lights[li].position;
    if (!(lp[3] != 0.f) || !std::isfinite(lp[0] + lp[1] + lp[2])) return RT_OK;  // directional (or garbage): no centre of projection
    const double L[3] = {lp[0], lp[1], lp[2]};
    SetupTrace lap("light tiles");
    const uint32_t n = c->n_objs;
    const double inf = std::numeric_limits<double>::infinity();
    // The fp32 shadow ray - start fl(P + 0.01 n), direction fl(L - P) - misses the exact line through the light by about
    // 1e-7 x (the coordinates involved + the distance to the light): the pad every registration sphere gets for it, and the
    // absolute slack of the kernels' distance cut, scale with the scene like the grid's own radii do (a fixed
    // 1e-3 is too little once coordinates or light distances reach 1e4).
    double coord_max = std::sqrt(L[0] * L[0] + L[1] * L[1] + L[2] * L[2]), reach_max = 0;
    for (uint32_t i = 0; i < n; ++i) {
        const double r = c->h_grid_spheres[4 * i + 3];
        if (!(r >= 0) || r == inf) continue;
        double cl = 0, dl = 0;
        for (int a = 0; a < 3; ++a) {
            cl += c->h_grid_spheres[4 * i + a] * c->h_grid_spheres[4 * i + a];
            dl += (c->h_grid_spheres[4 * i + a] - L[a]) * (c->h_grid_spheres[4 * i + a] - L[a]);
        }
        coord_max = std::max(coord_max, std::sqrt(cl) + r);
        reach_max = std::max(reach_max, std::sqrt(dl) + r);
    }
    const double kPad = std::max(1e-3, 4e-7 * (2.0 * coord_max + reach_max)), kFront = 0.05;
    // projection axis: every registered object strictly in front of the plane through the light
    int best_axis = -1;
    double best_sign = 0, best_clear = 0;
    for (int a = 0; a < 3; ++a)
        for (double sg : {-1.0, 1.0}) {
            double clear = inf;  // min over objects of (signed depth - radius)
            for (uint32_t i = 0; i < n && clear > kFront; ++i) {
                const double r = c->h_grid_spheres[4 * i + 3];
                if (!(r >= 0) || r == inf) continue;
                clear = std::min(clear, sg * (c->h_grid_spheres[4 * i + a] - L[a]) - (r + kPad));
            }
            if (clear > kFront && clear != inf && clear > best_clear) { best_clear = clear; best_axis = a; best_sign = sg; }
        }
    if (best_axis < 0) return RT_OK;
    // light-local frame: z' = -sign * (p - L)[axis] (objects at z' < 0), x', y' = the other two components
    const uint32_t az = (uint32_t)best_axis, ax = (az + 1u) % 3u, ay = (az + 2u) % 3u;
    const double szn = -best_sign;
    struct Rect { double u0, u1, v0, v1; };
    std::vector<Rect> rect(n, Rect{1, -1, 1, -1});
    double U0 = inf, U1 = -inf, V0 = inf, V1 = -inf;
    auto span = [&](double cx, double cz, double r, double& lo, double& hi) -> bool {
        // directions (x', z') through the origin that meet the disc (cx, cz; r), as x' / -z': tan of [phi - alpha, phi + alpha]
        const double rho = std::sqrt(cx * cx + cz * cz);
        if (!(rho > r)) return false;
        const double phi = std::atan2(cx, -cz), alpha = std::asin(std::min(1.0, r / rho));
        if (!(std::fabs(phi) + alpha < 1.5533)) return false;  // within 89 degrees of the axis, or no usable tangent
        lo = std::tan(phi - alpha);
        hi = std::tan(phi + alpha);
        // + what the kernel's fp32 (u, v) of a ray can be off by (~3e-7 (1 + |u|)), 30 times over
        lo -= 1e-5 * (1.0 + std::fabs(lo));
        hi += 1e-5 * (1.0 + std::fabs(hi));
        return true;
    };
    std::atomic<bool> no_span{false};  // an object without a usable tangent: no light tiles for this scene
    parallel_for(n, 8192, [&](size_t i0, size_t i1) {
        for (size_t i = i0; i < i1 && !no_span.load(std::memory_order_relaxed); ++i) {
            const double r0 = c->h_grid_spheres[4 * i + 3];
            if (!(r0 >= 0) || r0 == inf) continue;  // can never be hit
            const double r = r0 + kPad;
            const double q[3] = {c->h_grid_spheres[4 * i] - L[0], c->h_grid_spheres[4 * i + 1] - L[1], c->h_grid_spheres[4 * i + 2] - L[2]};
            const double qx = q[ax], qy = q[ay], qz = szn * q[az];
            Rect rc;
            if (!span(qx, qz, r, rc.u0, rc.u1) || !span(qy, qz, r, rc.v0, rc.v1)) { no_span.store(true, std::memory_order_relaxed); break; }
            rect[i] = rc;
        }
    });
    if (no_span.load()) return RT_OK;
    for (uint32_t i = 0; i < n; ++i) {
        const Rect& rc = rect[i];
        if (!(rc.u1 >= rc.u0)) continue;
        U0 = std::min(U0, rc.u0); U1 = std::max(U1, rc.u1); V0 = std::min(V0, rc.v0); V1 = std::max(V1, rc.v1);
    }
    if (!(U1 > U0) || !(V1 > V0) || !std::isfinite(U0 + U1 + V0 + V1)) return RT_OK;
    lap("axis + rectangles");
    // tile count: ~1.6 sqrt(n) per axis, halved while the lists would hold more than 24 entries per object
    double tile_factor = 1.6;
    if (const char* env = std::getenv("RT_LT_TILE_FACTOR")) {  // tuning knob (results do not depend on it)
        const double v = std::atof(env);
        if (v >= 0.1 && v <= 8.0) tile_factor = v;
    }
    uint32_t T = (uint32_t)std::min(1024.0, std::max(16.0, tile_factor * std::sqrt((double)n)));
    std::vector<uint32_t> start, entries, fill;
    uint64_t total = 0;
    float u0f = 0, v0f = 0, inv_du = 0, inv_dv = 0;
    auto tile_span = [&](double lo, double hi, float base, float inv, uint32_t& t0, uint32_t& t1) {
        // the fp32 expression the kernel evaluates, in double, with 0.01 tile of slack for the kernel's own rounding of it
        // (its (u, v) error is already inside the rectangle's padding; the product and the subtraction add < 1e-3 tile)
        const double a = std::floor((lo - (double)base) * (double)inv - 0.01), b = std::floor((hi - (double)base) * (double)inv + 0.01);
        t0 = (uint32_t)std::max(0.0, a);
        t1 = (uint32_t)std::min((double)T - 1.0, std::max(0.0, b));
    };
    for (;;) {
        const double du = (U1 - U0) / T * (1.0 + 1e-6), dv = (V1 - V0) / T * (1.0 + 1e-6);
        u0f = std::nextafter((float)U0, -std::numeric_limits<float>::infinity());
        v0f = std::nextafter((float)V0, -std::numeric_limits<float>::infinity());
        inv_du = (float)(1.0 / du);
        inv_dv = (float)(1.0 / dv);
        start.assign((size_t)T * T + 1, 0);
        total = 0;
        for (uint32_t i = 0; i < n; ++i) {
            if (!(rect[i].u1 >= rect[i].u0)) continue;
            uint32_t a0, a1, b0, b1;
            tile_span(rect[i].u0, rect[i].u1, u0f, inv_du, a0, a1);
            tile_span(rect[i].v0, rect[i].v1, v0f, inv_dv, b0, b1);
            total += (uint64_t)(a1 - a0 + 1) * (b1 - b0 + 1);
        }
        if (total <= 24ull * n + 4096ull || T <= 16u) break;
        T /= 2u;
    }
    if (total > 64ull * n + 4096ull || total * 32ull >= 0xffffffffull) return RT_OK;  // objects too wide as seen from the light / table beyond 32-bit byte offsets
    for (int pass = 0; pass < 2; ++pass) {
        if (pass == 1) {
            for (size_t k = 0; k < (size_t)T * T; ++k) start[k + 1] += start[k];
            entries.assign((size_t)total, 0);
            fill.assign(start.begin(), start.end() - 1);
        }
        for (uint32_t i = 0; i < n; ++i) {
            if (!(rect[i].u1 >= rect[i].u0)) continue;
            uint32_t a0, a1, b0, b1;
            tile_span(rect[i].u0, rect[i].u1, u0f, inv_du, a0, a1);
            tile_span(rect[i].v0, rect[i].v1, v0f, inv_dv, b0, b1);
            for (uint32_t b = b0; b <= b1; ++b)
                for (uint32_t a = a0; a <= a1; ++a) {
                    if (pass == 0) start[(size_t)b * T + a + 1] += 1;
                    else entries[fill[(size_t)b * T + a]++] = i;
                }
        }
    }
    std::vector<uint2> ranges((size_t)T * T);
    lap("count + fill");
    for (size_t k = 0; k < ranges.size(); ++k) ranges[k] = make_uint2(start[k], start[k + 1] - start[k]);
    // a tile's entries ordered by how far from the light the object starts: a ray's list ends at the first one that starts
    // beyond its own origin (an occluder's hit point lies between origin and light, within the registration radius of its centre)
    std::vector<float> key(n, 0.f);
    for (uint32_t i = 0; i < n; ++i) {
        const double r0 = c->h_grid_spheres[4 * i + 3];
        if (!(r0 >= 0) || r0 == inf) continue;
        const double q[3] = {c->h_grid_spheres[4 * i] - L[0], c->h_grid_spheres[4 * i + 1] - L[1], c->h_grid_spheres[4 * i + 2] - L[2]};
        const double d = std::sqrt(q[0] * q[0] + q[1] * q[1] + q[2] * q[2]) - (r0 + kPad);
        key[i] = std::nextafter((float)(d * (1.0 - 1e-6)), -std::numeric_limits<float>::infinity());
    }
    parallel_for(ranges.size(), 1024, [&](size_t t0, size_t t1) {
        for (size_t t = t0; t < t1; ++t)
            std::sort(entries.begin() + start[t], entries.begin() + start[t + 1], [&](uint32_t a, uint32_t b) { return key[a] < key[b] || (key[a] == key[b] && a < b); });
    });
    lap("keys + per-tile sorts");
    // The lists as RECORDS (tile_range + two float4 per entry) are what the kernels read where the block form below is not
    // built; with blocks they are never touched, and 56 MB of them (cfg4) are neither made nor uploaded.
    std::vector<float4> recs;
    auto upload_records = [&]() -> int {
        recs.resize(2 * (size_t)total);
        for (size_t k = 0; k < (size_t)total; ++k) {
            const uint32_t i = entries[k];
            float id_bits;
            std::memcpy(&id_bits, &i, 4);
            recs[2 * k] = make_float4((float)c->h_grid_spheres[4 * i], (float)c->h_grid_spheres[4 * i + 1], (float)c->h_grid_spheres[4 * i + 2], c->h_grid_pre[i]);
            recs[2 * k + 1] = make_float4(key[i], id_bits, 0.f, 0.f);
        }
        RT_HIP(c, hipMalloc((void**)&c->d_lt_range, sizeof(uint2) * ranges.size()));
        RT_HIP(c, hipMalloc((void**)&c->d_lt_records, sizeof(float4) * (recs.size() + 2)));
        RT_HIP(c, hipMemcpy(c->d_lt_range, ranges.data(), sizeof(uint2) * ranges.size(), hipMemcpyHostToDevice));
        if (total) RT_HIP(c, hipMemcpy(c->d_lt_records, recs.data(), sizeof(float4) * recs.size(), hipMemcpyHostToDevice));
        c->light_tiles.tile_range = c->d_lt_range;
        c->light_tiles.records = c->d_lt_records;
        return RT_OK;
    };
    rt::LightTiles& lt = c->light_tiles;
    lt.lx = lp[0]; lt.ly = lp[1]; lt.lz = lp[2];
    lt.u0 = u0f; lt.v0 = v0f; lt.inv_du = inv_du; lt.inv_dv = inv_dv;
    lt.tiles_u = T; lt.tiles_v = T;
    lt.ax = ax; lt.ay = ay; lt.az = az;
    lt.sx = 1.f; lt.sy = 1.f; lt.sz = (float)szn;
    lt.light = li;
    lt.cut_pad = (float)std::max(1e-4, 4e-7 * (2.0 * coord_max + reach_max));
    lt.enabled = 1u;
    if ((!c->has_triangles || !std::getenv("RT_NO_TRI_BLOCKS")) && total && !std::getenv("RT_NO_LT_BLOCKS")) {
        // ... and as 32-byte blocks of three candidates (LightTiles::blocks). Lattice: 16 bits per axis over the grid box; every
        // sphere rounded outwards exactly as build_walk_blocks does it (the device's own fma for the centre, the known
        // quantisation error d added to the radius with the cross term of the distance-dependent tolerance:
        // w'^2 = (w + d)^2 + 2 d sqrt(a) D + a (2 d D + d^2), a = the pre-test's alpha, D = the grid box's diagonal).
        const rt::GridDesc& g = c->grid;
        const float lof[3] = {g.lox, g.loy, g.loz};
        const double ext = (double)g.cell * std::max(g.nx, std::max(g.ny, g.nz));
        const volatile float stepv = (float)(ext / 65535.0 * (1.0 + 1e-6));
        const float stepf = stepv;
        const double Dbox = std::sqrt((double)g.cell * g.nx * (double)g.cell * g.nx + (double)g.cell * g.ny * (double)g.cell * g.ny +
                                      (double)g.cell * g.nz * (double)g.cell * g.nz);
        const double a0 = (double)g.pretest_alpha, sa0 = std::sqrt(a0);
        std::vector<uint16_t> q(3 * (size_t)n, 0);
        std::vector<double> wq(n, 0.0);
        double rmax = 0.0, kmax = 0.0;
        bool ok = stepf > 0.f && std::isfinite(stepf);
        for (uint32_t i = 0; i < n && ok; ++i) {
            if (!(rect[i].u1 >= rect[i].u0)) continue;
            double d2 = 0;
            for (int a = 0; a < 3; ++a) {
                const double cc = c->h_grid_spheres[4 * i + a];
                double u = std::floor((cc - (double)lof[a]) / (double)stepf + 0.5);
                if (!(u >= 0.0) || !(u <= 65535.0)) { ok = false; break; }  // (a centre outside the grid box: cannot happen for registered objects)
                q[3 * i + a] = (uint16_t)u;
                const double dec = (double)std::fmaf((float)u, stepf, lof[a]);
                d2 += (cc - dec) * (cc - dec);
            }
            const double w = std::fabs((double)c->h_grid_pre[i]), d = std::sqrt(d2);  // (the ray is not transformed here: no further term)
            const double w2 = (w + d) * (w + d) + 2.0 * d * sa0 * Dbox + a0 * (2.0 * d * Dbox + d * d);
            wq[i] = std::sqrt(w2) * (1.0 + 2e-6);
            rmax = std::max(rmax, wq[i]);
            kmax = std::max(kmax, (double)key[i]);
        }
        const volatile float rstepv = (float)(rmax / 255.0 * (1.0 + 1e-5)), kstepv = (float)(std::max(kmax, 1e-3) / 255.0 * (1.0 + 1e-5));
        const float rstepf = rstepv, kstepf = kstepv;
        // Block indices travel in 24 bits: the walks keep the position inside a block they come back to in bits 24+ of their
        // cursor (rt_grid.h: kLtBlockIndexBits; trace_segment / walk_segment mask with kLtBlockIndexMask). Count the blocks the
        // table will hold EXACTLY - a head per tile plus the further blocks of every chain - and keep the record form beyond
        // that (round 3 checked heads + total / 3 against 2^30, which the masks do not honour: ADVICE r3).
        uint64_t n_lt_blocks = ranges.size();
        for (size_t t = 0; t < ranges.size(); ++t) n_lt_blocks += ranges[t].y > 3u ? (ranges[t].y - 1u) / 3u : 0u;
        ok = ok && rstepf > 0.f && std::isfinite(rstepf) && std::isfinite(kstepf) && rt::light_tile_blocks_fit(n_lt_blocks);
        if (ok) {
            const size_t heads = ranges.size();
            // what an entry's two words hold depends on its object alone: the centre on the lattice, the radius rounded UP and
            // the key rounded DOWN to their 8-bit steps - once per object, not once per entry (1.75 M entries for 100 k objects)
            std::vector<uint2> packed(n, make_uint2(0u, 0u));
            for (uint32_t i = 0; i < n; ++i) {
                if (!(rect[i].u1 >= rect[i].u0)) continue;
                uint32_t r8 = (uint32_t)std::ceil(wq[i] / (double)rstepf);
                while (r8 < 255u && (double)((float)r8 * rstepf) < wq[i]) ++r8;  // (the device's own product must not fall short)
                if (r8 > 255u) r8 = 255u;
                double kk = std::floor(std::max((double)key[i], 0.0) / (double)kstepf * (1.0 - 1e-6));
                uint32_t k8 = (uint32_t)std::min(255.0, std::max(0.0, kk));
                while (k8 > 0u && (double)((float)k8 * kstepf) > (double)key[i]) --k8;  // (rounded DOWN: an entry may only look nearer to the light)
                packed[i] = make_uint2((uint32_t)q[3 * i] | ((uint32_t)q[3 * i + 1] << 16), (uint32_t)q[3 * i + 2] | (r8 << 16) | (k8 << 24));
            }
            // tile t's head is block t; the further blocks of the chains follow the heads in tile order
            std::vector<uint32_t> chain_at(heads + 1, 0u);
            for (size_t t = 0; t < heads; ++t) chain_at[t + 1] = chain_at[t] + (ranges[t].y > 3u ? (ranges[t].y - 1u) / 3u : 0u);
            const size_t n_blocks = heads + chain_at[heads];
            std::vector<uint32_t> blk(8 * n_blocks, 0u), bid(4 * n_blocks, c->n_objs);
            const uint32_t empty_hi = 0xff000000u;
            parallel_for(n_blocks, 1u << 12, [&](size_t b0, size_t b1) {
                for (size_t b = b0; b < b1; ++b) blk[8 * b + 3] = blk[8 * b + 5] = blk[8 * b + 7] = empty_hi;
            });
            parallel_for(heads, 1024, [&](size_t t0, size_t t1) {
                for (size_t t = t0; t < t1; ++t) {
                    size_t at = t, next = heads + chain_at[t];
                    for (uint32_t j = 0; j < ranges[t].y; ++j) {
                        const uint32_t slot = j % 3u;
                        if (j && slot == 0u) {  // the chain goes on in its next block behind the heads
                            blk[8 * at] = (uint32_t)next;
                            at = next++;
                        }
                        const uint32_t i = entries[(size_t)ranges[t].x + j];
                        blk[8 * at + 2 + 2 * slot] = packed[i].x;
                        blk[8 * at + 3 + 2 * slot] = packed[i].y;
                        bid[4 * at + slot] = i;
                    }
                }
            });
            RT_HIP(c, hipMalloc((void**)&c->d_lt_blocks, sizeof(uint32_t) * blk.size()));
            RT_HIP(c, hipMalloc((void**)&c->d_lt_block_ids, sizeof(uint32_t) * bid.size()));
            RT_HIP(c, hipMemcpy(c->d_lt_blocks, blk.data(), sizeof(uint32_t) * blk.size(), hipMemcpyHostToDevice));
            RT_HIP(c, hipMemcpy(c->d_lt_block_ids, bid.data(), sizeof(uint32_t) * bid.size(), hipMemcpyHostToDevice));
            lt.blocks = c->d_lt_blocks;
            lt.block_ids = c->d_lt_block_ids;
            lt.lat_lox = lof[0]; lt.lat_loy = lof[1]; lt.lat_loz = lof[2];
            lt.lat_step = stepf; lt.rstep = rstepf; lt.kstep = kstepf;
            lt.blocks_enabled = 1u;
            lap("blocks + uploads");
            if (std::getenv("RT_WALK_STATS"))
                std::fprintf(stderr, "[light tiles] %u x %u tiles, %llu entries, %zu blocks (%zu behind the heads), lattice step %g, radius step %g, key step %g\n",
                             T, T, (unsigned long long)total, blk.size() / 8, blk.size() / 8 - heads, (double)stepf, (double)rstepf, (double)kstepf);
        }
    }
    if (!lt.blocks_enabled) {
        const int rc = upload_records();
        if (rc != RT_OK) return rc;
        lap("records + uploads");
    }
    if (!c->h_walk.empty() && total && !lt.blocks_enabled) {  // (no blocks - RT_NO_LT_BLOCKS: the lists as records of the unified walk, each tile's chained to its end)
        const uint64_t base = c->h_walk.size() / 2;
        if ((base + total) * 32ull < 0xf0000000ull) {
            c->h_walk.resize(2 * (size_t)(base + total));
            for (size_t t = 0; t < ranges.size(); ++t)
                for (uint32_t j = 0; j < ranges[t].y; ++j) {
                    const size_t k = (size_t)ranges[t].x + j;
                    c->h_walk[2 * (base + k)] = walk_sphere(recs[2 * k]);
                    c->h_walk[2 * (base + k) + 1] = walk_link(entries[k], j + 1u < ranges[t].y ? (uint32_t)(base + k + 1) : 0u, recs[2 * k + 1].x);
                }
            lt.walk_base = (uint32_t)base;
        }
    }
    return RT_OK;
}

}  // namespace

extern "C" {

int rt_abi_version(void) { return RT_ABI_VERSION; }

const char* rt_last_error(const rt_context* ctx) { return ctx ? ctx->error.c_str() : g_create_error.c_str(); }

int rt_create(rt_context** out_ctx, const void* objs, uint32_t n_objs, const void* lights, uint32_t n_lights,
              const void* rays, uint64_t n_rays, uint32_t max_bounces, int kernel, int device, uint32_t flags) {
    g_create_error.clear();
    if (!out_ctx) return fail(nullptr, RT_ERR_INVALID_ARGUMENT, "ctx is NULL");
    *out_ctx = nullptr;
    if (kernel < 0 || kernel > 2) return fail(nullptr, RT_ERR_INVALID_ARGUMENT, "kernel must be 0, 1 or 2");
    if ((n_objs && !objs) || (n_lights && !lights))
        return fail(nullptr, RT_ERR_INVALID_ARGUMENT, "objs/lights is NULL with a non-zero count");
    if (flags & ~(RT_FLAG_UNFUSED | RT_FLAG_LITERAL | RT_FLAG_NO_RAYGEN | RT_FLAG_WAVEFRONT | RT_FLAG_MONOLITHIC | RT_FLAG_NO_GRID | RT_FLAG_FAST_PHONG))
        return fail(nullptr, RT_ERR_INVALID_ARGUMENT, "unknown flag bits");
    if (n_lights >= (1u << 22))  // the large-scene path keeps a pixel's light index in 22 bits of its phase word
        return fail(nullptr, RT_ERR_INVALID_ARGUMENT, "more than 4 194 303 lights");
    if ((flags & RT_FLAG_WAVEFRONT) && (flags & RT_FLAG_MONOLITHIC))
        return fail(nullptr, RT_ERR_INVALID_ARGUMENT, "RT_FLAG_WAVEFRONT and RT_FLAG_MONOLITHIC are exclusive");

    int n_dev = 0;
    hipError_t e = hipGetDeviceCount(&n_dev);
    if (e != hipSuccess || n_dev <= 0)
        return fail(nullptr, RT_ERR_NO_DEVICE, "no HIP device available (this backend has no CPU fallback)");
    if (device < 0 || device >= n_dev) return fail(nullptr, RT_ERR_INVALID_ARGUMENT, "device ordinal out of range");

    rt_context* c = new (std::nothrow) rt_context();
    if (!c) return fail(nullptr, RT_ERR_OUT_OF_MEMORY, "host allocation failed");
    c->device = device;
    c->flags = flags;
    c->kernel = kernel;
    c->n_objs = n_objs;
    c->n_lights = n_lights;
    c->n_rays = n_rays;
    c->n_local = n_rays;
    c->max_bounces = max_bounces;

    StopWatch sw_total, sw;
    int rc = RT_OK;
    auto bail = [&](int code) {
        g_create_error = c->error;
        rt_destroy(c);
        return code;
    };
#define RT_TRY(call)                                      \
    do {                                                  \
        hipError_t e2_ = (call);                          \
        if (e2_ != hipSuccess) { rc = fail_hip(c, e2_, #call); return bail(rc); } \
    } while (0)

    const bool trace = std::getenv("RT_SETUP_TRACE") != nullptr;  // engineering aid: where rt_create's time goes
    StopWatch swt;
    auto lap = [&](const char* what) { if (trace) std::fprintf(stderr, "[rt_create] %-28s %8.2f ms\n", what, swt.lap_ms()); };
    DeviceGuard guard(device);  // restores the caller's current device on every return below
    if (!guard.ok) { rc = fail_hip(c, guard.err, "hipSetDevice"); return bail(rc); }
    RT_TRY(hipStreamCreateWithFlags(&c->stream, hipStreamNonBlocking));
    lap("device + stream");
    // (when rt_create is the process's first HIP call this is where the runtime starts up and the device context is made: ~140 ms
    //  on the test boxes, MEASURED - it stays in create_ms, which is what the caller waits for, but is no part of the upload)
    (void)sw.lap_ms();
    for (uint32_t i = 0; i < kTimingSlots; ++i) {
        RT_TRY(hipEventCreate(&c->ev_begin[i]));
        c->ev_begin_made = i + 1;
        RT_TRY(hipEventCreate(&c->ev_end[i]));
        c->ev_end_made = i + 1;
    }
    RT_TRY(hipMalloc((void**)&c->d_counters, sizeof(rt::Counters)));
    lap("events + counters");

    {
        std::vector<rt::HotPair> pairs;
        std::vector<rt::HotObject> hot;
        std::vector<rt::ColdObject> cold;
        repack_objects(static_cast<const rt_object_data*>(objs), n_objs, pairs, hot, cold);
        c->n_pairs = (uint32_t)pairs.size();
        lap("repack_objects");
        RT_TRY(hipMalloc((void**)&c->d_pairs, sizeof(rt::HotPair) * (pairs.size() + 1)));
        RT_TRY(hipMalloc((void**)&c->d_shadow_pairs, sizeof(rt::HotPair) * (pairs.size() + 1)));
        if (!pairs.empty()) {
            RT_TRY(hipMemcpy(c->d_pairs, pairs.data(), sizeof(rt::HotPair) * pairs.size(), hipMemcpyHostToDevice));
            // shadow stream: same records, objects ordered by decreasing size (stable), re-paired
            const rt_object_data* od = static_cast<const rt_object_data*>(objs);
            std::vector<uint32_t> order(n_objs);
            std::vector<double> size(n_objs);
            for (uint32_t i = 0; i < n_objs; ++i) { order[i] = i; size[i] = size_proxy(od[i]); }
            std::stable_sort(order.begin(), order.end(), [&](uint32_t a, uint32_t b) { return size[a] > size[b]; });
            std::vector<rt::HotPair> spairs;
            pack_pairs(od, order.data(), n_objs, spairs);
            RT_TRY(hipMemcpy(c->d_shadow_pairs, spairs.data(), sizeof(rt::HotPair) * spairs.size(), hipMemcpyHostToDevice));
        }
        lap("pair streams (sort, upload)");
        // one spare record keeps the arrays non-null for n_objs == 0
        RT_TRY(hipMalloc((void**)&c->d_hot, sizeof(rt::HotObject) * (size_t)(n_objs + 1)));
        RT_TRY(hipMalloc((void**)&c->d_cold, sizeof(rt::ColdObject) * (size_t)(n_objs + 1)));
        if (n_objs <= 64) {  // per-bundle culling is only used for scenes this small
            c->h_spheres.resize(4 * (size_t)n_objs);
            for (uint32_t i = 0; i < n_objs; ++i) {
                const Sphere sp = bounding_sphere(static_cast<const rt_object_data*>(objs)[i]);
                c->h_spheres[4 * i] = sp.x; c->h_spheres[4 * i + 1] = sp.y; c->h_spheres[4 * i + 2] = sp.z; c->h_spheres[4 * i + 3] = sp.r;
            }
        }
        RT_TRY(hipMalloc((void**)&c->d_bounds, sizeof(float4) * 65));
        {   // the spare HotObject behind the last one can never be hit (unknown type): records of the unified walk that hold
            // no candidate point at it (GridDesc::walk_none)
            rt::HotObject none{};
            none.type = 0xffffffffu;
            hot.push_back(none);
            RT_TRY(hipMemcpy(c->d_hot, hot.data(), sizeof(rt::HotObject) * (size_t)(n_objs + 1), hipMemcpyHostToDevice));
        }
        if (n_objs) RT_TRY(hipMemcpy(c->d_cold, cold.data(), sizeof(rt::ColdObject) * n_objs, hipMemcpyHostToDevice));
        {   // matrix rows of both directions + absorption side by side (ObjectRecord)
            std::vector<rt::ObjectRecord> rec((size_t)n_objs + 1);
            std::memset(rec.data(), 0, sizeof(rt::ObjectRecord) * rec.size());
            parallel_for(n_objs, 16384, [&](size_t i0, size_t i1) {
                for (size_t i = i0; i < i1; ++i) {
                    rec[i].inv_row[0] = hot[i].row0; rec[i].inv_row[1] = hot[i].row1; rec[i].inv_row[2] = hot[i].row2;
                    rec[i].type = hot[i].type;
                    rec[i].pad0 = hot[i].pad[0];
                    rec[i].absorption = cold[i].amb_absorb.w;
                    for (int r = 0; r < 3; ++r) rec[i].mv_row[r] = cold[i].mv_row[r];
                }
            });
            rec[n_objs].type = 0xffffffffu;
            RT_TRY(hipMalloc((void**)&c->d_objrec, sizeof(rt::ObjectRecord) * rec.size()));
            RT_TRY(hipMemcpy(c->d_objrec, rec.data(), sizeof(rt::ObjectRecord) * rec.size(), hipMemcpyHostToDevice));
        }
    }
    lap("hot / cold / object records");
    RT_TRY(hipMalloc((void**)&c->d_lights, sizeof(rt::LightRec) * (size_t)(n_lights + 1)));
    if (n_lights) RT_TRY(hipMemcpy(c->d_lights, lights, sizeof(rt::LightRec) * n_lights, hipMemcpyHostToDevice));

    for (uint32_t i = 0; i < n_objs; ++i)
        if (static_cast<const rt_object_data*>(objs)[i].type == 2u) { c->has_triangles = true; break; }
    for (uint32_t i = n_objs; i-- > 0;) {
        const uint32_t ty = static_cast<const rt_object_data*>(objs)[i].type;
        if (ty <= 1u) { c->nan_winner = (int)i; c->nan_winner_sphere = (ty == 0u); break; }
    }
    for (uint32_t i = 0; i < n_objs && c->affine_w; ++i) {
        const rt_object_data& o = static_cast<const rt_object_data*>(objs)[i];
        if (o.type == 2u) continue;  // vertices, not matrices
        c->affine_w = o.mv[3] == 0.f && o.mv[7] == 0.f && o.mv[11] == 0.f && o.mv[15] == 1.f && o.mvInverse[3] == 0.f &&
                      o.mvInverse[7] == 0.f && o.mvInverse[11] == 0.f && o.mvInverse[15] == 1.f;
    }
    // An instance that can produce a NaN hit time for a FINITE ray (non-finite or singular rows x,y,z of mvInverse:
    // a scale of 0, garbage) makes the reference's result depend on the ORDER its loop meets the objects in - a NaN
    // time overwrites and is overwritten. The exact eliminations (any-hit shadow rays on a size-sorted stream, the
    // grid) assume finite times, so such scenes are rendered the literal way: every ray, every object, in order.
    if (!(c->flags & RT_FLAG_LITERAL)) {
        std::atomic<bool> degenerate{false};
        parallel_for(n_objs, 8192, [&](size_t i0, size_t i1) {
            for (size_t i = i0; i < i1 && !degenerate.load(std::memory_order_relaxed); ++i) {
                const rt_object_data& o = static_cast<const rt_object_data*>(objs)[i];
                if (o.type <= 1u && !std::isfinite(object_bound(o).r)) degenerate.store(true, std::memory_order_relaxed);
            }
        });
        if (degenerate.load()) { c->flags |= RT_FLAG_LITERAL; c->forced_literal = true; }
    }
    lap("instance checks");
    c->base_flags = c->flags;
    if (rays && n_rays) {
        const rt_ray* r = static_cast<const rt_ray*>(rays);
        uint32_t W = 0, H = 0;
        float z = 0.f;
        if (!(flags & RT_FLAG_NO_RAYGEN) && detect_pinhole(r, n_rays, W, H, z)) {
            c->pinhole = true;
            c->width = W;
            c->height = H;
            c->z = z;
            c->camera_out_of_domain = !camera_in_domain(W, H, z);
        } else {
            bool w0 = true;
            for (uint64_t i = 0; i < n_rays && w0; ++i) w0 = (r[i].direction[3] == 0.0f);
            c->dir_w_zero = w0;
            for (uint64_t i = 0; i < n_rays && !c->rays_out_of_domain; ++i)
                c->rays_out_of_domain = !direction_in_domain(r[i].direction[0], r[i].direction[1], r[i].direction[2]);
            for (uint64_t i = 0; i < n_rays; ++i) {  // where do primary rays start? (grid margins need it)
                if (r[i].start[3] != 1.0f || !std::isfinite(r[i].start[0] + r[i].start[1] + r[i].start[2])) { c->primary_w_one = false; break; }
                for (int a = 0; a < 3; ++a) {
                    c->origin_lo[a] = std::min(c->origin_lo[a], (double)r[i].start[a]);
                    c->origin_hi[a] = std::max(c->origin_hi[a], (double)r[i].start[a]);
                }
            }
            RT_TRY(hipMalloc((void**)&c->d_rays, sizeof(rt_ray) * (size_t)n_rays));
            RT_TRY(hipMemcpy(c->d_rays, rays, sizeof(rt_ray) * (size_t)n_rays, hipMemcpyHostToDevice));
            c->have_rays = true;
        }
    }
#undef RT_TRY
    apply_ray_domain(c);
    lap("ray scan + upload");
    (void)hipDeviceSynchronize();
    lap("device synchronise");
    c->setup.upload_ms = sw.lap_ms();
    if (n_objs >= kWavefrontGridMinObjects || (flags & RT_FLAG_WAVEFRONT) || c->has_triangles) {
        rc = build_grid(c, static_cast<const rt_object_data*>(objs), n_objs);
        if (rc != RT_OK) return bail(rc);
        c->setup.grid_ms = sw.lap_ms() - c->setup.blocks_ms;  // (build_grid ends with the block grid, which times itself)
        rc = build_light_tiles(c, static_cast<const rt_light*>(lights));
        if (rc != RT_OK) return bail(rc);
        c->setup.light_tiles_ms = sw.lap_ms();
        rc = upload_walk_records(c);
        if (rc != RT_OK) return bail(rc);
        c->setup.grid_ms += sw.lap_ms();
    }
    if (c->has_triangles && (!c->grid.enabled || (c->flags & (RT_FLAG_LITERAL | RT_FLAG_MONOLITHIC | RT_FLAG_NO_GRID)))) {
        fail(c, RT_ERR_INVALID_ARGUMENT,
             "triangle records (type 2, an extension of the reference's two primitives) are traced by the grid path only: "
             "it needs affine instances, ray w = 1, primary directions with 1e-30 < |d|^2 < 1e30, and none of RT_FLAG_LITERAL / "
             "RT_FLAG_MONOLITHIC / RT_FLAG_NO_GRID");
        return bail(RT_ERR_INVALID_ARGUMENT);
    }
    c->setup.create_ms = sw_total.lap_ms();
    *out_ctx = c;
    return RT_OK;
}

int rt_get_setup_times(rt_context* c, rt_setup_times_t* t) {
    if (!c || !t) return RT_ERR_INVALID_ARGUMENT;
    *t = c->setup;
    return RT_OK;
}

int rt_set_camera(rt_context* c, uint32_t width, uint32_t height, float z) {
    if (!c) return RT_ERR_INVALID_ARGUMENT;
    if (width == 0 || height == 0 || (uint64_t)width * height != c->n_rays)
        return fail(c, RT_ERR_INVALID_ARGUMENT, "width*height must equal n_rays");
    if (width > 0x1000000u || height > 0x1000000u) return fail(c, RT_ERR_INVALID_ARGUMENT, "grid too large");
    const bool out = !camera_in_domain(width, height, z);
    if (out && c->has_triangles)
        return fail(c, RT_ERR_INVALID_ARGUMENT, "a camera with a direction of |d|^2 outside (1e-30, 1e30) needs the literal loops, which do not know triangle records");
    c->camera_out_of_domain = out;
    c->pinhole = true;
    apply_ray_domain(c);
    c->width = width;
    c->height = height;
    c->z = z;
    c->rects_dirty = true;
    c->tiles_dirty = true;
    return RT_OK;
}

int rt_set_shard(rt_context* c, uint64_t tile_rays, uint32_t rank, uint32_t world) {
    if (!c) return RT_ERR_INVALID_ARGUMENT;
    if (world == 0 || rank >= world || (world > 1 && tile_rays == 0))
        return fail(c, RT_ERR_INVALID_ARGUMENT, "need world >= 1, rank < world, tile_rays > 0");
    c->tile_rays = tile_rays;
    c->rank = rank;
    c->world = world;
    c->n_local = local_count(c->n_rays, tile_rays, rank, world);
    return RT_OK;
}

uint64_t rt_local_rays(const rt_context* c) { return c ? c->n_local : 0; }

int rt_set_aux_device(rt_context* c, void* d_hit_t, void* d_hit_index) {
    if (!c) return RT_ERR_INVALID_ARGUMENT;
    c->aux_t = static_cast<float*>(d_hit_t);
    c->aux_index = static_cast<int32_t*>(d_hit_index);
    return RT_OK;
}

int rt_render_device(rt_context* c, void* d_out, void* hip_stream) {
    if (!c) return RT_ERR_INVALID_ARGUMENT;
    if (!d_out && c->n_local) return fail(c, RT_ERR_INVALID_ARGUMENT, "d_out is NULL");
    // NULL is the legacy default stream - NOT the context's private stream: a caller that passes its framework's
    // "current stream" handle (0 for torch's default stream) gets a render that is ordered with its own work
    return do_launch(c, d_out, static_cast<hipStream_t>(hip_stream), false);
}

// The synchronous Render() of a LARGE frame, in passes: the frame is cut into interleaved 16-row tiles as for several GPUs
// (rt_set_shard's partition; a pass stands for consecutive ranks - RenderParams::run_rays), pass 0 renders three tiles of every
// four, pass 1 the fourth, and pass 0's tiles travel to the pinned host frame - one strided device-to-host copy on a stream of its
// own - WHILE pass 1 renders; only the last quarter's copy is left behind the kernels. The blocking read-back of 268 MB
// (OpenCLRaytracer.cpp:94) is 4.9 ms behind an 11.4 ms cfg4 render. Measured, cfg4 (tools/ab/render_split.py,
// profiles/r04_experiments/render_split*.txt): one pass 16.5 ms, "1,1" (round 4's first form) 14.8, "3,1" 13.55, "2,1" 13.9,
// "5,2,1" 13.8, "7,1" 15.1 - an unequal split wins because a small pass renders less efficiently than a large one (a quarter of the
// frame takes 3.5 ms, not 2.85) while the copy it hides is proportional to the pass before it: 3/4 of the copy (3.7 ms) fits behind
// the last quarter's render. Only for frames of the large-scene path with >= 4 M rays that the caller has not sharded himself;
// RT_RENDER_PASSES=1 switches it off, RT_RENDER_SPLIT="a,b,.." chooses another split. The pixels are the one-pass frame's, bit for
// bit (a shard is the same arithmetic on a subset of the rays).
static int render_in_passes(rt_context* c, const float** out) {
    // the split: spans of consecutive ranks of a world of their sum ("3,1": three tiles of every four, then the fourth)
    uint32_t spans[kMaxPasses] = {3, 1, 0, 0}, K = 2, world = 0;
    if (const char* env = std::getenv("RT_RENDER_SPLIT")) {
        K = 0;
        for (const char* q = env; *q && K < kMaxPasses;) {
            const long v = std::strtol(q, const_cast<char**>(&q), 10);
            if (v <= 0 || v > 64) { K = 0; break; }
            spans[K++] = (uint32_t)v;
            if (*q == ',') ++q;
        }
        if (K == 0) return fail(c, RT_ERR_INVALID_ARGUMENT, "RT_RENDER_SPLIT: up to 4 comma-separated spans of 1..64 tiles");
    }
    for (uint32_t k = 0; k < K; ++k) world += spans[k];
    const uint64_t n_rays = c->n_rays;
    const uint64_t tile_rays = (c->pinhole && c->width) ? 16ull * c->width : 65536ull;
    const uint64_t tiles = (n_rays + tile_rays - 1) / tile_rays;
    const size_t elem = elem_bytes(c), tile_bytes = (size_t)tile_rays * elem;
    const size_t frame_bytes = (size_t)tiles * tile_bytes;  // whole tiles: the ragged last one is padded behind the frame's end
    if (frame_bytes > c->h_out_bytes) {
        if (c->h_out) (void)hipHostFree(c->h_out);
        c->h_out = nullptr;
        c->h_out_bytes = 0;
        RT_HIP(c, hipHostMalloc(&c->h_out, frame_bytes, hipHostMallocDefault));
        c->h_out_bytes = frame_bytes;
    }
    if (frame_bytes > c->d_out_bytes) {  // the passes' outputs one behind the other
        if (c->d_out) (void)hipFree(c->d_out);
        c->d_out = nullptr;
        c->d_out_bytes = 0;
        RT_HIP(c, hipMalloc(&c->d_out, frame_bytes));
        c->d_out_bytes = frame_bytes;
    }
    if (!c->copy_stream) RT_HIP(c, hipStreamCreateWithFlags(&c->copy_stream, hipStreamNonBlocking));
    for (uint32_t k = 0; k < K; ++k)
        if (!c->ev_pass[k]) RT_HIP(c, hipEventCreateWithFlags(&c->ev_pass[k], hipEventDisableTiming));
    int rc = RT_OK;
    size_t at = 0;        // this pass's output within d_out
    uint32_t rank = 0;    // its first rank
    const uint64_t groups = tiles / world, rest = tiles % world;
    for (uint32_t k = 0; k < K && rc == RT_OK; rank += spans[k], ++k) {
        c->tile_rays = tile_rays;
        c->rank = rank;
        c->world = world;
        c->span = spans[k];
        c->n_local = local_count(n_rays, tile_rays, rank, world, spans[k]);
        char* buf = static_cast<char*>(c->d_out) + at;
        const size_t run_bytes = (size_t)spans[k] * tile_bytes;
        at += (size_t)c->n_local * elem;
        if (c->n_local == 0) continue;
        rc = do_launch(c, buf, c->stream, false);
        if (rc != RT_OK) break;
        hipError_t e = hipEventRecord(c->ev_pass[k], c->stream);
        if (e == hipSuccess) e = hipStreamWaitEvent(c->copy_stream, c->ev_pass[k], 0);
        char* dst = static_cast<char*>(c->h_out) + (size_t)rank * tile_bytes;
        if (e == hipSuccess && groups)  // run j of this pass is tiles j * world + rank ... of the frame
            e = hipMemcpy2DAsync(dst, (size_t)world * tile_bytes, buf, run_bytes, run_bytes, (size_t)groups, hipMemcpyDeviceToHost, c->copy_stream);
        if (e == hipSuccess && rest > rank)  // the short run of the frame's last, incomplete group of tiles
            e = hipMemcpyAsync(dst + (size_t)groups * world * tile_bytes, buf + (size_t)groups * run_bytes,
                               (size_t)std::min<uint64_t>(rest - rank, spans[k]) * tile_bytes, hipMemcpyDeviceToHost, c->copy_stream);
        if (e != hipSuccess) rc = fail_hip(c, e, "read-back of a pass");
    }
    c->tile_rays = 0;
    c->rank = 0;
    c->world = 1;
    c->span = 1;
    c->n_local = n_rays;
    hipError_t e = hipStreamSynchronize(c->stream);
    const hipError_t e2 = hipStreamSynchronize(c->copy_stream);  // Render() is synchronous (OpenCLRaytracer.cpp:94)
    if (rc != RT_OK) return rc;
    if (e == hipSuccess) e = e2;
    if (e != hipSuccess) return fail_hip(c, e, "hipStreamSynchronize");
    *out = static_cast<const float*>(c->h_out);
    return RT_OK;
}

int rt_render(rt_context* c, const float** out) {
    if (!c || !out) return RT_ERR_INVALID_ARGUMENT;
    RT_DEVICE(c);
    {
        const char* env = std::getenv("RT_RENDER_PASSES");  // "1": one pass whatever the frame; "2": two passes whatever its size (tests)
        const bool off = env && env[0] == '1', forced = env && env[0] == '2';
        if (!off && c->world <= 1 && (forced || c->n_rays >= (1ull << 22)) && c->n_rays > 0 && c->n_local == c->n_rays && use_wavefront(c) &&
            (c->pinhole || c->have_rays) && !c->aux_t && !c->aux_index)  // (aux buffers are indexed by work-item of ONE whole-frame launch)
            return render_in_passes(c, out);
    }
    int rc = ensure_out(c);
    if (rc) return rc;
    rc = ensure_host_out(c);
    if (rc) return rc;
    rc = do_launch(c, c->d_out, c->stream, false);
    if (rc) return rc;
    const size_t bytes = (size_t)c->n_local * elem_bytes(c);
    if (bytes) RT_HIP(c, hipMemcpyAsync(c->h_out, c->d_out, bytes, hipMemcpyDeviceToHost, c->stream));
    RT_HIP(c, hipStreamSynchronize(c->stream));  // Render() is synchronous (OpenCLRaytracer.cpp:94)
    *out = static_cast<const float*>(c->h_out);
    return RT_OK;
}

int rt_render_aux(rt_context* c, float* hit_t, int32_t* hit_index) {
    if (!c) return RT_ERR_INVALID_ARGUMENT;
    int rc = ensure_out(c);
    if (rc) return rc;
    float* d_t = nullptr;
    int32_t* d_i = nullptr;
    const size_t n = (size_t)c->n_local;
    RT_DEVICE(c);
    if (hit_t) RT_HIP(c, hipMalloc((void**)&d_t, n ? n * 4 : 4));
    if (hit_index) {
        hipError_t e = hipMalloc((void**)&d_i, n ? n * 4 : 4);
        if (e != hipSuccess) { if (d_t) (void)hipFree(d_t); return fail_hip(c, e, "hipMalloc aux"); }
    }
    c->aux_t = d_t;
    c->aux_index = d_i;
    rc = do_launch(c, c->d_out, c->stream, false);
    hipError_t e = hipSuccess;
    if (rc == RT_OK) e = hipStreamSynchronize(c->stream);
    if (rc == RT_OK && e == hipSuccess && hit_t && n) e = hipMemcpy(hit_t, d_t, n * 4, hipMemcpyDeviceToHost);
    if (rc == RT_OK && e == hipSuccess && hit_index && n) e = hipMemcpy(hit_index, d_i, n * 4, hipMemcpyDeviceToHost);
    if (d_t) (void)hipFree(d_t);
    if (d_i) (void)hipFree(d_i);
    if (rc) return rc;
    if (e != hipSuccess) return fail_hip(c, e, "aux read-back");
    return RT_OK;
}

int rt_count_rays(rt_context* c) {
    if (!c) return RT_ERR_INVALID_ARGUMENT;
    RT_DEVICE(c);
    int rc = ensure_out(c);
    if (rc) return rc;
    rc = do_launch(c, c->d_out, c->stream, true);
    if (rc) return rc;
    if (c->n_local == 0) return RT_OK;
    RT_HIP(c, hipStreamSynchronize(c->stream));
    RT_HIP(c, hipMemcpy(&c->counters, c->d_counters, sizeof(rt::Counters), hipMemcpyDeviceToHost));
    if (std::getenv("RT_WALK_STATS")) {  // engineering aid: what the grid walk did in the counted frame
        if (c->grid.enabled) std::fprintf(stderr, "[grid] %d x %d x %d cells, edge %g\n", c->grid.nx, c->grid.ny, c->grid.nz, (double)c->grid.cell);
        static const char* names[8] = {"rays", "wave trips", "live lane-trips", "cell fetches", "pre-tests", "exact tests", "exact rounds", "hand-out rounds"};
        for (int k = 0; k < 2; ++k) {
            const unsigned long long* v = c->counters.walk[k];
            if (!v[0]) continue;
            std::fprintf(stderr, "[walk %s]", k ? "any" : "closest");
            for (int j = 0; j < 8; ++j) std::fprintf(stderr, " %s %llu", names[j], v[j]);
            std::fprintf(stderr, " | per ray: fetches %.2f pre-tests %.2f exact %.2f lane-trips %.2f | live lanes/trip %.1f\n",
                         (double)v[3] / v[0], (double)v[4] / v[0], (double)v[5] / v[0], (double)v[2] / v[0], (double)v[2] / (v[1] ? v[1] : 1));
        }
    }
    return RT_OK;
}

int rt_get_stats(rt_context* c, rt_stats_t* s) {
    if (!c || !s) return RT_ERR_INVALID_ARGUMENT;
    RT_DEVICE(c);
    if (c->ev_count) {
        const uint32_t slot = (c->ev_count - 1) % kTimingSlots;
        RT_HIP(c, hipEventSynchronize(c->ev_end[slot]));
        RT_HIP(c, hipEventElapsedTime(&c->last_ms, c->ev_begin[slot], c->ev_end[slot]));
    }
    s->rays_traced = c->counters.traced;
    s->rays_reference = c->counters.reference;
    s->hit_pixels = c->counters.hits;
    s->last_kernel_ms = c->last_ms;
    s->pinhole = c->pinhole ? 1u : 0u;
    s->width = c->pinhole ? c->width : 0;
    s->height = c->pinhole ? c->height : 0;
    s->local_rays = c->n_local;
    s->wavefront = c->last_wavefront ? 1u : 0u;
    s->rounds = c->last_rounds;
    s->object_tests = c->counters.tests;
    return RT_OK;
}

int rt_timing_reset(rt_context* c) {
    if (!c) return RT_ERR_INVALID_ARGUMENT;
    c->ev_count = 0;
    return RT_OK;
}

int rt_timing_summary(rt_context* c, double* sum_ms, uint32_t* launches) {
    if (!c || !sum_ms || !launches) return RT_ERR_INVALID_ARGUMENT;
    RT_DEVICE(c);
    const uint32_t n = c->ev_count < kTimingSlots ? c->ev_count : kTimingSlots;
    double total = 0.0;
    for (uint32_t i = 0; i < n; ++i) {
        float ms = 0.f;
        RT_HIP(c, hipEventSynchronize(c->ev_end[i]));
        RT_HIP(c, hipEventElapsedTime(&ms, c->ev_begin[i], c->ev_end[i]));
        total += ms;
    }
    *sum_ms = total;
    *launches = n;
    return RT_OK;
}

void rt_destroy(rt_context* c) {
    if (!c) return;
    DeviceGuard guard(c->device);
    if (c->stream) (void)hipStreamSynchronize(c->stream);
    if (c->d_pairs) (void)hipFree(c->d_pairs);
    if (c->d_shadow_pairs) (void)hipFree(c->d_shadow_pairs);
    if (c->d_hot) (void)hipFree(c->d_hot);
    if (c->d_cold) (void)hipFree(c->d_cold);
    if (c->d_objrec) (void)hipFree(c->d_objrec);
    if (c->d_bounds) (void)hipFree(c->d_bounds);
    if (c->d_grid_cell_range) (void)hipFree(c->d_grid_cell_range);
    if (c->d_grid_cell_rec) (void)hipFree(c->d_grid_cell_rec);
    if (c->d_grid_entries) (void)hipFree(c->d_grid_entries);
    if (c->d_grid_always) (void)hipFree(c->d_grid_always);
    if (c->d_grid_entry_sphere) (void)hipFree(c->d_grid_entry_sphere);
    if (c->d_tile_start) (void)hipFree(c->d_tile_start);
    if (c->d_tile_entries) (void)hipFree(c->d_tile_entries);
    if (c->d_lt_range) (void)hipFree(c->d_lt_range);
    if (c->d_lt_records) (void)hipFree(c->d_lt_records);
    if (c->d_lt_blocks) (void)hipFree(c->d_lt_blocks);
    if (c->d_lt_block_ids) (void)hipFree(c->d_lt_block_ids);
    if (c->d_walk_rec) (void)hipFree(c->d_walk_rec);
    if (c->d_walk_blocks) (void)hipFree(c->d_walk_blocks);
    if (c->d_walk_ids) (void)hipFree(c->d_walk_ids);
    if (c->d_lights) (void)hipFree(c->d_lights);
    if (c->d_rays) (void)hipFree(c->d_rays);
    if (c->d_out) (void)hipFree(c->d_out);
    if (c->h_out) (void)hipHostFree(c->h_out);
    if (c->copy_stream) (void)hipStreamDestroy(c->copy_stream);
    for (hipEvent_t ev : c->ev_pass) if (ev) (void)hipEventDestroy(ev);
    if (c->d_counters) (void)hipFree(c->d_counters);
    free_wavefront(c);
    for (uint32_t i = 0; i < c->ev_begin_made; ++i) (void)hipEventDestroy(c->ev_begin[i]);
    for (uint32_t i = 0; i < c->ev_end_made; ++i) (void)hipEventDestroy(c->ev_end[i]);
    if (c->stream) (void)hipStreamDestroy(c->stream);
    delete c;
}

}  // extern "C"

// ---- several GPUs from one process (hip_raytracer.h: rt_create_multi ...) ------------------------------------------------
struct rt_multi {
    std::vector<rt_context*> ctx;
    std::vector<int> devices;
    std::vector<void*> d_local;      // per context: its packed tiles, on its own device
    std::vector<char> peer_ok;       // per context: devices[0] and its device can address each other's memory
    uint64_t n_rays = 0, tile_rays = 0, tiles = 0;
    size_t elem = 16;
    void* h_frame = nullptr;         // rt_render_multi's frame: pinned, portable host memory (whole tiles) every device copies its tiles into
    std::string error;
    // One host thread per further shard, alive from rt_create_multi to rt_destroy_multi (round 3 created and joined n - 1
    // threads per frame). A frame = one job: every worker renders its shard and puts its tiles in place, the calling thread
    // does shard 0 and waits for the others.
    std::vector<std::thread> workers;
    std::mutex mu;
    std::condition_variable cv_go, cv_done;
    uint64_t generation = 0;         // bumped per job
    uint32_t pending = 0;            // workers that have not finished the current job
    bool quit = false;
    void* job_target = nullptr;      // where the tiles go: a frame on devices[0], or the pinned host frame
    bool job_to_host = false;
    std::vector<int> rcs;
    std::vector<std::string> errs;
};

namespace {

thread_local std::string g_multi_error;

int multi_fail(rt_multi* m, int code, const std::string& msg) {
    if (m) m->error = msg;
    else g_multi_error = msg;
    return code;
}

// one shard: render on the context's own stream, then put its tiles where they belong - in the frame on devices[0]
// (rt_render_multi_device) or STRAIGHT in the pinned host frame (rt_render_multi: the reference's blocking read-back,
// OpenCLRaytracer.cpp:94, over every GPU's own PCIe link at once instead of a hop to devices[0] and one link for the lot)
int multi_render_shard(rt_multi* m, uint32_t r, void* frame, bool to_host, std::string& err) {
    rt_context* c = m->ctx[r];
    DeviceGuard guard(c->device);
    if (!guard.ok) { err = std::string("hipSetDevice: ") + hipGetErrorString(guard.err); return RT_ERR_HIP; }
    int rc = rt_render_device(c, m->d_local[r], c->stream);
    if (rc != RT_OK) { err = c->error; return rc; }
    const uint32_t n = (uint32_t)m->ctx.size();
    const uint64_t mine = m->tiles / n + ((m->tiles % n) > r ? 1 : 0);
    const size_t tile_bytes = (size_t)m->tile_rays * m->elem;
    hipError_t e = hipSuccess;
    if (mine) {
        // tile j of this shard is tile j * n + r of the frame: one strided copy
        char* dst = static_cast<char*>(frame) + (size_t)r * tile_bytes;
        if (to_host) {
            e = hipMemcpy2DAsync(dst, (size_t)n * tile_bytes, m->d_local[r], tile_bytes, tile_bytes, (size_t)mine, hipMemcpyDeviceToHost, c->stream);
        } else if (c->device == m->devices[0] || m->peer_ok[r]) {
            e = hipMemcpy2DAsync(dst, (size_t)n * tile_bytes, m->d_local[r], tile_bytes, tile_bytes, (size_t)mine, hipMemcpyDeviceToDevice, c->stream);
        } else {
            for (uint64_t j = 0; j < mine && e == hipSuccess; ++j)
                e = hipMemcpyPeerAsync(dst + (size_t)j * n * tile_bytes, m->devices[0], static_cast<char*>(m->d_local[r]) + (size_t)j * tile_bytes,
                                       c->device, tile_bytes, c->stream);
        }
    }
    if (e == hipSuccess) e = hipStreamSynchronize(c->stream);
    if (e != hipSuccess) { err = std::string("tile placement: ") + hipGetErrorString(e); return RT_ERR_HIP; }
    return RT_OK;
}

void multi_worker(rt_multi* m, uint32_t r) {
    uint64_t seen = 0;
    for (;;) {
        void* target;
        bool to_host;
        {
            std::unique_lock<std::mutex> lk(m->mu);
            m->cv_go.wait(lk, [&] { return m->quit || m->generation != seen; });
            if (m->quit) return;
            seen = m->generation;
            target = m->job_target;
            to_host = m->job_to_host;
        }
        std::string err;
        const int rc = multi_render_shard(m, r, target, to_host, err);
        {
            std::lock_guard<std::mutex> lk(m->mu);
            m->rcs[r] = rc;
            m->errs[r] = err;
            if (--m->pending == 0) m->cv_done.notify_all();
        }
    }
}

// every shard renders and places its tiles; returns when the frame is complete
int multi_run_frame(rt_multi* m, void* target, bool to_host) {
    const uint32_t n = (uint32_t)m->ctx.size();
    {
        std::lock_guard<std::mutex> lk(m->mu);
        m->job_target = target;
        m->job_to_host = to_host;
        m->pending = (uint32_t)m->workers.size();
        m->generation += 1;
    }
    m->cv_go.notify_all();
    std::string err0;
    const int rc0 = multi_render_shard(m, 0, target, to_host, err0);
    {
        std::unique_lock<std::mutex> lk(m->mu);
        m->cv_done.wait(lk, [&] { return m->pending == 0; });
        m->rcs[0] = rc0;
        m->errs[0] = err0;
    }
    for (uint32_t r = 0; r < n; ++r)
        if (m->rcs[r] != RT_OK) return multi_fail(m, m->rcs[r], "shard " + std::to_string(r) + ": " + m->errs[r]);
    return RT_OK;
}

}  // namespace

extern "C" {

const char* rt_multi_last_error(const rt_multi* m) { return m ? m->error.c_str() : g_multi_error.c_str(); }

void rt_destroy_multi(rt_multi* m) {
    if (!m) return;
    {
        std::lock_guard<std::mutex> lk(m->mu);
        m->quit = true;
    }
    m->cv_go.notify_all();
    for (std::thread& t : m->workers) t.join();
    for (size_t r = 0; r < m->ctx.size(); ++r) {
        if (m->ctx[r] && r < m->d_local.size() && m->d_local[r]) {
            DeviceGuard guard(m->ctx[r]->device);
            (void)hipFree(m->d_local[r]);
        }
        rt_destroy(m->ctx[r]);
    }
    if (!m->devices.empty()) {
        DeviceGuard guard(m->devices[0]);
        if (m->h_frame) (void)hipHostFree(m->h_frame);
    }
    delete m;
}

int rt_create_multi(rt_multi** out, const void* objs, uint32_t n_objs, const void* lights, uint32_t n_lights, const void* rays,
                    uint64_t n_rays, uint32_t max_bounces, int kernel, const int* devices, uint32_t n_devices, uint64_t tile_rays,
                    uint32_t flags) {
    g_multi_error.clear();
    if (!out) return multi_fail(nullptr, RT_ERR_INVALID_ARGUMENT, "m is NULL");
    *out = nullptr;
    if (!devices || n_devices == 0 || n_devices > 64) return multi_fail(nullptr, RT_ERR_INVALID_ARGUMENT, "need 1..64 device ordinals");
    rt_multi* m = new (std::nothrow) rt_multi();
    if (!m) return multi_fail(nullptr, RT_ERR_OUT_OF_MEMORY, "host allocation failed");
    m->devices.assign(devices, devices + n_devices);
    m->ctx.assign(n_devices, nullptr);
    m->d_local.assign(n_devices, nullptr);
    m->peer_ok.assign(n_devices, 0);
    m->n_rays = n_rays;
    m->elem = kernel == RT_KERNEL_HITTEST ? sizeof(float) : 4 * sizeof(float);
    // the contexts are built side by side: each one's grid / tile builders run on a host thread of their own
    std::vector<int> rcs(n_devices, RT_OK);
    std::vector<std::string> errs(n_devices);
    {
        std::vector<std::thread> workers;
        for (uint32_t r = 0; r < n_devices; ++r)
            workers.emplace_back([&, r]() {
                rcs[r] = rt_create(&m->ctx[r], objs, n_objs, lights, n_lights, rays, n_rays, max_bounces, kernel, m->devices[r], flags);
                if (rcs[r] != RT_OK) errs[r] = rt_last_error(nullptr);
            });
        for (std::thread& t : workers) t.join();
    }
    for (uint32_t r = 0; r < n_devices; ++r)
        if (rcs[r] != RT_OK) {
            const int rc = multi_fail(nullptr, rcs[r], "context " + std::to_string(r) + " (device " + std::to_string(m->devices[r]) + "): " + errs[r]);
            rt_destroy_multi(m);
            return rc;
        }
    // tiles: the caller's, or row-tiles of 16 rows when the rays are the pinhole grid, else 65 536 rays
    if (tile_rays == 0) {
        const rt_context* c0 = m->ctx[0];
        tile_rays = c0->pinhole && c0->width ? 16ull * c0->width : 65536ull;
    }
    m->tile_rays = tile_rays;
    m->tiles = (n_rays + tile_rays - 1) / tile_rays;
    for (uint32_t r = 0; r < n_devices; ++r) {
        rt_context* c = m->ctx[r];
        int rc = rt_set_shard(c, tile_rays, r, n_devices);
        hipError_t e = hipSuccess;
        if (rc == RT_OK) {
            DeviceGuard guard(c->device);
            const size_t bytes = (size_t)c->n_local * m->elem;
            e = hipMalloc(&m->d_local[r], bytes ? bytes : 16);
            if (e == hipSuccess && c->device != m->devices[0]) {  // both directions; "already enabled" is fine
                int can = 0;
                if (hipDeviceCanAccessPeer(&can, c->device, m->devices[0]) == hipSuccess && can) {
                    const hipError_t pe = hipDeviceEnablePeerAccess(m->devices[0], 0);
                    if (pe == hipSuccess || pe == hipErrorPeerAccessAlreadyEnabled) m->peer_ok[r] = 1;
                    (void)hipGetLastError();
                }
            }
        }
        if (rc != RT_OK || e != hipSuccess) {
            const int code = multi_fail(nullptr, rc != RT_OK ? rc : RT_ERR_HIP, rc != RT_OK ? c->error : std::string("hipMalloc: ") + hipGetErrorString(e));
            rt_destroy_multi(m);
            return code;
        }
    }
    m->rcs.assign(n_devices, RT_OK);
    m->errs.assign(n_devices, std::string());
    for (uint32_t r = 1; r < n_devices; ++r) m->workers.emplace_back(multi_worker, m, r);
    *out = m;
    return RT_OK;
}

int rt_set_camera_multi(rt_multi* m, uint32_t width, uint32_t height, float z) {
    if (!m) return RT_ERR_INVALID_ARGUMENT;
    for (rt_context* c : m->ctx) {
        const int rc = rt_set_camera(c, width, height, z);
        if (rc != RT_OK) return multi_fail(m, rc, c->error);
    }
    return RT_OK;
}

uint64_t rt_multi_frame_elems(const rt_multi* m) { return m ? m->tiles * m->tile_rays : 0; }

rt_context* rt_multi_context(rt_multi* m, uint32_t r) { return (m && r < m->ctx.size()) ? m->ctx[r] : nullptr; }

int rt_render_multi_device(rt_multi* m, void* d_frame) {
    if (!m) return RT_ERR_INVALID_ARGUMENT;
    if (!d_frame && m->n_rays) return multi_fail(m, RT_ERR_INVALID_ARGUMENT, "d_frame is NULL");
    return multi_run_frame(m, d_frame, false);
}

int rt_render_multi(rt_multi* m, const float** out) {
    if (!m || !out) return RT_ERR_INVALID_ARGUMENT;
    if (!m->h_frame) {
        // whole tiles (the last one may be ragged: its padding work-items are written like pixels), pinned and PORTABLE: every
        // device of the node copies into it
        DeviceGuard guard(m->devices[0]);
        if (!guard.ok) return multi_fail(m, RT_ERR_HIP, std::string("hipSetDevice: ") + hipGetErrorString(guard.err));
        const size_t frame_bytes = (size_t)rt_multi_frame_elems(m) * m->elem;
        const hipError_t e = hipHostMalloc(&m->h_frame, frame_bytes ? frame_bytes : 16, hipHostMallocPortable);
        if (e != hipSuccess) return multi_fail(m, e == hipErrorOutOfMemory ? RT_ERR_OUT_OF_MEMORY : RT_ERR_HIP, std::string("host frame: ") + hipGetErrorString(e));
    }
    const int rc = multi_run_frame(m, m->h_frame, true);  // Render() is synchronous (OpenCLRaytracer.cpp:94): every shard has waited for its copy
    if (rc != RT_OK) return rc;
    *out = static_cast<const float*>(m->h_frame);
    return RT_OK;
}

/* the counters of the last counted render summed over the shards, the slowest shard's kernel time */
int rt_get_stats_multi(rt_multi* m, rt_stats_t* out) {
    if (!m || !out) return RT_ERR_INVALID_ARGUMENT;
    rt_stats_t sum;
    std::memset(&sum, 0, sizeof(sum));
    for (size_t r = 0; r < m->ctx.size(); ++r) {
        rt_stats_t s;
        const int rc = rt_get_stats(m->ctx[r], &s);
        if (rc != RT_OK) return multi_fail(m, rc, m->ctx[r]->error);
        if (r == 0) sum = s;
        else {
            sum.rays_traced += s.rays_traced;
            sum.rays_reference += s.rays_reference;
            sum.hit_pixels += s.hit_pixels;
            sum.object_tests += s.object_tests;
            sum.local_rays += s.local_rays;
            sum.last_kernel_ms = std::max(sum.last_kernel_ms, s.last_kernel_ms);
            sum.rounds = std::max(sum.rounds, s.rounds);
            sum.wavefront = sum.wavefront | s.wavefront;
        }
    }
    *out = sum;
    return RT_OK;
}

int rt_count_rays_multi(rt_multi* m) {
    if (!m) return RT_ERR_INVALID_ARGUMENT;
    for (rt_context* c : m->ctx) {
        const int rc = rt_count_rays(c);
        if (rc != RT_OK) return multi_fail(m, rc, c->error);
    }
    return RT_OK;
}

}  // extern "C"
