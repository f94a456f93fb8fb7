// rt_wavefront.hip - the large-N path: traversal and shading in separate gfx950 kernels.
//
// With thousands of objects a ray spends >99.9 % of its instructions in the object loop. Inside the
// monolithic per-pixel kernel (rt_kernels.hip) that loop drags the whole shading state along (151 VGPRs -> 3
// waves per SIMD); as its own kernel it needs 48 VGPRs and runs 1.5-2x faster per ray-object test. So for
// big scenes a pixel becomes a small resumable state machine whose state lives in HBM (SoA, ~180 B/pixel):
//
//   wf_begin    : primary ray of every work-item -> ray slot, queued for a closest-hit trace
//   wf_trace_*  : lean traversal kernels over a queue of pixel ids (closest hit / any hit with t < 1)
//   wf_resume   : consumes each pixel's trace result, advances its state machine to the next ray it needs
//                 (shadow ray of the next light in the scan, or the next reflection ray) or writes the pixel
//
// A pixel has one closest-hit ray and/or one shadow ray in flight: in shade_and_reflect the reflection ray that
// leaves a hit is queued together with the hit's first shadow ray (begin_shade_lit), so a frame takes about D + 2
// big rounds (+1 per extra light a stale-specular scan needs); the other kernels / literal mode take one ray per
// round. The arithmetic
// is the same set of device functions as the monolithic kernel (rt_device.h), only the control flow is cut
// at the traversal calls; tests require both paths to agree bit for bit.
#include "rt_kernels.h"
#include "rt_grid.h"

#include <algorithm>
#include <cstdio>
#include <cstdlib>
#include <cstring>

namespace rt {

// ---- per-pixel state ------------------------------------------------------------------------------------------------
// Two 32-byte ray records per pixel (store_ray), then 16-byte BLOCKS of four fields, each block an array over the
// pixels (field f of pixel i at st[16 n + (f - 16 & ~3) n + 4 i + (f & 3)]): the fields a resume step reads or writes
// together sit in one block, so a lane moves them with ONE 16-byte access and a wave with one 1-KB request, instead of
// four 256-byte requests to four arrays (the kernel is bound by the memory system, not by its arithmetic). The default
// shade_and_reflect flow touches the first four blocks only; the rest belongs to literal mode, the `shade` kernel and
// stale-specular light scans.
enum : uint32_t {
    F_SX, F_SY, F_SZ, F_SW, F_DX, F_DY, F_DZ, F_DW,          // closest-hit ray in flight (primary / reflection): ONE 32-byte record per pixel
    F_A0, F_A1, F_A2, F_A3, F_A4, F_A5, F_A6, F_A7,          // shadow ray in flight: a second 32-byte record (store_ray, slot 1)
    F_PX, F_PY, F_PZ, F_HIDX,                                  // hit being shaded: view-space point, object
    F_NX, F_NY, F_NZ, F_PHASE,                                 // ... its normal; phase word (phase | flags | light index << 10)
    F_ABR, F_ABG, F_ABB, F_AP,                                 // shade_and_reflect: absorbColor, absorptionPercent
    F_RES_T, F_RES_I, F_RES_ANY, F_BOUNCES,                    // trace results (closest: t / index; shadow: 1 lit, 0 blocked); bounces left
    F_RX, F_RY, F_RZ, F_PW,                                    // reflection vector and point.w (only when the ray does not leave with the hit)
    F_RR, F_RG, F_RB, F_SPARE,                                 // reflectColor across a reflection trace (literal mode)
    F_AR, F_AG, F_AB, F_DR, F_DG, F_DB, F_SR, F_SG,            // light-loop terms carried between rounds
    F_SB, F_CR, F_CG, F_CB,                                    // ... and the forward (literal / shade kernel) colour
    F_COUNT
};
static_assert(F_COUNT % 4 == 0 && F_PX == 16, "blocks of four after the two ray records");
enum : uint32_t { PH_DONE = 0, PH_PRIMARY = 1, PH_SHADOW_PRIMARY = 2, PH_SHADOW_REFLECT = 3, PH_REFLECT = 4 };
// flag on the phase word: the reflection ray that leaves the hit being shaded is already in flight / traced (its ray
// in slot 0, its result in F_RES_T / F_RES_I) - see begin_shade
constexpr uint32_t PH_FLAG_REFLECTION_SENT = 0x100u;
// ... and it has not been traced yet (set for exactly the round that follows its emission; wf_finish needs to know)
constexpr uint32_t PH_FLAG_REFLECTION_PENDING = 0x200u;
// the light whose shadow ray is in flight rides in the same word
constexpr uint32_t kPhaseLightShift = 10;
static_assert(kPhaseLightShift == 10, "rt_create limits the light count to 2^22 - 1");

// Round state, in device memory (WavefrontBuffers::counts). The host enqueues a frame's rounds WITHOUT waiting for the
// queue lengths: every kernel of a round reads them from here (resolve_round) and a one-block kernel between rounds
// (wf_advance) turns the queues a round has filled into the input of the next one. Launch grids are sized for the
// most a queue can hold; surplus workgroups exit on their first comparison.
enum : uint32_t {
    RS_NEXT_CLOSEST = 0, RS_NEXT_ANY = 1,   // append counters of the queues being filled (block_push)
    RS_SLICE_A = 2, RS_SLICE_B = 3,         // survivors of a shadow slice (brute-force path), ping-pong
    RS_N_CLOSEST = 4, RS_N_ANY = 5,         // lengths of the queues the current round consumes
    RS_CUR = 6,                             // which queue pair that is
    RS_FINISH = 7,                          // 0: rounds go on, 1: the rest belongs to wf_finish, 2: nothing left
    RS_ROUNDS = 8,                          // rounds that had work (statistics)
    RS_FRAME_STUCK = 9,                     // wf_frame: waves that gave up on a state in which nothing could move (a logic error: the host fails the frame)
    RS_WORDS = 16
};

struct WfParams {
    RenderParams rp;
    float* st;            // F_COUNT x n_local floats
    uint32_t* qs[2][2];   // two queue pairs {closest-hit, any-hit} of pixel ids; a round consumes one pair and fills the other
    uint32_t* q_closest;  // -- resolved per kernel from the round state (resolve_round): the queues being filled ...
    uint32_t* q_any;
    uint32_t* q_prev_closest;  // ... and the queues being consumed, with their lengths
    uint32_t* q_prev_any;
    uint32_t* counts;     // the round state (RS_*) followed by the run-ticket counters of the grid walk
    const HotPair* shadow_pairs;  // pair stream sorted by decreasing size, for order-free shadow tests
    GridDesc grid;                // conservative uniform grid (enabled = 0: brute force)
    ScreenTiles tiles;            // screen-tile object lists for pinhole primary rays
    LightTiles ltiles;            // light tiles for the last light's shadow rays
    BlockGrid bgrid;              // the closest-hit walk's coarse grid of 32-byte blocks (enabled = 0: walk_segment / trace_segment)
    uint32_t n_prev_closest, n_prev_any;
    int kernel;
    uint32_t first_round;  // the closest-hit rays of this round are the primary rays (never stored: closest_ray())
    uint32_t identity_queue;  // ... and its queue is implicit: entry t is work-item t, every work-item is in phase PH_PRIMARY
                              // (no wf_begin, no queue to write and read back, no phase words - frames without padding work-items)
    uint32_t count_rays;  // instrumentation on
};

// Kernel prologue: the round's queues and their lengths, read from device memory (wave-uniform scalar loads; the
// values were written by wf_advance, an earlier kernel of the same stream). False: the frame is past its rounds.
__device__ __forceinline__ bool resolve_round(WfParams& w) {
    const uint32_t* rs = w.counts;
    const bool odd = (rs[RS_CUR] & 1u) != 0u;  // (selects, not a run-time index: the struct must stay in scalar registers)
    w.q_prev_closest = odd ? w.qs[1][0] : w.qs[0][0];
    w.q_prev_any = odd ? w.qs[1][1] : w.qs[0][1];
    w.q_closest = odd ? w.qs[0][0] : w.qs[1][0];
    w.q_any = odd ? w.qs[0][1] : w.qs[1][1];
    w.n_prev_closest = rs[RS_N_CLOSEST];
    w.n_prev_any = rs[RS_N_ANY];
    return rs[RS_FINISH] == 0u;
}

// Work-item -> pixel of this rank's frame part. Pinhole frames are walked in 8x8-pixel tiles (work-items 64k ..
// 64k+63 = one tile), so that the 64 rays a wave traces together, and the rays in flight on the chip, are
// neighbours in the image in both directions; pixel state and queues are indexed by work-item, only the primary
// ray and the final stores need the pixel.
__device__ __forceinline__ uint64_t pixel_of(const RenderParams& p, uint64_t t) {
    if (!p.wf_tile_order) return t;
    const uint32_t tt = (uint32_t)t;
    const uint32_t tile = tt >> 6, within = tt & 63u;
    const uint32_t trow = tile / p.bundles_x, tcol = tile - trow * p.bundles_x;
    return (uint64_t)(((trow << 3) + (within >> 3)) * p.width + (tcol << 3) + (within & 7u));
}

__device__ __forceinline__ uint64_t field_at(const WfParams& w, uint32_t f, uint64_t i) {
    return (uint64_t)(f & ~3u) * w.rp.n_local + 4u * i + (f & 3u);  // (f >= 16: block (f - 16) / 4 behind the 16 n ray words)
}
__device__ __forceinline__ float& F(const WfParams& w, uint32_t f, uint64_t i) { return w.st[field_at(w, f, i)]; }
__device__ __forceinline__ uint32_t& U(const WfParams& w, uint32_t f, uint64_t i) {
    return reinterpret_cast<uint32_t*>(w.st)[field_at(w, f, i)];
}
// whole blocks (16-byte aligned: hipMalloc'ed base, every block starts at a multiple of 4 floats)
__device__ __forceinline__ float4 load_block(const WfParams& w, uint32_t f0, uint64_t i) {
    return *reinterpret_cast<const float4*>(w.st + field_at(w, f0, i));
}
__device__ __forceinline__ void store_block(const WfParams& w, uint32_t f0, uint64_t i, float4 v) {
    *reinterpret_cast<float4*>(w.st + field_at(w, f0, i)) = v;
}

// A ray in flight is the one record the trace kernels read per lane at scattered pixel ids (a lane takes a new
// ray whenever its old one ends), so it is kept as one 32-byte record per pixel: two 16-byte loads from one cache
// line instead of eight 4-byte loads from eight lines. Two slots, in the space of the first sixteen fields: 0 for
// the closest-hit ray (primary / reflection), 1 for the shadow ray - a pixel can have one of each in flight.
constexpr uint32_t kSlotClosest = 0, kSlotShadow = 1;
__device__ __forceinline__ void store_ray(const WfParams& w, uint64_t i, const Ray& r, uint32_t slot) {
    float4* rec = reinterpret_cast<float4*>(w.st) + 2 * ((uint64_t)slot * w.rp.n_local + i);
    rec[0] = make_float4(r.sx, r.sy, r.sz, r.sw);
    rec[1] = make_float4(r.dx, r.dy, r.dz, r.dw);
}
__device__ __forceinline__ Ray load_ray(const WfParams& w, uint64_t i, uint32_t slot) {
    const float4* rec = reinterpret_cast<const float4*>(w.st) + 2 * ((uint64_t)slot * w.rp.n_local + i);
    const float4 s = rec[0], d = rec[1];
    Ray r;
    r.sx = s.x; r.sy = s.y; r.sz = s.z; r.sw = s.w;
    r.dx = d.x; r.dy = d.y; r.dz = d.z; r.dw = d.w;
    return r;
}
// The hit being shaded. Its point and normal are rewritten with the phase word (which carries the light index) by
// emit_shadow: two 16-byte stores. The reflection vector and point.w are only kept when the reflection ray has not
// been sent with the hit (begin_shade_lit).
__device__ __forceinline__ void store_hit_extra(const WfParams& w, uint64_t i, const HitRec& h) {
    store_block(w, F_RX, i, make_float4(h.rx, h.ry, h.rz, h.pw));
}
// trace results: (t, index) of a closest-hit ray as one 8-byte store / load, the shadow flag next to them
__device__ __forceinline__ void store_closest_result(const WfParams& w, uint64_t i, float T, int idx) {
    *reinterpret_cast<float2*>(w.st + field_at(w, F_RES_T, i)) = make_float2(T, __uint_as_float((uint32_t)idx));
}
__device__ __forceinline__ void load_closest_result(const WfParams& w, uint64_t i, float& T, int& idx) {
    const float2 r = *reinterpret_cast<const float2*>(w.st + field_at(w, F_RES_T, i));
    T = r.x;
    idx = (int)__float_as_uint(r.y);
}

// The shadow ray a pixel has in flight on the grid path, rebuilt from what emit_shadow left in the pixel's state: the hit
// point (F_PX block) and the light (the phase word's light index; `last_light` = the queue entry says it is the light of
// the light tiles, and the phase word is not read). Bit for bit the ray emit_shadow computed: same function, same inputs.
__device__ __forceinline__ bool shadow_rays_rebuilt(const WfParams& w) {
    return w.grid.enabled && !w.rp.scene.literal && !w.grid.has_triangles;
}
template <bool FUSED>
__device__ __forceinline__ Ray shadow_of_pixel(const WfParams& w, uint64_t i, bool last_light, uint32_t& li) {
    const float4 hp = load_block(w, F_PX, i);
    li = last_light ? w.ltiles.light : (U(w, F_PHASE, i) >> kPhaseLightShift);
    Ray ray;
    float nlx, nly, nlz;
    shadow_ray_to<FUSED>(w.rp.scene.lights[li], hp.x, hp.y, hp.z, ray, nlx, nly, nlz);
    return ray;
}

// append pixel i to a queue (wave-aggregated by the compiler: one atomic per wave per call site)
__device__ __forceinline__ void push(uint32_t* queue, uint32_t* counter, uint64_t i) {
    const uint32_t slot = atomicAdd(counter, 1u);
    queue[slot] = (uint32_t)i;
}

// Queue appends of a whole workgroup with ONE atomic per queue: a single global counter takes ~88 returning
// atomics per microsecond on this chip, so one per wave (262 144 waves per round at 4096^2) would cost 3 ms a
// round. Order inside the workgroup is preserved (neighbouring pixels stay neighbours in the next trace).
#ifndef RT_RESUME_THREADS
#define RT_RESUME_THREADS 512   // threads per workgroup of wf_begin / wf_resume / wf_step. MEASURED (round 4, cfg4 frame, two runs): 1024 (round 2's
#endif                          // choice, when one atomic per workgroup mattered most): 11.87 / 11.77 ms, 512: 11.65 / 11.56, 256: 12.05 / 11.96 - a
                                // 1 024-thread workgroup is ALL sixteen wave slots of a CU at 4 waves per SIMD: the CU idles until its slowest wave is done

constexpr int kResumeThreads = RT_RESUME_THREADS;
constexpr uint32_t kQueueAlsoShadow = 0x80000000u;  // flag on a closest-queue entry (pixel ids are < 2^31: launch_wavefront checks)
constexpr uint32_t kQueuePixel = 0x7fffffffu;
constexpr uint32_t kQueueLastLight = 0x80000000u;  // flag on a shadow-queue entry: the ray goes to the light of the light tiles
__device__ __forceinline__ void block_push(bool want_closest, bool want_any, uint32_t id, uint32_t* __restrict__ q_closest,
                                           uint32_t* __restrict__ q_any, uint32_t* __restrict__ counts, uint32_t any_flag = 0u) {
    __shared__ uint32_t s_cnt[2][kResumeThreads / 64];
    __shared__ uint32_t s_base[2];
    const uint32_t wave = threadIdx.x >> 6, lane = threadIdx.x & 63u;
    const uint32_t n_waves = (blockDim.x + 63u) >> 6;
    const unsigned long long bc = __ballot(want_closest), ba = __ballot(want_any);
    if (lane == 0u) { s_cnt[0][wave] = (uint32_t)__popcll(bc); s_cnt[1][wave] = (uint32_t)__popcll(ba); }
    __syncthreads();
    if (threadIdx.x < 2u) {
        uint32_t total = 0;
        for (uint32_t k = 0; k < n_waves; ++k) { const uint32_t c = s_cnt[threadIdx.x][k]; s_cnt[threadIdx.x][k] = total; total += c; }
        s_base[threadIdx.x] = total ? atomicAdd(&counts[threadIdx.x], total) : 0u;
    }
    __syncthreads();
    const unsigned long long below = (1ull << lane) - 1ull;
    // a pixel that queues a shadow ray AND a reflection ray sits in both queues and is resumed from its shadow entry: its
    // closest-queue entry says so in its top bit, so that wf_resume can drop it without touching the pixel's state
    if (want_closest) q_closest[s_base[0] + s_cnt[0][wave] + (uint32_t)__popcll(bc & below)] = want_any ? (id | kQueueAlsoShadow) : id;
    if (want_any) q_any[s_base[1] + s_cnt[1][wave] + (uint32_t)__popcll(ba & below)] = id | any_flag;
}

__device__ __forceinline__ unsigned long long wave_sum64(unsigned long long v) {
#pragma unroll
    for (int off = 32; off > 0; off >>= 1) v += __shfl_xor(v, off, 64);
    return v;
}

__device__ __forceinline__ bool begin_pixel(const WfParams& w, uint64_t i);

// ---- wf_begin: primary rays --------------------------------------------------------------------------------------
__global__ __launch_bounds__(kResumeThreads) void wf_begin(const WfParams wk) {
    WfParams w = wk;
    (void)resolve_round(w);
    const RenderParams& p = w.rp;
    const uint64_t i = (uint64_t)blockIdx.x * kResumeThreads + threadIdx.x;
    bool want = false;
    if (i < p.n_local) want = begin_pixel(w, i);
    block_push(want, false, (uint32_t)i, w.q_closest, w.q_any, w.counts);
}

// global ray index of work-item i (shards: interleaved tiles), and the primary ray itself - regenerated (pinhole) or
// re-read from the upload buffer wherever it is needed; it is never copied into the pixel state
__device__ __forceinline__ uint64_t global_ray_of(const RenderParams& p, uint64_t i) {
    const uint64_t px = pixel_of(p, i);
    if (p.world <= 1u) return px;
    const uint64_t run = px / p.run_rays;  // (a shard: run_rays == tile_rays; a pass of rt_render may stand for several consecutive ranks)
    return (run * p.world + p.rank) * p.tile_rays + (px - run * p.run_rays);
}
__device__ __forceinline__ Ray primary_ray(const RenderParams& p, uint64_t g) {
    Ray ray;
    if (p.pinhole) {
        const uint32_t gi = (uint32_t)g;
        const uint32_t row = gi / p.width;
        const uint32_t col = gi - row * p.width;
        ray.sx = 0.f; ray.sy = 0.f; ray.sz = 0.f; ray.sw = 1.f;
        ray.dx = (float)col - p.half_w;
        ray.dy = (p.height_f - (float)row) - p.half_h;
        ray.dz = p.z;
        ray.dw = 0.f;
    } else {
        const float4 s = p.rays[2 * g];
        const float4 d = p.rays[2 * g + 1];
        ray.sx = s.x; ray.sy = s.y; ray.sz = s.z; ray.sw = s.w;
        ray.dx = d.x; ray.dy = d.y; ray.dz = d.z; ray.dw = d.w;
    }
    return ray;
}
// the closest-hit ray pixel i has (or had) in flight
__device__ __forceinline__ Ray closest_ray(const WfParams& w, uint64_t i, bool primary) {
    return primary ? primary_ray(w.rp, global_ray_of(w.rp, i)) : load_ray(w, i, kSlotClosest);
}

__device__ __forceinline__ bool begin_pixel(const WfParams& w, uint64_t i) {
    const RenderParams& p = w.rp;
    if (global_ray_of(p, i) >= p.n_rays) {  // padding work-item of a ragged last tile
        const uint64_t px = pixel_of(p, i);
        if (w.kernel == 0) reinterpret_cast<float*>(p.out)[px] = kMaxFloat;
        else reinterpret_cast<float4*>(p.out)[px] = make_float4(0.f, 0.f, 0.f, 1.0f);
        if (p.aux_t) p.aux_t[px] = kMaxFloat;
        if (p.aux_index) p.aux_index[px] = -1;
        store_block(w, F_NX, i, make_float4(0.f, 0.f, 0.f, __uint_as_float(PH_DONE)));
        return false;
    }
    store_block(w, F_NX, i, make_float4(0.f, 0.f, 0.f, __uint_as_float(PH_PRIMARY)));  // (a whole block: full lines instead of strided words)
    return true;
}

// ---- lean traversal kernels ----------------------------------------------------------------------------------------
template <bool FUSED, bool DW0>
__global__ __launch_bounds__(256) void wf_trace_closest(const WfParams wk) {
    WfParams w = wk;
    if (!resolve_round(w)) return;
    const uint32_t t = blockIdx.x * 256u + threadIdx.x;
    if (t >= w.n_prev_closest) return;
    const uint64_t i = w.identity_queue ? t : (w.q_prev_closest[t] & kQueuePixel);
    const Ray ray = closest_ray(w, i, w.first_round != 0u);
    float T = kMaxFloat;
    int idx = -1;
    closest_hit<FUSED, DW0>(w.rp.scene.pairs, w.rp.scene.n_pairs, ray, T, idx);
    store_closest_result(w, i, T, idx);
    const unsigned long long lanes = (unsigned long long)__popcll(__ballot(true));
    if (w.count_rays && (threadIdx.x & 63u) == 0u)  // 2 tests per pair for every ray of the wave
        atomicAdd(&w.rp.counters->tests, 2ull * w.rp.scene.n_pairs * lanes);
}

// one object of a wave-uniform list (scalar loads) against this lane's ray
template <bool FUSED>
__device__ __forceinline__ void tile_candidate(const RT_CONST HotObjectC* o, int k, const Ray& ray, float& T, int& idx, bool& cur_sphere) {
    const uint32_t type = o->type;
    if (type == 2u) {  // triangle (extension)
        const f4 r0 = o->row0, r1 = o->row1, r2 = o->row2;
        float t;
        if (triangle_candidate(r0.x, r0.y, r0.z, r1.x, r1.y, r1.z, r2.x, r2.y, r2.z, r0.w, r1.w, r2.w, __uint_as_float(o->pad[0]), ray, t))
            closest_take(t, k, false, T, idx, cur_sphere);
        return;
    }
    float sx, sy, sz, dx, dy, dz;
    object_space_one<FUSED, true>(o, ray, sx, sy, sz, dx, dy, dz);
    closest_update_unordered<FUSED>(type, sx, sy, sz, dx, dy, dz, k, T, idx, cur_sphere);
}

// First round of a pinhole frame: a wave holds 64 consecutive pixels of one row, i.e. one 64x8 screen tile, and
// walks that tile's object list with wave-uniform scalar loads (plus the always-list). Waves that straddle tiles
// (ragged ends) fall back to the per-lane grid walk. Order-free tie rules, so the result is the same either way.
template <bool FUSED>
__global__ __launch_bounds__(256) void wf_trace_primary_tiles(const WfParams wk) {
    WfParams w = wk;
    if (!resolve_round(w)) return;
    const uint32_t t = blockIdx.x * 256u + threadIdx.x;
    if (t >= w.n_prev_closest) return;
    const RenderParams& p = w.rp;
    const uint64_t i = w.identity_queue ? t : (w.q_prev_closest[t] & kQueuePixel);
    const uint64_t g = global_ray_of(p, i);
    const Ray ray = primary_ray(p, g);
    const uint32_t row = (uint32_t)g / p.width, col = (uint32_t)g - row * p.width;
    const uint32_t tile = (row >> 3) * w.tiles.tiles_x + (col >> w.tiles.col_shift);
    const uint32_t first = __builtin_amdgcn_readfirstlane(tile);
    float T = kMaxFloat;
    int idx = -1;
    uint32_t tested = 0;
    if (__ballot(tile != first) == 0ull) {
        bool cur_sphere = false;
        const RT_CONST HotObjectC* hot = (const RT_CONST HotObjectC*)(p.scene.hot);
        for (uint32_t a = 0; a < w.grid.n_always; ++a) {
            const int k = (int)w.grid.always[a];
            tile_candidate<FUSED>(hot + k, k, ray, T, idx, cur_sphere);
        }
        for (uint32_t a = 0; a < w.tiles.n_global; ++a) {  // objects that project onto the whole screen
            const int k = (int)w.tiles.entries[w.tiles.global_begin + a];
            tile_candidate<FUSED>(hot + k, k, ray, T, idx, cur_sphere);
        }
        const uint32_t e0 = w.tiles.tile_start[first], e1 = w.tiles.tile_start[first + 1];
        tested = w.grid.n_always + w.tiles.n_global + (e1 - e0);
        for (uint32_t e = e0; e < e1; ++e) {
            const int k = (int)w.tiles.entries[e];
            tile_candidate<FUSED>(hot + k, k, ray, T, idx, cur_sphere);
        }
    } else {
        closest_hit_grid<FUSED, true>(w.grid, p.scene.hot, ray, T, idx, tested);
    }
    store_closest_result(w, i, T, idx);
    if (w.count_rays) atomicAdd(&w.rp.counters->tests, (unsigned long long)tested);
}

// ---- grid traversal, persistent waves ----------------------------------------------------------------------------
// The straightforward per-lane walk (one thread = one ray, nested cell / candidate loops) keeps only ~9 of 64
// lanes busy (PMC: SQ_THREAD_CYCLES_VALU / SQ_ACTIVE_INST_VALU): rays differ in length, cells in population, and a
// finished lane idles until the longest ray of its wave is done. Here a wave owns a segment of kSegment
// consecutive queue entries; a lane that finishes its ray immediately takes the next one of the segment (a
// wave-uniform cursor, no atomics), and the walk is a single loop in which every lane does at most one cell fetch
// and one candidate per trip. Same cells, same candidates, same order-free update -> same results.
// (Tried and dropped: scheduling one phase per trip by the number of lanes waiting in each state - refill / step
// / pre-test / full test - so that the full test only runs with half the wave ready: 2x SLOWER, the extra trips
// and ballots cost more than the better packing of the expensive phase saves.)
#ifndef RT_SEGMENT
#define RT_SEGMENT 256  // queue entries per run (round 2, with 8 ticket regions: 128: 20.4 ms per cfg4 frame, 256: 20.3, 512: 20.6+; one region: 128: 22.1, 256: 20.7, 1024: 22.1, 2048: 25.0)
#endif
constexpr uint32_t kSegment = RT_SEGMENT;
#ifndef RT_TICKET_REGIONS
#define RT_TICKET_REGIONS 8  // ticket counters per launch, one region of the queue per XCD: a single counter's returning atomics (~88 per microsecond) were costing 1.6 ms per cfg4 frame at 128-entry runs once the shadow rays had left the grid walk
#endif
constexpr uint32_t kTicketRegions = RT_TICKET_REGIONS;
constexpr uint32_t kTicketStride = 16;   // one counter per 64-byte line
constexpr uint32_t kTicketWords = 16u * kTicketStride;  // room for up to 16 regions per launch
constexpr uint32_t kTicketBase = 16;     // counts[16 ..]: closest-hit launch, counts[16 + kTicketWords ..]: shadow launch

#ifndef RT_WALK_DEDUPE
#define RT_WALK_DEDUPE 1   // an object met again in the next cell is not parked / tested twice
#endif
#ifndef RT_REFILL_MIN
#define RT_REFILL_MIN 16  // new rays are handed out once this many lanes are idle (setting a ray up is ~150 instructions)
#endif
#ifndef RT_SKIP_CAP
#define RT_SKIP_CAP 2  // empty-space steps a lane takes per trip beyond the first (the other lanes wait for it; cfg5: 0: 94.7 ms, 2: 91.2, 6: 91.6, 12: 92.1)
#endif
#ifndef RT_INLINE_LAST_SHADOW
#define RT_INLINE_LAST_SHADOW 1  // a path's last hit tests its last-light shadow ray inside wf_resume (shade_last_light_inline)
#endif
#ifndef RT_FLAT_PARK
#define RT_FLAT_PARK 1
#endif
#ifndef RT_CELL_INLINE
#define RT_CELL_INLINE 1  // the persistent walk reads 32-byte cell records that hold the first entry (GridDesc::cell_rec) - range and first
                          // candidate in ONE request instead of two dependent ones: cfg4 17.5 -> 17.0 ms. Not for sparsely occupied
                          // grids (meshes): four times the bytes per EMPTY cell cost cfg5 60.4 -> 62.6 ms
#endif
#ifndef RT_SKIP_FEW_LANES
#define RT_SKIP_FEW_LANES 12  // ... unless at most this many lanes of the wave are still walking: then as many as the cell allows
#endif
#ifndef RT_WALK_FIRST_CELL
#define RT_WALK_FIRST_CELL 0  // 1: mesh scenes test an entry only in the first cell of the walk that holds it (rt_grid.h: entered_inside). cfg5: exact tests per closest-hit ray 19.7 -> 13.1, frame 64.1 -> 66.7 ms - the rule costs every trip more than the repeated tests did
#endif
#ifndef RT_DEFER_PENDING
#define RT_DEFER_PENDING 8  // run the exact tests once this many lanes hold a candidate ... (8: 20.4, 16: 20.7, 28: 22.9 ms - fewer candidates wait since repeated objects are skipped)
#endif
#ifndef RT_DEFER_STUCK_SHIFT
#define RT_DEFER_STUCK_SHIFT 2  // ... or once a quarter of the live lanes cannot move without theirs
#endif

// (the pre-test of rt_grid.h with the squared radius taken from the record; w2 = -inf: never passes)
__device__ __forceinline__ bool misses_bounding_sphere2(const float4 s, float sx, float sy, float sz, float dx, float dy, float dz,
                                                        float dd, float alpha) {
    const float ox = s.x - sx, oy = s.y - sy, oz = s.z - sz;
    const float oo = __builtin_fmaf(oz, oz, __builtin_fmaf(oy, oy, ox * ox));
    const float od = __builtin_fmaf(oz, dz, __builtin_fmaf(oy, dy, ox * dx));
    const float c = oo - s.w;
    const float disc = __builtin_fmaf(od, od, -(dd * __builtin_fmaf(-alpha, oo, c)));
    return disc < 0.f || (od < 0.f && c > 1.0e-5f * oo);
}

// one candidate of a light-tile block (LightTiles::blocks): decoded sphere and distance key
__device__ __forceinline__ void lt_block_entry(const LightTiles& lt, uint32_t lo, uint32_t hi, float4& sphere, float& key) {
    const float r = (float)((hi >> 16) & 0xffu) * lt.rstep;
    sphere = make_float4(__builtin_fmaf((float)(lo & 0xffffu), lt.lat_step, lt.lat_lox), __builtin_fmaf((float)(lo >> 16), lt.lat_step, lt.lat_loy),
                         __builtin_fmaf((float)(hi & 0xffffu), lt.lat_step, lt.lat_loz), r * r);
    key = (float)(hi >> 24) * lt.kstep;
}

// The queue of a persistent walk, cut in runs of `seg` entries (all three walks: trace_segment, walk_segment, block_segment).
// Few runs: wave i traces run i. Many: the resident waves draw run after run from ticket counters, in queue order - so the
// chip stays balanced to the end of the launch without the per-wave tails of long fixed segments, and the rays in flight at any
// moment are neighbours in the queue (= in the image), which is what keeps cells and objects in L2. Run length: kSegment while
// the queue holds at least two such runs per launched wave; shorter (down to 64) for small queues - the tail of a frame, or one
// rank's share of it on a multi-GPU node - so that every wave still gets work and no wave ends the launch alone with a long run.
// Dynamic hand-out is XCD-affine: the queue is cut into kTicketRegions contiguous regions with one ticket counter each; a wave
// draws from the region of the XCD it runs on (HW_REG_XCC_ID - placement is only a speed matter) and moves on to the next
// region when its own is used up. Each XCD has its own L2: this way the rays in flight on one XCD are neighbours in the queue.
// (A single counter's returning atomics, ~88 per microsecond, were costing 1.6 ms per cfg4 frame at 128-entry runs.)
struct RunCursor {
    uint32_t seg, n_runs, next, seg_end, region, regions_tried;
    bool dynamic;
    __device__ __forceinline__ bool grab(uint32_t n_queue, uint32_t* __restrict__ run_ctr, uint32_t lane) {
        for (;;) {
            const uint32_t lo = (uint32_t)(((uint64_t)n_runs * region) / kTicketRegions);
            const uint32_t hi = (uint32_t)(((uint64_t)n_runs * (region + 1u)) / kTicketRegions);
            uint32_t r = 0;
            if (lane == 0u) r = atomicAdd(run_ctr + region * kTicketStride, 1u);
            r = __builtin_amdgcn_readfirstlane(r) + lo;
            if (r < hi) {
                next = r * seg;
                seg_end = (n_queue - next < seg) ? n_queue : next + seg;
                return true;
            }
            if (++regions_tried >= kTicketRegions) return false;
            region = (region + 1u == kTicketRegions) ? 0u : region + 1u;
        }
    }
    // false: nothing for this wave to do
    __device__ __forceinline__ bool begin(uint32_t n_queue, uint32_t wave, uint32_t n_waves, uint32_t* __restrict__ run_ctr, uint32_t lane) {
        seg = kSegment;
        while (seg > 64u && n_queue < 2u * n_waves * seg) seg >>= 1;
        n_runs = (n_queue + seg - 1u) / seg;
        dynamic = n_runs > n_waves;
        next = 0; seg_end = 0; region = 0; regions_tried = 0;
        if (kTicketRegions > 1u) {
            uint32_t xcc;
            asm volatile("s_getreg_b32 %0, hwreg(HW_REG_XCC_ID)" : "=s"(xcc));
            region = (xcc & 0xfu) % kTicketRegions;
        }
        if (dynamic) return grab(n_queue, run_ctr, lane);
        if (wave >= n_runs) return false;
        next = wave * seg;
        seg_end = (n_queue - next < seg) ? n_queue : next + seg;
        return true;
    }
};

// One segment of a queue, traced by one wave. A candidate that passes the 16-byte pre-test is not tested on the
// spot: the lane parks it (`pend`) and keeps walking; the reference's exact test - a 52-byte gather and ~100
// instructions - runs for the whole wave when enough lanes hold one (or are stuck behind theirs), so it executes
// with tens of lanes instead of the 3-5 that happen to need it in any single trip. The closest-hit update is
// order-free and T only ever shrinks, so a late update can only make a lane look at MORE cells than necessary.
template <bool FUSED, bool ANY, bool STATS, bool TRI>
__device__ __forceinline__ void trace_segment(const WfParams& w, const uint32_t* __restrict__ queue, uint32_t n_queue,
                                              uint32_t wave, uint32_t n_waves, uint32_t* __restrict__ run_ctr,
                                              unsigned long long& tested) {
    const uint32_t lane = threadIdx.x & 63u;
    RunCursor rc;   // runs of queue entries and their tickets (wave-uniform): see RunCursor
    if (!rc.begin(n_queue, wave, n_waves, run_ctr, lane)) return;
    uint32_t& next = rc.next;
    const uint32_t& seg_end = rc.seg_end;
    auto grab = [&]() -> bool { return rc.grab(n_queue, run_ctr, lane); };
    bool more = rc.dynamic;  // may another run be drawn?
    const GridDesc& g = w.grid;
    const HotObject* __restrict__ hot = w.rp.scene.hot;

    unsigned long long s_rays = 0, s_trips = 0, s_live = 0, s_fetch = 0, s_pre = 0, s_flush = 0, s_refill = 0;  // STATS only
    int st = 0;            // 0 idle (needs a ray), 1 at a cell whose list has not been fetched, 2 walking a list, 3 walk over
    uint32_t pix = 0;      // pixel whose ray this lane is tracing
    Ray ray = {};
    LeanWalk wk = {};
    uint32_t e = 0, e1 = 0;      // the current cell's list: next entry, end
    float T = kMaxFloat, slack = 0.f, dd = 0.f;
    int idx = -1;
    bool cur_sphere = false;
    bool in_lt = false;    // shadow rays of the last light: the lane's list is a light tile (no cell walk: the list is all there is)
    bool lt_blocks = false;  // ... in block form (LightTiles::blocks): the lane is in state 4
    bool pend = false;     // a candidate that passed the pre-test and awaits the exact test
    uint32_t pend_k = 0;   // ... its object, and the object of the last exact test: an object is registered in every cell its
    uint32_t done_k = 0xffffffffu;  // sphere reaches, so a ray meets it again in the next cell(s) - once is enough

    for (;;) {
        // ---- hand out rays to idle lanes ----
        const unsigned long long idle = __ballot(st == 0);
        if (next < seg_end && ((uint32_t)__popcll(idle) >= (uint32_t)RT_REFILL_MIN || idle == ~0ull)) {
            // rank of this lane among the idle ones (mbcnt: set bits of the mask below this lane)
            const uint32_t mine = next + __builtin_amdgcn_mbcnt_hi((uint32_t)(idle >> 32), __builtin_amdgcn_mbcnt_lo((uint32_t)idle, 0u));
            if (STATS && lane == 0u) ++s_refill;
            if (st == 0 && mine < seg_end) {
                if (STATS) ++s_rays;
                const uint32_t entry = (!ANY && w.identity_queue) ? mine : queue[mine];
                pix = (!ANY && w.identity_queue) ? mine : (entry & kQueuePixel);
                uint32_t ray_light;  // shadow rays: the light they go to; reflection rays: begin_shade_lit's note (the object they leave)
                if (ANY && !TRI) {
                    ray = shadow_of_pixel<FUSED>(w, pix, (entry & kQueueLastLight) != 0u, ray_light);
                } else if (ANY) {
                    ray = load_ray(w, pix, kSlotShadow);
                    ray_light = __float_as_uint(ray.dw);
                } else {
                    ray = closest_ray(w, pix, w.first_round != 0u);
                    ray_light = __float_as_uint(ray.dw);
                }
                ray.sw = 1.0f; ray.dw = 0.0f;  // what every ray of a grid-able frame carries (rt_create checks the preconditions)
                in_lt = false;
                lt_blocks = false;
                // reflection rays: the object the ray leaves has had its exact test already (begin_shade_lit)
                if (!ANY) done_k = (w.first_round == 0u) ? ray_light : 0xffffffffu;
                T = kMaxFloat; idx = -1; cur_sphere = false; pend = false; if (ANY) done_k = 0xffffffffu;
                bool done = false;
                for (uint32_t a = 0; a < g.n_always && !done; ++a) {  // objects every ray must test (usually none)
                    const int k = (int)g.always[a];
                    float t;
                    bool sphere;
                    const bool cand = lane_candidate<FUSED, true, TRI>(hot + k, ray, t, sphere);
                    if (STATS) ++tested;
                    if (ANY) done = cand && !(t >= 1.f);
                    else if (cand) closest_take(t, k, sphere, T, idx, cur_sphere);
                }
                dd = ray.dx * ray.dx + ray.dy * ray.dy + ray.dz * ray.dz;
                slack = dd > 0.f ? kWalkSlackCells * g.cell * __builtin_amdgcn_rsqf(dd) * 1.0001f : 3.0e38f;
                Walk w0 = {};
                if (ANY && w.ltiles.enabled && ray_light == w.ltiles.light) {
                    // the last light's shadow ray: one tile of the light's own perspective holds every object it can meet
                    uint32_t tile;
                    float cut;
                    const float chk = ((ray.sx + ray.sy) + ray.sz) + ((ray.dx + ray.dy) + ray.dz);
                    if (!done && chk == chk && light_tile_of(w.ltiles, ray.sx, ray.sy, ray.sz, tile, cut)) {
                        slack = cut;  // (the slot is free in this mode: no cell walk) how far from the light an occluder can start
                        if (w.ltiles.blocks_enabled) {  // the tile's chain of three-candidate blocks, head = block `tile` (state 4)
                            e = tile;
                            w0.alive = true;
                            in_lt = true;
                            lt_blocks = true;
                        } else {
                            const uint2 range = table_at(w.ltiles.tile_range, tile);
                            e = range.x;
                            e1 = range.x + range.y;
                            if (range.y != 0u) { w0.alive = true; in_lt = true; }
                        }
                    }
                } else if (!done) {
                    w0 = walk_begin(g, ray, ANY ? 1.0f + slack : 3.0e38f);
                }
                wk = lean_walk(g, w0);
                if (done || !w0.alive) {  // occluded by an always-object, or the ray misses the grid box / has no light tile
                    if (ANY) U(w, F_RES_ANY, pix) = (done || nan_shadow_blocked(w.rp.scene, ray)) ? 0u : 1u;
                    else store_closest_result(w, pix, T, idx);
                } else {
                    st = lt_blocks ? 4 : (in_lt ? 2 : 1);
                }
            }
            next += (uint32_t)__popcll(idle);
            if (next >= seg_end && more) more = grab();  // on to another run, if any is left
        }
        const unsigned long long live = __ballot(st != 0);
        if (live == 0ull) {
            if (next >= seg_end) break;
            continue;
        }
        bool advance = false, blocked = false;
        if (ANY && st == 4) {
            // ---- a light tile's block (state 4): three candidates per 32-byte fetch, as walk_segment's light-tile lanes ----
            const LightTiles& lt = w.ltiles;
            const uint32_t b = e & kLtBlockIndexMask, pos = e >> kLtBlockIndexBits;
            const uint4 q0 = table_at(lt.blocks, 2u * b), q1 = table_at(lt.blocks, 2u * b + 1u);
            // (the block's ids are requested WITH the block, not after its pre-tests: 0.8 of a shadow ray's 1.1 blocks have a
            // candidate that passes, and the walk is a chain of dependent fetches - 16 bytes more per block, one step less per ray)
            const uint4 kk = table_at(reinterpret_cast<const uint4*>(lt.block_ids), b);
            const uint32_t nxt = q0.x;
            const uint32_t lo[3] = {q0.z, q1.x, q1.z}, hi[3] = {q0.w, q1.y, q1.w};
            uint32_t pm = 0u;
            bool beyond = false;  // sorted by distance from the light: an entry beyond the ray's origin ends the list
#pragma unroll
            for (uint32_t j = 0; j < 3u; ++j) {
                float4 sphere;
                float key;
                lt_block_entry(lt, lo[j], hi[j], sphere, key);
                beyond = beyond || key > slack;
                if (STATS) ++s_pre;
                const bool pass = !beyond && !misses_bounding_sphere2(sphere, ray.sx, ray.sy, ray.sz, ray.dx, ray.dy, ray.dz, dd, g.pretest_alpha);
                pm |= pass ? (1u << j) : 0u;
            }
            pm &= 0x7u << pos;  // (a block the lane comes back to: the entries before `pos` have been dealt with)
            uint32_t stalled = 0u, back = 0u, parked = pend ? 1u : 0u;
            while (pm != 0u) {  // the ids of the candidates that passed
                const uint32_t j = (uint32_t)__builtin_ctz(pm);
                const uint32_t k = j == 0u ? kk.x : (j == 1u ? kk.y : kk.z);
                const bool dup = (k == done_k) || (parked != 0u && k == pend_k);
                const bool wait = !dup && parked != 0u;   // one parking slot: wait for the exact tests, resume at this entry
                const bool take = !dup && parked == 0u;
                pend_k = take ? k : pend_k;
                parked = take ? 1u : parked;
                stalled = wait ? 1u : stalled;
                back = wait ? j : back;
                pm = wait ? 0u : (pm & (pm - 1u));
            }
            pend = parked != 0u;
            blocked = stalled != 0u;
            st = (!blocked && (beyond || nxt == 0u)) ? 3 : 4;
            e = blocked ? (b | (back << kLtBlockIndexBits)) : nxt;
        }
        uint32_t skip = 0u;
        // how many of those a lane takes in one trip: few while the wave is full (the other lanes wait for it), all of them
        // once it is nearly empty - the end of a launch is a handful of rays on their way out of the scene, ~100 dependent
        // cell fetches each at the old cap, and no launch can be shorter than its longest ray
        const uint32_t skip_cap = (uint32_t)__popcll(live) <= (uint32_t)RT_SKIP_FEW_LANES ? 255u : (uint32_t)RT_SKIP_CAP;
        if (STATS) { if (lane == 0u) { ++s_trips; s_live += (unsigned long long)__popcll(live); } if (st == 1) ++s_fetch; }
        // ---- phase A: fetch the current cell's list ----
        bool fresh = false;                // this trip fetched the lane's cell: its first candidate came with the record
        float4 first = make_float4(0.f, 0.f, 0.f, 0.f);
        uint32_t first_k = 0u;
        if (st == 1) {
            uint2 range;
            if (RT_CELL_INLINE && !TRI) {  // (compile-time: with both forms in one kernel - a run-time switch - the frame took 18.6 ms)
                const float4 r1 = table_at(g.cell_rec, 2u * (uint32_t)wk.c + 1u);
                first = table_at(g.cell_rec, 2u * (uint32_t)wk.c);
                range = make_uint2(__float_as_uint(r1.x), __float_as_uint(r1.y));
                first_k = __float_as_uint(r1.z);
                fresh = range.y != 0u;
            } else {
                range = table_at(g.cell_range, (uint32_t)wk.c);
            }
            e = range.x;
            e1 = range.x + range.y;
            {   // (branch-free, like the parking logic below) an empty cell's offset word says how many further steps stay in empty cells
                const bool occupied = range.y != 0u;
                st = occupied ? 2 : 1;
                advance = !occupied;
                skip = occupied ? 0u : (range.x < skip_cap ? range.x : skip_cap);
            }
        }
        // ---- phase B: pre-test one candidate (also for a lane that has just fetched a non-empty cell) ----
        if (st == 2) {
            float4 bound, aux = make_float4(0.f, 0.f, 0.f, 0.f);
            if (ANY && in_lt) { bound = table_at(w.ltiles.records, 2u * e); aux = table_at(w.ltiles.records, 2u * e + 1u); }
            else if (RT_CELL_INLINE && fresh) bound = first;
            else bound = table_at(g.entry_sphere, e);
            if (STATS) ++s_pre;
            bool pass = !misses_bounding_sphere(bound, ray, dd, g.pretest_alpha);
            if (ANY) {  // light tiles: sorted by distance from the light - this entry and all after it lie beyond the ray's origin
                const bool beyond = in_lt && aux.x > slack;
                pass = pass && !beyond;
                e = beyond ? e1 - 1u : e;
            }
            uint32_t k = 0u;
            if (pass) {  // the same object again (parked, or tested a cell ago)? its result is known or on its way
                k = (ANY && in_lt) ? __float_as_uint(aux.y) : ((RT_CELL_INLINE && fresh) ? first_k : table_at(g.entries, e));
#if RT_WALK_DEDUPE
                if (k == done_k || (pend && k == pend_k)) pass = false;
#endif
                if (RT_WALK_FIRST_CELL && TRI && !(ANY && in_lt) && entered_inside(bound, ray, wk.t_enter, g.own_shrink)) pass = false;
            }
#if RT_FLAT_PARK
            {   // (branch-free: every nested divergent `if` costs this loop half a dozen scalar instructions of exec-mask bookkeeping)
                const bool stall = pass && pend;   // one parking slot: wait for the exact tests
                const bool park = pass && !pend;
                blocked = stall;
                pend_k = park ? k : pend_k;
                pend = pend || park;
                e += stall ? 0u : 1u;
                advance = advance || (!stall && e == e1);
            }
#else
            if (pass && pend) {
                blocked = true;  // one parking slot: wait for the exact tests
            } else {
                if (pass) { pend = true; pend_k = k; }
                ++e;
                if (e == e1) advance = true;
            }
#endif
        }
        // ---- step to the next cell, or end the walk ----
        if (advance) {
            const float limit = ANY ? 1.0f + slack : T + slack;
            bool stop = (ANY && in_lt) || !lean_next(g, wk) || wk.t_enter > limit;
            while (skip != 0u && !stop) {  // empty space: step on without fetching (same cells, same order, same checks)
                --skip;
                stop = !lean_next(g, wk) || wk.t_enter > limit;
            }
            skip = 0u;
            st = stop ? 3 : 1;
        }
        // ---- the exact tests, when enough lanes wait for them ----
        const unsigned long long pending = __ballot(pend);
        if (pending != 0ull) {
            const unsigned long long stuck = __ballot(pend && (blocked || st == 3));
            const uint32_t n_live = (uint32_t)__popcll(live);
            if ((uint32_t)__popcll(pending) >= (uint32_t)RT_DEFER_PENDING ||
                ((uint32_t)__popcll(stuck) << RT_DEFER_STUCK_SHIFT) >= n_live) {
                if (STATS && lane == 0u) ++s_flush;
                if (pend) {
                    float t;
                    bool sphere;
                    const bool cand = lane_candidate<FUSED, true, TRI>(hot + pend_k, ray, t, sphere);
                    if (STATS) ++tested;
                    pend = false;
                    done_k = pend_k;
                    if (ANY) {
                        if (cand && !(t >= 1.f)) { U(w, F_RES_ANY, pix) = 0u; st = 0; }
                    } else if (cand) {
                        closest_take(t, (int)pend_k, sphere, T, idx, cur_sphere);
                    }
                }
            }
        }
        // ---- a finished walk with nothing parked: the ray is done ----
        if (st == 3 && !pend) {
            if (ANY) U(w, F_RES_ANY, pix) = 1u;  // nothing in the way
            else store_closest_result(w, pix, T, idx);
            st = 0;
        }
    }
    if (STATS) {
        unsigned long long* acc = w.rp.counters->walk[ANY ? 1 : 0];
        const unsigned long long v[8] = {wave_sum64(s_rays), s_trips, s_live, wave_sum64(s_fetch), wave_sum64(s_pre), wave_sum64(tested), s_flush, s_refill};
        if (lane == 0u)
            for (int j = 0; j < 8; ++j)
                if (v[j]) atomicAdd(&acc[j], v[j]);
    }
}

#ifndef RT_WAVES_PER_EU
#define RT_WAVES_PER_EU 6      // <= 80 VGPRs, no spills (round 2: 19.75-19.8 ms per cfg4 frame vs 20.0-20.07 at 7, where the walk spills ~8 VGPRs + 30-40 SGPRs)
#endif
#ifndef RT_WAVES_PER_EU_ANY
#define RT_WAVES_PER_EU_ANY RT_WAVES_PER_EU  // the shadow-ray variant (it also carries the light-tile path)
#endif
#ifndef RT_WAVES_PER_EU_TRI
#define RT_WAVES_PER_EU_TRI 6  // the variants that know triangles would like ~90 VGPRs; 5 waves without spills measured slower (97.6 vs 93 ms, cfg5)
#endif
template <bool FUSED, bool ANY, bool STATS, bool TRI>
__global__ __launch_bounds__(256) __attribute__((amdgpu_waves_per_eu(TRI ? RT_WAVES_PER_EU_TRI : (ANY ? RT_WAVES_PER_EU_ANY : RT_WAVES_PER_EU)))) void wf_trace_grid_persistent(const WfParams wk,
                                                                 uint32_t* __restrict__ run_ctr) {
    WfParams w = wk;
    if (!resolve_round(w)) return;
    const uint32_t* queue = ANY ? w.q_prev_any : w.q_prev_closest;
    const uint32_t n_queue = ANY ? w.n_prev_any : w.n_prev_closest;
    if (n_queue == 0u) return;
    const uint32_t wave = (blockIdx.x * 256u + threadIdx.x) >> 6;
    const uint32_t n_waves = (gridDim.x * 256u) >> 6;
    unsigned long long tested = 0;
    trace_segment<FUSED, ANY, STATS, TRI>(w, queue, n_queue, wave, n_waves, run_ctr, tested);
    if (STATS && tested) atomicAdd(&w.rp.counters->tests, tested);
}

// ---- grid traversal, unified form (scenes without triangles) -----------------------------------------------------------
// What the profile of trace_segment said (round 2: ~170 vector + ~130 scalar instructions per wave-trip with 32 of 64 lanes
// active): the trip count is near its floor, the cost of a trip is not - a lane is EITHER fetching a cell OR pre-testing an
// entry OR stepping, every state is a divergent branch, and half the lanes sit out each of them. Here every walking lane
// does the same thing in every trip: fetch the 32-byte record under its cursor (GridDesc::walk_rec: cell heads, the further
// entries of a cell and the light tiles' entries share one format), pre-test it, move the cursor along `next` - and, when
// the list ends, to the next cell of a DDA whose step is computed by every lane in every trip and taken by select. No
// per-axis step budgets: the record table carries two empty cells around the grid, and a walk ends once it passes the
// parameter at which the ray leaves the grid box plus a quarter of its smallest cell step (walk2_limits), which happens
// inside that border. Same cells in the same order, same pre-test, same exact tests, same order-free update as
// trace_segment -> same results; the tests compare both with the brute-force path bit for bit.
#ifndef RT_WALK2_WAVES
#define RT_WALK2_WAVES 6
#endif
#ifndef RT_WALK2_REFILL_MIN
#define RT_WALK2_REFILL_MIN RT_REFILL_MIN
#endif
#ifndef RT_WALK2_DEFER_PENDING
#define RT_WALK2_DEFER_PENDING RT_DEFER_PENDING
#endif



template <bool FUSED, bool ANY, bool STATS>
__device__ __forceinline__ void walk_segment(const WfParams& w, const uint32_t* __restrict__ queue, uint32_t n_queue,
                                             uint32_t wave, uint32_t n_waves, uint32_t* __restrict__ run_ctr,
                                             unsigned long long& tested) {
    const uint32_t lane = threadIdx.x & 63u;
    RunCursor rc;   // runs of queue entries and their tickets (wave-uniform)
    if (!rc.begin(n_queue, wave, n_waves, run_ctr, lane)) return;
    uint32_t& next = rc.next;
    const uint32_t& seg_end = rc.seg_end;
    auto grab = [&]() -> bool { return rc.grab(n_queue, run_ctr, lane); };
    bool more = rc.dynamic;
    const GridDesc& g = w.grid;
    const HotObject* __restrict__ hot = w.rp.scene.hot;
    const float4* __restrict__ rec = g.walk_rec;

    unsigned long long s_rays = 0, s_trips = 0, s_live = 0, s_fetch = 0, s_pre = 0, s_flush = 0, s_refill = 0;  // STATS only
    bool alive = false;   // the lane holds a ray ...
    bool over = false;    // ... whose walk has ended (it may still wait for its parked exact test)
    uint32_t pix = 0;
    float rsx = 0.f, rsy = 0.f, rsz = 0.f, rdx = 0.f, rdy = 0.f, rdz = 0.f, dd = 0.f;
    uint32_t cursor = 0, cell = 0;            // record to look at next; head record of the current cell
    float tx = 0.f, ty = 0.f, tz = 0.f, dtx = 0.f, dty = 0.f, dtz = 0.f;
    int stx = 0, sty = 0, stz = 0;
    float T = kMaxFloat, limit = 0.f, t_stop = 0.f;
    float slack = 0.f;                        // closest: walk slack past the best hit; shadow rays: the light tile's cut (+inf: none)
    int idx = -1;
    bool cur_sphere = false;
    bool pend = false;
    bool in_lt_blocks = false;  // shadow ray of the last light whose list is a chain of three-candidate blocks (LightTiles::blocks)
    uint32_t pend_k = 0, done_k = 0xffffffffu;

    for (;;) {
        // ---- hand out rays to idle lanes ----
        const unsigned long long idle = __ballot(!alive);
        if (next < seg_end && ((uint32_t)__popcll(idle) >= (uint32_t)RT_WALK2_REFILL_MIN || idle == ~0ull)) {
            const uint32_t mine = next + __builtin_amdgcn_mbcnt_hi((uint32_t)(idle >> 32), __builtin_amdgcn_mbcnt_lo((uint32_t)idle, 0u));
            if (STATS && lane == 0u) ++s_refill;
            if (!alive && mine < seg_end) {
                if (STATS) ++s_rays;
                const uint32_t entry = (!ANY && w.identity_queue) ? mine : queue[mine];
                pix = (!ANY && w.identity_queue) ? mine : (entry & kQueuePixel);
                uint32_t note;  // shadow rays: the light they go to; reflection rays: begin_shade_lit's note (the object they leave)
                Ray ray;
                if (ANY) {
                    ray = shadow_of_pixel<FUSED>(w, pix, (entry & kQueueLastLight) != 0u, note);
                } else {
                    ray = closest_ray(w, pix, w.first_round != 0u);
                    note = __float_as_uint(ray.dw);
                }
                ray.sw = 1.0f; ray.dw = 0.0f;  // what every ray of a grid-able frame carries (rt_create checks the preconditions)
                rsx = ray.sx; rsy = ray.sy; rsz = ray.sz; rdx = ray.dx; rdy = ray.dy; rdz = ray.dz;
                T = kMaxFloat; idx = -1; cur_sphere = false; pend = false; over = false; in_lt_blocks = false;
                done_k = (!ANY && w.first_round == 0u) ? note : 0xffffffffu;
                dd = rdx * rdx + rdy * rdy + rdz * rdz;
                const float slk = dd > 0.f ? kWalkSlackCells * g.cell * __builtin_amdgcn_rsqf(dd) * 1.0001f : 3.0e38f;
                bool start = false;   // a walk begins
                bool brute = false;   // not this walk's kind of ray: every object is tested here and now
                if (ANY && w.ltiles.enabled && note == w.ltiles.light) {
                    // the last light's shadow ray: one tile of the light's own perspective holds every object it can meet
                    uint32_t tile;
                    float cut;
                    const float chk = ((rsx + rsy) + rsz) + ((rdx + rdy) + rdz);
                    if (chk == chk && light_tile_of(w.ltiles, rsx, rsy, rsz, tile, cut)) {
                        if (w.ltiles.blocks_enabled) {
                            // the tile's chain of three-candidate blocks; its head is block `tile` (an empty tile's head holds nothing
                            // nearer than "beyond": the first trip ends the walk)
                            start = true;
                            cursor = tile;
                            in_lt_blocks = true;
                            slack = cut;
                            limit = -1.0f;
                        } else {
                            const uint2 range = table_at(w.ltiles.tile_range, tile);
                            if (range.y != 0u) {
                                start = true;
                                cursor = w.ltiles.walk_base + range.x;
                                slack = cut;        // how far from the light an occluder can start
                                limit = -1.0f;      // the list is all there is: its end ends the walk (0 > -1)
                                tx = ty = tz = 0.f; dtx = dty = dtz = 0.f;
                                stx = sty = stz = 0; cell = 0u;
                            }
                        }
                    }
                } else {
                    const Walk w0 = walk_begin(g, ray, ANY ? 1.0f + slk : 3.0e38f);  // (a ray with a NaN in it: not alive)
                    if (w0.alive) {
                        // The unified walk is made for rays whose cell steps are exact enough to end inside the table's
                        // two-cell border: a direction of sane magnitude and an origin in (or near) the grid box - every
                        // ray a frame produces by itself. Anything else (caller-made primary rays from far away, with
                        // directions of 1e20 ...) tests every object instead: rare, slow, and the same result.
                        const float dmin = __builtin_fminf(__builtin_fminf(w0.dtx, w0.dty), w0.dtz);
                        const bool tame = dd > 1.0e-30f && dd < 1.0e30f && w0.t_enter <= 4096.f * dmin;
                        brute = !tame;
                        start = tame;
                        cell = ((uint32_t)(w0.iz + 2) * g.walk_nxy) + ((uint32_t)(w0.iy + 2) * g.walk_nx) + (uint32_t)(w0.ix + 2);
                        cursor = cell;
                        tx = w0.tx; ty = w0.ty; tz = w0.tz; dtx = w0.dtx; dty = w0.dty; dtz = w0.dtz;
                        stx = w0.stepx > 0 ? 1 : -1;
                        sty = w0.stepy > 0 ? (int)g.walk_nx : -(int)g.walk_nx;
                        stz = w0.stepz > 0 ? (int)g.walk_nxy : -(int)g.walk_nxy;
                        // past this parameter the ray is outside the grid box (t_exit), by at most a quarter of a cell step along
                        // any axis: still inside the border, no further cell of the box ahead
                        t_stop = __builtin_fminf(w0.t_exit + 0.25f * dmin, 3.0e38f);
                        limit = t_stop;  // (closest: T is +MAX until something is hit; shadow: t_exit <= 1 + slack already)
                        slack = ANY ? __builtin_inff() : slk;
                    }
                }
                bool done = false;    // shadow rays: occluded already
                {   // objects every ray must test (usually none); a ray the walk is not made for tests them all
                    const uint32_t n_loop = brute ? w.rp.scene.n_objs : g.n_always;
                    for (uint32_t a = 0; a < n_loop && !done; ++a) {
                        const int k = brute ? (int)a : (int)g.always[a];
                        float t;
                        bool sphere;
                        const bool cand = lane_candidate<FUSED, true, false>(hot + k, ray, t, sphere);
                        if (STATS) ++tested;
                        if (ANY) done = cand && !(t >= 1.f);
                        else if (cand) closest_take(t, k, sphere, T, idx, cur_sphere);
                    }
                    if (!ANY) limit = __builtin_fminf(T + slack, t_stop);
                }
                if (done || !start) {  // occluded by an always-object, no cell / light tile to look at, or everything tested already
                    if (ANY) U(w, F_RES_ANY, pix) = (done || nan_shadow_blocked(w.rp.scene, ray)) ? 0u : 1u;
                    else store_closest_result(w, pix, T, idx);
                } else {
                    alive = true;
                }
            }
            next += (uint32_t)__popcll(idle);
            if (next >= seg_end && more) more = grab();  // on to another run, if any is left
        }
        const unsigned long long live = __ballot(alive);
        if (live == 0ull) {
            if (next >= seg_end) break;
            continue;
        }
        // ---- one trip: the record under the cursor ----
        const bool walking = alive && !over;
        bool stall = false;
        if (STATS) { if (lane == 0u) { ++s_trips; s_live += (unsigned long long)__popcll(live); } if (walking) { ++s_pre; if (cursor < g.walk_cells) ++s_fetch; } }
        if (ANY && walking && in_lt_blocks) {
            // ---- a light tile's block: three candidates per 32-byte fetch (the shadow walk was paying for 2.4 record fetches per ray)
            const LightTiles& lt = w.ltiles;
            const uint32_t b = cursor & kLtBlockIndexMask, pos = cursor >> kLtBlockIndexBits;
            const uint4 q0 = table_at(lt.blocks, 2u * b), q1 = table_at(lt.blocks, 2u * b + 1u);
            // (the block's ids are requested WITH the block, not after its pre-tests: 0.8 of a shadow ray's 1.1 blocks have a
            // candidate that passes, and the walk is a chain of dependent fetches - 16 bytes more per block, one step less per ray)
            const uint4 kk = table_at(reinterpret_cast<const uint4*>(lt.block_ids), b);
            const uint32_t nxt = q0.x;
            const uint32_t lo[3] = {q0.z, q1.x, q1.z}, hi[3] = {q0.w, q1.y, q1.w};
            uint32_t pm = 0u;
            bool beyond = false;  // sorted by distance from the light: an entry beyond the ray's origin ends the list
#pragma unroll
            for (uint32_t e = 0; e < 3u; ++e) {
                float4 sphere;
                float key;
                lt_block_entry(lt, lo[e], hi[e], sphere, key);
                beyond = beyond || key > slack;
                const bool pass = !beyond && !misses_bounding_sphere2(sphere, rsx, rsy, rsz, rdx, rdy, rdz, dd, g.pretest_alpha);
                pm |= pass ? (1u << e) : 0u;
            }
            pm &= 0x7u << pos;  // (a block the lane comes back to: the entries before `pos` have been dealt with)
            uint32_t stalled = 0u, back = 0u, parked = pend ? 1u : 0u;
            while (pm != 0u) {  // the ids of the candidates that passed
                const uint32_t e = (uint32_t)__builtin_ctz(pm);
                const uint32_t k = e == 0u ? kk.x : (e == 1u ? kk.y : kk.z);
                const bool dup = (k == done_k) || (parked != 0u && k == pend_k);
                const bool wait = !dup && parked != 0u;   // one parking slot: wait for the exact tests, resume at this entry
                const bool take = !dup && parked == 0u;
                pend_k = take ? k : pend_k;
                parked = take ? 1u : parked;
                stalled = wait ? 1u : stalled;
                back = wait ? e : back;
                pm = wait ? 0u : (pm & (pm - 1u));
            }
            pend = parked != 0u;
            stall = stalled != 0u;
            over = !stall && (beyond || nxt == 0u);
            cursor = stall ? (b | (back << kLtBlockIndexBits)) : nxt;
        } else if (walking) {
            const float4 a = table_at(rec, 2u * cursor);
            const float4 b = table_at(rec, 2u * cursor + 1u);
            const uint32_t k = __float_as_uint(b.x), nxt = __float_as_uint(b.y);
            bool pass = !misses_bounding_sphere2(a, rsx, rsy, rsz, rdx, rdy, rdz, dd, g.pretest_alpha);
            bool beyond = false;
            if (ANY) {  // light tiles: sorted by distance from the light - this entry and all after it lie beyond the ray's origin
                beyond = b.z > slack;
                pass = pass && !beyond;
            }
            // the same object again (parked, or tested a cell ago)? its result is known or on its way
            pass = pass && (k != done_k) && !(pend && k == pend_k);
            stall = pass && pend;               // one parking slot: wait for the exact tests, look at this record again
            const bool park = pass && !pend;
            pend_k = park ? k : pend_k;
            pend = pend || park;
            const bool adv = !stall && nxt == 0u;  // the list ends here: on to the next cell
            // the DDA step, computed by every lane, taken by those that advance (ties: x before y before z, as walk_next)
            const float tmin = __builtin_fminf(__builtin_fminf(tx, ty), tz);
            const bool ax = (tx <= ty) && (tx <= tz);
            const bool ay = !ax && (ty <= tz);
            const bool az = !ax && !ay;
            tx += (adv && ax) ? dtx : 0.f;
            ty += (adv && ay) ? dty : 0.f;
            tz += (adv && az) ? dtz : 0.f;
            const int step = ax ? stx : (ay ? sty : stz);
            cell += adv ? (uint32_t)step : 0u;
            over = (adv && tmin > limit) || beyond;
            cursor = stall ? cursor : (adv ? cell : nxt);
        }
        // ---- the exact tests, when enough lanes wait for them ----
        const unsigned long long pending = __ballot(pend);
        if (pending != 0ull) {
            const unsigned long long stuck = __ballot(pend && (stall || over));
            const uint32_t n_live = (uint32_t)__popcll(live);
            if ((uint32_t)__popcll(pending) >= (uint32_t)RT_WALK2_DEFER_PENDING ||
                ((uint32_t)__popcll(stuck) << RT_DEFER_STUCK_SHIFT) >= n_live) {
                if (STATS && lane == 0u) ++s_flush;
                if (pend) {
                    float t;
                    bool sphere;
                    const Ray ray = {rsx, rsy, rsz, 1.0f, rdx, rdy, rdz, 0.0f};
                    const bool cand = lane_candidate<FUSED, true, false>(hot + pend_k, ray, t, sphere);
                    if (STATS) ++tested;
                    pend = false;
                    done_k = pend_k;
                    if (ANY) {
                        if (cand && !(t >= 1.f)) { U(w, F_RES_ANY, pix) = 0u; alive = false; }
                    } else if (cand) {
                        closest_take(t, (int)pend_k, sphere, T, idx, cur_sphere);
                        limit = __builtin_fminf(T + slack, t_stop);
                    }
                }
            }
        }
        // ---- a finished walk with nothing parked: the ray is done ----
        if (alive && over && !pend) {
            if (ANY) U(w, F_RES_ANY, pix) = 1u;  // nothing in the way
            else store_closest_result(w, pix, T, idx);
            alive = false;
        }
    }
    if (STATS) {
        unsigned long long* acc = w.rp.counters->walk[ANY ? 1 : 0];
        const unsigned long long v[8] = {wave_sum64(s_rays), s_trips, s_live, wave_sum64(s_fetch), wave_sum64(s_pre), wave_sum64(tested), s_flush, s_refill};
        if (lane == 0u)
            for (int j = 0; j < 8; ++j)
                if (v[j]) atomicAdd(&acc[j], v[j]);
    }
}

template <bool FUSED, bool ANY, bool STATS>
__global__ __launch_bounds__(256) __attribute__((amdgpu_waves_per_eu(RT_WALK2_WAVES))) void wf_walk(const WfParams wk, uint32_t* __restrict__ run_ctr) {
    WfParams w = wk;
    if (!resolve_round(w)) return;
    const uint32_t* queue = ANY ? w.q_prev_any : w.q_prev_closest;
    const uint32_t n_queue = ANY ? w.n_prev_any : w.n_prev_closest;
    if (n_queue == 0u) return;
    const uint32_t wave = (blockIdx.x * 256u + threadIdx.x) >> 6;
    const uint32_t n_waves = (gridDim.x * 256u) >> 6;
    unsigned long long tested = 0;
    walk_segment<FUSED, ANY, STATS>(w, queue, n_queue, wave, n_waves, run_ctr, tested);
    if (STATS && tested) atomicAdd(&w.rp.counters->tests, tested);
}

// ---- grid traversal, block form (closest-hit rays, scenes without triangles) ----------------------------------------------
// rt_grid.h: BlockGrid says why: the walk is bound by the cache lines its lanes pull, so a trip fetches ONE 32-byte block
// with up to seven candidates (8-bit lattice spheres) and pre-tests them all; object ids are fetched only for the few that
// pass. Everything else is walk_segment: one parking slot, exact tests run for the whole wave, order-free update, walks that
// end by parameter inside the table's empty border.
#ifndef RT_WALK3_WAVES
#define RT_WALK3_WAVES 6
#endif
#ifndef RT_WALK3_REFILL_MIN
#define RT_WALK3_REFILL_MIN RT_WALK2_REFILL_MIN
#endif
#ifndef RT_WALK3_DEFER_PENDING
#define RT_WALK3_DEFER_PENDING 16  // exact tests of the block walk once this many lanes hold a candidate (a trip is ~300 instructions here, the
#endif                             // test ~150): cfg4 frame with 8 / 12 / 16 / 24: 12.07 / 11.93 / 11.99 / 12.06 ms, 16 with half the lanes stuck: 11.92
#ifndef RT_WALK3_STUCK_SHIFT
#define RT_WALK3_STUCK_SHIFT 1     // ... or once HALF of the live lanes cannot move without theirs (walk_segment: a quarter)
#endif
#ifndef RT_WALK3_DEFER_PENDING_TRI
#define RT_WALK3_DEFER_PENDING_TRI RT_WALK2_DEFER_PENDING  // meshes keep the earlier pair (8, a quarter): cfg5 38.97 ms against 40.15 with 16 / half
#endif
#ifndef RT_WALK3_STUCK_SHIFT_TRI
#define RT_WALK3_STUCK_SHIFT_TRI RT_DEFER_STUCK_SHIFT
#endif
#ifndef RT_STATS_TAIL
#define RT_STATS_TAIL 0
#endif
#ifndef RT_BLOCK_SKIP_CAP
#define RT_BLOCK_SKIP_CAP 8   // empty-space steps a lane takes per trip beyond the first (the other lanes of the wave wait for it)
#endif

#ifndef RT_LATTICE_BEHIND
#define RT_LATTICE_BEHIND 0  // 1: also reject spheres entirely behind the origin (3 more instructions per candidate, ~15 % fewer exact tests: cfg4 +0.33 ms)
#endif
// candidate `word` = {x, y, z, r} (8 bits each) of a block's lattice against a ray whose origin is (olx, oly, olz) in lattice
// coordinates: the pre-test of misses_bounding_sphere (a similarity transform leaves every sign it looks at unchanged),
// written without boolean plumbing - the walk runs seven of these per trip and was paying ~8 scalar instructions of mask
// logic for each. With oo = |o - c|^2, od = (c - o) . d:
//     line misses:      od^2 - dd (oo (1 - alpha) - r^2) < 0
//     entirely behind:  od < 0  and  oo (1 - 1e-5) - r^2 > 0
//     miss  <=>  max(-disc, min(-od, behind)) > 0          (no NaN can arise for the rays this walk accepts)
// Returns pm with the verdict shifted in from the right (1 = passes): the caller feeds the entries last to first.
__device__ __forceinline__ uint32_t lattice_pretest(uint32_t pm, uint32_t word, float olx, float oly, float olz, float dx, float dy, float dz,
                                                    float neg_dd, float one_minus_alpha) {
    const float x = (float)(word & 0xffu), y = (float)((word >> 8) & 0xffu), z = (float)((word >> 16) & 0xffu), r = (float)(word >> 24);
    const float ox = x - olx, oy = y - oly, oz = z - olz;
    const float oo = __builtin_fmaf(oz, oz, __builtin_fmaf(oy, oy, ox * ox));
    const float od = __builtin_fmaf(oz, dz, __builtin_fmaf(oy, dy, ox * dx));
    const float r2 = r * r;
    const float disc = __builtin_fmaf(neg_dd, __builtin_fmaf(one_minus_alpha, oo, -r2), od * od);
#if RT_LATTICE_BEHIND
    const float behind = __builtin_fmaf(0.99999f, oo, -r2);
    const float s = __builtin_fmaxf(-disc, __builtin_fminf(-od, behind));
    return pm + pm + (s > 0.f ? 0u : 1u);
#else
    return pm + pm + (disc < 0.f ? 0u : 1u);
#endif
}

// all candidates of one block (entries last to first, so that bit e of the result is entry e)
__device__ __forceinline__ uint32_t block_pretests(const uint4 q0, const uint4 q1, float olx, float oly, float olz, float dx, float dy, float dz,
                                                   float neg_dd, float one_minus_alpha) {
    uint32_t pm = 0u;
    if constexpr (kBlockEntries > 6) pm = lattice_pretest(pm, q1.w, olx, oly, olz, dx, dy, dz, neg_dd, one_minus_alpha);
    if constexpr (kBlockEntries > 5) pm = lattice_pretest(pm, q1.z, olx, oly, olz, dx, dy, dz, neg_dd, one_minus_alpha);
    if constexpr (kBlockEntries > 4) pm = lattice_pretest(pm, q1.y, olx, oly, olz, dx, dy, dz, neg_dd, one_minus_alpha);
    if constexpr (kBlockEntries > 3) pm = lattice_pretest(pm, q1.x, olx, oly, olz, dx, dy, dz, neg_dd, one_minus_alpha);
    if constexpr (kBlockEntries > 2) pm = lattice_pretest(pm, q0.w, olx, oly, olz, dx, dy, dz, neg_dd, one_minus_alpha);
    if constexpr (kBlockEntries > 1) pm = lattice_pretest(pm, q0.z, olx, oly, olz, dx, dy, dz, neg_dd, one_minus_alpha);
    return lattice_pretest(pm, q0.y, olx, oly, olz, dx, dy, dz, neg_dd, one_minus_alpha);
}

template <bool FUSED, bool STATS, bool TRI>
__device__ __forceinline__ void block_segment(const WfParams& w, const uint32_t* __restrict__ queue, uint32_t n_queue,
                                              uint32_t wave, uint32_t n_waves, uint32_t* __restrict__ run_ctr,
                                              unsigned long long& tested) {
    const uint32_t lane = threadIdx.x & 63u;
    RunCursor rc;   // runs of queue entries and their tickets (wave-uniform)
    if (!rc.begin(n_queue, wave, n_waves, run_ctr, lane)) return;
    uint32_t& next = rc.next;
    const uint32_t& seg_end = rc.seg_end;
    auto grab = [&]() -> bool { return rc.grab(n_queue, run_ctr, lane); };
    bool more = rc.dynamic;
    const GridDesc& g = w.grid;
    const BlockGrid& bg = w.bgrid;
    const HotObject* __restrict__ hot = w.rp.scene.hot;
    const float wnx_f = (float)bg.wnx, wny_f = (float)bg.wny;

    unsigned long long s_rays = 0, s_trips = 0, s_live = 0, s_fetch = 0, s_pre = 0, s_flush = 0, s_refill = 0;  // STATS only
    unsigned long long t_dry = 0; uint32_t my_trips = 0, max_trips = 0, trips_after = 0;  // STATS + RT_STATS_TAIL
    // The lane's state bits live in ONE vector register and are changed with vector and / or: as `bool`s carried round
    // the loop the compiler keeps them as 64-bit lane masks in scalar registers, and every merge of divergent paths costs
    // an andn2 / and / or triple per flag (round 2's walk: 0.75 scalar instructions per vector instruction).
    constexpr uint32_t kAlive = 1u;    // the lane holds a ray ...
    constexpr uint32_t kOver = 2u;     // ... whose walk has ended (it may still wait for its parked exact test)
    constexpr uint32_t kPend = 4u;     // a candidate is parked for the next round of exact tests (pend_k)
    constexpr uint32_t kSphere = 8u;   // the current best hit is a sphere (closest_take's tie rule)
    uint32_t fl = 0u;
    uint32_t pix = 0;
    float rsx = 0.f, rsy = 0.f, rsz = 0.f, rdx = 0.f, rdy = 0.f, rdz = 0.f, dd = 0.f;
    uint32_t cur = 0;                         // block to look at next (bits 0-23) and the entry to resume it at (bits 24-26)
    float fx = 0.f, fy = 0.f, fz = 0.f;       // current cell, padded coordinates (whole numbers)
    float tx = 0.f, ty = 0.f, tz = 0.f, dtx = 0.f, dty = 0.f, dtz = 0.f;
    float T = kMaxFloat, limit = 0.f, t_stop = 0.f, slack = 0.f;
    int idx = -1;
    uint32_t pend_k = 0, done_k = 0xffffffffu;

    for (;;) {
        // ---- hand out rays to idle lanes ----
        const unsigned long long idle = __ballot((fl & kAlive) == 0u);
        if (next < seg_end && ((uint32_t)__popcll(idle) >= (uint32_t)RT_WALK3_REFILL_MIN || idle == ~0ull)) {
            const uint32_t mine = next + __builtin_amdgcn_mbcnt_hi((uint32_t)(idle >> 32), __builtin_amdgcn_mbcnt_lo((uint32_t)idle, 0u));
            if (STATS && lane == 0u) ++s_refill;
            if ((fl & kAlive) == 0u && mine < seg_end) {
                if (STATS) { ++s_rays; my_trips = 0; }
                const uint32_t entry = w.identity_queue ? mine : queue[mine];
                pix = w.identity_queue ? mine : (entry & kQueuePixel);
                Ray ray = closest_ray(w, pix, w.first_round != 0u);
                const uint32_t note = __float_as_uint(ray.dw);  // begin_shade_lit's note: the object a reflection ray leaves
                ray.sw = 1.0f; ray.dw = 0.0f;  // what every ray of a grid-able frame carries (rt_create checks the preconditions)
                rsx = ray.sx; rsy = ray.sy; rsz = ray.sz; rdx = ray.dx; rdy = ray.dy; rdz = ray.dz;
                T = kMaxFloat; idx = -1;
                bool cur_sphere = false;
                done_k = (w.first_round == 0u) ? note : 0xffffffffu;
                dd = rdx * rdx + rdy * rdy + rdz * rdz;
                slack = dd > 0.f ? kWalkSlackCells * bg.cell * __builtin_amdgcn_rsqf(dd) * 1.0001f : 3.0e38f;
                bool start = false, brute = false;
                const Walk w0 = walk_begin(bg, ray, 3.0e38f);  // (a ray with a NaN in it: not alive)
                if (w0.alive) {
                    // (walk_segment: the rays this walk is made for; anything else tests every object here and now. Since round 4 a
                    // frame whose PRIMARY directions fail this window is rendered by the literal loops as a whole - rt_create's ray scan,
                    // rt_set_camera - and every origin lies in the grid box, so what is left for this branch are secondary rays with
                    // |d|^2 outside (1e-30, 1e30): a light within 1e-15 of a hit point. Rare, slow, and the same result.)
                    const float dmin = __builtin_fminf(__builtin_fminf(w0.dtx, w0.dty), w0.dtz);
                    const bool tame = dd > 1.0e-30f && dd < 1.0e30f && w0.t_enter <= 4096.f * dmin;
                    brute = !tame;
                    start = tame;
                    fx = (float)(w0.ix + (int)kBlockBorder); fy = (float)(w0.iy + (int)kBlockBorder); fz = (float)(w0.iz + (int)kBlockBorder);
                    cur = ((uint32_t)(w0.iz + (int)kBlockBorder) * bg.wny + (uint32_t)(w0.iy + (int)kBlockBorder)) * bg.wnx + (uint32_t)(w0.ix + (int)kBlockBorder);
                    tx = w0.tx; ty = w0.ty; tz = w0.tz; dtx = w0.dtx; dty = w0.dty; dtz = w0.dtz;
                    t_stop = __builtin_fminf(w0.t_exit + 0.25f * dmin, 3.0e38f);
                }
                {   // objects every ray must test (usually none); a ray the walk is not made for tests them all
                    const uint32_t n_loop = brute ? w.rp.scene.n_objs : g.n_always;
                    for (uint32_t a = 0; a < n_loop; ++a) {
                        const int k = brute ? (int)a : (int)g.always[a];
                        float t;
                        bool sphere;
                        const bool cand = lane_candidate<FUSED, true, TRI>(hot + k, ray, t, sphere);
                        if (STATS) ++tested;
                        if (cand) closest_take(t, k, sphere, T, idx, cur_sphere);
                    }
                    limit = __builtin_fminf(T + slack, t_stop);
                }
                if (!start) store_closest_result(w, pix, T, idx);  // no cell to look at, or everything tested already
                fl = start ? (kAlive | (cur_sphere ? kSphere : 0u)) : 0u;
            }
            next += (uint32_t)__popcll(idle);
            if (next >= seg_end && more) more = grab();  // on to another run, if any is left
        }
        const unsigned long long live = __ballot((fl & kAlive) != 0u);
        if (live == 0ull) {
            if (next >= seg_end) break;
            continue;
        }
        // ---- one trip: the block under the cursor ----
        const bool walking = (fl & (kAlive | kOver)) == kAlive;
        uint32_t stalled = 0u;
        if (STATS) { if (lane == 0u) { ++s_trips; s_live += (unsigned long long)__popcll(live); } if (walking) { s_pre += kBlockEntries; ++my_trips; max_trips = my_trips > max_trips ? my_trips : max_trips; if ((cur & 0xffffffu) < bg.n_cells) ++s_fetch; } if (RT_STATS_TAIL && t_dry == 0 && next >= seg_end && !more) t_dry = wall_clock64(); if (t_dry != 0) ++trips_after; }
        if (walking) {
            const uint32_t b = cur & 0xffffffu, pos = cur >> 24;
            const uint4 q0 = table_at(bg.blocks, 2u * b);
            const uint4 q1 = table_at(bg.blocks, 2u * b + 1u);
            const uint32_t nxt = q0.x & 0xffffffu;
            const float inv = __builtin_ldexpf(bg.inv_step, -(int)((q0.x >> 27) & 3u));
            // the ray's origin in the lattice of this block: 256 steps centred on the current cell
            const float olx = __builtin_fmaf(rsx - __builtin_fmaf(fx, bg.cell, bg.c0x), inv, 128.0f);
            const float oly = __builtin_fmaf(rsy - __builtin_fmaf(fy, bg.cell, bg.c0y), inv, 128.0f);
            const float olz = __builtin_fmaf(rsz - __builtin_fmaf(fz, bg.cell, bg.c0z), inv, 128.0f);
            // candidates that pass the pre-test: bit e = entry e
            uint32_t pm = block_pretests(q0, q1, olx, oly, olz, rdx, rdy, rdz, -dd, 1.0f - g.pretest_alpha);
            pm &= 0x7fu << pos;  // (a block the lane comes back to: the entries before `pos` have been dealt with)
            uint32_t back = 0u;
            while (pm != 0u) {  // few lanes, seldom more than once: the ids of the candidates that passed
                const uint32_t e = (uint32_t)__builtin_ctz(pm);
                const uint32_t k = table_at(bg.ids, 8u * b + e);
                // the same object again (parked, or tested a cell ago)? its result is known or on its way
                const bool parked = (fl & kPend) != 0u;
                const bool dup = (k == done_k) || (parked && k == pend_k);
                const bool wait = !dup && parked;   // one parking slot: wait for the exact tests, resume at this entry
                const bool take = !dup && !parked;
                pend_k = take ? k : pend_k;
                fl |= take ? kPend : 0u;
                stalled = wait ? 1u : stalled;
                back = wait ? e : back;
                pm = wait ? 0u : (pm & (pm - 1u));
            }
            const bool adv = stalled == 0u && nxt == 0u;  // the chain ends here: on to the next cell
            // the DDA step, computed by every lane, taken by those that advance (ties: x before y before z, as walk_next)
            const float tmin = __builtin_fminf(__builtin_fminf(tx, ty), tz);
            const bool ax = (tx <= ty) && (tx <= tz);
            const bool ay = !ax && (ty <= tz);
            const bool az = !ax && !ay;
            tx += (adv && ax) ? dtx : 0.f;
            ty += (adv && ay) ? dty : 0.f;
            tz += (adv && az) ? dtz : 0.f;
            fx += (adv && ax) ? __builtin_copysignf(1.0f, rdx) : 0.f;
            fy += (adv && ay) ? __builtin_copysignf(1.0f, rdy) : 0.f;
            fz += (adv && az) ? __builtin_copysignf(1.0f, rdz) : 0.f;
            fl |= (adv && tmin > limit) ? kOver : 0u;
            {   // empty space: an empty cell's header says how many further steps stay in empty cells - taken here without fetching
                // their blocks (same cells, same order, same end test; at most RT_BLOCK_SKIP_CAP per trip: the other lanes wait)
                uint32_t more = ((q0.x >> 24) & 7u) | ((q0.x >> 26) & 0x38u);
                more = (adv && (fl & kOver) == 0u) ? (more < (uint32_t)RT_BLOCK_SKIP_CAP ? more : (uint32_t)RT_BLOCK_SKIP_CAP) : 0u;
                if (!bg.take_skips) more = 0u;  // (wave-uniform: crowded grids do not bother)
                while (more != 0u) {
                    const float tm = __builtin_fminf(__builtin_fminf(tx, ty), tz);
                    const bool sx = (tx <= ty) && (tx <= tz);
                    const bool sy = !sx && (ty <= tz);
                    const bool sz = !sx && !sy;
                    tx += sx ? dtx : 0.f;
                    ty += sy ? dty : 0.f;
                    tz += sz ? dtz : 0.f;
                    fx += sx ? __builtin_copysignf(1.0f, rdx) : 0.f;
                    fy += sy ? __builtin_copysignf(1.0f, rdy) : 0.f;
                    fz += sz ? __builtin_copysignf(1.0f, rdz) : 0.f;
                    const bool out = tm > limit;
                    fl |= out ? kOver : 0u;
                    more = out ? 0u : more - 1u;
                }
            }
            const uint32_t cell = (uint32_t)__builtin_fmaf(__builtin_fmaf(fz, wny_f, fy), wnx_f, fx);  // (whole numbers below 2^24: exact)
            cur = stalled != 0u ? (b | (back << 24)) : (adv ? cell : nxt);
        }
        // ---- the exact tests, when enough lanes wait for them ----
        const unsigned long long pending = __ballot((fl & kPend) != 0u);
        if (pending != 0ull) {
            const unsigned long long stuck = __ballot((fl & kPend) != 0u && (stalled != 0u || (fl & kOver) != 0u));
            const uint32_t n_live = (uint32_t)__popcll(live);
            if ((uint32_t)__popcll(pending) >= (uint32_t)(TRI ? RT_WALK3_DEFER_PENDING_TRI : RT_WALK3_DEFER_PENDING) ||
                ((uint32_t)__popcll(stuck) << (TRI ? RT_WALK3_STUCK_SHIFT_TRI : RT_WALK3_STUCK_SHIFT)) >= n_live) {
                if (STATS && lane == 0u) ++s_flush;
                if ((fl & kPend) != 0u) {
                    float t;
                    bool sphere;
                    const Ray ray = {rsx, rsy, rsz, 1.0f, rdx, rdy, rdz, 0.0f};
                    const bool cand = lane_candidate<FUSED, true, TRI>(hot + pend_k, ray, t, sphere);
                    if (STATS) ++tested;
                    done_k = pend_k;
                    bool cur_sphere = (fl & kSphere) != 0u;
                    if (cand) {
                        closest_take(t, (int)pend_k, sphere, T, idx, cur_sphere);
                        limit = __builtin_fminf(T + slack, t_stop);
                    }
                    fl = (fl & ~(kPend | kSphere)) | (cur_sphere ? kSphere : 0u);
                }
            }
        }
        // ---- a finished walk with nothing parked: the ray is done ----
        if ((fl & (kAlive | kOver | kPend)) == (kAlive | kOver)) {
            store_closest_result(w, pix, T, idx);
            fl = 0u;
        }
    }
    if (STATS) {
        unsigned long long* acc = w.rp.counters->walk[0];
#if RT_STATS_TAIL  // (engineering build: the last two columns become the longest end-game of a wave - 10 ns ticks between its queue running dry and its exit - in the low 32 bits + its trips in that time in the high ones; and the longest ray in trips)
        const unsigned long long after = t_dry ? wall_clock64() - t_dry : 0ull;
        if (lane == 0u) { atomicMax(&acc[6], (after << 32) | (unsigned long long)trips_after); }
        uint32_t mt = max_trips;
        for (int o = 32; o > 0; o >>= 1) { const uint32_t other = (uint32_t)__shfl_xor((int)mt, o); mt = other > mt ? other : mt; }
        if (lane == 0u) atomicMax(&acc[7], (unsigned long long)mt);
        const unsigned long long v[8] = {wave_sum64(s_rays), s_trips, s_live, wave_sum64(s_fetch), wave_sum64(s_pre), wave_sum64(tested), 0ull, 0ull};
#else
        const unsigned long long v[8] = {wave_sum64(s_rays), s_trips, s_live, wave_sum64(s_fetch), wave_sum64(s_pre), wave_sum64(tested), s_flush, s_refill};
#endif
        if (lane == 0u)
            for (int j = 0; j < 8; ++j)
                if (v[j]) atomicAdd(&acc[j], v[j]);
    }
}

// The same walk for ONE ray in one thread (wf_finish: the tail of a frame, where a ray's time is its chain of dependent
// fetches - cfg5: 28 blocks + the empty-space steps their headers allow, against 64 cells of the fine grid). Same blocks, same
// pre-test, same exact tests, order-free update: the result of closest_hit_grid. False: not this walk's kind of ray
// (block_segment's `tame`), nothing was done.
template <bool FUSED>
__device__ __forceinline__ bool closest_hit_blocks(const BlockGrid& bg, const GridDesc& g, const HotObject* __restrict__ hot, const Ray& ray,
                                                   uint32_t skip_object, float& T, int& index, uint32_t& tested) {
    const float dd = ray.dx * ray.dx + ray.dy * ray.dy + ray.dz * ray.dz;
    const Walk w0 = walk_begin(bg, ray, 3.0e38f);
    if (!w0.alive) return true;  // misses the grid box, or a NaN in it: no candidates (closest_result patches NaN rays)
    const float dmin = __builtin_fminf(__builtin_fminf(w0.dtx, w0.dty), w0.dtz);
    if (!(dd > 1.0e-30f && dd < 1.0e30f && w0.t_enter <= 4096.f * dmin)) return false;
    const float slack = kWalkSlackCells * bg.cell * __builtin_amdgcn_rsqf(dd) * 1.0001f;
    const float t_stop = __builtin_fminf(w0.t_exit + 0.25f * dmin, 3.0e38f);
    float fx = (float)(w0.ix + (int)kBlockBorder), fy = (float)(w0.iy + (int)kBlockBorder), fz = (float)(w0.iz + (int)kBlockBorder);
    float tx = w0.tx, ty = w0.ty, tz = w0.tz;
    const float wnx_f = (float)bg.wnx, wny_f = (float)bg.wny;
    const float neg_dd = -dd, oma = 1.0f - g.pretest_alpha;
    bool cur_sphere = false;
    uint32_t done_k = skip_object;
    uint32_t b = (uint32_t)__builtin_fmaf(__builtin_fmaf(fz, wny_f, fy), wnx_f, fx);
    for (;;) {
        const uint4 q0 = table_at(bg.blocks, 2u * b);
        const uint4 q1 = table_at(bg.blocks, 2u * b + 1u);
        const float inv = __builtin_ldexpf(bg.inv_step, -(int)((q0.x >> 27) & 3u));
        const float olx = __builtin_fmaf(ray.sx - __builtin_fmaf(fx, bg.cell, bg.c0x), inv, 128.0f);
        const float oly = __builtin_fmaf(ray.sy - __builtin_fmaf(fy, bg.cell, bg.c0y), inv, 128.0f);
        const float olz = __builtin_fmaf(ray.sz - __builtin_fmaf(fz, bg.cell, bg.c0z), inv, 128.0f);
        uint32_t pm = block_pretests(q0, q1, olx, oly, olz, ray.dx, ray.dy, ray.dz, neg_dd, oma);
        while (pm != 0u) {
            const uint32_t k = table_at(bg.ids, 8u * b + (uint32_t)__builtin_ctz(pm));
            pm &= pm - 1u;
            if (k == done_k) continue;  // tested a moment ago (the same object in the next cell; a reflection ray's own object)
            done_k = k;
            float t;
            bool sphere;
            ++tested;
            if (lane_candidate<FUSED, true, true>(hot + k, ray, t, sphere)) closest_take(t, (int)k, sphere, T, index, cur_sphere);
        }
        const uint32_t nxt = q0.x & 0xffffffu;
        if (nxt != 0u) { b = nxt; continue; }
        uint32_t steps = 1u + (((q0.x >> 24) & 7u) | ((q0.x >> 26) & 0x38u));  // (an empty cell: + the steps sure to stay in empty cells)
        bool stop = false;
        const float limit = __builtin_fminf(T + slack, t_stop);
        while (steps-- != 0u && !stop) {
            const float tmin = __builtin_fminf(__builtin_fminf(tx, ty), tz);
            stop = tmin > limit;
            if (stop) break;
            if (tx <= ty && tx <= tz) { tx += w0.dtx; fx += __builtin_copysignf(1.0f, ray.dx); }
            else if (ty <= tz) { ty += w0.dty; fy += __builtin_copysignf(1.0f, ray.dy); }
            else { tz += w0.dtz; fz += __builtin_copysignf(1.0f, ray.dz); }
        }
        if (stop) break;
        b = (uint32_t)__builtin_fmaf(__builtin_fmaf(fz, wny_f, fy), wnx_f, fx);
    }
    return true;
}

// ... and "is anything in the way before t = 1" for one shadow ray in one thread (wf_finish: the light scans of the frame's
// tail go to lights that have no tiles). Same answer as any_hit_grid. `done`: false = not this walk's kind of ray.
template <bool FUSED>
__device__ __forceinline__ bool any_hit_blocks(const BlockGrid& bg, const GridDesc& g, const HotObject* __restrict__ hot, const Ray& ray,
                                               bool& done, uint32_t& tested) {
    done = true;
    const float dd = ray.dx * ray.dx + ray.dy * ray.dy + ray.dz * ray.dz;
    const float slack = dd > 0.f ? kWalkSlackCells * bg.cell * __builtin_amdgcn_rsqf(dd) * 1.0001f : 3.0e38f;
    const Walk w0 = walk_begin(bg, ray, 1.0f + slack);
    if (!w0.alive) { done = false; return false; }  // (off the grid or a NaN ray: the caller's any_hit_grid knows what that means)
    const float dmin = __builtin_fminf(__builtin_fminf(w0.dtx, w0.dty), w0.dtz);
    if (!(dd > 1.0e-30f && dd < 1.0e30f && w0.t_enter <= 4096.f * dmin)) { done = false; return false; }
    const float limit = __builtin_fminf(w0.t_exit + 0.25f * dmin, 3.0e38f);  // (t_exit <= 1 + slack already)
    float fx = (float)(w0.ix + (int)kBlockBorder), fy = (float)(w0.iy + (int)kBlockBorder), fz = (float)(w0.iz + (int)kBlockBorder);
    float tx = w0.tx, ty = w0.ty, tz = w0.tz;
    const float wnx_f = (float)bg.wnx, wny_f = (float)bg.wny;
    const float neg_dd = -dd, oma = 1.0f - g.pretest_alpha;
    uint32_t done_k = 0xffffffffu;
    uint32_t b = (uint32_t)__builtin_fmaf(__builtin_fmaf(fz, wny_f, fy), wnx_f, fx);
    for (;;) {
        const uint4 q0 = table_at(bg.blocks, 2u * b);
        const uint4 q1 = table_at(bg.blocks, 2u * b + 1u);
        const float inv = __builtin_ldexpf(bg.inv_step, -(int)((q0.x >> 27) & 3u));
        const float olx = __builtin_fmaf(ray.sx - __builtin_fmaf(fx, bg.cell, bg.c0x), inv, 128.0f);
        const float oly = __builtin_fmaf(ray.sy - __builtin_fmaf(fy, bg.cell, bg.c0y), inv, 128.0f);
        const float olz = __builtin_fmaf(ray.sz - __builtin_fmaf(fz, bg.cell, bg.c0z), inv, 128.0f);
        uint32_t pm = block_pretests(q0, q1, olx, oly, olz, ray.dx, ray.dy, ray.dz, neg_dd, oma);
        while (pm != 0u) {
            const uint32_t k = table_at(bg.ids, 8u * b + (uint32_t)__builtin_ctz(pm));
            pm &= pm - 1u;
            if (k == done_k) continue;
            done_k = k;
            float t;
            bool sphere;
            ++tested;
            if (lane_candidate<FUSED, true, true>(hot + k, ray, t, sphere) && !(t >= 1.f)) return true;
        }
        const uint32_t nxt = q0.x & 0xffffffu;
        if (nxt != 0u) { b = nxt; continue; }
        uint32_t steps = 1u + (((q0.x >> 24) & 7u) | ((q0.x >> 26) & 0x38u));
        bool stop = false;
        while (steps-- != 0u && !stop) {
            const float tmin = __builtin_fminf(__builtin_fminf(tx, ty), tz);
            stop = tmin > limit;
            if (stop) break;
            if (tx <= ty && tx <= tz) { tx += w0.dtx; fx += __builtin_copysignf(1.0f, ray.dx); }
            else if (ty <= tz) { ty += w0.dty; fy += __builtin_copysignf(1.0f, ray.dy); }
            else { tz += w0.dtz; fz += __builtin_copysignf(1.0f, ray.dz); }
        }
        if (stop) return false;
        b = (uint32_t)__builtin_fmaf(__builtin_fmaf(fz, wny_f, fy), wnx_f, fx);
    }
}

#ifndef RT_WALK3_WAVES_TRI
#define RT_WALK3_WAVES_TRI RT_WALK3_WAVES   // the variant with the triangle branch wants ~95 registers: 6 waves = 17 spilled
#endif
template <bool FUSED, bool STATS, bool TRI>
__global__ __launch_bounds__(256) __attribute__((amdgpu_waves_per_eu(TRI ? RT_WALK3_WAVES_TRI : RT_WALK3_WAVES))) void wf_walk_blocks(const WfParams wk, uint32_t* __restrict__ run_ctr) {
    WfParams w = wk;
    if (!resolve_round(w)) return;
#ifdef RT_WALK3_LDS_PAD  // measurement knob: LDS nobody uses, so that only 160 KB / pad workgroups (x 4 waves) fit a CU
    __shared__ uint32_t occupancy_pad[RT_WALK3_LDS_PAD / 4];
    if (w.n_prev_closest == 0xffffffffu) occupancy_pad[threadIdx.x] = threadIdx.x;
    asm volatile("" ::"v"(&occupancy_pad[0]));
#endif
    const uint32_t n_queue = w.n_prev_closest;
    if (n_queue == 0u) return;
    const uint32_t wave = (blockIdx.x * 256u + threadIdx.x) >> 6;
    const uint32_t n_waves = (gridDim.x * 256u) >> 6;
    unsigned long long tested = 0;
    block_segment<FUSED, STATS, TRI>(w, w.q_prev_closest, n_queue, wave, n_waves, run_ctr, tested);
    if (STATS && tested) atomicAdd(&w.rp.counters->tests, tested);
}

// Literal shadow test: the reference's full closest hit, then its `time >= 1 || time < 0` (:229).
template <bool FUSED>
__global__ __launch_bounds__(256) void wf_trace_any_literal(const WfParams wk) {
    WfParams w = wk;
    if (!resolve_round(w)) return;
    const uint32_t t = blockIdx.x * 256u + threadIdx.x;
    if (t >= w.n_prev_any) return;
    const uint64_t i = w.q_prev_any[t];
    const Ray ray = load_ray(w, i, kSlotShadow);
    float T = kMaxFloat;
    int idx = -1;
    closest_hit<FUSED, true>(w.rp.scene.pairs, w.rp.scene.n_pairs, ray, T, idx);
    U(w, F_RES_ANY, i) = (T >= 1.f || T < 0) ? 1u : 0u;
    const unsigned long long lanes = (unsigned long long)__popcll(__ballot(true));
    if (w.count_rays && (threadIdx.x & 63u) == 0u)
        atomicAdd(&w.rp.counters->tests, 2ull * w.rp.scene.n_pairs * lanes);
}

// Shadow rays, one SLICE of the occluder stream per launch. "Is anything in the way" does not depend on the
// order the objects are asked in, so shadow rays use their own copy of the pair stream, sorted by decreasing
// size (big occluders first), cut into slices. A ray that finds an occluder in a slice is finished
// (visibility 0); only the survivors are queued for the next slice. Every ray thus stops within one slice of
// its first occluder no matter what the other 63 lanes of its wave are doing - the wave-wide early exit of a
// single long loop almost never fires, because one lit lane keeps the whole wave going.
template <bool FUSED>
__global__ __launch_bounds__(256) void wf_trace_any_slice(const WfParams wk, const uint32_t* __restrict__ q_in, const uint32_t* __restrict__ n_in,
                                                          uint32_t pair_lo, uint32_t pair_hi, uint32_t* __restrict__ q_out,
                                                          uint32_t* __restrict__ out_count, uint32_t first_slice) {
    WfParams w = wk;
    if (!resolve_round(w)) return;
    // the first slice reads the round's shadow queue, later ones the survivors of the slice before
    const uint32_t n_queue = first_slice ? w.n_prev_any : *n_in;
    if (first_slice) q_in = w.q_prev_any;
    const uint32_t t = blockIdx.x * 256u + threadIdx.x;
    if (t >= n_queue) return;
    const uint32_t i = q_in[t];
    const Ray ray = load_ray(w, i, kSlotShadow);
    uint32_t visited = 0;
    const int nan_kind = nan_ray_outcome(ray, w.rp.scene.nan_winner, w.rp.scene.nan_winner_sphere);
    if (__ballot(nan_kind != kNanRayNone) != 0ull && nan_kind != kNanRayNone) {  // the size-sorted stream is order-free, a NaN ray's outcome is not
        U(w, F_RES_ANY, i) = (nan_kind == kNanRayTimeNaN) ? 0u : 1u;
        return;
    }
    const bool occluded = any_hit_before_one<FUSED>(w.shadow_pairs + pair_lo, pair_hi - pair_lo, ray, &visited);
    if (w.count_rays) {  // `visited` is wave-uniform: every lane rides along until the wave leaves
        const unsigned long long lanes = (unsigned long long)__popcll(__ballot(true));
        if ((threadIdx.x & 63u) == (uint32_t)__builtin_ctzll(__ballot(true)))
            atomicAdd(&w.rp.counters->tests, 2ull * visited * lanes);
    }
    if (occluded) {
        U(w, F_RES_ANY, i) = 0u;
    } else {
        if (first_slice) U(w, F_RES_ANY, i) = 1u;  // lit unless a later slice says otherwise
        if (q_out) push(q_out, out_count, i);
    }
}

// ---- the resumable pixel -------------------------------------------------------------------------------------------
// Closest-hit result of pixel i. A ray with a NaN in it walks no cells (walk_begin) and comes back as a miss; the
// reference's loop gives such a ray t = NaN on the LAST sphere / box of the scene, because every rejection in its
// tests is a comparison that NaN fails. In practice these are reflections off a box hit whose object-space point
// has no coordinate beyond 0.4998 (tiny far boxes: 0/0 normal). The brute-force kernels reproduce that by
// construction; for the grid path it is patched in here, where the result is consumed.
__device__ __forceinline__ void patch_nan_result(const WfParams& w, const Ray& ray, float& T, int& idx) {
    if (T == kMaxFloat && w.grid.enabled && !w.rp.scene.literal && w.rp.scene.nan_winner >= 0) {
        if (nan_ray_outcome(ray, w.rp.scene.nan_winner, w.rp.scene.nan_winner_sphere) == kNanRayTimeNaN) {
            T = __builtin_nanf("");
            idx = w.rp.scene.nan_winner;
        }
    }
}
__device__ __forceinline__ void closest_result(const WfParams& w, uint64_t i, bool primary, float& T, int& idx) {
    load_closest_result(w, i, T, idx);
    if (T == kMaxFloat && w.grid.enabled && !w.rp.scene.literal && w.rp.scene.nan_winner >= 0) {
        const Ray ray = closest_ray(w, i, primary);
        if (nan_ray_outcome(ray, w.rp.scene.nan_winner, w.rp.scene.nan_winner_sphere) == kNanRayTimeNaN) {
            T = __builtin_nanf("");
            idx = w.rp.scene.nan_winner;
        }
    }
}

struct Ctx {
    const WfParams& w;
    uint64_t i;
    unsigned long long traced, reference, hits;
    bool want_closest, want_any;  // the pixel queued a ray for the next round (appended by block_push at the end)
    uint32_t flags;               // PH_FLAG_* bits of the pixel's phase word
    uint32_t li;                  // ... and its light index
    float4 nblock;                // the block the phase word came in: normal of the hit being shaded
    // Shadow phases: everything the step will read from the pixel's own state, requested in ONE go at the top of the
    // step (resume_pixel) instead of where each piece is used - the step is a chain of dependent memory waits (queue
    // entry -> phase word -> hit -> material -> accumulators -> trace result -> ray -> next hit's records) at 4 waves
    // per SIMD, and these five requests depend on the pixel index only.
    bool pre;
    float4 pre_hit, pre_res, pre_acc;
    Ray pre_ray0;
    unsigned long long tests = 0;  // exact object tests run by the step itself (last_light_blocked)
    uint32_t any_flag = 0;         // kQueueLastLight on the pixel's shadow-queue entry (emit_shadow)
};

template <int KERNEL>
__device__ __forceinline__ void write_pixel(Ctx& c, float r, float g, float b) {
    reinterpret_cast<float4*>(c.w.rp.out)[pixel_of(c.w.rp, c.i)] = make_float4(r, g, b, 1.0f);
    U(c.w, F_PHASE, c.i) = PH_DONE;
}

// queue the shadow ray of light `li` for the hit `h` (its geometry is recomputed when the result arrives)
template <bool FUSED>
__device__ __forceinline__ void emit_shadow(Ctx& c, const HitRec& h, uint32_t li, uint32_t phase, bool new_hit) {
    const Scene& S = c.w.rp.scene;
    // Through the grid the shadow ray is not stored: the trace kernels rebuild it from the hit point (shadow_of_pixel) - 32 bytes
    // per shadow ray that wf_resume does not write and 16 that the walk does not read - and nothing of the light's geometry is
    // needed here. The brute-force / literal trace kernels read the stored record; direction.w (0 for a shadow ray, read by no
    // any-hit test) carries the light index there.
    // (Not for meshes: their shadow walk is the heavier kernel, and the rebuild cost cfg5 58.5 -> 59.8 ms; cfg4: 16.74 -> 16.38.)
    if (shadow_rays_rebuilt(c.w)) {
        c.any_flag = (c.w.ltiles.enabled && li == c.w.ltiles.light) ? kQueueLastLight : 0u;
    } else {
        float nvx = h.nx, nvy = h.ny, nvz = h.nz;
        normalize3(nvx, nvy, nvz);
        float vvx = -h.px, vvy = -h.py, vvz = -h.pz;
        normalize3(vvx, vvy, vvz);
        LightGeom g;
        light_geometry<FUSED>(S.lights[li], h, nvx, nvy, nvz, vvx, vvy, vvz, g);
        g.shadow.dw = __uint_as_float(li);
        store_ray(c.w, c.i, g.shadow, kSlotShadow);
    }
    if (new_hit) store_block(c.w, F_PX, c.i, make_float4(h.px, h.py, h.pz, __uint_as_float((uint32_t)h.index)));
    store_block(c.w, F_NX, c.i, make_float4(h.nx, h.ny, h.nz, __uint_as_float(phase | (li << kPhaseLightShift))));
    c.want_any = true;
    c.traced += 1;
}

template <int KERNEL, bool FUSED> __device__ __forceinline__ void shade_done(Ctx& c, const HitRec& h, bool primary, float cr, float cg, float cb);
__device__ __forceinline__ void finish_reflect(Ctx& c, bool fused, float abr, float abg, float abb, float rr, float rg, float rb, float ap, uint32_t bounces);

// Is the shadow ray towards the LAST light (the light of the light tiles) blocked? One thread, the whole list: the same
// tile, the same cut, the same pre-test and the same exact tests as the light-tile mode of the persistent walk
// (trace_segment, in_lt) - an any-hit answer does not depend on the order or on who asks. (Light tiles are only built for
// scenes without always-tested objects: build_light_tiles.)
template <bool FUSED>
__device__ __forceinline__ bool last_light_blocked(const WfParams& w, Ray ray, uint32_t& tests) {
    ray.sw = 1.0f; ray.dw = 0.0f;  // as the walk's hand-out (rt_create checks the preconditions of a grid-able frame)
    const HotObject* __restrict__ hot = w.rp.scene.hot;
    const float chk = ((ray.sx + ray.sy) + ray.sz) + ((ray.dx + ray.dy) + ray.dz);
    uint32_t tile;
    float cut;
    if (w.ltiles.blocks_enabled && chk == chk && light_tile_of(w.ltiles, ray.sx, ray.sy, ray.sz, tile, cut)) {
        // the tile's chain of 32-byte blocks, three candidates each (the head block sits in an L2-sized table)
        const LightTiles& lt = w.ltiles;
        const float dd = ray.dx * ray.dx + ray.dy * ray.dy + ray.dz * ray.dz;
        for (uint32_t b = tile;;) {
            const uint4 q0 = table_at(lt.blocks, 2u * b), q1 = table_at(lt.blocks, 2u * b + 1u);
            const uint4 kk = table_at(reinterpret_cast<const uint4*>(lt.block_ids), b);  // (with the block, not after its pre-tests: walk_segment)
            const uint32_t lo[3] = {q0.z, q1.x, q1.z}, hi[3] = {q0.w, q1.y, q1.w}, ids[3] = {kk.x, kk.y, kk.z};
#pragma unroll
            for (uint32_t e = 0; e < 3u; ++e) {
                float4 sphere;
                float key;
                lt_block_entry(lt, lo[e], hi[e], sphere, key);
                if (key > cut) return false;  // sorted by distance from the light: this entry and all after it lie beyond the ray's origin
                if (misses_bounding_sphere2(sphere, ray.sx, ray.sy, ray.sz, ray.dx, ray.dy, ray.dz, dd, w.grid.pretest_alpha)) continue;
                float t;
                bool sphere_type;
                ++tests;
                if (lane_candidate<FUSED, true, true>(hot + ids[e], ray, t, sphere_type) && !(t >= 1.f)) return true;
            }
            if (q0.x == 0u) return false;
            b = q0.x;
        }
    }
    if (chk == chk && light_tile_of(w.ltiles, ray.sx, ray.sy, ray.sz, tile, cut)) {
        const float dd = ray.dx * ray.dx + ray.dy * ray.dy + ray.dz * ray.dz;
        const uint2 range = table_at(w.ltiles.tile_range, tile);
        for (uint32_t e = range.x; e != range.x + range.y; ++e) {
            const float4 bound = table_at(w.ltiles.records, 2u * e), aux = table_at(w.ltiles.records, 2u * e + 1u);
            if (aux.x > cut) break;  // sorted by distance from the light: this entry and all after it lie beyond the ray's origin
            if (misses_bounding_sphere(bound, ray, dd, w.grid.pretest_alpha)) continue;
            float t;
            bool sphere;
            ++tests;
            if (lane_candidate<FUSED, true, true>(hot + __float_as_uint(aux.y), ray, t, sphere) && !(t >= 1.f)) return true;
        }
        return false;
    }
    return nan_shadow_blocked(w.rp.scene, ray);  // no tile in that direction: nothing in the way - unless the ray is a NaN ray
}

// shade_and_reflect, a hit that sends no reflection ray (its path ends here: no bounces left, or absorbed): nothing else of
// the pixel would be in flight next to its shadow ray, and the whole frame would go through one more round - a shadow
// walk, a wf_resume pass - for the last hit of every path. With light tiles the last light's shadow test is a short list:
// it is run HERE, and the light loop's backward scan (shade_last_light_wins) finishes on the spot. False (nothing done,
// nothing counted) when the scan has to go on to an earlier light (stale specular: lit, nDotL <= 0) - the queued path takes over.
template <bool FUSED>
__device__ __forceinline__ bool shade_last_light_inline(Ctx& c, const HitRec& h, bool primary) {
    const Scene& S = c.w.rp.scene;
    const uint32_t li = S.n_lights - 1u;
    float nDotL, rDotV;
    bool lit;
    uint32_t tests = 0;
    {   // the shadow test first, with as little else alive as possible (the step runs at 4 waves per SIMD, 128 registers)
        float nvx = h.nx, nvy = h.ny, nvz = h.nz;
        normalize3_shading(S.fast_phong != 0u, nvx, nvy, nvz);
        float vvx = -h.px, vvy = -h.py, vvz = -h.pz;
        normalize3_shading(S.fast_phong != 0u, vvx, vvy, vvz);
        LightGeom g;
        light_geometry<FUSED>(S.lights[li], h, nvx, nvy, nvz, vvx, vvy, vvz, g, S.fast_phong != 0u);
        nDotL = g.nDotL; rDotV = g.rDotV;
        lit = !last_light_blocked<FUSED>(c.w, g.shadow, tests);
    }
    const LightRec L = S.lights[li];
    const ColdObject* co = S.cold + h.index;
    const float4 amb = co->amb_absorb, dif = co->dif_shine, spec = co->spec_type;
    // (resume_shadow's backward scan at its first light, statement for statement)
    float dr = 0.f, dg = 0.f, db = 0.f, sr = 0.f, sg = 0.f, sb = 0.f;
    const float ar = amb.x * L.ambient.x, ag = amb.y * L.ambient.y, ab = amb.z * L.ambient.z;
    if (lit) {
        const float nd = __builtin_fmaxf(nDotL, 0.f);
        dr = (dif.x * L.diffuse.x) * nd; dg = (dif.y * L.diffuse.y) * nd; db = (dif.z * L.diffuse.z) * nd;
    }
    bool need_specular = true;
    if (!lit) {
        need_specular = false;
    } else if (nDotL > 0) {
        const float pw = specular_power(rDotV, dif.w, S.fast_phong != 0u);
        sr = (spec.x * L.specular.x) * pw; sg = (spec.y * L.specular.y) * pw; sb = (spec.z * L.specular.z) * pw;
        need_specular = false;
    }
    if (need_specular && li > 0u) return false;
    const float cr = (ar + dr) + sr, cg = (ag + dg) + sg, cb = (ab + db) + sb;
    // shade_done() and the top of loop_step(), written out: this function is reached FROM loop_step (via begin_shade_lit),
    // and going back into it would make the call graph recursive (a device stack of unknown depth). The path ends here.
    float abr, abg, abb, rr, rg, rb, ap;
    uint32_t bounces;
    if (primary) {
        ap = amb.w;
        abr = cr * ap; abg = cg * ap; abb = cb * ap;
        rr = 0.f; rg = 0.f; rb = 0.f;
        bounces = c.w.rp.max_bounces;
    } else {
        const float4 acc = c.pre ? c.pre_acc : load_block(c.w, F_ABR, c.i);
        ap = acc.w;
        abr = acc.x; abg = acc.y; abb = acc.z;
        const float ra = (1.f - ap) * amb.w;
        abr = fma_<FUSED>(ra, cr, abr); abg = fma_<FUSED>(ra, cg, abg); abb = fma_<FUSED>(ra, cb, abb);
        ap = ap + ra;
        rr = cr; rg = cg; rb = cb;
        bounces = c.pre ? __float_as_uint(c.pre_res.w) : U(c.w, F_BOUNCES, c.i);
    }
    if (bounces > 0u && ap <= 0.999f) return false;  // the loop would go on (the caller's `sends` says it cannot): not this function's case
    c.traced += 1;
    c.tests += tests;
    if (bounces > 0u) c.reference += 1;  // (loop_step: the iteration whose absorption test ends the loop)
    finish_reflect(c, FUSED, abr, abg, abb, rr, rg, rb, ap, bounces - 1u);
    return true;
}

// Start shading hit `h` when there is at least one light (light loop of shade(), shade_and_reflect_kernel.cl:193):
// queue the first shadow ray. In shade_and_reflect the reflection ray that leaves this hit depends on the hit and
// on the loop state (bounces left, absorption so far), not on the colour the light loop is about to produce - so
// it is queued NOW, next to the shadow ray, with exactly the condition loop_step() will evaluate after the
// shading (`bounces > 0 && absorptionPercent <= 0.999`, the latter already including this hit). A frame then needs
// D + 2 big rounds instead of 2D + 2, every round traces shadow and reflection rays side by side, and wf_resume
// touches the pixel state half as often. `spec_bounces` / `spec_ap`: the loop state as it will be after this hit.
template <int KERNEL, bool FUSED>
__device__ __forceinline__ void begin_shade_lit(Ctx& c, const HitRec& h, bool primary, uint32_t spec_bounces, float spec_ap, const ObjRows& rows) {
    const Scene& S = c.w.rp.scene;
    c.reference += S.n_lights;
    uint32_t flag = 0u;
    // shade_and_reflect outside literal mode never needs the stored reflection vector: either the ray leaves now, or
    // loop_step() will decide - under this same condition - that there is none
    if (!(KERNEL == 2 && !S.literal)) store_hit_extra(c.w, c.i, h);
    const bool sends = KERNEL == 2 && !S.literal && spec_bounces > 0u && spec_ap <= 0.999f;
    if (KERNEL == 2 && RT_INLINE_LAST_SHADOW && !S.literal && !sends && c.w.ltiles.enabled && S.n_lights - 1u == c.w.ltiles.light) {
        if (shade_last_light_inline<FUSED>(c, h, primary)) return;
    }
    if (sends) {
        Ray ray;
        reflection_ray<FUSED>(h, ray);
        if (c.w.grid.enabled) {
            // The ray starts a skin's width off the object it leaves, i.e. inside that object's registration sphere: the
            // grid walk would park it as its first candidate and spend a (sparsely filled) exact-test round on it. Its
            // matrix is in registers here, in a kernel whose lanes all have work: run the reference's exact test against
            // the ray's own object NOW and tell the walk (direction.w of a reflection ray is 0 and no traversal reads the
            // slot) that this object is done. If the test ever reports a hit, nothing is said and the walk tests it as usual.
            float t_self;
            bool sphere_self;
            const bool self_hit = rows_candidate<FUSED, true>(rows, ray, t_self, sphere_self);  // (the record materialise() just read)
            ray.dw = __uint_as_float(self_hit ? 0xffffffffu : (uint32_t)h.index);
        }
        store_ray(c.w, c.i, ray, kSlotClosest);
        c.want_closest = true;
        c.traced += 1;
        flag = PH_FLAG_REFLECTION_SENT | PH_FLAG_REFLECTION_PENDING;
    }
    const bool forward = (KERNEL == 1) || S.literal;
    // the carried light-loop terms start at zero: resume_shadow() knows the first light of a scan and does not
    // read them, so nothing is written here
    emit_shadow<FUSED>(c, h, forward ? 0u : S.n_lights - 1u, (primary ? PH_SHADOW_PRIMARY : PH_SHADOW_REFLECT) | flag, true);
}

// ... or, without lights, go straight on (shade() returns black)
template <int KERNEL, bool FUSED>
__device__ __forceinline__ void begin_shade(Ctx& c, const HitRec& h, bool primary, uint32_t spec_bounces, float spec_ap, const ObjRows& rows) {
    if (c.w.rp.scene.n_lights == 0) { shade_done<KERNEL, FUSED>(c, h, primary, 0.f, 0.f, 0.f); return; }  // (h travels in registers)
    begin_shade_lit<KERNEL, FUSED>(c, h, primary, spec_bounces, spec_ap, rows);
}

// one light-loop iteration, resumed with the visibility of light `li`; mirrors shade_forward /
// shade_last_light_wins in rt_device.h statement for statement
template <int KERNEL, bool FUSED>
__device__ __forceinline__ void resume_shadow(Ctx& c, bool primary) {
    const Scene& S = c.w.rp.scene;
    const uint64_t i = c.i;
    HitRec h;
    {   // point + object, normal (came in with the phase word), and - where it is kept - the reflection vector
        const float4 a = c.pre ? c.pre_hit : load_block(c.w, F_PX, i);
        h.px = a.x; h.py = a.y; h.pz = a.z; h.index = (int)__float_as_uint(a.w);
        h.nx = c.nblock.x; h.ny = c.nblock.y; h.nz = c.nblock.z;
        h.rx = h.ry = h.rz = 0.f;
        h.pw = 1.0f;
        if (!(KERNEL == 2 && !S.literal)) { const float4 r = load_block(c.w, F_RX, i); h.rx = r.x; h.ry = r.y; h.rz = r.z; h.pw = r.w; }
    }
    const uint32_t li = c.li;
    const bool lit = (c.pre ? __float_as_uint(c.pre_res.z) : U(c.w, F_RES_ANY, i)) != 0u;
    const ColdObject* co = S.cold + h.index;
    const float4 amb = co->amb_absorb, dif = co->dif_shine, spec = co->spec_type;
    float nvx = h.nx, nvy = h.ny, nvz = h.nz;
    normalize3_shading(S.fast_phong != 0u, nvx, nvy, nvz);
    float vvx = -h.px, vvy = -h.py, vvz = -h.pz;
    normalize3_shading(S.fast_phong != 0u, vvx, vvy, vvz);
    const LightRec L = S.lights[li];
    LightGeom g;
    light_geometry<FUSED>(L, h, nvx, nvy, nvz, vvx, vvy, vvz, g, S.fast_phong != 0u);
    // (by the time a shadow result is resumed, a reflection ray sent with the hit's first shadow ray has been traced)
    const uint32_t phase = (primary ? PH_SHADOW_PRIMARY : PH_SHADOW_REFLECT) | (c.flags & ~PH_FLAG_REFLECTION_PENDING);
    const bool forward = (KERNEL == 1) || S.literal;
    const bool first_of_scan = forward ? (li == 0u) : (li == S.n_lights - 1u);  // carried terms are still all zero
    float sr = 0.f, sg = 0.f, sb = 0.f;
    if (!first_of_scan) { sr = F(c.w, F_SR, i); sg = F(c.w, F_SG, i); sb = F(c.w, F_SB, i); }
    if (forward) {
        const float ar = amb.x * L.ambient.x, ag = amb.y * L.ambient.y, ab = amb.z * L.ambient.z;
        float dr, dg, db;
        if (lit) {
            const float nd = __builtin_fmaxf(g.nDotL, 0.f);
            dr = (dif.x * L.diffuse.x) * nd; dg = (dif.y * L.diffuse.y) * nd; db = (dif.z * L.diffuse.z) * nd;
            if (g.nDotL > 0) {
                const float pw = specular_power(g.rDotV, dif.w, S.fast_phong != 0u);
                sr = (spec.x * L.specular.x) * pw; sg = (spec.y * L.specular.y) * pw; sb = (spec.z * L.specular.z) * pw;
            }
        } else {
            dr = 0.f; dg = 0.f; db = 0.f;
            sr = 0.f; sg = 0.f; sb = 0.f;
        }
        float cr = 0.f, cg = 0.f, cb = 0.f;
        if (!first_of_scan) { cr = F(c.w, F_CR, i); cg = F(c.w, F_CG, i); cb = F(c.w, F_CB, i); }
        if (KERNEL == 1) { cr = ((cr + ar) + dr) + sr; cg = ((cg + ag) + dg) + sg; cb = ((cb + ab) + db) + sb; }
        else { cr = (ar + dr) + sr; cg = (ag + dg) + sg; cb = (ab + db) + sb; }
        if (li + 1u < S.n_lights) {
            F(c.w, F_SR, i) = sr; F(c.w, F_SG, i) = sg; F(c.w, F_SB, i) = sb;
            F(c.w, F_CR, i) = cr; F(c.w, F_CG, i) = cg; F(c.w, F_CB, i) = cb;
            emit_shadow<FUSED>(c, h, li + 1u, phase, false);
        } else {
            shade_done<KERNEL, FUSED>(c, h, primary, cr, cg, cb);
        }
        return;
    }
    // backward scan (shade_last_light_wins)
    float ar = 0.f, ag = 0.f, ab = 0.f, dr = 0.f, dg = 0.f, db = 0.f;
    if (!first_of_scan) {
        ar = F(c.w, F_AR, i); ag = F(c.w, F_AG, i); ab = F(c.w, F_AB, i);
        dr = F(c.w, F_DR, i); dg = F(c.w, F_DG, i); db = F(c.w, F_DB, i);
    }
    bool need_specular = true;
    if (li == S.n_lights - 1u) {
        ar = amb.x * L.ambient.x; ag = amb.y * L.ambient.y; ab = amb.z * L.ambient.z;
        if (lit) {
            const float nd = __builtin_fmaxf(g.nDotL, 0.f);
            dr = (dif.x * L.diffuse.x) * nd; dg = (dif.y * L.diffuse.y) * nd; db = (dif.z * L.diffuse.z) * nd;
        }
    }
    if (!lit) {
        need_specular = false;
    } else if (g.nDotL > 0) {
        const float pw = specular_power(g.rDotV, dif.w, S.fast_phong != 0u);
        sr = (spec.x * L.specular.x) * pw; sg = (spec.y * L.specular.y) * pw; sb = (spec.z * L.specular.z) * pw;
        need_specular = false;
    }
    if (need_specular && li > 0u) {
        F(c.w, F_AR, i) = ar; F(c.w, F_AG, i) = ag; F(c.w, F_AB, i) = ab;
        F(c.w, F_DR, i) = dr; F(c.w, F_DG, i) = dg; F(c.w, F_DB, i) = db;
        F(c.w, F_SR, i) = sr; F(c.w, F_SG, i) = sg; F(c.w, F_SB, i) = sb;
        emit_shadow<FUSED>(c, h, li - 1u, phase, false);
    } else {
        shade_done<KERNEL, FUSED>(c, h, primary, (ar + dr) + sr, (ag + dg) + sg, (ab + db) + sb);
    }
}

// the tail of shade_and_reflect (:281-284)
__device__ __forceinline__ void finish_reflect(Ctx& c, bool fused, float abr, float abg, float abb, float rr, float rg,
                                               float rb, float ap, uint32_t bounces) {
    if (bounces == 0u && ap < 1.f) {
        const float wgt = 1.f - ap;
        if (fused) { abr = fma_<true>(wgt, rr, abr); abg = fma_<true>(wgt, rg, abg); abb = fma_<true>(wgt, rb, abb); }
        else { abr = fma_<false>(wgt, rr, abr); abg = fma_<false>(wgt, rg, abg); abb = fma_<false>(wgt, rb, abb); }
    }
    write_pixel<2>(c, abr, abg, abb);
}

// the state a pixel carries across a reflection trace
__device__ __forceinline__ void store_loop_state(Ctx& c, float abr, float abg, float abb, float rr, float rg, float rb, float ap,
                                                 uint32_t bounces, bool keep_reflect_color) {
    const uint64_t i = c.i;
    store_block(c.w, F_ABR, i, make_float4(abr, abg, abb, ap));
    U(c.w, F_BOUNCES, i) = bounces;
    // reflectColor only has to survive a trace when the NEXT resume step may end the loop without shading another hit
    // (PH_REFLECT); a ray that left with its hit is followed by a shading step that replaces it
    if (keep_reflect_color) store_block(c.w, F_RR, i, make_float4(rr, rg, rb, 0.f));
}

// top of one iteration of `while (bounces-- > 0 && raycast(...) && absorptionPercent <= 0.999f)` (:268)
template <bool FUSED>
__device__ __forceinline__ void loop_step(Ctx& c, const HitRec& from, float abr, float abg, float abb, float rr, float rg,
                                          float rb, float ap, uint32_t bounces) {
    const Scene& S = c.w.rp.scene;
    const uint32_t before = bounces;
    bounces = bounces - 1u;
    if (!(before > 0u)) { finish_reflect(c, FUSED, abr, abg, abb, rr, rg, rb, ap, bounces); return; }
    c.reference += 1;
    const bool absorbing = (ap <= 0.999f);
    if (!absorbing && !S.literal) { finish_reflect(c, FUSED, abr, abg, abb, rr, rg, rb, ap, bounces); return; }
    if (c.flags & PH_FLAG_REFLECTION_SENT) {
        // the ray left together with this hit's first shadow ray (begin_shade_lit) and has been traced: carry on
        // with its result, as the PH_REFLECT branch of wf_resume does one round later in the other modes
        const uint64_t i = c.i;
        float T;
        int idx;
        Ray ray;
        if (c.pre) {
            T = c.pre_res.x;
            idx = (int)__float_as_uint(c.pre_res.y);
            ray = c.pre_ray0;
            patch_nan_result(c.w, ray, T, idx);
        } else {
            closest_result(c.w, i, false, T, idx);
        }
        if (T == kMaxFloat) { finish_reflect(c, FUSED, abr, abg, abb, rr, rg, rb, ap, bounces); return; }
        if (!c.pre) ray = load_ray(c.w, i, kSlotClosest);
        ray.dw = 0.0f;  // (a reflection ray's direction.w; the slot may carry begin_shade_lit's note to the walk)
        HitRec rh;
        ObjRows rows;
        float absorb_rh;
        materialise<FUSED>(S.objrec, S.cold, idx, T, ray, rh, S.affine != 0u, &rows, &absorb_rh);
        store_loop_state(c, abr, abg, abb, rr, rg, rb, ap, bounces, false);
        if (c.pre) {  // the step may go on to shade rh on the spot (shade_last_light_inline): what it preloaded is now this
            c.pre_acc = make_float4(abr, abg, abb, ap);
            c.pre_res.w = __uint_as_float(bounces);
        }
        const float ra = (1.f - ap) * absorb_rh;  // shade_done's update, ahead of time (the absorption came with the record)
        begin_shade_lit<2, FUSED>(c, rh, false, bounces, ap + ra, rows);
        return;
    }
    Ray ray;
    reflection_ray<FUSED>(from, ray);
    ray.dw = __uint_as_float(0xffffffffu);  // (no note for the walk - see begin_shade_lit; no traversal reads direction.w of a reflection ray)
    store_ray(c.w, c.i, ray, kSlotClosest);
    store_loop_state(c, abr, abg, abb, rr, rg, rb, ap, bounces, true);
    U(c.w, F_PHASE, c.i) = PH_REFLECT;
    c.want_closest = true;
    c.traced += 1;
}

template <int KERNEL, bool FUSED>
__device__ __forceinline__ void shade_done(Ctx& c, const HitRec& h, bool primary, float cr, float cg, float cb) {
    const Scene& S = c.w.rp.scene;
    if (KERNEL == 1) { write_pixel<1>(c, cr, cg, cb); return; }
    const uint64_t i = c.i;
    if (primary) {
        // absorbColor = hit.mat.absorption * shade(hit) (:255-256); bounces = MAX_BOUNCES (:258)
        const float ap = S.cold[h.index].amb_absorb.w;
        loop_step<FUSED>(c, h, cr * ap, cg * ap, cb * ap, 0.f, 0.f, 0.f, ap, c.w.rp.max_bounces);
    } else {
        const float4 acc = c.pre ? c.pre_acc : load_block(c.w, F_ABR, i);
        float ap = acc.w;
        float abr = acc.x, abg = acc.y, abb = acc.z;
        const float ra = (1.f - ap) * S.cold[h.index].amb_absorb.w;
        abr = fma_<FUSED>(ra, cr, abr); abg = fma_<FUSED>(ra, cg, abg); abb = fma_<FUSED>(ra, cb, abb);
        ap = ap + ra;
        loop_step<FUSED>(c, h, abr, abg, abb, cr, cg, cb, ap, c.pre ? __float_as_uint(c.pre_res.w) : U(c.w, F_BOUNCES, i));
    }
}

// one step of a pixel's state machine: consume the results of the ray(s) it had in flight, queue what it needs next
template <int KERNEL, bool FUSED>
__device__ __forceinline__ void resume_pixel(Ctx& c) {
    const WfParams& w = c.w;
    const uint64_t i = c.i;
    const Scene& S = w.rp.scene;
    c.pre = false;
    c.nblock = w.identity_queue ? make_float4(0.f, 0.f, 0.f, __uint_as_float(PH_PRIMARY)) : load_block(w, F_NX, i);
    const uint32_t word = __float_as_uint(c.nblock.w);
    const uint32_t phase = word & 0xffu;
    c.flags = word & (PH_FLAG_REFLECTION_SENT | PH_FLAG_REFLECTION_PENDING);
    c.li = word >> kPhaseLightShift;
    if (phase == PH_PRIMARY) {
        float T;
        int idx;
        closest_result(w, i, true, T, idx);
        const bool hit = (KERNEL == 2) ? !(T == kMaxFloat) : (T < kMaxFloat);
        c.traced += 1; c.reference += 1; c.hits += hit ? 1 : 0;
        if (w.rp.aux_t) w.rp.aux_t[pixel_of(w.rp, i)] = T;
        if (w.rp.aux_index) w.rp.aux_index[pixel_of(w.rp, i)] = hit ? idx : -1;
        if (KERNEL == 0) {
            reinterpret_cast<float*>(w.rp.out)[pixel_of(w.rp, i)] = hit ? T : kMaxFloat;
            U(w, F_PHASE, i) = PH_DONE;
        } else if (!hit) {
            write_pixel<KERNEL>(c, 0.f, 0.f, 0.f);
        } else {
            const Ray ray = closest_ray(w, i, true);
            HitRec h;
            ObjRows rows;
            float absorb_h;
            materialise<FUSED>(S.objrec, S.cold, idx, T, ray, h, S.affine != 0u, &rows, &absorb_h);
            begin_shade<KERNEL, FUSED>(c, h, true, w.rp.max_bounces, absorb_h, rows);
        }
    } else if (phase == PH_SHADOW_PRIMARY || phase == PH_SHADOW_REFLECT) {
        c.pre = true;
        c.pre_hit = load_block(w, F_PX, i);
        c.pre_res = load_block(w, F_RES_T, i);
        c.pre_acc = load_block(w, F_ABR, i);       // (not yet written in the primary phase: not read there either)
        c.pre_ray0 = load_ray(w, i, kSlotClosest);  // (only meaningful when a reflection ray left with the hit)
        resume_shadow<KERNEL, FUSED>(c, phase == PH_SHADOW_PRIMARY);
    } else if (phase == PH_REFLECT) {
        float T;
        int idx;
        closest_result(w, i, false, T, idx);
        const float4 acc = load_block(w, F_ABR, i), rc = load_block(w, F_RR, i);
        const float ap = acc.w;
        const uint32_t bounces = U(w, F_BOUNCES, i);
        const float abr = acc.x, abg = acc.y, abb = acc.z;
        const float rr = rc.x, rg = rc.y, rb = rc.z;
        if (T == kMaxFloat || !(ap <= 0.999f)) {  // raycast() false, or the absorption test of the loop condition
            finish_reflect(c, FUSED, abr, abg, abb, rr, rg, rb, ap, bounces);
        } else {
            Ray ray = load_ray(w, i, kSlotClosest);
            ray.dw = 0.0f;  // (a reflection ray's direction.w; the slot may carry begin_shade_lit's note to the walk)
            HitRec rh;
            ObjRows rows;
            float absorb_rh;
            materialise<FUSED>(S.objrec, S.cold, idx, T, ray, rh, S.affine != 0u, &rows, &absorb_rh);
            begin_shade<KERNEL, FUSED>(c, rh, false, bounces, ap + (1.f - ap) * absorb_rh, rows);
        }
    }
}

// a pixel with a shadow AND a reflection ray in flight sits in both queues: it is resumed from its shadow entry
__device__ __forceinline__ bool duplicate_entry(const WfParams& w, uint32_t t, uint32_t entry) {
    return (t < w.n_prev_closest) && (entry & kQueueAlsoShadow) != 0u;
}

__device__ __forceinline__ void add_ray_counters(const WfParams& w, const Ctx& c) {
    const unsigned long long a = wave_sum64(c.traced), b = wave_sum64(c.reference), h = wave_sum64(c.hits), t = wave_sum64(c.tests);
    if ((threadIdx.x & 63u) == 0u && (a | b | h | t)) {
        atomicAdd(&w.rp.counters->traced, a);
        atomicAdd(&w.rp.counters->reference, b);
        atomicAdd(&w.rp.counters->hits, h);
        if (t) atomicAdd(&w.rp.counters->tests, t);
    }
}

#ifndef RT_RESUME_WAVES_PER_EU
#define RT_RESUME_WAVES_PER_EU 4
#endif
template <int KERNEL, bool FUSED>
__global__ __launch_bounds__(kResumeThreads) __attribute__((amdgpu_waves_per_eu(RT_RESUME_WAVES_PER_EU))) void wf_resume(const WfParams wk) {
    WfParams w = wk;
    if (!resolve_round(w)) return;
    if (blockIdx.x * kResumeThreads >= w.n_prev_closest + w.n_prev_any) return;  // the grid is sized for the most the queues can hold
    const uint32_t t = blockIdx.x * kResumeThreads + threadIdx.x;
    const uint32_t total = w.n_prev_closest + w.n_prev_any;
    Ctx c{w, 0, 0ull, 0ull, 0ull, false, false, 0u, 0u, make_float4(0.f, 0.f, 0.f, 0.f), false, {}, {}, {}, {}};
    if (t < total) {
        const uint32_t entry = w.identity_queue ? t : ((t < w.n_prev_closest) ? w.q_prev_closest[t] : w.q_prev_any[t - w.n_prev_closest]);
        c.i = entry & kQueuePixel;
        if (!duplicate_entry(w, t, entry)) resume_pixel<KERNEL, FUSED>(c);
    }
    block_push(c.want_closest, c.want_any, (uint32_t)c.i, w.q_closest, w.q_any, w.counts, c.any_flag);
    if (w.count_rays) add_ray_counters(w, c);
}

// The tail of a frame: once only a sliver of the pixels is still alive (stale-specular light scans, the odd long
// path), a round costs its fixed price - launch, ~0.1 ms of latency per ray, a host round trip - for almost no work,
// and that price does not shrink when the frame is split over more GPUs. Here every remaining pixel runs its state
// machine to the end in one thread, tracing its rays itself through the grid (closest_hit_grid / any_hit_grid: same
// cells, same exact tests, same results as the wave-level walk).
template <int KERNEL, bool FUSED>
__global__ __launch_bounds__(256) void wf_finish(const WfParams wk) {
    WfParams w = wk;
    (void)resolve_round(w);
    if (w.counts[RS_FINISH] != 1u) return;  // only once wf_advance has handed the rest of the frame over
    const uint32_t t = blockIdx.x * 256u + threadIdx.x;
    const uint32_t total = w.n_prev_closest + w.n_prev_any;
    Ctx c{w, 0, 0ull, 0ull, 0ull, false, false, 0u, 0u, make_float4(0.f, 0.f, 0.f, 0.f), false, {}, {}, {}, {}};
    uint32_t tested = 0;
    if (t < total) {
        const uint32_t entry = (t < w.n_prev_closest) ? w.q_prev_closest[t] : w.q_prev_any[t - w.n_prev_closest];
        c.i = entry & kQueuePixel;
        if (!duplicate_entry(w, t, entry)) {
            // the queues hold rays that have NOT been traced yet: trace first, then resume, and so on to the end
            bool do_closest = (t < w.n_prev_closest) || (U(w, F_PHASE, c.i) & PH_FLAG_REFLECTION_PENDING) != 0u;
            bool do_any = !(t < w.n_prev_closest);
            for (;;) {
                if (do_any) {
                    uint32_t li;
                    Ray ray;
                    if (shadow_rays_rebuilt(w)) {
                        ray = shadow_of_pixel<FUSED>(w, c.i, false, li);
                    } else {
                        ray = load_ray(w, c.i, kSlotShadow);
                        li = __float_as_uint(ray.dw);
                    }
                    bool blocked;
                    if (w.ltiles.enabled && li == w.ltiles.light) {
                        blocked = last_light_blocked<FUSED>(w, ray, tested);   // (the walk's choice: trace_segment)
                    } else {
                        bool walked = false;
                        blocked = false;
                        if (w.bgrid.enabled && w.grid.n_always == 0u) {
                            Ray r1 = ray;
                            r1.sw = 1.0f; r1.dw = 0.0f;
                            blocked = any_hit_blocks<FUSED>(w.bgrid, w.grid, w.rp.scene.hot, r1, walked, tested);
                        }
                        if (!walked) blocked = any_hit_grid<FUSED>(w.grid, w.rp.scene, ray, tested);
                    }
                    U(w, F_RES_ANY, c.i) = blocked ? 0u : 1u;
                }
                if (do_closest) {
                    Ray ray = load_ray(w, c.i, kSlotClosest);
                    float T = kMaxFloat;
                    int idx = -1;
                    bool walked = false;
                    if (w.bgrid.enabled && w.grid.n_always == 0u) {  // the block grid (the ray's note names the object it leaves)
                        const uint32_t note = __float_as_uint(ray.dw);
                        ray.sw = 1.0f; ray.dw = 0.0f;
                        walked = closest_hit_blocks<FUSED>(w.bgrid, w.grid, w.rp.scene.hot, ray, note, T, idx, tested);
                    }
                    if (!walked) closest_hit_grid<FUSED, true>(w.grid, w.rp.scene.hot, ray, T, idx, tested);
                    store_closest_result(w, c.i, T, idx);
                }
                c.want_closest = false;
                c.want_any = false;
                resume_pixel<KERNEL, FUSED>(c);
                if (!c.want_closest && !c.want_any) break;  // the pixel has been written
                do_closest = c.want_closest;
                do_any = c.want_any;
            }
        }
    }
    if (w.count_rays) {
        add_ray_counters(w, c);
        if (tested) atomicAdd(&w.rp.counters->tests, (unsigned long long)tested);
    }
}

// ---- the frame in ONE persistent kernel (round 4) ------------------------------------------------------------------------
// The reference's work-item runs primary ray -> shade -> up to D reflection iterations without ever waiting for another
// pixel (shade_and_reflect_kernel.cl:244-285). The round machine above cuts that loop at every ray and puts a device-wide
// barrier there: 5 rounds x {two walks, wf_resume, wf_advance} = 21 launches per cfg4 frame, every persistent walk draining
// from 54 to 0 live lanes before the next kernel may start (~0.15 ms per launch that no split of the frame over more GPUs
// removes), and a memory-bound wf_resume that runs alone between issue-bound walks. Here a pixel goes from
// its primary hit to its final store inside ONE persistent launch, without ever meeting a barrier:
//   * the waves of a workgroup take roles (below: "roles"): WALKERS trace closest-hit rays - block_segment's loop, trip for
//     trip (same blocks, same pre-tests, same parked exact tests, same order-free update), so (t, index) of every ray are the
//     round machine's bits - and a STEPPER runs the pixels' steps, a full wave of them at a time (frame_step: materialise the
//     hit, build the reflection ray and test it against the object it leaves, test the light loop's shadow rays on the spot -
//     the last light's through its light tile, the list wf_resume's last hits and wf_finish already walk in one thread - shade,
//     accumulate, write the pixel or queue its reflection ray). The step calls the round machine's own device functions, so
//     every colour is the round machine's too;
//   * rays and pixel state travel between the two through rings in LDS: no queue and no pixel state in HBM, no second ray
//     kind in flight (the shadow rays never leave the step);
//   * primary rays keep their own first kernel (screen tiles: wave-uniform lists), whose (t, index) per work-item the stepper
//     reads when it admits a pixel.
// Used for shade_and_reflect frames through grid + block grid + block-form light tiles without triangles or always-tested
// objects (launch: frame_kernel_applies); everything else keeps the rounds.
#ifndef RT_FRAME_WAVES
#define RT_FRAME_WAVES 4          // waves per SIMD the kernel is compiled for. MEASURED (the block walk alone, occupancy forced by unused LDS,
                                  // cfg4 frame): 6 waves 11.80 ms, 5: 12.23, 4: 12.24, 3: 13.47 - four cost the walk 7 %, three 26 %
#endif
#ifndef RT_FRAME_REFILL_MIN
#define RT_FRAME_REFILL_MIN 16    // a walker parks / takes rays once this many of its lanes are not walking (a ray's set-up is ~150 instructions)
#endif
struct FrameAcc {
    float abr, abg, abb, ap;      // absorbColor, absorptionPercent (shade_and_reflect_kernel.cl:255-257)
    float rr, rg, rb;             // reflectColor
    uint32_t bounces;             // as the loop's unsigned counter AFTER its post-decrement
};

// shade()'s light loop for hit h, shade_and_reflect's way (shade_last_light_wins, rt_device.h, statement for statement; the
// round machine spreads the same loop over resume_shadow's visits): ambient and diffuse are the LAST light's, the specular
// belongs to the last light that was either blocked or lit with nDotL > 0 - scan backwards and stop there. The last light's
// shadow ray goes through its light tile (last_light_blocked), an earlier light's - the stale-specular case, a per cent of
// the hits - through the fine grid in one thread (any_hit_grid, as wf_finish traces it where the block walk does not apply).
template <bool FUSED, bool BLOCKS>
__device__ __forceinline__ void shade_scan_now(const WfParams& w, const HitRec& h, float& cr, float& cg, float& cb, float& absorb, uint32_t& tests,
                                               uint32_t& shadow_rays) {
    const Scene& S = w.rp.scene;
    const ColdObject* co = S.cold + h.index;
    float nvx = h.nx, nvy = h.ny, nvz = h.nz;
    normalize3_shading(S.fast_phong != 0u, nvx, nvy, nvz);
    float vvx = -h.px, vvy = -h.py, vvz = -h.pz;
    normalize3_shading(S.fast_phong != 0u, vvx, vvy, vvz);
    float sr = 0.f, sg = 0.f, sb = 0.f;
    float dr = 0.f, dg = 0.f, db = 0.f;
    float ar = 0.f, ag = 0.f, ab = 0.f;
    bool need_specular = true;
    for (uint32_t li = S.n_lights; li-- > 0 && need_specular;) {
        const LightRec L = S.lights[li];
        LightGeom g;
        light_geometry<FUSED>(L, h, nvx, nvy, nvz, vvx, vvy, vvz, g, S.fast_phong != 0u);
        shadow_rays += 1u;
        bool blocked;
        if (li == w.ltiles.light) {
            blocked = last_light_blocked<FUSED>(w, g.shadow, tests);
        } else {
            // a light without tiles: one thread through the block grid, or - rays that walk is not made for, and always in wf_frame's
            // stepper (BLOCKS = false), where the second walk inlined cost every step 56 spilled registers, as a real call 151 -
            // through the fine grid: wf_finish's pair of walks, same answer either way
            bool walked = false;
            blocked = false;
            if (BLOCKS) {
                Ray r1 = g.shadow;
                r1.sw = 1.0f; r1.dw = 0.0f;
                blocked = any_hit_blocks<FUSED>(w.bgrid, w.grid, S.hot, r1, walked, tests);
            }
            if (!walked) blocked = any_hit_grid<FUSED>(w.grid, S, g.shadow, tests);
        }
        const bool lit = !blocked;
        if (li == S.n_lights - 1u) {
            const float4 amb = co->amb_absorb, dif = co->dif_shine;
            ar = amb.x * L.ambient.x; ag = amb.y * L.ambient.y; ab = amb.z * L.ambient.z;
            if (lit) {
                const float nd = __builtin_fmaxf(g.nDotL, 0.f);
                dr = (dif.x * L.diffuse.x) * nd; dg = (dif.y * L.diffuse.y) * nd; db = (dif.z * L.diffuse.z) * nd;
            }
        }
        if (!lit) {
            need_specular = false;  // zeroed here, nothing later re-assigns it
        } else if (g.nDotL > 0) {
            const float4 dif = co->dif_shine, spec = co->spec_type;
            const float pw = specular_power(g.rDotV, dif.w, S.fast_phong != 0u);
            sr = (spec.x * L.specular.x) * pw; sg = (spec.y * L.specular.y) * pw; sb = (spec.z * L.specular.z) * pw;
            need_specular = false;
        }
        // lit with nDotL <= 0: the specular of an earlier light is still live - keep scanning
    }
    cr = (ar + dr) + sr; cg = (ag + dg) + sg; cb = (ab + db) + sb;
    absorb = co->amb_absorb.w;
}

// A wave's two rings, in LDS (nobody but the wave touches them: no barrier, no atomics - the counters are wave-uniform
// registers). A ray whose walk has ended is PARKED in `due` by its lane, which then takes the next ray waiting in `rdy`;
// once enough pixels are due, the WHOLE wave runs their steps - one pixel per lane, every lane busy - and what goes on is
// queued in `rdy`. Round 4's first form let a lane step its own pixel at hand-out time: ~800 instructions at 10-16 of 64
// lanes, twice per hand-out - 24.3 ms per cfg4 frame against the round machine's 11.9.
#ifndef RT_FRAME_RING
#define RT_FRAME_RING 96   // entries per ring: 132 bytes x entries x walkers per workgroup x workgroups per CU <= 160 KB
#endif
constexpr uint32_t kRing = RT_FRAME_RING;
static_assert(kRing >= 64, "a full wave of steps");
// Pixels a wave has in flight (walking + due + waiting). With at most 64 + kRing of them no state exists in which nothing can
// move: all 64 lanes holding a finished ray they cannot park (due full) AND rays waiting in rdy would be 65 + kRing pixels.
constexpr uint32_t kPixelsInFlight = 64u + kRing;
struct FrameRings {
    float4 due[4][kRing];      // {start, T}, {direction, index}, {absorbColor, absorptionPercent}, {reflectColor, bounces}
    uint32_t due_pix[kRing];
    float4 rdy[4][kRing];      // {start, note}, {direction, pixel}, accumulators as above
};
static_assert(sizeof(FrameRings) == 132u * kRing, "layout");

// Where a step finds its pixel and leaves what goes on: wf_frame's LDS rings (RingIO) or the round machine's pixel state in
// HBM (StateIO, wf_step). traced(): the ray that came back, its result and the work-item; ap_bounces(): absorptionPercent and
// bounces before this hit; acc(): all accumulators; put_ray() / put_acc(): the reflection ray (with the note for the walk) and
// the accumulators of a pixel that goes on.
struct RingIO {
    FrameRings& R;
    uint32_t dslot, pslot;
    __device__ __forceinline__ void traced(Ray& ray, float& T, int& idx, uint32_t& pix) const {
        const float4 e0 = R.due[0][dslot], e1 = R.due[1][dslot];
        pix = R.due_pix[dslot];
        ray.sx = e0.x; ray.sy = e0.y; ray.sz = e0.z; T = e0.w;
        ray.dx = e1.x; ray.dy = e1.y; ray.dz = e1.z; idx = (int)__float_as_uint(e1.w);
    }
    __device__ __forceinline__ void ap_bounces(float& ap, uint32_t& bounces) const { ap = R.due[2][dslot].w; bounces = __float_as_uint(R.due[3][dslot].w); }
    __device__ __forceinline__ void acc(FrameAcc& a) const {
        const float4 e2 = R.due[2][dslot], e3 = R.due[3][dslot];
        a.abr = e2.x; a.abg = e2.y; a.abb = e2.z; a.ap = e2.w;
        a.rr = e3.x; a.rg = e3.y; a.rb = e3.z; a.bounces = __float_as_uint(e3.w);
    }
    __device__ __forceinline__ void put_ray(const Ray& nr, uint32_t note, uint32_t pix) const {
        R.rdy[0][pslot] = make_float4(nr.sx, nr.sy, nr.sz, __uint_as_float(note));
        R.rdy[1][pslot] = make_float4(nr.dx, nr.dy, nr.dz, __uint_as_float(pix));
    }
    __device__ __forceinline__ void put_acc(const FrameAcc& a) const {
        R.rdy[2][pslot] = make_float4(a.abr, a.abg, a.abb, a.ap);
        R.rdy[3][pslot] = make_float4(a.rr, a.rg, a.rb, __uint_as_float(a.bounces));
    }
};
struct StateIO {   // ray slot 0, F_RES_T / F_RES_I, the F_ABR block, the F_RR block with the bounces in its spare word
    const WfParams& w;
    uint32_t pix0;
    __device__ __forceinline__ void traced(Ray& ray, float& T, int& idx, uint32_t& pix) const {
        pix = pix0;
        load_closest_result(w, pix0, T, idx);
        const Ray r = load_ray(w, pix0, kSlotClosest);
        ray.sx = r.sx; ray.sy = r.sy; ray.sz = r.sz; ray.dx = r.dx; ray.dy = r.dy; ray.dz = r.dz;  // (w = 1 / 0: the slot carried the walk's note)
    }
    __device__ __forceinline__ void ap_bounces(float& ap, uint32_t& bounces) const { ap = F(w, F_AP, pix0); bounces = U(w, F_SPARE, pix0); }
    __device__ __forceinline__ void acc(FrameAcc& a) const {
        const float4 e2 = load_block(w, F_ABR, pix0), e3 = load_block(w, F_RR, pix0);
        a.abr = e2.x; a.abg = e2.y; a.abb = e2.z; a.ap = e2.w;
        a.rr = e3.x; a.rg = e3.y; a.rb = e3.z; a.bounces = __float_as_uint(e3.w);
    }
    __device__ __forceinline__ void put_ray(const Ray& nr, uint32_t note, uint32_t) const {
        Ray r = nr;
        r.sw = 1.0f; r.dw = __uint_as_float(note);
        store_ray(w, pix0, r, kSlotClosest);
    }
    __device__ __forceinline__ void put_acc(const FrameAcc& a) const {
        store_block(w, F_ABR, pix0, make_float4(a.abr, a.abg, a.abb, a.ap));
        store_block(w, F_RR, pix0, make_float4(a.rr, a.rg, a.rb, __uint_as_float(a.bounces)));
    }
};

// One step of a pixel = shade_and_reflect's loop body (:253-284) for a pixel whose closest-hit ray came back with (T, idx):
// either a ray `io` knows, or - a NEW pixel, `primary` - work-item `pix` with the first kernel's result. True: the pixel goes
// on; its reflection ray, the note for the walk (the object it leaves, already tested, or ~0) and its accumulators have been
// put down through `io`. False: the pixel has been written.
// In wf_frame the step runs next to walks on a 128-register budget, and what is kept in registers across its heavy part (the
// shadow test) decides how much goes to scratch (a first ring version spilled ~70 registers per lane and step: 18 GB of scratch
// writes per frame). So `io` doubles as the step's memory: the accumulators stay where they are until the colour is known, and
// the reflection ray is built and tested against its own object right after materialise() - while the object's rows are in
// registers - and put down before the shading starts (whether the loop will cast it depends on the loop state, not on the
// colour: begin_shade_lit's `sends`).
template <bool FUSED, bool BLOCKS, typename IO>
__device__ __forceinline__ bool frame_step(const WfParams& w, const IO& io, bool primary, uint32_t pix, Ctx& cnt) {
    const RenderParams& p = w.rp;
    const Scene& S = p.scene;
    float4* out = reinterpret_cast<float4*>(p.out);
    Ray ray = {0.f, 0.f, 0.f, 1.0f, 0.f, 0.f, 0.f, 0.0f};  // (what every ray of a grid-able frame carries: rt_create checks)
    float T;
    int idx;
    if (!primary) {
        io.traced(ray, T, idx, pix);
    } else {
        load_closest_result(w, pix, T, idx);   // the first kernel's (t, index) of this work-item's primary ray
        const Ray pr = closest_ray(w, pix, true);
        ray.sx = pr.sx; ray.sy = pr.sy; ray.sz = pr.sz; ray.dx = pr.dx; ray.dy = pr.dy; ray.dz = pr.dz;
    }
    if (T == kMaxFloat) patch_nan_result(w, ray, T, idx);
    const bool hit = !(T == kMaxFloat);
    const uint64_t px = pixel_of(p, pix);
    auto finish = [&](const FrameAcc& a) {  // the tail of shade_and_reflect (:281-284): finish_reflect without the phase word
        float abr = a.abr, abg = a.abg, abb = a.abb;
        if (a.bounces == 0u && a.ap < 1.f) {
            const float wgt = 1.f - a.ap;
            abr = fma_<FUSED>(wgt, a.rr, abr); abg = fma_<FUSED>(wgt, a.rg, abg); abb = fma_<FUSED>(wgt, a.rb, abb);
        }
        out[px] = make_float4(abr, abg, abb, 1.0f);
    };
    if (primary) {
        if (p.aux_t) p.aux_t[px] = T;
        if (p.aux_index) p.aux_index[px] = hit ? idx : -1;
        cnt.traced += 1; cnt.reference += 1; cnt.hits += hit ? 1 : 0;
        if (!hit) { out[px] = make_float4(0.f, 0.f, 0.f, 1.0f); return false; }
    } else if (!hit) {  // raycast() false (:268)
        FrameAcc a;
        io.acc(a);
        finish(a);
        return false;
    }
    float hpx, hpy, hpz, hnx, hny, hnz, absorb_h;
    int hindex;
    bool sends;
    {
        HitRec h;
        ObjRows rows;
        materialise<FUSED>(S.objrec, S.cold, idx, T, ray, h, S.affine != 0u, &rows, &absorb_h);
        hpx = h.px; hpy = h.py; hpz = h.pz; hnx = h.nx; hny = h.ny; hnz = h.nz; hindex = h.index;
        // Will the loop cast this hit's reflection ray? (the statements below are shade_done's and loop_step's, ahead of time)
        uint32_t bounces_in = p.max_bounces;
        float ap_after = absorb_h;
        if (!primary) {
            float ap_old;
            io.ap_bounces(ap_old, bounces_in);
            const float ra = (1.f - ap_old) * absorb_h;
            ap_after = ap_old + ra;
        }
        sends = bounces_in > 0u && ap_after <= 0.999f;
        if (sends) {
            Ray nr;
            reflection_ray<FUSED>(h, nr);
            float t_self;
            bool sphere_self;
            const bool self_hit = rows_candidate<FUSED, true>(rows, nr, t_self, sphere_self);  // begin_shade_lit: the ray's own object, tested here
            io.put_ray(nr, self_hit ? 0xffffffffu : (uint32_t)h.index, pix);
        }
    }
    float cr, cg, cb, absorb;
    uint32_t tests = 0, shadow_rays = 0;
    {
        HitRec h;
        h.px = hpx; h.py = hpy; h.pz = hpz; h.pw = 1.0f; h.nx = hnx; h.ny = hny; h.nz = hnz; h.rx = 0.f; h.ry = 0.f; h.rz = 0.f; h.index = hindex;
        shade_scan_now<FUSED, BLOCKS>(w, h, cr, cg, cb, absorb, tests, shadow_rays);
    }
    cnt.traced += shadow_rays;     // the shadow rays just tested
    cnt.reference += S.n_lights;   // the light loop's rays (begin_shade_lit)
    cnt.tests += tests;
    FrameAcc a;
    if (primary) {  // absorbColor = hit.mat.absorption * shade(hit) (:255-256); bounces = MAX_BOUNCES (:258)
        a.ap = absorb;
        a.abr = cr * a.ap; a.abg = cg * a.ap; a.abb = cb * a.ap;
        a.rr = 0.f; a.rg = 0.f; a.rb = 0.f;
        a.bounces = p.max_bounces;
    } else {        // (:270-274)
        io.acc(a);
        const float ra = (1.f - a.ap) * absorb;
        a.abr = fma_<FUSED>(ra, cr, a.abr); a.abg = fma_<FUSED>(ra, cg, a.abg); a.abb = fma_<FUSED>(ra, cb, a.abb);
        a.ap = a.ap + ra;
        a.rr = cr; a.rg = cg; a.rb = cb;
    }
    // top of `while (bounces-- > 0 && raycast(...) && absorptionPercent <= 0.999f)` (:268), as loop_step
    const uint32_t before = a.bounces;
    a.bounces = before - 1u;
    if (!(before > 0u)) { finish(a); return false; }
    cnt.reference += 1;
    if (!(a.ap <= 0.999f)) { finish(a); return false; }  // the reference casts this ray but never reads its result
    // (`sends` said so - same inputs, same arithmetic - and the ray has been put down already)
    io.put_acc(a);
    cnt.traced += 1;
    return sends;
}

// ---- roles ----
// A step needs ~125 registers (it is wf_resume's arithmetic), a walking lane ~45 that stay live across it: a wave that did both
// had 189 to hold in 128 and spilled 70-120 of them per lane and step (18 GB of scratch writes per frame, round 4's second
// form: 15.8 ms). So the waves of a workgroup take ROLES: wave 0 is the STEPPER, waves 1-3 are WALKERS. Each walker owns a
// `due` and a `rdy` ring in LDS (FrameRings, single producer / single consumer each: the walker parks finished rays in `due`
// and takes rays from `rdy`, the stepper does the opposite), four counters say how far either side has got (FrameCtl; written
// with release, read with acquire at workgroup scope - LDS is one in-order memory per CU, the fences are for the compiler).
// A pixel stays with its walker from admission to the final store; the stepper admits new pixels (it draws the runs) while a
// walker has fewer than kPixelsInFlight of them. max(walk, step) registers instead of their sum; nothing ever waits on a
// device-wide barrier, and nothing but the framebuffer, the first kernel's results and the scene tables touches memory.
struct FrameCtl {
    uint32_t due_tail;   // walker: rays parked in `due` so far          (slot = count mod kRing; all counts only grow)
    uint32_t due_head;   // stepper: ... of which it has consumed
    uint32_t rdy_tail;   // stepper: rays queued in `rdy` so far
    uint32_t rdy_head;   // walker: ... of which it has taken
    uint32_t starving;   // walker: nothing walks and nothing waits (the stepper then steps whatever is due, however little)
    uint32_t quit;       // stepper: this walker's part of the frame is finished
    uint32_t pad[2];
};
constexpr uint32_t kFrameWalkers = 3;   // walker waves per workgroup (+ one stepper)
static_assert(132u * kRing * kFrameWalkers + (sizeof(FrameCtl) + 32u) * kFrameWalkers <= 40u * 1024u, "four workgroups per CU");

__device__ __forceinline__ uint32_t ctl_load(const uint32_t* p) { return __hip_atomic_load(p, __ATOMIC_ACQUIRE, __HIP_MEMORY_SCOPE_WORKGROUP); }
__device__ __forceinline__ void ctl_store(uint32_t* p, uint32_t v) { __hip_atomic_store(p, v, __ATOMIC_RELEASE, __HIP_MEMORY_SCOPE_WORKGROUP); }
constexpr uint32_t kFrameSpinLimit = 1u << 22;  // idle polls in a row before a wave gives up (a logic error's cost bound; each poll sleeps)

// A WALKER: block_segment's loop, fed from its `rdy` ring instead of a queue, parking results in its `due` ring.
template <bool FUSED, bool COUNT>
__device__ __forceinline__ void frame_walker(const WfParams& w, FrameRings& R, FrameCtl& C, unsigned long long& tested) {
    const uint32_t lane = threadIdx.x & 63u;
    const GridDesc& g = w.grid;
    const BlockGrid& bg = w.bgrid;
    const HotObject* __restrict__ hot = w.rp.scene.hot;
    const float wnx_f = (float)bg.wnx, wny_f = (float)bg.wny;

    constexpr uint32_t kAlive = 1u;    // the lane holds a ray that is being walked ...
    constexpr uint32_t kOver = 2u;     // ... whose walk has ended (it may still wait for its parked exact test)
    constexpr uint32_t kPend = 4u;     // a candidate is parked for the next round of exact tests (pend_k)
    constexpr uint32_t kSphere = 8u;   // the current best hit is a sphere (closest_take's tie rule)
    constexpr uint32_t kDone = 16u;    // the lane holds a ray whose result (T, idx) is final and not yet parked in `due`
    uint32_t fl = 0u;
    uint32_t pix = 0;
    float rsx = 0.f, rsy = 0.f, rsz = 0.f, rdx = 0.f, rdy = 0.f, rdz = 0.f, dd = 0.f;
    uint32_t cur = 0;
    float fx = 0.f, fy = 0.f, fz = 0.f;
    float tx = 0.f, ty = 0.f, tz = 0.f, dtx = 0.f, dty = 0.f, dtz = 0.f;
    float T = kMaxFloat, limit = 0.f, t_stop = 0.f, slack = 0.f;
    int idx = -1;
    uint32_t pend_k = 0, done_k = 0xffffffffu;
    float4 acc0 = make_float4(0.f, 0.f, 0.f, 0.f), acc1 = make_float4(0.f, 0.f, 0.f, 0.f);  // the pixel's accumulators ride along
    // this side's cursors (wave-uniform): counts as published in C, slots wrapped
    uint32_t due_tail = 0, due_slot = 0, rdy_head = 0, rdy_slot = 0;
    uint32_t spins = 0;
    bool said_starving = false;
    unsigned long long s_rays = 0, s_trips = 0, s_live = 0, s_flush = 0, s_refill = 0, s_starved = 0, s_stalled = 0;  // COUNT only

    for (;;) {
        // ---- hand-out: finished rays are parked, lanes without a ray take the next one that waits ----
        const unsigned long long walking_m = __ballot((fl & kAlive) != 0u);
        if (64u - (uint32_t)__popcll(walking_m) >= (uint32_t)RT_FRAME_REFILL_MIN) {
            const bool fin = (fl & kDone) != 0u;
            const unsigned long long fm = __ballot(fin);
            if (fm != 0ull) {
                const uint32_t room = kRing - (due_tail - ctl_load(&C.due_head));
                const uint32_t rank = __builtin_amdgcn_mbcnt_hi((uint32_t)(fm >> 32), __builtin_amdgcn_mbcnt_lo((uint32_t)fm, 0u));
                if (COUNT && lane == 0u) { ++s_refill; s_stalled += (uint32_t)__popcll(fm) > room ? (uint32_t)__popcll(fm) - room : 0u; }
                if (fin && rank < room) {
                    uint32_t slot = due_slot + rank;
                    slot -= slot >= kRing ? kRing : 0u;
                    R.due[0][slot] = make_float4(rsx, rsy, rsz, T);
                    R.due[1][slot] = make_float4(rdx, rdy, rdz, __uint_as_float((uint32_t)idx));
                    R.due[2][slot] = acc0;
                    R.due[3][slot] = acc1;
                    R.due_pix[slot] = pix;
                    fl = 0u;
                }
                const uint32_t n_fin = (uint32_t)__popcll(fm);
                const uint32_t parked = n_fin < room ? n_fin : room;
                if (parked != 0u) {
                    due_tail += parked;
                    due_slot += parked;
                    due_slot -= due_slot >= kRing ? kRing : 0u;
                    ctl_store(&C.due_tail, due_tail);  // (release: the entries above are in LDS before the count says so)
                }
            }
            const bool empty = (fl & (kAlive | kDone)) == 0u;
            const unsigned long long em = __ballot(empty);
            if (em != 0ull) {
                const uint32_t waiting = ctl_load(&C.rdy_tail) - rdy_head;
                const uint32_t n_em = (uint32_t)__popcll(em);
                const uint32_t taken = n_em < waiting ? n_em : waiting;
                const uint32_t k = __builtin_amdgcn_mbcnt_hi((uint32_t)(em >> 32), __builtin_amdgcn_mbcnt_lo((uint32_t)em, 0u));
                if (empty && k < taken) {
                    uint32_t slot = rdy_slot + k;
                    slot -= slot >= kRing ? kRing : 0u;
                    const float4 a0 = R.rdy[0][slot], a1 = R.rdy[1][slot];
                    acc0 = R.rdy[2][slot];
                    acc1 = R.rdy[3][slot];
                    rsx = a0.x; rsy = a0.y; rsz = a0.z; rdx = a1.x; rdy = a1.y; rdz = a1.z;
                    pix = __float_as_uint(a1.w);
                    if (COUNT) ++s_rays;
                    // ---- the ray's walk (block_segment's hand-out) ----
                    const Ray ray = {rsx, rsy, rsz, 1.0f, rdx, rdy, rdz, 0.0f};  // (what every ray of a grid-able frame carries: rt_create checks)
                    T = kMaxFloat; idx = -1;
                    bool cur_sphere = false;
                    done_k = __float_as_uint(a0.w);  // frame_step's note: the object the ray leaves has had its exact test
                    dd = rdx * rdx + rdy * rdy + rdz * rdz;
                    slack = dd > 0.f ? kWalkSlackCells * bg.cell * __builtin_amdgcn_rsqf(dd) * 1.0001f : 3.0e38f;
                    bool start = false, brute = false;
                    const Walk w0 = walk_begin(bg, ray, 3.0e38f);  // (a ray with a NaN in it: not alive)
                    if (w0.alive) {
                        const float dmin = __builtin_fminf(__builtin_fminf(w0.dtx, w0.dty), w0.dtz);
                        const bool tame = dd > 1.0e-30f && dd < 1.0e30f && w0.t_enter <= 4096.f * dmin;
                        brute = !tame;
                        start = tame;
                        fx = (float)(w0.ix + (int)kBlockBorder); fy = (float)(w0.iy + (int)kBlockBorder); fz = (float)(w0.iz + (int)kBlockBorder);
                        cur = ((uint32_t)(w0.iz + (int)kBlockBorder) * bg.wny + (uint32_t)(w0.iy + (int)kBlockBorder)) * bg.wnx + (uint32_t)(w0.ix + (int)kBlockBorder);
                        tx = w0.tx; ty = w0.ty; tz = w0.tz; dtx = w0.dtx; dty = w0.dty; dtz = w0.dtz;
                        t_stop = __builtin_fminf(w0.t_exit + 0.25f * dmin, 3.0e38f);
                    }
                    if (brute) {  // a ray the walk is not made for (a direction of absurd magnitude) tests every object, here and now
                        for (uint32_t k2 = 0; k2 < w.rp.scene.n_objs; ++k2) {
                            float t;
                            bool sphere;
                            const bool cand = lane_candidate<FUSED, true, false>(hot + k2, ray, t, sphere);
                            if (COUNT) ++tested;
                            if (cand) closest_take(t, (int)k2, sphere, T, idx, cur_sphere);
                        }
                    }
                    limit = __builtin_fminf(T + slack, t_stop);
                    // no cell to look at, or everything tested already: the result is final
                    fl = start ? (kAlive | (cur_sphere ? kSphere : 0u)) : kDone;
                }
                if (taken != 0u) {
                    rdy_head += taken;
                    rdy_slot += taken;
                    rdy_slot -= rdy_slot >= kRing ? kRing : 0u;
                    ctl_store(&C.rdy_head, rdy_head);  // (release: the entries have been read before the slots are given back)
                }
            }
        }
        const unsigned long long live = __ballot((fl & kAlive) != 0u);
        if (live == 0ull) {
            // nothing to walk: wait for the stepper (more rays, room in `due`, or the end of this walker's part of the frame)
            const bool holding = __ballot((fl & kDone) != 0u) != 0ull;
            if (!holding && ctl_load(&C.quit) != 0u && ctl_load(&C.rdy_tail) == rdy_head) break;
            if (!holding && !said_starving) { ctl_store(&C.starving, 1u); said_starving = true; }
            if (COUNT && lane == 0u) ++s_starved;
            if (++spins > kFrameSpinLimit) { if (lane == 0u) atomicAdd(&w.counts[RS_FRAME_STUCK], 1u); break; }
            __builtin_amdgcn_s_sleep(8);
            continue;
        }
        if (said_starving) { ctl_store(&C.starving, 0u); said_starving = false; }
#ifndef RT_FRAME_YIELD_LIVE
#define RT_FRAME_YIELD_LIVE 0
#endif
        if (RT_FRAME_YIELD_LIVE != 0 && (uint32_t)__popcll(live) < (uint32_t)RT_FRAME_YIELD_LIVE && ctl_load(&C.due_head) != due_tail && spins < 64u) {
            // few lanes walking, nothing waiting, and the stepper has not yet got to what this wave has parked: a trip costs the
            // same ~370 instructions with 20 lanes as with 60 - leave the issue slots to the stepper for a moment (bounded: 64 naps)
            ++spins;
            __builtin_amdgcn_s_sleep(4);
            continue;
        }
        spins = 0u;
        // ---- one trip: the block under the cursor (block_segment, trip for trip) ----
        const bool walking = (fl & (kAlive | kOver)) == kAlive;
        uint32_t stalled = 0u;
        if (COUNT && lane == 0u) { ++s_trips; s_live += (unsigned long long)__popcll(live); }
        if (walking) {
            const uint32_t b = cur & 0xffffffu, pos = cur >> 24;
            const uint4 q0 = table_at(bg.blocks, 2u * b);
            const uint4 q1 = table_at(bg.blocks, 2u * b + 1u);
            const uint32_t nxt = q0.x & 0xffffffu;
            const float inv = __builtin_ldexpf(bg.inv_step, -(int)((q0.x >> 27) & 3u));
            const float olx = __builtin_fmaf(rsx - __builtin_fmaf(fx, bg.cell, bg.c0x), inv, 128.0f);
            const float oly = __builtin_fmaf(rsy - __builtin_fmaf(fy, bg.cell, bg.c0y), inv, 128.0f);
            const float olz = __builtin_fmaf(rsz - __builtin_fmaf(fz, bg.cell, bg.c0z), inv, 128.0f);
            uint32_t pm = block_pretests(q0, q1, olx, oly, olz, rdx, rdy, rdz, -dd, 1.0f - g.pretest_alpha);
            pm &= 0x7fu << pos;
            uint32_t back = 0u;
            while (pm != 0u) {
                const uint32_t e = (uint32_t)__builtin_ctz(pm);
                const uint32_t k = table_at(bg.ids, 8u * b + e);
                const bool parked = (fl & kPend) != 0u;
                const bool dup = (k == done_k) || (parked && k == pend_k);
                const bool wait = !dup && parked;
                const bool take = !dup && !parked;
                pend_k = take ? k : pend_k;
                fl |= take ? kPend : 0u;
                stalled = wait ? 1u : stalled;
                back = wait ? e : back;
                pm = wait ? 0u : (pm & (pm - 1u));
            }
            const bool adv = stalled == 0u && nxt == 0u;
            const float tmin = __builtin_fminf(__builtin_fminf(tx, ty), tz);
            const bool ax = (tx <= ty) && (tx <= tz);
            const bool ay = !ax && (ty <= tz);
            const bool az = !ax && !ay;
            tx += (adv && ax) ? dtx : 0.f;
            ty += (adv && ay) ? dty : 0.f;
            tz += (adv && az) ? dtz : 0.f;
            fx += (adv && ax) ? __builtin_copysignf(1.0f, rdx) : 0.f;
            fy += (adv && ay) ? __builtin_copysignf(1.0f, rdy) : 0.f;
            fz += (adv && az) ? __builtin_copysignf(1.0f, rdz) : 0.f;
            fl |= (adv && tmin > limit) ? kOver : 0u;
            {
                uint32_t skips = ((q0.x >> 24) & 7u) | ((q0.x >> 26) & 0x38u);
                skips = (adv && (fl & kOver) == 0u) ? (skips < (uint32_t)RT_BLOCK_SKIP_CAP ? skips : (uint32_t)RT_BLOCK_SKIP_CAP) : 0u;
                if (!bg.take_skips) skips = 0u;
                while (skips != 0u) {
                    const float tm = __builtin_fminf(__builtin_fminf(tx, ty), tz);
                    const bool sx = (tx <= ty) && (tx <= tz);
                    const bool sy = !sx && (ty <= tz);
                    const bool sz = !sx && !sy;
                    tx += sx ? dtx : 0.f;
                    ty += sy ? dty : 0.f;
                    tz += sz ? dtz : 0.f;
                    fx += sx ? __builtin_copysignf(1.0f, rdx) : 0.f;
                    fy += sy ? __builtin_copysignf(1.0f, rdy) : 0.f;
                    fz += sz ? __builtin_copysignf(1.0f, rdz) : 0.f;
                    const bool outside = tm > limit;
                    fl |= outside ? kOver : 0u;
                    skips = outside ? 0u : skips - 1u;
                }
            }
            const uint32_t cell = (uint32_t)__builtin_fmaf(__builtin_fmaf(fz, wny_f, fy), wnx_f, fx);
            cur = stalled != 0u ? (b | (back << 24)) : (adv ? cell : nxt);
        }
        // ---- the exact tests, when enough lanes wait for them ----
        const unsigned long long pending = __ballot((fl & kPend) != 0u);
        if (pending != 0ull) {
            const unsigned long long stuck = __ballot((fl & kPend) != 0u && (stalled != 0u || (fl & kOver) != 0u));
            const uint32_t n_live = (uint32_t)__popcll(live);
            if ((uint32_t)__popcll(pending) >= (uint32_t)RT_WALK3_DEFER_PENDING || ((uint32_t)__popcll(stuck) << RT_WALK3_STUCK_SHIFT) >= n_live) {
                if (COUNT && lane == 0u) ++s_flush;
                if ((fl & kPend) != 0u) {
                    float t;
                    bool sphere;
                    const Ray ray = {rsx, rsy, rsz, 1.0f, rdx, rdy, rdz, 0.0f};
                    const bool cand = lane_candidate<FUSED, true, false>(hot + pend_k, ray, t, sphere);
                    if (COUNT) ++tested;
                    done_k = pend_k;
                    bool cur_sphere = (fl & kSphere) != 0u;
                    if (cand) {
                        closest_take(t, (int)pend_k, sphere, T, idx, cur_sphere);
                        limit = __builtin_fminf(T + slack, t_stop);
                    }
                    fl = (fl & ~(kPend | kSphere)) | (cur_sphere ? kSphere : 0u);
                }
            }
        }
        // ---- a finished walk with nothing parked: the ray's result is final (parked in `due` at the next hand-out) ----
        if ((fl & (kAlive | kOver | kPend)) == (kAlive | kOver)) fl = kDone;
    }
    if (COUNT) {  // engineering aid (RT_WALK_STATS), row 0: as the walks' rows
        const unsigned long long r = wave_sum64(s_rays);
        if (lane == 0u) {
            unsigned long long* acc = w.rp.counters->walk[0];
            const unsigned long long v0[8] = {r, s_trips, s_live, s_starved, s_stalled, 0ull, s_flush, s_refill};
            for (int j = 0; j < 8; ++j) if (v0[j]) atomicAdd(&acc[j], v0[j]);
        }
    }
}

// The STEPPER of a workgroup: draws the runs of new pixels, and for each of its walkers in turn runs the steps that are due -
// one pixel per lane, topped up with new pixels - and queues what goes on in that walker's `rdy`.
template <bool FUSED, bool COUNT>
__device__ __forceinline__ void frame_stepper(const WfParams& w, const uint32_t* __restrict__ queue, uint32_t n_queue, uint32_t wave, uint32_t n_waves,
                                              uint32_t* __restrict__ run_ctr, FrameRings* __restrict__ rings, FrameCtl* __restrict__ ctl, uint32_t* __restrict__ mine_state, Ctx& cnt) {
    const uint32_t lane = threadIdx.x & 63u;
#ifndef RT_FRAME_STEPPER_PRIO
#define RT_FRAME_STEPPER_PRIO 3
#endif
    // One stepper feeds three walkers: whenever it can issue, it should (the walkers of its SIMD are issue-hungry, and a stepper
    // that gets a quarter of the issue slots starves all three: 28 of 64 lanes walking, 15.7 ms per cfg4 frame)
    __builtin_amdgcn_s_setprio(RT_FRAME_STEPPER_PRIO);
    RunCursor rc;
    bool more = false;
    if (rc.begin(n_queue, wave, n_waves, run_ctr, lane)) more = rc.dynamic;
    else { rc.next = 0u; rc.seg_end = 0u; }
    uint32_t& next = rc.next;
    const uint32_t& seg_end = rc.seg_end;
    // per walker, 8 words of this wave's own in LDS (`mine_state`, zeroed by the kernel; a rolled loop over the walkers - the step is
    // ~1500 instructions, three copies of it would not fit the instruction cache): what this side has published (due_head,
    // rdy_tail), its wrapped slots, the pixels the walker has in flight
    uint32_t spins = 0;
    unsigned long long s_batches = 0, s_step_lanes = 0, s_passes = 0, s_idle = 0;  // COUNT only

    for (;;) {
        bool stepped = false;
#pragma nounroll
        for (uint32_t v = 0; v < kFrameWalkers; ++v) {
            uint32_t* st = mine_state + 8u * v;
            uint32_t due_head_v = st[0], due_slot_v = st[1], rdy_tail_v = st[2], rdy_slot_v = st[3], in_flight_v = st[4];
            if (next >= seg_end && more) more = rc.grab(n_queue, run_ctr, lane);  // on to another run, if any is left
            FrameRings& R = rings[v];
            FrameCtl& C = ctl[v];
            const uint32_t due_n = ctl_load(&C.due_tail) - due_head_v;
            const uint32_t rdy_n = rdy_tail_v - ctl_load(&C.rdy_head);
            const uint32_t left = next < seg_end ? seg_end - next : 0u;
            const uint32_t headroom = kPixelsInFlight - in_flight_v;
            const uint32_t admissible = left < headroom ? left : headroom;
            const uint32_t room = kRing - rdy_n;
            const uint32_t want = room < 64u ? room : 64u;   // what goes on is queued in rdy: never more than it can take
            const uint32_t avail = due_n + admissible;
            const bool normal = avail >= 64u && want >= 48u;                       // a full wave of steps, and room for what they produce
            const bool pressure = due_n + 16u >= kRing && want != 0u;               // due is about to fill up
            const bool flush = avail != 0u && want != 0u && rdy_n == 0u && ctl_load(&C.starving) != 0u;  // the walker has nothing: whatever there is
            if (COUNT && lane == 0u) ++s_passes;
            if (!(normal || pressure || flush)) continue;
            stepped = true;
            const uint32_t d = due_n < want ? due_n : want;
            const uint32_t a = (want - d) < admissible ? (want - d) : admissible;
            const bool run = lane < d + a;
            const bool primary = lane >= d;
            if (COUNT && lane == 0u) { ++s_batches; s_step_lanes += d + a; }
            // what goes on is put down in `rdy` behind its tail, lane by lane (d + a <= want <= its room), and closed up afterwards
            uint32_t pslot = rdy_slot_v + lane;
            pslot -= pslot >= kRing ? kRing : 0u;
            uint32_t dslot = due_slot_v + lane;
            dslot -= dslot >= kRing ? kRing : 0u;
            bool goes_on = false;
            if (run) {
                uint32_t spix = 0u;
                if (primary) {
                    const uint32_t mine = next + (lane - d);
                    const uint32_t entry = w.identity_queue ? mine : queue[mine];
                    spix = w.identity_queue ? mine : (entry & kQueuePixel);
                }
                goes_on = frame_step<FUSED, false>(w, RingIO{R, dslot, pslot}, primary, spix, cnt);
            }
            const unsigned long long gm = __ballot(goes_on);
            {   // close the gaps the pixels that ended have left (reads before writes: one wave, LDS in order)
                const uint32_t rank = __builtin_amdgcn_mbcnt_hi((uint32_t)(gm >> 32), __builtin_amdgcn_mbcnt_lo((uint32_t)gm, 0u));
                const bool move = goes_on && rank != lane;
                float4 m0, m1, m2, m3;
                if (move) { m0 = R.rdy[0][pslot]; m1 = R.rdy[1][pslot]; m2 = R.rdy[2][pslot]; m3 = R.rdy[3][pslot]; }
                if (move) {
                    uint32_t to = rdy_slot_v + rank;
                    to -= to >= kRing ? kRing : 0u;
                    R.rdy[0][to] = m0; R.rdy[1][to] = m1; R.rdy[2][to] = m2; R.rdy[3][to] = m3;
                }
            }
            const uint32_t n_on = (uint32_t)__popcll(gm);
            if (n_on != 0u) {
                rdy_tail_v += n_on;
                rdy_slot_v += n_on;
                rdy_slot_v -= rdy_slot_v >= kRing ? kRing : 0u;
                ctl_store(&C.rdy_tail, rdy_tail_v);  // (release: the rays are in LDS before the count says so)
            }
            if (d != 0u) {
                due_head_v += d;
                due_slot_v += d;
                due_slot_v -= due_slot_v >= kRing ? kRing : 0u;
                ctl_store(&C.due_head, due_head_v);  // (release: the entries have been read before the slots are given back)
            }
            next += a;
            in_flight_v = in_flight_v + a - (d + a - n_on);
            st[0] = due_head_v; st[1] = due_slot_v; st[2] = rdy_tail_v; st[3] = rdy_slot_v; st[4] = in_flight_v;
        }
        if (stepped) { spins = 0u; continue; }
        // nothing was due anywhere
        bool all_idle = next >= seg_end && !more;
        for (uint32_t v = 0; v < kFrameWalkers; ++v) all_idle = all_idle && mine_state[8u * v + 4u] == 0u;
        if (all_idle) break;
        if (COUNT && lane == 0u) ++s_idle;
        if (++spins > kFrameSpinLimit) { if (lane == 0u) atomicAdd(&w.counts[RS_FRAME_STUCK], 1u); break; }
        __builtin_amdgcn_s_sleep(4);
    }
#pragma unroll
    for (uint32_t v = 0; v < kFrameWalkers; ++v) ctl_store(&ctl[v].quit, 1u);
    if (COUNT && lane == 0u) {  // row 1: step batches, passes over a walker, step lanes, idle polls
        unsigned long long* acc1 = w.rp.counters->walk[1];
        const unsigned long long v1[8] = {s_batches, s_passes, s_step_lanes, s_idle, 0ull, 0ull, 0ull, 0ull};
        for (int j = 0; j < 8; ++j) if (v1[j]) atomicAdd(&acc1[j], v1[j]);
    }
}

template <bool FUSED, bool COUNT>
__global__ __launch_bounds__(256) __attribute__((amdgpu_waves_per_eu(RT_FRAME_WAVES, RT_FRAME_WAVES))) void wf_frame(const WfParams wk, uint32_t* __restrict__ run_ctr) {
    WfParams w = wk;
    if (!resolve_round(w)) return;
    __shared__ FrameRings s_rings[kFrameWalkers];
    __shared__ FrameCtl s_ctl[kFrameWalkers];
    __shared__ uint32_t s_stepper[8 * kFrameWalkers];
    const uint32_t n_queue = w.n_prev_closest;
    if (n_queue == 0u) return;
    if (threadIdx.x < kFrameWalkers * (uint32_t)(sizeof(FrameCtl) / 4u)) reinterpret_cast<uint32_t*>(s_ctl)[threadIdx.x] = 0u;
    if (threadIdx.x < 8u * kFrameWalkers) s_stepper[threadIdx.x] = 0u;
    __syncthreads();   // (the only barrier of the kernel: the counters start from zero)
    const uint32_t role = threadIdx.x >> 6;
    Ctx cnt{w, 0, 0ull, 0ull, 0ull, false, false, 0u, 0u, make_float4(0.f, 0.f, 0.f, 0.f), false, {}, {}, {}, {}};
    unsigned long long tested = 0;
    if (role == 0u) frame_stepper<FUSED, COUNT>(w, w.q_prev_closest, n_queue, blockIdx.x, gridDim.x, run_ctr, s_rings, s_ctl, s_stepper, cnt);
    else frame_walker<FUSED, COUNT>(w, s_rings[role - 1u], s_ctl[role - 1u], tested);
    if (COUNT) {
        add_ray_counters(w, cnt);
        if (tested) atomicAdd(&w.rp.counters->tests, tested);
    }
}

// ---- rounds of {closest-hit walk, step} (round 4) --------------------------------------------------------------------------
// What wf_frame's measurements said: fusing walk and step into one launch costs the walk a third of its
// waves (the step needs 128 registers, the walk 80, a kernel has one allocation) and that outweighs every barrier it removes.
// What does carry over is the STEP itself: with the light loop's shadow rays tested on the spot - through the last light's tile,
// the way wf_resume already treats a path's last hit - a pixel has ONE ray kind in flight, ever. The round machine shrinks to
// {walk the closest-hit queue, step} per bounce: no shadow queue and no shadow walk (1.8 ms of launches per cfg4 frame, each
// ray rebuilt from its hit point first), no second visit of a hit a round later (its point, normal, phase word stored and
// read back, its material gathered twice), no stale-specular stragglers (the backward scan finishes inside the step), hence
// no wf_finish. Per pixel and round the step reads the ray and its result (40 bytes) + the accumulators (32) and writes the
// next ray + the accumulators (64): 136 bytes where wf_resume moves ~180 and the shadow walk another 48.
// The step is frame_step with the pixel state in HBM (StateIO).
// one step of every pixel of the round's closest-hit queue; what goes on is appended to the next round's
#ifndef RT_STEP_BLOCKS
#define RT_STEP_BLOCKS 0   // 1: the stale-specular scans' shadow rays through the block grid first (as wf_finish), 0: the fine grid only
#endif
template <bool FUSED, bool COUNT>
__global__ __launch_bounds__(kResumeThreads) __attribute__((amdgpu_waves_per_eu(RT_RESUME_WAVES_PER_EU))) void wf_step(const WfParams wk) {
    WfParams w = wk;
    if (!resolve_round(w)) return;
    if (blockIdx.x * kResumeThreads >= w.n_prev_closest) return;  // the grid is sized for the most the queue can hold
    const uint32_t t = blockIdx.x * kResumeThreads + threadIdx.x;
    Ctx cnt{w, 0, 0ull, 0ull, 0ull, false, false, 0u, 0u, make_float4(0.f, 0.f, 0.f, 0.f), false, {}, {}, {}, {}};
    bool goes_on = false;
    uint32_t pix = 0u;
    if (t < w.n_prev_closest) {
        const uint32_t entry = w.identity_queue ? t : w.q_prev_closest[t];
        pix = entry & kQueuePixel;
        goes_on = frame_step<FUSED, RT_STEP_BLOCKS != 0>(w, StateIO{w, pix}, w.first_round != 0u, pix, cnt);
    }
    block_push(goes_on, false, pix, w.q_closest, w.q_any, w.counts);
    if (COUNT) add_ray_counters(w, cnt);
}

// the frames wf_frame / the {walk, step} rounds are made for (everything else keeps the round machine)
static bool step_rounds_possible(const WfParams& w, int kernel);
static bool frame_kernel_applies(const WfParams& w, int kernel) {
    const char* env = std::getenv("RT_FRAME_KERNEL");  // "1": on (read per launch: tests render both ways in one process)
    if (!(env && env[0] == '1')) return false;
    return step_rounds_possible(w, kernel);
}
static bool step_rounds_apply(const WfParams& w, int kernel) {
    const char* env = std::getenv("RT_STEP_ROUNDS");  // "1": on (read per launch)
    if (!(env && env[0] == '1')) return false;
    return step_rounds_possible(w, kernel);
}
static bool step_rounds_possible(const WfParams& w, int kernel) {
    if (std::getenv("RT_WALK2") || std::getenv("RT_WALK3")) return false;  // (a run that asks for one of the round machine's walks by name)
    return kernel == 2 && w.grid.enabled && !w.rp.scene.literal && !w.grid.has_triangles && w.grid.n_always == 0u && w.bgrid.enabled &&
           w.ltiles.enabled && w.ltiles.blocks_enabled && w.rp.scene.n_lights >= 1u && w.ltiles.light == w.rp.scene.n_lights - 1u;
}

constexpr uint32_t kMinSlicePairs = 2048;
constexpr uint32_t kMaxSlices = 16;

// Between two rounds (one workgroup): the queues the round has filled become the next round's input, their append
// counters and the grid walk's run tickets start from zero again, and the frame's tail is handed to wf_finish once
// few enough pixels are alive.
__global__ __launch_bounds__(256) void wf_advance(uint32_t* __restrict__ rs, uint32_t finish_threshold, uint32_t allow_finish) {
    for (uint32_t k = threadIdx.x; k < 2u * kTicketWords; k += 256u) rs[kTicketBase + k] = 0u;
    if (threadIdx.x != 0u) return;
    if (rs[RS_FINISH] != 0u) return;  // the frame is already past its rounds: leave the hand-over as it is
    const uint32_t nc = rs[RS_NEXT_CLOSEST], na = rs[RS_NEXT_ANY];
    rs[RS_N_CLOSEST] = nc;
    rs[RS_N_ANY] = na;
    rs[RS_NEXT_CLOSEST] = 0u;
    rs[RS_NEXT_ANY] = 0u;
    rs[RS_CUR] ^= 1u;
    if (nc + na == 0u) rs[RS_FINISH] = 2u;
    else if (allow_finish && nc + na <= finish_threshold) rs[RS_FINISH] = 1u;
    else rs[RS_ROUNDS] += 1u;
}

// First round of a frame without padding work-items: the queue is the identity (WfParams::identity_queue), only its length
// has to be put down.
__global__ void wf_identity_round(uint32_t* __restrict__ rs, uint32_t n) {
    rs[RS_N_CLOSEST] = n;
    rs[RS_ROUNDS] = 1u;
}

// ---- host driver -----------------------------------------------------------------------------------------------------
static inline dim3 grid_for(uint64_t n) { return dim3((uint32_t)((n + 255u) / 256u)); }
// one wave per kSegment queue entries, four waves per workgroup
#ifndef RT_MAX_WAVES
#define RT_MAX_WAVES 8192  // 256 CUs x 4 SIMDs x 8: every wave that can be resident
#endif
// A round's two walks run on two streams. Both are persistent: whichever gets the wave slots first keeps them until its
// queue is empty, so launched at full size they run one after the other. The closest-hit walk is bound by instruction
// issue, the shadow walk (light tiles: a few dependent loads per ray) by memory latency - given a share of the slots
// each, the second might hide in the first. MEASURED (cfg4 frame, closest / shadow wave caps): 8192 / 8192 (launched at full
// size, in effect one after the other) 19.6 ms; 6144 / 2048 20.15; 5120 / 2048 20.2; 5120 / 1024 20.8; 4096 / 2048 21.7 -
// the closest-hit walk needs every slot it can get, the knobs stay at "full size".
#ifndef RT_MAX_WAVES_CLOSEST_SHARED
#define RT_MAX_WAVES_CLOSEST_SHARED 5120   // round 4 (block walk + light-tile blocks): the closest-hit walk of a big round at FIVE waves per SIMD, the sixth
#endif                                     // slot left to the shadow walk - MEASURED, closest / shadow caps, cfg4 frame, two runs each: 8192 / 4096 (round 3's)
                                           // 11.68 / 11.70 ms; 5120 / 2560: 11.38 / 11.48; 5120 / 3072: 11.44 / 11.50; 5120 / 4096: 11.50 / 11.57;
                                           // 4864 / 3072: 11.53 / 11.58; 5376 / 3072: 11.54 / 11.55; 4096 / 4096: 12.08 (tools/ab/env_sweep_caps.py)
#ifndef RT_MAX_WAVES_ANY_SHARED
#define RT_MAX_WAVES_ANY_SHARED RT_MAX_WAVES
#endif
static inline dim3 persistent_grid(uint64_t n, uint64_t max_waves) {
    uint64_t waves = (n + 63) / 64;  // (the kernel shortens its runs to 64 entries for small queues)
    if (waves > max_waves) waves = max_waves;
    if (waves == 0) waves = 1;
    return dim3((uint32_t)((waves + 3u) / 4u));
}

size_t wavefront_state_bytes(uint64_t n_local) { return (size_t)F_COUNT * sizeof(float) * (size_t)n_local; }
size_t wavefront_queue_bytes(uint64_t n_local) { return sizeof(uint32_t) * (size_t)n_local; }
size_t wavefront_counter_bytes() { return sizeof(uint32_t) * (size_t)(kTicketBase + 2u * kTicketWords); }

// one launch of the persistent grid walk (the template arguments pick the compiled variant); `n_max` = the most the
// queue can hold, the kernel reads its real length from the round state
template <bool FUSED, bool ANY>
static void launch_persistent(const WfParams& w, uint64_t n_max, uint32_t* ticket, hipStream_t s, bool shared) {
    // The shadow walk (light tiles: two or three short trips per ray) is bound by what a wave costs to start, not by how many
    // rays are in flight: MEASURED, rank 0's share of the cfg4 frame at world = 1 / 2 / 4 / 8 with its wave cap at 8192:
    // 17.44 / 9.51 / 5.57 / 3.59 ms; 4096: 17.40 / 9.34 / 5.31 / 3.39; 2048: 18.15 / 9.57 / 5.36 / 3.32; 1024: 20.2 / 10.6 / 5.9 / 3.56.
    const bool mesh = w.grid.has_triangles != 0u;  // (meshes keep round 3's caps - their shadow walk is the heavier kernel: cfg5 38.4 / 38.9 ms against 40.0 / 39.4 with the new ones)
    uint64_t any_cap = std::min<uint64_t>(RT_MAX_WAVES_ANY_SHARED, n_max <= (3ull << 20) ? 2048u : (mesh ? 4096u : 2560u));  // (round 4: 2560 beside a closest-hit walk of 5120, see above)
    // Round 3, block walk: a SMALL closest-hit queue (one rank's share of the frame at 8 ranks: 2.1 M pixels) is better served by
    // 4 waves per SIMD than by all 6 - the queue holds only ~3 rays per lane slot, the launch is mostly its own tail, and the
    // shadow walk finds free slots beside it instead of running in that tail. MEASURED, rank 0's share at world = 1 / 2 / 4 / 8,
    // closest / shadow wave caps: 8192 / 4096|2048: 12.66 / 6.84 / 3.94 / 2.55 ms; 4096 / 2048 everywhere: 12.97 / 6.98 / 3.98 / 2.32;
    // 5120 / 1024: 13.49 / 7.10 / 3.96 / 2.39.
    uint64_t closest_cap = shared ? (n_max <= (3ull << 20) ? std::min<uint64_t>(RT_MAX_WAVES_CLOSEST_SHARED, 4096u) : (mesh ? RT_MAX_WAVES : RT_MAX_WAVES_CLOSEST_SHARED)) : RT_MAX_WAVES;
    if (const char* env = std::getenv("RT_WAVES_CLOSEST")) closest_cap = (uint64_t)std::max(64, std::atoi(env));  // measurement knobs
    if (const char* env = std::getenv("RT_WAVES_ANY")) any_cap = (uint64_t)std::max(64, std::atoi(env));
    const dim3 grid = persistent_grid(n_max, ANY ? any_cap : closest_cap), block(256);
    const bool tri = w.grid.has_triangles != 0u;
    // scenes without triangles: the unified walk (walk_segment), where its record table was built; RT_WALK2=closest / any /
    // none picks which of the two walks use it (measurement knob)
    const char* walk2_env = std::getenv("RT_WALK2");  // (read per launch: tests switch walks inside one process)
    const char* walk3_env = std::getenv("RT_WALK3");  // "0": closest-hit rays through walk_segment / trace_segment instead
    if (!ANY && w.bgrid.enabled && !(walk3_env && walk3_env[0] == '0')) {
        if (tri) {  // (meshes: the same walk with the triangle branch in its exact tests)
            if (w.count_rays) hipLaunchKernelGGL((wf_walk_blocks<FUSED, true, true>), grid, block, 0, s, w, ticket);
            else hipLaunchKernelGGL((wf_walk_blocks<FUSED, false, true>), grid, block, 0, s, w, ticket);
        } else {
            if (w.count_rays) hipLaunchKernelGGL((wf_walk_blocks<FUSED, true, false>), grid, block, 0, s, w, ticket);
            else hipLaunchKernelGGL((wf_walk_blocks<FUSED, false, false>), grid, block, 0, s, w, ticket);
        }
        return;
    }
    const bool walk2_allowed = !walk2_env || (ANY ? std::strcmp(walk2_env, "any") == 0 : std::strcmp(walk2_env, "closest") == 0) || std::strcmp(walk2_env, "both") == 0;
    if (!tri && w.grid.walk_rec && walk2_allowed && (!ANY || !w.ltiles.enabled || w.ltiles.walk_base != 0u || w.ltiles.blocks_enabled)) {
        if (w.count_rays) hipLaunchKernelGGL((wf_walk<FUSED, ANY, true>), grid, block, 0, s, w, ticket);
        else hipLaunchKernelGGL((wf_walk<FUSED, ANY, false>), grid, block, 0, s, w, ticket);
        return;
    }
    if (w.count_rays) {
        if (tri) hipLaunchKernelGGL((wf_trace_grid_persistent<FUSED, ANY, true, true>), grid, block, 0, s, w, ticket);
        else hipLaunchKernelGGL((wf_trace_grid_persistent<FUSED, ANY, true, false>), grid, block, 0, s, w, ticket);
    } else {
        if (tri) hipLaunchKernelGGL((wf_trace_grid_persistent<FUSED, ANY, false, true>), grid, block, 0, s, w, ticket);
        else hipLaunchKernelGGL((wf_trace_grid_persistent<FUSED, ANY, false, false>), grid, block, 0, s, w, ticket);
    }
}

// A frame = wf_begin, then rounds of {trace closest || trace any} -> wf_resume -> wf_advance, then wf_finish. The
// rounds are enqueued in batches without looking at the queue lengths (device-side round state); the host reads the
// state back once per batch - once per frame when the first batch (as many rounds as the kernel's control flow
// needs at least) gets the frame down to wf_finish's share, which is the normal case.
template <int KERNEL, bool FUSED>
static hipError_t run_wavefront(WfParams w, WavefrontBuffers& buf, hipStream_t stream, uint32_t* rounds_out) {
    hipError_t e;
    const uint64_t n = w.rp.n_local;
    w.st = buf.state;
    w.counts = buf.counts;
    w.shadow_pairs = buf.shadow_pairs;
    w.grid = buf.grid;
    w.tiles = buf.tiles;
    w.ltiles = buf.light_tiles;
    w.bgrid = buf.blocks;
    for (int a = 0; a < 2; ++a) { w.qs[a][0] = buf.q_closest[a]; w.qs[a][1] = buf.q_any[a]; }
    uint32_t* rs = buf.counts;
    if ((e = hipMemsetAsync(rs, 0, wavefront_counter_bytes(), stream)) != hipSuccess) return e;
    // at or below this many live pixels the frame is finished by wf_finish instead of further rounds
    uint64_t finish_threshold = std::max<uint64_t>(2048, n / 128);
    if (const char* env = std::getenv("RT_WF_FINISH_THRESHOLD")) finish_threshold = (uint64_t)std::atoll(env);
    if (finish_threshold > 0xffffffffull) finish_threshold = 0xffffffffull;
    const bool use_grid = w.grid.enabled && !w.rp.scene.literal;
    // padding work-items exist only in the ragged last tile of a sharded frame: they must not be traced (their global ray
    // does not exist), so such frames go through wf_begin, which leaves them out of the first queue
    // (work-item -> ray is monotonic, so the last work-item decides; n is a whole number of tiles, the last run may be a short one)
    const uint64_t last_run = n ? (n - 1u) / w.rp.run_rays : 0u;
    const bool padded = w.rp.world > 1u && n > 0u && (last_run * w.rp.world + w.rp.rank) * w.rp.tile_rays + (n - last_run * w.rp.run_rays) > w.rp.n_rays;
    static const bool always_begin = std::getenv("RT_WF_ALWAYS_BEGIN") != nullptr;  // measurement knob
    const bool identity = !padded && !always_begin && n > 0;
    if (identity) {
        hipLaunchKernelGGL(wf_identity_round, dim3(1), dim3(1), 0, stream, rs, (uint32_t)n);
        if ((e = hipGetLastError()) != hipSuccess) return e;
    } else {
        hipLaunchKernelGGL(wf_begin, dim3((uint32_t)((n + kResumeThreads - 1) / kResumeThreads)), dim3(kResumeThreads), 0, stream, w);
        if ((e = hipGetLastError()) != hipSuccess) return e;
        hipLaunchKernelGGL(wf_advance, dim3(1), dim3(256), 0, stream, rs, (uint32_t)finish_threshold, 0u);
        if ((e = hipGetLastError()) != hipSuccess) return e;
    }

    static const bool one_stream = std::getenv("RT_WF_ONE_STREAM") != nullptr;  // measurement knob
    const bool use_frame = KERNEL == 2 && frame_kernel_applies(w, KERNEL);
    const bool use_step = KERNEL == 2 && !use_frame && step_rounds_apply(w, KERNEL);  // rounds of {closest-hit walk, wf_step}: no shadow queue, no wf_finish
    auto enqueue_round = [&](bool first, uint64_t nc_max, uint64_t na_max) -> hipError_t {
        hipError_t e2;
        w.first_round = first ? 1u : 0u;
        w.identity_queue = (first && identity) ? 1u : 0u;
        // The two launches of a round are independent (different rays, different result words): through the grid they
        // run side by side on two streams, so that each fills the other's tail and a small light-scan queue hides
        // behind a big reflection queue.
        const bool side_by_side = use_grid && nc_max && na_max && buf.side_stream && !one_stream;
        hipStream_t any_stream = stream;
        if (side_by_side) {
            any_stream = buf.side_stream;
            if ((e2 = hipEventRecord(buf.ev_fork, stream)) != hipSuccess) return e2;
            if ((e2 = hipStreamWaitEvent(any_stream, buf.ev_fork, 0)) != hipSuccess) return e2;
            launch_persistent<FUSED, true>(w, na_max, rs + kTicketBase + kTicketWords, any_stream, true);
            if ((e2 = hipGetLastError()) != hipSuccess) return e2;
            if ((e2 = hipEventRecord(buf.ev_join, any_stream)) != hipSuccess) return e2;
        }
        if (nc_max) {
            if (use_grid && first && w.tiles.enabled && w.rp.pinhole) {
                hipLaunchKernelGGL((wf_trace_primary_tiles<FUSED>), grid_for(nc_max), dim3(256), 0, stream, w);
            } else if (use_grid) {
                launch_persistent<FUSED, false>(w, nc_max, rs + kTicketBase, stream, side_by_side);  // (a grid implies direction.w = 0)
            } else {
                if (first && !w.rp.dir_w_zero) hipLaunchKernelGGL((wf_trace_closest<FUSED, false>), grid_for(nc_max), dim3(256), 0, stream, w);
                else hipLaunchKernelGGL((wf_trace_closest<FUSED, true>), grid_for(nc_max), dim3(256), 0, stream, w);
            }
            if ((e2 = hipGetLastError()) != hipSuccess) return e2;
        }
        if (side_by_side) {
            if ((e2 = hipStreamWaitEvent(stream, buf.ev_join, 0)) != hipSuccess) return e2;
        } else if (na_max) {
            if (w.rp.scene.literal) {
                hipLaunchKernelGGL((wf_trace_any_literal<FUSED>), grid_for(na_max), dim3(256), 0, stream, w);
                if ((e2 = hipGetLastError()) != hipSuccess) return e2;
            } else if (use_grid) {
                launch_persistent<FUSED, true>(w, na_max, rs + kTicketBase + kTicketWords, stream, false);
                if ((e2 = hipGetLastError()) != hipSuccess) return e2;
            } else {
                // slices of >= kMinSlicePairs pairs (amortises each launch's pipeline fill), at most kMaxSlices; the
                // survivors of a slice are the next one's queue (lengths stay on the device)
                const uint32_t n_pairs = w.rp.scene.n_pairs;
                uint32_t n_slices = n_pairs / kMinSlicePairs;
                n_slices = n_slices < 1u ? 1u : (n_slices > kMaxSlices ? kMaxSlices : n_slices);
                for (uint32_t sl = 0; sl < n_slices; ++sl) {
                    const uint32_t lo = (uint32_t)((uint64_t)n_pairs * sl / n_slices);
                    const uint32_t hi = (uint32_t)((uint64_t)n_pairs * (sl + 1) / n_slices);
                    const bool last = (sl + 1 == n_slices);
                    uint32_t* q_out = last ? nullptr : buf.q_slice[sl & 1u];
                    uint32_t* n_out = rs + (sl & 1u ? RS_SLICE_B : RS_SLICE_A);
                    const uint32_t* q_in = sl ? buf.q_slice[(sl - 1u) & 1u] : nullptr;
                    const uint32_t* n_in = rs + ((sl - 1u) & 1u ? RS_SLICE_B : RS_SLICE_A);
                    if (!last && (e2 = hipMemsetAsync(n_out, 0, sizeof(uint32_t), stream)) != hipSuccess) return e2;
                    hipLaunchKernelGGL((wf_trace_any_slice<FUSED>), grid_for(na_max), dim3(256), 0, stream, w, q_in, n_in, lo, hi,
                                       q_out, n_out, sl == 0 ? 1u : 0u);
                    if ((e2 = hipGetLastError()) != hipSuccess) return e2;
                }
            }
        }
        const uint64_t total_max = nc_max + na_max;
        if (first && use_frame) {
            // wf_frame: every pixel from its primary hit to its final store in one persistent launch (the first round's shadow
            // tickets are unused: the first round traces primary rays only)
            uint64_t waves = 256ull * 4ull * (uint64_t)RT_FRAME_WAVES;  // every wave that can be resident
            if (const char* env = std::getenv("RT_WAVES_FRAME")) waves = (uint64_t)std::max(64, std::atoi(env));  // measurement knob
            const dim3 grid = persistent_grid(nc_max, waves);
            if (w.count_rays) hipLaunchKernelGGL((wf_frame<FUSED, true>), grid, dim3(256), 0, stream, w, rs + kTicketBase + kTicketWords);
            else hipLaunchKernelGGL((wf_frame<FUSED, false>), grid, dim3(256), 0, stream, w, rs + kTicketBase + kTicketWords);
        } else if (use_step) {
            const dim3 grid((uint32_t)((nc_max + kResumeThreads - 1) / kResumeThreads));
            if (w.count_rays) hipLaunchKernelGGL((wf_step<FUSED, true>), grid, dim3(kResumeThreads), 0, stream, w);
            else hipLaunchKernelGGL((wf_step<FUSED, false>), grid, dim3(kResumeThreads), 0, stream, w);
        } else {
            hipLaunchKernelGGL((wf_resume<KERNEL, FUSED>), dim3((uint32_t)((total_max + kResumeThreads - 1) / kResumeThreads)),
                               dim3(kResumeThreads), 0, stream, w);
        }
        if ((e2 = hipGetLastError()) != hipSuccess) return e2;
        hipLaunchKernelGGL(wf_advance, dim3(1), dim3(256), 0, stream, rs, (uint32_t)finish_threshold, (use_grid && !use_step) ? 1u : 0u);
        return hipGetLastError();
    };

    // rounds the kernel's control flow needs at least (then the host looks): hittest 1; shade one per light; shade_and_reflect
    // one per bounce + 2 (the reflection ray leaves with the hit's first shadow ray), more in literal mode
    uint32_t batch = 1;
    if (KERNEL == 1) batch = w.rp.scene.n_lights + 1u;
    else if (KERNEL == 2) batch = w.rp.scene.literal ? 8u : w.rp.max_bounces + 2u;
    if (batch > 12u) batch = 12u;
    if (use_step) batch = std::min<uint32_t>(w.rp.max_bounces, 11u) + 1u;  // a step per ray of the deepest path: the primary ray and one per bounce
    if (use_frame) batch = 1u;  // the first "round" is the whole frame (wf_frame leaves nothing behind: the queues it would fill stay empty)
    if (const char* env = std::getenv("RT_WF_BATCH")) batch = (uint32_t)std::max(1, std::atoi(env));
    bool first = true;
    uint64_t nc_max = n, na_max = 0;  // the first round traces the primary rays only
    for (;;) {
        for (uint32_t r = 0; r < batch; ++r) {
            if ((e = enqueue_round(first, nc_max, na_max)) != hipSuccess) return e;
            if (first) { first = false; nc_max = (KERNEL == 0) ? 0 : n; na_max = (KERNEL == 0 || use_step) ? 0 : n; }
            if (nc_max + na_max == 0) break;
        }
        if (use_grid && !use_step) {
            // (wf_finish resumes pixels from their stored phase words and rays: never as a first round - with a batch of one
            //  round, RT_WF_BATCH=1, `w` would still carry the first round's flags here)
            w.first_round = 0u;
            w.identity_queue = 0u;
            hipLaunchKernelGGL((wf_finish<KERNEL, FUSED>), grid_for(finish_threshold ? finish_threshold : 1), dim3(256), 0, stream, w);
            if ((e = hipGetLastError()) != hipSuccess) return e;
        }
        if ((e = hipMemcpyAsync(buf.h_counts, rs, RS_WORDS * sizeof(uint32_t), hipMemcpyDeviceToHost, stream)) != hipSuccess) return e;
        if ((e = hipStreamSynchronize(stream)) != hipSuccess) return e;
        if (std::getenv("RT_ROUND_STATS"))  // engineering aid (with RT_WF_BATCH=1: after every round)
            std::fprintf(stderr, "[rounds] after %u round(s): closest queue %u, shadow queue %u, hand-over %u\n", buf.h_counts[RS_ROUNDS],
                         buf.h_counts[RS_N_CLOSEST], buf.h_counts[RS_N_ANY], buf.h_counts[RS_FINISH]);
        if (buf.h_counts[RS_FRAME_STUCK] != 0u) return hipErrorLaunchFailure;  // wf_frame gave up on pixels (a logic error): no frame is better than a wrong one
        if (buf.h_counts[RS_FINISH] != 0u) break;  // 2: the queues ran empty; 1: wf_finish (enqueued above) took the rest
        if (nc_max + na_max == 0) break;
        // still going (long light scans, deep bounce chains): what is alive bounds every later queue
        const uint64_t alive = (uint64_t)buf.h_counts[RS_N_CLOSEST] + buf.h_counts[RS_N_ANY];
        nc_max = std::min<uint64_t>(n, alive);
        na_max = use_step ? 0 : std::min<uint64_t>(n, alive);
        batch = std::getenv("RT_ROUND_STATS") ? 1u : 8u;
    }
    if (rounds_out) *rounds_out = buf.h_counts[RS_ROUNDS];
    return hipSuccess;
}

hipError_t launch_wavefront(const RenderParams& p, int kernel, bool fused, bool count, WavefrontBuffers& buf,
                            hipStream_t stream, uint32_t* rounds_out) {
    if (p.n_local >= 0x7fffffffull) return hipErrorInvalidValue;  // queue entries are pixel ids in 31 bits + a flag
    WfParams w;
    std::memset(&w, 0, sizeof(w));
    w.rp = p;
    w.kernel = kernel;
    w.count_rays = count ? 1u : 0u;
    switch (kernel) {
        case 0: return fused ? run_wavefront<0, true>(w, buf, stream, rounds_out) : run_wavefront<0, false>(w, buf, stream, rounds_out);
        case 1: return fused ? run_wavefront<1, true>(w, buf, stream, rounds_out) : run_wavefront<1, false>(w, buf, stream, rounds_out);
        case 2: return fused ? run_wavefront<2, true>(w, buf, stream, rounds_out) : run_wavefront<2, false>(w, buf, stream, rounds_out);
        default: return hipErrorInvalidValue;
    }
}

}  // namespace rt
