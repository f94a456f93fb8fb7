// rt_kernels.h - launch interface between the C-ABI layer (rt_api.cpp) and the gfx950 kernels.
#pragma once

#include "rt_device.h"
#include "rt_grid.h"

namespace rt {

struct RenderParams {
    Scene scene;
    const float4* __restrict__ rays;  // 2 x float4 per ray (start, direction); null in pinhole mode
    uint64_t n_rays;                  // rays of the whole frame
    uint64_t n_local;                 // work-items of this launch (this rank's share)
    uint64_t tile_rays;               // shard tile length in rays
    uint64_t run_rays;                // consecutive rays of the frame a launch covers per `world` tiles: tile_rays x the ranks it stands for (1: a shard)
    uint32_t rank, world;
    uint32_t pinhole;                 // generate the reference's pinhole grid in-kernel
    uint32_t width;
    float half_w, half_h, height_f, z;
    uint32_t dir_w_zero;              // every primary direction has w == 0 exactly
    uint32_t wf_tile_order;           // wavefront path: work-item t is pixel (8x8 tile t/64, position t%64) instead of pixel t
    uint32_t tile2d;                  // pinhole + row-tile shards: one 8x8-pixel bundle per wave iteration
    uint32_t tile_cull;               // per-bundle screen-rectangle test of every object (scene.bounds holds rectangles)
    uint32_t bundles_x;               // 8-pixel bundle columns per bundle row
    uint32_t local_rows;              // rows this rank renders
    uint32_t tile_rows;               // rows per shard tile
    uint32_t n_bundles;               // 8x8 bundles (tile2d) or 64-ray chunks of this launch
    uint32_t max_bounces;
    void* out;                        // float4 per work-item (kernels 1,2) or float (kernel 0)
    float* aux_t;                     // optional
    int32_t* aux_index;               // optional
    Counters* counters;               // used by counted launches only
};

hipError_t launch_render(const RenderParams& p, int kernel, bool fused, bool count, hipStream_t stream);

// large-N path (rt_wavefront.hip): traversal and shading as separate kernels, pixel state in HBM
struct WavefrontBuffers {
    float* state = nullptr;                        // wavefront_state_bytes(n_local)
    uint32_t* q_closest[2] = {nullptr, nullptr};   // wavefront_queue_bytes(n_local) each
    uint32_t* q_any[2] = {nullptr, nullptr};
    uint32_t* q_slice[2] = {nullptr, nullptr};     // survivors of a shadow slice (ping-pong)
    const HotPair* shadow_pairs = nullptr;         // size-sorted pair stream (owned by the context)
    GridDesc grid = {};                            // conservative grid (owned by the context); enabled = 0 -> brute force
    ScreenTiles tiles = {};                        // per-screen-tile object lists for pinhole primary rays
    LightTiles light_tiles = {};                   // per-direction object lists for the shadow rays of the last light
    BlockGrid blocks = {};                         // the closest-hit walk's coarse grid of 32-byte blocks (owned by the context)
    uint32_t* counts = nullptr;                    // device-side round state (queue lengths, hand-over flags) + run-ticket counters
    uint32_t* h_counts = nullptr;                  // 16 x uint32 pinned host mirror of the round state
    uint64_t capacity = 0;                         // n_local the buffers were sized for
    hipStream_t side_stream = nullptr;             // a round's shadow-ray launch runs here, next to the closest-hit launch
    hipEvent_t ev_fork = nullptr, ev_join = nullptr;
};
size_t wavefront_state_bytes(uint64_t n_local);
size_t wavefront_queue_bytes(uint64_t n_local);
size_t wavefront_counter_bytes();
// Runs a whole frame; the rounds are enqueued without host round trips (device-side round state), `stream` is
// synchronised once per batch of rounds - once per frame in the normal case.
hipError_t launch_wavefront(const RenderParams& p, int kernel, bool fused, bool count, WavefrontBuffers& buf,
                            hipStream_t stream, uint32_t* rounds_out);

}  // namespace rt
