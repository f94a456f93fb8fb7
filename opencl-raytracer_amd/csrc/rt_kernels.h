// rt_kernels.h - launch interface between the C-ABI layer (rt_api.cpp) and the gfx950 kernels.
#pragma once

#include "rt_device.h"

namespace rt {

struct RenderParams {
    Scene scene;
    const float4* __restrict__ rays;  // 2 x float4 per ray (start, direction); null in pinhole mode
    uint64_t n_rays;                  // rays of the whole frame
    uint64_t n_local;                 // work-items of this launch (this rank's share)
    uint64_t tile_rays;               // shard tile length in rays
    uint32_t rank, world;
    uint32_t pinhole;                 // generate the reference's pinhole grid in-kernel
    uint32_t width;
    float half_w, half_h, height_f, z;
    uint32_t dir_w_zero;              // every primary direction has w == 0 exactly
    uint32_t max_bounces;
    void* out;                        // float4 per work-item (kernels 1,2) or float (kernel 0)
    float* aux_t;                     // optional
    int32_t* aux_index;               // optional
    Counters* counters;               // used by counted launches only
};

hipError_t launch_render(const RenderParams& p, int kernel, bool fused, bool count, hipStream_t stream);

}  // namespace rt
