// rt_kernels.hip - gfx950 kernels of the IRaytracer hot path and their launcher.
//
// render_pixels<KERNEL, FUSED, COUNT>: one work-item per ray, 256-thread workgroups (4 wave64). Replaces the
// reference's three NDRange kernels (hittest_kernel.cl:54, shade_kernel.cl:180,
// shade_and_reflect_kernel.cl:244; launched 1-D with local size 32 at OpenCLRaytracer.cpp:89-91).
// The whole bounce loop of a pixel stays in registers; object records arrive through scalar loads
// (wave-uniform index), every work-item stores exactly one 16-byte pixel, coalesced (1 KiB per wave).
#include "rt_kernels.h"

namespace rt {

// wave64 sum of a 64-bit counter
__device__ __forceinline__ unsigned long long wave_sum(unsigned long long v) {
#pragma unroll
    for (int off = 32; off > 0; off >>= 1) v += __shfl_xor(v, off, 64);
    return v;
}

template <int KERNEL, bool FUSED, bool COUNT>
__global__ __launch_bounds__(256) void render_pixels(const RenderParams p) {
    const uint64_t local = (uint64_t)blockIdx.x * 256u + threadIdx.x;
    const bool active = local < p.n_local;

    // ---- work-item -> global ray index (interleaved tile ownership for multi-GPU shards) ----
    uint64_t g = local;
    if (p.world > 1u) {
        const uint64_t tile = local / p.tile_rays;
        const uint64_t off = local - tile * p.tile_rays;
        g = (tile * p.world + p.rank) * p.tile_rays + off;
    }
    const bool valid = active && g < p.n_rays;

    Counters ctr = {0ull, 0ull, 0ull};
    float outr = 0.f, outg = 0.f, outb = 0.f;
    float T = kMaxFloat;
    int idx = -1;
    bool hit = false;

    if (valid) {
        Ray ray;
        if (p.pinhole) {
            // main()'s ray loop (OpenCL-Raytracer.cpp:18-26,68-72): exact in fp32 (small half-integers)
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

        const Scene& S = p.scene;
        if (p.dir_w_zero) closest_hit<FUSED, true>(S.pairs, S.n_pairs, ray, T, idx);
        else closest_hit<FUSED, false>(S.pairs, S.n_pairs, ray, T, idx);
        if constexpr (COUNT) { ctr.traced += 1; ctr.reference += 1; }

        // raycast()'s return value: shade_and_reflect_kernel.cl:173 vs shade_kernel.cl:167 / hittest_kernel.cl:149
        hit = (KERNEL == 2) ? !(T == kMaxFloat) : (T < kMaxFloat);
        if constexpr (COUNT) ctr.hits += hit ? 1 : 0;

        if constexpr (KERNEL != 0) {
            if (hit) {
                HitRec h;
                materialise<FUSED>(S.hot, S.cold, idx, T, ray, h);
                if constexpr (KERNEL == 1) shade_forward<FUSED, true, COUNT>(S, h, outr, outg, outb, ctr);
                else shade_and_reflect_pixel<FUSED, COUNT>(S, p.max_bounces, h, outr, outg, outb, ctr);
            }
        }
    }

    if (active) {
        if constexpr (KERNEL == 0) {
            reinterpret_cast<float*>(p.out)[local] = hit ? T : kMaxFloat;
        } else {
            // background = the reference's upload-time pixel {0,0,0,1} (OpenCLRaytracer.cpp:32, Q11)
            reinterpret_cast<float4*>(p.out)[local] = make_float4(outr, outg, outb, 1.0f);
        }
        if (p.aux_t) p.aux_t[local] = T;
        if (p.aux_index) p.aux_index[local] = hit ? idx : -1;
    }

    if constexpr (COUNT) {
        const unsigned long long a = wave_sum(ctr.traced), b = wave_sum(ctr.reference), c = wave_sum(ctr.hits);
        if ((threadIdx.x & 63u) == 0u) {
            atomicAdd(&p.counters->traced, a);
            atomicAdd(&p.counters->reference, b);
            atomicAdd(&p.counters->hits, c);
        }
    }
}

template <int KERNEL, bool FUSED>
static hipError_t launch2(const RenderParams& p, bool count, hipStream_t stream) {
    const uint64_t blocks64 = (p.n_local + 255u) / 256u;
    if (blocks64 == 0) return hipSuccess;
    if (blocks64 > 0x7fffffffull) return hipErrorInvalidValue;
    const dim3 grid((uint32_t)blocks64), block(256);
    if (count) hipLaunchKernelGGL((render_pixels<KERNEL, FUSED, true>), grid, block, 0, stream, p);
    else hipLaunchKernelGGL((render_pixels<KERNEL, FUSED, false>), grid, block, 0, stream, p);
    return hipGetLastError();
}

hipError_t launch_render(const RenderParams& p, int kernel, bool fused, bool count, hipStream_t stream) {
    switch (kernel) {
        case 0: return fused ? launch2<0, true>(p, count, stream) : launch2<0, false>(p, count, stream);
        case 1: return fused ? launch2<1, true>(p, count, stream) : launch2<1, false>(p, count, stream);
        case 2: return fused ? launch2<2, true>(p, count, stream) : launch2<2, false>(p, count, stream);
        default: return hipErrorInvalidValue;
    }
}

}  // namespace rt
