// rt_kernels.hip - gfx950 kernels of the IRaytracer hot path and their launcher.
//
// render_pixels<KERNEL, FUSED, COUNT>: one work-item per ray, 256-thread workgroups (4 wave64). Replaces the
// reference's three NDRange kernels (hittest_kernel.cl:54, shade_kernel.cl:180,
// shade_and_reflect_kernel.cl:244; launched 1-D with local size 32 at OpenCLRaytracer.cpp:89-91).
// The whole bounce loop of a pixel stays in registers; object records arrive through scalar loads
// (wave-uniform index), every work-item stores exactly one 16-byte pixel. This is the small-scene path
// (a few objects: shading dominates); scenes with hundreds of objects and more go through rt_wavefront.hip.
// For the reference's pinhole grid the work is laid out in 16x16-pixel workgroups / 8x8-pixel waves and each
// wave first asks which objects its 64 rays can possibly hit (bundle_candidates).
#include "rt_kernels.h"

namespace rt {

// wave64 sum of a 64-bit counter
__device__ __forceinline__ unsigned long long wave_sum(unsigned long long v) {
#pragma unroll
    for (int off = 32; off > 0; off >>= 1) v += __shfl_xor(v, off, 64);
    return v;
}

// Which objects can any ray of this wave's 8x8 pixel bundle hit? Lane k compares the bundle's direction
// rectangle [xl,xr] x [yb,yt] (pinhole rays: direction = (x, y, z), z fixed) with object k's screen rectangle,
// i.e. the conservative projection of its inflated bounding sphere computed on the host in double precision
// (rt_api.cpp: screen_rect). Four compares per object per wave.
__device__ __forceinline__ uint64_t bundle_candidates(const float4* __restrict__ rects, uint32_t n, float xl, float xr,
                                                      float yb, float yt) {
    const uint32_t lane = threadIdx.x & 63u;
    bool keep = false;
    if (lane < n) {
        const float4 r = rects[lane];  // (xmin, xmax, ymin, ymax)
        keep = (xr >= r.x) && (xl <= r.y) && (yt >= r.z) && (yb <= r.w);
    }
    return __ballot(keep);
}

// render_pixels: one wave = one 8x8-pixel bundle (pinhole grids; a workgroup is a 16x16 block) or 64 consecutive
// rays (ray lists). Bundles cost anything from ~30 instructions (nothing to hit) to ~1000 per pixel (hit,
// shadow ray, reflection), so the distribution is left to the hardware workgroup dispatcher: measured here, a
// persistent grid with static striding loses 25 % to imbalance, and one drawing bundles from an atomic counter
// saturates that counter at ~88 tickets/us (3 ms for a 4096^2 frame).
template <int KERNEL, bool FUSED, bool COUNT>
__global__ __launch_bounds__(256) void render_pixels(const RenderParams p) {
    Counters ctr = {};
    const uint32_t lane = threadIdx.x & 63u;
    const Scene& S = p.scene;
    {
        const uint32_t wave = threadIdx.x >> 6;
        // ---- work-item -> (local slot, global ray) ----
        uint32_t local, col = 0, row = 0;
        uint64_t g;
        bool active;
        bool use_mask = false;
        uint64_t mask = 0ull;
        if (p.tile2d) {
            const uint32_t blocks_x = (p.bundles_x + 1u) >> 1;
            const uint32_t wy = blockIdx.x / blocks_x;
            const uint32_t by = wy * 2u + (wave >> 1);
            const uint32_t bx = (blockIdx.x - wy * blocks_x) * 2u + (wave & 1u);
            col = bx * 8u + (lane & 7u);
            const uint32_t lrow = by * 8u + (lane >> 3);
            active = col < p.width && lrow < p.local_rows;
            local = lrow * p.width + col;
            uint32_t r0 = by * 8u;  // first row of the bundle, mapped to the global grid (shard tiles are 8-row aligned)
            if (p.world > 1u) {
                const uint32_t t = r0 / p.tile_rows;
                r0 = (t * p.world + p.rank) * p.tile_rows + (r0 - t * p.tile_rows);
            }
            row = r0 + (lane >> 3);
            g = (uint64_t)row * p.width + col;
            if (p.tile_cull) {
                const float xl = (float)(bx * 8u) - p.half_w, xr = xl + 7.0f;
                const float yt = (p.height_f - (float)r0) - p.half_h, yb = yt - 7.0f;
                mask = bundle_candidates(S.bounds, S.n_objs, xl, xr, yb, yt);
                use_mask = true;
            }
        } else {
            local = blockIdx.x * 256u + threadIdx.x;
            active = local < p.n_local;
            g = local;
            if (p.world > 1u) {
                const uint64_t run = (uint64_t)local / p.run_rays;
                const uint64_t off = (uint64_t)local - run * p.run_rays;
                g = (run * p.world + p.rank) * p.tile_rays + off;
            }
            if (p.pinhole) {
                const uint32_t gi = (uint32_t)g;
                row = gi / p.width;
                col = gi - row * p.width;
            }
        }
        const bool valid = active && g < p.n_rays;

        float outr = 0.f, outg = 0.f, outb = 0.f;
        float T = kMaxFloat;
        int idx = -1;
        bool hit = false;

        if (valid && !(use_mask && mask == 0ull)) {
            Ray ray;
            if (p.pinhole) {
                // main()'s ray loop (OpenCL-Raytracer.cpp:18-26,68-72): exact in fp32 (small half-integers)
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

            if (p.dir_w_zero) closest_hit_small<FUSED, true>(S.hot, S.n_objs, use_mask, mask, ray, T, idx);
            else closest_hit_small<FUSED, false>(S.hot, S.n_objs, use_mask, mask, ray, T, idx);

            // raycast()'s return value: shade_and_reflect_kernel.cl:173 vs shade_kernel.cl:167 / hittest_kernel.cl:149
            hit = (KERNEL == 2) ? !(T == kMaxFloat) : (T < kMaxFloat);

            if constexpr (KERNEL != 0) {
                if (hit) {
                    HitRec h;
                    materialise<FUSED>(S.objrec, S.cold, idx, T, ray, h, S.affine != 0u);
                    if constexpr (KERNEL == 1) shade_forward<FUSED, true, COUNT>(S, h, outr, outg, outb, ctr);
                    else shade_and_reflect_pixel<FUSED, COUNT>(S, p.max_bounces, h, outr, outg, outb, ctr);
                }
            }
        }
        if constexpr (COUNT) {
            if (valid) { ctr.traced += 1; ctr.reference += 1; ctr.hits += hit ? 1 : 0; }
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
    }

    if constexpr (COUNT) {
        const unsigned long long a = wave_sum(ctr.traced), b = wave_sum(ctr.reference), c = wave_sum(ctr.hits);
        if (lane == 0u) {
            atomicAdd(&p.counters->traced, a);
            atomicAdd(&p.counters->reference, b);
            atomicAdd(&p.counters->hits, c);
        }
    }
}

template <int KERNEL, bool FUSED>
static hipError_t launch2(const RenderParams& p, bool count, hipStream_t stream) {
    if (p.n_bundles == 0) return hipSuccess;
    uint32_t blocks;
    if (p.tile2d) blocks = ((p.bundles_x + 1u) / 2u) * ((((p.local_rows + 7u) / 8u) + 1u) / 2u);  // 2x2 bundles each
    else blocks = (p.n_bundles + 3u) / 4u;
    const dim3 grid(blocks), block(256);
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
