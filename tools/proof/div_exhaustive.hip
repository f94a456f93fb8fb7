// Exhaustive check, on the GPU, of shared-reciprocal division sequences against IEEE-754 division (fp32, round to nearest
// even): for EVERY pair of significands (a, b in [1, 2): 2^23 x 2^23 = 2^46 pairs) is
//     y  = rcp(b) refined by one Newton step            (shared by all numerators of one denominator)
//     q0 = a * y;  r = fma(-q0, b, a);  q = fma(r, y, q0)
// bit-identical to a / b?  Exponents do not matter as long as nothing overflows, underflows or goes subnormal on the way
// (the sequences' operations are all exact scalings of the [1, 2) case), which the callers guard; signs are symmetric.
// Build: hipcc -O3 --offload-arch=gfx950 -fhip-fp32-correctly-rounded-divide-sqrt -ffp-contract=off tools/proof/div_exhaustive.hip -o tools/proof/div_exhaustive
// Run:   ./div_exhaustive [log2 of the number of denominators to test, default 23 = all]
#include <hip/hip_runtime.h>
#include <cstdint>
#include <cstdio>
#include <cstdlib>

struct Tally {
    unsigned long long bad_one_step;      // y after ONE Newton step, one correction of q
    unsigned long long bad_two_corr;      // ... two corrections of q
    unsigned long long bad_recip;         // y != RN(1 / b)
    unsigned long long pairs;
    uint32_t first_a, first_b;
};

__device__ __forceinline__ float rcp_hw(float x) { return __builtin_amdgcn_rcpf(x); }

__global__ __launch_bounds__(256) void check(uint32_t b_first, uint32_t b_count, Tally* out) {
    const uint32_t gid = blockIdx.x * 256u + threadIdx.x;
    const uint32_t n_threads = gridDim.x * 256u;
    unsigned long long bad1 = 0, bad2 = 0, badr = 0, pairs = 0;
    uint32_t fa = 0, fb = 0;
    // a thread owns numerators gid, gid + n_threads, ... for every denominator of the launch
    for (uint32_t bi = 0; bi < b_count; ++bi) {
        const uint32_t mb = b_first + bi;
        const float b = __uint_as_float(0x3f800000u | mb);
        const float y0 = rcp_hw(b);
        const float e = __builtin_fmaf(-b, y0, 1.0f);
        const float y = __builtin_fmaf(e, y0, y0);
        if (gid == 0 && y != 1.0f / b) ++badr;
        for (uint32_t ma = gid; ma < (1u << 23); ma += n_threads) {
            const float a = __uint_as_float(0x3f800000u | ma);
            const float want = a / b;
            const float q0 = a * y;
            const float r0 = __builtin_fmaf(-q0, b, a);
            const float q1 = __builtin_fmaf(r0, y, q0);
            const float r1 = __builtin_fmaf(-q1, b, a);
            const float q2 = __builtin_fmaf(r1, y, q1);
            ++pairs;
            if (__float_as_uint(q1) != __float_as_uint(want)) { if (!bad1) { fa = ma; fb = mb; } ++bad1; }
            if (__float_as_uint(q2) != __float_as_uint(want)) ++bad2;
        }
    }
    if (bad1) { atomicAdd(&out->bad_one_step, bad1); out->first_a = fa; out->first_b = fb; }
    if (bad2) atomicAdd(&out->bad_two_corr, bad2);
    if (badr) atomicAdd(&out->bad_recip, badr);
    atomicAdd(&out->pairs, pairs);
}

int main(int argc, char** argv) {
    const int lg = argc > 1 ? atoi(argv[1]) : 23;
    const uint32_t n_b = 1u << lg;
    const uint32_t stride = (1u << 23) / n_b;  // a subset: evenly spaced denominators, plus the all-ones significand
    Tally* d;
    hipMalloc(&d, sizeof(Tally));
    hipMemset(d, 0, sizeof(Tally));
    const uint32_t chunk = 4096;  // denominators per launch (a launch = 2^35 pairs when all numerators are tested)
    const dim3 grid(256 * 16), block(256);
    unsigned long long launched = 0;
    if (stride == 1) {
        for (uint32_t b0 = 0; b0 < (1u << 23); b0 += chunk) {
            hipLaunchKernelGGL(check, grid, block, 0, 0, b0, chunk, d);
            if (((b0 / chunk) & 63u) == 63u) {
                hipDeviceSynchronize();
                Tally h;
                hipMemcpy(&h, d, sizeof(h), hipMemcpyDeviceToHost);
                printf("denominators %u / %u: pairs %llu  mismatches: one correction %llu, two corrections %llu, reciprocal %llu\n",
                       b0 + chunk, 1u << 23, h.pairs, h.bad_one_step, h.bad_two_corr, h.bad_recip);
                fflush(stdout);
            }
            launched += chunk;
        }
    } else {
        for (uint32_t k = 0; k < n_b; ++k) hipLaunchKernelGGL(check, grid, block, 0, 0, k * stride, 1u, d);
        hipLaunchKernelGGL(check, grid, block, 0, 0, (1u << 23) - 1u, 1u, d);
    }
    hipDeviceSynchronize();
    Tally h;
    hipMemcpy(&h, d, sizeof(h), hipMemcpyDeviceToHost);
    printf("TOTAL pairs %llu (2^46 = %llu)  mismatches: one correction %llu (first a %06x b %06x), two corrections %llu, reciprocal not RN %llu\n",
           h.pairs, 1ull << 46, h.bad_one_step, h.first_a, h.first_b, h.bad_two_corr, h.bad_recip);
    return 0;
}
