// What does one DEPENDENT random record fetch cost a CU on gfx950? The grid walk's trip is exactly that: every lane of
// every resident wave follows its own chain of 16..128-byte records scattered over a table of some tens of MB.
// Measured: chip-wide lane-fetches per second and CU cycles per lane-fetch, as a function of
//   LOADS  dwordx4 loads issued per record (1, 2, 4, 8 = 16 .. 128 bytes of one 128-byte block)
//   SPLIT  0: all loads of a fetch go to ONE 128-byte block; 1: every load to a block of its own (same count of
//          instructions, LOADS times the lines)
//   table size (2 MB: L2-resident, 8 / 32 / 128 MB: beyond an XCD's 4 MB L2, inside the 256 MB Infinity Cache)
//   waves per SIMD (4, 6, 8)
// Build / run on an MI355X:  hipcc --offload-arch=gfx950 -O3 gather_rate.hip -o gather_rate && ./gather_rate
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#include <vector>
#define CK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("hip error %d (%s) at %d\n", (int)e, hipGetErrorString(e), __LINE__); return 1; } } while (0)

// table: n_blocks blocks of 128 bytes = 8 x uint4; word 0 of every 16-byte piece = a random block index (the chain)
template <int LOADS, int SPLIT>
__global__ __launch_bounds__(256) void chase(const uint4* __restrict__ table, uint32_t mask, int iters, uint32_t* __restrict__ out) {
    uint32_t idx = (blockIdx.x * 256u + threadIdx.x) * 2654435761u;
    uint32_t acc = 0;
    for (int it = 0; it < iters; ++it) {
        uint4 v[LOADS];
#pragma unroll
        for (int l = 0; l < LOADS; ++l) {
            const uint32_t b = SPLIT ? ((idx + (uint32_t)l * 0x9e3779b9u) & mask) : (idx & mask);
            v[l] = table[(size_t)b * 8u + (SPLIT ? 0u : (uint32_t)l)];
        }
        uint32_t nx = 0;
#pragma unroll
        for (int l = 0; l < LOADS; ++l) { nx ^= v[l].x; acc += v[l].y ^ v[l].w; }
        idx = nx + (uint32_t)it;
    }
    out[blockIdx.x * 256u + threadIdx.x] = acc + idx;
}

template <int LOADS, int SPLIT>
int run(const uint4* d_table, uint32_t n_blocks, int waves_per_simd, uint32_t* d_out, double table_mb) {
    const int iters = 400;
    dim3 grid(256 * waves_per_simd), block(256);  // 256 CUs x 4 SIMDs x waves / 4 waves per workgroup
    hipEvent_t e0, e1;
    CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));
    float best = 1e30f;
    for (int rep = 0; rep < 3; ++rep) {
        CK(hipEventRecord(e0));
        chase<LOADS, SPLIT><<<grid, block>>>(d_table, n_blocks - 1u, iters, d_out);
        CK(hipEventRecord(e1));
        CK(hipEventSynchronize(e1));
        float ms;
        CK(hipEventElapsedTime(&ms, e0, e1));
        if (ms < best) best = ms;
    }
    const double fetches = (double)grid.x * 256.0 * iters;
    const double per_s = fetches / (best * 1e-3);
    const double cyc = 256.0 * 2.4e9 / per_s;  // CU cycles per lane-fetch
    printf("table %6.0f MB  waves/SIMD %d  loads/fetch %d (%s)  %7.3f ms  %7.1f G lane-fetches/s  %6.2f CU-cycles per lane-fetch  %6.2f per load  latency/fetch %.0f ns\n",
           table_mb, waves_per_simd, LOADS, SPLIT ? "own line each" : "one 128-B block", best, per_s / 1e9, cyc, cyc / LOADS,
           best * 1e6 / iters);
    return 0;
}

int main() {
    const size_t max_blocks = (size_t)1 << 20;  // 128 MB
    std::vector<uint4> h(max_blocks * 8);
    uint64_t s = 88172645463325252ull;
    for (size_t i = 0; i < h.size(); ++i) {
        s ^= s << 13; s ^= s >> 7; s ^= s << 17;
        h[i] = make_uint4((uint32_t)(s >> 11), (uint32_t)s, 0u, (uint32_t)(s >> 40));
    }
    uint4* d_table;
    uint32_t* d_out;
    CK(hipMalloc((void**)&d_table, h.size() * sizeof(uint4)));
    CK(hipMalloc((void**)&d_out, 256 * 8 * 256 * sizeof(uint32_t)));
    CK(hipMemcpy(d_table, h.data(), h.size() * sizeof(uint4), hipMemcpyHostToDevice));
    for (uint32_t lg : {14u, 16u, 18u, 20u}) {
        const uint32_t nb = 1u << lg;
        const double mb = nb * 128.0 / 1048576.0;
        for (int w : {4, 6, 8}) {
            if (run<1, 0>(d_table, nb, w, d_out, mb)) return 1;
            if (run<2, 0>(d_table, nb, w, d_out, mb)) return 1;
            if (run<4, 0>(d_table, nb, w, d_out, mb)) return 1;
            if (run<8, 0>(d_table, nb, w, d_out, mb)) return 1;
            if (run<2, 1>(d_table, nb, w, d_out, mb)) return 1;
            if (run<4, 1>(d_table, nb, w, d_out, mb)) return 1;
        }
    }
    return 0;
}
