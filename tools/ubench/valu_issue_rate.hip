// VALU issue-rate microbenchmark on gfx950: per-instruction cost with 1..8 waves/SIMD, plus in-kernel clock.
#include <hip/hip_runtime.h>
#include <cstdio>
#include <algorithm>
#define CK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("hip error %d\n", (int)e); return 1; } } while (0)
typedef float f2 __attribute__((ext_vector_type(2)));

#define REP16(S) S(0) S(1) S(2) S(3) S(4) S(5) S(6) S(7) S(8) S(9) S(10) S(11) S(12) S(13) S(14) S(15)

template <int MODE>
__global__ __launch_bounds__(256) void k(float* out, unsigned long long* clk, int iters, float a0, float b0) {
    float x[16];
    f2 y[16];
#pragma unroll
    for (int i = 0; i < 16; ++i) { x[i] = threadIdx.x * 0.001f + i; y[i] = f2{x[i], x[i] + 1.f}; }
    f2 a = {a0, a0 * 1.0001f}, b = {b0, b0 * 0.999f};
    unsigned long long t0 = __builtin_amdgcn_s_memtime(), r0 = __builtin_amdgcn_s_memrealtime();
    for (int it = 0; it < iters; ++it) {
#define S(i) \
        if constexpr (MODE == 0) asm volatile("v_fma_f32 %0, %0, %1, %2" : "+v"(x[i]) : "v"(a0), "v"(b0)); \
        else if constexpr (MODE == 1) asm volatile("v_fma_f32 %0, %1, %0, %2" : "+v"(x[i]) : "s"(a0), "v"(b0)); \
        else if constexpr (MODE == 2) asm volatile("v_fmac_f32_e32 %0, %1, %2" : "+v"(x[i]) : "v"(a0), "v"(b0)); \
        else if constexpr (MODE == 3) asm volatile("v_fmac_f32_e32 %0, %1, %2" : "+v"(x[i]) : "s"(a0), "v"(b0)); \
        else if constexpr (MODE == 4) asm volatile("v_mul_f32_e32 %0, %1, %0" : "+v"(x[i]) : "s"(a0)); \
        else if constexpr (MODE == 5) asm volatile("v_pk_fma_f32 %0, %0, %1, %2" : "+v"(y[i]) : "v"(a), "v"(b)); \
        else if constexpr (MODE == 6) asm volatile("v_add_f32_e32 %0, %1, %0" : "+v"(x[i]) : "v"(b0)); \
        else if constexpr (MODE == 7) asm volatile("v_pk_fma_f32 %0, %1, %0, %2" : "+v"(y[i]) : "s"(a), "v"(b));
        REP16(S)
#undef S
    }
    unsigned long long t1 = __builtin_amdgcn_s_memtime(), r1 = __builtin_amdgcn_s_memrealtime();
    float s = 0;
#pragma unroll
    for (int i = 0; i < 16; ++i) s += x[i] + y[i].x + y[i].y;
    out[blockIdx.x * 256 + threadIdx.x] = s;
    if (threadIdx.x == 0 && blockIdx.x == 0) { clk[0] = t1 - t0; clk[1] = r1 - r0; }
}

template <int MODE>
int run(const char* name, int w, float* d, unsigned long long* dclk) {
    int iters = 100000;
    dim3 grid(256 * w), block(256);
    hipEvent_t e0, e1;
    CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));
    float best = 1e30f;
    for (int rep = 0; rep < 3; ++rep) {
        CK(hipEventRecord(e0));
        k<MODE><<<grid, block>>>(d, dclk, iters, 1.0001f, 0.5f);
        CK(hipEventRecord(e1));
        CK(hipEventSynchronize(e1));
        float ms; CK(hipEventElapsedTime(&ms, e0, e1));
        best = std::min(best, ms);
    }
    unsigned long long clk[2];
    CK(hipMemcpy(clk, dclk, 16, hipMemcpyDeviceToHost));
    double ghz = (double)clk[0] / ((double)clk[1] * 10.0);  // memrealtime ticks at 100 MHz
    double per_simd = (double)iters * 16 * grid.x * 4 / 1024.0;
    double ns = best * 1e6 / per_simd;
    printf("%-22s w/SIMD=%d %8.3f ms  clock %.2f GHz  %.2f cycles per wave-instr per SIMD\n", name, w, best, ghz, ns * ghz);
    return 0;
}

int main() {
    float* d; unsigned long long* dclk;
    CK(hipMalloc(&d, 256 * 8 * 256 * 4 * 4)); CK(hipMalloc(&dclk, 16));
    for (int w : {1, 2, 4, 8}) {
        if (run<0>("v_fma_f32 vvv", w, d, dclk)) return 1;
        run<1>("v_fma_f32 svv", w, d, dclk);
        run<2>("v_fmac_f32_e32 vv", w, d, dclk);
        run<3>("v_fmac_f32_e32 sv", w, d, dclk);
        run<4>("v_mul_f32_e32 sv", w, d, dclk);
        run<6>("v_add_f32_e32 vv", w, d, dclk);
        run<5>("v_pk_fma_f32 vvv", w, d, dclk);
        run<7>("v_pk_fma_f32 svv", w, d, dclk);
    }
    return 0;
}
