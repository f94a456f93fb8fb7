// Issue cost of the instruction FORMS the grid walk is made of, on gfx950: cycles per wave-instruction per SIMD
// at 1..7 waves/SIMD. Forms: all-VGPR VALU, SGPR / literal / inline-constant sources, compares that write VCC or an
// SGPR pair, selects that read an SGPR-pair mask, compare + select pairs, three-source and packed forms.
// (Modes 4 and 10-12 of the switch are not run: a select on a VCC nobody wrote and the SALU-only loops did not measure
// what they were meant to.)
#include <hip/hip_runtime.h>
#include <cstdio>
#include <algorithm>
#define CK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("hip error %d at %d\n", (int)e, __LINE__); return 1; } } while (0)
#define REP16(S) S(0) S(1) S(2) S(3) S(4) S(5) S(6) S(7) S(8) S(9) S(10) S(11) S(12) S(13) S(14) S(15)

template <int MODE>
__global__ __launch_bounds__(256) void k(float* out, unsigned long long* clk, int iters, float a0, float b0) {
    float x[16];
#pragma unroll
    for (int i = 0; i < 16; ++i) x[i] = threadIdx.x * 0.001f + i;
    unsigned long long m0 = 0x5555555555555555ull, m1 = 0x3333333333333333ull;
    unsigned long long t0 = __builtin_amdgcn_s_memtime(), r0 = __builtin_amdgcn_s_memrealtime();
    for (int it = 0; it < iters; ++it) {
#define S(i) \
        if constexpr (MODE == 0) asm volatile("v_fma_f32 %0, %0, %1, %2" : "+v"(x[i]) : "v"(a0), "v"(b0)); \
        else if constexpr (MODE == 1) asm volatile("v_fma_f32 %0, %1, %0, %2" : "+v"(x[i]) : "s"(a0), "v"(b0)); \
        else if constexpr (MODE == 2) asm volatile("v_add_f32_e32 %0, 1.0, %0" : "+v"(x[i])); \
        else if constexpr (MODE == 3) asm volatile("v_add_f32_e32 %0, 0x3727c5ac, %0" : "+v"(x[i])); \
        else if constexpr (MODE == 4) asm volatile("v_cndmask_b32_e32 %0, %0, %1, vcc" : "+v"(x[i]) : "v"(a0) : ); \
        else if constexpr (MODE == 5) asm volatile("v_cndmask_b32_e64 %0, %0, %1, %2" : "+v"(x[i]) : "v"(a0), "s"(m0)); \
        else if constexpr (MODE == 6) asm volatile("v_cmp_lt_f32_e32 vcc, %0, %1" : : "v"(x[i]), "v"(a0) : "vcc"); \
        else if constexpr (MODE == 7) asm volatile("v_cmp_lt_f32_e64 %0, %1, %2" : "=s"(m1) : "v"(x[i]), "v"(a0)); \
        else if constexpr (MODE == 8) asm volatile("v_add_u32_e32 %0, %0, %1" : "+v"(x[i]) : "v"(a0)); \
        else if constexpr (MODE == 9) asm volatile("v_min3_f32 %0, %0, %1, %2" : "+v"(x[i]) : "v"(a0), "v"(b0)); \
        else if constexpr (MODE == 10) asm volatile("v_fma_f32 %0, %0, %2, %3\n s_and_b64 %1, %1, %4" : "+v"(x[i]), "+s"(m0) : "v"(a0), "v"(b0), "s"(m1)); \
        else if constexpr (MODE == 11) asm volatile("v_fma_f32 %0, %0, %2, %3\n s_and_b64 %1, %1, %4\n s_or_b64 %1, %1, %4" : "+v"(x[i]), "+s"(m0) : "v"(a0), "v"(b0), "s"(m1)); \
        else if constexpr (MODE == 12) asm volatile("s_and_b64 %0, %0, %1" : "+s"(m0) : "s"(m1)); \
        else if constexpr (MODE == 13) asm volatile("v_cmp_lt_f32_e32 vcc, %0, %1\n v_cndmask_b32_e32 %0, %0, %1, vcc" : "+v"(x[i]) : "v"(a0) : "vcc"); \
        else if constexpr (MODE == 14) asm volatile("v_mul_f32_e64 %0, %0, -%1" : "+v"(x[i]) : "v"(a0)); \
        else if constexpr (MODE == 15) asm volatile("v_pk_add_f32 %0, %0, %1" : "+v"(*(double*)&x[i & 14]) : "v"(*(double*)&x[(i + 2) & 14]));
        REP16(S)
#undef S
    }
    unsigned long long t1 = __builtin_amdgcn_s_memtime(), r1 = __builtin_amdgcn_s_memrealtime();
    float s = (float)(m0 & 1) + (float)(m1 & 1);
#pragma unroll
    for (int i = 0; i < 16; ++i) s += x[i];
    out[blockIdx.x * 256 + threadIdx.x] = s;
    if (threadIdx.x == 0 && blockIdx.x == 0) { clk[0] = t1 - t0; clk[1] = r1 - r0; }
}

template <int MODE>
int run(const char* name, int per_rep, int w, float* d, unsigned long long* dclk) {
    int iters = 20000;
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
    double groups = (double)iters * 16 * grid.x * 4 / 1024.0;   // asm groups per SIMD
    double ns = best * 1e6 / groups;
    printf("%-40s w/SIMD=%d %8.3f ms  clock %.2f GHz  %6.2f cycles per group (%d instr)\n", name, w, best, ghz, ns * ghz, per_rep);
    return 0;
}

int main() {
    float* d; unsigned long long* dclk;
    CK(hipMalloc(&d, 256 * 8 * 256 * 4 * 4)); CK(hipMalloc(&dclk, 16));
    for (int w : {1, 2, 4, 7}) {
        if (run<0>("v_fma_f32 v,v,v", 1, w, d, dclk)) return 1;
        run<1>("v_fma_f32 s,v,v", 1, w, d, dclk);
        run<2>("v_add_f32 inline-const", 1, w, d, dclk);
        run<3>("v_add_f32 literal", 1, w, d, dclk);
        run<5>("v_cndmask e64 (reads sgpr pair)", 1, w, d, dclk);
        run<6>("v_cmp e32 (writes vcc)", 1, w, d, dclk);
        run<7>("v_cmp e64 (writes sgpr pair)", 1, w, d, dclk);
        run<8>("v_add_u32 v,v", 1, w, d, dclk);
        run<9>("v_min3_f32 v,v,v", 1, w, d, dclk);
        run<13>("v_cmp vcc + v_cndmask vcc", 2, w, d, dclk);
        run<14>("v_mul_f32 e64 neg modifier", 1, w, d, dclk);
        run<15>("v_pk_add_f32 v,v", 1, w, d, dclk);
    }
    return 0;
}
