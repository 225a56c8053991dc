// Do the matrix pipe and the vector ALU of one SIMD overlap ACROSS waves?  One workgroup of 8 waves (two per SIMD): waves 0-3
// run NM independent 16x16x32 bf16 MFMAs per iteration (4 accumulators, round robin), waves 4-7 run NV VALU operations per
// iteration (4 independent chains of v_med3 / v_cvt_pk / v_fma -- the epilogue mix).  Three runs: MFMA waves alone, VALU waves
// alone, both.  If the pipes overlap, "both" costs max(alone); if a SIMD issues one or the other, it costs the sum.
// build: hipcc --offload-arch=gfx950 -O3 -o tools/ubench/xwave_overlap tools/ubench/xwave_overlap.hip
#include <hip/hip_runtime.h>
#include <cstdio>
using f32x4 = __attribute__((ext_vector_type(4))) float;
using bf16x8 = __attribute__((ext_vector_type(8))) __bf16;

template <int MODE, int OP>   // MODE 1: MFMA waves work, 2: VALU waves work, 3: both; OP: -1 = the mix, 0..5 = one instruction kind
__global__ __launch_bounds__(512) void k(float* out, unsigned long long* cyc, int iters) {
    const int wave = threadIdx.x >> 6;
    f32x4 acc[4] = {{0, 0, 0, 0}, {0, 0, 0, 0}, {0, 0, 0, 0}, {0, 0, 0, 0}};
    bf16x8 a, b;
    for (int i = 0; i < 8; ++i) { a[i] = (__bf16)(threadIdx.x * 0.001f + i); b[i] = (__bf16)(i * 0.5f); }
    float v[4] = {threadIdx.x * 0.5f, 1.f, 2.f, 3.f};
    unsigned p[4] = {0, 0, 0, 0};
    unsigned long long t0, t1;
    __syncthreads();
    asm volatile("s_memtime %0\n\ts_waitcnt lgkmcnt(0)" : "=s"(t0)::"memory");
    if (wave < 4) {
        if (MODE & 1)
            for (int it = 0; it < iters; ++it) {
#pragma unroll
                for (int j = 0; j < 16; ++j) acc[j & 3] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(a, b, acc[j & 3], 0, 0, 0);
            }
    } else {
        if (MODE & 2)
            for (int it = 0; it < iters; ++it) {
#pragma unroll
                for (int j = 0; j < 16; ++j) {
                    const int c = j & 3;
                    const int kind = OP < 0 ? (j >> 2) : OP;
                    if (kind == 0) asm volatile("v_med3_f32 %0, %0, 0, %1" : "+v"(v[c]) : "v"(6.0f));
                    if (kind == 1) asm volatile("v_cvt_pk_f16_f32 %0, %1, %1" : "=v"(p[c]) : "v"(v[c]));
                    if (kind == 2) asm volatile("v_fma_f32 %0, %0, %1, %0" : "+v"(v[c]) : "v"(1.0001f));
                    if (kind == 3) asm volatile("v_pk_add_f16 %0, %0, %1" : "+v"(p[c]) : "v"(p[(c + 1) & 3]));
                    if (kind == 4) asm volatile("v_cvt_pk_bf16_f32 %0, %1, %1" : "=v"(p[c]) : "v"(v[c]));
                    if (kind == 5) asm volatile("v_add_f32 %0, %0, %1" : "+v"(v[c]) : "v"(1.5f));
                    if (kind == 6) asm volatile("v_mov_b32 %0, %1" : "=v"(p[c]) : "v"(p[(c + 1) & 3]));
                    if (kind == 7) asm volatile("s_nop 0\n\tv_mov_b32_dpp %0, %1 wave_shl:1 row_mask:0xf bank_mask:0xf bound_ctrl:1" : "=v"(p[c]) : "v"(p[(c + 1) & 3]));
                }
            }
    }
    asm volatile("s_memtime %0\n\ts_waitcnt lgkmcnt(0)" : "=s"(t1)::"memory");
    out[threadIdx.x] = acc[0][0] + acc[1][1] + acc[2][2] + acc[3][3] + v[0] + v[1] + v[2] + v[3] + (float)(p[0] ^ p[1] ^ p[2] ^ p[3]);
    if ((threadIdx.x & 63) == 0) cyc[wave] = t1 - t0;
}
template <int MODE, int OP>
static void run(const char* name, float* out, unsigned long long* cyc) {
    const int iters = 4096;
    k<MODE, OP><<<1, 512>>>(out, cyc, iters);
    hipDeviceSynchronize();
    k<MODE, OP><<<1, 512>>>(out, cyc, iters);
    hipDeviceSynchronize();
    unsigned long long h[8];
    hipMemcpy(h, cyc, sizeof h, hipMemcpyDeviceToHost);
    printf("%-28s MFMA wave 0: %6.2f cycles per MFMA   VALU wave 4: %6.2f cycles per VALU op\n", name, (double)h[0] / (iters * 16.0),
           (double)h[4] / (iters * 16.0));
}
int main() {
    float* out; unsigned long long* cyc;
    hipMalloc(&out, 4096); hipMalloc(&cyc, 64);
    run<1, -1>("MFMA waves alone", out, cyc);
    run<2, -1>("VALU waves alone (mix)", out, cyc);
    run<3, -1>("both (mix)", out, cyc);
    run<2, 0>("v_med3_f32 alone", out, cyc);       run<3, 0>("v_med3_f32 + MFMA", out, cyc);
    run<2, 1>("v_cvt_pk_f16_f32 alone", out, cyc); run<3, 1>("v_cvt_pk_f16_f32 + MFMA", out, cyc);
    run<2, 4>("v_cvt_pk_bf16_f32 alone", out, cyc); run<3, 4>("v_cvt_pk_bf16_f32 + MFMA", out, cyc);
    run<2, 2>("v_fma_f32 alone", out, cyc);        run<3, 2>("v_fma_f32 + MFMA", out, cyc);
    run<2, 5>("v_add_f32 alone", out, cyc);        run<3, 5>("v_add_f32 + MFMA", out, cyc);
    run<2, 3>("v_pk_add_f16 alone", out, cyc);     run<3, 3>("v_pk_add_f16 + MFMA", out, cyc);
    run<2, 6>("v_mov_b32 alone", out, cyc);        run<3, 6>("v_mov_b32 + MFMA", out, cyc);
    run<2, 7>("v_mov_b32_dpp alone", out, cyc);    run<3, 7>("v_mov_b32_dpp + MFMA", out, cyc);
    return 0;
}
