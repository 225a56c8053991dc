// What does a DPP lane shift cost on gfx950?  One wave runs a dependent chain of N operations of one kind and reads
// s_memtime before and after (cycles per op); a second variant runs two waves per SIMD.
//   v_add_f32 (plain) | v_mov_b32_dpp row_shl:1 | v_mov_b32_dpp wave_shl:1 | v_add_f32_dpp wave_shl:1 | v_med3 + s_nop 1 + dpp wave_shl
// build: hipcc --offload-arch=gfx950 -O3 -o tools/ubench/dpp_cost tools/ubench/dpp_cost.hip
#include <hip/hip_runtime.h>
#include <cstdio>

template <int KIND>
__global__ void k(float* out, unsigned long long* cyc, int n) {
    float v = threadIdx.x * 0.5f, w = 1.0f;
    unsigned long long t0, t1;
    asm volatile("s_memtime %0\n\ts_waitcnt lgkmcnt(0)" : "=s"(t0)::"memory");
    for (int i = 0; i < n; ++i) {
#pragma unroll
        for (int j = 0; j < 16; ++j) {
            if (KIND == 0) asm volatile("v_add_f32 %0, %0, %1" : "+v"(v) : "v"(w));
            if (KIND == 1) asm volatile("s_nop 1\n\tv_mov_b32_dpp %0, %0 row_shl:1 row_mask:0xf bank_mask:0xf bound_ctrl:1" : "+v"(v));
            if (KIND == 2) asm volatile("s_nop 1\n\tv_mov_b32_dpp %0, %0 wave_shl:1 row_mask:0xf bank_mask:0xf bound_ctrl:1" : "+v"(v));
            if (KIND == 3) asm volatile("s_nop 1\n\tv_add_f32_dpp %0, %0, %1 wave_shl:1 row_mask:0xf bank_mask:0xf bound_ctrl:1" : "+v"(v) : "v"(w));
            if (KIND == 4) asm volatile("s_nop 1\n\tv_mov_b32_dpp %0, %0 row_shr:1 row_mask:0xf bank_mask:0xf bound_ctrl:1" : "+v"(v));
            if (KIND == 5) asm volatile("s_nop 1\n\tv_mov_b32_dpp %0, %0 wave_ror:1 row_mask:0xf bank_mask:0xf" : "+v"(v));
            if (KIND == 6) asm volatile("s_nop 1" : "+v"(v));
        }
    }
    asm volatile("s_memtime %0\n\ts_waitcnt lgkmcnt(0)" : "=s"(t1)::"memory");
    out[blockIdx.x * blockDim.x + threadIdx.x] = v;
    if (threadIdx.x % 64 == 0) cyc[blockIdx.x * (blockDim.x / 64) + threadIdx.x / 64] = t1 - t0;
}

template <int KIND>
static void run(const char* name, float* out, unsigned long long* cyc, int threads) {
    const int n = 4096;
    k<KIND><<<1, threads>>>(out, cyc, n);
    hipDeviceSynchronize();
    k<KIND><<<1, threads>>>(out, cyc, n);
    hipDeviceSynchronize();
    unsigned long long h[16];
    hipMemcpy(h, cyc, sizeof h, hipMemcpyDeviceToHost);
    printf("%-52s %d waves/SIMD: %6.2f cycles per op (wave 0)\n", name, threads / 256, (double)h[0] / (n * 16.0));
}

int main() {
    float* out;
    unsigned long long* cyc;
    hipMalloc(&out, 1 << 16);
    hipMalloc(&cyc, 1 << 12);
    for (int threads : {256, 512}) {
        run<0>("v_add_f32 (dependent chain)", out, cyc, threads);
        run<6>("s_nop 1 alone", out, cyc, threads);
        run<1>("s_nop 1 + v_mov_b32_dpp row_shl:1", out, cyc, threads);
        run<4>("s_nop 1 + v_mov_b32_dpp row_shr:1", out, cyc, threads);
        run<2>("s_nop 1 + v_mov_b32_dpp wave_shl:1", out, cyc, threads);
        run<3>("s_nop 1 + v_add_f32_dpp wave_shl:1", out, cyc, threads);
        run<5>("s_nop 1 + v_mov_b32_dpp wave_ror:1", out, cyc, threads);
    }
    return 0;
}
