// Microbenchmark: does ONE wave overlap its own VALU instructions with its in-flight MFMA?
// Loop body: one dependent v_mfma_f32_32x32x16_bf16 followed by NV independent VALU instructions (other registers).
// Accumulator in VGPRs ("v" form) vs AGPRs ("a" form).  Prints cycles per (MFMA + NV VALU) group.
#include <hip/hip_runtime.h>
#include <algorithm>
#include <cstdio>
#include <vector>
using f32x16 = __attribute__((ext_vector_type(16))) float;
using i32x4 = __attribute__((ext_vector_type(4))) int;

template <int NV, bool DPP>
__device__ __forceinline__ void valu(float (&x)[8]) {
    if constexpr (NV > 0) {
        if constexpr (DPP)
            asm volatile("v_add_f32_dpp %0, %0, %0 wave_shl:1 row_mask:0xf bank_mask:0xf bound_ctrl:1" : "+v"(x[0]));
        else
            asm volatile("v_add_f32 %0, %0, %0" : "+v"(x[0]));
    }
    if constexpr (NV > 1) asm volatile("v_add_f32 %0, %0, %0" : "+v"(x[1]));
    if constexpr (NV > 2) asm volatile("v_med3_f32 %0, %0, 0, %1" : "+v"(x[2]) : "v"(x[7]));
    if constexpr (NV > 3) asm volatile("v_fma_f32 %0, %0, %1, %0" : "+v"(x[3]) : "v"(x[7]));
    if constexpr (NV > 4) {
        if constexpr (DPP)
            asm volatile("v_add_f32_dpp %0, %0, %0 wave_shl:1 row_mask:0xf bank_mask:0xf bound_ctrl:1" : "+v"(x[4]));
        else
            asm volatile("v_add_f32 %0, %0, %0" : "+v"(x[4]));
    }
    if constexpr (NV > 5) asm volatile("v_add_f32 %0, %0, %0" : "+v"(x[5]));
    if constexpr (NV > 6) asm volatile("v_med3_f32 %0, %0, 0, %1" : "+v"(x[6]) : "v"(x[7]));
    if constexpr (NV > 7) asm volatile("v_fma_f32 %0, %0, %1, %0" : "+v"(x[3]) : "v"(x[7]));
    if constexpr (NV > 8) valu<NV - 8, DPP>(x);
}

template <int NV, bool AGPR, bool DPP, int WAVES>
__global__ __launch_bounds__(64 * WAVES) void k(float* out, unsigned long long* cyc, int iters) {
    float x[8];
    for (int i = 0; i < 8; ++i) x[i] = 1.0f + threadIdx.x * 1e-6f + i;
    f32x16 acc = {};
    i32x4 wa = {0x3f803f80, 0x3f803f80, 0x3f803f80, (int)threadIdx.x}, wb = {0x3c003c00, 1, 2, (int)threadIdx.x};
    __syncthreads();
    unsigned long long t0 = __builtin_amdgcn_s_memtime();
    for (int it = 0; it < iters; ++it) {
#pragma unroll
        for (int i = 0; i < 8; ++i) {
            if constexpr (AGPR)
                asm volatile("v_mfma_f32_32x32x16_bf16 %0, %1, %2, %0" : "+a"(acc) : "v"(wa), "v"(wb));
            else
                asm volatile("v_mfma_f32_32x32x16_bf16 %0, %1, %2, %0" : "+v"(acc) : "v"(wa), "v"(wb));
            valu<NV, DPP>(x);
        }
    }
    unsigned long long t1 = __builtin_amdgcn_s_memtime();
    asm volatile("s_nop 15\n\ts_nop 15" ::: "memory");
    float s = 0;
    for (int i = 0; i < 16; ++i) s += acc[i];
    for (int i = 0; i < 8; ++i) s += x[i];
    out[blockIdx.x * blockDim.x + threadIdx.x] = s;
    if ((threadIdx.x & 63) == 0) cyc[blockIdx.x * WAVES + (threadIdx.x >> 6)] = t1 - t0;
}

template <int NV, bool AGPR, bool DPP, int WAVES>
void run() {
    const int iters = 400, nb = 256;
    float* out; unsigned long long* cyc;
    hipMalloc(&out, nb * 64 * WAVES * 4); hipMalloc(&cyc, nb * WAVES * 8);
    for (int rep = 0; rep < 2; ++rep) hipLaunchKernelGGL((k<NV, AGPR, DPP, WAVES>), dim3(nb), dim3(64 * WAVES), 0, 0, out, cyc, iters);
    hipDeviceSynchronize();
    std::vector<unsigned long long> h(nb * WAVES);
    hipMemcpy(h.data(), cyc, h.size() * 8, hipMemcpyDeviceToHost);
    std::sort(h.begin(), h.end());
    printf("NV %2d  acc in %s  %s  waves/SIMD %d: %6.1f cycles per (MFMA + %d VALU)\n", NV, AGPR ? "AGPR" : "VGPR", DPP ? "dpp mix" : "plain  ",
           WAVES / 4, h[h.size() / 2] / (iters * 8.0), NV);
    hipFree(out); hipFree(cyc);
}
template <int NV>
void sweep() {
    run<NV, false, false, 4>(); run<NV, true, false, 4>();
    run<NV, false, true, 4>(); run<NV, true, true, 4>();
    run<NV, false, true, 8>(); run<NV, true, true, 8>();
}
int main() {
    sweep<0>(); sweep<4>(); sweep<8>(); sweep<12>(); sweep<16>();
    return 0;
}
