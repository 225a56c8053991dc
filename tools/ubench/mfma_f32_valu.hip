// Microbenchmark: do VALU instructions hide behind v_mfma_f32_32x32x2_f32 (the fp32 matrix instruction of rn_stage_f32m.hip)?
// Loop body: one dependent fp32 MFMA followed by NV independent VALU instructions (other registers), 1 or 2 waves per SIMD.
// Prints shader cycles per (MFMA + NV VALU) group of one wave.  64 cycles = the MFMA alone (16 passes).
#include <hip/hip_runtime.h>
#include <algorithm>
#include <cstdio>
#include <vector>
using f32x16 = __attribute__((ext_vector_type(16))) float;

template <int NV>
__device__ __forceinline__ void valu(float (&x)[8]) {
    if constexpr (NV > 0) asm volatile("v_add_f32 %0, %0, %0" : "+v"(x[0]));
    if constexpr (NV > 1) asm volatile("v_fma_f32 %0, %0, %1, %0" : "+v"(x[1]) : "v"(x[7]));
    if constexpr (NV > 2) asm volatile("v_med3_f32 %0, %0, 0, %1" : "+v"(x[2]) : "v"(x[7]));
    if constexpr (NV > 3) asm volatile("v_add_f32 %0, %0, %0" : "+v"(x[3]));
    if constexpr (NV > 4) valu<NV - 4>(x);
}

template <int NV, int WAVES>
__global__ __launch_bounds__(64 * WAVES) void k(float* out, unsigned long long* cyc, int iters) {
    float x[8];
    for (int i = 0; i < 8; ++i) x[i] = 1.0f + threadIdx.x * 1e-6f + i;
    f32x16 acc = {};
    float wa = 1.0f + threadIdx.x * 1e-3f, wb = 0.5f;
    __syncthreads();
    unsigned long long t0 = __builtin_amdgcn_s_memtime();
    for (int it = 0; it < iters; ++it) {
#pragma unroll
        for (int i = 0; i < 8; ++i) {
            asm volatile("v_mfma_f32_32x32x2_f32 %0, %1, %2, %0" : "+v"(acc) : "v"(wa), "v"(wb));
            valu<NV>(x);
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

template <int NV, int WAVES>
void run() {
    const int iters = 400, nb = 256;
    float* out;
    unsigned long long* cyc;
    hipMalloc(&out, nb * 64 * WAVES * 4);
    hipMalloc(&cyc, nb * WAVES * 8);
    for (int rep = 0; rep < 2; ++rep) hipLaunchKernelGGL((k<NV, WAVES>), dim3(nb), dim3(64 * WAVES), 0, 0, out, cyc, iters);
    hipDeviceSynchronize();
    std::vector<unsigned long long> h(nb * WAVES);
    hipMemcpy(h.data(), cyc, h.size() * 8, hipMemcpyDeviceToHost);
    std::sort(h.begin(), h.end());
    printf("NV %2d  waves/SIMD %d: %6.1f cycles per (fp32 MFMA + %d VALU) of one wave  -> %5.1f cycles of the SIMD per MFMA\n", NV, WAVES / 4,
           h[h.size() / 2] / (iters * 8.0), NV, h[h.size() / 2] / (iters * 8.0) / (WAVES / 4));
    hipFree(out);
    hipFree(cyc);
}

int main() {
    run<0, 4>(); run<2, 4>(); run<4, 4>(); run<8, 4>(); run<12, 4>(); run<16, 4>();
    run<0, 8>(); run<2, 8>(); run<4, 8>(); run<8, 8>(); run<12, 8>(); run<16, 8>();
    return 0;
}
