// Microbenchmark: what does the chip SUSTAIN at its power cap?  Runs each load for a few seconds and prints the rate per
// interval: (a) bare v_mfma_f32_32x32x16_{bf16,f16} on random operands, 4 independent accumulators per wave, 2 waves per
// SIMD; (b) the same with every B operand re-read from LDS (ds_read_b128); (c) a float4 HBM copy.  Sample
// `rocm-smi --showclocks --showpower` next to it (tools/power_cap.sh) to see the clock the governor settles at.
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#include <vector>
#include <chrono>
using i32x4 = __attribute__((ext_vector_type(4))) int;
using f32x16 = __attribute__((ext_vector_type(16))) float;
using bf16x8 = __attribute__((ext_vector_type(8))) __bf16;
using f16x8 = __attribute__((ext_vector_type(8))) _Float16;

template <bool F16, bool LDS>
__global__ __launch_bounds__(512, 2) void mfma_k(const int* __restrict__ seed, float* out, int iters) {
    __shared__ __attribute__((aligned(16))) int smem[16 * 1024];
    for (int i = threadIdx.x; i < 16 * 1024; i += blockDim.x) smem[i] = seed[i];
    __syncthreads();
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    i32x4 a[4], b[4];
    for (int i = 0; i < 4; ++i) {
        a[i] = *reinterpret_cast<const i32x4*>(&smem[((wave * 4 + i) * 64 + lane) * 4 % (16 * 1024 - 4)]);
        b[i] = *reinterpret_cast<const i32x4*>(&smem[((wave * 4 + i + 32) * 64 + lane) * 4 % (16 * 1024 - 4)]);
    }
    f32x16 acc[4] = {};
    const unsigned base = static_cast<unsigned>(reinterpret_cast<uintptr_t>((__attribute__((address_space(3))) char*)smem)) + lane * 16 + wave * 4096;
    for (int it = 0; it < iters; ++it) {
#pragma unroll
        for (int u = 0; u < 8; ++u) {
#pragma unroll
            for (int i = 0; i < 4; ++i) {
                i32x4 bv = b[i];
                if constexpr (LDS) asm volatile("ds_read_b128 %0, %1 offset:%2\n\ts_waitcnt lgkmcnt(0)" : "=v"(bv) : "v"(base), "n"(0));
                if constexpr (F16)
                    acc[i] = __builtin_amdgcn_mfma_f32_32x32x16_f16(__builtin_bit_cast(f16x8, a[i]), __builtin_bit_cast(f16x8, bv), acc[i], 0, 0, 0);
                else
                    acc[i] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(__builtin_bit_cast(bf16x8, a[i]), __builtin_bit_cast(bf16x8, bv), acc[i], 0, 0, 0);
            }
        }
    }
    float s = 0;
    for (int i = 0; i < 4; ++i)
        for (int j = 0; j < 16; ++j) s += acc[i][j];
    out[blockIdx.x * blockDim.x + threadIdx.x] = s;
}

// the same loop on v_mfma_f32_16x16x32_bf16: 8 independent 16x16 accumulators (4 registers each) per wave
using f32x4 = __attribute__((ext_vector_type(4))) float;
template <bool LDS>
__global__ __launch_bounds__(512, 2) void mfma16_k(const int* __restrict__ seed, float* out, int iters) {
    __shared__ __attribute__((aligned(16))) int smem[16 * 1024];
    for (int i = threadIdx.x; i < 16 * 1024; i += blockDim.x) smem[i] = seed[i];
    __syncthreads();
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    i32x4 a[4], b[4];
    for (int i = 0; i < 4; ++i) {
        a[i] = *reinterpret_cast<const i32x4*>(&smem[((wave * 4 + i) * 64 + lane) * 4 % (16 * 1024 - 4)]);
        b[i] = *reinterpret_cast<const i32x4*>(&smem[((wave * 4 + i + 32) * 64 + lane) * 4 % (16 * 1024 - 4)]);
    }
    f32x4 acc[8] = {};
    const unsigned base = static_cast<unsigned>(reinterpret_cast<uintptr_t>((__attribute__((address_space(3))) char*)smem)) + lane * 16 + wave * 4096;
    for (int it = 0; it < iters; ++it) {
#pragma unroll
        for (int u = 0; u < 8; ++u) {
#pragma unroll
            for (int i = 0; i < 8; ++i) {
                i32x4 bv = b[i & 3];
                if constexpr (LDS) {
                    if ((i & 1) == 0) asm volatile("ds_read_b128 %0, %1 offset:%2\n\ts_waitcnt lgkmcnt(0)" : "=v"(bv) : "v"(base), "n"(0));
                }
                acc[i] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(__builtin_bit_cast(bf16x8, a[i & 3]), __builtin_bit_cast(bf16x8, bv), acc[i], 0, 0, 0);
            }
        }
    }
    float s = 0;
    for (int i = 0; i < 8; ++i)
        for (int j = 0; j < 4; ++j) s += acc[i][j];
    out[blockIdx.x * blockDim.x + threadIdx.x] = s;
}

__global__ void copy_k(const float4* __restrict__ in, float4* __restrict__ out, size_t n) {
    for (size_t i = blockIdx.x * (size_t)blockDim.x + threadIdx.x; i < n; i += (size_t)gridDim.x * blockDim.x) out[i] = in[i];
}

template <typename F>
void sustain(const char* name, double unit_per_launch, const char* unit, double seconds, F launch) {
    hipEvent_t e0, e1;
    hipEventCreate(&e0);
    hipEventCreate(&e1);
    const auto t_end = std::chrono::steady_clock::now() + std::chrono::duration<double>(seconds);
    int k = 0;
    while (std::chrono::steady_clock::now() < t_end) {
        hipEventRecord(e0);
        for (int r = 0; r < 20; ++r) launch();
        hipEventRecord(e1);
        hipEventSynchronize(e1);
        float ms;
        hipEventElapsedTime(&ms, e0, e1);
        if (k++ % 8 == 0) printf("%-34s %8.1f %s\n", name, 20 * unit_per_launch / (ms * 1e-3) * 1e-12, unit);
        fflush(stdout);
    }
}

int main(int argc, char** argv) {
    const double secs = argc > 1 ? atof(argv[1]) : 4.0;
    std::vector<int> h(16 * 1024);
    srand(1);
    for (auto& v : h) {   // two random 16-bit floats of magnitude ~1 per dword (valid in both formats)
        const unsigned lo = 0x3c00u + (rand() & 0x3ff) + ((rand() & 1) << 15), hi = 0x3c00u + (rand() & 0x3ff) + ((rand() & 1) << 15);
        v = static_cast<int>(lo | (hi << 16));
    }
    int* seed;
    float* out;
    hipMalloc(&seed, h.size() * 4);
    hipMemcpy(seed, h.data(), h.size() * 4, hipMemcpyHostToDevice);
    hipMalloc(&out, 4096 * 512 * 4);
    const int iters = 2000, nb = 1024;
    const double flop = double(nb) * 8 * iters * 32 * 2.0 * 32 * 32 * 16;
    sustain("mfma bf16 32x32x16, registers", flop, "TFLOP/s", secs, [&] { hipLaunchKernelGGL((mfma_k<false, false>), dim3(nb), dim3(512), 0, 0, seed, out, iters); });
    sustain("mfma f16  32x32x16, registers", flop, "TFLOP/s", secs, [&] { hipLaunchKernelGGL((mfma_k<true, false>), dim3(nb), dim3(512), 0, 0, seed, out, iters); });
    const double flop16 = double(nb) * 8 * iters * 64 * 2.0 * 16 * 16 * 32;
    sustain("mfma bf16 16x16x32, registers", flop16, "TFLOP/s", secs, [&] { hipLaunchKernelGGL((mfma16_k<false>), dim3(nb), dim3(512), 0, 0, seed, out, iters); });
    sustain("mfma bf16 16x16x32, B from LDS (1 read per 2 MFMAs)", flop16, "TFLOP/s", secs, [&] { hipLaunchKernelGGL((mfma16_k<true>), dim3(nb), dim3(512), 0, 0, seed, out, iters); });
    sustain("mfma bf16 32x32x16, B from LDS", flop, "TFLOP/s", secs, [&] { hipLaunchKernelGGL((mfma_k<false, true>), dim3(nb), dim3(512), 0, 0, seed, out, iters); });
    const size_t n4 = (size_t(1) << 30) / 16;
    float4 *ci, *co;
    hipMalloc(&ci, n4 * 16);
    hipMalloc(&co, n4 * 16);
    hipMemset(ci, 1, n4 * 16);
    sustain("float4 copy 1 GiB (read + write)", 2.0 * n4 * 16, "TB/s", secs, [&] { hipLaunchKernelGGL(copy_k, dim3(4096), dim3(256), 0, 0, ci, co, n4); });
    return 0;
}
