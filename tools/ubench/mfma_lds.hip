// Microbenchmark: cycles per v_mfma_f32_32x32x16_bf16 in a dependent chain whose B operand is
// read from LDS (ds_read_b128 via inline asm) D chunks ahead, A operand in registers.
// One wave per SIMD (256-thread workgroups, one per CU).  Prints cycles per MFMA per variant.
#include <hip/hip_runtime.h>
#include <cstdio>
#include <vector>
#include <algorithm>
using i32x4 = __attribute__((ext_vector_type(4))) int;
using f32x16 = __attribute__((ext_vector_type(16))) float;
using bf16x8 = __attribute__((ext_vector_type(8))) __bf16;

template <int D, bool USE_LDS, int WAVES>
__global__ __launch_bounds__(64 * WAVES) void k(float* out, unsigned long long* cyc, int iters) {
    extern __shared__ __attribute__((aligned(16))) char smem[];
    for (int i = threadIdx.x; i < 48 * 1024 / 4; i += blockDim.x) reinterpret_cast<int*>(smem)[i] = 0x3c003c00 + i;
    __syncthreads();
    constexpr int KC = 36;
    i32x4 w[KC];
    for (int i = 0; i < KC; ++i) w[i] = i32x4{0x3f803f80 + i, 0x3f803f80, 0x3f803f80, (int)threadIdx.x};
    const unsigned base = static_cast<unsigned>(reinterpret_cast<uintptr_t>((__attribute__((address_space(3))) char*)smem)) + (threadIdx.x & 63) * 16 + (threadIdx.x >> 6) * 1024;
    f32x16 acc = {};
    unsigned long long t0 = __builtin_amdgcn_s_memtime();
    for (int it = 0; it < iters; ++it) {
        i32x4 b[KC];
        auto rd = [&](int kc) {
            i32x4 v;
            if constexpr (USE_LDS) {
                asm volatile("ds_read_b128 %0, %1 offset:%2" : "=v"(v) : "v"(base), "n"(0));
            } else {
                v = i32x4{(int)base, kc, it, 1};
                asm volatile("" : "+v"(v));
            }
            return v;
        };
#pragma unroll
        for (int i = 0; i < D; ++i) b[i] = rd(i);
#pragma unroll
        for (int i = 0; i < KC; ++i) {
            if (i + D < KC) b[i + D] = rd(i + D);
            if constexpr (USE_LDS) {
                const int newer = (KC - 1 - i) < D ? (KC - 1 - i) : D;
                switch (newer) {   // constant after unrolling
                    case 0: asm volatile("s_waitcnt lgkmcnt(0)" : "+v"(b[i])); break;
                    case 1: asm volatile("s_waitcnt lgkmcnt(1)" : "+v"(b[i])); break;
                    case 2: asm volatile("s_waitcnt lgkmcnt(2)" : "+v"(b[i])); break;
                    case 3: asm volatile("s_waitcnt lgkmcnt(3)" : "+v"(b[i])); break;
                    case 4: asm volatile("s_waitcnt lgkmcnt(4)" : "+v"(b[i])); break;
                    case 5: asm volatile("s_waitcnt lgkmcnt(5)" : "+v"(b[i])); break;
                    case 6: asm volatile("s_waitcnt lgkmcnt(6)" : "+v"(b[i])); break;
                    case 7: asm volatile("s_waitcnt lgkmcnt(7)" : "+v"(b[i])); break;
                    default: asm volatile("s_waitcnt lgkmcnt(8)" : "+v"(b[i])); break;
                }
            }
            acc = __builtin_amdgcn_mfma_f32_32x32x16_bf16(__builtin_bit_cast(bf16x8, w[i]), __builtin_bit_cast(bf16x8, b[i]), acc, 0, 0, 0);
        }
    }
    unsigned long long t1 = __builtin_amdgcn_s_memtime();
    float s = 0;
    for (int i = 0; i < 16; ++i) s += acc[i];
    out[blockIdx.x * blockDim.x + threadIdx.x] = s;
    if ((threadIdx.x & 63) == 0) cyc[blockIdx.x * WAVES + (threadIdx.x >> 6)] = t1 - t0;
}

template <int D, bool USE_LDS, int WAVES>
void run(const char* name) {
    const int iters = 200, nb = 256;
    float* out;
    unsigned long long* cyc;
    hipMalloc(&out, nb * 64 * WAVES * 4);
    hipMalloc(&cyc, nb * WAVES * 8);
    hipFuncSetAttribute(reinterpret_cast<const void*>(k<D, USE_LDS, WAVES>), hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024);
    for (int rep = 0; rep < 2; ++rep) hipLaunchKernelGGL((k<D, USE_LDS, WAVES>), dim3(nb), dim3(64 * WAVES), 96 * 1024, 0, out, cyc, iters);
    hipDeviceSynchronize();
    std::vector<unsigned long long> h(nb * WAVES);
    hipMemcpy(h.data(), cyc, h.size() * 8, hipMemcpyDeviceToHost);
    std::sort(h.begin(), h.end());
    printf("%-28s waves/CU %d: median %.1f cycles per MFMA (min %.1f max %.1f)\n", name, WAVES, h[h.size() / 2] / (200.0 * 36),
           h.front() / (200.0 * 36), h.back() / (200.0 * 36));
    hipFree(out);
    hipFree(cyc);
}

int main() {
    run<1, false, 4>("regs only");
    run<1, true, 4>("LDS depth 1");
    run<2, true, 4>("LDS depth 2");
    run<4, true, 4>("LDS depth 4");
    run<8, true, 4>("LDS depth 8");
    run<4, true, 8>("LDS depth 4");
    run<8, true, 8>("LDS depth 8");
    run<1, false, 8>("regs only");
    return 0;
}
