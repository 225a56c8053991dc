// Does an LDS-DMA load (global_load_lds_dwordx4 / buffer_load_dwordx4 ... lds) count in LGKM_CNT?
// One wave per workgroup issues N loads from addresses it never touched before (HBM misses) and after each one
//   mode 0: waits for nothing          (issue rate)
//   mode 1: s_waitcnt vmcnt(0)         (serial memory latency)
//   mode 2: s_waitcnt lgkmcnt(0)       (= mode 0 if LGKM_CNT does not see the load, = mode 1 if it does)
//   mode 3: one ds_read_b128 + s_waitcnt lgkmcnt(0)   (what an LDS fragment read behind a pending DMA costs)
// build: hipcc --offload-arch=gfx950 -O3 -o tools/ubench/dma_lgkm tools/ubench/dma_lgkm.hip
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>

typedef __attribute__((address_space(3))) void* lds_ptr;

template <int MODE, int KIND>
__global__ void k(const char* src, float* out, int n, size_t stride) {
    extern __shared__ __attribute__((aligned(16))) char lds[];
    const int lane = threadIdx.x;
    const char* p = src + (size_t)blockIdx.x * n * stride + lane * 16;
    float acc = 0.f;
    __amdgpu_buffer_rsrc_t rs = __builtin_amdgcn_make_buffer_rsrc(const_cast<char*>(src), 0, 0x7fffffff, 0x00020000);
    unsigned voff = (unsigned)((size_t)blockIdx.x * n * stride) + lane * 16;
    for (int i = 0; i < n; ++i) {
        if (KIND == 0) {
            __builtin_amdgcn_global_load_lds(p, (lds_ptr)(lds + (i & 3) * 1024), 16, 0, 0);
        } else {
            __builtin_amdgcn_raw_ptr_buffer_load_lds(rs, (lds_ptr)(lds + (i & 3) * 1024), 16, voff, 0, 0, 0);
            voff += (unsigned)stride;
        }
        p += stride;
        if (MODE == 1) asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        if (MODE == 2) asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
        if (MODE == 3) {
            float4 v;
            asm volatile("ds_read_b128 %0, %1 offset:8192\n\ts_waitcnt lgkmcnt(0)" : "=v"(v) : "v"(lane * 16) : "memory");
            acc += v.x;
        }
    }
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    out[blockIdx.x * 64 + lane] = acc + lds[lane];
}

template <int MODE, int KIND>
static void run(const char* name, const char* src, float* out, int n, size_t stride, int blocks) {
    hipEvent_t a, b;
    hipEventCreate(&a);
    hipEventCreate(&b);
    k<MODE, KIND><<<blocks, 64, 16384>>>(src, out, 16, stride);
    hipDeviceSynchronize();
    hipEventRecord(a);
    k<MODE, KIND><<<blocks, 64, 16384>>>(src, out, n, stride);
    hipEventRecord(b);
    hipEventSynchronize(b);
    float ms;
    hipEventElapsedTime(&ms, a, b);
    printf("%-44s %8.1f ns per load (one wave, %d loads)\n", name, ms * 1e6 / n, n);
}

int main() {
    const int n = 4096, blocks = 1;
    const size_t stride = 1 << 16;                 // every load a new 64 KB region: nothing is cached
    char* src;
    float* out;
    hipMalloc(&src, (size_t)n * stride * blocks + (1 << 20));
    hipMalloc(&out, 64 * 4 * blocks);
    hipMemset(src, 1, (size_t)n * stride * blocks + (1 << 20));
    run<0, 0>("global_load_lds  no wait", src, out, n, stride, blocks);
    run<1, 0>("global_load_lds  + s_waitcnt vmcnt(0)", src, out, n, stride, blocks);
    run<2, 0>("global_load_lds  + s_waitcnt lgkmcnt(0)", src, out, n, stride, blocks);
    run<3, 0>("global_load_lds  + ds_read + lgkmcnt(0)", src, out, n, stride, blocks);
    run<0, 1>("buffer_load lds  no wait", src, out, n, stride, blocks);
    run<1, 1>("buffer_load lds  + s_waitcnt vmcnt(0)", src, out, n, stride, blocks);
    run<2, 1>("buffer_load lds  + s_waitcnt lgkmcnt(0)", src, out, n, stride, blocks);
    run<3, 1>("buffer_load lds  + ds_read + lgkmcnt(0)", src, out, n, stride, blocks);
    return 0;
}
