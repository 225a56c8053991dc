// Microbenchmark: what the HBM system of this box delivers for plain streaming kernels
// (read-only sum, write-only fill, float4 copy) at the sizes of one stage's tensors (~0.75 GB).
#include <hip/hip_runtime.h>
#include <cstdio>
using f4 = __attribute__((ext_vector_type(4))) float;
__global__ __launch_bounds__(256) void k_read(const f4* __restrict__ in, float* out, size_t n) {
    f4 acc = {0, 0, 0, 0};
    for (size_t i = blockIdx.x * 256ull + threadIdx.x; i < n; i += gridDim.x * 256ull) acc += in[i];
    if (acc[0] + acc[1] + acc[2] + acc[3] == 12345.f) out[0] = 1.f;
}
__global__ __launch_bounds__(256) void k_write(f4* __restrict__ out, size_t n) {
    const f4 v = {1.f, 2.f, 3.f, (float)threadIdx.x};
    for (size_t i = blockIdx.x * 256ull + threadIdx.x; i < n; i += gridDim.x * 256ull) out[i] = v;
}
__global__ __launch_bounds__(256) void k_copy(const f4* __restrict__ in, f4* __restrict__ out, size_t n) {
    for (size_t i = blockIdx.x * 256ull + threadIdx.x; i < n; i += gridDim.x * 256ull) out[i] = in[i];
}
// each workgroup streams its own contiguous chunk (like one image per workgroup)
__global__ __launch_bounds__(256) void k_copy_chunk(const f4* __restrict__ in, f4* __restrict__ out, size_t per_wg) {
    const f4* a = in + blockIdx.x * per_wg;
    f4* b = out + blockIdx.x * per_wg;
    for (size_t i = threadIdx.x; i < per_wg; i += 256) b[i] = a[i];
}
template <typename F>
float timeit(F f, int reps = 10) {
    hipEvent_t e0, e1;
    hipEventCreate(&e0); hipEventCreate(&e1);
    f(); f();
    hipEventRecord(e0);
    for (int i = 0; i < reps; ++i) f();
    hipEventRecord(e1);
    hipEventSynchronize(e1);
    float ms; hipEventElapsedTime(&ms, e0, e1);
    return ms / reps;
}
int main() {
    const size_t bytes = 768ull << 20, n = bytes / 16;
    f4 *a, *b; float* o;
    hipMalloc(&a, bytes); hipMalloc(&b, bytes); hipMalloc(&o, 4);
    hipMemset(a, 0, bytes); hipMemset(b, 0, bytes);
    for (int wgs : {256, 512, 1024, 2048, 8192}) {
        float r = timeit([&] { hipLaunchKernelGGL(k_read, dim3(wgs), dim3(256), 0, 0, a, o, n); });
        float w = timeit([&] { hipLaunchKernelGGL(k_write, dim3(wgs), dim3(256), 0, 0, b, n); });
        float c = timeit([&] { hipLaunchKernelGGL(k_copy, dim3(wgs), dim3(256), 0, 0, a, b, n); });
        float cc = timeit([&] { hipLaunchKernelGGL(k_copy_chunk, dim3(wgs), dim3(256), 0, 0, a, b, n / wgs); });
        printf("wgs %5d: read %.2f TB/s   write %.2f TB/s   copy %.2f TB/s (r+w)   chunked copy %.2f TB/s\n", wgs,
               bytes / r / 1e9, bytes / w / 1e9, 2 * bytes / c / 1e9, 2 * bytes / cc / 1e9);
    }
    return 0;
}
