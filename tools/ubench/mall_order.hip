// Does the 256 MB memory-side cache (MALL) keep what a launch WROTE, and does the order in which the next launch reads it
// matter?  Launch A writes a buffer of X bytes front to back (every workgroup owns a contiguous 1/G slice and walks it
// upwards, like the stage kernels walk the rows of their image); launch B reads it (a) in the same direction, (b) every
// slice from its END downwards -- the bytes A wrote last are the ones B touches first.  Timed: B alone (events), per X.
//   hipcc --offload-arch=gfx950 -O3 tools/ubench/mall_order.hip -o tools/ubench/mall_order && tools/ubench/mall_order
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#define CK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("%s: %s\n", #x, hipGetErrorString(e)); exit(1); } } while (0)

__global__ void wr(float4* p, size_t n4_per_wg, float v) {
    float4* q = p + blockIdx.x * n4_per_wg;
    for (size_t i = threadIdx.x; i < n4_per_wg; i += blockDim.x) q[i] = make_float4(v, v + 1, v + 2, (float)i);
}
template <int REV>
__global__ void rd(const float4* p, size_t n4_per_wg, float* out) {
    const float4* q = p + blockIdx.x * n4_per_wg;
    float4 a = make_float4(0, 0, 0, 0);
    const size_t chunks = n4_per_wg / blockDim.x;
    for (size_t c = 0; c < chunks; ++c) {
        const size_t cc = REV ? chunks - 1 - c : c;
        const float4 v = q[cc * blockDim.x + threadIdx.x];
        a.x += v.x; a.y += v.y; a.z += v.z; a.w += v.w;
    }
    if (a.x + a.y + a.z + a.w == 12345.678f) out[0] = a.x;
}
int main() {
    const int G = 256, T = 512;
    float* out; CK(hipMalloc(&out, 4));
    hipEvent_t e0, e1; CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));
    for (size_t mb : {64, 128, 192, 256, 384, 512, 768, 1024}) {
        const size_t bytes = mb << 20, n4 = bytes / 16, per = n4 / G;
        float4* buf; CK(hipMalloc(&buf, bytes));
        float best[2] = {1e9f, 1e9f}, wbest = 1e9f;
        for (int rep = 0; rep < 6; ++rep)
            for (int rev = 0; rev < 2; ++rev) {
                CK(hipEventRecord(e0));
                hipLaunchKernelGGL(wr, dim3(G), dim3(T), 0, 0, buf, per, (float)rep);
                CK(hipEventRecord(e1)); CK(hipEventSynchronize(e1));
                float ms; CK(hipEventElapsedTime(&ms, e0, e1)); if (ms < wbest) wbest = ms;
                CK(hipEventRecord(e0));
                if (rev) hipLaunchKernelGGL(rd<1>, dim3(G), dim3(T), 0, 0, buf, per, out);
                else hipLaunchKernelGGL(rd<0>, dim3(G), dim3(T), 0, 0, buf, per, out);
                CK(hipEventRecord(e1)); CK(hipEventSynchronize(e1));
                CK(hipEventElapsedTime(&ms, e0, e1)); if (ms < best[rev]) best[rev] = ms;
            }
        printf("%5zu MB: write %.3f ms (%.2f TB/s)   read same order %.3f ms (%.2f TB/s)   read reversed %.3f ms (%.2f TB/s)\n", mb, wbest,
               bytes / wbest * 1e-9, best[0], bytes / best[0] * 1e-9, best[1], bytes / best[1] * 1e-9);
        CK(hipFree(buf));
    }
    return 0;
}
