// Unfused float32 kernels: one launch per graph node, every node tappable.
// This is the RN_DTYPE_F32 execution path (BASELINE config 2: batch-1 fp32 forward,
// per-layer correctness) and the on-device cross-check for the fused MFMA path.
//
// Node semantics follow the TF-1.13 CPU kernels the reference calls:
//   conv3x3_relu6   network.py:184-186  tf.layers.conv2d(3, strides=1, VALID, no bias, relu6)
//   avgpool         network.py:189      tf.nn.avg_pool VALID, divisor k*k
//   bn              network.py:193/:202 FusedBatchNorm(is_training=False): (x-mean)*inv+beta
//   resize_add      network.py:199      out + resize_bilinear(skip) (legacy, align_corners=False)
//   head            network.py:231-237, :44-45  flatten, 4 dense blocks, softmax, argmax
#include "rn_internal.h"
#include <hip/hip_bf16.h>
#include <hip/hip_fp16.h>

namespace {

__device__ __forceinline__ float relu6f(float v) { return fminf(fmaxf(v, 0.f), 6.f); }

__device__ __forceinline__ float load_as_f32(const void* p, int dtype, int64_t i) {
    if (dtype == RN_DTYPE_F32) return reinterpret_cast<const float*>(p)[i];
    if (dtype == RN_DTYPE_BF16) {
        const unsigned short u = reinterpret_cast<const unsigned short*>(p)[i];
        return __uint_as_float(static_cast<unsigned>(u) << 16);
    }
    return __half2float(reinterpret_cast<const __half*>(p)[i]);
}

// uint8 BGR -> float32 RGB through the 256-entry table fp32(fp64(v)/255*2-1)
__global__ __launch_bounds__(256) void preprocess_u8_kernel(const uint8_t* __restrict__ bgr,
                                                            float* __restrict__ rgb,
                                                            const float* __restrict__ lut, int64_t npix) {
    const int64_t p = static_cast<int64_t>(blockIdx.x) * blockDim.x + threadIdx.x;
    if (p >= npix) return;
    const uint8_t b = bgr[3 * p + 0], g = bgr[3 * p + 1], r = bgr[3 * p + 2];
    rgb[3 * p + 0] = lut[r];
    rgb[3 * p + 1] = lut[g];
    rgb[3 * p + 2] = lut[b];
}

// Direct 3x3 VALID convolution + ReLU6.  16x16 output pixels per block, one pixel per
// thread, CO_T output channels per thread; input tile staged in LDS channel-major so
// that a wave's 16 x-neighbours read consecutive banks; weights are wave-uniform and
// come in through scalar loads.
template <int CO_T>
__global__ __launch_bounds__(256) void conv3x3_relu6_f32_kernel(const float* __restrict__ in,
                                                                const float* __restrict__ w,
                                                                float* __restrict__ out, int H, int W,
                                                                int Cin, int Cout, int co_groups) {
    constexpr int TILE = 16, IT = TILE + 2, CCH = 8;
    __shared__ float tile[CCH * IT * IT];
    const int n = blockIdx.z / co_groups, cog = blockIdx.z % co_groups;
    const int ox0 = blockIdx.x * TILE, oy0 = blockIdx.y * TILE;
    const int tx = threadIdx.x & 15, ty = threadIdx.x >> 4;
    const int Ho = H - 2, Wo = W - 2;
    float acc[CO_T];
#pragma unroll
    for (int o = 0; o < CO_T; ++o) acc[o] = 0.f;
    for (int c0 = 0; c0 < Cin; c0 += CCH) {
        const int cc = min(CCH, Cin - c0);
        for (int i = threadIdx.x; i < IT * IT * CCH; i += 256) {
            const int c = i % CCH, p = i / CCH;
            const int ix = p % IT, iy = p / IT;
            const int gy = oy0 + iy, gx = ox0 + ix;
            float v = 0.f;
            if (c < cc && gy < H && gx < W)
                v = in[((static_cast<int64_t>(n) * H + gy) * W + gx) * Cin + c0 + c];
            tile[c * IT * IT + iy * IT + ix] = v;
        }
        __syncthreads();
        for (int ky = 0; ky < 3; ++ky)
            for (int kx = 0; kx < 3; ++kx)
                for (int c = 0; c < cc; ++c) {
                    const float v = tile[c * IT * IT + (ty + ky) * IT + tx + kx];
                    const float* wr = w + (static_cast<int64_t>(ky * 3 + kx) * Cin + c0 + c) * Cout + cog * CO_T;
#pragma unroll
                    for (int o = 0; o < CO_T; ++o) acc[o] = fmaf(v, wr[o], acc[o]);
                }
        __syncthreads();
    }
    const int oy = oy0 + ty, ox = ox0 + tx;
    if (oy < Ho && ox < Wo) {
        float* op = out + ((static_cast<int64_t>(n) * Ho + oy) * Wo + ox) * Cout + cog * CO_T;
#pragma unroll
        for (int o = 0; o < CO_T; ++o) op[o] = relu6f(acc[o]);
    }
}

__global__ __launch_bounds__(256) void avgpool_f32_kernel(const float* __restrict__ in, float* __restrict__ out,
                                                          int H, int W, int C, int k, int s, int Ho, int Wo,
                                                          int64_t total) {
    const int64_t i = static_cast<int64_t>(blockIdx.x) * blockDim.x + threadIdx.x;
    if (i >= total) return;
    const int c = i % C;
    int64_t p = i / C;
    const int x = p % Wo;
    p /= Wo;
    const int y = p % Ho;
    const int64_t n = p / Ho;
    float acc = 0.f;
    for (int ky = 0; ky < k; ++ky)
        for (int kx = 0; kx < k; ++kx)
            acc += in[((n * H + (y * s + ky)) * W + (x * s + kx)) * C + c];
    out[i] = acc / static_cast<float>(k * k);
}

__global__ __launch_bounds__(256) void bn_f32_kernel(const float* __restrict__ in, float* __restrict__ out,
                                                     int C, const float* __restrict__ mean,
                                                     const float* __restrict__ inv,
                                                     const float* __restrict__ beta, int64_t total) {
    const int64_t i = static_cast<int64_t>(blockIdx.x) * blockDim.x + threadIdx.x;
    if (i >= total) return;
    const int c = i % C;
    // (x - mean) * inv + beta, un-contracted like the reference's Eigen expression
    out[i] = __fadd_rn(__fmul_rn(__fsub_rn(in[i], mean[c]), inv[c]), beta[c]);
}

__global__ __launch_bounds__(256) void resize_add_f32_kernel(const float* __restrict__ x,
                                                             const float* __restrict__ skip,
                                                             float* __restrict__ out, int side, int sside,
                                                             int C, const int32_t* __restrict__ lo,
                                                             const int32_t* __restrict__ hi,
                                                             const float* __restrict__ lerp, int64_t total) {
    const int64_t i = static_cast<int64_t>(blockIdx.x) * blockDim.x + threadIdx.x;
    if (i >= total) return;
    const int c = i % C;
    int64_t p = i / C;
    const int ox = p % side;
    p /= side;
    const int oy = p % side;
    const int64_t n = p / side;
    const float* r0 = skip + (n * sside + lo[oy]) * sside * C;
    const float* r1 = skip + (n * sside + hi[oy]) * sside * C;
    const float tl = r0[lo[ox] * C + c], tr = r0[hi[ox] * C + c];
    const float bl = r1[lo[ox] * C + c], br = r1[hi[ox] * C + c];
    const float xl = lerp[ox], yl = lerp[oy];
    const float top = __fadd_rn(tl, __fmul_rn(__fsub_rn(tr, tl), xl));
    const float bottom = __fadd_rn(bl, __fmul_rn(__fsub_rn(br, bl), xl));
    const float r = __fadd_rn(top, __fmul_rn(__fsub_rn(bottom, top), yl));
    out[i] = __fadd_rn(x[i], r);
}

// Stage 0 of a float32 handle without taps in ONE launch: conv3x3 (3 -> 8) -> ReLU6 -> avg-pool 3/1 -> BN, the four per-node
// kernels' arithmetic in their order (fmaf chain over ky, kx, c; window sum over ky, kx, then / 9; un-contracted BN): same bits.
// 27 MACs per conv output are not matrix-core work (K = 27, 8 couts); what the per-node path pays for is its three round
// trips through HBM (6.5 MB per image; this kernel: input 0.6 MB + output 1.55 MB).  One workgroup = a 30 x 14 output tile =
// 32 x 16 conv pixels (two full passes of the 256 threads); the 216 weights are wave-uniform scalar operands of the fmas (as
// LDS broadcasts they were 216 ds_reads per conv pixel next to its 27 input reads: the kernel was bound by LDS issue).
__global__ __launch_bounds__(256) void stage0_fused_f32_kernel(const float* __restrict__ in, const float* __restrict__ w,
                                                               float* __restrict__ out, int S, const float* __restrict__ mean,
                                                               const float* __restrict__ inv, const float* __restrict__ beta) {
    constexpr int TX = 30, TY = 14, CX = TX + 2, CY = TY + 2, IX = TX + 4, IY = TY + 4, CO = 8;
    static_assert(CX * CY == 512, "two passes of 256 threads");
    __shared__ float tin[IY * IX * 3];
    __shared__ __attribute__((aligned(16))) float tconv[CY * CX * CO];
    const int n = blockIdx.z, ox0 = blockIdx.x * TX, oy0 = blockIdx.y * TY;
    const int So = S - 4;                                      // conv S - 2, pool 3/1: S - 4
    const float* img = in + static_cast<int64_t>(n) * S * S * 3;
    for (int i = threadIdx.x; i < IY * IX * 3; i += 256) {
        const int c = i % 3, p = i / 3, ix = p % IX, iy = p / IX;
        const int gy = min(oy0 + iy, S - 1), gx = min(ox0 + ix, S - 1);      // (clamped columns / rows only feed outputs that are not stored)
        tin[i] = img[(static_cast<int64_t>(gy) * S + gx) * 3 + c];
    }
    __syncthreads();
#pragma unroll
    for (int pass = 0; pass < 2; ++pass) {
        const int p = threadIdx.x + 256 * pass;
        const int cx = p % CX, cy = p / CX;
        float acc[CO];
#pragma unroll
        for (int o = 0; o < CO; ++o) acc[o] = 0.f;
#pragma unroll
        for (int ky = 0; ky < 3; ++ky)
#pragma unroll
            for (int kx = 0; kx < 3; ++kx)
#pragma unroll
                for (int c = 0; c < 3; ++c) {
                    const float v = tin[((cy + ky) * IX + cx + kx) * 3 + c];
#pragma unroll
                    for (int o = 0; o < CO; ++o) acc[o] = fmaf(v, w[((ky * 3 + kx) * 3 + c) * CO + o], acc[o]);
                }
        // (two 16-byte writes per pixel: as eight scalar writes at a stride of eight floats they were 8-way bank conflicts)
        float4* tp = reinterpret_cast<float4*>(tconv + p * CO);
        tp[0] = make_float4(relu6f(acc[0]), relu6f(acc[1]), relu6f(acc[2]), relu6f(acc[3]));
        tp[1] = make_float4(relu6f(acc[4]), relu6f(acc[5]), relu6f(acc[6]), relu6f(acc[7]));
    }
    __syncthreads();
    for (int q = threadIdx.x; q < TX * TY; q += 256) {
        const int tx = q % TX, ty = q / TX;
        const int ox = ox0 + tx, oy = oy0 + ty;
        if (ox >= So || oy >= So) continue;
        float* op = out + ((static_cast<int64_t>(n) * So + oy) * So + ox) * CO;
        float y[CO], acc[CO];
#pragma unroll
        for (int o = 0; o < CO; ++o) acc[o] = 0.f;
        // window sum in the per-node kernel's order (ky, then kx) for every cout; the nine pixels come as 16-byte reads
#pragma unroll
        for (int ky = 0; ky < 3; ++ky)
#pragma unroll
            for (int kx = 0; kx < 3; ++kx) {
                const float4* tp = reinterpret_cast<const float4*>(tconv + ((ty + ky) * CX + tx + kx) * CO);
                const float4 a0 = tp[0], a1 = tp[1];
                acc[0] += a0.x;
                acc[1] += a0.y;
                acc[2] += a0.z;
                acc[3] += a0.w;
                acc[4] += a1.x;
                acc[5] += a1.y;
                acc[6] += a1.z;
                acc[7] += a1.w;
            }
#pragma unroll
        for (int o = 0; o < CO; ++o) {
            const float pooled = acc[o] / 9.0f;
            y[o] = __fadd_rn(__fmul_rn(__fsub_rn(pooled, mean[o]), inv[o]), beta[o]);
        }
        *reinterpret_cast<float4*>(op) = make_float4(y[0], y[1], y[2], y[3]);
        *reinterpret_cast<float4*>(op + 4) = make_float4(y[4], y[5], y[6], y[7]);
    }
}

__global__ __launch_bounds__(256) void convert_to_f32_kernel(const void* __restrict__ in, int dtype,
                                                             float* __restrict__ out, int64_t total) {
    const int64_t i = static_cast<int64_t>(blockIdx.x) * blockDim.x + threadIdx.x;
    if (i < total) out[i] = load_as_f32(in, dtype, i);
}

// One 64-thread block (one wavefront) per image: flatten, dense chain, softmax, argmax.
constexpr int HEAD_MAX_FLAT = 4096;
// Launched with 64 threads (one wavefront) when the flatten is short (224 input: 64 values), with 1024 when it is long
// (600 input: 3136 values; 256 until round 4: 40 us per launch at 64 images, one workgroup per image on a quarter of the
// chip): the first dense layer's K loop is then split over threads / nout partitions whose partial sums
// meet in LDS -- with one wavefront the 3136-step dependent fma chain of that layer alone took 1.06 ms per launch at
// 64 x 600x600 (21 % of the forward pass, profiles/r2_600_kernel_stats.csv).  Layers with nin <= 64 run as before.
__global__ __launch_bounds__(1024) void head_kernel(const void* __restrict__ flat, int flat_dtype, HeadArgs a,
                                                   float* __restrict__ probs, int64_t* __restrict__ ids) {
    __shared__ float buf0[HEAD_MAX_FLAT];
    __shared__ float small[2][64];
    __shared__ float part[1024];
    // dense kernels staged in LDS with coalesced, independent loads (all layers that fit): read from global
    // inside the k loop, the 64 + 32 + 16 + 8 dependent steps each paid an L2 round trip (40 us per launch)
    constexpr int HEAD_W_LDS = 3072;
    __shared__ float wl[HEAD_W_LDS];
    const int img = blockIdx.x;
    const int lane = threadIdx.x;
    const int nthr = blockDim.x;
    const int nin0 = a.nin[0];
    for (int i = lane; i < nin0; i += nthr) buf0[i] = load_as_f32(flat, flat_dtype, static_cast<int64_t>(img) * nin0 + i);
    int w_off[RN_MAX_DENSE];
    {
        int off = 0;
        for (int d = 0; d < a.n_dense; ++d) {
            const int cnt = a.nin[d] * a.nout[d];
            if (off + cnt <= HEAD_W_LDS) {
                w_off[d] = off;
                for (int i = lane; i < cnt; i += nthr) wl[off + i] = a.w[d][i];
                off += cnt;
            } else {
                w_off[d] = -1;
            }
        }
    }
    __syncthreads();
    const float* cur = buf0;
    for (int d = 0; d < a.n_dense; ++d) {
        const int nin = a.nin[d], nout = a.nout[d];
        float* dst = small[d & 1];
        const float* wd = w_off[d] >= 0 ? wl + w_off[d] : a.w[d];
        const int parts = (nthr > 64 && nin > 64) ? nthr / nout : 1;      // K partitions of this layer
        if (parts > 1) {
            // partition p takes k = p, p + parts, ...: lanes of one partition read consecutive outputs (coalesced)
            const int o = lane % nout, p = lane / nout;
            float v = 0.f;
            if (p < parts) {
                // eight weight loads in flight per trip, accumulated in the original k order (same bits): one L2 round
                // trip per dependent step made this loop 0.155 ms per launch at 64 x 600x600
                int k = p;
                for (; k + 7 * parts < nin; k += 8 * parts) {
                    float wv[8];
#pragma unroll
                    for (int u = 0; u < 8; ++u) wv[u] = wd[(k + u * parts) * nout + o];
#pragma unroll
                    for (int u = 0; u < 8; ++u) v = fmaf(cur[k + u * parts], wv[u], v);
                }
                for (; k < nin; k += parts) v = fmaf(cur[k], wd[k * nout + o], v);
            }
            part[lane] = v;
            __syncthreads();
        }
        if (lane < nout) {
            float v = 0.f;
            if (parts > 1) {
                for (int p = 0; p < parts; ++p) v += part[p * nout + lane];
            } else {
                for (int k = 0; k < nin; ++k) v = fmaf(cur[k], wd[k * nout + lane], v);
            }
            if (a.bias[d]) v = __fadd_rn(v, a.bias[d][lane]);
            if (a.tap_mm[d]) a.tap_mm[d][static_cast<int64_t>(img) * nout + lane] = v;
            v = relu6f(v);
            if (a.tap_relu[d]) a.tap_relu[d][static_cast<int64_t>(img) * nout + lane] = v;
            if (a.inv[d]) {
                v = __fadd_rn(__fmul_rn(v, a.inv[d][lane]), a.shift[d][lane]);
                if (a.tap_bn[d]) a.tap_bn[d][static_cast<int64_t>(img) * nout + lane] = v;
            }
            dst[lane] = v;
        }
        __syncthreads();
        cur = dst;
    }
    // softmax + argmax over the num_classes logits (wave-level: every lane of the first wavefront holds one class)
    if (lane >= 64) return;
    const int nc = a.nout[a.n_dense - 1];
    const float logit = lane < nc ? cur[lane] : -INFINITY;
    float mx = logit;
    for (int off = 32; off > 0; off >>= 1) mx = fmaxf(mx, __shfl_xor(mx, off));
    const float e = lane < nc ? expf(logit - mx) : 0.f;
    float sum = e;
    for (int off = 32; off > 0; off >>= 1) sum += __shfl_xor(sum, off);
    const float p = e / sum;
    if (lane < nc) probs[static_cast<int64_t>(img) * nc + lane] = p;
    // argmax with lowest-index tie-break (tf.argmax)
    float bestv = lane < nc ? p : -1.f;
    int besti = lane < nc ? lane : 0x7fffffff;
    for (int off = 32; off > 0; off >>= 1) {
        const float ov = __shfl_xor(bestv, off);
        const int oi = __shfl_xor(besti, off);
        if (ov > bestv || (ov == bestv && oi < besti)) {
            bestv = ov;
            besti = oi;
        }
    }
    if (lane == 0) ids[img] = besti;
}

inline unsigned blocks_for(int64_t total, int bs = 256) { return static_cast<unsigned>((total + bs - 1) / bs); }

}  // namespace

int rn_launch_preprocess_u8(hipStream_t s, const uint8_t* bgr, float* rgb, const float* lut, int64_t npix) {
    hipLaunchKernelGGL(preprocess_u8_kernel, dim3(blocks_for(npix)), dim3(256), 0, s, bgr, rgb, lut, npix);
    RN_CHECK_LAUNCH();
    return RN_OK;
}

int rn_launch_conv3x3_relu6_f32(hipStream_t s, const float* in, const float* w, float* out, int n, int h,
                                int wd, int cin, int cout) {
    const int ho = h - 2, wo = wd - 2;
    dim3 block(256);
    if (cout % 16 == 0) {
        const int cg = cout / 16;
        dim3 grid((wo + 15) / 16, (ho + 15) / 16, n * cg);
        hipLaunchKernelGGL(conv3x3_relu6_f32_kernel<16>, grid, block, 0, s, in, w, out, h, wd, cin, cout, cg);
    } else if (cout % 8 == 0) {
        const int cg = cout / 8;
        dim3 grid((wo + 15) / 16, (ho + 15) / 16, n * cg);
        hipLaunchKernelGGL(conv3x3_relu6_f32_kernel<8>, grid, block, 0, s, in, w, out, h, wd, cin, cout, cg);
    } else {
        const int cg = cout;
        dim3 grid((wo + 15) / 16, (ho + 15) / 16, n * cg);
        hipLaunchKernelGGL(conv3x3_relu6_f32_kernel<1>, grid, block, 0, s, in, w, out, h, wd, cin, cout, cg);
    }
    RN_CHECK_LAUNCH();
    return RN_OK;
}

int rn_launch_avgpool_f32(hipStream_t s, const float* in, float* out, int n, int h, int w, int c, int k, int st) {
    const int ho = (h - k) / st + 1, wo = (w - k) / st + 1;
    const int64_t total = static_cast<int64_t>(n) * ho * wo * c;
    hipLaunchKernelGGL(avgpool_f32_kernel, dim3(blocks_for(total)), dim3(256), 0, s, in, out, h, w, c, k, st, ho,
                       wo, total);
    RN_CHECK_LAUNCH();
    return RN_OK;
}

int rn_launch_stage0_fused_f32(hipStream_t s, const float* in, const float* w, float* out, int n, int side, const BnDev& bn) {
    const int so = side - 4;
    hipLaunchKernelGGL(stage0_fused_f32_kernel, dim3((so + 29) / 30, (so + 13) / 14, n), dim3(256), 0, s, in, w, out, side, bn.mean, bn.inv, bn.beta);
    RN_CHECK_LAUNCH();
    return RN_OK;
}

int rn_launch_bn_f32(hipStream_t s, const float* in, float* out, int64_t npix, int c, const BnDev& bn) {
    const int64_t total = npix * c;
    hipLaunchKernelGGL(bn_f32_kernel, dim3(blocks_for(total)), dim3(256), 0, s, in, out, c, bn.mean, bn.inv,
                       bn.beta, total);
    RN_CHECK_LAUNCH();
    return RN_OK;
}

int rn_launch_resize_add_f32(hipStream_t s, const float* x, const float* skip, float* out, int n, int side,
                             int skip_side, int c, const ResizeTab& rt) {
    const int64_t total = static_cast<int64_t>(n) * side * side * c;
    hipLaunchKernelGGL(resize_add_f32_kernel, dim3(blocks_for(total)), dim3(256), 0, s, x, skip, out, side,
                       skip_side, c, rt.lo, rt.hi, rt.lerp, total);
    RN_CHECK_LAUNCH();
    return RN_OK;
}

int rn_launch_convert_to_f32(hipStream_t s, const void* in, int dtype, float* out, int64_t n) {
    hipLaunchKernelGGL(convert_to_f32_kernel, dim3(blocks_for(n)), dim3(256), 0, s, in, dtype, out, n);
    RN_CHECK_LAUNCH();
    return RN_OK;
}

int rn_launch_head(hipStream_t s, const void* flat, int flat_dtype, int n, const HeadArgs& a, float* probs,
                   int64_t* ids) {
    if (a.nin[0] > HEAD_MAX_FLAT) {
        rn_set_error("flatten length %d exceeds head kernel capacity %d", a.nin[0], HEAD_MAX_FLAT);
        return RN_E_INVALID;
    }
    for (int d = 0; d < a.n_dense; ++d)
        if (a.nout[d] > 64 || (d > 0 && a.nin[d] > 64)) {
            rn_set_error("dense layer %d wider than 64 is not supported by the head kernel", d);
            return RN_E_INVALID;
        }
    const int threads = (a.nin[0] > 256 && a.nout[0] <= 64 && 1024 % a.nout[0] == 0) ? 1024 : 64;
    hipLaunchKernelGGL(head_kernel, dim3(n), dim3(threads), 0, s, flat, flat_dtype, a, probs, ids);
    RN_CHECK_LAUNCH();
    return RN_OK;
}
