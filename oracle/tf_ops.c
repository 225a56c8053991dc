/*
 * ORACLE (test infrastructure, not product code) -- plain-C restatement of the
 * TensorFlow 1.13.1 CPU op semantics that the reference's forward pass executes
 * (the ops are called from /root/reference/network.py; the kernels themselves
 * live in the un-vendored pip dependency tensorflow==1.13.1).
 *
 * PARITY UNPINNED: TensorFlow cannot be installed here and the reference ships
 * no golden vectors, so this file is checked against a second independent
 * restatement (oracle/roomnet_ref.py) and a torch-CPU cross-check, not against
 * TensorFlow itself.
 *
 * Each function is one graph-node type; oracle/c_oracle.py wires them into the
 * graph of network.py:225-237.  All tensors are float32, NHWC, C-contiguous.
 * Used only by tests/, __graft_entry__.smoke() and bench.py's cpu_baseline leg.
 *
 *   rn_ref_conv2d_valid      network.py:184  tf.layers.conv2d(k=3,s=1,VALID,no bias)
 *   rn_ref_relu6             network.py:185 / :214  tf.nn.relu6
 *   rn_ref_avg_pool_valid    network.py:189  tf.nn.avg_pool(VALID)
 *   rn_ref_fused_bn_infer    network.py:193,:202  FusedBatchNorm(is_training=False)
 *   rn_ref_resize_bilinear   network.py:199  tf.image.resize_bilinear (legacy)
 *   rn_ref_add               network.py:199  "+"
 *   rn_ref_matmul            network.py:212  tf.layers.dense
 *   rn_ref_bias_add          network.py:212  use_bias=True (last layer)
 *   rn_ref_bn_2d             network.py:217  tf.nn.batch_normalization (rank-2 path)
 *   rn_ref_softmax           network.py:44
 *   rn_ref_argmax            network.py:45   lowest index on ties
 *   rn_ref_preprocess        network.py:129/:153  ((x[...,[2,1,0]]/255.)*2)-1 in fp64 -> fp32
 */
#include <math.h>
#include <stdint.h>
#include <string.h>
#include <stdlib.h>

#ifdef _OPENMP
#include <omp.h>
#endif

#define API __attribute__((visibility("default")))

API int rn_ref_max_threads(void) {
#ifdef _OPENMP
    return omp_get_max_threads();
#else
    return 1;
#endif
}

API void rn_ref_set_threads(int n) {
#ifdef _OPENMP
    if (n > 0) omp_set_num_threads(n);
#else
    (void)n;
#endif
}

/* uint8 BGR -> float32 RGB in [-1,1]; the reference evaluates the expression in
 * float64 (NumPy true division) and TensorFlow casts to float32 at the feed. */
API void rn_ref_preprocess(const uint8_t* bgr, float* rgb, int64_t npix) {
    float lut[256];
    for (int v = 0; v < 256; ++v) lut[v] = (float)((((double)v / 255.) * 2) - 1);
#pragma omp parallel for schedule(static)
    for (int64_t i = 0; i < npix; ++i) {
        rgb[3 * i + 0] = lut[bgr[3 * i + 2]];
        rgb[3 * i + 1] = lut[bgr[3 * i + 1]];
        rgb[3 * i + 2] = lut[bgr[3 * i + 0]];
    }
}

/* out[n,y,x,o] = sum_{ky,kx,c} in[n,y+ky,x+kx,c] * w[ky,kx,c,o]; float32 accumulate */
API void rn_ref_conv2d_valid(const float* in, const float* w, float* out,
                             int n, int h, int wd, int cin, int cout, int kh, int kw) {
    const int ho = h - kh + 1, wo = wd - kw + 1;
#pragma omp parallel for collapse(2) schedule(static)
    for (int b = 0; b < n; ++b) {
        for (int y = 0; y < ho; ++y) {
            float acc[512];
            for (int x = 0; x < wo; ++x) {
                for (int o = 0; o < cout; ++o) acc[o] = 0.f;
                for (int ky = 0; ky < kh; ++ky) {
                    for (int kx = 0; kx < kw; ++kx) {
                        const float* ip = in + (((int64_t)b * h + (y + ky)) * wd + (x + kx)) * cin;
                        const float* wp = w + ((int64_t)(ky * kw + kx) * cin) * cout;
                        for (int c = 0; c < cin; ++c) {
                            const float v = ip[c];
                            const float* wr = wp + (int64_t)c * cout;
                            for (int o = 0; o < cout; ++o) acc[o] += v * wr[o];
                        }
                    }
                }
                float* op = out + (((int64_t)b * ho + y) * wo + x) * cout;
                for (int o = 0; o < cout; ++o) op[o] = acc[o];
            }
        }
    }
}

API void rn_ref_relu6(float* x, int64_t n) {
#pragma omp parallel for schedule(static)
    for (int64_t i = 0; i < n; ++i) {
        float v = x[i];
        v = v < 0.f ? 0.f : v;
        x[i] = v > 6.f ? 6.f : v;
    }
}

/* VALID average pool, window summed in raster order, divided by k*k */
API void rn_ref_avg_pool_valid(const float* in, float* out, int n, int h, int w, int c, int k, int s) {
    const int ho = (h - k) / s + 1, wo = (w - k) / s + 1;
    const float div = (float)(k * k);
#pragma omp parallel for collapse(2) schedule(static)
    for (int b = 0; b < n; ++b) {
        for (int y = 0; y < ho; ++y) {
            for (int x = 0; x < wo; ++x) {
                float* op = out + (((int64_t)b * ho + y) * wo + x) * c;
                for (int ch = 0; ch < c; ++ch) op[ch] = 0.f;
                for (int ky = 0; ky < k; ++ky)
                    for (int kx = 0; kx < k; ++kx) {
                        const float* ip = in + (((int64_t)b * h + (y * s + ky)) * w + (x * s + kx)) * c;
                        for (int ch = 0; ch < c; ++ch) op[ch] += ip[ch];
                    }
                for (int ch = 0; ch < c; ++ch) op[ch] = op[ch] / div;
            }
        }
    }
}

/* y = (x - mean) * (rsqrt(var + eps) * gamma) + beta, per channel (last dim) */
API void rn_ref_fused_bn_infer(const float* in, float* out, int64_t npix, int c,
                               const float* gamma, const float* beta,
                               const float* mean, const float* var, float eps) {
    float inv[512];
    for (int ch = 0; ch < c; ++ch) inv[ch] = (1.0f / sqrtf(var[ch] + eps)) * gamma[ch];
#pragma omp parallel for schedule(static)
    for (int64_t p = 0; p < npix; ++p) {
        const float* ip = in + p * c;
        float* op = out + p * c;
        for (int ch = 0; ch < c; ++ch) op[ch] = (ip[ch] - mean[ch]) * inv[ch] + beta[ch];
    }
}

/* legacy (TF<=1.13) bilinear resize, align_corners=False, no half-pixel centres */
static void interp_weights(int out_size, int in_size, int64_t* lo, int64_t* hi, float* lerp) {
    const float scale = (float)in_size / (float)out_size;
    for (int i = out_size - 1; i >= 0; --i) {
        const float src = (float)i * scale;
        lo[i] = (int64_t)src;
        hi[i] = lo[i] + 1 < in_size - 1 ? lo[i] + 1 : in_size - 1;
        lerp[i] = src - (float)lo[i];
    }
}

API void rn_ref_resize_bilinear(const float* in, float* out, int n, int h, int w, int c, int oh, int ow) {
    int64_t* ylo = (int64_t*)malloc(sizeof(int64_t) * (size_t)(2 * oh + 2 * ow));
    int64_t* yhi = ylo + oh;
    int64_t* xlo = yhi + oh;
    int64_t* xhi = xlo + ow;
    float* yl = (float*)malloc(sizeof(float) * (size_t)(oh + ow));
    float* xl = yl + oh;
    interp_weights(oh, h, ylo, yhi, yl);
    interp_weights(ow, w, xlo, xhi, xl);
#pragma omp parallel for collapse(2) schedule(static)
    for (int b = 0; b < n; ++b) {
        for (int y = 0; y < oh; ++y) {
            const float* r0 = in + ((int64_t)b * h + ylo[y]) * w * c;
            const float* r1 = in + ((int64_t)b * h + yhi[y]) * w * c;
            for (int x = 0; x < ow; ++x) {
                float* op = out + (((int64_t)b * oh + y) * ow + x) * c;
                for (int ch = 0; ch < c; ++ch) {
                    const float tl = r0[xlo[x] * c + ch], tr = r0[xhi[x] * c + ch];
                    const float bl = r1[xlo[x] * c + ch], br = r1[xhi[x] * c + ch];
                    const float top = tl + (tr - tl) * xl[x];
                    const float bottom = bl + (br - bl) * xl[x];
                    op[ch] = top + (bottom - top) * yl[y];
                }
            }
        }
    }
    free(ylo);
    free(yl);
}

API void rn_ref_add(const float* a, const float* b, float* out, int64_t n) {
#pragma omp parallel for schedule(static)
    for (int64_t i = 0; i < n; ++i) out[i] = a[i] + b[i];
}

/* out[m,n] = sum_k a[m,k] * b[k,n] */
API void rn_ref_matmul(const float* a, const float* b, float* out, int m, int k, int n) {
#pragma omp parallel for schedule(static)
    for (int i = 0; i < m; ++i) {
        float* op = out + (int64_t)i * n;
        for (int j = 0; j < n; ++j) op[j] = 0.f;
        for (int kk = 0; kk < k; ++kk) {
            const float v = a[(int64_t)i * k + kk];
            for (int j = 0; j < n; ++j) op[j] += v * b[(int64_t)kk * n + j];
        }
    }
}

API void rn_ref_bias_add(float* x, const float* bias, int m, int n) {
    for (int i = 0; i < m; ++i)
        for (int j = 0; j < n; ++j) x[(int64_t)i * n + j] += bias[j];
}

/* tf.nn.batch_normalization: inv = rsqrt(var+eps)*gamma; y = x*inv + (beta - mean*inv) */
API void rn_ref_bn_2d(const float* in, float* out, int m, int n, const float* gamma, const float* beta,
                      const float* mean, const float* var, float eps) {
    for (int j = 0; j < n; ++j) {
        const float inv = (1.0f / sqrtf(var[j] + eps)) * gamma[j];
        const float shift = beta[j] - mean[j] * inv;
        for (int i = 0; i < m; ++i) out[(int64_t)i * n + j] = in[(int64_t)i * n + j] * inv + shift;
    }
}

API void rn_ref_softmax(const float* in, float* out, int m, int n) {
    for (int i = 0; i < m; ++i) {
        const float* ip = in + (int64_t)i * n;
        float* op = out + (int64_t)i * n;
        float mx = ip[0];
        for (int j = 1; j < n; ++j) mx = ip[j] > mx ? ip[j] : mx;
        float sum = 0.f;
        for (int j = 0; j < n; ++j) {
            op[j] = expf(ip[j] - mx);
            sum += op[j];
        }
        for (int j = 0; j < n; ++j) op[j] = op[j] / sum;
    }
}

API void rn_ref_argmax(const float* in, int64_t* out, int m, int n) {
    for (int i = 0; i < m; ++i) {
        const float* ip = in + (int64_t)i * n;
        int best = 0;
        for (int j = 1; j < n; ++j)
            if (ip[j] > ip[best]) best = j;
        out[i] = best;
    }
}
