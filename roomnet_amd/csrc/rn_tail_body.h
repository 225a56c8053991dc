// Device code of the tail of the 16-bit path (the last two 16 -> 16 conv stages + flatten + dense head + softmax / argmax,
// reference network.py:230-237, :44-45), shared by tail_kernel (rn_tail.hip: a launch of its own) and the fused back end
// (rn_backend.hip: stages 6 -> 7 -> 8 -> 9 -> head in one launch).  See rn_tail.hip for the description.
#pragma once
#include "rn_fused.h"
#include "rn_stage.h"

namespace rnk {


constexpr int T_MAXSIDE = 34;              // widest stage input: its conv row (side - 2) must fit one 32-column MFMA tile
constexpr int T_W_LDS = 3072;              // floats of dense kernels staged in LDS (all four at 224: 2736)

struct TailStage {
    const i32x4* wfrag;                    // [9 taps][64 lanes] A fragments (16 input channels = one 16-deep K chunk per tap)
    const float* mean;
    const float* inv;
    const float* beta;
    unsigned short* out;                   // [N, So, So, 16]
    int Si, So;                            // input side, output side = (Si - 2 - 4) / 2 + 1
};

struct TailArgs {
    const unsigned short* in;              // [N, S, S, 16]: output of the block's first step (input AND skip tensor)
    TailStage a, b;                        // the block's second and third step
    const float* mean2;                    // second BN of the residual step
    const float* inv2;
    const float* beta2;
    const int32_t* rlo;                    // legacy bilinear tables a.Si -> b.So
    const int32_t* rhi;
    const float* rlerp;
    HeadArgs head;
    float* probs;
    int64_t* ids;
};

// one conv3x3 (16 -> 16) + ReLU6 + avg-pool 4/2 + BN [+ residual + BN] stage for the pooled rows [yo_a, yo_b) of this wave.
// src: LDS image [Si][Si][16] of 16-bit values; dst: LDS image [So][So][16]; skip: LDS image of the block's first output
template <int DT, bool RES>
__device__ __forceinline__ void tail_stage(const unsigned short* src, unsigned short* dst, const TailStage& st, const TailArgs& a,
                                           const unsigned short* skip, int skip_side, int img, int yo_a, int yo_b, int lane, const float* ttab) {
    constexpr int PK = 4, PS = 2, RING = 3;
    const int r = lane & 31, hh = lane >> 5;
    const int Si = st.Si, So = st.So;
    if (yo_a >= yo_b) return;
    i32x4 wreg[9];
#pragma unroll
    for (int t = 0; t < 9; ++t) wreg[t] = st.wfrag[t * 64 + lane];
    const int xo = r / PS;
    const bool lane_out = (r % PS == 0 && r <= 32 - PK) && xo < So;
    int rx_lo = 0, rx_hi = 0;
    float rx_l = 0.f;
    if constexpr (RES) {
        const int xq = min(xo, So - 1);
        rx_lo = a.rlo[xq];
        rx_hi = a.rhi[xq];
        rx_l = a.rlerp[xq];
    }
    // the per-channel tables come from an LDS copy (`ttab`: [mean | inv | beta] of this stage, then of the second BN), staged by
    // tail_phase in front of the conv phases: at their point of use as global loads they sat behind the output stores (the
    // compiler may not move a load across a store it cannot prove disjoint) and every emitted row paid several dependent
    // global round trips (~1 us each on a kernel that is nothing but latency); held in registers they cost 48 VGPRs, which
    // the 11-wave back-end kernel (168 per lane) does not have
    // vertical interpolation of the first pooled row; the next row's is fetched right behind each emitted row
    [[maybe_unused]] float yl_n = 0.f;
    [[maybe_unused]] int ylo_n = 0, yhi_n = 0;
    if constexpr (RES) {
        yl_n = a.rlerp[yo_a];
        ylo_n = a.rlo[yo_a];
        yhi_n = a.rhi[yo_a];
    }
    float vring[RING][16];
#pragma unroll
    for (int i = 0; i < RING; ++i)
#pragma unroll
        for (int g = 0; g < 16; ++g) vring[i][g] = 0.f;
    const int yc0 = yo_a * PS;
    const int nconv = (yo_b - yo_a - 1) * PS + PK;
    int boff[3];
#pragma unroll
    for (int kx = 0; kx < 3; ++kx) boff[kx] = min(r + kx, Si - 1) * 16 + hh * 8;     // element offset inside a row
    for (int it = 0; it < nconv; ++it) {
        f32x16 acc;
#pragma unroll
        for (int g = 0; g < 16; ++g) acc[g] = 0.f;
#pragma unroll
        for (int tap = 0; tap < 9; ++tap) {
            const int ky = tap / 3, kx = tap % 3;
            const i32x4 b = *reinterpret_cast<const i32x4*>(src + (yc0 + it + ky) * Si * 16 + boff[kx]);
            acc = mfma32<DT>(wreg[tap], b, acc);
        }
        const int lrow = it;
        const bool emit = lrow >= PK - 1 && ((lrow - (PK - 1)) % PS) == 0;
        const int yo = yo_a + (lrow - (PK - 1)) / PS;
        float hs[16];
#pragma unroll
        for (int g = 0; g < 16; ++g) {
            const float v = relu6f(acc[g]);
            const float t = v + lane_next(v);
            hs[g] = t + lane_next(lane_next(t));
        }
        float tot[16];
#pragma unroll
        for (int g = 0; g < 16; ++g) {
            float t = vring[0][g];
#pragma unroll
            for (int i = 1; i < RING; ++i) t += vring[i][g];
            tot[g] = t + hs[g];
#pragma unroll
            for (int i = 0; i + 1 < RING; ++i) vring[i][g] = vring[i + 1][g];
            vring[RING - 1][g] = hs[g];
        }
        if (emit) {
            constexpr float inv_area = 1.0f / static_cast<float>(PK * PK);
            float yl = 0.f;
            const unsigned short* sk0 = nullptr;
            const unsigned short* sk1 = nullptr;
            if constexpr (RES) {
                yl = yl_n;
                sk0 = skip + ylo_n * skip_side * 16;
                sk1 = skip + yhi_n * skip_side * 16;
                const int yn = min(yo + 1, So - 1);
                yl_n = a.rlerp[yn];
                ylo_n = a.rlo[yn];
                yhi_n = a.rhi[yn];
            }
            unsigned short* orow = st.out + ((static_cast<int64_t>(img) * So + yo) * So + xo) * 16;
#pragma unroll
            for (int g = 0; g < 2; ++g) {
                const int c0 = 4 * hh + 8 * g;
                const f32x4 mean = *reinterpret_cast<const f32x4*>(ttab + c0), inv = *reinterpret_cast<const f32x4*>(ttab + 16 + c0),
                            beta = *reinterpret_cast<const f32x4*>(ttab + 32 + c0);
                float y[4];
#pragma unroll
                for (int j = 0; j < 4; ++j) y[j] = (tot[4 * g + j] * inv_area - mean[j]) * inv[j] + beta[j];
                if constexpr (RES) {
                    if (lane_out) {
                        const f32x4 tl = unpack4<DT>(*reinterpret_cast<const uint2*>(sk0 + rx_lo * 16 + c0));
                        const f32x4 tr = unpack4<DT>(*reinterpret_cast<const uint2*>(sk0 + rx_hi * 16 + c0));
                        const f32x4 bl = unpack4<DT>(*reinterpret_cast<const uint2*>(sk1 + rx_lo * 16 + c0));
                        const f32x4 br = unpack4<DT>(*reinterpret_cast<const uint2*>(sk1 + rx_hi * 16 + c0));
                        const f32x4 mean2 = *reinterpret_cast<const f32x4*>(ttab + 48 + c0), inv2 = *reinterpret_cast<const f32x4*>(ttab + 64 + c0),
                                    beta2 = *reinterpret_cast<const f32x4*>(ttab + 80 + c0);
#pragma unroll
                        for (int j = 0; j < 4; ++j) {
                            const float top = tl[j] + (tr[j] - tl[j]) * rx_l;
                            const float bot = bl[j] + (br[j] - bl[j]) * rx_l;
                            const float rs = top + (bot - top) * yl;
                            y[j] = ((y[j] + rs) - mean2[j]) * inv2[j] + beta2[j];
                        }
                    }
                }
                if (lane_out) {
                    const uint2 pk = pack4<DT>(y[0], y[1], y[2], y[3]);
                    *reinterpret_cast<uint2*>(orow + c0) = pk;
                    *reinterpret_cast<uint2*>(dst + (yo * So + xo) * 16 + c0) = pk;
                }
            }
        }
    }
}

// Everything behind the staging of the block's first output `xin` [S][S][16] (LDS) and of the dense kernels `wl` (LDS, offsets
// w_off, -1 = read from global): the two conv stages on the first four waves of the workgroup, the dense chain, softmax and
// argmax.  Called by all of the workgroup's first 256 threads; contains workgroup barriers.
template <int DT>
__device__ __forceinline__ void tail_phase(const TailArgs& a, const unsigned short* xin, unsigned short* xa, unsigned short* xb, float* buf0,
                                           float (*small)[64], const float* wl, const int (&w_off)[RN_MAX_DENSE], float* ttab, int img, int tid) {
    const int lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int S = a.a.Si;
    // (every barrier of this phase is lds_barrier(): the phases talk through LDS only; __syncthreads() also drained the stage
    //  outputs' and taps' global stores -- a write acknowledgement of 1-2 us at each of its eight barriers)
    // per-channel tables -> LDS: stage a at ttab[0..47], stage b at ttab[48..95] followed by its second BN at ttab[96..143]
    if (tid < 144) {
        const int t = tid / 16, c = tid % 16;
        const float* srcs[9] = {a.a.mean, a.a.inv, a.a.beta, a.b.mean, a.b.inv, a.b.beta, a.mean2, a.inv2, a.beta2};
        ttab[tid] = srcs[t][c];
    }
    // the dense layers' per-output constants of this thread: fetched now, used after the conv phases (the first HD_REG layers:
    // 3 registers each, live across both conv phases of a kernel that has 168 per lane; deeper heads fetch theirs at the layer)
    constexpr int HD_REG = 4;
    float hd_bias[HD_REG], hd_inv[HD_REG], hd_shift[HD_REG];
#pragma unroll
    for (int d = 0; d < HD_REG; ++d) {
        const bool on = d < a.head.n_dense && tid < a.head.nout[d];
        hd_bias[d] = on && a.head.bias[d] ? a.head.bias[d][tid] : 0.f;
        hd_inv[d] = on && a.head.inv[d] ? a.head.inv[d][tid] : 0.f;
        hd_shift[d] = on && a.head.inv[d] ? a.head.shift[d][tid] : 0.f;
    }
    lds_barrier();
    // ---- second and third step of the block: pooled rows dealt to the four waves
    {
        const int per = (a.a.So + 3) / 4;
        tail_stage<DT, false>(xin, xa, a.a, a, nullptr, 0, img, min(wave * per, a.a.So), min((wave + 1) * per, a.a.So), lane, ttab);
    }
    lds_barrier();
    {
        const int per = (a.b.So + 3) / 4;
        tail_stage<DT, true>(xa, xb, a.b, a, xin, S, img, min(wave * per, a.b.So), min((wave + 1) * per, a.b.So), lane, ttab + 48);
    }
    lds_barrier();
    // ---- flatten + dense chain + softmax + argmax: head_kernel's arithmetic, wave 0 computes
    const HeadArgs& h = a.head;
    const int nin0 = h.nin[0];
    for (int i = tid; i < nin0; i += 256) buf0[i] = from16<DT>(xb[i]);
    wait_vmcnt<0>();          // the dense kernels' LDS-DMA of this wave (tail_stage_dense_dma) has landed; the barrier publishes it
    lds_barrier();
    const float* cur = buf0;
#pragma unroll
    for (int d = 0; d < RN_MAX_DENSE; ++d) {       // (unrolled: the per-layer constants above stay in registers)
        if (d >= h.n_dense) break;
        const int nin = h.nin[d], nout = h.nout[d];
        float* dst = small[d & 1];
        if (tid < nout) {
            float v = 0.f;
            const float* wd = w_off[d] >= 0 ? wl + w_off[d] : h.w[d];
            // the same fma chain over k = 0 .. nin - 1, its operands read eight steps ahead of their use (one step at a time every
            // fma waited for its own two LDS reads: 2 500 cycles for the 64-input layer; in-kernel stamps)
            int k = 0;
            for (; k + 8 <= nin; k += 8) {
                float cv[8], wv[8];
#pragma unroll
                for (int j = 0; j < 8; ++j) {
                    cv[j] = cur[k + j];
                    wv[j] = wd[(k + j) * nout + tid];
                }
#pragma unroll
                for (int j = 0; j < 8; ++j) v = fmaf(cv[j], wv[j], v);
            }
            for (; k < nin; ++k) v = fmaf(cur[k], wd[k * nout + tid], v);
            if (h.bias[d]) v = __fadd_rn(v, d < HD_REG ? hd_bias[d] : h.bias[d][tid]);
            if (h.tap_mm[d]) h.tap_mm[d][static_cast<int64_t>(img) * nout + tid] = v;
            v = fminf(fmaxf(v, 0.f), 6.f);
            if (h.tap_relu[d]) h.tap_relu[d][static_cast<int64_t>(img) * nout + tid] = v;
            if (h.inv[d]) {
                v = __fadd_rn(__fmul_rn(v, d < HD_REG ? hd_inv[d] : h.inv[d][tid]), d < HD_REG ? hd_shift[d] : h.shift[d][tid]);
                if (h.tap_bn[d]) h.tap_bn[d][static_cast<int64_t>(img) * nout + tid] = v;
            }
            dst[tid] = v;
        }
        lds_barrier();
        cur = dst;
    }
    if (wave != 0) return;
    const int nc = h.nout[h.n_dense - 1];
    const float logit = lane < nc ? cur[lane] : -INFINITY;
    // butterflies over the smallest power-of-two group of lanes that holds the nc classes (lanes past it hold the neutral element
    // and are never read back): three exchange steps instead of six for the six classes
    int off0 = 1;
    while (off0 < nc) off0 <<= 1;
    off0 >>= 1;
    float mx = logit;
    for (int off = off0; off > 0; off >>= 1) mx = fmaxf(mx, __shfl_xor(mx, off));
    const float e = lane < nc ? expf(logit - mx) : 0.f;
    float sum = e;
    for (int off = off0; off > 0; off >>= 1) sum += __shfl_xor(sum, off);
    const float p = e / sum;
    if (lane < nc) a.probs[static_cast<int64_t>(img) * nc + lane] = p;
    float bestv = lane < nc ? p : -1.f;
    int besti = lane < nc ? lane : 0x7fffffff;
    for (int off = off0; off > 0; off >>= 1) {
        const float ov = __shfl_xor(bestv, off);
        const int oi = __shfl_xor(besti, off);
        if (ov > bestv || (ov == bestv && oi < besti)) {
            bestv = ov;
            besti = oi;
        }
    }
    if (lane == 0) a.ids[img] = besti;
}

// dense kernels -> LDS (all that fit T_W_LDS floats), by `nthreads` threads
__device__ __forceinline__ void tail_stage_dense(const HeadArgs& h, float* wl, int (&w_off)[RN_MAX_DENSE], int tid, int nthreads) {
    int off = 0;
#pragma unroll
    for (int d = 0; d < RN_MAX_DENSE; ++d) {
        w_off[d] = -1;
        if (d >= h.n_dense) continue;
        const int cnt = h.nin[d] * h.nout[d];
        if (off + cnt <= T_W_LDS) {
            w_off[d] = off;
            // 16-byte pieces, every load of a thread issued before its first LDS store (element by element each store waited for
            // its own load: eleven dependent L2 round trips per thread in front of the tail kernel's first barrier)
            if (cnt % 4 == 0 && off % 4 == 0 && (reinterpret_cast<uintptr_t>(h.w[d]) & 15) == 0) {
                const f32x4* src = reinterpret_cast<const f32x4*>(h.w[d]);
                f32x4* dst = reinterpret_cast<f32x4*>(wl + off);
                const int n4 = cnt / 4;
                for (int i0 = 0; i0 < n4; i0 += 4 * nthreads) {
                    f32x4 t[4];
#pragma unroll
                    for (int u = 0; u < 4; ++u)
                        if (i0 + u * nthreads + tid < n4) t[u] = src[i0 + u * nthreads + tid];
#pragma unroll
                    for (int u = 0; u < 4; ++u)
                        if (i0 + u * nthreads + tid < n4) dst[i0 + u * nthreads + tid] = t[u];
                }
            } else {
                for (int i = tid; i < cnt; i += nthreads) wl[off + i] = h.w[d][i];
            }
            off += cnt;
        }
    }
}

// The same staging by LDS-DMA (global_load_lds_dwordx4: no registers, no LDS store instructions): `nwaves` waves, wave `wave`
// takes every nwaves-th kilobyte.  The caller retires it (vmcnt) before the first read.  In the back end it is issued in front
// of the weight-fragment loads and retires WITH them at the prologue's `s_waitcnt vmcnt(0)` (the row loop's counted waits start
// from an empty queue), so it does not ride under the row loop: what it saves against register loads + LDS stores in front of
// the loop (every wave ~10 000 cycles there, in-kernel stamps) is the second round trip and the store issue, ~2 us of the
// launch.  Layers that are not 16-byte granular fall back to tail_stage_dense's element loop.
__device__ __forceinline__ void tail_stage_dense_dma(const HeadArgs& h, float* wl, int (&w_off)[RN_MAX_DENSE], int wave, int lane, int nwaves) {
    int off = 0;
#pragma unroll
    for (int d = 0; d < RN_MAX_DENSE; ++d) {
        w_off[d] = -1;
        if (d >= h.n_dense) continue;
        const int cnt = h.nin[d] * h.nout[d];
        if (off + cnt <= T_W_LDS) {
            w_off[d] = off;
            if (cnt % 4 == 0 && off % 4 == 0 && (reinterpret_cast<uintptr_t>(h.w[d]) & 15) == 0) {
                const int n4 = cnt / 4;
                const char* src = reinterpret_cast<const char*>(h.w[d]);
                char* dst = reinterpret_cast<char*>(wl + off);
                for (int i0 = wave * 64; i0 < n4; i0 += nwaves * 64) {
                    const unsigned long long mask = n4 - i0 >= 64 ? ~0ull : (1ull << (n4 - i0)) - 1ull;
                    dma16_masked(src + static_cast<size_t>(i0 + lane) * 16, dst + static_cast<size_t>(i0) * 16, mask);
                }
            } else {
                for (int i = wave * 64 + lane; i < cnt; i += nwaves * 64) wl[off + i] = h.w[d][i];
            }
            off += cnt;
        }
    }
}

}  // namespace rnk

void rn_tail_fill_args(rn_handle* h, const rnk::i32x4* wfrag_a, const rnk::i32x4* wfrag_b, const HeadArgs& head, float* d_probs, int64_t* d_ids,
                       rnk::TailArgs* out);
