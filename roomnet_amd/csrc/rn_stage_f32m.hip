// Float32 conv stages on the matrix cores: the throughput path of RN_DTYPE_F32 handles (reference network.py:28: the
// graph's arithmetic type is float32; :183-203 one conv_block step).
//
//   in fp32 [N, H, W, CIN] -> conv3x3 VALID -> ReLU6 -> [avg-pool k/s] -> BN [-> + legacy-bilinear(skip) -> BN] = out fp32
//
// One launch per stage, every operation of the stage in fp32 exactly as the per-node kernels of rn_kernels_f32.hip state
// them (un-contracted BN and residual expressions); only the ORDER of the convolution's K sum and of the pooling window sum
// differs from the per-node path (within the 1e-4-of-abs-max per-node tolerance of BASELINE config 2; tests/test_hip_f32.py).
// The per-node path stays what RN_FLAG_TAPS handles run: every graph node tappable.
//
// Kernel: the row-streaming implicit GEMM of stage_mfma_kernel (rn_fused.hip) with v_mfma_f32_32x32x2_f32 -- fp32 in, fp32
// accumulate, bitwise an fmaf chain, 64 FLOP / clock / SIMD = 1/16 of the bf16 rate (MI355X_MICROARCH.md): the stage is bound by
// the matrix pipe, not by memory (fp32 is priced against the 157 TFLOP/s matrix-fp32 roofline).
//   * workgroup = image x band of output rows x block of columns x group of 32 couts; one wave = one tile of 32 conv columns;
//   * the last three input rows live in an LDS ring (fp32 NHWC, 16-byte chunks XOR-swizzled inside a pixel); the B operand of
//     FOUR MFMAs is one ds_read_b128: lane (pixel, h) reads channels 8 q + 4 h .. + 3 of a tap, MFMA i of the four contracts
//     channels {8 q + i, 8 q + 4 + i};
//   * weights sit in LDS in fragment order (host-packed): one lane-linear ds_read_b128 = the A operands of the same four MFMAs;
//   * accumulator layout = pixel on the lane, 16 couts in registers: ReLU6 per register, horizontal pool by DPP wave shifts,
//     vertical pool in a register ring, BN / residual / store per group of 4 consecutive couts (one 16-byte store).
#include "rn_fused.h"
#include "rn_stage.h"

#include <atomic>
#include <cstring>

using namespace rnk;

namespace {

struct F32StageArgs {
    const float* in;              // [N, H, W, CIN]
    float* out;                   // [N, Ho, Wo, COUT]
    const f32x4* wfrag;           // [9 CIN / 8][CT][64 lanes] x 4 floats
    const float* bn_mean;
    const float* bn_inv;
    const float* bn_beta;
    const float* skip;            // [N, Ss, Ss, COUT] (residual stages)
    const float* bn2_mean;
    const float* bn2_inv;
    const float* bn2_beta;
    const int32_t* rlo;
    const int32_t* rhi;
    const float* rlerp;
    const float* cinit;           // [COUT] start value of the accumulators (KQL < CIN / 8: the frozen input channels' contribution)
    int H, W, Ho, Wo, Ss;
    int rows_per_band, n_bands, n_colblocks, n_ctg, npt;
    int ringcols;                 // columns of a ring row: (npt - 1) tile strides + 34, but never more than the input row
};

__device__ __forceinline__ f32x16 mfma_f32(float a, float b, f32x16 c) { return __builtin_amdgcn_mfma_f32_32x32x2f32(a, b, c, 0, 0, 0); }

// NSL = ring slots: 4 (three live rows + the one being filled behind the compute) or 3 (the fill waits behind a second barrier:
// the 64-channel stages, whose weights take 72 KB of the LDS)
// LPT = 16-byte chunks of a ring row one thread fetches (the host sizes the workgroup so that ringcols * CIN / 4 <= LPT * threads)
// KS = waves per pixel tile: 2 = the tile's K sum is split by channel halves over two waves (the 64 -> 128 stage has two tiles per
// row: two waves would leave half the CU's matrix pipes idle); the second wave's partial accumulators cross through LDS
// KQL = chunk pairs (8 input channels each) per tap that are contracted: CIN / 8, or fewer where the producer's BN freezes the last
// CIN - 8 KQL channels to constants (rn_create relabels them to the end): a VALID convolution sees every tap of every pixel, so
// they add one constant per cout -- a.cinit, the accumulators' start value
template <int CIN, int COUT, int PK, int PS, bool RES, int NSL, int LPT, int KS, int KQL>
__global__ __launch_bounds__(512) void stage_f32m_kernel(const F32StageArgs a) {
    constexpr int CP = CIN / 4;                          // 16-byte chunks (4 floats) per pixel
    constexpr int KQ = CIN / 8;                          // chunk pairs per tap = ds_read_b128 per tap and lane
    static_assert(KQL == KQ || (KQL >= 1 && KQL < KQ && KS == 1 && COUT % 32 == 0), "input-channel fold: one wave per tile");
    constexpr int KC = 9 * KQ;                           // weight fragments (1 KB each) per 32-cout tile
    constexpr int CT = (COUT + 31) / 32;
    constexpr int NG = COUT >= 32 ? 4 : COUT / 8;
    constexpr int TSTRIDE = tile_stride(PK, PS);
    constexpr int NOUT_T = tile_nout(PK, PS);
    constexpr int RING = PK ? PK - 1 : 0;
    constexpr int PIXB = CIN * 4;
    constexpr int LPT_MAX = LPT;
    static_assert(CIN % 8 == 0 && COUT % 8 == 0, "channels must be multiples of 8");

    extern __shared__ __attribute__((aligned(16))) char smem[];
    const int tid = threadIdx.x;
    const int lane = tid & 63;
    const int wave_all = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int wave = wave_all / KS, kh = wave_all % KS;      // pixel tile / K half of this wave
    const int r = lane & 31, hh = lane >> 5;
    const int nthreads = blockDim.x;
    const int npt = a.npt;

    int bid = blockIdx.x;
    const int ctg = bid % a.n_ctg;
    bid /= a.n_ctg;
    const int cb = bid % a.n_colblocks;
    const int band = bid / a.n_colblocks;
    const int n = blockIdx.y;

    // (a ring row never holds more columns than the input has: the last tile's lanes right of the image then read past the row
    //  -- other rows' activations, or zero past the workgroup's LDS -- and only feed output columns that are not stored)
    const int ringcols = a.ringcols;
    const int rowbytes = ringcols * PIXB;
    char* const wl = smem;                                  // weights [KC][64] x 16 B of this workgroup's cout tile
    float* const tabs = reinterpret_cast<float*>(smem + KC * 1024);      // [6][32] per-channel tables of this cout tile
    char* const ring = smem + KC * 1024 + 1024;             // NSL rows
    [[maybe_unused]] char* const pbuf = ring + NSL * (a.ringcols * CIN * 4) + wave * 4096 + lane * 16;      // KS == 2: partial sums, 4 KB per tile

    const int yo0 = band * a.rows_per_band;
    const int yo1 = min(a.Ho, yo0 + a.rows_per_band);
    const int yc0 = PK ? yo0 * PS : yo0;
    const int nconv = PK ? (yo1 - yo0 - 1) * PS + PK : (yo1 - yo0);
    const int nin = nconv + 2;
    const int x0c = cb * npt * TSTRIDE;
    const int xo_blk0 = PK ? x0c / PS : x0c;

    for (int i = tid; i < KC * 64; i += nthreads) reinterpret_cast<f32x4*>(wl)[i] = a.wfrag[((i >> 6) * CT + ctg) * 64 + (i & 63)];
    // per-channel tables of the tile's couts -> LDS (in registers they cost up to 96 VGPRs per lane)
    for (int i = tid; i < (KQL < KQ ? 7 : 6) * 32; i += nthreads) {
        const int t = i / 32, c = ctg * 32 + i % 32;
        const float* src = t == 0 ? a.bn_mean : t == 1 ? a.bn_inv : t == 2 ? a.bn_beta : t == 3 ? a.bn2_mean : t == 4 ? a.bn2_inv : t == 5 ? a.bn2_beta : a.cinit;
        tabs[i] = (c < COUT && (RES || t < 3 || t == 6)) ? src[c] : 0.f;
    }

    // ---- input-row loader: a thread owns up to LPT_MAX 16-byte chunks of a ring row
    const int nchunks = ringcols * CP;
    const float* const in_img = a.in + static_cast<int64_t>(n) * a.H * a.W * CIN;
    int ld_goff[LPT_MAX], ld_loff[LPT_MAX];
#pragma unroll
    for (int i = 0; i < LPT_MAX; ++i) {
        const int q = tid + i * nthreads;
        const int p = q / CP, c4 = q % CP;
        ld_loff[i] = q < nchunks ? (p * CP + (c4 ^ chunk_swz<CP>(p))) * 16 : -1;
        ld_goff[i] = (q < nchunks ? min(x0c + p, a.W - 1) : 0) * CIN + c4 * 4;      // columns past the edge feed discarded lanes
    }
    f32x4 pre[LPT_MAX];
    auto fetch_row = [&](int j) {
        const float* row = in_img + static_cast<int64_t>(yc0 + j) * a.W * CIN;
        // (unconditional: a thread without a chunk re-reads pixel 0 of the row; as predicated loads every one of them sat in a
        //  basic block of its own behind an s_waitcnt vmcnt(0) that also drained the epilogue's stores)
#pragma unroll
        for (int i = 0; i < LPT_MAX; ++i) pre[i] = *reinterpret_cast<const f32x4*>(row + ld_goff[i]);
    };
    auto store_row = [&](int j) {
        char* dst = ring + (j % NSL) * rowbytes;
#pragma unroll
        for (int i = 0; i < LPT_MAX; ++i)
            if (ld_loff[i] >= 0) *reinterpret_cast<f32x4*>(dst + ld_loff[i]) = pre[i];
    };
    for (int j = 0; j < 3; ++j) {
        fetch_row(j);
        store_row(j);
    }
    __syncthreads();

    const int xrel0 = wave * TSTRIDE + r;
    int boff[3], bswz[3];
#pragma unroll
    for (int kx = 0; kx < 3; ++kx) {
        boff[kx] = (xrel0 + kx) * PIXB;
        bswz[kx] = chunk_swz<CP>(xrel0 + kx);
    }
    // byte offset of the lane's B chunk inside a ring row, per kernel column and chunk pair of this wave (lane constants: computed
    // per group in the row loop they were two VALU instructions each -- and VALU instructions do not hide behind the fp32 MFMA,
    // tools/ubench/mfma_f32_valu.hip)
    constexpr bool BQ_PRE = KQ / KS <= 4;      // (the 64-channel one-wave-per-tile stage has no 24 registers to spare)
    [[maybe_unused]] int bq[3][BQ_PRE ? KQ / KS : 1];
    if constexpr (BQ_PRE) {
#pragma unroll
        for (int kx = 0; kx < 3; ++kx)
#pragma unroll
            for (int qi = 0; qi < KQ / KS; ++qi) {
                const int q = KS == 2 ? 2 * qi + kh : qi;
                bq[kx][qi] = boff[kx] + (((2 * q + hh) ^ bswz[kx]) << 4);
            }
    }
    const int xc = x0c + xrel0;
    const int xo = PK ? xc / PS : xc;
    const bool lane_out = (PK ? (r % PS == 0 && r <= 32 - PK) : true) && xo < a.Wo && (xo - xo_blk0) < npt * NOUT_T;
    const int cout_lane = ctg * 32 + 4 * hh;              // + 8 g + j

    int rx_lo = 0, rx_hi = 0;
    float rx_l = 0.f;
    if constexpr (RES) {
        const int xq = min(xo, a.Wo - 1);
        rx_lo = a.rlo[xq];
        rx_hi = a.rhi[xq];
        rx_l = a.rlerp[xq];
    }
    float vring[RING > 0 ? RING : 1][16];
#pragma unroll
    for (int i = 0; i < (RING > 0 ? RING : 1); ++i)
#pragma unroll
        for (int g = 0; g < 16; ++g) vring[i][g] = 0.f;

    const char* const wl_lane = wl + lane * 16;

    // one row step; PH = it mod 3 for the pooled stages without residual (RING_ROT: the row loop is unrolled by three): the three
    // rows of the vertical pooling ring are vring[PH] (oldest), vring[(PH + 1) % 3], vring[(PH + 2) % 3] -- the new row replaces
    // the oldest in place instead of shifting the ring (32 v_mov per row).  The residual stages keep the shifting form: unrolled
    // they spill.
    constexpr bool RING_ROT = RING == 3 && !RES;
    auto step = [&](auto PHC, int it) __attribute__((always_inline)) {
        [[maybe_unused]] constexpr int PH = decltype(PHC)::value;
        const bool have_next = it + 3 < nin;
        if (have_next) fetch_row(it + 3);
        // which output row this conv row completes; a residual stage fetches the four skip neighbours of the lane's pixel NOW, in
        // front of the row's MFMAs (at their point of use every emitted row waited for 16 dependent L2 round trips)
        bool emit;
        int yo;
        if constexpr (PK > 0) {
            emit = it >= PK - 1 && ((it - (PK - 1)) % PS) == 0;
            yo = yo0 + (it - (PK - 1)) / PS;
        } else {
            emit = true;
            yo = yo0 + it;
        }
        [[maybe_unused]] f32x4 sk_tl[NG], sk_tr[NG], sk_bl[NG], sk_br[NG];
        [[maybe_unused]] float yl = 0.f;
        if constexpr (RES) {
            if (emit && lane_out) {
                yl = a.rlerp[yo];
                const float* skn = a.skip + static_cast<int64_t>(n) * a.Ss * a.Ss * COUT;
                const float* sk0 = skn + static_cast<int64_t>(a.rlo[yo]) * a.Ss * COUT;
                const float* sk1 = skn + static_cast<int64_t>(a.rhi[yo]) * a.Ss * COUT;
#pragma unroll
                for (int g = 0; g < NG; ++g) {
                    const int c0 = cout_lane + 8 * g;
                    sk_tl[g] = *reinterpret_cast<const f32x4*>(sk0 + rx_lo * COUT + c0);
                    sk_tr[g] = *reinterpret_cast<const f32x4*>(sk0 + rx_hi * COUT + c0);
                    sk_bl[g] = *reinterpret_cast<const f32x4*>(sk1 + rx_lo * COUT + c0);
                    sk_br[g] = *reinterpret_cast<const f32x4*>(sk1 + rx_hi * COUT + c0);
                }
            }
        }

        f32x16 acc = {0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f};      // (the first MFMA's inline-zero C operand)
        if constexpr (KQL < KQ) {
#pragma unroll
            for (int g = 0; g < 4; ++g) {
                const f32x4 c = *reinterpret_cast<const f32x4*>(tabs + 192 + 4 * hh + 8 * g);       // couts cout_lane + 8 g .. + 3
#pragma unroll
                for (int j = 0; j < 4; ++j) acc[4 * g + j] = c[j];
            }
        }
        const char* rowp[3];
#pragma unroll
        for (int ky = 0; ky < 3; ++ky) rowp[ky] = ring + ((it + ky) % NSL) * rowbytes;
#pragma unroll
        for (int tap = 0; tap < 9; ++tap) {
            const int ky = tap / 3, kx = tap % 3;
#pragma unroll
            for (int qi = 0; qi < (KQL < KQ ? KQL : KQ / KS); ++qi) {
                const int q = KS == 2 ? 2 * qi + kh : qi;   // (wave-uniform: this wave's channel half)
                const f32x4 b = *reinterpret_cast<const f32x4*>(rowp[ky] + (BQ_PRE ? bq[kx][BQ_PRE ? qi : 0] : boff[kx] + (((2 * q + hh) ^ bswz[kx]) << 4)));
                const f32x4 wv = *reinterpret_cast<const f32x4*>(wl_lane + (tap * KQ + q) * 1024);
#pragma unroll
                for (int i = 0; i < 4; ++i) acc = mfma_f32(wv[i], b[i], acc);
            }
        }

        // ---------------- ReLU6 + horizontal pool sum (lanes) + vertical pool sum (register ring)
        if constexpr (KS == 2) {
            // the second wave of the tile hands its partial sums over; the first adds them (fixed order) and runs the epilogue
            if (kh == 1) {
#pragma unroll
                for (int i = 0; i < 4; ++i) *reinterpret_cast<f32x4*>(pbuf + i * 1024) = f32x4{acc[4 * i], acc[4 * i + 1], acc[4 * i + 2], acc[4 * i + 3]};
            }
            lds_barrier();
            if (kh == 0) {
#pragma unroll
                for (int i = 0; i < 4; ++i) {
                    const f32x4 p = *reinterpret_cast<const f32x4*>(pbuf + i * 1024);
#pragma unroll
                    for (int j = 0; j < 4; ++j) acc[4 * i + j] += p[j];
                }
            }
        }
        float tot[16];
#pragma unroll
        for (int g = 0; g < 16; ++g) {
            const float v = relu6f(acc[g]);
            float hs;
            if constexpr (PK == 4) {
                const float t = v + lane_next(v);
                hs = t + lane_next(lane_next(t));
            } else {
                hs = v;
            }
            float s = hs;
            if constexpr (RING_ROT) {
                static_assert(!RING_ROT || RING == 3, "pool 4");
                float t = vring[PH % 3][g];                  // oldest first, as in the shifting form
#pragma unroll
                for (int i = 1; i < 3; ++i) t += vring[(PH + i) % 3][g];
                s = t + s;
                vring[PH % 3][g] = hs;
            } else if constexpr (RING > 0) {
                float t = vring[0][g];
#pragma unroll
                for (int i = 1; i < RING; ++i) t += vring[i][g];
                s = t + s;
#pragma unroll
                for (int i = 0; i + 1 < RING; ++i) vring[i][g] = vring[i + 1][g];
                vring[RING - 1][g] = hs;
            }
            tot[g] = s;
        }
        if (emit && kh == 0) {
            constexpr float inv_area = PK ? 1.0f / static_cast<float>(PK * PK) : 1.0f;      // pool 4: exact (a power of two)
            float* orow = a.out + ((static_cast<int64_t>(n) * a.Ho + yo) * a.Wo + xo) * COUT;
#pragma unroll
            for (int g = 0; g < NG; ++g) {
                const int c0 = cout_lane + 8 * g;
                const float* tl0 = tabs + 4 * hh + 8 * g;
                const f32x4 t_mean = *reinterpret_cast<const f32x4*>(tl0), t_inv = *reinterpret_cast<const f32x4*>(tl0 + 32),
                            t_beta = *reinterpret_cast<const f32x4*>(tl0 + 64);
                f32x4 y;
                // (x - mean) * inv + beta, un-contracted like bn_f32_kernel
#pragma unroll
                for (int j = 0; j < 4; ++j)
                    y[j] = __fadd_rn(__fmul_rn(__fsub_rn(__fmul_rn(tot[4 * g + j], inv_area), t_mean[j]), t_inv[j]), t_beta[j]);
                if constexpr (RES) {
                    if (lane_out) {
                        const f32x4 t_mean2 = *reinterpret_cast<const f32x4*>(tl0 + 96), t_inv2 = *reinterpret_cast<const f32x4*>(tl0 + 128),
                                    t_beta2 = *reinterpret_cast<const f32x4*>(tl0 + 160);
                        const f32x4 tl = sk_tl[g], tr = sk_tr[g], bl = sk_bl[g], br = sk_br[g];
#pragma unroll
                        for (int j = 0; j < 4; ++j) {
                            // resize_add_f32_kernel's expression, then the second BN
                            const float top = __fadd_rn(tl[j], __fmul_rn(__fsub_rn(tr[j], tl[j]), rx_l));
                            const float bot = __fadd_rn(bl[j], __fmul_rn(__fsub_rn(br[j], bl[j]), rx_l));
                            const float rs = __fadd_rn(top, __fmul_rn(__fsub_rn(bot, top), yl));
                            y[j] = __fadd_rn(__fmul_rn(__fsub_rn(__fadd_rn(y[j], rs), t_mean2[j]), t_inv2[j]), t_beta2[j]);
                        }
                    }
                }
                if (lane_out) *reinterpret_cast<f32x4*>(orow + c0) = y;
            }
        }

        // (lds_barrier: the waves talk through LDS only; the row's output stores stay in flight across it)
        if constexpr (NSL == 3) lds_barrier();            // everybody is past row `it` before its slot is refilled
        if (have_next) store_row(it + 3);
        lds_barrier();
    };
    if constexpr (RING_ROT) {
        for (int it = 0; it < nconv; it += 3) {
            step(IC<0>{}, it);
            if (it + 1 < nconv) step(IC<1>{}, it + 1);
            if (it + 2 < nconv) step(IC<2>{}, it + 2);
        }
    } else {
        for (int it = 0; it < nconv; ++it) step(IC<0>{}, it);
    }
}


// ---- 16-cout stages (128 -> 16 and 16 -> 16, avg-pool 4/2, no residual) on v_mfma_f32_16x16x4_f32: a 32 x 32 x 2 tile would
// idle half of its rows on 16 couts, and its weights (9 x 128 x 32 floats) would not fit the LDS next to the ring.  Same
// structure as stage_f32m_kernel: wave = tile of 16 conv columns (7 pooled columns: stride 14), lane = (pixel, k group), one
// ds_read_b128 of the ring is the B operand of four MFMAs (lane (pixel, k) reads channels 16 q + 4 k .. + 3 of a tap, MFMA i
// contracts channels {16 q + 4 k + i, k = 0..3}), weights in LDS in fragment order; the accumulator holds 4 couts of the
// lane's pixel: ReLU6, horizontal window by row-local DPP shifts, vertical window in a register ring, un-contracted BN, one
// 16-byte store.  KS = 3: the K sum of the 128-channel stage is split by KERNEL ROW over three waves per tile (one image row of
// the ring each), partial sums meet in LDS and are added in a fixed order by the first.
__device__ __forceinline__ f32x4 mfma_f32_16(float a, float b, f32x4 c) { return __builtin_amdgcn_mfma_f32_16x16x4f32(a, b, c, 0, 0, 0); }

// chunk swizzle of the 16-column kernel's ring: a ds_read_b128 lane group mixes two k groups here (lanes {0-3, 12-15} of k group 0 with
// {20-27} of k group 1, ...), so the 32-channel ring (8 chunks per pixel) takes ((pixel >> 1) & 3) << 1 -- every group reads 16 distinct
// 16-byte slots of the bank row at every tile offset (chunk_swz<8>, made for 32 pixels reading one chunk index, left 1.7 LDS cycles per
// read: 38 % conflict cycles, profiles/r5_c_f32_mfma_busy.txt)
template <int CP>
__device__ __forceinline__ int m16_swz(int pix) {
    if constexpr (CP == 8)
        return ((pix >> 1) & 3) << 1;
    else
        return chunk_swz<CP>(pix);
}

// PS = 1, CFZ = 16: the first 32 -> 32 stage (avg-pool 4/1) of a handle whose rn_create proved 16 of its output channels constant
// and relabelled them to the end: the 16 live couts are convolved (13 pooled columns per tile: stride 13), the other 16 channels of
// the 32-channel output pixel are written as their constant (beta) -- the tensor in HBM is complete, the next stage does not
// contract them (stage_f32m_kernel's KQL)
template <int CIN, int KS, int LPT, int PS = 2, int CFZ = 0>
__global__ __launch_bounds__(576) void stage_f32m16_kernel(const F32StageArgs a) {
    constexpr int COUT = 16, PK = 4;
    constexpr int NSL = CFZ > 0 ? 4 : 3;                 // ring slots: 4 = the next row lands behind the compute, one barrier per row
    constexpr int CP = CIN / 4;                          // 16-byte chunks per pixel
    constexpr int KG = CIN / 16;                         // 16-channel groups per tap = ds_read_b128 per tap and lane
    constexpr int KC = 9 * KG;                           // weight fragments (1 KB each)
    constexpr int TSTRIDE = PS == 2 ? 14 : 13, NOUT_T = PS == 2 ? 7 : 13;
    static_assert(PS == 1 || PS == 2, "pool stride");
    static_assert(CFZ == 0 || CFZ == 16, "frozen channels written beside the 16 live ones");
    constexpr int PIXB = CIN * 4;
    static_assert(CIN % 16 == 0 && (KS == 1 || KS == 3), "channel groups of 16; K split = one kernel row per wave");

    extern __shared__ __attribute__((aligned(16))) char smem[];
    const int tid = threadIdx.x;
    const int lane = tid & 63;
    const int wave_all = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int wave = wave_all / KS, kh = wave_all % KS;
    const int px = lane & 15, kq = lane >> 4;
    const int nthreads = blockDim.x;
    const int npt = a.npt;
    const int cb = blockIdx.x % a.n_colblocks, band = blockIdx.x / a.n_colblocks, n = blockIdx.y;
    const int ringcols = a.ringcols;
    const int rowbytes = ringcols * PIXB;
    char* const wl = smem;
    float* const tabs = reinterpret_cast<float*>(smem + KC * 1024);          // [3][16]: mean, inv, beta
    char* const ring = smem + KC * 1024 + 256;
    [[maybe_unused]] char* const pbuf = ring + NSL * rowbytes;                 // KS == 3: [2][npt] partial tiles of 1 KB

    const int yo0 = band * a.rows_per_band;
    const int yo1 = min(a.Ho, yo0 + a.rows_per_band);
    const int yc0 = yo0 * PS;
    const int nconv = (yo1 - yo0 - 1) * PS + PK;
    const int nin = nconv + 2;
    const int x0c = cb * npt * TSTRIDE;
    const int xo_blk0 = x0c / PS;

    // weights: in LDS in fragment order (read per use), or -- where the 9 KG fragments fit the register file (KS = 1, 32 channels:
    // 72 registers) -- in registers: the B operand then is the only LDS read of the chain (per row and workgroup the weight reads
    // were half of its LDS traffic, which ran at half of the matrix time)
    constexpr bool WREG = KS == 1 && KC <= 18 && CFZ > 0;
    [[maybe_unused]] f32x4 wreg[WREG ? KC : 1];
    if constexpr (WREG) {
#pragma unroll
        for (int i = 0; i < KC; ++i) wreg[i] = a.wfrag[i * 64 + lane];
    } else {
        for (int i = tid; i < KC * 64; i += nthreads) reinterpret_cast<f32x4*>(wl)[i] = a.wfrag[i];
    }
    for (int i = tid; i < 48; i += nthreads) tabs[i] = (i < 16 ? a.bn_mean : i < 32 ? a.bn_inv : a.bn_beta)[i & 15];

    const int nchunks = ringcols * CP;
    const float* const in_img = a.in + static_cast<int64_t>(n) * a.H * a.W * CIN;
    int ld_goff[LPT], ld_loff[LPT];
#pragma unroll
    for (int i = 0; i < LPT; ++i) {
        const int q = tid + i * nthreads;
        const int p = q / CP, c4 = q % CP;
        ld_loff[i] = q < nchunks ? (p * CP + (c4 ^ m16_swz<CP>(p))) * 16 : -1;
        ld_goff[i] = (q < nchunks ? min(x0c + p, a.W - 1) : 0) * CIN + c4 * 4;
    }
    f32x4 pre[LPT];
    auto fetch_row = [&](int j) {
        const float* row = in_img + static_cast<int64_t>(yc0 + j) * a.W * CIN;
#pragma unroll
        for (int i = 0; i < LPT; ++i) pre[i] = *reinterpret_cast<const f32x4*>(row + ld_goff[i]);
    };
    auto store_row = [&](int j) {
        char* dst = ring + (j % NSL) * rowbytes;
#pragma unroll
        for (int i = 0; i < LPT; ++i)
            if (ld_loff[i] >= 0) *reinterpret_cast<f32x4*>(dst + ld_loff[i]) = pre[i];
    };
    for (int j = 0; j < 3; ++j) {
        fetch_row(j);
        store_row(j);
    }
    __syncthreads();

    const int xrel0 = wave * TSTRIDE + px;
    int boff[3], bswz[3];
#pragma unroll
    for (int kx = 0; kx < 3; ++kx) {
        boff[kx] = (xrel0 + kx) * PIXB;
        bswz[kx] = m16_swz<CP>(xrel0 + kx);
    }
    const int xc = x0c + xrel0;
    const int xo = xc / PS;
    const bool lane_out = (px % PS == 0 && px <= 16 - PK) && xo < a.Wo && (xo - xo_blk0) < npt * NOUT_T;
    float vring[3][4];
#pragma unroll
    for (int i = 0; i < 3; ++i)
#pragma unroll
        for (int g = 0; g < 4; ++g) vring[i][g] = 0.f;
    const char* const wl_lane = wl + lane * 16;
    const f32x4 t_mean = *reinterpret_cast<const f32x4*>(tabs + 4 * kq), t_inv = *reinterpret_cast<const f32x4*>(tabs + 16 + 4 * kq),
                t_beta = *reinterpret_cast<const f32x4*>(tabs + 32 + 4 * kq);
    [[maybe_unused]] f32x4 t_frozen = {0.f, 0.f, 0.f, 0.f};
    if constexpr (CFZ > 0) t_frozen = *reinterpret_cast<const f32x4*>(a.bn_beta + COUT + 4 * kq);

    for (int it = 0; it < nconv; ++it) {
        const bool have_next = it + 3 < nin;
        if (have_next) fetch_row(it + 3);
        const bool emit = it >= PK - 1 && ((it - (PK - 1)) % PS) == 0;
        const int yo = yo0 + (it - (PK - 1)) / PS;
        f32x4 acc = {0.f, 0.f, 0.f, 0.f};
#pragma unroll
        for (int t3 = 0; t3 < (KS == 3 ? 3 : 9); ++t3) {
            const int ky = KS == 3 ? kh : t3 / 3, kx = t3 % 3;        // (KS == 3: ky is wave-uniform)
            const char* rowp = ring + ((it + ky) % NSL) * rowbytes;
            const char* wrow = wl_lane + (ky * 3 + kx) * KG * 1024;
#pragma unroll
            for (int q = 0; q < KG; ++q) {
                const f32x4 b = *reinterpret_cast<const f32x4*>(rowp + boff[kx] + (((4 * q + kq) ^ bswz[kx]) << 4));
                f32x4 wv;
                if constexpr (WREG)
                    wv = wreg[(WREG ? t3 : 0) * KG + q];
                else
                    wv = *reinterpret_cast<const f32x4*>(wrow + q * 1024);
#pragma unroll
                for (int i = 0; i < 4; ++i) acc = mfma_f32_16(wv[i], b[i], acc);
            }
        }
        if constexpr (KS == 3) {
            if (kh > 0) *reinterpret_cast<f32x4*>(pbuf + ((kh - 1) * npt + wave) * 1024 + lane * 16) = acc;
            lds_barrier();
            if (kh == 0) {
                const f32x4 p1 = *reinterpret_cast<const f32x4*>(pbuf + wave * 1024 + lane * 16);
                const f32x4 p2 = *reinterpret_cast<const f32x4*>(pbuf + (npt + wave) * 1024 + lane * 16);
#pragma unroll
                for (int j = 0; j < 4; ++j) acc[j] = (acc[j] + p1[j]) + p2[j];
            }
        }
        if (kh == 0) {
            float tot[4];
#pragma unroll
            for (int g = 0; g < 4; ++g) {
                const float v = relu6f(acc[g]);
                const float t = v + row_next<1>(v);
                const float hs = t + row_next<2>(t);
                float s = vring[0][g];
                s += vring[1][g];
                s += vring[2][g];
                tot[g] = s + hs;
                vring[0][g] = vring[1][g];
                vring[1][g] = vring[2][g];
                vring[2][g] = hs;
            }
            if (emit && lane_out) {
                f32x4 y;
#pragma unroll
                for (int j = 0; j < 4; ++j)
                    y[j] = __fadd_rn(__fmul_rn(__fsub_rn(__fmul_rn(tot[j], 1.0f / 16.0f), t_mean[j]), t_inv[j]), t_beta[j]);
                float* opix = a.out + ((static_cast<int64_t>(n) * a.Ho + yo) * a.Wo + xo) * (COUT + CFZ) + 4 * kq;
                *reinterpret_cast<f32x4*>(opix) = y;
                if constexpr (CFZ > 0) *reinterpret_cast<f32x4*>(opix + COUT) = t_frozen;
            }
        }
        if constexpr (NSL == 3) lds_barrier();            // everybody is past row `it` (and the partial tiles) before its slot is refilled
        if (have_next) store_row(it + 3);                 // (NSL == 4: slot (it + 3) % 4 held row it - 1, which the last barrier retired)
        lds_barrier();
    }
}

// The residual epilogue ALONE for the couts of a residual stage whose first BN is frozen (rn_create: y1 = beta for every input):
// out = BN2(beta + legacy-bilinear(skip)), the expressions of stage_f32m_kernel's epilogue un-contracted; one thread = 4 couts of
// one output pixel.
__global__ __launch_bounds__(256) void f32_frozen_residual_kernel(const float* __restrict__ skip, float* __restrict__ out, const float* bn_beta,
                                                                  const float* bn2_mean, const float* bn2_inv, const float* bn2_beta, const int32_t* rlo,
                                                                  const int32_t* rhi, const float* rlerp, int64_t n_pix, int Ho, int Wo, int Ss, int cout,
                                                                  int c_begin, int c_count) {
    const int groups = c_count / 4;
    const int64_t idx = static_cast<int64_t>(blockIdx.x) * 256 + threadIdx.x;
    if (idx >= n_pix * groups) return;
    const int64_t p = idx / groups;
    const int c0 = c_begin + 4 * static_cast<int>(idx % groups);
    const int xo = static_cast<int>(p % Wo), yo = static_cast<int>((p / Wo) % Ho);
    const int64_t n = p / (static_cast<int64_t>(Wo) * Ho);
    const float xl = rlerp[xo], yl = rlerp[yo];
    const float* skn = skip + n * Ss * Ss * cout;
    const float* sk0 = skn + static_cast<int64_t>(rlo[yo]) * Ss * cout;
    const float* sk1 = skn + static_cast<int64_t>(rhi[yo]) * Ss * cout;
    const f32x4 tl = *reinterpret_cast<const f32x4*>(sk0 + rlo[xo] * cout + c0), tr = *reinterpret_cast<const f32x4*>(sk0 + rhi[xo] * cout + c0);
    const f32x4 bl = *reinterpret_cast<const f32x4*>(sk1 + rlo[xo] * cout + c0), br = *reinterpret_cast<const f32x4*>(sk1 + rhi[xo] * cout + c0);
    const f32x4 beta = *reinterpret_cast<const f32x4*>(bn_beta + c0), m2 = *reinterpret_cast<const f32x4*>(bn2_mean + c0),
                i2 = *reinterpret_cast<const f32x4*>(bn2_inv + c0), b2 = *reinterpret_cast<const f32x4*>(bn2_beta + c0);
    f32x4 y;
#pragma unroll
    for (int j = 0; j < 4; ++j) {
        const float top = __fadd_rn(tl[j], __fmul_rn(__fsub_rn(tr[j], tl[j]), xl));
        const float bot = __fadd_rn(bl[j], __fmul_rn(__fsub_rn(br[j], bl[j]), xl));
        const float rs = __fadd_rn(top, __fmul_rn(__fsub_rn(bot, top), yl));
        y[j] = __fadd_rn(__fmul_rn(__fsub_rn(__fadd_rn(beta[j], rs), m2[j]), i2[j]), b2[j]);
    }
    *reinterpret_cast<f32x4*>(out + p * cout + c0) = y;
}

struct F32mStage {
    bool on = false;
    bool m16 = false;             // stage_f32m16_kernel (16 couts): ks / lpt16 below
    int ks16 = 1;
    f32x4* wfrag = nullptr;
    float* cinit = nullptr;       // per cout: the frozen input channels' contribution (variants with live_cin < cin)
    int variant = -1, npt = 1, n_colblocks = 1, n_ctg = 1, nsl = 4, ringcols = 34;
    size_t lds = 0;
};

struct F32mState {
    std::vector<F32mStage> st;
};

using F32LaunchFn = void (*)(const F32StageArgs&, dim3, dim3, size_t, hipStream_t);

template <int CIN, int COUT, int PK, int PS, bool RES, int NSL, int LPT, int KS, int KQL = CIN / 8>
void launch_f32m(const F32StageArgs& a, dim3 grid, dim3 block, size_t lds, hipStream_t s) {
    auto kern = stage_f32m_kernel<CIN, COUT, PK, PS, RES, NSL, LPT, KS, KQL>;
    static std::atomic<unsigned long long> attr_devices{0};
    int dev = 0;
    (void)hipGetDevice(&dev);
    if (!(attr_devices.load(std::memory_order_acquire) >> (dev & 63) & 1ull)) {
        (void)hipFuncSetAttribute(reinterpret_cast<const void*>(kern), hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024);
        attr_devices.fetch_or(1ull << (dev & 63), std::memory_order_release);
    }
    hipLaunchKernelGGL(kern, grid, block, lds, s, a);
}

struct F32Variant {
    int cin, cout, pk, ps, res, nsl, lpt, ks;
    int live_cin;                 // input channels contracted (== cin unless the variant folds frozen input channels)
    F32LaunchFn fn;
};
const F32Variant kF32Variants[] = {
    {8, 32, 4, 1, 0, 4, 4, 1, 8, launch_f32m<8, 32, 4, 1, false, 4, 4, 1>},         // stage 1
    {32, 32, 4, 1, 0, 4, 4, 1, 32, launch_f32m<32, 32, 4, 1, false, 4, 4, 1>},      // stage 2
    {32, 32, 4, 1, 1, 4, 4, 1, 32, launch_f32m<32, 32, 4, 1, true, 4, 4, 1>},       // stage 3
    {32, 32, 4, 1, 1, 4, 4, 1, 16, launch_f32m<32, 32, 4, 1, true, 4, 4, 1, 2>},    // stage 3, 16 of its input channels frozen
    {32, 64, 4, 2, 0, 4, 4, 1, 32, launch_f32m<32, 64, 4, 2, false, 4, 4, 1>},      // stage 4
    {64, 64, 4, 2, 1, 3, 8, 1, 64, launch_f32m<64, 64, 4, 2, true, 3, 8, 1>},       // stage 5
    {64, 128, 0, 1, 0, 3, 8, 2, 64, launch_f32m<64, 128, 0, 1, false, 3, 8, 2>},    // stage 6
};

template <int CIN, int KS, int LPT, int PS = 2, int CFZ = 0>
void launch_f32m16(const F32StageArgs& a, dim3 grid, dim3 block, size_t lds, hipStream_t s) {
    auto kern = stage_f32m16_kernel<CIN, KS, LPT, PS, CFZ>;
    static std::atomic<unsigned long long> attr_devices{0};
    int dev = 0;
    (void)hipGetDevice(&dev);
    if (!(attr_devices.load(std::memory_order_acquire) >> (dev & 63) & 1ull)) {
        (void)hipFuncSetAttribute(reinterpret_cast<const void*>(kern), hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024);
        attr_devices.fetch_or(1ull << (dev & 63), std::memory_order_release);
    }
    hipLaunchKernelGGL(kern, grid, block, lds, s, a);
}
struct F32Variant16 {
    int cin, ks, lpt;
    int ps, frozen;               // pool stride; frozen couts written as constants beside the 16 live ones
    int nsl, max_tiles;           // ring slots; pixel tiles per workgroup at most
    F32LaunchFn fn;
};
const F32Variant16 kF32Variants16[] = {
    {128, 3, 4, 2, 0, 3, 3, launch_f32m16<128, 3, 4>},          // stage 7 (npt = 3: 576 threads, a 46-pixel ring row = 1 472 chunks)
    {16, 1, 2, 2, 0, 3, 9, launch_f32m16<16, 1, 2>},            // stage 8
    // stage 2 with 16 frozen couts (weights in registers, four ring slots).  Tiles per workgroup, measured twice (ms): 9 1.53-1.57 |
    // 8 (blocks of 8, 8, 1) 1.75 | 6 1.66-1.71 | 5 2.11 | 4 (4, 4, 4, 4, 1: three workgroups per CU) 1.35-1.39, the next stage 0.06
    // slower, the pass +1.4 % | 3 1.53 | 2 1.85
    {32, 1, 2, 1, 16, 4, 4, launch_f32m16<32, 1, 2, 1, 16>},
};

}  // namespace

void rn_f32m_release(rn_handle* h) {
    delete static_cast<F32mState*>(h->f32m);
    h->f32m = nullptr;
}

// Pack the weights of every stage the kernel covers: frag[tap * CIN / 8 + q][cout tile][lane][i] =
// W[tap][channel 8 q + 4 (lane / 32) + i][cout 32 tile + lane % 32]
int rn_f32m_prepare(rn_handle* h, const rn_weights* w) {
    auto* fs = new F32mState();
    fs->st.resize(h->stages.size());
    h->f32m = fs;
    auto upload = [&](const std::vector<float>& v, const char* what, float** out) -> int {
        void* d = nullptr;
        if (hipMalloc(&d, v.size() * 4) != hipSuccess) {
            rn_set_error("hipMalloc(%s) failed", what);
            return RN_E_NOMEM;
        }
        h->allocs.push_back(d);
        RN_HIP(hipMemcpy(d, v.data(), v.size() * 4, hipMemcpyHostToDevice));
        *out = static_cast<float*>(d);
        return RN_OK;
    };
    auto prepare_stage = [&](size_t si) -> int {
        const StagePlan& s = h->stages[si];
        F32mStage& f = fs->st[si];
        // the producer of a stage that does not contract its frozen input channels need not convolve them either: 16 live couts on
        // the 16 x 16 x 4 tiles, the frozen ones written as constants (only once the consumer's variant is in place)
        const bool live16 = static_cast<int>(si) + 1 == h->f32_kfold_stage && fs->st[si + 1].on && fs->st[si + 1].cinit && h->f32_kfold_live == 16 &&
                            s.cout == 32 && s.pool_k == 4 && s.pool_s == 1 && s.skip_stage < 0;
        // (rn_create relabelled a stage's frozen input channels to the end: the variant that contracts the others only, if there is one)
        int want_live = static_cast<int>(si) == h->f32_kfold_stage ? h->f32_kfold_live : s.cin;
        for (int pass = 0; pass < 2 && f.variant < 0 && !live16; ++pass) {
            for (size_t v = 0; v < sizeof(kF32Variants) / sizeof(kF32Variants[0]); ++v) {
                const F32Variant& k = kF32Variants[v];
                if (k.cin == s.cin && k.cout == s.cout && k.pk == s.pool_k && (s.pool_k == 0 || k.ps == s.pool_s) &&
                    k.res == (s.skip_stage >= 0 ? 1 : 0) && k.live_cin == want_live)
                    f.variant = static_cast<int>(v);
            }
            if (f.variant < 0 && want_live != s.cin) {
                h->f32_kfold_stage = -1;          // no such variant: every channel is contracted (the relabelling alone changes nothing)
                want_live = s.cin;
            }
        }
        if (f.variant < 0 && (live16 || (s.cout == 16 && s.pool_k == 4 && s.pool_s == 2 && s.skip_stage < 0))) {
            // 16-cout stages: stage_f32m16_kernel; frag[tap * CIN / 16 + q][lane][i] = W[tap][channel 16 q + 4 (lane / 16) + i][cout lane % 16]
            const int frozen = live16 ? 16 : 0;
            for (size_t v = 0; v < sizeof(kF32Variants16) / sizeof(kF32Variants16[0]); ++v) {
                const F32Variant16& k = kF32Variants16[v];
                if (k.cin != s.cin || k.ps != s.pool_s || k.frozen != frozen) continue;
                const int tstride = k.ps == 2 ? 14 : 13, nout_t = k.ps == 2 ? 7 : 13;
                const int kg = s.cin / 16, tiles = (s.out_side + nout_t - 1) / nout_t;
                int npt = std::min(tiles, k.max_tiles);
                npt = std::min(npt, (tiles + (tiles + npt - 1) / npt - 1) / ((tiles + npt - 1) / npt));      // even column blocks
                size_t lds = 0;
                int ringcols = 0;
                for (; npt >= 1; --npt) {
                    ringcols = std::min((npt - 1) * tstride + 18, s.in_side);
                    lds = static_cast<size_t>(9 * kg) * 1024 + 256 + static_cast<size_t>(k.nsl) * ringcols * s.cin * 4 + (k.ks == 3 ? 2 * npt * 1024 : 0);
                    if (lds <= 160 * 1024 && ringcols * (s.cin / 4) <= k.lpt * 64 * npt * k.ks) break;
                }
                if (npt < 1) continue;
                std::vector<float> frag(static_cast<size_t>(9 * kg) * 64 * 4, 0.f);
                const float* wsrc = w->stages[si].kernel;      // HWIO = [tap][cin][cout]: the first 16 couts
                for (int tap = 0; tap < 9; ++tap)
                    for (int q = 0; q < kg; ++q)
                        for (int l = 0; l < 64; ++l)
                            for (int i = 0; i < 4; ++i)
                                frag[((static_cast<size_t>(tap) * kg + q) * 64 + l) * 4 + i] =
                                    wsrc[(static_cast<size_t>(tap) * s.cin + 16 * q + 4 * (l >> 4) + i) * s.cout + (l & 15)];
                float* d = nullptr;
                const int rc = upload(frag, "fp32 MFMA weights", &d);
                if (rc != RN_OK) return rc;
                f.wfrag = reinterpret_cast<f32x4*>(d);
                f.m16 = true;
                f.variant = static_cast<int>(v);
                f.ks16 = k.ks;
                f.npt = npt;
                f.ringcols = ringcols;
                f.lds = lds;
                f.n_colblocks = (tiles + npt - 1) / npt;
                f.n_ctg = 1;
                f.on = true;
                break;
            }
            if (f.on || !live16) return RN_OK;
            // (no 16-cout variant for this producer: it convolves every channel like any other stage)
            for (size_t v = 0; v < sizeof(kF32Variants) / sizeof(kF32Variants[0]); ++v) {
                const F32Variant& k = kF32Variants[v];
                if (k.cin == s.cin && k.cout == s.cout && k.pk == s.pool_k && k.ps == s.pool_s && k.res == 0 && k.live_cin == s.cin) f.variant = static_cast<int>(v);
            }
        }
        if (f.variant < 0) return RN_OK;
        if (s.skip_stage >= 0 && h->stages[s.skip_stage].node_bn2 >= 0) return RN_OK;      // skip source = a first BN output
        const int kq = s.cin / 8, kc = 9 * kq, ct_n = (s.cout + 31) / 32;
        f.nsl = kF32Variants[f.variant].nsl;
        // pixel tiles (= waves) per workgroup: as many as fit the LDS next to the weights, at most 8
        const int tstride = tile_stride(s.pool_k, s.pool_s), nout_t = tile_nout(s.pool_k, s.pool_s);
        const int tiles = (s.out_side + nout_t - 1) / nout_t;
        f.npt = std::min(tiles, 8);
        for (;;) {
            f.ringcols = std::min((f.npt - 1) * tstride + 34, s.in_side);
            const int ks = kF32Variants[f.variant].ks;
            f.lds = static_cast<size_t>(kc) * 1024 + 1024 + static_cast<size_t>(f.nsl) * f.ringcols * s.cin * 4 + (ks == 2 ? f.npt * 4096 : 0);
            const int chunks = f.ringcols * (s.cin / 4);
            if ((f.lds <= 160 * 1024 && chunks <= kF32Variants[f.variant].lpt * 64 * f.npt * ks && f.npt * ks <= 8) || f.npt == 1) break;
            --f.npt;
        }
        if (f.lds > 160 * 1024 || f.ringcols * (s.cin / 4) > kF32Variants[f.variant].lpt * 64 * f.npt * kF32Variants[f.variant].ks)
            return RN_OK;      // not coverable
        f.n_colblocks = (tiles + f.npt - 1) / f.npt;
        f.n_ctg = ct_n;
        std::vector<float> frag(static_cast<size_t>(kc) * ct_n * 64 * 4, 0.f);
        const float* wsrc = w->stages[si].kernel;      // HWIO = [tap][cin][cout]
        for (int tap = 0; tap < 9; ++tap)
            for (int q = 0; q < kq; ++q)
                for (int t = 0; t < ct_n; ++t)
                    for (int l = 0; l < 64; ++l)
                        for (int i = 0; i < 4; ++i) {
                            const int c = 8 * q + 4 * (l >> 5) + i, co = 32 * t + (l & 31);
                            if (co < s.cout)
                                frag[((static_cast<size_t>(tap * kq + q) * ct_n + t) * 64 + l) * 4 + i] =
                                    wsrc[(static_cast<size_t>(tap) * s.cin + c) * s.cout + co];
                        }
        float* d = nullptr;
        int rc = upload(frag, "fp32 MFMA weights", &d);
        if (rc != RN_OK) return rc;
        f.wfrag = reinterpret_cast<f32x4*>(d);
        if (want_live < s.cin) {
            // the producer writes beta[c] for its frozen channel c at every pixel: sum over the nine taps, in double
            const float* beta = w->stages[si - 1].beta;
            std::vector<float> ci(static_cast<size_t>(ct_n) * 32, 0.f);
            for (int co = 0; co < s.cout; ++co) {
                double acc = 0.0;
                for (int tap = 0; tap < 9; ++tap)
                    for (int c = want_live; c < s.cin; ++c)
                        acc += static_cast<double>(wsrc[(static_cast<size_t>(tap) * s.cin + c) * s.cout + co]) * static_cast<double>(beta[c]);
                ci[co] = static_cast<float>(acc);
            }
            if ((rc = upload(ci, "fp32 frozen-input constants", &f.cinit)) != RN_OK) return rc;
        }
        f.on = true;
        return RN_OK;
    };
    // the stage with frozen input channels first: whether its producer may leave them out depends on it
    const int kf = h->f32_kfold_stage;
    if (kf >= 0) {
        const int rc = prepare_stage(static_cast<size_t>(kf));
        if (rc != RN_OK) return rc;
        if (!fs->st[kf].cinit) h->f32_kfold_stage = -1;      // (the stage is not covered, or has no variant for that channel count)
    }
    for (size_t si = 0; si < h->stages.size(); ++si) {
        if (static_cast<int>(si) == kf) continue;
        const int rc = prepare_stage(si);
        if (rc != RN_OK) return rc;
    }
    return RN_OK;
}

bool rn_f32m_covers(const rn_handle* h, int stage) {
    const F32mState* fs = static_cast<const F32mState*>(h->f32m);
    return fs && stage >= 0 && stage < static_cast<int>(fs->st.size()) && fs->st[stage].on;
}

int rn_f32m_launch(rn_handle* h, int stage, const float* in, int n) {
    const F32mState* fs = static_cast<const F32mState*>(h->f32m);
    const F32mStage& f = fs->st[stage];
    const StagePlan& s = h->stages[stage];
    F32StageArgs a{};
    a.in = in;
    a.out = static_cast<float*>(h->nodes[s.node_bn2 >= 0 ? s.node_bn2 : s.node_bn].ptr);
    a.wfrag = f.wfrag;
    a.cinit = f.cinit;
    a.bn_mean = s.bn.mean;
    a.bn_inv = s.bn.inv;
    a.bn_beta = s.bn.beta;
    if (s.skip_stage >= 0) {
        a.skip = static_cast<const float*>(h->nodes[h->stages[s.skip_stage].node_bn].ptr);
        a.bn2_mean = s.bn2.mean;
        a.bn2_inv = s.bn2.inv;
        a.bn2_beta = s.bn2.beta;
        a.rlo = s.rt.lo;
        a.rhi = s.rt.hi;
        a.rlerp = s.rt.lerp;
        a.Ss = s.skip_side;
    }
    a.H = a.W = s.in_side;
    a.Ho = a.Wo = s.out_side;
    a.npt = f.npt;
    a.ringcols = f.ringcols;
    a.n_colblocks = f.n_colblocks;
    a.n_ctg = f.n_ctg;
    // frozen cout tiles of this stage (rn_create proved y1 = beta for their channels and relabelled them to the end): not convolved
    const bool fold = !f.m16 && stage == h->f32_fold_stage && s.skip_stage >= 0 && h->f32_fold_live > 0 && h->f32_fold_live < s.cout &&
                      h->f32_fold_live % 32 == 0;
    if (fold) a.n_ctg = h->f32_fold_live / 32;
    // bands: whole rounds of the chip (one workgroup per CU: the weights and the ring fill most of its LDS); a band costs its
    // rows plus the rows its neighbour reads again
    const long per_band = static_cast<long>(n) * f.n_colblocks * a.n_ctg;
    const int rows_in = s.pool_k ? s.pool_s : 1, overlap = s.pool_k ? 5 : 2;
    const int max_bands = std::max(1, s.out_side / 4);
    int bands = 1;
    long best = -1;
    for (int b = 1; b <= 16 && b <= max_bands; ++b) {
        const long rounds = (per_band * b + h->n_cu - 1) / h->n_cu;
        const long cost = rounds * (rows_in * ((s.out_side + b - 1) / b) + overlap);
        if (best < 0 || cost < best) {
            best = cost;
            bands = b;
        }
    }
    if (per_band * bands < h->n_cu) bands = static_cast<int>(std::min<long>((h->n_cu + per_band - 1) / per_band, max_bands));
    a.rows_per_band = (s.out_side + bands - 1) / bands;
    a.n_bands = (s.out_side + a.rows_per_band - 1) / a.rows_per_band;
    if (f.m16)
        kF32Variants16[f.variant].fn(a, dim3(a.n_bands * a.n_colblocks, n), dim3(64 * f.npt * f.ks16), f.lds, h->stream);
    else
        kF32Variants[f.variant].fn(a, dim3(a.n_bands * a.n_colblocks * a.n_ctg, n), dim3(64 * f.npt * kF32Variants[f.variant].ks), f.lds, h->stream);
    RN_CHECK_LAUNCH();
    if (fold) {
        const int c_begin = h->f32_fold_live, c_count = s.cout - c_begin;
        const int64_t n_pix = static_cast<int64_t>(n) * s.out_side * s.out_side;
        const int64_t threads = n_pix * (c_count / 4);
        hipLaunchKernelGGL(f32_frozen_residual_kernel, dim3(static_cast<unsigned>((threads + 255) / 256)), dim3(256), 0, h->stream, a.skip, a.out, a.bn_beta,
                           a.bn2_mean, a.bn2_inv, a.bn2_beta, a.rlo, a.rhi, a.rlerp, n_pix, s.out_side, s.out_side, s.skip_side, s.cout, c_begin, c_count);
        RN_CHECK_LAUNCH();
    }
    return RN_OK;
}
