// Fused 16-bit execution path (RN_DTYPE_BF16 / RN_DTYPE_F16): one launch per conv stage.
//
// A stage is   conv3x3 VALID s1 (no bias) -> ReLU6 -> [avg-pool k x k / s] -> BN
//              -> [ + legacy-bilinear(skip) -> BN ]          (reference network.py:172-208)
// Activations live in HBM as NHWC 16-bit tensors; a stage reads its input once and
// writes its post-BN output once (the stage-boundary traffic model of SURVEY.md 8d).
// Accumulation, ReLU6, pooling, BN and the residual are float32 in registers.
//
// Kernel structure (stages 1..N, `stage_mfma_kernel`):
//   * a workgroup owns one image, one band of output rows and one block of columns and
//     walks down its band one conv row per iteration ("row streaming"): the last
//     3 input rows live in an LDS ring, the next row is prefetched into registers while
//     the current one is computed -- every input row is fetched once per band.
//   * implicit GEMM on the matrix cores, D[cout][pixel] = W^T[cout][k] * im2col[k][pixel]
//     with v_mfma_f32_32x32x16_{bf16,f16}: a wave owns a tile of 32 consecutive conv
//     columns; the B operand (8 consecutive channels of one tap of one pixel = 16 B) is a
//     single ds_read_b128 from the NHWC ring, made bank-conflict free by XOR-swizzling the
//     16-byte channel chunk inside each pixel; the A operand (weights) is pre-packed on
//     the host in fragment order and read from LDS with lane-linear ds_read_b128.
//   * the accumulator layout puts the pixel on the lane and the channel in the
//     register, so ReLU6 is per register, the horizontal pool sum is two DPP wave shifts
//     per register, the vertical pool sum is a register ring across iterations, BN is
//     an fma, and the pooled tile never touches LDS or HBM before its final store.
//   * neighbouring pixel tiles overlap by k-1 columns so that no cross-wave exchange is
//     needed for the horizontal pool.
// Stage 0 (3 input channels, K = 27) runs on the matrix cores too (`stage0_kernel` below): its operand is the
// uint8 pixel value, the uint8 -> [-1,1] pre-processing of network.py:129 is folded into its weights.
#include "rn_fused.h"
#include <map>
#include <vector>
#include <algorithm>
#include <atomic>
#include "rn_stage.h"

#include <cmath>
#include <cstring>
#include <cstdlib>

using namespace rnk;

namespace {

// ------------------------------------------------------------------------------ stage 0
// uint8 BGR [N,S,S,3] -> table -> conv3x3 (3 -> 8) -> ReLU6 -> avg-pool 3x3/1 -> BN -> 16-bit NHWC.
//
// Row streaming on the matrix cores with the im2col built entirely in registers:
//   * K is laid out as (ky, kx in 0..3, c in 0..3) = 48 (kx = 3 and c = 3 are zero weights), i.e.
//     ONE 16-deep MFMA K-chunk per input row ky.  For v_mfma_f32_32x32x16 the B operand of lane
//     (r, h) is then: h = 0: pixels r and r+1 (4 x 16-bit each), h = 1: pixel r+2 and zeros.
//   * lane (r, h) loads ITS pixel x0 + r + 2h (3 bytes) and packs the byte values as fp16 numbers
//     (R,G | B,0) -- exact; the pre-processing table of network.py:129 is folded into the weights
//     (s0_pixel_halves, rn_stage.h); the lower half-wave gets pixel r+1 from its neighbour lane
//     with a DPP shift.  The fragments of the last 3 input rows stay in 12 VGPRs.
//   * D[cout][pixel]: rows 0..7 hold the hi halves of the folded weights, rows 8..15 the lo halves,
//     so a lane owns 4 channels of one conv pixel in 4 + 4 accumulator registers: add, ReLU6,
//     3-wide horizontal sum by DPP, 3-row vertical sum in a register ring, one fma for BN, one
//     8-byte store.  No LDS traffic.
// A wave owns 32 conv columns (29 output columns, tiles overlap by 3) and walks down a band of
// rows; a workgroup is up to 8 such waves side by side.
// The MFMA inputs of this stage are ALWAYS fp16, whatever the storage type of the activations:
// fp16 holds the 256 input levels exactly and the weights as hi + lo pairs, so the stage computes
// the fp32 convolution of the exact input (bf16's 8-bit significand cannot represent the levels).
constexpr int S0_CO = 8;
constexpr int S0_TSTRIDE = 29;      // output columns per 32-column tile: 32 - (3 - 1) - 1
constexpr int S0_AHEAD = 8;         // input rows prefetched (one dword per lane and row in registers)

struct Stage0Args {
    const uint8_t* bgr;             // [N, S, S, 3]
    const i32x4* wfrag;             // [3 (ky)][64 lanes] A fragments, 8 x fp16: cout rows 0..7 hi, 8..15 lo
    const float* ptab;              // [2][8] folded BN: scale (inv / 9 / 2^8), shift
    unsigned short* out;            // [N, So, So, 8]
    int S, So;
    int rows_per_band, n_bands, n_colblocks, npt;
};

template <int DT>
__global__ __launch_bounds__(512) void stage0_kernel(const Stage0Args a) {
    const int tid = threadIdx.x;
    const int lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int r = lane & 31, hh = lane >> 5;
    const int cb = blockIdx.x % a.n_colblocks;
    const int band = blockIdx.x / a.n_colblocks;
    const int n = blockIdx.y;

    const int yo0 = band * a.rows_per_band;
    const int yo1 = min(a.So, yo0 + a.rows_per_band);
    const int nconv = (yo1 - yo0) + 2;                 // conv rows of the band (pool 3, stride 1)
    const int nin = nconv + 2;                         // input rows
    const int xt0 = (cb * a.npt + wave) * S0_TSTRIDE;  // first conv / input column of this wave's tile
    const int px = min(xt0 + r + 2 * hh, a.S - 1);     // this lane's input column (clamped at the edge)
    // One (unaligned) dword load per lane and row covers the pixel's 3 bytes -- three byte loads cost the
    // address coalescer three passes per row.  The last column reads one byte early and shifts, so no lane
    // ever touches the byte behind the caller's buffer.
    const int sh0 = px == a.S - 1 ? 8 : 0;
    const uint8_t* src = a.bgr + (static_cast<int64_t>(n) * a.S * a.S + static_cast<int64_t>(yo0) * a.S + px) * 3 - (sh0 >> 3);
    const int row_bytes = a.S * 3;

    i32x4 wreg[3];
#pragma unroll
    for (int ky = 0; ky < 3; ++ky) wreg[ky] = a.wfrag[ky * 64 + lane];
    const f32x4 scale = *reinterpret_cast<const f32x4*>(a.ptab + 4 * hh);
    const f32x4 shift = *reinterpret_cast<const f32x4*>(a.ptab + 8 + 4 * hh);
    const int xo = xt0 + r;
    const bool lane_out = r < S0_TSTRIDE && xo < a.So && (xo - cb * a.npt * S0_TSTRIDE) < a.npt * S0_TSTRIDE;
    unsigned short* out_lane = a.out + (static_cast<int64_t>(n) * a.So * a.So + xo) * S0_CO + 4 * hh;
    const unsigned nb_mask = hh ? 0u : 0xffffffffu;    // the upper half-wave's second pixel slot is zero

    // prefetch queue of raw bytes
    unsigned pw[S0_AHEAD];
    auto load_px = [&](int j) -> unsigned {
        unsigned w;
        __builtin_memcpy(&w, src + static_cast<int64_t>(min(j, nin - 1)) * row_bytes, 4);   // unaligned dword
        return w;
    };
#pragma unroll
    for (int i = 0; i < S0_AHEAD; ++i) pw[i] = load_px(i);

    i32x4 bfr[3];                                      // B fragments of the 3 live input rows
    float h1[4], h2[4];                                // horizontal sums of the two previous conv rows
#pragma unroll
    for (int j = 0; j < 4; ++j) h1[j] = h2[j] = 0.f;
#pragma unroll
    for (int i = 0; i < 3; ++i) bfr[i] = i32x4{0, 0, 0, 0};

    // consume the oldest prefetched row into a B fragment, refill the queue slot
    auto next_frag = [&](int jrow) -> i32x4 {
        const unsigned w = pw[0] >> sh0;                       // bytes: B, G, R
        int d0, d1;
        s0_pixel_halves(w, d0, d1);
#pragma unroll
        for (int i = 0; i + 1 < S0_AHEAD; ++i) pw[i] = pw[i + 1];
        pw[S0_AHEAD - 1] = load_px(jrow + S0_AHEAD);
        i32x4 f;
        f[0] = d0;
        f[1] = d1;
        f[2] = static_cast<int>(static_cast<unsigned>(__builtin_amdgcn_update_dpp(0, d0, 0x130, 0xf, 0xf, true)) & nb_mask);
        f[3] = static_cast<int>(static_cast<unsigned>(__builtin_amdgcn_update_dpp(0, d1, 0x130, 0xf, 0xf, true)) & nb_mask);
        return f;
    };
    bfr[0] = next_frag(0);
    bfr[1] = next_frag(1);

    const f32x16 zero = {0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f};
    for (int it = 0; it < nconv; ++it) {
        bfr[2] = next_frag(it + 2);
        f32x16 acc = mfma32<RN_DTYPE_F16>(wreg[0], bfr[0], zero);
        acc = mfma32<RN_DTYPE_F16>(wreg[1], bfr[1], acc);
        acc = mfma32<RN_DTYPE_F16>(wreg[2], bfr[2], acc);
        bfr[0] = bfr[1];
        bfr[1] = bfr[2];
        float y[4];
#pragma unroll
        for (int j = 0; j < 4; ++j) {
            const float v = s0_relu6(acc, j);
            const float v1 = lane_next(v);
            const float hs = (v + v1) + lane_next(v1);
            y[j] = fmaf((h2[j] + h1[j]) + hs, scale[j], shift[j]);
            h2[j] = h1[j];
            h1[j] = hs;
        }
        if (it >= 2 && lane_out)
            *reinterpret_cast<uint2*>(out_lane + static_cast<int64_t>(yo0 + it - 2) * a.So * S0_CO) =
                pack4<DT>(y[0], y[1], y[2], y[3]);
    }
}

// ------------------------------------------------------------------------ MFMA stage
template <int DT, int CIN, int COUT, int PK, int PS, bool RES, int CTW>
__global__ __launch_bounds__(512) void stage_mfma_kernel(const StageArgs a) {
    using G = StageGeom<CIN>;
    constexpr int CP = G::CP, KC = G::KC, LPT = G::LPT;
    constexpr int CT = (COUT + 31) / 32;                 // 32-wide cout tiles in the stage
    constexpr int NG = COUT >= 32 ? 4 : COUT / 8;        // groups of 4 consecutive couts per lane half-row
    constexpr int TSTRIDE = tile_stride(PK, PS);
    constexpr int NOUT_T = tile_nout(PK, PS);
    constexpr int RING = PK ? PK - 1 : 0;
    constexpr int PIXB = CIN * 2;                        // bytes per pixel
    static_assert(CIN % 8 == 0 && COUT % 8 == 0, "channels must be multiples of 8");
    static_assert(CT % CTW == 0, "cout tiles must split evenly over workgroups");
    static_assert(!PK || PS == 1 || PS == 2, "pool stride 1 or 2");

    extern __shared__ __attribute__((aligned(16))) char smem[];
    const int tid = threadIdx.x;
    const int lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int r = lane & 31, hh = lane >> 5;
    const int nthreads = blockDim.x;
    const int npt = a.npt;

    int bid = blockIdx.x;
    const int ctg = bid % a.n_ctg;
    bid /= a.n_ctg;
    const int cb = bid % a.n_colblocks;
    const int band = bid / a.n_colblocks;
    const int n = blockIdx.y;

    const int ringcols = (npt - 1) * TSTRIDE + 34;
    const int rowbytes = ringcols * PIXB;
    char* const wl = smem;                                  // weights [KC][CTW][64] x 16 B
    char* const ring = smem + KC * CTW * 1024;              // NSLOT rows

    // rows of this band
    const int yo0 = band * a.rows_per_band;
    const int yo1 = min(a.Ho, yo0 + a.rows_per_band);
    const int yc0 = PK ? yo0 * PS : yo0;
    const int nconv = PK ? (yo1 - yo0 - 1) * PS + PK : (yo1 - yo0);
    const int nin = nconv + 2;
    // columns of this block
    const int x0c = cb * npt * TSTRIDE;                     // first conv / input column of the block
    const int xo_blk0 = PK ? x0c / PS : x0c;

    // ---- weights -> LDS (fragment order, lane linear)
    {
        const i32x4* src = a.wfrag;
        for (int i = tid; i < KC * CTW * 64; i += nthreads) {
            const int l = i & 63, t = i >> 6;
            const int ct = t % CTW, kc = t / CTW;
            reinterpret_cast<i32x4*>(wl)[i] = src[(kc * CT + ctg * CTW + ct) * 64 + l];
        }
    }

    // ---- input-row loader: thread owns up to LPT 16-byte chunks of a ring row
    const int nchunks = ringcols * CP;
    const unsigned short* const in_img = a.in + static_cast<int64_t>(n) * a.H * a.W * CIN;
    int ld_goff[LPT];     // element offset inside an input row, or -1 (zero fill / not mine)
    int ld_loff[LPT];     // byte offset inside a ring row
#pragma unroll
    for (int i = 0; i < LPT; ++i) {
        const int q = tid + i * nthreads;
        const int p = q / CP, c8 = q % CP;
        ld_loff[i] = q < nchunks ? (p * CP + (c8 ^ chunk_swz<CP>(p))) * 16 : -1;
        // columns past the image edge only feed discarded lanes: clamp instead of branching
        ld_goff[i] = (q < nchunks ? min(x0c + p, a.W - 1) : 0) * CIN + c8 * 8;
    }
    i32x4 pre[LPT];
    auto fetch_row = [&](int j) {   // input row yc0 + j -> registers
        const unsigned short* row = in_img + static_cast<int64_t>(yc0 + j) * a.W * CIN;
#pragma unroll
        for (int i = 0; i < LPT; ++i) pre[i] = *reinterpret_cast<const i32x4*>(row + ld_goff[i]);
    };
    auto store_row = [&](int j) {   // registers -> ring slot j % NSLOT
        char* dst = ring + (j & (NSLOT - 1)) * rowbytes;
#pragma unroll
        for (int i = 0; i < LPT; ++i)
            if (ld_loff[i] >= 0) *reinterpret_cast<i32x4*>(dst + ld_loff[i]) = pre[i];
    };
    for (int j = 0; j < 3; ++j) {
        fetch_row(j);
        store_row(j);
    }
    __syncthreads();

    // ---- per-lane constants of this wave's pixel tile
    const int xrel0 = wave * TSTRIDE + r;             // ring column of conv column (tap kx = 0)
    int boff[3];                                      // byte offset of pixel (xrel0 + kx) chunk 0
    int bswz[3];
#pragma unroll
    for (int kx = 0; kx < 3; ++kx) {
        boff[kx] = (xrel0 + kx) * PIXB;
        bswz[kx] = chunk_swz<CP>(xrel0 + kx);
    }
    const int xc = x0c + xrel0;                       // conv column of this lane
    const int xo = PK ? xc / PS : xc;                 // output column of this lane
    const bool lane_out = (PK ? (r % PS == 0 && r <= 32 - PK) : true) && xo < a.Wo &&
                          (xo - xo_blk0) < npt * NOUT_T;
    const int cout_lane = ctg * CTW * 32 + 4 * hh;    // + ct*32 + 8*g + j

    int rx_lo = 0, rx_hi = 0;
    float rx_l = 0.f;
    if constexpr (RES) {
        const int xq = min(xo, a.Wo - 1);
        rx_lo = a.rlo[xq];
        rx_hi = a.rhi[xq];
        rx_l = a.rlerp[xq];
    }

    float vring[RING > 0 ? RING : 1][CTW][16];
#pragma unroll
    for (int i = 0; i < (RING > 0 ? RING : 1); ++i)
#pragma unroll
        for (int ct = 0; ct < CTW; ++ct)
#pragma unroll
            for (int g = 0; g < 16; ++g) vring[i][ct][g] = 0.f;

    const char* const wl_lane = wl + lane * 16;

    for (int it = 0; it < nconv; ++it) {
        const bool have_next = it + 3 < nin;
        if (have_next) fetch_row(it + 3);

        // ---------------- implicit GEMM for conv row yc0 + it
        f32x16 acc[CTW];
#pragma unroll
        for (int ct = 0; ct < CTW; ++ct)
#pragma unroll
            for (int g = 0; g < 16; ++g) acc[ct][g] = 0.f;

        const char* rowp[3];
#pragma unroll
        for (int ky = 0; ky < 3; ++ky) rowp[ky] = ring + ((it + ky) & (NSLOT - 1)) * rowbytes;

        if constexpr (CIN >= 16) {
#pragma unroll
            for (int tap = 0; tap < 9; ++tap) {
                const int ky = tap / 3, kx = tap % 3;
                const char* pb = rowp[ky] + boff[kx];
#pragma unroll
                for (int cc = 0; cc < CIN / 16; ++cc) {
                    const int kc = tap * (CIN / 16) + cc;
                    const int c8 = cc * 2 + hh;
                    const i32x4 b = *reinterpret_cast<const i32x4*>(pb + ((c8 ^ bswz[kx]) << 4));
#pragma unroll
                    for (int ct = 0; ct < CTW; ++ct) {
                        const i32x4 wv = *reinterpret_cast<const i32x4*>(wl_lane + (kc * CTW + ct) * 1024);
                        acc[ct] = mfma32<DT>(wv, b, acc[ct]);
                    }
                }
            }
        } else {
            // CIN == 8: a 16-deep chunk spans two taps; the lane half selects the tap
#pragma unroll
            for (int kc = 0; kc < KC; ++kc) {
                int tap = 2 * kc + hh;
                tap = tap > 8 ? 8 : tap;                      // K padded 72 -> 80: weights are zero there
                const int ky = tap / 3, kx = tap - ky * 3;
                const char* pb = ring + ((it + ky) & (NSLOT - 1)) * rowbytes + (xrel0 + kx) * PIXB;
                const i32x4 b = *reinterpret_cast<const i32x4*>(pb);
#pragma unroll
                for (int ct = 0; ct < CTW; ++ct) {
                    const i32x4 wv = *reinterpret_cast<const i32x4*>(wl_lane + (kc * CTW + ct) * 1024);
                    acc[ct] = mfma32<DT>(wv, b, acc[ct]);
                }
            }
        }

        // ---------------- ReLU6 + horizontal pool sum (lanes) + vertical pool sum (register ring)
        const int lrow = it;
        bool emit;
        int yo;
        if constexpr (PK > 0) {
            emit = lrow >= PK - 1 && ((lrow - (PK - 1)) % PS) == 0;
            yo = yo0 + (lrow - (PK - 1)) / PS;
        } else {
            emit = true;
            yo = yo0 + lrow;
        }
#pragma unroll
        for (int ct = 0; ct < CTW; ++ct) {
            float hs[16];
#pragma unroll
            for (int g = 0; g < 16; ++g) {
                const float v = relu6f(acc[ct][g]);
                if constexpr (PK == 4) {
                    const float t = v + lane_next(v);
                    hs[g] = t + lane_next(lane_next(t));
                } else if constexpr (PK == 3) {
                    const float v1 = lane_next(v);
                    hs[g] = (v + v1) + lane_next(v1);
                } else if constexpr (PK == 2) {
                    hs[g] = v + lane_next(v);
                } else {
                    hs[g] = v;
                }
            }
            float tot[16];
#pragma unroll
            for (int g = 0; g < 16; ++g) {
                float s = hs[g];
                if constexpr (RING > 0) {
                    float t = vring[0][ct][g];
#pragma unroll
                    for (int i = 1; i < RING; ++i) t += vring[i][ct][g];
                    s = t + s;
#pragma unroll
                    for (int i = 0; i + 1 < RING; ++i) vring[i][ct][g] = vring[i + 1][ct][g];
                    vring[RING - 1][ct][g] = hs[g];
                }
                tot[g] = s;
            }
            if (emit) {
                // ---------------- BN (+ residual + BN) + store, 4 consecutive channels at a time
                constexpr float inv_area = PK ? 1.0f / static_cast<float>(PK * PK) : 1.0f;
                float yl = 0.f;
                const unsigned short* sk0 = nullptr;
                const unsigned short* sk1 = nullptr;
                if constexpr (RES) {
                    const int ylo = a.rlo[yo], yhi = a.rhi[yo];
                    yl = a.rlerp[yo];
                    const unsigned short* skn = a.skip + static_cast<int64_t>(n) * a.Ss * a.Ss * COUT;
                    sk0 = skn + static_cast<int64_t>(ylo) * a.Ss * COUT;
                    sk1 = skn + static_cast<int64_t>(yhi) * a.Ss * COUT;
                }
                unsigned short* orow = a.out + ((static_cast<int64_t>(n) * a.Ho + yo) * a.Wo + xo) * COUT;
#pragma unroll
                for (int g = 0; g < NG; ++g) {
                    const int c0 = cout_lane + ct * 32 + 8 * g;
                    const f32x4 mean = *reinterpret_cast<const f32x4*>(a.bn_mean + c0);
                    const f32x4 inv = *reinterpret_cast<const f32x4*>(a.bn_inv + c0);
                    const f32x4 beta = *reinterpret_cast<const f32x4*>(a.bn_beta + c0);
                    float y[4];
#pragma unroll
                    for (int j = 0; j < 4; ++j) y[j] = (tot[4 * g + j] * inv_area - mean[j]) * inv[j] + beta[j];
                    if constexpr (RES) {
                        if (lane_out) {
                            const f32x4 tl = unpack4<DT>(*reinterpret_cast<const uint2*>(sk0 + rx_lo * COUT + c0));
                            const f32x4 tr = unpack4<DT>(*reinterpret_cast<const uint2*>(sk0 + rx_hi * COUT + c0));
                            const f32x4 bl = unpack4<DT>(*reinterpret_cast<const uint2*>(sk1 + rx_lo * COUT + c0));
                            const f32x4 br = unpack4<DT>(*reinterpret_cast<const uint2*>(sk1 + rx_hi * COUT + c0));
                            const f32x4 mean2 = *reinterpret_cast<const f32x4*>(a.bn2_mean + c0);
                            const f32x4 inv2 = *reinterpret_cast<const f32x4*>(a.bn2_inv + c0);
                            const f32x4 beta2 = *reinterpret_cast<const f32x4*>(a.bn2_beta + c0);
#pragma unroll
                            for (int j = 0; j < 4; ++j) {
                                const float top = tl[j] + (tr[j] - tl[j]) * rx_l;
                                const float bot = bl[j] + (br[j] - bl[j]) * rx_l;
                                const float rs = top + (bot - top) * yl;
                                y[j] = ((y[j] + rs) - mean2[j]) * inv2[j] + beta2[j];
                            }
                        }
                    }
                    if (lane_out) *reinterpret_cast<uint2*>(orow + c0) = pack4<DT>(y[0], y[1], y[2], y[3]);
                }
            }
        }

        if (have_next) store_row(it + 3);
        __syncthreads();
    }
}

// ------------------------------------------------------------------------- host side
unsigned short f32_to_bf16(float f) {
    uint32_t u;
    std::memcpy(&u, &f, 4);
    if ((u & 0x7fffffffu) > 0x7f800000u) return static_cast<unsigned short>((u >> 16) | 0x40);   // NaN
    u += 0x7fffu + ((u >> 16) & 1u);
    return static_cast<unsigned short>(u >> 16);
}

unsigned short f32_to_f16(float f) {
    uint32_t x;
    std::memcpy(&x, &f, 4);
    const uint32_t sign = (x >> 16) & 0x8000u;
    x &= 0x7fffffffu;
    if (x >= 0x7f800000u) return static_cast<unsigned short>(sign | 0x7c00u | (x > 0x7f800000u ? 0x200u : 0));
    if (x >= 0x477ff000u) return static_cast<unsigned short>(sign | 0x7c00u);          // overflow -> inf
    if (x < 0x33000001u) return static_cast<unsigned short>(sign);                     // underflow -> 0
    int e = static_cast<int>(x >> 23) - 127;
    uint32_t m = (x & 0x7fffffu) | 0x800000u;
    int shift;
    if (e < -14) {
        shift = 13 + (-14 - e);
        e = -15;
    } else {
        shift = 13;
    }
    uint32_t half = m >> shift;
    const uint32_t rem = m & ((1u << shift) - 1), halfway = 1u << (shift - 1);
    if (rem > halfway || (rem == halfway && (half & 1))) ++half;
    uint32_t out;
    if (e == -15)
        out = half;                                   // subnormal (may carry into exponent 1)
    else
        out = (static_cast<uint32_t>(e + 15) << 10) + (half - 0x400u);
    return static_cast<unsigned short>(sign | out);
}

float bf16_bits_to_f32(unsigned short u) {
    const unsigned bits = static_cast<unsigned>(u) << 16;
    float f;
    std::memcpy(&f, &bits, 4);
    return f;
}

float f16_to_f32(unsigned short h) {
    const uint32_t sign = static_cast<uint32_t>(h & 0x8000u) << 16;
    const int e = (h >> 10) & 0x1f;
    const uint32_t m = h & 0x3ffu;
    float mag;
    if (e == 0)
        mag = std::ldexp(static_cast<float>(m), -24);                    // zero / subnormal
    else if (e == 31)
        mag = m ? std::nanf("") : INFINITY;
    else
        mag = std::ldexp(static_cast<float>(m | 0x400u), e - 25);
    uint32_t u;
    std::memcpy(&u, &mag, 4);
    u |= sign;
    std::memcpy(&mag, &u, 4);
    return mag;
}

// Cost of running `wgs` equal workgroups of `rows` row steps each with `slots` of them resident at a time, for the
// variants with several small workgroups per CU (they are back-filled as slots free up, so a launch does not run in
// whole rounds of the chip).  Fitted to band-count sweeps on the GPU (NOTES.md, rounds 1-2, "Band counts"):
//   * 2-wave workgroups, four per CU (32->64 stage): the fractional number of rounds plus an eighth of a round for the
//     ragged end, 1.5 row steps of prologue per workgroup (224: 2 bands; 600: 4);
//   * 4-wave workgroups, two per CU (64->128 stage): a partial last round costs at least 0.6 of a round (1.125 and 2.25
//     rounds measured as bad as 2 and 3), 3 row steps of prologue (224: 2 bands; 600: 5).
static double rn_backfill_cost(long wgs, long slots, int rows, int wgs_per_cu) {
    const double r = std::max(1.0, static_cast<double>(wgs) / static_cast<double>(slots));
    if (wgs_per_cu >= 4) return (r + 0.12) * (rows + 1.5);
    const double whole = std::floor(r), frac = r - whole;
    return (whole + (frac > 1e-9 ? std::max(frac, 0.6) : 0.0)) * (rows + 3.0);
}

#ifdef RN_CLOCK
// diagnostic build: every launch of a forward pass gets its own region of one host-visible buffer for the per-workgroup
// (delta s_memtime, delta s_memrealtime) pairs; nothing is synchronised or printed unless RN_CLOCK_REPORT is set in the
// environment WHEN the pass is enqueued (tools/gpu_clock.sh sets it for the last pass of a multi-second run), so the
// stamped pass runs back to back with the ones before it
struct ClockRegion {
    char what[32];
    size_t off, nwg;
};
static unsigned long long* g_clock_buf = nullptr;
static std::vector<ClockRegion> g_clock_regions;
static size_t g_clock_used = 0;
static unsigned long long* rn_clock_region(const char* what, size_t nwg) {
    if (!g_clock_buf) (void)hipHostMalloc(reinterpret_cast<void**>(&g_clock_buf), 16u << 20, 0);
    ClockRegion r{};
    snprintf(r.what, sizeof r.what, "%s", what);
    r.off = g_clock_used;
    r.nwg = nwg;
    g_clock_used += 2 * nwg;
    g_clock_regions.push_back(r);
    return g_clock_buf + r.off;
}
static void rn_clock_begin() {
    g_clock_regions.clear();
    g_clock_used = 0;
}
static void rn_clock_end(hipStream_t stream) {
    if (!getenv("RN_CLOCK_REPORT")) return;
    (void)hipStreamSynchronize(stream);
    for (const ClockRegion& r : g_clock_regions) {
        const unsigned long long* buf = g_clock_buf + r.off;
        std::vector<double> ghz, us;
        for (size_t k = 0; k < r.nwg; ++k)
            if (buf[2 * k + 1]) {
                ghz.push_back(static_cast<double>(buf[2 * k]) / static_cast<double>(buf[2 * k + 1]) * 0.1);
                us.push_back(static_cast<double>(buf[2 * k + 1]) * 0.01);
            }
        if (ghz.empty()) continue;
        std::sort(ghz.begin(), ghz.end());
        std::sort(us.begin(), us.end());
        fprintf(stderr, "[clock] %-12s in-kernel clock %.3f GHz (median of %zu workgroups; 10th / 90th percentile %.3f / %.3f), workgroup lifetime %.1f us median\n",
                r.what, ghz[ghz.size() / 2], ghz.size(), ghz[ghz.size() / 10], ghz[ghz.size() * 9 / 10], us[us.size() / 2]);
    }
}
#endif

struct FusedStage {
    bool use_rw = false;         // register-weights kernel (rn_stage_rw.hip) covers this stage
    bool use_c16 = false;        // 16x16x32-tile kernel (rn_conv16.hip) runs this stage instead
    bool use_c16p = false;       // ... its pooled 128 -> 16 sibling
    bool use_s5x = false;        // the 64 -> 64 residual stage on 16x16x32 tiles with row-register blocking (rn_stage5x.hip)
    bool use_s4x = false;        // the 32 -> 64 stage likewise (rn_stage4x.hip)
    bool use_s6x = false;        // the un-pooled 64 -> 128 stage likewise (rn_stage6x.hip)
    bool sixth = false;          // conv weights stored / 6, folded BN scale x 6 (pack2_relu6_sixth in the stage's kernel)
    i32x4* wfrag16 = nullptr;    // its weight fragments
    RwPlan rw;
    float* ptab = nullptr;       // folded BN tables for the rw kernel
    int variant = -1;            // index into the dispatch table
    int ctw = 1;                 // cout tiles per workgroup
    int npt = 1;                 // pixel tiles (= waves) per workgroup
    int n_colblocks = 1;
    size_t lds_bytes = 0;
    i32x4* wfrag = nullptr;
};

using LaunchFn = int (*)(hipStream_t, const StageArgs&, dim3, dim3, size_t);

template <int DT, int CIN, int COUT, int PK, int PS, bool RES, int CTW>
int launch_variant(hipStream_t s, const StageArgs& a, dim3 grid, dim3 block, size_t lds) {
    auto kern = stage_mfma_kernel<DT, CIN, COUT, PK, PS, RES, CTW>;
    static std::atomic<unsigned long long> attr_devices{0};     // per device, see launch_rw
    int dev = 0;
    RN_HIP(hipGetDevice(&dev));
    if (!(attr_devices.load(std::memory_order_acquire) >> (dev & 63) & 1ull)) {
        RN_HIP(hipFuncSetAttribute(reinterpret_cast<const void*>(kern), hipFuncAttributeMaxDynamicSharedMemorySize,
                                   160 * 1024));
        attr_devices.fetch_or(1ull << (dev & 63), std::memory_order_release);
    }
    hipLaunchKernelGGL(kern, grid, block, lds, s, a);
    RN_CHECK_LAUNCH();
    return RN_OK;
}

struct Variant {
    int cin, cout, pk, ps, res, ctw;
    LaunchFn fn[2];   // [bf16, f16]
};

#define RN_VARIANT(CIN, COUT, PK, PS, RES, CTW)                                                   \
    {                                                                                             \
        CIN, COUT, PK, PS, RES, CTW, {                                                            \
            launch_variant<RN_DTYPE_BF16, CIN, COUT, PK, PS, RES != 0, CTW>,                      \
                launch_variant<RN_DTYPE_F16, CIN, COUT, PK, PS, RES != 0, CTW>                    \
        }                                                                                         \
    }

const Variant kVariants[] = {
    RN_VARIANT(8, 32, 4, 1, 0, 1),    // stage 1
    RN_VARIANT(32, 32, 4, 1, 0, 1),   // stage 2
    RN_VARIANT(32, 32, 4, 1, 1, 1),   // stage 3 (+ residual)
    RN_VARIANT(32, 64, 4, 2, 0, 2),   // stage 4
    RN_VARIANT(64, 64, 4, 2, 1, 2),   // stage 5 (+ residual)
    RN_VARIANT(64, 128, 0, 1, 0, 2),  // stage 6 (no pool; cout tiles split over 2 workgroups)
    RN_VARIANT(128, 16, 4, 2, 0, 1),  // stage 7
    RN_VARIANT(16, 16, 4, 2, 0, 1),   // stage 8
    RN_VARIANT(16, 16, 4, 2, 1, 1),   // stage 9 (+ residual)
};
constexpr int kNumVariants = sizeof(kVariants) / sizeof(kVariants[0]);

struct FusedState {
    std::vector<FusedStage> st;
    // cross-stage fusion: launch_rep[i] = the stage under which the launch that computes stage i reports its time
    // (== i for a launch of its own)
    std::vector<int> launch_rep;
    bool fuse_s0 = false;        // stage 0 is computed inside stage 1's kernel (rn_stage_rw.hip, S0F)
    bool use_tail = false;       // last two stages + head in one launch (rn_tail.hip)
    // the whole back end (stage 6 -> 7 -> 8 -> 9 -> head) in one launch, one workgroup per image (rn_backend.hip): taken when the
    // batch fills at least half the chip; smaller batches keep the launches that cut an image into bands
    bool use_backend = false;
    bool last_backend = false;   // the last forward pass ran it: stages 6 and 7 were not written
    // cross-stage fused pair (rn_stage23.hip): stages pair_first, pair_first + 1 run as one launch
    int pair_first = -1;
    float* pair_ptab = nullptr;  // [5][32]: the first stage's scale, shift | the second stage's scale', shift', scale2
    bool pair_x16 = false;       // the pair runs on 16x16x32 tiles (rn_stage23x.hip) with the fragments below
    i32x4* pair_wfrag_a = nullptr;
    i32x4* pair_wfrag_b = nullptr;
    // frozen first-BN channels of the 64 -> 64 residual stage (rn_fused_prepare): its index (or -1), the 16-cout quarters whose
    // convolution still runs, and the channel relabelling of the tensors it touches (node id -> position p holds channel perm[p])
    int fold5_stage = -1;
    int fold5_live_q = 4;
    // constant channels of the stage in front of it (round 6; rn_fused_prepare): the relabelling puts 16 channels whose 16-BIT
    // STORE is one number for every input into the last cout quarter of that stage -- all of them frozen channels of the residual
    // stage too, so the same positions of the residual stage's output are constants as well.  Neither kernel computes them: both
    // tensors are filled once (rn_fused_post_alloc), the residual stage contracts 48 input channels and starts its accumulators
    // from the constants' contribution.
    // round 6: refined rounding of the 16-bit handles (default; off under RN_FLAG_NO_DITHER and on the legacy comparison arms):
    // `refine` = conv weights rounded with the residual carried from tap to tap (diffuse_taps), `dither` (bf16 only) = the stores of
    // the large stage outputs go through v_cvt_sr_bf16_f32 with a seed that depends on the output row (rn_stage.h)
    bool refine = false;
    bool dither = false;
    std::vector<char> dither_out;    // per conv stage: its output rows are dithered
    int relabel_stage = -1;          // the residual stage whose channels (and its neighbours') are stored relabelled -- also on the
                                     // arm that computes every channel (same channel order in both arms: same MFMA summation order)
    bool const_layout = false;       // positions 48..63 of the two relabelled tensors hold the constant channels (both arms)
    int const4_proven = 0;           // channels of that stage with a constant 16-bit store on this handle
    bool const4 = false;             // 16 of them sit in positions 48..63 and are not computed
    unsigned short const4_val[16] = {};   // their stored values (s4.bn positions 48..63)
    unsigned short const5_val[16] = {};   // ... and of the residual stage's output (s5.bn2 positions 48..63)
    unsigned short* const_vals_dev = nullptr;   // [2][16] on the device: const4_val | const5_val (StageArgs::cvals)
    i32x4* s6_wfrag48 = nullptr;     // the stage behind the residual stage without ITS constant input channels (rn_stage6x_pack48) ...
    float* s6_cstart = nullptr;      // ... and their contribution [128]
    float* s5_cstart = nullptr;      // [64] what the 16 constant input channels add to every conv output of the residual stage
    i32x4* s5_wfrag48 = nullptr;     // the residual stage's fragments without them (rn_stage5x_pack48)
    std::map<int, std::vector<int>> node_perm;
    float* pair_ptab_x = nullptr;    // rn_stage23x.hip's table: pair_ptab with the first stage's channels in the B ring's order
    int pair_producer_halves = 2;    // 1: 16 channels of the pair's on-chip tensor are frozen and not computed (Stage23Args)
    int pair_narrow = 0;             // with it: 1 = the ring holds 16 channels, 2 = eight (24 constant channels; round 6)
    int pair_frozen = 0;             // how many channels of it are frozen on this handle
    // stage 0
    i32x4* s0_wfrag = nullptr;
    float* s0_ptab = nullptr;
};

}  // namespace

void rn_fused_release(rn_handle* h) {
    delete static_cast<FusedState*>(h->fused);
    h->fused = nullptr;
}

// Conv weights [tap][cin][cout] -> the handle's 16-bit values with the rounding residual CARRIED from tap to tap (centre first): the
// nine taps of a (cin, cout) pair then sum to the exact sum within half an ulp of ONE weight instead of the sum of nine independent
// roundings -- smooth image content sees the tap sum (on the parity set the weight share of the bf16 logit error goes from 0.061 to
// 0.007, tools/sim16.py).  The array is rewritten with exactly representable values: every later conversion is exact, every kernel
// family packs the same numbers.
static void diffuse_taps(float* w, int cin, int cout, int dtype) {
    static const int order[9] = {4, 0, 8, 2, 6, 1, 7, 3, 5};
    for (int ci = 0; ci < cin; ++ci)
        for (int co = 0; co < cout; ++co) {
            float carry = 0.f;
            for (int t : order) {
                float& x = w[(static_cast<size_t>(t) * cin + ci) * cout + co];
                const float v = x + carry;
                const float q = dtype == RN_DTYPE_BF16 ? bf16_bits_to_f32(f32_to_bf16(v)) : f16_to_f32(f32_to_f16(v));
                carry = std::isfinite(q) ? v - q : 0.f;
                x = q;
            }
        }
}

int rn_fused_prepare(rn_handle* h, const rn_weights* w_in) {
    const int dti = h->dtype == RN_DTYPE_BF16 ? 0 : 1;
    (void)dti;
    auto* fs = new FusedState();
    fs->st.resize(h->stages.size());
    h->fused = fs;
    fs->refine = !(h->flags & (RN_FLAG_GENERIC_KERNELS | RN_FLAG_PAIR_32X32 | RN_FLAG_NO_DITHER));
    fs->dither = fs->refine && h->dtype == RN_DTYPE_BF16;
    fs->dither_out.assign(h->stages.size(), 0);
    {
        // dithered outputs: the large stage tensors in front of the back end (the last four stages run in one launch per image and
        // keep their tensors in LDS), except stage 0 (it lives in LDS rings inside stage 1's kernel) and the first stage of a
        // fusable 32 -> 32 pair (its output is the pair's on-chip tensor; the frozen-channel fold relies on its plain rounding)
        const int ns = static_cast<int>(h->stages.size());
        for (int i = 1; i + 4 < ns && fs->dither; ++i) {
            const StagePlan& s = h->stages[i];
            const bool pair_first = i + 1 < ns && s.cin == 32 && s.cout == 32 && s.pool_k == 4 && s.pool_s == 1 && s.skip_stage < 0 &&
                                    h->stages[i + 1].cin == 32 && h->stages[i + 1].cout == 32 && h->stages[i + 1].skip_stage == i - 1;
            // (the stages whose kernels carry the SR store: 32+ channels in and out, residual or stride-2 pooling -- stages 3, 4, 5;
            //  the first block's first step stays plain: its 1.5 M values per image cost more as SR stores than their dither returned)
            const bool srp = s.cin >= 32 && s.cout >= 32 && (s.skip_stage >= 0 || s.pool_s == 2);
            fs->dither_out[i] = (pair_first || !srp) ? 0 : 1;
        }
    }
    // the handle's 16-bit store of a value whose rounding is not dithered (constant channels, tables)
    const auto cv_rne = [&](float v) -> unsigned short { return h->dtype == RN_DTYPE_BF16 ? f32_to_bf16(v) : f32_to_f16(v); };
    // ... and of the tensors whose kernels store through v_cvt_sr_bf16_f32 (the 64-channel block's two outputs)
    const auto cv_store = [&](float v) -> unsigned short {
        // (bf16: the row-blocked stage kernels store through v_cvt_sr_bf16_f32 whether the handle dithers or not)
        if (h->dtype == RN_DTYPE_BF16) return rn_sr_bf16_host(v, RN_SEED_PLAIN);
        return h->dtype == RN_DTYPE_BF16 ? f32_to_bf16(v) : f32_to_f16(v);
    };
    // ---- frozen first-BN channels of a 64 -> 64 residual stage (stage 5 of the network; rn_stage5x.hip).  Its epilogue forms
    // y1 = fma(H, sc1', sh1') with H = a pooled sum of ReLU6 / 6 values in [0, 16]: where |sc1'| * 16 < 2^-25 |sh1'| the fma
    // returns sh1' EXACTLY in float32 for every input -- the channel's convolution cannot change a bit of the output (the
    // reference's L2 regulariser drove 44 of the 64 gammas of the shipped checkpoint to ~1e-30).  With >= 32 such channels the
    // channels of this stage's output are RELABELLED (a permutation of the weights of this stage, of the cout of the stage
    // before -- whose output is this stage's input AND its skip tensor, paired channel by channel -- and of the cin of the stage
    // behind) so that the last two 16-cout quarters are all frozen: their waves skip the convolution and its pooling.  The
    // relabelling happens HERE, on a copy of the weight arrays, in front of everything else: every kernel family of every arm
    // sees one consistent network; rn_tap puts the channels of the two affected tensors back in the reference's order.
    std::vector<rn_conv_stage> stg(w_in->stages, w_in->stages + w_in->n_stages);
    std::vector<std::vector<float>> owned;
    rn_weights wp = *w_in;
    wp.stages = stg.data();
    const rn_weights* const w = &wp;
    const bool fold_ok = !(h->flags & RN_FLAG_COMPUTE_FROZEN);      // (the computing arm keeps the relabelling and folds nothing)
    if (!(h->flags & (RN_FLAG_GENERIC_KERNELS | RN_FLAG_PAIR_32X32))) {
        for (int r = 2; r + 1 < w_in->n_stages; ++r) {
            const rn_conv_stage& s5 = w_in->stages[r];
            const rn_conv_stage& s4 = w_in->stages[r - 1];
            const rn_conv_stage& s6 = w_in->stages[r + 1];
            const StagePlan& p5 = h->stages[r];
            if (!(s5.cin == 64 && s5.cout == 64 && s5.pool_k == 4 && s5.pool_s == 2 && s5.skip_stage == r - 1 && s5.gamma2 && s4.cout == 64 &&
                  s4.skip_stage < 0 && s6.cin == 64 && s6.skip_stage < 0 &&
                  rn_stage5x_supported(s5.cin, s5.cout, s5.pool_k, s5.pool_s, true, p5.in_side, p5.skip_side)))
                continue;
            bool other_use = false;          // nobody else may pair with the relabelled tensors
            for (int k = 0; k < w_in->n_stages; ++k) other_use |= (k != r && w_in->stages[k].skip_stage == r - 1) || w_in->stages[k].skip_stage == r;
            if (other_use) continue;
            std::vector<int> frozen, live;
            for (int c = 0; c < 64; ++c) {
                // the table values exactly as the stage loop below builds them (float arithmetic, `sixth` weights)
                const float inv = (1.0f / sqrtf(s5.variance[c] + w_in->bn_epsilon)) * s5.gamma[c];
                float t0 = inv / 16.0f, t1 = s5.beta[c] - s5.mean[c] * inv;
                const float inv2 = (1.0f / sqrtf(s5.variance2[c] + w_in->bn_epsilon)) * s5.gamma2[c];
                const float sh2 = s5.beta2[c] - s5.mean2[c] * inv2;
                t1 = t1 * inv2 + sh2;
                t0 = t0 * inv2;
                t0 *= 6.0f;
                const bool fz = static_cast<double>(std::fabs(t0)) * 16.0 * (1.0 + 1e-6) < static_cast<double>(std::fabs(t1)) * 2.98023223876953125e-8;      // 2^-25
                (fz ? frozen : live).push_back(c);
            }
            if (frozen.size() < 32) continue;
            // ---- constant channels of the stage in front (round 6).  Its kernel (rn_stage4x.hip) stores
            // pack2<DT>(fma(H, sc, sh)) with H = a pooled sum of ReLU6 / 6 values in [0, 16]; fma and the 16-bit conversion are
            // monotone in H, so where the two ends H = 0 and H = 16 convert to the same 16-bit number every input does: the
            // stored channel is that number at every pixel of every image, bit for bit what the kernel that computes it stores
            // (the shipped checkpoint: 26 channels in bf16, 23 in fp16 -- its L2 regulariser left their BN scale below half an
            // ulp of the shift -- all of them among this stage's frozen channels).  Table values as the stage loop below builds them.
            std::vector<int> cst;
            std::vector<unsigned short> cst_val(64, 0);
            const bool s4x_ok = rn_stage4x_supported(s4.cin, s4.cout, s4.pool_k, s4.pool_s, false, h->stages[r - 1].in_side);
            for (int c = 0; c < 64 && s4x_ok; ++c) {
                const float inv = (1.0f / sqrtf(s4.variance[c] + w_in->bn_epsilon)) * s4.gamma[c];
                float sc = inv / 16.0f;
                const float sh = s4.beta[c] - s4.mean[c] * inv;
                sc *= 6.0f;
                const unsigned short v0 = cv_store(std::fmaf(0.0f, sc, sh)), v16 = cv_store(std::fmaf(16.0f, sc, sh));
                cst_val[c] = v0;
                if (v0 == v16 && std::isfinite(sh)) cst.push_back(c);
            }
            fs->const4_proven = static_cast<int>(cst.size());
            {
                // frozen channels that are constants of the stage in front go LAST (positions 48..63 when there are 16 of them)
                std::vector<int> both, only;
                for (int c : frozen) (std::find(cst.begin(), cst.end(), c) != cst.end() ? both : only).push_back(c);
                fs->const_layout = both.size() >= 16;
                fs->const4 = fs->const_layout && fold_ok;
                frozen = only;
                frozen.insert(frozen.end(), both.begin(), both.end());
            }
            std::vector<int> pi(64);
            while (frozen.size() > 32) {             // (the spare frozen channels -- taken from the front -- are computed like live ones)
                live.push_back(frozen.front());
                frozen.erase(frozen.begin());
            }
            std::sort(live.begin(), live.end());
            for (int p = 0; p < 32; ++p) pi[p] = live[p];
            for (int p = 0; p < 32; ++p) pi[32 + p] = frozen[p];
            if (fs->const_layout)
                for (int p = 0; p < 16; ++p) fs->const4_val[p] = cst_val[pi[48 + p]];
            auto perm_vec = [&](const float* src) -> const float* {
                if (!src) return nullptr;
                owned.emplace_back(64);
                for (int p = 0; p < 64; ++p) owned.back()[p] = src[pi[p]];
                return owned.back().data();
            };
            auto perm_kernel = [&](const float* src, int cin, int cout, bool pin, bool pout) -> const float* {
                owned.emplace_back(static_cast<size_t>(9) * cin * cout);
                std::vector<float>& dst = owned.back();
                for (int tap = 0; tap < 9; ++tap)
                    for (int ci = 0; ci < cin; ++ci)
                        for (int co = 0; co < cout; ++co)
                            dst[(static_cast<size_t>(tap) * cin + ci) * cout + co] = src[(static_cast<size_t>(tap) * cin + (pin ? pi[ci] : ci)) * cout + (pout ? pi[co] : co)];
                return dst.data();
            };
            stg[r - 1].kernel = perm_kernel(s4.kernel, s4.cin, 64, false, true);
            stg[r - 1].gamma = perm_vec(s4.gamma);
            stg[r - 1].beta = perm_vec(s4.beta);
            stg[r - 1].mean = perm_vec(s4.mean);
            stg[r - 1].variance = perm_vec(s4.variance);
            stg[r].kernel = perm_kernel(s5.kernel, 64, 64, true, true);
            stg[r].gamma = perm_vec(s5.gamma);
            stg[r].beta = perm_vec(s5.beta);
            stg[r].mean = perm_vec(s5.mean);
            stg[r].variance = perm_vec(s5.variance);
            stg[r].gamma2 = perm_vec(s5.gamma2);
            stg[r].beta2 = perm_vec(s5.beta2);
            stg[r].mean2 = perm_vec(s5.mean2);
            stg[r].variance2 = perm_vec(s5.variance2);
            stg[r + 1].kernel = perm_kernel(s6.kernel, 64, s6.cout, true, false);
            fs->relabel_stage = r;
            if (fold_ok) {
                fs->fold5_stage = r;
                fs->fold5_live_q = 2;
            }
            fs->node_perm[h->stages[r - 1].node_bn] = pi;
            fs->node_perm[h->stages[r].node_bn2] = pi;
            break;
        }
    }
    if (h->stages[0].cin != 3 || h->stages[0].cout != S0_CO || h->stages[0].pool_k != 3 ||
        h->stages[0].pool_s != 1 || h->stages[0].skip_stage >= 0) {
        rn_set_error("16-bit path: stage 0 must be conv(3->8) + pool 3/1 (got %d->%d pool %d/%d)", h->stages[0].cin,
                     h->stages[0].cout, h->stages[0].pool_k, h->stages[0].pool_s);
        return RN_E_INVALID;
    }
    {
        // stage-0 tables (s0_pixel_halves, rn_stage.h): A fragments with K = (ky, kx<4, c<4) of the folded weights
        // 2^8 (2 w / 255) as fp16 hi (cout rows 0..7) + lo (rows 8..15) pairs with the constant -2^8 sum(w) in the fourth
        // channel slot of (ky, kx) = (0, 0); folded BN
        std::vector<unsigned short> frag(3 * 64 * 8, 0);
        const float* w0 = w->stages[0].kernel;       // [ky][kx][c][cout]
        for (int ky = 0; ky < 3; ++ky)
            for (int l = 0; l < 64; ++l)
                for (int j = 0; j < 8; ++j) {
                    const int kk = 8 * (l >> 5) + j, kx = kk / 4, c = kk % 4, row = l & 31, co = row & 7, part = row >> 3;
                    if (part >= 2) continue;
                    double v;
                    if (kx < 3 && c < 3) {
                        v = static_cast<double>(w0[((ky * 3 + kx) * 3 + c) * S0_CO + co]) * (2.0 / 255.0) * S0_WSCALE;
                    } else if (ky == 0 && kx == 0 && c == 3) {      // the constant slot: B = 1.0
                        v = 0.0;
                        for (int t = 0; t < 27; ++t) v -= static_cast<double>(w0[t * S0_CO + co]);
                        v *= S0_WSCALE;
                    } else {
                        continue;
                    }
                    // the folded values live in fp16 hi + lo pairs: a checkpoint whose stage-0 weights (or their sum over the
                    // 27 taps) reach 65504 / 2^8 = 255.9 would turn into infinities and NaN outputs without a word.  (Small
                    // weights are safe: hi + lo resolves 2^-24 absolutely, below the fp32 ulp of any weight above 2^-1.)
                    if (!(std::fabs(v) < 65504.0)) {
                        rn_set_error("16-bit path: stage-0 weight fold out of fp16 range (cout %d: |%g| >= 65504 after the 2^8 scale; "
                                     "conv2d/kernel must stay below ~255 per weight and per 27-tap sum) -- use RN_DTYPE_F32", co, v);
                        return RN_E_INVALID;
                    }
                    const unsigned short hi = f32_to_f16(static_cast<float>(v));
                    const unsigned short lo = f32_to_f16(static_cast<float>(v - static_cast<double>(f16_to_f32(hi))));
                    frag[(ky * 64 + l) * 8 + j] = part == 0 ? hi : lo;
                }
        const rn_conv_stage& ws = w->stages[0];
        std::vector<float> tab(16);
        for (int c = 0; c < S0_CO; ++c) {
            const float inv = (1.0f / sqrtf(ws.variance[c] + w->bn_epsilon)) * ws.gamma[c];
            tab[c] = inv / 9.0f / S0_WSCALE;
            tab[8 + c] = ws.beta[c] - ws.mean[c] * inv;
        }
        void* d = nullptr;
        auto up = [&](const void* src, size_t bytes, void** out) -> int {
            hipError_t e = hipMalloc(out, bytes);
            if (e != hipSuccess) {
                rn_set_error("hipMalloc(stage 0 tables) failed: %s", hipGetErrorString(e));
                return RN_E_NOMEM;
            }
            h->allocs.push_back(*out);
            RN_HIP(hipMemcpy(*out, src, bytes, hipMemcpyHostToDevice));
            return RN_OK;
        };
        int rc;
        if ((rc = up(frag.data(), frag.size() * 2, &d)) != RN_OK) return rc;
        fs->s0_wfrag = static_cast<i32x4*>(d);
        if ((rc = up(tab.data(), tab.size() * 4, &d)) != RN_OK) return rc;
        fs->s0_ptab = static_cast<float*>(d);
    }
    for (size_t i = 1; i < h->stages.size(); ++i) {
        StagePlan& s = h->stages[i];
        FusedStage& f = fs->st[i];
        for (int v = 0; v < kNumVariants; ++v) {
            const Variant& k = kVariants[v];
            if (k.cin == s.cin && k.cout == s.cout && k.pk == s.pool_k && (s.pool_k == 0 || k.ps == s.pool_s) &&
                k.res == (s.skip_stage >= 0 ? 1 : 0)) {
                f.variant = v;
                break;
            }
        }
        if (f.variant < 0) {
            rn_set_error("16-bit path: no kernel variant for stage %zu (cin %d cout %d pool %d/%d res %d)", i, s.cin,
                         s.cout, s.pool_k, s.pool_s, s.skip_stage >= 0);
            return RN_E_INVALID;
        }
        f.use_rw = rn_rw_supported(s.cin, s.cout, s.pool_k, s.pool_s, s.skip_stage >= 0, s.out_side, s.skip_side,
                                   &f.rw) && !(h->flags & RN_FLAG_GENERIC_KERNELS);
        // Stages whose kernel pools fp16 ReLU6 outputs on the matrix cores (the pool 4/1 register-weights variants and the
        // cross-stage kernels built on them; rn_stage4x / rn_stage5x) store their conv weights divided by 6: the ReLU6 is
        // then the free [0, 1] clamp of the fp16 conversion (pack2_relu6_sixth) and the folded BN scale carries the 6.
        const bool want_s4x = f.use_rw && !(h->flags & (RN_FLAG_GENERIC_KERNELS | RN_FLAG_PAIR_32X32)) &&
                              rn_stage4x_supported(s.cin, s.cout, s.pool_k, s.pool_s, s.skip_stage >= 0, s.in_side);
        const bool want_s5x = f.use_rw && !(h->flags & (RN_FLAG_GENERIC_KERNELS | RN_FLAG_PAIR_32X32)) && s.skip_stage >= 0 &&
                              rn_stage5x_supported(s.cin, s.cout, s.pool_k, s.pool_s, true, s.in_side, s.skip_side) &&
                              s.skip_stage == static_cast<int>(i) - 1 && h->stages[s.skip_stage].node_bn2 < 0;
        f.sixth = f.use_rw && ((s.pool_k == 4 && s.pool_s == 1) || want_s4x || want_s5x);
        if (f.use_rw) {
            // y = S * (inv / k^2) + (beta - mean * inv);  y2 = (y + r) * inv2 + (beta2 - mean2 * inv2)
            const rn_conv_stage& ws = w->stages[i];
            std::vector<float> tab(static_cast<size_t>(4) * s.cout, 0.f);
            for (int c = 0; c < s.cout; ++c) {
                const float inv = (1.0f / sqrtf(ws.variance[c] + w->bn_epsilon)) * ws.gamma[c];
                tab[c] = s.pool_k ? inv / static_cast<float>(s.pool_k * s.pool_k) : inv;
                tab[s.cout + c] = ws.beta[c] - ws.mean[c] * inv;
                if (s.skip_stage >= 0) {
                    // residual stages: y2 = (S * sc1 + sh1 + R) * sc2 + sh2 = S * (sc1 sc2) + R * sc2 + (sh1 sc2 + sh2):
                    // tables 0/1 hold the products, so the epilogue is two fmas around the resized skip value
                    const float inv2 = (1.0f / sqrtf(ws.variance2[c] + w->bn_epsilon)) * ws.gamma2[c];
                    const float sh2 = ws.beta2[c] - ws.mean2[c] * inv2;
                    tab[2 * s.cout + c] = inv2;
                    tab[3 * s.cout + c] = sh2;
                    tab[s.cout + c] = tab[s.cout + c] * inv2 + sh2;
                    tab[c] = tab[c] * inv2;
                }
                if (f.sixth) tab[c] *= 6.0f;
            }
            void* dt = nullptr;
            hipError_t e2 = hipMalloc(&dt, tab.size() * 4);
            if (e2 != hipSuccess) {
                rn_set_error("hipMalloc(ptab) failed: %s", hipGetErrorString(e2));
                return RN_E_NOMEM;
            }
            h->allocs.push_back(dt);
            RN_HIP(hipMemcpy(dt, tab.data(), tab.size() * 4, hipMemcpyHostToDevice));
            f.ptab = static_cast<float*>(dt);
            if (fs->const4 && static_cast<int>(i) == fs->fold5_stage && s.cout == 64) {
                // the residual stage's output at the positions of the constant channels: y1 = fma(0, sc1', sh1') = sh1' (frozen first
                // BN), the bilinear resize of a constant channel is the constant (its two weights are exact 16-bit numbers that
                // sum to 1, the products are exact in float32), y = fma(v, sc2, y1) -- what rn_stage5x.hip computes for them
                for (int p = 0; p < 16; ++p) {
                    const float v = h->dtype == RN_DTYPE_BF16 ? bf16_bits_to_f32(fs->const4_val[p]) : f16_to_f32(fs->const4_val[p]);
                    fs->const5_val[p] = cv_store(std::fmaf(v, tab[2 * 64 + 48 + p], tab[64 + 48 + p]));
                }
            }
        }
        const Variant& k = kVariants[f.variant];
        f.ctw = k.ctw;
        const int tstride = tile_stride(s.pool_k, s.pool_s), nout_t = tile_nout(s.pool_k, s.pool_s);
        const int tiles = (s.out_side + nout_t - 1) / nout_t;
        const int kc = (9 * s.cin + 15) / 16;
        // generic kernel: as many pixel tiles per workgroup as fit the LDS next to the weights
        f.npt = tiles >= 8 ? 8 : tiles;
        for (;;) {
            const int ringcols = (f.npt - 1) * tstride + 34;
            f.lds_bytes = static_cast<size_t>(kc) * f.ctw * 1024 + static_cast<size_t>(NSLOT) * ringcols * s.cin * 2;
            if (f.lds_bytes <= 160 * 1024 || f.npt == 1) break;
            --f.npt;
        }
        f.n_colblocks = (tiles + f.npt - 1) / f.npt;
        if (!f.use_rw && f.lds_bytes > 160 * 1024) {
            rn_set_error("16-bit path: stage %zu needs %zu bytes of LDS", i, f.lds_bytes);
            return RN_E_INVALID;
        }
        // pack weights: frag[kc][ct][lane][j] = W[k = kc*16 + 8*(lane>>5) + j][cout = ct*32 + (lane&31)]
        const int ct_n = (s.cout + 31) / 32;
        std::vector<unsigned short> frag(static_cast<size_t>(kc) * ct_n * 64 * 8, 0);
        const float* wsrc = w->stages[i].kernel;   // HWIO == [k = tap*cin + c][cout]
        const int K = 9 * s.cin;
        std::vector<float> wsixth;
        if (f.sixth || fs->refine) {
            wsixth.assign(wsrc, wsrc + static_cast<size_t>(K) * s.cout);
            if (f.sixth)
                for (float& v : wsixth) v /= 6.0f;
#ifndef RN_REFINE_UPTO
#define RN_REFINE_UPTO 99        // (diagnostic builds: the last conv stage whose weights get the carried rounding)
#endif
            if (fs->refine && static_cast<int>(i) <= RN_REFINE_UPTO) diffuse_taps(wsixth.data(), s.cin, s.cout, h->dtype);      // (every pack below converts exactly)
            wsrc = wsixth.data();
        }
        for (int c = 0; c < kc; ++c)
            for (int t = 0; t < ct_n; ++t)
                for (int l = 0; l < 64; ++l)
                    for (int j = 0; j < 8; ++j) {
                        const int kk = c * 16 + 8 * (l >> 5) + j, co = t * 32 + (l & 31);
                        float v = 0.f;
                        if (kk < K && co < s.cout) v = wsrc[static_cast<size_t>(kk) * s.cout + co];
                        frag[((static_cast<size_t>(c) * ct_n + t) * 64 + l) * 8 + j] =
                            h->dtype == RN_DTYPE_BF16 ? f32_to_bf16(v) : f32_to_f16(v);
                    }
        void* d = nullptr;
        hipError_t e = hipMalloc(&d, frag.size() * 2);
        if (e != hipSuccess) {
            rn_set_error("hipMalloc(weights) failed: %s", hipGetErrorString(e));
            return RN_E_NOMEM;
        }
        h->allocs.push_back(d);
        RN_HIP(hipMemcpy(d, frag.data(), frag.size() * 2, hipMemcpyHostToDevice));
        f.wfrag = static_cast<i32x4*>(d);
        // the un-pooled 64 -> 128 stage runs on 16x16x32 tiles (rn_conv16.hip) unless the comparison flags ask for the
        // one-kernel-family paths
        if (f.use_rw && f.ptab && rn_conv16_supported(s.cin, s.cout, s.pool_k, s.skip_stage >= 0) &&
            !(h->flags & RN_FLAG_GENERIC_KERNELS)) {
            std::vector<unsigned short> f16;
            rn_conv16_pack(wsrc, h->dtype, f32_to_bf16, f32_to_f16, &f16);
            void* d16 = nullptr;
            if (hipMalloc(&d16, f16.size() * 2) != hipSuccess) {
                rn_set_error("hipMalloc(conv16 weights) failed");
                return RN_E_NOMEM;
            }
            h->allocs.push_back(d16);
            RN_HIP(hipMemcpy(d16, f16.data(), f16.size() * 2, hipMemcpyHostToDevice));
            f.wfrag16 = static_cast<i32x4*>(d16);
            f.use_c16 = true;
        }
        if (f.use_rw && f.ptab && !(h->flags & (RN_FLAG_GENERIC_KERNELS | RN_FLAG_PAIR_32X32)) &&
            rn_stage6x_supported(s.cin, s.cout, s.pool_k, s.skip_stage >= 0, s.in_side)) {
            std::vector<unsigned short> f16;
            rn_stage6x_pack(wsrc, h->dtype, f32_to_bf16, f32_to_f16, &f16);
            void* d16 = nullptr;
            if (hipMalloc(&d16, f16.size() * 2) != hipSuccess) {
                rn_set_error("hipMalloc(stage6x weights) failed");
                return RN_E_NOMEM;
            }
            h->allocs.push_back(d16);
            RN_HIP(hipMemcpy(d16, f16.data(), f16.size() * 2, hipMemcpyHostToDevice));
            f.wfrag16 = static_cast<i32x4*>(d16);
            f.use_s6x = true;
            f.use_c16 = false;
            if (fs->const4 && static_cast<int>(i) == fs->fold5_stage + 1 && s.cin == 64 && s.cout == 128) {
                // the residual stage's output channels 48..63 are constants (const5_val): this stage contracts 48 input channels and
                // starts from their contribution, like the residual stage itself
                rn_stage6x_pack48(wsrc, h->dtype, f32_to_bf16, f32_to_f16, &f16);
                void* d48 = nullptr;
                if (hipMalloc(&d48, f16.size() * 2) != hipSuccess) {
                    rn_set_error("hipMalloc(stage6x weights) failed");
                    return RN_E_NOMEM;
                }
                h->allocs.push_back(d48);
                RN_HIP(hipMemcpy(d48, f16.data(), f16.size() * 2, hipMemcpyHostToDevice));
                fs->s6_wfrag48 = static_cast<i32x4*>(d48);
                const auto cv = [&](float v) { return h->dtype == RN_DTYPE_BF16 ? f32_to_bf16(v) : f32_to_f16(v); };
                const auto bk = [&](unsigned short u) { return h->dtype == RN_DTYPE_BF16 ? bf16_bits_to_f32(u) : f16_to_f32(u); };
                std::vector<float> cst(128);
                for (int co = 0; co < 128; ++co) {
                    double sum = 0.0;
                    for (int tap = 0; tap < 9; ++tap)
                        for (int p = 48; p < 64; ++p)
                            sum += static_cast<double>(bk(cv(wsrc[(static_cast<size_t>(tap) * 64 + p) * 128 + co]))) * static_cast<double>(bk(fs->const5_val[p - 48]));
                    cst[co] = static_cast<float>(sum);
                }
                void* dc = nullptr;
                if (hipMalloc(&dc, cst.size() * 4) != hipSuccess) {
                    rn_set_error("hipMalloc(stage6x constants) failed");
                    return RN_E_NOMEM;
                }
                h->allocs.push_back(dc);
                RN_HIP(hipMemcpy(dc, cst.data(), cst.size() * 4, hipMemcpyHostToDevice));
                fs->s6_cstart = static_cast<float*>(dc);
            }
        }
        if (f.ptab && want_s4x) {
            std::vector<unsigned short> f16;
            rn_stage4x_pack(wsrc, h->dtype, f32_to_bf16, f32_to_f16, &f16);
            void* d16 = nullptr;
            if (hipMalloc(&d16, f16.size() * 2) != hipSuccess) {
                rn_set_error("hipMalloc(stage4x weights) failed");
                return RN_E_NOMEM;
            }
            h->allocs.push_back(d16);
            RN_HIP(hipMemcpy(d16, f16.data(), f16.size() * 2, hipMemcpyHostToDevice));
            f.wfrag16 = static_cast<i32x4*>(d16);
            f.use_s4x = true;
        }
        if (f.ptab && want_s5x) {
            // (the skip tensor must be the stage's own input: the kernel interpolates it from its input ring)
            std::vector<unsigned short> f16;
            rn_stage5x_pack(wsrc, h->dtype, f32_to_bf16, f32_to_f16, &f16);
            void* d16 = nullptr;
            if (hipMalloc(&d16, f16.size() * 2) != hipSuccess) {
                rn_set_error("hipMalloc(stage5x weights) failed");
                return RN_E_NOMEM;
            }
            h->allocs.push_back(d16);
            RN_HIP(hipMemcpy(d16, f16.data(), f16.size() * 2, hipMemcpyHostToDevice));
            f.wfrag16 = static_cast<i32x4*>(d16);
            f.use_s5x = true;
            if (fs->const4 && static_cast<int>(i) == fs->fold5_stage) {
                // without the 16 constant input channels (positions 48..63): 15 fragments per cout quarter instead of 18, and what
                // those channels add to every conv output -- sum over the nine taps of (16-bit weight / 6) x (stored 16-bit value),
                // the products the matrix cores would form, summed here in double
                rn_stage5x_pack48(wsrc, h->dtype, f32_to_bf16, f32_to_f16, &f16);
                void* d48 = nullptr;
                if (hipMalloc(&d48, f16.size() * 2) != hipSuccess) {
                    rn_set_error("hipMalloc(stage5x weights) failed");
                    return RN_E_NOMEM;
                }
                h->allocs.push_back(d48);
                RN_HIP(hipMemcpy(d48, f16.data(), f16.size() * 2, hipMemcpyHostToDevice));
                fs->s5_wfrag48 = static_cast<i32x4*>(d48);
                const auto cv = [&](float v) { return h->dtype == RN_DTYPE_BF16 ? f32_to_bf16(v) : f32_to_f16(v); };
                const auto bk = [&](unsigned short u) { return h->dtype == RN_DTYPE_BF16 ? bf16_bits_to_f32(u) : f16_to_f32(u); };
                std::vector<float> cst(64);
                for (int co = 0; co < 64; ++co) {
                    double sum = 0.0;
                    for (int tap = 0; tap < 9; ++tap)
                        for (int p = 48; p < 64; ++p)
                            sum += static_cast<double>(bk(cv(wsrc[(static_cast<size_t>(tap) * 64 + p) * 64 + co]))) * static_cast<double>(bk(fs->const4_val[p - 48]));
                    cst[co] = static_cast<float>(sum);
                }
                void* dc = nullptr;
                if (hipMalloc(&dc, cst.size() * 4) != hipSuccess) {
                    rn_set_error("hipMalloc(stage5x constants) failed");
                    return RN_E_NOMEM;
                }
                h->allocs.push_back(dc);
                RN_HIP(hipMemcpy(dc, cst.data(), cst.size() * 4, hipMemcpyHostToDevice));
                fs->s5_cstart = static_cast<float*>(dc);
            }
        }
        if (f.use_rw && f.ptab && rn_conv16p_supported(s.cin, s.cout, s.pool_k, s.pool_s, s.skip_stage >= 0) &&
            !(h->flags & RN_FLAG_GENERIC_KERNELS)) {
            std::vector<unsigned short> f16;
            rn_conv16p_pack(wsrc, h->dtype, f32_to_bf16, f32_to_f16, &f16);
            void* d16 = nullptr;
            if (hipMalloc(&d16, f16.size() * 2) != hipSuccess) {
                rn_set_error("hipMalloc(conv16p weights) failed");
                return RN_E_NOMEM;
            }
            h->allocs.push_back(d16);
            RN_HIP(hipMemcpy(d16, f16.data(), f16.size() * 2, hipMemcpyHostToDevice));
            f.wfrag16 = static_cast<i32x4*>(d16);
            f.use_c16p = true;
        }
    }
    if (fs->const4 && !(fs->fold5_stage >= 1 && fs->st[fs->fold5_stage - 1].use_s4x && fs->st[fs->fold5_stage].use_s5x && fs->s5_wfrag48 && fs->s5_cstart))
        fs->const4 = false;          // (another kernel family runs one of the two stages: every channel is computed)
    if (!fs->const4) fs->s6_wfrag48 = nullptr, fs->s6_cstart = nullptr;
    if (fs->const4) {
        unsigned short both[32];
        std::memcpy(both, fs->const4_val, 32);
        std::memcpy(both + 16, fs->const5_val, 32);
        void* dv = nullptr;
        if (hipMalloc(&dv, sizeof both) != hipSuccess) {
            rn_set_error("hipMalloc(constant channel values) failed");
            return RN_E_NOMEM;
        }
        h->allocs.push_back(dv);
        RN_HIP(hipMemcpy(dv, both, sizeof both, hipMemcpyHostToDevice));
        fs->const_vals_dev = static_cast<unsigned short*>(dv);
    }
    // ---- cross-stage fusion: the last two steps of a depth-3 block (network.py:183-203 with block_depth = 3):
    // stage i (32->32, pool 4/1) feeds only stage i+1 (32->32, pool 4/1 + residual), whose skip tensor is stage i's
    // INPUT.  One kernel runs both; stage i's output never reaches HBM.
    if (!(h->flags & (RN_FLAG_STAGE_LAUNCHES | RN_FLAG_GENERIC_KERNELS)))
        for (size_t i = 2; i + 1 < h->stages.size(); ++i) {
            const StagePlan& s1 = h->stages[i];
            const StagePlan& s2 = h->stages[i + 1];
            auto is3232 = [](const StagePlan& s) { return s.cin == 32 && s.cout == 32 && s.pool_k == 4 && s.pool_s == 1; };
            if (!is3232(s1) || !is3232(s2) || s1.skip_stage >= 0 || s2.skip_stage != static_cast<int>(i) - 1) continue;
            if (!fs->st[i].use_rw || !fs->st[i + 1].use_rw || !rn_stage23_supported(s1.in_side)) continue;
            bool feeds_others = false;      // stage i's output must have no other consumer
            for (size_t k = i + 2; k < h->stages.size(); ++k) feeds_others |= h->stages[k].skip_stage == static_cast<int>(i);
            if (feeds_others) continue;
            std::vector<float> t1(4 * 32), t2(4 * 32), tab(5 * 32);
            RN_HIP(hipMemcpy(t1.data(), fs->st[i].ptab, t1.size() * 4, hipMemcpyDeviceToHost));
            RN_HIP(hipMemcpy(t2.data(), fs->st[i + 1].ptab, t2.size() * 4, hipMemcpyDeviceToHost));
            std::copy(t1.begin(), t1.begin() + 64, tab.begin());
            std::copy(t2.begin(), t2.begin() + 96, tab.begin() + 64);
            void* dt = nullptr;
            if (hipMalloc(&dt, tab.size() * 4) != hipSuccess) {
                rn_set_error("hipMalloc(fused pair tables) failed");
                return RN_E_NOMEM;
            }
            h->allocs.push_back(dt);
            RN_HIP(hipMemcpy(dt, tab.data(), tab.size() * 4, hipMemcpyHostToDevice));
            fs->pair_ptab = static_cast<float*>(dt);
            fs->pair_first = static_cast<int>(i);
            if (!(h->flags & RN_FLAG_PAIR_32X32)) {
                // ---- frozen channels of the pair's on-chip tensor B (the first stage's output).  The epilogue stores
                // to16(fma(H, sc, sh)) with H = a sum of 16 ReLU6 / 6 values in [0, 16]: where |sc| * 16 < 2^-25 |sh| the fma
                // returns sh EXACTLY in float32 for every H the convolution can produce -- the channel is the constant to16(sh)
                // whatever the image, in this kernel's arithmetic bit for bit (and to 1e-10 of an O(1) tensor in the reference's
                // float32, where the same product vanishes against the same addend).  The shipped checkpoint has 18 such channels
                // of 32 (its L2 regulariser drove their BN gamma to ~1e-20): with >= 16 of them the first conv computes half of
                // its couts.  perm[p] = the channel at B-ring position p; the positions (p & 7) >= 4 -- the second half of every
                // 8-cout group -- take frozen channels.
                // both convs' weights / 6 (pool 4/1 stages), with the handle's refined rounding when it is on: every use below -- the
                // two fragment packs and the frozen channels' constant -- reads THESE arrays
                std::vector<float> wq[2];
                for (int which = 0; which < 2; ++which) {
                    const float* raw = w->stages[i + which].kernel;       // [tap][cin][cout]
                    wq[which].assign(raw, raw + static_cast<size_t>(9) * 32 * 32);
                    for (float& v : wq[which]) v /= 6.0f;
                    if (fs->refine) diffuse_taps(wq[which].data(), 32, 32, h->dtype);
                }
                int perm[32];
                {
                    // Round 6: the criterion is the tensor's 16-BIT STORE (as for the constant quarter of the 64-channel block above): the
                    // channel is constant when the two ends of the pooled sum's range, H = 0 and H = 16, store the same 16-bit number
                    // (fma and the conversion are monotone in H) -- every channel whose fma returns its addend in float32 (round 5's
                    // criterion: 18 on the shipped checkpoint) and those whose scale is below half an ulp of the shift (26 in bf16,
                    // 25 in fp16).  With >= 24 of them the ring holds EIGHT channels (live ones at positions 0..3, 8..11: the lane
                    // groups 0, 1 of the producer's half), with >= 16 sixteen (positions (p & 7) < 4).
                    std::vector<int> frozen, live;
                    for (int c = 0; c < 32; ++c) {
                        const float sc = t1[c], sh = t1[32 + c];
                        const bool fz = std::isfinite(sh) && cv_rne(std::fmaf(0.0f, sc, sh)) == cv_rne(std::fmaf(16.0f, sc, sh));      // (the B ring's store: round to nearest even)
                        (fz ? frozen : live).push_back(c);
                    }
                    fs->pair_frozen = static_cast<int>(frozen.size());
                    const bool fold_pair = !(h->flags & RN_FLAG_COMPUTE_FROZEN);
                    fs->pair_producer_halves = (frozen.size() >= 16 && fold_pair) ? 1 : 2;
                    fs->pair_narrow = fs->pair_producer_halves == 1 ? (frozen.size() >= 24 ? 2 : 1) : 0;
                    const int n_fold = fs->pair_narrow == 2 ? 24 : 16;
                    const auto live_pos = [&](int p) { return fs->pair_narrow == 2 ? ((p & 7) < 4 && p < 16) : (p & 7) < 4; };
                    if (fs->pair_producer_halves == 1) {
                        while (static_cast<int>(frozen.size()) > n_fold) {             // the spare constant channels are computed like live ones
                            live.push_back(frozen.back());
                            frozen.pop_back();
                        }
                        std::sort(live.begin(), live.end());
                        size_t nl = 0, nf = 0;
                        for (int p = 0; p < 32; ++p) perm[p] = live_pos(p) ? live[nl++] : frozen[nf++];
                    } else {
                        for (int p = 0; p < 32; ++p) perm[p] = p;
                    }
                    std::vector<float> tabx(tab);
                    tabx.resize(6 * 32, 0.f);
                    for (int p = 0; p < 32; ++p) {
                        tabx[p] = t1[perm[p]];
                        tabx[32 + p] = t1[32 + perm[p]];
                    }
                    if (fs->pair_producer_halves == 1) {
                        // row 5: what the 16 frozen channels (B positions (p & 7) >= 4) add to every output of the second conv:
                        // sum over the nine taps of (16-bit weight / 6) x (the channel's stored 16-bit value) -- the products the
                        // matrix cores would form, summed here in double
                        const auto cv = [&](float v) { return h->dtype == RN_DTYPE_BF16 ? f32_to_bf16(v) : f32_to_f16(v); };
                        const auto bk = [&](unsigned short u) {
                            if (h->dtype != RN_DTYPE_BF16) return f16_to_f32(u);
                            const unsigned bits = static_cast<unsigned>(u) << 16;
                            float f;
                            std::memcpy(&f, &bits, 4);
                            return f;
                        };
                        const float* w3src = wq[1].data();                  // [tap][cin][cout], already / 6
                        for (int co = 0; co < 32; ++co) {
                            double sum = 0.0;
                            for (int p = 0; p < 32; ++p) {
                                if (live_pos(p)) continue;
                                const double val = bk(cv_rne(t1[32 + perm[p]]));        // (the channel's stored value: the kernels' own store)
                                for (int tap = 0; tap < 9; ++tap)
                                    sum += static_cast<double>(bk(cv(w3src[(static_cast<size_t>(tap) * 32 + perm[p]) * 32 + co]))) * val;
                            }
                            tabx[160 + co] = static_cast<float>(sum);
                        }
                    }
                    void* dx = nullptr;
                    if (hipMalloc(&dx, tabx.size() * 4) != hipSuccess) {
                        rn_set_error("hipMalloc(fused pair tables) failed");
                        return RN_E_NOMEM;
                    }
                    h->allocs.push_back(dx);
                    RN_HIP(hipMemcpy(dx, tabx.data(), tabx.size() * 4, hipMemcpyHostToDevice));
                    fs->pair_ptab_x = static_cast<float*>(dx);
                }
                for (int which = 0; which < 2; ++which) {
                    std::vector<unsigned short> f16;
                    // (both stages are pool 4/1 register-weights stages: `sixth` weights, like their fragments above)
                    std::vector<float> w6(static_cast<size_t>(9) * 32 * 32);
                    const float* wsrc6 = wq[which].data();                  // [tap][cin][cout], already / 6
                    for (int tap = 0; tap < 9; ++tap)
                        for (int ci = 0; ci < 32; ++ci)
                            for (int co = 0; co < 32; ++co)
                                w6[(static_cast<size_t>(tap) * 32 + ci) * 32 + co] =
                                    which == 0 ? wsrc6[(static_cast<size_t>(tap) * 32 + ci) * 32 + perm[co]]        // B's channels = the first conv's couts
                                               : wsrc6[(static_cast<size_t>(tap) * 32 + perm[ci]) * 32 + co];       // ... and the second conv's cins
                    if (which == 1 && fs->pair_producer_halves == 1) {
                        int ring_cin[16];
                        for (int r = 0; r < 16; ++r) ring_cin[r] = perm[8 * (r >> 2) + (r & 3)];       // (eight-channel ring: r < 8)
                        if (fs->pair_narrow == 2)
                            rn_stage23x_pack_narrow8(wq[1].data(), ring_cin, h->dtype, f32_to_bf16, f32_to_f16, &f16);
                        else
                            rn_stage23x_pack_narrow(wq[1].data(), ring_cin, h->dtype, f32_to_bf16, f32_to_f16, &f16);
                    } else
                        rn_stage23x_pack(w6.data(), h->dtype, f32_to_bf16, f32_to_f16, &f16);
                    void* d16 = nullptr;
                    if (hipMalloc(&d16, f16.size() * 2) != hipSuccess) {
                        rn_set_error("hipMalloc(fused pair weights) failed");
                        return RN_E_NOMEM;
                    }
                    h->allocs.push_back(d16);
                    RN_HIP(hipMemcpy(d16, f16.data(), f16.size() * 2, hipMemcpyHostToDevice));
                    (which ? fs->pair_wfrag_b : fs->pair_wfrag_a) = static_cast<i32x4*>(d16);
                }
                fs->pair_x16 = true;
            }
            break;
        }
    // stage 0 inside stage 1's kernel: the 8-channel register-weights variant computes stage 0 for its own ring columns
    // (network.py:226 feeding :227's first step)
    if (!(h->flags & (RN_FLAG_STAGE_LAUNCHES | RN_FLAG_GENERIC_KERNELS)) && h->stages.size() > 1) {
        const StagePlan& s1 = h->stages[1];
        bool feeds_others = false;
        for (size_t k = 2; k < h->stages.size(); ++k) feeds_others |= h->stages[k].skip_stage == 0;
        fs->fuse_s0 = fs->st[1].use_rw && fs->st[1].rw.variant == 0 && s1.skip_stage < 0 && !feeds_others;
    }
    fs->use_tail = !(h->flags & (RN_FLAG_STAGE_LAUNCHES | RN_FLAG_GENERIC_KERNELS)) && rn_tail_supported(h);
    {
        const size_t ns = h->stages.size();
        fs->use_backend = fs->use_tail && !(h->flags & RN_FLAG_PAIR_32X32) && ns >= 5 && rn_backend_supported(h) && fs->st[ns - 4].use_s6x &&
                          fs->st[ns - 3].use_c16p;
    }
    fs->launch_rep.resize(h->stages.size());
    for (size_t i = 0; i < h->stages.size(); ++i) fs->launch_rep[i] = static_cast<int>(i);
    if (fs->fuse_s0) fs->launch_rep[0] = 1;
    if (fs->use_tail) fs->launch_rep[h->stages.size() - 2] = static_cast<int>(h->stages.size()) - 1;
    if (fs->pair_first >= 0) fs->launch_rep[fs->pair_first] = fs->pair_first + 1;
    return RN_OK;
}

void rn_fused_frozen_info(const rn_handle* h, int info[4]) {
    const FusedState* fs = static_cast<const FusedState*>(h->fused);
    info[0] = info[1] = 0;
    info[2] = -1;
    info[3] = 4;
    if (!fs) return;
    info[0] = fs->pair_x16 && fs->pair_producer_halves == 1 ? (fs->pair_narrow == 2 ? 24 : 16) : 0;
    info[1] = fs->pair_frozen;
    info[2] = fs->fold5_stage;
    info[3] = fs->fold5_stage >= 0 ? fs->fold5_live_q : 4;
}

void rn_fused_const_info(const rn_handle* h, int info[4]) {
    const FusedState* fs = static_cast<const FusedState*>(h->fused);
    info[0] = -1;
    info[1] = info[2] = info[3] = 0;
    if (!fs || fs->fold5_stage < 1) return;
    info[1] = fs->const4_proven;
    if (!fs->const4) return;
    info[0] = fs->fold5_stage - 1;
    info[2] = 16;
    info[3] = 48;
}

namespace {
// channels c0 .. c0 + 15 of a [npix, 64] 16-bit tensor <- vals[0..15]
__global__ void fill_channels16_kernel(unsigned short* base, int64_t npix, int c0, const i32x4 v0, const i32x4 v1) {
    const int64_t p = static_cast<int64_t>(blockIdx.x) * blockDim.x + threadIdx.x;
    if (p >= npix) return;
    i32x4* dst = reinterpret_cast<i32x4*>(base + p * 64 + c0);
    dst[0] = v0;
    dst[1] = v1;
}
}  // namespace

// After alloc_buffers: the constant channels of the two tensors nobody computes (const4) are written once, for every image slot
// of the handle; the kernels never touch these positions again, rn_tap and the consumers read complete tensors.
int rn_fused_post_alloc(rn_handle* h) {
    FusedState* fs = static_cast<FusedState*>(h->fused);
    if (!fs || !fs->const4) return RN_OK;
    const int r = fs->fold5_stage;
    const StagePlan& s4 = h->stages[r - 1];
    const StagePlan& s5 = h->stages[r];
    struct Job { int node; int side; const unsigned short* vals; };
    const Job jobs[2] = {{s4.node_bn, s4.out_side, fs->const4_val}, {s5.node_bn2, s5.out_side, fs->const5_val}};
    for (const Job& j : jobs) {
        unsigned short* base = static_cast<unsigned short*>(h->nodes[j.node].ptr);
        if (!base) {
            rn_set_error("constant channels: node %d has no buffer", j.node);
            return RN_E_STATE;
        }
        i32x4 v[2];
        std::memcpy(v, j.vals, 32);
        const int64_t npix = static_cast<int64_t>(h->max_batch) * j.side * j.side;
        hipLaunchKernelGGL(fill_channels16_kernel, dim3(static_cast<unsigned>((npix + 255) / 256)), dim3(256), 0, h->stream, base, npix, 48, v[0], v[1]);
        RN_CHECK_LAUNCH();
    }
    RN_HIP(hipStreamSynchronize(h->stream));
    return RN_OK;
}

// channel relabelling of a tensor on this handle (position p of the stored tensor holds channel perm[p] of the reference's), or null
const int* rn_fused_node_perm(const rn_handle* h, int node_id) {
    const FusedState* fs = static_cast<const FusedState*>(h->fused);
    if (!fs) return nullptr;
    auto it = fs->node_perm.find(node_id);
    return it == fs->node_perm.end() ? nullptr : it->second.data();
}

// true when the stage's output tensor is never written to HBM on this handle (it lives in LDS inside a fused launch)
bool rn_fused_stage_elided(const rn_handle* h, int stage) {
    const FusedState* fs = static_cast<const FusedState*>(h->fused);
    if (!fs) return false;
    const int ns = static_cast<int>(h->stages.size());
    if (fs->last_backend && (stage == ns - 4 || stage == ns - 3)) return true;
    return (fs->fuse_s0 && stage == 0) || (fs->pair_first >= 0 && stage == fs->pair_first);
}

int rn_fused_launch_rep(const rn_handle* h, int stage) {
    const FusedState* fs = static_cast<const FusedState*>(h->fused);
    if (!fs || stage < 0 || stage >= static_cast<int>(fs->launch_rep.size())) return stage;
    const int ns = static_cast<int>(fs->launch_rep.size());
    if (fs->last_backend && stage >= ns - 4) return ns - 1;      // (the last forward pass ran stage 6 .. head as one launch)
    return fs->launch_rep[stage];
}

int rn_fused_forward(rn_handle* h, const uint8_t* d_bgr, const float* d_rgb, int n, float* d_probs,
                     int64_t* d_ids) {
    if (!d_bgr || d_rgb) {
        rn_set_error("16-bit handles take uint8 BGR input (the pre-processing is folded into stage 0's weights)");
        return RN_E_STATE;
    }
    FusedState* fs = static_cast<FusedState*>(h->fused);
    if (!fs) {
        rn_set_error("fused plan missing");
        return RN_E_STATE;
    }
    const int dti = h->dtype == RN_DTYPE_BF16 ? 0 : 1;
#ifdef RN_CLOCK
    rn_clock_begin();
#endif
    // stage 0 (a launch of its own unless stage 1's kernel computes it)
    if (fs->fuse_s0) {
        rn_record_event(h, 2);
    } else {
        const StagePlan& s = h->stages[0];
        Stage0Args a0{};
        a0.bgr = d_bgr;
        a0.wfrag = fs->s0_wfrag;
        a0.ptab = fs->s0_ptab;
        a0.out = static_cast<unsigned short*>(h->nodes[s.node_bn].ptr);
        a0.S = s.in_side;
        a0.So = s.out_side;
        const int tiles = (s.out_side + S0_TSTRIDE - 1) / S0_TSTRIDE;
        a0.npt = tiles >= 8 ? 8 : tiles;
        a0.n_colblocks = (tiles + a0.npt - 1) / a0.npt;
        // ~4 workgroups of 8 waves per CU across the launch, at least 8 output rows per band
        const int per_band = n * a0.n_colblocks;
        int bands = (1024 + per_band - 1) / per_band;
        const int max_bands = (s.out_side + 7) / 8;
        if (bands > max_bands) bands = max_bands;
        if (bands < 1) bands = 1;
        a0.rows_per_band = (s.out_side + bands - 1) / bands;
        a0.n_bands = (s.out_side + a0.rows_per_band - 1) / a0.rows_per_band;
        dim3 grid(a0.n_bands * a0.n_colblocks, n);
        if (dti == 0)
            hipLaunchKernelGGL(stage0_kernel<RN_DTYPE_BF16>, grid, dim3(64 * a0.npt), 0, h->stream, a0);
        else
            hipLaunchKernelGGL(stage0_kernel<RN_DTYPE_F16>, grid, dim3(64 * a0.npt), 0, h->stream, a0);
        RN_CHECK_LAUNCH();
        rn_record_event(h, 2);
    }
    for (size_t i = 1; i < h->stages.size(); ++i) {
        const StagePlan& s = h->stages[i];
        const FusedStage& f = fs->st[i];
        const StagePlan& prev = h->stages[i - 1];
        if (fs->use_backend && i + 4 == h->stages.size() && 2 * n >= h->n_cu) {
            // stage 6 .. head in one launch: reported under the last stage
            const size_t ns = h->stages.size();
            for (size_t k = i; k + 1 < ns; ++k) rn_record_event(h, 2 + static_cast<int>(k));
            HeadArgs head;
            rn_fill_head_args(h, &head);
            const bool k48_6 = fs->s6_cstart && static_cast<int>(ns) - 4 == fs->fold5_stage + 1;
            int rc = rn_backend_launch(h, k48_6 ? fs->s6_wfrag48 : fs->st[ns - 4].wfrag16, fs->st[ns - 4].ptab, k48_6 ? fs->s6_cstart : nullptr,
                                       fs->st[ns - 3].wfrag16, fs->st[ns - 3].ptab, fs->st[ns - 2].wfrag, fs->st[ns - 1].wfrag, head, n, d_probs, d_ids);
            if (rc != RN_OK) return rc;
            fs->last_backend = true;         // (only now: a failed launch must not report s6.bn / s7.bn as elided)
            rn_record_event(h, 2 + static_cast<int>(ns - 1));
            rn_record_event(h, 2 + static_cast<int>(ns));
#ifdef RN_CLOCK
            rn_clock_end(h->stream);
#endif
            return RN_OK;
        }
        fs->last_backend = false;
        if (fs->use_tail && i + 2 == h->stages.size()) {
            // the last two stages, the flatten, the dense head, softmax and argmax in one launch: reported under the last
            // stage, the head's own slot reads ~0
            rn_record_event(h, 2 + static_cast<int>(i));
            HeadArgs head;
            rn_fill_head_args(h, &head);
            int rc = rn_tail_launch(h, fs->st[i].wfrag, fs->st[i + 1].wfrag, head, n, d_probs, d_ids);
            if (rc != RN_OK) return rc;
            rn_record_event(h, 2 + static_cast<int>(i) + 1);
            rn_record_event(h, 2 + static_cast<int>(h->stages.size()));
#ifdef RN_CLOCK
            rn_clock_end(h->stream);
#endif
            return RN_OK;
        }
        if (static_cast<int>(i) == fs->pair_first) {
            // both stages of the pair in one launch; the event of the first stage is recorded in front of it, so
            // rn_timing reports the whole launch under the second stage
            rn_record_event(h, 2 + static_cast<int>(i));
            const StagePlan& s2 = h->stages[i + 1];
            Stage23Args fa{};
            fa.in = static_cast<const unsigned short*>(h->nodes[prev.node_bn2 >= 0 ? prev.node_bn2 : prev.node_bn].ptr);
            fa.out = static_cast<unsigned short*>(h->nodes[s2.node_bn2].ptr);
            fa.wfrag2 = fs->pair_x16 ? fs->pair_wfrag_a : f.wfrag;
            fa.wfrag3 = fs->pair_x16 ? fs->pair_wfrag_b : fs->st[i + 1].wfrag;
            fa.ptab = fs->pair_x16 ? fs->pair_ptab_x : fs->pair_ptab;
            fa.producer_halves = fs->pair_x16 ? fs->pair_producer_halves : 2;
            fa.dither = fs->dither_out[i + 1];
            fa.narrow_b = fa.producer_halves == 1 ? fs->pair_narrow : 0;
            fa.rlo = s2.rt.lo;
            fa.rhi = s2.rt.hi;
            fa.rlerp = s2.rt.lerp;
            fa.rscale = static_cast<float>(s2.skip_side) / static_cast<float>(s2.out_side);
            fa.W = s.in_side;
            fa.Wo = s2.out_side;
            if (!rn_stage23_plan(s.in_side, &fa.n_cblocks, fa.cb_x0, fa.cb_wo)) {
                rn_set_error("fused stage pair: no column-block plan for input side %d", s.in_side);
                return RN_E_STATE;
            }
            // one workgroup per CU: workgroup = image x column block x band of rows; bands only to fill the chip / even
            // out the rounds (a band costs its rows plus 11 steps of pipeline fill)
            const int n_cu = h->n_cu;
            const long per_band = static_cast<long>(n) * fa.n_cblocks;
            int bands = 1;
            long best_cost = -1;
            const int max_bands = (s2.out_side + 7) / 8;
            for (int b = 1; b <= 8 && b <= max_bands; ++b) {
                const long rounds = (per_band * b + n_cu - 1) / n_cu;
                const long cost = rounds * ((s2.out_side + b - 1) / b + 11);
                if (best_cost < 0 || cost < best_cost) {
                    best_cost = cost;
                    bands = b;
                }
            }
            if (per_band * bands < n_cu) {
                bands = static_cast<int>((n_cu + per_band - 1) / per_band);
                if (bands > max_bands) bands = max_bands;
            }
            fa.rows_per_band = (s2.out_side + bands - 1) / bands;
            fa.n_bands = (s2.out_side + fa.rows_per_band - 1) / fa.rows_per_band;
#ifdef RN_STAMPS
            static unsigned long long* stamp_host23 = nullptr;
            const size_t nwaves23 = static_cast<size_t>(fa.n_bands) * fa.n_cblocks * n * 8;
            if (!stamp_host23) (void)hipHostMalloc(reinterpret_cast<void**>(&stamp_host23), 16u << 20, 0);
            std::memset(stamp_host23, 0, nwaves23 * 96);
            fa.stamp_buf = stamp_host23;
#endif
#ifdef RN_CLOCK
            if (fs->pair_x16) fa.stamp_buf = rn_clock_region("stages 2+3", static_cast<size_t>(fa.n_bands) * fa.n_cblocks * n);
#endif
#ifdef RN_ROUND2_ARMS
            int rc = fs->pair_x16 ? rn_stage23x_launch(h->dtype, h->stream, fa, n) : rn_stage23_launch(h->dtype, h->stream, fa, n);
#else
            // (the round-2 pair kernel, rn_stage23.hip, is part of the test / A-B library only: rn_create refuses RN_FLAG_PAIR_32X32 here)
            int rc = rn_stage23x_launch(h->dtype, h->stream, fa, n);
#endif
            if (rc != RN_OK) return rc;
#ifdef RN_STAMPS
            (void)hipStreamSynchronize(h->stream);
            {
                double wtot[8] = {0}, wbar[8] = {0}, wsteps[8] = {0};
                for (size_t k = 0; k < nwaves23; ++k) {
                    wtot[k & 7] += stamp_host23[k * 4];
                    wbar[k & 7] += stamp_host23[k * 4 + 2];
                    wsteps[k & 7] += stamp_host23[k * 4 + 3];
                }
                {
                    double seg[2][6] = {{0}, {0}};
                    double st[2] = {0, 0};
                    const unsigned long long* base = stamp_host23 + nwaves23 * 4;
                    for (size_t k = 0; k < nwaves23; ++k) {
                        const int role = (k & 7) < 4 ? 0 : 1;
                        st[role] += stamp_host23[k * 4 + 3];
                        for (int j = 0; j < 6; ++j) seg[role][j] += base[k * 8 + j];
                    }
                    fprintf(stderr, "[stamps]   producer segments (chain0 epi0 chain1 epi1): %.0f %.0f %.0f %.0f\n", seg[0][0] / st[0], seg[0][1] / st[0],
                            seg[0][2] / st[0], seg[0][3] / st[0]);
                    fprintf(stderr, "[stamps]   consumer segments (bookkeeping+own-fetch chain0 skip-wait epi0 chain1+epi1): %.0f %.0f %.0f %.0f %.0f\n", seg[1][0] / st[1],
                            seg[1][2] / st[1], seg[1][3] / st[1], seg[1][4] / st[1], seg[1][5] / st[1]);
                }
                for (int w8 = 0; w8 < 8; ++w8)
                    if (wsteps[w8] > 0)
                        fprintf(stderr, "[stamps] fused stages %zu+%zu wave %d (%s): cycles/step %.0f, of which barrier wait %.0f\n", i, i + 1, w8,
                                w8 < 4 ? "producer" : "consumer", wtot[w8] / wsteps[w8], wbar[w8] / wsteps[w8]);
            }
#endif
            rn_record_event(h, 2 + static_cast<int>(i) + 1);
            ++i;
            continue;
        }
        StageArgs a{};
        a.in = static_cast<const unsigned short*>(h->nodes[prev.node_bn2 >= 0 ? prev.node_bn2 : prev.node_bn].ptr);
        a.out = static_cast<unsigned short*>(h->nodes[s.node_bn2 >= 0 ? s.node_bn2 : s.node_bn].ptr);
        a.wfrag = f.wfrag;
        a.bn_mean = s.bn.mean;
        a.bn_inv = s.bn.inv;
        a.bn_beta = s.bn.beta;
        if (s.skip_stage >= 0) {
            const StagePlan& sk = h->stages[s.skip_stage];
            // the skip source is the first BN output of the block (network.py:195-196)
            if (sk.node_bn2 >= 0) {
                rn_set_error("16-bit path: skip source with its own residual is not supported");
                return RN_E_INVALID;
            }
            a.skip = static_cast<const unsigned short*>(h->nodes[sk.node_bn].ptr);
            a.bn2_mean = s.bn2.mean;
            a.bn2_inv = s.bn2.inv;
            a.bn2_beta = s.bn2.beta;
            a.rlo = s.rt.lo;
            a.rhi = s.rt.hi;
            a.rlerp = s.rt.lerp;
            a.Ss = s.skip_side;
            a.rscale = static_cast<float>(s.skip_side) / static_cast<float>(s.out_side);
        }
        a.H = a.W = s.in_side;
        a.Ho = a.Wo = s.out_side;
        a.dither = fs->dither_out[i];
        a.plain_q = -1;
        // (both arms: the constant channels sit in the last quarter of the two relabelled tensors and keep the plain rounding)
        if (fs->const_layout && (static_cast<int>(i) == fs->relabel_stage || static_cast<int>(i) + 1 == fs->relabel_stage)) a.plain_q = 3;
        if (f.use_c16) {
            Conv16Args ca{};
            ca.in = a.in;
            ca.out = a.out;
            ca.wfrag = f.wfrag16;
            ca.ptab = f.ptab;
            ca.H = ca.W = s.in_side;
            ca.Ho = ca.Wo = s.out_side;
            ca.n_colblocks = rn_conv16_colblocks(s.out_side);
            // 4-wave workgroups, two per CU; back-filled, so the cost of a band count is the fractional number of rounds
            const long per_band = static_cast<long>(n) * ca.n_colblocks;
            const long slots = 2L * h->n_cu;
            const int max_bands = (s.out_side + 3) / 4;
            int bands = 1;
            double best_cost = -1;
            for (int b = 1; b <= 8 && b <= max_bands; ++b) {
                if (per_band * b < slots && b < max_bands) continue;          // fill the chip first
                const double cost = rn_backfill_cost(per_band * b, slots, (s.out_side + b - 1) / b, 2);
                if (best_cost < 0 || cost < best_cost) {
                    best_cost = cost;
                    bands = b;
                }
            }
            if (per_band * bands < slots) {
                bands = static_cast<int>((slots + per_band - 1) / per_band);
                if (bands > max_bands) bands = max_bands;
            }
            ca.rows_per_band = (s.out_side + bands - 1) / bands;
            ca.n_bands = (s.out_side + ca.rows_per_band - 1) / ca.rows_per_band;
            int rc = rn_conv16_launch(h->dtype, h->stream, ca, n);
            if (rc != RN_OK) return rc;
            rn_record_event(h, 2 + static_cast<int>(i));
            continue;
        }
        if (f.use_s5x || f.use_s4x || f.use_s6x) {
            a.wfrag = f.wfrag16;
            a.ptab = f.ptab;
            const bool planned = f.use_s5x   ? rn_stage5x_plan(s.out_side, &a.n_cb, a.cb_xo0, a.cb_wo)
                                 : f.use_s4x ? rn_stage4x_plan(s.out_side, &a.n_cb, a.cb_xo0, a.cb_wo)
                                             : rn_stage6x_plan(s.out_side, &a.n_cb, a.cb_xo0, a.cb_wo);
            if (!planned) {
                rn_set_error("stage %zu: no column-block plan for output side %d", i, s.out_side);
                return RN_E_STATE;
            }
            // one workgroup (8 waves) per CU, workgroup = image x column block x band of rows: whole rounds of the chip.  A
            // band costs its input rows plus the rows its neighbour reads again (6 of the pooled stages, 2 of the un-pooled
            // one); small batches take as many bands as it needs to fill the chip.
            const long per_band = static_cast<long>(n) * a.n_cb;
            const int rows_in = s.pool_k ? 2 : 1, overlap = s.pool_k ? 6 : 2;
            const int max_bands = std::max(1, s.out_side / 4);
            int bands = 1;
            long best_cost = -1;
            for (int b = 1; b <= 8 && b <= max_bands; ++b) {
                const long rounds = (per_band * b + h->n_cu - 1) / h->n_cu;
                const long cost = rounds * (rows_in * ((s.out_side + b - 1) / b) + overlap);
                if (best_cost < 0 || cost < best_cost) {
                    best_cost = cost;
                    bands = b;
                }
            }
            if (per_band * bands < h->n_cu) bands = static_cast<int>(std::min<long>((h->n_cu + per_band - 1) / per_band, max_bands));
            a.rows_per_band = (s.out_side + bands - 1) / bands;
            a.n_bands = (s.out_side + a.rows_per_band - 1) / a.rows_per_band;
#ifdef RN_CLOCK
            {
                char what[32];
                snprintf(what, sizeof what, "stage %d", static_cast<int>(i));
                a.stamp_buf = rn_clock_region(what, static_cast<size_t>(a.n_bands) * a.n_cb * n);
            }
#endif
            a.live_q = (f.use_s5x && static_cast<int>(i) == fs->fold5_stage) ? fs->fold5_live_q : 4;
            if (fs->const4 && f.use_s4x && static_cast<int>(i) + 1 == fs->fold5_stage) {      // its last cout quarter is constant
                a.live_q = 3;
                a.cvals = fs->const_vals_dev;
            }
            if (fs->s6_cstart && f.use_s6x && static_cast<int>(i) == fs->fold5_stage + 1) {
                a.wfrag = fs->s6_wfrag48;
                a.cstart = fs->s6_cstart;
            }
            if (fs->const4 && f.use_s5x && static_cast<int>(i) == fs->fold5_stage) {
                a.wfrag = fs->s5_wfrag48;
                a.cstart = fs->s5_cstart;
                a.cvals = fs->const_vals_dev + 16;
            }
            int rc = f.use_s5x   ? rn_stage5x_launch(h->dtype, h->stream, a, n)
                     : f.use_s4x ? rn_stage4x_launch(h->dtype, h->stream, a, n)
                                 : rn_stage6x_launch(h->dtype, h->stream, a, n);
            if (rc != RN_OK) return rc;
            rn_record_event(h, 2 + static_cast<int>(i));
            continue;
        }
        if (f.use_c16p) {
            Conv16Args ca{};
            ca.in = a.in;
            ca.out = a.out;
            ca.wfrag = f.wfrag16;
            ca.ptab = f.ptab;
            ca.H = ca.W = s.in_side;
            ca.Ho = ca.Wo = s.out_side;
            ca.n_colblocks = rn_conv16p_colblocks(s.out_side);
            const long per_band = static_cast<long>(n) * ca.n_colblocks;
            const int per_cu = rn_conv16p_wgs_per_cu(s.out_side);    // 3-wave workgroups (72 KB of LDS): two per CU; 5-wave ones (123 KB): one
            const long slots = static_cast<long>(per_cu) * h->n_cu;
            const int max_bands = (s.out_side + 3) / 4;
            int bands = 1;
            double best_cost = -1;
            for (int b = 1; b <= 8 && b <= max_bands; ++b) {
                if (per_band * b < slots && b < max_bands) continue;          // fill the chip first
                const double cost = rn_backfill_cost(per_band * b, slots, 2 * ((s.out_side + b - 1) / b) + 2, per_cu);
                if (best_cost < 0 || cost < best_cost) {
                    best_cost = cost;
                    bands = b;
                }
            }
            if (per_band * bands < slots) {
                bands = static_cast<int>((slots + per_band - 1) / per_band);
                if (bands > max_bands) bands = max_bands;
            }
            ca.rows_per_band = (s.out_side + bands - 1) / bands;
            ca.n_bands = (s.out_side + ca.rows_per_band - 1) / ca.rows_per_band;
            int rc = rn_conv16p_launch(h->dtype, h->stream, ca, n);
            if (rc != RN_OK) return rc;
            rn_record_event(h, 2 + static_cast<int>(i));
            continue;
        }
        if (f.use_rw) {
            // `sixth` weights (/ 6, BN scale x 6) are only right for kernels that clamp to [0, 1]: the pool 4/1 variants of the
            // register-weights kernel and rn_stage4x / 5x (handled above).  Its stride-2 (DPP) variants clamp at 6.
            if (f.sixth && !(s.pool_k == 4 && s.pool_s == 1)) {
                rn_set_error("16-bit path: stage %zu has weights / 6 but would run a kernel that applies ReLU6 at 6", i);
                return RN_E_STATE;
            }
#ifdef RN_DIAG
            if (const char* dbg = getenv("RN_DEBUG_FLAGS")) a.dbg_flags = atoi(dbg);   // diagnostic builds only
#endif
            if (i == 1 && fs->fuse_s0) {
                a.s0_bgr = d_bgr;
                a.s0_wfrag = fs->s0_wfrag;
                a.s0_ptab = fs->s0_ptab;
                a.s0_S = h->stages[0].in_side;
                a.s0_private = (h->flags & RN_FLAG_PAIR_32X32) ? 1 : 0;
            }
            a.ptab = f.ptab;
            a.skipcols = f.rw.skipcols;
            a.n_colblocks = f.rw.n_colblocks;
            // the shared-ring form of the fused stages 0 + 1 cuts rows into 227-column blocks (eight 29-column tiles minus the
            // 5 halo columns of the last one)
            if (a.s0_bgr && !a.s0_private && f.rw.variant == 0 && f.rw.npt == 8) a.n_colblocks = rn_rw_s0sh_colblocks(s.out_side);
            a.npt = f.rw.npt;
            a.n_ctg = 1;
            // Workgroups per CU: one (8-wave variants; checked with HW_ID stamps) or four (the 2-wave workgroups of the
            // 32->64 stage).  A band costs its rows plus ~10 rows of prologue / pool warm-up; pick the band count that
            // gives the least time.  One workgroup per CU runs in whole rounds of the chip: 1 band at batch 256 x 224^2,
            // 2 when e.g. 64 x 600^2 images x 6 column blocks = 384 workgroups would otherwise run 1.5 rounds.  Small
            // workgroups are back-filled as slots free up, so their cost is the fractional number of rounds (>= 1).
            const int per_band = n * a.n_colblocks;
            const long slots = static_cast<long>(h->n_cu) * f.rw.wgs_per_cu;
            const int max_bands = (s.out_side + 7) / 8;
            int bands = 1;
            double best_cost = -1;
            for (int b = 1; b <= 8 && b <= max_bands; ++b) {
                const long wgs = static_cast<long>(per_band) * b;
                const long rows_b = (s.out_side + b - 1) / b * (s.pool_k ? s.pool_s : 1);     // conv rows of a band
                const double cost = f.rw.wgs_per_cu == 1 ? static_cast<double>((wgs + slots - 1) / slots) * static_cast<double>(rows_b + 10)
                                                         : rn_backfill_cost(wgs, slots, static_cast<int>(rows_b) + 2, f.rw.wgs_per_cu);
                if (best_cost < 0 || cost < best_cost) {
                    best_cost = cost;
                    bands = b;
                }
            }
            if (static_cast<long>(per_band) * bands < slots) {      // small batches: fill the chip first
                bands = static_cast<int>((slots + per_band - 1) / per_band);
                if (bands > max_bands) bands = max_bands;
            }
            a.rows_per_band = (s.out_side + bands - 1) / bands;
            if (s.pool_k == 0 && a.rows_per_band < 4) a.rows_per_band = 4;   // ring prologue depth
            a.n_bands = (s.out_side + a.rows_per_band - 1) / a.rows_per_band;
            dim3 grid(a.n_bands * a.n_colblocks, n);
#ifdef RN_STAMPS
            // diagnostic build: per-wave phase cycle sums, printed per stage after the launch
            static unsigned long long* stamp_host = nullptr;
            const size_t nwaves = static_cast<size_t>(grid.x) * grid.y * 16;
            if (!stamp_host) (void)hipHostMalloc(reinterpret_cast<void**>(&stamp_host), 64u << 20, 0);
            std::memset(stamp_host, 0, nwaves * 32);
            a.stamp_buf = stamp_host;
#endif
#ifdef RN_CLOCK
            {
                char what[32];
                snprintf(what, sizeof what, a.s0_bgr ? "stages 0+%zu" : "stage %zu", i);
                a.stamp_buf = rn_clock_region(what, static_cast<size_t>(grid.x) * grid.y);
            }
#endif
            int rc = rn_rw_launch(f.rw, h->dtype, h->stream, a, grid);
            if (rc != RN_OK) return rc;
#ifdef RN_STAMPS
            (void)hipStreamSynchronize(h->stream);
            {
                double w = 0, d = 0, b = 0, rows = 0, ch = 0, life = 0, pro = 0;
                size_t cnt = 0;
                for (size_t k = 0; k < nwaves; ++k)
                    if (stamp_host[k * 4 + 3]) {
                        w += stamp_host[k * 4];
                        d += static_cast<double>(stamp_host[k * 4 + 1] & 0xffffffffull);
                        life += static_cast<double>(stamp_host[k * 4 + 1] >> 32);
                        pro += static_cast<double>(stamp_host[k * 4 + 3] >> 32);
                        b += static_cast<double>(stamp_host[k * 4 + 2] & 0xffffffffull);
                        ch += static_cast<double>(stamp_host[k * 4 + 2] >> 32);
                        rows += static_cast<double>(stamp_host[k * 4 + 3] & 0xffffffffull);
                        ++cnt;
                    }
#ifdef RN_STAMP_HWID
                {
                    // how many workgroups of this launch were resident on one CU at the same time?
                    struct Iv { unsigned long long t0, t1; unsigned cu; };
                    std::vector<Iv> iv;
                    for (size_t k = 0; k < nwaves; ++k)
                        if (stamp_host[k * 4 + 3] & 0xffffffffull) {
                            const unsigned hw = static_cast<unsigned>(stamp_host[k * 4 + 3] >> 32);
                            const unsigned xcc = static_cast<unsigned>(stamp_host[k * 4 + 1]) & 0xf;
                            // gfx9 HW_ID: wave[3:0] simd[5:4] pipe[7:6] cu[11:8] sh[12] se[15:13]
                            const unsigned cu = ((hw >> 8) & 0xf) | (((hw >> 12) & 1) << 4) | (((hw >> 13) & 7) << 5) | (xcc << 8);
                            iv.push_back({stamp_host[k * 4 + 0], stamp_host[k * 4 + 2], cu});
                        }
                    std::map<unsigned, std::vector<Iv>> per;
                    for (auto& v : iv) per[v.cu].push_back(v);
                    double avg_conc = 0; size_t ncu = 0; size_t maxc = 0;
                    for (auto& kv : per) {
                        // time-weighted mean number of resident waves on this CU
                        std::vector<std::pair<unsigned long long, int>> ev;
                        for (auto& v : kv.second) { ev.push_back({v.t0, 1}); ev.push_back({v.t1, -1}); }
                        std::sort(ev.begin(), ev.end());
                        double area = 0; int cur = 0; unsigned long long last = ev.front().first; size_t mx = 0;
                        for (auto& e : ev) { area += static_cast<double>(e.first - last) * cur; last = e.first; cur += e.second; if (static_cast<size_t>(cur) > mx) mx = cur; }
                        avg_conc += area / static_cast<double>(ev.back().first - ev.front().first);
                        maxc = std::max(maxc, mx); ++ncu;
                    }
                    fprintf(stderr, "[hwid] stage %zu: %zu waves on %zu distinct CUs; resident waves per CU: mean %.2f, max %zu\n", i, iv.size(), ncu,
                            avg_conc / ncu, maxc);
                }
#endif
                if (cnt && getenv("RN_STAMPS_PER_WAVE")) {
                    // per wave index inside the workgroup: who is the straggler the others wait for at the barrier?
                    const size_t wpw = 16;      // stamp slots per workgroup (rn_stage_rw.hip writes NTHREADS / 64 of them)
                    double ww[16] = {0}, bb[16] = {0}, rr[16] = {0};
                    const size_t real = static_cast<size_t>(getenv("RN_STAMPS_WAVES") ? atoi(getenv("RN_STAMPS_WAVES")) : 8);
                    for (size_t k = 0; k < nwaves; ++k)
                        if (stamp_host[k * 4 + 3]) {
                            const size_t wi = k % real;
                            ww[wi] += stamp_host[k * 4];
                            bb[wi] += static_cast<double>(stamp_host[k * 4 + 2] & 0xffffffffull);
                            rr[wi] += static_cast<double>(stamp_host[k * 4 + 3] & 0xffffffffull);
                        }
                    (void)wpw;
                    for (size_t wi = 0; wi < real; ++wi)
                        if (rr[wi] > 0) fprintf(stderr, "[stamps]   stage %zu wave %zu: work %.0f barrier %.0f\n", i, wi, ww[wi] / rr[wi], bb[wi] / rr[wi]);
                }
                if (cnt)
                    fprintf(stderr, "[stamps] stage %zu: waves %zu, cycles/step: work %.0f (MFMA chain alone %.0f)  dma-wait %.0f  barrier %.0f  (steps/wave %.0f)\n",
                            i, cnt, w / rows, ch / rows, d / rows, b / rows, rows / cnt);
                if (cnt)
                    fprintf(stderr, "[stamps]   stage %zu: wave lifetime %.0f cycles, of which prologue %.0f, row loop %.0f\n", i, life / cnt,
                            pro / cnt, (w + d + b) / cnt);
            }
#endif
            rn_record_event(h, 2 + static_cast<int>(i));
            continue;
        }
        const int n_ctg = ((s.cout + 31) / 32) / f.ctw;
        // bands: aim for >= ~2 workgroups per CU across the launch, at least 4 output rows per band
        const int per_band_wgs = n * f.n_colblocks * n_ctg;
        int bands = (768 + per_band_wgs - 1) / per_band_wgs;
        // tiny stages are pure latency chains (one wave per workgroup, a global-load round trip per row): give
        // every output row its own workgroup instead of 4 rows each
        const int max_bands = s.out_side <= 8 ? s.out_side : (s.out_side + 3) / 4;
        if (bands > max_bands) bands = max_bands;
        if (s.out_side <= 8) bands = max_bands;
        if (bands < 1) bands = 1;
        a.rows_per_band = (s.out_side + bands - 1) / bands;
        a.n_bands = (s.out_side + a.rows_per_band - 1) / a.rows_per_band;
        a.n_colblocks = f.n_colblocks;
        a.n_ctg = n_ctg;
        a.npt = f.npt;
        dim3 grid(a.n_bands * a.n_colblocks * a.n_ctg, n);
        int rc = kVariants[f.variant].fn[dti](h->stream, a, grid, dim3(64 * f.npt), f.lds_bytes);
        if (rc != RN_OK) return rc;
        rn_record_event(h, 2 + static_cast<int>(i));
    }
    int rc = rn_run_head(h, n, d_probs, d_ids);
    if (rc != RN_OK) return rc;
    rn_record_event(h, 2 + static_cast<int>(h->stages.size()));
#ifdef RN_CLOCK
    rn_clock_end(h->stream);
#endif
    return RN_OK;
}
