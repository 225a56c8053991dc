// 64 -> 64 residual stage with avg-pool 4/2 (reference network.py:228, second step of conv_block(64, pool 4/2, depth 2)) on
// 16x16x32 matrix tiles with ROW-REGISTER BLOCKING:
//
//   A = s4.bn [N, W, W, 64] -> conv3x3 VALID -> ReLU6 -> avg-pool 4x4 stride 2 -> BN -> + legacy-bilinear(A) -> BN  = out [N, Wo, Wo, 64]
//
// Why a kernel of its own: the row-streaming template (rn_stage_rw.hip) runs this stage with ONE wave per SIMD (144 weight
// registers per 32-cout tile), 32-pixel tiles of which the fourth is mostly empty (98 conv columns) and a staged copy of
// the skip rows; its matrix pipe is 57 % busy.  Here:
//  * a wave owns 16 couts (72 weight registers: 9 taps x 2 channel halves x 4) for 3 or 4 ADJACENT 16-pixel tiles; eight
//    waves = 4 cout quarters x 2 pixel halves, two per SIMD; 7 tiles cover the 98 conv columns (32-pixel form: 128);
//  * every operand fragment (16 pixels x 32 channels of the NEWEST input row, one ds_read_b128) feeds THREE accumulators --
//    kernel row 0 of conv row s, row 1 of conv row s-1, row 2 of conv row s-2 -- so a step reads 6 fragments per tile for
//    18 MFMAs: the four cout-quarter waves re-reading every fragment stay far below the LDS bandwidth, and the three chains
//    are independent (no dependent-issue stalls).  The partial accumulators of all tiles live across steps (48 registers);
//  * pooling (stride 2) is the band-matrix MFMA of rn_stage23x.hip: vertical pair sums as fp16 pairs ARE the A operand, one
//    MFMA accumulates the 16 pooled columns of a tile PAIR from the pair's two tiles and the first two columns of the next;
//  * the residual's skip tensor is the stage's own input: its two source rows are still in the input ring (9 slots), read
//    transposed (ds_read_b64_tr_b16) and interpolated horizontally on the matrix cores (K = 64 source columns per tile pair);
//    the lerp fraction is quantised like in the fused stage pair (res_quant_lerp) so that its two weights are one exact
//    16-bit operand.
// K48 (round 6): the input channels 48..63 are CONSTANTS on the handle (StageArgs::cstart; rn_fused_prepare): a row step reads
// five fragments per tile instead of six -- (kx, channels 0..31) for kx = 0, 1, 2, then [kx 0 | kx 1] and [kx 2 | zero weights] of
// channels 32..47 (lane groups 0, 1 take one tap's two chunks, groups 2, 3 the next tap's) -- for 15 MFMAs instead of 18; the
// accumulators start from the constants' contribution, and the waves of the last cout quarter, whose output channels are
// constants as well (frozen first BN + constant skip channel), idle.
// One workgroup = one image x one band of output rows x one COLUMN BLOCK of 31..53 pooled columns (66..110 input columns:
// the whole row of the 224 x 224 network, three blocks at 600; rn_stage5x_plan checks that every block's skip columns lie
// inside the input columns it stages).
#include "rn_fused.h"
#include "rn_stage.h"

#include <atomic>
#include <utility>

using namespace rnk;

namespace {

constexpr int V_NS = 9;                           // ring slots: rows s-5 .. s (conv window + residual rows) + 3 in flight
constexpr int V_AHEAD = 3;
constexpr int V_RINGPX = 112;                     // pixels per ring row (7 tiles + halo; input rows are <= 110 wide)
constexpr int V_ROW = V_RINGPX * 128;             // bytes per ring row (64 channels x 16 bit per pixel)
constexpr int V_TAB_BYTES = 4 * 64 * 4;           // folded BN tables [4][64] (rn_fused_prepare: sc1', sh1', sc2, sh2)
constexpr int V_RING_OFF = V_TAB_BYTES;
constexpr int V_LDS = V_RING_OFF + V_NS * V_ROW;
constexpr int V_WMIN = 66, V_WMAX = 110;
static_assert(V_LDS <= 160 * 1024, "LDS budget");

__device__ __forceinline__ unsigned lds_addr(const void* p) {
    return static_cast<unsigned>(reinterpret_cast<uintptr_t>((__attribute__((address_space(3))) const char*)p));
}
// 8 chunks per 128-byte pixel: chunk ^ (pixel & 7) is conflict-free for the 16x16x32 operand read (rn_conv16.hip)
__device__ __forceinline__ int swz8(int pix) { return pix & 7; }

using i32x2 = __attribute__((ext_vector_type(2))) int;

template <int DT>
__device__ __forceinline__ f32x4 mfma16(i32x4 a, i32x4 b, f32x4 c) {
    if constexpr (DT == RN_DTYPE_BF16)
        return __builtin_amdgcn_mfma_f32_16x16x32_bf16(__builtin_bit_cast(bf16x8, a), __builtin_bit_cast(bf16x8, b), c, 0, 0, 0);
    else
        return __builtin_amdgcn_mfma_f32_16x16x32_f16(__builtin_bit_cast(f16x8, a), __builtin_bit_cast(f16x8, b), c, 0, 0, 0);
}

template <int DT, bool K48>
__global__ __launch_bounds__(512, 2) void stage5x_kernel(const StageArgs a) {
    constexpr int NF = K48 ? 5 : 6;                   // operand fragments per tile and row step
#ifdef RN_CLOCK
    unsigned long long ck_t0, ck_r0;
    clock_pair(ck_t0, ck_r0);
#endif
    extern __shared__ __attribute__((aligned(64))) char smem[];
    const int tid = threadIdx.x;
    const int lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    // cout quarter / pixel half.  All quarters live: waves w and w + 4 (one SIMD) are the two pixel halves of quarter w (4 + 3
    // tiles).  Two live quarters (StageArgs::live_q): waves 0..3 = the live (quarter, half) units, one per SIMD, waves 4..7 =
    // the frozen quarters' units, whose step is the residual epilogue only.
    const bool fold = a.live_q == 2;
    const int cq = fold ? (wave < 4 ? (wave >> 1) : 2 + ((wave - 4) >> 1)) : (wave & 3);
    const int ph = fold ? (wave & 1) : (wave >> 2);
    const bool conv_live = !fold || cq < 2;
    const bool epi_live = !K48 || cq < 3;                 // (K48: the last quarter's output channels are constants, written at rn_create)
    const int px16 = lane & 15, g = lane >> 4;
    const int cb = blockIdx.x % a.n_cb, band = blockIdx.x / a.n_cb, n = blockIdx.y;
    const int Win = a.W, Wo_full = a.Wo, Ho = a.Ho;
    const int xo0 = a.cb_xo0[cb], Wo = a.cb_wo[cb];       // this block's pooled columns
    const int x0 = 2 * xo0;                               // its first input column
    const int W = min(2 * Wo + 6, Win - x0);              // input columns it stages (conv window + the residual's reach)
    const int yo0 = band * a.rows_per_band;
    const int nrows = min(Ho, yo0 + a.rows_per_band) - yo0;
    const int y0 = 2 * yo0;                               // first input row = first conv row of the band
    const int nconv = 2 * (nrows - 1) + 4;
    const int nin = nconv + 2;
    const f32x4 zero4 = {0.f, 0.f, 0.f, 0.f};
    constexpr int OOB = 0x40000000;

    float* const tab = reinterpret_cast<float*>(smem);
    if (tid < 256) tab[tid] = a.ptab[tid];
    char* const ring = smem + V_RING_OFF;
    const unsigned ring_lds = lds_addr(ring);
    // pixels W .. 111 of every slot are never written by the row DMA: zero them once (the conv fragments of the discarded
    // right-hand columns read them; uninitialised LDS could hold NaN patterns)
    for (int i = tid; i < V_NS * (V_RINGPX - V_WMIN) * 8; i += 512) {
        const int slot = i / ((V_RINGPX - V_WMIN) * 8), rest = i % ((V_RINGPX - V_WMIN) * 8);
        const int p = V_WMIN + rest / 8, c = rest % 8;
        if (p >= W) *reinterpret_cast<i32x4*>(ring + slot * V_ROW + p * 128 + c * 16) = i32x4{0, 0, 0, 0};
    }

    // ---- this wave's tiles: pixel half 0 = 4 tiles from column 0 (31 pooled columns), half 1 = 3 tiles from column 62
    // (23 pooled columns from 31): the runs overlap by the 2 columns a stride-2 window reaches across
    const bool has4 = ph == 0;
    const int xw = ph ? 62 : 0;
    const int xo_run = ph ? 31 : 0;
    const int nout_run = has4 ? 31 : 23;

    // ---- weights: fragment f = (ky * 3 + kx) * 2 + ch (K48: ky * 5 + j, rn_stage5x_pack48), B operand of D'[pixel][cout]
    i32x4 wf[3 * NF];
#pragma unroll
    for (int f = 0; f < 3 * NF; ++f) {
        const i32x4* src = a.wfrag + (f * 4 + cq) * 64 + lane;      // (frozen quarters: loaded, never used)
        asm volatile("global_load_dwordx4 %0, %1, off" : "=v"(wf[f]) : "v"(src) : "memory");
    }

    // ---- input rows by LDS-DMA: two pieces per wave and row (piece 0: chunks tid = pixels 0..63; piece 1: the remaining
    // (W - 64) x 8 chunks, W - 64 per wave, lane-masked)
    const char* const in_img = reinterpret_cast<const char*>(a.in + static_cast<int64_t>(n) * Win * Win * 64) + x0 * 128;
    const int row_bytes = Win * 128;
    const unsigned goff0 = static_cast<unsigned>((tid >> 3) * 128 + (((tid & 7) ^ swz8(tid >> 3)) << 4));
    const int tailn = W - 64;
    const unsigned long long tail_mask = tailn >= 64 ? ~0ull : ((1ull << tailn) - 1ull);
    unsigned goff1;
    {
        const int q = 512 + wave * tailn + min(lane, tailn - 1);
        const int p = q >> 3, c = q & 7;
        goff1 = static_cast<unsigned>(p * 128 + ((c ^ swz8(p)) << 4));
    }
    auto issue_row = [&](int y, int slot) __attribute__((always_inline)) {
        const char* row = in_img + static_cast<int64_t>(y0 + min(y, nin - 1)) * row_bytes;
        unsigned o0 = goff0, o1 = goff1;
        asm volatile("" : "+v"(o0), "+v"(o1));
        dma16(row + o0, ring + slot * V_ROW + wave * 1024);
        dma16_masked(row + o1, ring + slot * V_ROW + (512 + wave * tailn) * 16, tail_mask);
    };

    // ---- operand read bases (slot 0): tap column kx, channel half ch; tile k adds 16 pixels = 2048 bytes (same swizzle)
    unsigned base[3][2];
#pragma unroll
    for (int kx = 0; kx < 3; ++kx)
#pragma unroll
        for (int ch = 0; ch < 2; ++ch) {
            const int p = xw + px16 + kx;
            base[kx][ch] = ring_lds + static_cast<unsigned>(p * 128 + (((4 * ch + g) ^ swz8(p)) << 4));
        }
    if constexpr (K48) {
        // fragments 3, 4: channels 32..47 of two taps -- lane groups 0, 1 read chunks 4, 5 of tap column 0 (fragment 4: column 2),
        // groups 2, 3 those of column 1 (fragment 4: zero weights; they read what groups 0, 1 read)
#pragma unroll
        for (int j = 0; j < 2; ++j) {
            const int p = xw + px16 + (j == 0 ? (g >> 1) : 2);
            base[j][1] = ring_lds + static_cast<unsigned>(p * 128 + (((4 + (g & 1)) ^ swz8(p)) << 4));
        }
    }

    // ---- pooling band matrices (stride 2).  The 16 pooled columns n of a tile PAIR: n < 8 start in the pair's first tile
    // (window = its columns 2n .. 2n+3; n = 7 ends in the second tile), n >= 8 in the second (n = 15 ends in the tile after).
    // K element 8 g + e of an operand = pixel 4 g + (e & 3) of the tile (older pair sum for e < 4, newer for e >= 4).
    i32x4 pmA, pmB, pmC;
#pragma unroll
    for (int d = 0; d < 4; ++d) {
        unsigned wa = 0, wb = 0, wc = 0;
#pragma unroll
        for (int e2 = 0; e2 < 2; ++e2) {
            const int e = 2 * d + e2;
            const int p = 4 * g + (e & 3);
            const int nn = px16;
            const bool inA = nn < 8 && p >= 2 * nn && p <= 2 * nn + 3;
            const bool inB = (nn < 8 && 16 + p >= 2 * nn && 16 + p <= 2 * nn + 3) || (nn >= 8 && p >= 2 * (nn - 8) && p <= 2 * (nn - 8) + 3);
            const bool inC = nn >= 8 && 16 + p >= 2 * (nn - 8) && 16 + p <= 2 * (nn - 8) + 3;
            wa |= (inA ? 0x3C00u : 0u) << (16 * e2);
            wb |= (inB ? 0x3C00u : 0u) << (16 * e2);
            wc |= (inC ? 0x3C00u : 0u) << (16 * e2);
        }
        pmA[d] = static_cast<int>(wa);
        pmB[d] = static_cast<int>(wb);
        pmC[d] = static_cast<int>(wc);
    }
    asm volatile("" : "+v"(pmA), "+v"(pmB), "+v"(pmC));

    // ---- residual: tile pair u interpolates its 16 pooled columns from 64 source columns starting at xs_u (kept inside the
    // row); two K halves.  Transposed reads: lane 4 q + p' of a 16-lane group supplies pixel row q, couts 16 cq + 4 p' .. + 3.
    // K element 8 g + 4 t2 + i of a half is source column 4 g + 16 t2 + i (not 8 g + 4 t2 + i): the two 16-lane groups of a
    // half-wave then read pixels 4 apart, whose chunk swizzles differ -- conflict-free; 8 apart they share every bank (the
    // 13.8 % LDS conflict rate of round 3; enumerated with the lane groups of MI355X_MICROARCH.md)
    unsigned a_tr[2][2][2];       // [pair][K half][block of 4 pixels]
    i32x4 wx[2][2];               // [pair][K half]
    int voff[2];
    {
        const int q = px16 >> 2, pp = px16 & 3;
#pragma unroll
        for (int u = 0; u < 2; ++u) {
            // (the bilinear tables are indexed by full-row output columns and hold full-row source columns: both block-relative here)
            const int xo_first = min(xo_run + 16 * u, Wo - 1);
            const int xs_u = min(a.rlo[xo0 + xo_first] - x0, W - 64);
            const int xo = xo_run + 16 * u + px16;
            const bool valid = 16 * u + px16 < nout_run && xo < Wo;
            voff[u] = valid ? ((xo0 + xo) * 64 + 16 * cq + 4 * g) * 2 : OOB;
            const int xq = xo0 + min(xo, Wo - 1);
            const int plo = a.rlo[xq] - x0, phi = a.rhi[xq] - x0;
            const float xlq = res_quant_lerp<DT>(a.rlerp[xq]);
#pragma unroll
            for (int kh = 0; kh < 2; ++kh) {
                unsigned short wh[8];
#pragma unroll
                for (int j = 0; j < 8; ++j) {
                    const int xin = xs_u + 32 * kh + 4 * g + 16 * (j >> 2) + (j & 3);
                    float w = 0.f;
                    if (xin == plo) w += 1.0f - xlq;
                    if (xin == phi) w += xlq;
                    wh[j] = to16<DT>(w);
                }
#pragma unroll
                for (int d = 0; d < 4; ++d) wx[u][kh][d] = static_cast<int>(static_cast<unsigned>(wh[2 * d]) | (static_cast<unsigned>(wh[2 * d + 1]) << 16));
#pragma unroll
                for (int t2 = 0; t2 < 2; ++t2) {
                    const int pix = xs_u + 32 * kh + 4 * g + 16 * t2 + q;
                    const int chunk = 2 * cq + (pp >> 1);
                    a_tr[u][kh][t2] = ring_lds + static_cast<unsigned>(pix * 128 + ((chunk ^ swz8(pix)) << 4) + (pp & 1) * 8);
                }
            }
        }
    }
    asm volatile("" : "+v"(wx[0][0]), "+v"(wx[0][1]), "+v"(wx[1][0]), "+v"(wx[1][1]));
    // folded BN of the lane's 4 couts (16 cq + 4 g + i): y = rs * sc2 + (S * sc1' + sh1')
    const f32x4 sc1 = *reinterpret_cast<const f32x4*>(a.ptab + 16 * cq + 4 * g);
    const f32x4 sh1 = *reinterpret_cast<const f32x4*>(a.ptab + 64 + 16 * cq + 4 * g);
    const f32x4 sc2 = *reinterpret_cast<const f32x4*>(a.ptab + 128 + 16 * cq + 4 * g);

    // ---- state
    f32x4 acc[3][4];              // partial accumulators of conv rows s, s-1, s-2 (index = conv row mod 3), per tile
    int hp[4][2], pp2[4][2];      // ReLU6'd even conv row (fp16 pairs) waiting for its odd partner; pair sum of the previous row pair
#pragma unroll
    for (int k = 0; k < 4; ++k) {
        hp[k][0] = hp[k][1] = pp2[k][0] = pp2[k][1] = 0;
#pragma unroll
        for (int r3 = 0; r3 < 3; ++r3) acc[r3][k] = zero4;
    }
    const int out_row_bytes = Wo_full * 128;
    const char* const out_img = reinterpret_cast<const char*>(a.out + static_cast<int64_t>(n) * Ho * Wo_full * 64);

#pragma unroll
    for (int j = 0; j < V_AHEAD; ++j) issue_row(j, j);
    wait_vmcnt<0>();
#pragma unroll
    for (int f = 0; f < 3 * NF; ++f) asm volatile("" : "+v"(wf[f]));
    // K48: what the constant input channels add to every conv output of the lane's cout (D'[pixel][cout]: cout = lane % 16)
    f32x4 cst4 = zero4;
    if constexpr (K48) {
        const float c0 = a.cstart[16 * cq + px16];
        cst4 = f32x4{c0, c0, c0, c0};
    }
    lds_barrier();

    int slot_cur = 0;             // ring slot of input row s
    // One step = one input row s (band-relative).  R = s mod 3 and the parity of s are compile time (unrolled by 6).
    auto step = [&](auto RC, auto PARC, int s) __attribute__((always_inline)) {
        constexpr int R = decltype(RC)::value, PAR = decltype(PARC)::value;
        constexpr int iN = R, iM = (R + 2) % 3, iO = (R + 1) % 3;      // accumulators of conv rows s, s-1, s-2
        // row s has landed.  VM_CNT counts the output stores too and retires in order: a step issues two DMA pieces at its
        // top and, on odd rows, two stores at its end.  Younger than the DMA of row s (issued at step s - 3): even s: stores
        // of s - 3, DMA of s - 2, DMA + stores of s - 1 = 2 + 2 + 2 + 2; odd s: DMA + stores of s - 2, DMA of s - 1 = 2 + 2 + 2.
        wait_vmcnt<PAR == 0 ? 8 : 6>();
        raw_barrier();
        {
            int sl = slot_cur + V_AHEAD;
            sl = sl >= V_NS ? sl - V_NS : sl;
            issue_row(s + V_AHEAD, sl);                                 // into the slot of row s-6: nobody reads it any more
        }
        const unsigned so = static_cast<unsigned>(slot_cur * V_ROW);
        unsigned bc[3][2];
#pragma unroll
        for (int kx = 0; kx < 3; ++kx)
#pragma unroll
            for (int ch = 0; ch < 2; ++ch) bc[kx][ch] = base[kx][ch] + so;
        // A tile's six operand fragments are read in two batches of three (channel half 0, then 1), double-buffered across
        // batches AND tiles: batch b + 1's reads go out before batch b's nine MFMAs, so the LDS round trip never stands at
        // the head of a chain (a batch is only 9 MFMAs long)
        i32x4 fq[2][3];
        // reads of batch b: three fragments (channel half 0), then three (half 1) -- K48: two (the merged fragments)
        auto reads = [&](auto BC) __attribute__((always_inline)) {
            constexpr int b = decltype(BC)::value, k = b >> 1, ch = b & 1;
            constexpr int NR = (K48 && ch) ? 2 : 3;
            auto& dst = fq[b & 1];
            auto& bcr = bc;                                            // (named outside the asm: implicit capture)
#pragma unroll
            for (int kx = 0; kx < NR; ++kx) asm volatile("ds_read_b128 %0, %1 offset:%2" : "=v"(dst[kx]) : "v"(bcr[kx][ch]), "n"(k * 2048));
        };
        auto batch = [&](auto BC, auto NEXTC) __attribute__((always_inline)) {
            constexpr int b = decltype(BC)::value, k = b >> 1, ch = b & 1;
            constexpr bool NEXT = decltype(NEXTC)::value != 0;        // batch b + 1's reads are issued here
            constexpr int NR = (K48 && ch) ? 2 : 3;                    // this batch's fragments
            constexpr int NRN = NEXT ? ((K48 && !ch) ? 2 : 3) : 0;     // the next batch's, in flight behind them
            if constexpr (NEXT) reads(IC<b + 1>{});
            auto& cur = fq[b & 1];
            [&]<int... KX>(std::integer_sequence<int, KX...>) {
                (([&] {
                     constexpr int fi = K48 ? (ch ? 3 + KX : KX) : KX * 2 + ch;      // fragment index inside a kernel row
                     asm volatile("s_waitcnt lgkmcnt(%1)" : "+v"(cur[KX]) : "n"(NRN + NR - 1 - KX));
                     acc[iN][k] = mfma16<DT>(cur[KX], wf[0 * NF + fi], (ch == 0 && KX == 0) ? cst4 : acc[iN][k]);
                     acc[iM][k] = mfma16<DT>(cur[KX], wf[1 * NF + fi], acc[iM][k]);
                     acc[iO][k] = mfma16<DT>(cur[KX], wf[2 * NF + fi], acc[iO][k]);
                 }()),
                 ...);
            }(std::make_integer_sequence<int, NR>{});
        };
        auto tile = [&](auto KC, auto LASTC) __attribute__((always_inline)) {
            constexpr int k = decltype(KC)::value;
            constexpr bool LAST = decltype(LASTC)::value != 0;        // no batch follows this tile's second one
            batch(IC<2 * k>{}, IC<1>{});
            batch(IC<2 * k + 1>{}, IC<(LAST ? 0 : 1)>{});
        };
        // conv row j = s - 2 is complete in acc[iO] after the tile's chain: ReLU6 -> fp16 pairs; even rows wait in hp, odd
        // rows form the pair sum and the pooling operand [previous pair sum | this pair sum]
        i32x4 op[4];
        op[3] = i32x4{0, 0, 0, 0};
        auto finish = [&](auto KC) __attribute__((always_inline)) {
            constexpr int k = decltype(KC)::value;
            const f32x4 v = acc[iO][k];
            const int v0 = static_cast<int>(pack2_relu6_sixth(v[0], v[1]));
            const int v1 = static_cast<int>(pack2_relu6_sixth(v[2], v[3]));
            if constexpr (PAR == 0) {
                hp[k][0] = v0;
                hp[k][1] = v1;
            } else {
                const int n0 = pk_add_f16(hp[k][0], v0), n1 = pk_add_f16(hp[k][1], v1);
                op[k] = i32x4{pp2[k][0], pp2[k][1], n0, n1};
                pp2[k][0] = n0;
                pp2[k][1] = n1;
            }
        };
        // K48: a tile's five fragments are ONE batch of 15 MFMAs, read a whole tile ahead (a (3, 2) split left the two-fragment batch's
        // six MFMAs = ~100 cycles to cover an LDS round trip: the stage ran 5 % slower than with 18 MFMAs per tile)
        i32x4 fq5[2][K48 ? 5 : 1];
        auto reads5 = [&](auto KC) __attribute__((always_inline)) {
            constexpr int k = decltype(KC)::value;
            auto& dst = fq5[k & 1];
            auto& bcr = bc;
#pragma unroll
            for (int f = 0; f < 5; ++f) asm volatile("ds_read_b128 %0, %1 offset:%2" : "=v"(dst[f]) : "v"(f < 3 ? bcr[f][0] : bcr[f - 3][1]), "n"(k * 2048));
        };
        auto tile5 = [&](auto KC, auto NEXTC) __attribute__((always_inline)) {
            constexpr int k = decltype(KC)::value;
            constexpr bool NEXT = decltype(NEXTC)::value != 0;        // tile k + 1's reads are issued here (5 more in flight)
            if constexpr (NEXT) reads5(IC<k + 1>{});
            auto& cur = fq5[k & 1];
            [&]<int... F>(std::integer_sequence<int, F...>) {
                (([&] {
                     asm volatile("s_waitcnt lgkmcnt(%1)" : "+v"(cur[F]) : "n"((NEXT ? 5 : 0) + 4 - F));
                     acc[iN][k] = mfma16<DT>(cur[F], wf[0 * NF + F], F == 0 ? cst4 : acc[iN][k]);
                     acc[iM][k] = mfma16<DT>(cur[F], wf[1 * NF + F], acc[iM][k]);
                     acc[iO][k] = mfma16<DT>(cur[F], wf[2 * NF + F], acc[iO][k]);
                 }()),
                 ...);
            }(std::make_integer_sequence<int, 5>{});
        };
        if constexpr (K48) {
            if (conv_live) {              // (wave-uniform)
                reads5(IC<0>{});
                tile5(IC<0>{}, IC<1>{});
                finish(IC<0>{});
                tile5(IC<1>{}, IC<1>{});
                finish(IC<1>{});
                tile5(IC<2>{}, IC<0>{});
                finish(IC<2>{});
                if (has4) {               // (the fourth tile's reads stay behind the branch)
                    reads5(IC<3>{});
                    tile5(IC<3>{}, IC<0>{});
                    finish(IC<3>{});
                }
            }
        } else
        if (conv_live) {                  // (wave-uniform)
            reads(IC<0>{});
            tile(IC<0>{}, IC<0>{});
            finish(IC<0>{});
            tile(IC<1>{}, IC<0>{});
            finish(IC<1>{});
            tile(IC<2>{}, IC<1>{});
            finish(IC<2>{});
            if (has4) {                   // (the fourth tile's first reads stay behind the branch)
                reads(IC<6>{});
                tile(IC<3>{}, IC<1>{});
                finish(IC<3>{});
            }
        }
        if constexpr (PAR == 1) {
            // odd conv row j = s - 2 >= 3 completes pooled row r = (j - 3) / 2
            const int r = (s - 5) >> 1;
            const bool emit = s >= 5 && r < nrows;
            const int rr = min(max(r, 0), nrows - 1);
            const float src = mul_rounded(static_cast<float>(yo0 + rr), a.rscale);
            const int ylo_v = static_cast<int>(src);
            const int ylo = __builtin_amdgcn_readfirstlane(ylo_v);
            const float yl = __int_as_float(__builtin_amdgcn_readfirstlane(__float_as_int(src - static_cast<float>(ylo_v))));
            const int yhi = min(ylo + 1, Win - 1);
            // ring slots of the two source rows (band-relative rows ylo - y0, yhi - y0; row s sits in slot_cur)
            int dlo = s - (ylo - y0), dhi = s - (yhi - y0);            // 0 .. 5 rows behind the newest
            dlo = min(max(dlo, 0), V_NS - 1);
            dhi = min(max(dhi, 0), V_NS - 1);
            int slo = slot_cur - dlo, shi = slot_cur - dhi;
            slo = slo < 0 ? slo + V_NS : slo;
            shi = shi < 0 ? shi + V_NS : shi;
            const unsigned olo = static_cast<unsigned>(slo * V_ROW), ohi = static_cast<unsigned>(shi * V_ROW);
            const char* orow = out_img + static_cast<int64_t>(yo0 + rr) * out_row_bytes;
            const __amdgpu_buffer_rsrc_t rs = __builtin_amdgcn_make_buffer_rsrc(const_cast<char*>(orow), 0, out_row_bytes, 0x00020000);
            const int emask = emit ? 0 : OOB;
            [[maybe_unused]] const unsigned seed = (!a.dither || cq == a.plain_q) ? RN_SEED_PLAIN : rn_dither_seed(yo0 + rr);      // (bf16 handles: rn_stage.h)
            if (!epi_live) {
                // (K48, last quarter: nothing to compute -- its waves store the constants, so that the pixels leave as full lines and
                //  VM_CNT counts two stores per odd step in every wave: the counted wait at the top of a step is one immediate)
                const i32x2 cv = *reinterpret_cast<const i32x2*>(a.cvals + 4 * g);     // (the lane's channels 48 + 4 g .. + 3; full lines)
                __builtin_amdgcn_raw_buffer_store_b64(cv, rs, voff[0] | emask, 0, 0);
                __builtin_amdgcn_raw_buffer_store_b64(cv, rs, voff[1] | emask, 0, 0);
            } else
#pragma unroll
            for (int u = 0; u < 2; ++u) {
                i32x2 tq[2][2][2];          // [lo / hi][K half][block]
#pragma unroll
                for (int kh = 0; kh < 2; ++kh)
#pragma unroll
                    for (int t2 = 0; t2 < 2; ++t2) {
                        const unsigned alo = a_tr[u][kh][t2] + olo, ahi = a_tr[u][kh][t2] + ohi;
                        asm volatile("ds_read_b64_tr_b16 %0, %1" : "=v"(tq[0][kh][t2]) : "v"(alo));
                        asm volatile("ds_read_b64_tr_b16 %0, %1" : "=v"(tq[1][kh][t2]) : "v"(ahi));
                    }
                f32x4 H = zero4;           // frozen quarters: fma(H, sc1, sh1) = sh1 for every H the convolution can produce
                if (conv_live) {
                    H = mfma16<RN_DTYPE_F16>(op[2 * u], pmA, zero4);
                    H = mfma16<RN_DTYPE_F16>(op[2 * u + 1], pmB, H);
                    if (u == 0) H = mfma16<RN_DTYPE_F16>(op[2], pmC, H);
                }
                asm volatile("s_waitcnt lgkmcnt(0)"
                             : "+v"(tq[0][0][0]), "+v"(tq[0][0][1]), "+v"(tq[0][1][0]), "+v"(tq[0][1][1]), "+v"(tq[1][0][0]), "+v"(tq[1][0][1]),
                               "+v"(tq[1][1][0]), "+v"(tq[1][1][1]));
                f32x4 r_lo = zero4, r_hi = zero4;
#pragma unroll
                for (int kh = 0; kh < 2; ++kh) {
                    const i32x4 al = {tq[0][kh][0][0], tq[0][kh][0][1], tq[0][kh][1][0], tq[0][kh][1][1]};
                    const i32x4 ah = {tq[1][kh][0][0], tq[1][kh][0][1], tq[1][kh][1][0], tq[1][kh][1][1]};
                    r_lo = mfma16<DT>(al, wx[u][kh], r_lo);
                    r_hi = mfma16<DT>(ah, wx[u][kh], r_hi);
                }
                float y[4];
#pragma unroll
                for (int i = 0; i < 4; ++i) {
                    const float y1 = __builtin_fmaf(H[i], sc1[i], sh1[i]);
                    const float lo = r_lo[i];
                    const float rsv = __builtin_fmaf(r_hi[i] - lo, yl, lo);
                    y[i] = __builtin_fmaf(rsv, sc2[i], y1);
                }
                i32x2 d;
                if constexpr (DT == RN_DTYPE_BF16)
                    d = i32x2{static_cast<int>(pack2_sr_bf16(y[0], y[1], seed)), static_cast<int>(pack2_sr_bf16(y[2], y[3], seed))};
                else
                    d = i32x2{static_cast<int>(pack2<DT>(y[0], y[1])), static_cast<int>(pack2<DT>(y[2], y[3]))};
                __builtin_amdgcn_raw_buffer_store_b64(d, rs, voff[u] | emask, 0, 0);
            }
        }
        slot_cur = slot_cur == V_NS - 1 ? 0 : slot_cur + 1;
        asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
    };
    int s = 0;
    for (; s + 5 < nin; s += 6) {
        step(IC<0>{}, IC<0>{}, s);
        step(IC<1>{}, IC<1>{}, s + 1);
        step(IC<2>{}, IC<0>{}, s + 2);
        step(IC<0>{}, IC<1>{}, s + 3);
        step(IC<1>{}, IC<0>{}, s + 4);
        step(IC<2>{}, IC<1>{}, s + 5);
    }
    const int rem = nin - s;
    if (rem > 0) step(IC<0>{}, IC<0>{}, s);
    if (rem > 1) step(IC<1>{}, IC<1>{}, s + 1);
    if (rem > 2) step(IC<2>{}, IC<0>{}, s + 2);
    if (rem > 3) step(IC<0>{}, IC<1>{}, s + 3);
    if (rem > 4) step(IC<1>{}, IC<0>{}, s + 4);
    wait_vmcnt<0>();
#ifdef RN_CLOCK
    if (a.stamp_buf && threadIdx.x == 256) {
        unsigned long long t1, r1;
        clock_pair(t1, r1);
        const int64_t wg = static_cast<int64_t>(blockIdx.y) * gridDim.x + blockIdx.x;
        a.stamp_buf[wg * 2 + 0] = t1 - ck_t0;
        a.stamp_buf[wg * 2 + 1] = r1 - ck_r0;
    }
#endif
}

}  // namespace

// pooled columns per block: 2 wo + 4 .. 2 wo + 6 staged input columns must lie in V_WMIN .. V_WMAX.  The residual reaches
// (in_side - 2 out_side) < 5 columns / rows past the conv window's first column / row: inside the staged columns and the
// 9-slot ring for every input the network can produce (in_side = 2 out_side + 4 or + 5).
bool rn_stage5x_plan(int out_side, int* n_cb, int* xo0, int* wo) {
    return rn_colblock_plan(out_side, (V_WMIN - 4 + 1) / 2, (V_WMAX - 4) / 2, n_cb, xo0, wo);
}

bool rn_stage5x_supported(int cin, int cout, int pool_k, int pool_s, bool res, int in_side, int skip_side) {
    int ncb, xo0[4], wo[4];
    const int out_side = (in_side - 6) / 2 + 1;
    return cin == 64 && cout == 64 && pool_k == 4 && pool_s == 2 && res && in_side >= V_WMIN && skip_side == in_side &&
           in_side - 2 * out_side <= 5 && rn_stage5x_plan(out_side, &ncb, xo0, wo);
}

// B-operand fragments: frag[f = (ky * 3 + kx) * 2 + ch][cout quarter q][lane][j] = W[tap ky * 3 + kx][channel 32 ch + 8 (lane / 16) + j][cout 16 q + lane % 16]
void rn_stage5x_pack(const float* w_hwio, int dtype, unsigned short (*cvt_bf16)(float), unsigned short (*cvt_f16)(float),
                     std::vector<unsigned short>* out) {
    out->assign(static_cast<size_t>(18) * 4 * 64 * 8, 0);
    for (int f = 0; f < 18; ++f)
        for (int q = 0; q < 4; ++q)
            for (int l = 0; l < 64; ++l)
                for (int j = 0; j < 8; ++j) {
                    const int tap = f >> 1, ch = f & 1;
                    const int k = tap * 64 + 32 * ch + 8 * (l >> 4) + j, co = 16 * q + (l & 15);
                    const float v = w_hwio[static_cast<size_t>(k) * 64 + co];
                    (*out)[((static_cast<size_t>(f) * 4 + q) * 64 + l) * 8 + j] = dtype == RN_DTYPE_BF16 ? cvt_bf16(v) : cvt_f16(v);
                }
}

// K48 fragments: frag[f = ky * 5 + j][cout quarter q][lane][e]: j < 3 = W[tap (ky, kx = j)][channel 8 (lane / 16) + e]; j = 3: lane groups
// 0, 1 = W[tap (ky, 0)][channel 32 + 8 g + e], groups 2, 3 = W[tap (ky, 1)][channel 32 + 8 (g - 2) + e]; j = 4: groups 0, 1 =
// W[tap (ky, 2)][channel 32 + 8 g + e], groups 2, 3 = 0.  Channels 48..63 (constants on the handle) do not appear.
void rn_stage5x_pack48(const float* w_hwio, int dtype, unsigned short (*cvt_bf16)(float), unsigned short (*cvt_f16)(float),
                       std::vector<unsigned short>* out) {
    out->assign(static_cast<size_t>(15) * 4 * 64 * 8, 0);
    for (int ky = 0; ky < 3; ++ky)
        for (int j = 0; j < 5; ++j)
            for (int q = 0; q < 4; ++q)
                for (int l = 0; l < 64; ++l)
                    for (int e = 0; e < 8; ++e) {
                        const int g = l >> 4, co = 16 * q + (l & 15);
                        int kx, ch;
                        if (j < 3) {
                            kx = j;
                            ch = 8 * g + e;
                        } else if (j == 3) {
                            kx = g >> 1;
                            ch = 32 + 8 * (g & 1) + e;
                        } else {
                            if (g >= 2) continue;
                            kx = 2;
                            ch = 32 + 8 * g + e;
                        }
                        const float v = w_hwio[(static_cast<size_t>(ky * 3 + kx) * 64 + ch) * 64 + co];
                        (*out)[((static_cast<size_t>(ky * 5 + j) * 4 + q) * 64 + l) * 8 + e] = dtype == RN_DTYPE_BF16 ? cvt_bf16(v) : cvt_f16(v);
                    }
}

int rn_stage5x_launch(int dtype, hipStream_t s, const StageArgs& a, int n) {
    auto launch = [&](auto kern) -> int {
        static std::atomic<unsigned long long> attr_devices{0};
        int dev = 0;
        RN_HIP(hipGetDevice(&dev));
        if (!(attr_devices.load(std::memory_order_acquire) >> (dev & 63) & 1ull)) {
            RN_HIP(hipFuncSetAttribute(reinterpret_cast<const void*>(kern), hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024));
            attr_devices.fetch_or(1ull << (dev & 63), std::memory_order_release);
        }
        hipLaunchKernelGGL(kern, dim3(a.n_bands * a.n_cb, n), dim3(512), V_LDS, s, a);
        RN_CHECK_LAUNCH();
        return RN_OK;
    };
    if (a.cstart && a.live_q == 2) {
        if (dtype == RN_DTYPE_BF16) return launch(stage5x_kernel<RN_DTYPE_BF16, true>);
        return launch(stage5x_kernel<RN_DTYPE_F16, true>);
    }
    if (dtype == RN_DTYPE_BF16) return launch(stage5x_kernel<RN_DTYPE_BF16, false>);
    return launch(stage5x_kernel<RN_DTYPE_F16, false>);
}
