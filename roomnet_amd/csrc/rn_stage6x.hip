// Un-pooled 64 -> 128 stage (reference network.py:229, conv_block(128, pooling=False)) on 16x16x32 matrix tiles with
// ROW-REGISTER BLOCKING (the design of rn_stage4x.hip / rn_stage5x.hip):
//
//   in [N, W, W, 64] -> conv3x3 VALID -> ReLU6 -> BN(inference)  = out [N, W-2, W-2, 128]
//
// One workgroup = one image (x band of rows); eight waves = the eight 16-cout groups, each for all three 16-pixel tiles of
// the row (46 conv columns at 224 x 224).  A wave keeps its 16 couts' weights in 72 registers (9 taps x 2 channel halves)
// and the partial accumulators of conv rows s, s-1, s-2 for its three tiles in 36; every operand fragment of the NEWEST
// input row feeds three MFMAs (kernel rows 0, 1, 2), so a step reads 6 fragments per tile for 18 MFMAs and the ring only
// holds the newest row and the rows in flight.  Operands are NOT swapped here (D[cout][pixel]: a lane holds 4 consecutive
// couts of one pixel = one 8-byte store): there is no pooling that would want the pixels in registers.
// Wider rows are cut into COLUMN BLOCKS of 33..48 output columns (rn_colblock_plan: one block at 224, three at 600):
// workgroup = image x band x block.  conv16_kernel (rn_conv16.hip) stays as the RN_FLAG_PAIR_32X32 arm.
#include "rn_fused.h"
#include "rn_stage.h"

#include <atomic>
#include <utility>

using namespace rnk;

namespace {

constexpr int S6_NS = 4, S6_AHEAD = 3;
constexpr int S6_RINGPX = 50;                     // 3 tiles + 2 halo columns
constexpr int S6_ROW = 52 * 128;                  // bytes per ring row (padded)
constexpr int S6_LDS = S6_NS * S6_ROW;
constexpr int S6_WMIN = 35, S6_WMAX = 50;

__device__ __forceinline__ unsigned lds_addr(const void* p) {
    return static_cast<unsigned>(reinterpret_cast<uintptr_t>((__attribute__((address_space(3))) const char*)p));
}
__device__ __forceinline__ int swz8(int pix) { return pix & 7; }

using i32x2 = __attribute__((ext_vector_type(2))) int;

template <int DT>
__device__ __forceinline__ f32x4 mfma16(i32x4 a, i32x4 b, f32x4 c) {
    if constexpr (DT == RN_DTYPE_BF16)
        return __builtin_amdgcn_mfma_f32_16x16x32_bf16(__builtin_bit_cast(bf16x8, a), __builtin_bit_cast(bf16x8, b), c, 0, 0, 0);
    else
        return __builtin_amdgcn_mfma_f32_16x16x32_f16(__builtin_bit_cast(f16x8, a), __builtin_bit_cast(f16x8, b), c, 0, 0, 0);
}

// K48 (round 6): the input channels 48..63 are constants of the handle (the residual stage in front writes them from a table,
// rn_fused_prepare): five operand fragments per tile and row step instead of six -- rn_stage5x.hip's scheme: (kx, channels 0..31)
// for kx = 0, 1, 2, then [kx 0 | kx 1] and [kx 2 | zero weights] of channels 32..47 -- read a whole tile ahead, and the
// accumulators start from the constants' contribution (StageArgs::cstart).
template <int DT, bool K48>
__global__ __launch_bounds__(512, 2) void stage6x_kernel(const StageArgs a) {
    constexpr int NF = K48 ? 5 : 6;
#ifdef RN_CLOCK
    unsigned long long ck_t0, ck_r0;
    clock_pair(ck_t0, ck_r0);
#endif
    extern __shared__ __attribute__((aligned(64))) char smem[];
    const int tid = threadIdx.x;
    const int lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);      // = cout group (16 couts)
    const int px16 = lane & 15, g = lane >> 4;
    const int cb = blockIdx.x % a.n_cb, band = blockIdx.x / a.n_cb, n = blockIdx.y;
    const int Win = a.W, Wo_full = a.Wo, Ho = a.Ho;
    const int x0 = a.cb_xo0[cb], Wo = a.cb_wo[cb];        // this block's output columns = its first input column
    const int W = Wo + 2;                                 // input columns it reads
    const int yo0 = band * a.rows_per_band;
    const int nrows = min(Ho, yo0 + a.rows_per_band) - yo0;
    const int nin = nrows + 2;
    const f32x4 zero4 = {0.f, 0.f, 0.f, 0.f};
    constexpr int OOB = 0x40000000;

    char* const ring = smem;
    const unsigned ring_lds = lds_addr(ring);
    // pixels W .. 51 of every slot are never written by the row DMA: zero them once
    for (int i = tid; i < S6_NS * (52 - S6_WMIN) * 8; i += 512) {
        const int slot = i / ((52 - S6_WMIN) * 8), rest = i % ((52 - S6_WMIN) * 8);
        const int p = S6_WMIN + rest / 8, c = rest % 8;
        if (p >= W) *reinterpret_cast<i32x4*>(ring + slot * S6_ROW + p * 128 + c * 16) = i32x4{0, 0, 0, 0};
    }

    // ---- weights: fragment f = (ky * 3 + kx) * 2 + ch (K48: ky * 5 + j, rn_stage6x_pack48), A operand of D[cout][pixel]
    i32x4 wf[3 * NF];
#pragma unroll
    for (int f = 0; f < 3 * NF; ++f) {
        const i32x4* src = a.wfrag + (f * 8 + wave) * 64 + lane;
        asm volatile("global_load_dwordx4 %0, %1, off" : "=v"(wf[f]) : "v"(src) : "memory");
    }

    // ---- input rows by LDS-DMA: one lane-masked piece per wave and row (W x 8 chunks, W per wave)
    const char* const in_img = reinterpret_cast<const char*>(a.in + static_cast<int64_t>(n) * Win * Win * 64) + x0 * 128;
    const int row_bytes = Win * 128;
    const unsigned long long dma_mask = (1ull << W) - 1ull;
    unsigned goff;
    {
        const int q = wave * W + min(lane, W - 1);
        const int p = q >> 3, c = q & 7;
        goff = static_cast<unsigned>(p * 128 + ((c ^ swz8(p)) << 4));
    }
    auto issue_row = [&](int y, int slot) __attribute__((always_inline)) {
        const char* row = in_img + static_cast<int64_t>(yo0 + min(y, nin - 1)) * row_bytes;
        unsigned o = goff;
        asm volatile("" : "+v"(o));
        dma16_masked(row + o, ring + slot * S6_ROW + wave * W * 16, dma_mask);
    };

    unsigned base[3][2];
#pragma unroll
    for (int kx = 0; kx < 3; ++kx)
#pragma unroll
        for (int ch = 0; ch < 2; ++ch) {
            const int p = px16 + kx;
            base[kx][ch] = ring_lds + static_cast<unsigned>(p * 128 + (((4 * ch + g) ^ swz8(p)) << 4));
        }
    f32x4 cst4 = zero4;
    if constexpr (K48) {
#pragma unroll
        for (int j = 0; j < 2; ++j) {          // merged fragments: lane groups 0, 1 = chunks 4, 5 of tap column 0 (j = 1: 2), groups 2, 3 of column 1
            const int p = px16 + (j == 0 ? (g >> 1) : 2);
            base[j][1] = ring_lds + static_cast<unsigned>(p * 128 + (((4 + (g & 1)) ^ swz8(p)) << 4));
        }
        cst4 = *reinterpret_cast<const f32x4*>(a.cstart + 16 * wave + 4 * g);      // (D[cout][pixel]: the lane's 4 couts)
    }
    int voff[3];
#pragma unroll
    for (int k = 0; k < 3; ++k) {
        const int xo = 16 * k + px16;
        voff[k] = xo < Wo ? ((x0 + xo) * 128 + 16 * wave + 4 * g) * 2 : OOB;
    }
    const f32x4 sc = *reinterpret_cast<const f32x4*>(a.ptab + 16 * wave + 4 * g);
    const f32x4 sh = *reinterpret_cast<const f32x4*>(a.ptab + 128 + 16 * wave + 4 * g);

    f32x4 acc[3][3];
#pragma unroll
    for (int r3 = 0; r3 < 3; ++r3)
#pragma unroll
        for (int k = 0; k < 3; ++k) acc[r3][k] = zero4;
    const int out_row_bytes = Wo_full * 256;
    const char* const out_img = reinterpret_cast<const char*>(a.out + static_cast<int64_t>(n) * Ho * Wo_full * 128);

#pragma unroll
    for (int j = 0; j < S6_AHEAD; ++j) issue_row(j, j);
    wait_vmcnt<0>();
#pragma unroll
    for (int f = 0; f < 3 * NF; ++f) asm volatile("" : "+v"(wf[f]));
    lds_barrier();

    int slot_cur = 0;
    auto step = [&](auto RC, int s) __attribute__((always_inline)) {
        constexpr int R = decltype(RC)::value;
        constexpr int iN = R, iM = (R + 2) % 3, iO = (R + 1) % 3;
        // row s has landed.  VM_CNT counts the output stores too and retires in order: per step this wave issues one DMA
        // piece and then three stores (always: masked lanes store out of range), so 3 + 4 + 4 operations are younger than
        // the DMA of row s.  (Waiting for all but AHEAD - 1 made every step wait for the previous step's stores.)
        wait_vmcnt<3 + 4 * (S6_AHEAD - 1)>();
        raw_barrier();
        {
            int sl = slot_cur + S6_AHEAD;
            sl = sl >= S6_NS ? sl - S6_NS : sl;
            issue_row(s + S6_AHEAD, sl);
        }
        const unsigned so = static_cast<unsigned>(slot_cur * S6_ROW);
        unsigned bc[3][2];
#pragma unroll
        for (int kx = 0; kx < 3; ++kx)
#pragma unroll
            for (int ch = 0; ch < 2; ++ch) bc[kx][ch] = base[kx][ch] + so;
        const int j = s - 2;                                            // conv row completed by this step
        const int jr = min(max(j, 0), nrows - 1);
        const char* orow = out_img + static_cast<int64_t>(yo0 + jr) * out_row_bytes;
        const __amdgpu_buffer_rsrc_t rs = __builtin_amdgcn_make_buffer_rsrc(const_cast<char*>(orow), 0, out_row_bytes, 0x00020000);
        const int emask = (j >= 0 && j < nrows) ? 0 : OOB;
        i32x4 fq[2][3];
        auto reads = [&](auto BC) __attribute__((always_inline)) {
            constexpr int b = decltype(BC)::value, k = b >> 1, ch = b & 1;
            auto& dst = fq[b & 1];
            auto& bcr = bc;
#pragma unroll
            for (int kx = 0; kx < 3; ++kx) asm volatile("ds_read_b128 %0, %1 offset:%2" : "=v"(dst[kx]) : "v"(bcr[kx][ch]), "n"(k * 2048));
        };
        auto batch = [&](auto BC, auto NEXTC) __attribute__((always_inline)) {
            constexpr int b = decltype(BC)::value, k = b >> 1, ch = b & 1;
            constexpr bool NEXT = decltype(NEXTC)::value != 0;
            if constexpr (NEXT) reads(IC<b + 1>{});
            auto& cur = fq[b & 1];
            [&]<int... KX>(std::integer_sequence<int, KX...>) {
                (([&] {
                     asm volatile("s_waitcnt lgkmcnt(%1)" : "+v"(cur[KX]) : "n"((NEXT ? 3 : 0) + 2 - KX));
                     acc[iN][k] = mfma16<DT>(wf[(0 * 3 + KX) * 2 + ch], cur[KX], (ch == 0 && KX == 0) ? zero4 : acc[iN][k]);
                     acc[iM][k] = mfma16<DT>(wf[(1 * 3 + KX) * 2 + ch], cur[KX], acc[iM][k]);
                     acc[iO][k] = mfma16<DT>(wf[(2 * 3 + KX) * 2 + ch], cur[KX], acc[iO][k]);
                 }()),
                 ...);
            }(std::make_integer_sequence<int, 3>{});
        };
        auto emit = [&](auto KC) __attribute__((always_inline)) {
            constexpr int k = decltype(KC)::value;
            const f32x4 v = acc[iO][k];
            float y[4];
#pragma unroll
            for (int i = 0; i < 4; ++i) y[i] = __builtin_fmaf(relu6f(v[i]), sc[i], sh[i]);
            const i32x2 d = {static_cast<int>(pack2<DT>(y[0], y[1])), static_cast<int>(pack2<DT>(y[2], y[3]))};
            __builtin_amdgcn_raw_buffer_store_b64(d, rs, voff[k] | emask, 0, 0);
        };
        [[maybe_unused]] i32x4 fq5[2][K48 ? 5 : 1];
        auto reads5 = [&](auto KC) __attribute__((always_inline)) {
            constexpr int k = decltype(KC)::value;
            auto& dst = fq5[k & 1];
            auto& bcr = bc;
#pragma unroll
            for (int f = 0; f < 5; ++f) asm volatile("ds_read_b128 %0, %1 offset:%2" : "=v"(dst[f]) : "v"(f < 3 ? bcr[f][0] : bcr[f - 3][1]), "n"(k * 2048));
        };
        auto tile5 = [&](auto KC, auto NEXTC) __attribute__((always_inline)) {
            constexpr int k = decltype(KC)::value;
            constexpr bool NEXT = decltype(NEXTC)::value != 0;
            if constexpr (NEXT) reads5(IC<k + 1>{});
            auto& cur = fq5[k & 1];
            [&]<int... F>(std::integer_sequence<int, F...>) {
                (([&] {
                     asm volatile("s_waitcnt lgkmcnt(%1)" : "+v"(cur[F]) : "n"((NEXT ? 5 : 0) + 4 - F));
                     acc[iN][k] = mfma16<DT>(wf[0 * NF + F], cur[F], F == 0 ? cst4 : acc[iN][k]);
                     acc[iM][k] = mfma16<DT>(wf[1 * NF + F], cur[F], acc[iM][k]);
                     acc[iO][k] = mfma16<DT>(wf[2 * NF + F], cur[F], acc[iO][k]);
                 }()),
                 ...);
            }(std::make_integer_sequence<int, 5>{});
        };
        if constexpr (K48) {
            reads5(IC<0>{});
            tile5(IC<0>{}, IC<1>{});
            emit(IC<0>{});
            tile5(IC<1>{}, IC<1>{});
            emit(IC<1>{});
            tile5(IC<2>{}, IC<0>{});
            emit(IC<2>{});
        } else {
            reads(IC<0>{});
            batch(IC<0>{}, IC<1>{});
            batch(IC<1>{}, IC<1>{});
            emit(IC<0>{});
            batch(IC<2>{}, IC<1>{});
            batch(IC<3>{}, IC<1>{});
            emit(IC<1>{});
            batch(IC<4>{}, IC<1>{});
            batch(IC<5>{}, IC<0>{});
            emit(IC<2>{});
        }
        slot_cur = slot_cur == S6_NS - 1 ? 0 : slot_cur + 1;
        asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
    };
    int s = 0;
    for (; s + 2 < nin; s += 3) {
        step(IC<0>{}, s);
        step(IC<1>{}, s + 1);
        step(IC<2>{}, s + 2);
    }
    const int rem = nin - s;
    if (rem > 0) step(IC<0>{}, s);
    if (rem > 1) step(IC<1>{}, s + 1);
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

bool rn_stage6x_plan(int out_side, int* n_cb, int* xo0, int* wo) {
    return rn_colblock_plan(out_side, S6_WMIN - 2, S6_WMAX - 2, n_cb, xo0, wo);
}

bool rn_stage6x_supported(int cin, int cout, int pool_k, bool res, int in_side) {
    int ncb, xo0[4], wo[4];
    return cin == 64 && cout == 128 && pool_k == 0 && !res && in_side >= S6_WMIN && rn_stage6x_plan(in_side - 2, &ncb, xo0, wo);
}

// A-operand fragments: frag[f = (ky * 3 + kx) * 2 + ch][cout group q][lane][j] = W[tap][channel 32 ch + 8 (lane / 16) + j][cout 16 q + lane % 16]
void rn_stage6x_pack(const float* w_hwio, int dtype, unsigned short (*cvt_bf16)(float), unsigned short (*cvt_f16)(float),
                     std::vector<unsigned short>* out) {
    out->assign(static_cast<size_t>(18) * 8 * 64 * 8, 0);
    for (int f = 0; f < 18; ++f)
        for (int q = 0; q < 8; ++q)
            for (int l = 0; l < 64; ++l)
                for (int j = 0; j < 8; ++j) {
                    const int tap = f >> 1, ch = f & 1;
                    const int k = tap * 64 + 32 * ch + 8 * (l >> 4) + j, co = 16 * q + (l & 15);
                    const float v = w_hwio[static_cast<size_t>(k) * 128 + co];
                    (*out)[((static_cast<size_t>(f) * 8 + q) * 64 + l) * 8 + j] = dtype == RN_DTYPE_BF16 ? cvt_bf16(v) : cvt_f16(v);
                }
}

// K48 fragments (rn_stage5x_pack48's layout with eight 16-cout groups): frag[f = ky * 5 + j][group q][lane][e]
void rn_stage6x_pack48(const float* w_hwio, int dtype, unsigned short (*cvt_bf16)(float), unsigned short (*cvt_f16)(float),
                       std::vector<unsigned short>* out) {
    out->assign(static_cast<size_t>(15) * 8 * 64 * 8, 0);
    for (int ky = 0; ky < 3; ++ky)
        for (int j = 0; j < 5; ++j)
            for (int q = 0; q < 8; ++q)
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
                        const float v = w_hwio[(static_cast<size_t>(ky * 3 + kx) * 64 + ch) * 128 + co];
                        (*out)[((static_cast<size_t>(ky * 5 + j) * 8 + q) * 64 + l) * 8 + e] = dtype == RN_DTYPE_BF16 ? cvt_bf16(v) : cvt_f16(v);
                    }
}

int rn_stage6x_launch(int dtype, hipStream_t s, const StageArgs& a, int n) {
    auto launch = [&](auto kern) -> int {
        hipLaunchKernelGGL(kern, dim3(a.n_bands * a.n_cb, n), dim3(512), S6_LDS, s, a);
        RN_CHECK_LAUNCH();
        return RN_OK;
    };
    if (a.cstart) {
        if (dtype == RN_DTYPE_BF16) return launch(stage6x_kernel<RN_DTYPE_BF16, true>);
        return launch(stage6x_kernel<RN_DTYPE_F16, true>);
    }
    if (dtype == RN_DTYPE_BF16) return launch(stage6x_kernel<RN_DTYPE_BF16, false>);
    return launch(stage6x_kernel<RN_DTYPE_F16, false>);
}
