// Device helpers and launch-argument block shared by the fused stage kernels.
#pragma once
#include "rn_internal.h"

#include <type_traits>

namespace rnk {

using i32x4 = __attribute__((ext_vector_type(4))) int;
using f32x16 = __attribute__((ext_vector_type(16))) float;
using f32x4 = __attribute__((ext_vector_type(4))) float;
using f32x2 = __attribute__((ext_vector_type(2))) float;

// Packed fp32 arithmetic (v_pk_fma_f32 / v_pk_add_f32: two IEEE fp32 operations per VALU slot, bit-identical to the
// scalar forms).  The epilogues are bound by instruction issue, so BN and the residual lerp run on register pairs.
__device__ __forceinline__ f32x2 pk_fma(f32x2 a, f32x2 b, f32x2 c) { return __builtin_elementwise_fma(a, b, c); }
// (hipcc splits a v2f32 fsub into two v_sub_f32, also when written as fma(b, -1, a); an inline-asm v_pk_add_f32 with
//  neg modifiers cost the fused stage pair 15 spills: the subtraction stays two scalar instructions)
__device__ __forceinline__ f32x2 pk_sub(f32x2 a, f32x2 b) { return a - b; }
__device__ __forceinline__ f32x2 pk_add(f32x2 a, f32x2 b) { return a + b; }
__device__ __forceinline__ f32x2 pk_splat(float v) { return f32x2{v, v}; }
using bf16x8 = __attribute__((ext_vector_type(8))) __bf16;
using f16x8 = __attribute__((ext_vector_type(8))) _Float16;

template <int DT>
__device__ __forceinline__ f32x16 mfma32(i32x4 a, i32x4 b, f32x16 c) {
    if constexpr (DT == RN_DTYPE_BF16)
        return __builtin_amdgcn_mfma_f32_32x32x16_bf16(__builtin_bit_cast(bf16x8, a), __builtin_bit_cast(bf16x8, b),
                                                       c, 0, 0, 0);
    else
        return __builtin_amdgcn_mfma_f32_32x32x16_f16(__builtin_bit_cast(f16x8, a), __builtin_bit_cast(f16x8, b), c,
                                                      0, 0, 0);
}

template <int DT>
__device__ __forceinline__ unsigned short to16(float v) {
    if constexpr (DT == RN_DTYPE_BF16)
        return __builtin_bit_cast(unsigned short, static_cast<__bf16>(v));
    else
        return __builtin_bit_cast(unsigned short, static_cast<_Float16>(v));
}

template <int DT>
__device__ __forceinline__ float from16(unsigned short u) {
    if constexpr (DT == RN_DTYPE_BF16)
        return __uint_as_float(static_cast<unsigned>(u) << 16);
    else
        return static_cast<float>(__builtin_bit_cast(_Float16, u));
}

// two floats -> one dword of two 16-bit values, round-to-nearest-even (one v_cvt_pk_*)
template <int DT>
__device__ __forceinline__ unsigned pack2(float a, float b) {
    const f32x2 v = {a, b};
    if constexpr (DT == RN_DTYPE_BF16) {
        using bf16x2 = __attribute__((ext_vector_type(2))) __bf16;
        return __builtin_bit_cast(unsigned, __builtin_convertvector(v, bf16x2));
    } else {
        using f16x2 = __attribute__((ext_vector_type(2))) _Float16;
        return __builtin_bit_cast(unsigned, __builtin_convertvector(v, f16x2));
    }
}

// ---- dithered 16-bit stores (bf16 handles, round 6).  A bf16 store keeps 8 significand bits: on smooth image
// content every pixel of a region rounds the same way and the errors ADD through the 3 x 3 windows of the next convolution
// (the worst logit errors of the parity set were solid-colour and gradient images).  v_cvt_sr_bf16_f32 computes
// (bits + (seed >> 16)) >> 16 (tools/ubench/cvt_sr.hip: no mismatch in 2 M conversions); with a seed that depends on the
// OUTPUT ROW only -- 1/6, 3/6, 5/6 of an ulp for rows 0, 1, 2 mod 3 -- the three rows of every window carry three different
// rounding offsets and their errors cancel to a third of an ulp of the window's mean: deterministic, position-dependent
// rounding, each stored value within one ulp of the exact one (instead of half).  One instruction per value (RNE: one per pair).
constexpr unsigned RN_SEED_PLAIN = 0x80000000u;      // the un-dithered form of the same instruction: round half up
__host__ __device__ __forceinline__ unsigned rn_dither_seed(int row) {
    const int p = row % 3;
    return p == 0 ? 0x2AAA0000u : (p == 1 ? 0x80000000u : 0xD5550000u);
}
// two floats -> one dword of two bf16 values through v_cvt_sr_bf16_f32 with a wave-uniform seed.  (Inline asm: the builtin wants
// the old destination as an operand and the compiler zeroes it first; the inputs are VALU results -- the BN fma -- never raw MFMA
// results, so no hazard needs padding inside the string.)
__device__ __forceinline__ unsigned pack2_sr_bf16(float a, float b, unsigned seed) {
#if defined(__HIP_DEVICE_COMPILE__)
    unsigned r;
    asm("v_cvt_sr_bf16_f32 %0, %1, %3\n\tv_cvt_sr_bf16_f32 %0, %2, %3 op_sel:[0,0,1]" : "=&v"(r) : "v"(a), "v"(b), "s"(seed));
    return r;
#else
    return 0;
#endif
}
// the same conversion on the host (rn_create's proofs of constant channels, the tests)
inline unsigned short rn_sr_bf16_host(float v, unsigned seed) {
    unsigned u;
    __builtin_memcpy(&u, &v, 4);
    return static_cast<unsigned short>((u + (seed >> 16)) >> 16);
}

// ReLU6 of two accumulators of a stage whose conv weights are stored DIVIDED BY 6 (rn_fused_prepare: `sixth`), as one
// instruction: relu6(6 x) / 6 = clamp(x, 0, 1), and the [0, 1] clamp is the free output modifier of the fp16 conversion
// (LLVM folds the packed min / max into `v_cvt_pk_f16_f32 ... clamp`).  The pooled sums stay scaled by 1/6; the folded BN
// scale of such a stage carries the 6.
__device__ __forceinline__ unsigned pack2_relu6_sixth(float a, float b) {
    using f16x2 = __attribute__((ext_vector_type(2))) _Float16;
    const f32x2 v = {a, b};
    f16x2 h = __builtin_convertvector(v, f16x2);
    const f16x2 z = {static_cast<_Float16>(0.f), static_cast<_Float16>(0.f)}, o = {static_cast<_Float16>(1.f), static_cast<_Float16>(1.f)};
    h = __builtin_elementwise_min(__builtin_elementwise_max(h, z), o);
    return __builtin_bit_cast(unsigned, h);
}

template <int DT>
__device__ __forceinline__ uint2 pack4(float a, float b, float c, float d) {
    uint2 r;
    r.x = pack2<DT>(a, b);
    r.y = pack2<DT>(c, d);
    return r;
}

// workgroup barrier that orders LDS traffic only: unlike __syncthreads() it does not drain
// outstanding global loads/stores (vmcnt) -- prefetches and output stores stay in flight
__device__ __forceinline__ void lds_barrier() {
    __builtin_amdgcn_fence(__ATOMIC_RELEASE, "workgroup", "local");
    __builtin_amdgcn_s_barrier();
    __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "workgroup", "local");
}

template <int DT>
__device__ __forceinline__ f32x4 unpack4(uint2 v) {
    f32x4 r;
    r[0] = from16<DT>(static_cast<unsigned short>(v.x & 0xffff));
    r[1] = from16<DT>(static_cast<unsigned short>(v.x >> 16));
    r[2] = from16<DT>(static_cast<unsigned short>(v.y & 0xffff));
    r[3] = from16<DT>(static_cast<unsigned short>(v.y >> 16));
    return r;
}

// value of lane+1 (DPP wave shift left by one; lane 63 reads 0)
__device__ __forceinline__ float lane_next(float v) {
    // bound_ctrl = 1: lanes without a source read 0, no "old" value to materialise, and the
    // shift folds into the consuming v_add_f32 as a DPP modifier
    return __int_as_float(__builtin_amdgcn_update_dpp(0, __float_as_int(v), 0x130, 0xf, 0xf, true));
}

// value of lane+N inside the 16-lane DPP row (row_shl:N; lanes without a source read 0)
template <int N>
__device__ __forceinline__ float row_next(float v) {
    return __int_as_float(__builtin_amdgcn_update_dpp(0, __float_as_int(v), 0x100 + N, 0xf, 0xf, true));
}

// packed fp16 add of two dwords holding two halves each (one v_pk_add_f16)
__device__ __forceinline__ int pk_add_f16(int a, int b) {
    using h2 = __attribute__((ext_vector_type(2))) _Float16;
    return __builtin_bit_cast(int, __builtin_bit_cast(h2, a) + __builtin_bit_cast(h2, b));
}

// Residual interpolation weight as ONE 16-bit MFMA operand: the lerp fraction t is rounded to a multiple of 2^-8
// (bf16) / 2^-11 (fp16), for which t and 1 - t are both exactly representable -- the two weights still sum to
// exactly 1, the interpolation position moves by at most 2^-9 (2^-12) of a pixel.
template <int DT>
__host__ __device__ __forceinline__ float res_quant_lerp(float t) {
    constexpr float s = DT == RN_DTYPE_BF16 ? 256.0f : 2048.0f;
    return __builtin_rintf(t * s) / s;
}

// a * b rounded to float32 BEFORE anything else uses it.  TF-1.13's compute_interpolation_weights rounds
// in = out_index * scale and then takes lerp = in - floor(in); left to -ffp-contract=fast, hipcc fuses the product
// into the subtraction (v_fma_f32 lerp, scale, index, -floor), which moves the lerp weight by an ulp of `in`.
__device__ __forceinline__ float mul_rounded(float a, float b) {
    float p = a * b;
    asm volatile("" : "+v"(p));
    return p;
}

__device__ __forceinline__ float relu6f(float v) { return __builtin_amdgcn_fmed3f(v, 0.f, 6.f); }

// ---- stage-0 operands (stage0_kernel in rn_fused.hip and the stage-0 fusion of stage_rw_kernel use the same sequence)
// Stage 0 is the one stage whose input is EXACT in 16 bits: the uint8 pixel values themselves.  The pre-processing
// ((x / 255.) * 2) - 1 of network.py:129 is folded into the weights and a per-cout constant,
//     conv(x') = sum w (2 x / 255 - 1) = sum (2 w / 255) x - sum w,
// the B operand is fp16(x) (an integer <= 255: exact), and each folded weight enters the MFMA as a PAIR of fp16 numbers
// hi + lo (hi = fp16(v), lo = fp16(v - hi): 22 significand bits) in two cout rows of the A operand -- rows 8..31 of the
// 32 x 32 tile were idle (the stage has 8 couts), so the pair costs no matrix instruction: row c holds hi, row 8 + c lo,
// and the lane adds its two accumulator registers.  The constant - sum w rides in the idle fourth channel slot of the
// (ky = 0, kx = 0) pixel: that B element is 1.0, its A elements are the constant's hi / lo halves.  Everything is scaled
// by 2^8 (hi stays a normal fp16 number for |w| >= 1.5e-5; the power of two is exact and comes off in the BN scale),
// ReLU6 clamps at 6 x 2^8.
// With it the stage computes the fp32 convolution of the exact input up to the fp32 accumulation order; before, the
// fp16-rounded input levels and weights were the largest single error of the 16-bit path (tools/scratch/dbg_logit_err.py).
constexpr float S0_WSCALE = 256.0f;
__device__ __forceinline__ void s0_pixel_halves(unsigned bgr, int& d0, int& d1) {
    typedef _Float16 h2 __attribute__((ext_vector_type(2)));
    // 0x6400 | x is the fp16 number 1024 + x: bytes (R, 0x64, G, 0x64) and (B, 0x64, 0x00, 0x3c), then - 1024 from the pixel
    // halves: (R, G) and (B, 1.0)
    const unsigned m0 = __builtin_amdgcn_perm(0x3c006400u, bgr, 0x05010502u);
    const unsigned m1 = __builtin_amdgcn_perm(0x3c006400u, bgr, 0x07060500u);
    const h2 a = __builtin_bit_cast(h2, m0) - h2{static_cast<_Float16>(1024.f), static_cast<_Float16>(1024.f)};
    const h2 b = __builtin_bit_cast(h2, m1) - h2{static_cast<_Float16>(1024.f), static_cast<_Float16>(0.f)};
    d0 = __builtin_bit_cast(int, a);
    d1 = __builtin_bit_cast(int, b);
}
// conv value of cout 4 h + j of the lane's pixel (hi row + lo row: one packed add per two couts), ReLU6 in the scaled domain
__device__ __forceinline__ float s0_relu6(const f32x16& acc, int j) {
    const int p = j & ~1;
    const f32x2 sum = f32x2{acc[p], acc[p + 1]} + f32x2{acc[4 + p], acc[5 + p]};
    return __builtin_amdgcn_fmed3f(sum[j & 1], 0.f, 6.f * S0_WSCALE);
}

#ifdef RN_CLOCK
// Diagnostic build (-DRN_CLOCK, tools/build_clock.sh; never shipped): the in-kernel clock of a launch is
// delta(s_memtime) / delta(s_memrealtime) x 100 MHz (MI355X_MICROARCH.md, DVFS give-back item 6), stamped once at the entry
// and once at the exit of one wave per workgroup; the values go to a buffer of their own, nothing is computed from them.
__device__ __forceinline__ void clock_pair(unsigned long long& t, unsigned long long& r) {
    asm volatile("s_memtime %0\n\ts_memrealtime %1\n\ts_waitcnt lgkmcnt(0)" : "=s"(t), "=s"(r)::"memory");
}
#endif

// ------------------------------------------------------------------------ LDS-DMA / barrier helpers
template <int P>
using IC = std::integral_constant<int, P>;

typedef __attribute__((address_space(3))) void* lds_void_ptr;

// Bare workgroup barrier.  No fence: a fence would make the compiler drain the LDS-DMA queue
// (vmcnt(0)) at every barrier.  Correctness is by construction: DMA data is retired by the
// counted s_waitcnt vmcnt(N) in front of it, every ds_read of a step has been consumed by
// an MFMA / VALU instruction of that step, and there are no ds_writes in the loop.
__device__ __forceinline__ void raw_barrier() {
    asm volatile("" ::: "memory");
    __builtin_amdgcn_s_barrier();
    asm volatile("" ::: "memory");
}

// 16-byte-per-lane LDS-DMA piece: LDS destination = wave-uniform `lds` + lane * 16.
// (Kept in an explicit __device__ function: used directly inside a lambda the builtin makes
// the host pass drop the kernel's launch stub without a diagnostic.)
__device__ __forceinline__ void dma16(const void* gsrc, char* lds) {
    __builtin_amdgcn_global_load_lds(gsrc, (lds_void_ptr)lds, 16, 0, 0);
}

// The same piece with only the lanes of `mask` active (tail piece of a wave-private ring row).
// Done with an explicit EXEC window instead of `if (lane_ok)`: a divergent branch would split the
// row step into several basic blocks and the MFMA / epilogue interleave stops at block borders.
// Only called from wave-uniform code with all lanes active (EXEC is restored to all ones).
__device__ __forceinline__ void dma16_masked(const void* gsrc, char* lds, unsigned long long mask) {
    const unsigned lds_addr = static_cast<unsigned>(reinterpret_cast<uintptr_t>((__attribute__((address_space(3))) char*)lds));
    // M0 and EXEC are reserved registers: a clobber entry for them is not honoured, so the statement leaves both
    // exactly as it found them (M0 saved and restored, EXEC back to all ones -- every caller runs with all lanes on).
    unsigned m0_save;
    asm volatile(
        "s_mov_b32 %0, m0\n\t"
        "s_mov_b32 m0, %2\n\t"
        "s_mov_b64 exec, %3\n\t"
        "global_load_lds_dwordx4 %1, off\n\t"
        "s_mov_b64 exec, -1\n\t"
        "s_mov_b32 m0, %0"
        : "=&s"(m0_save)
        : "v"(gsrc), "s"(lds_addr), "s"(mask)
        : "memory");
}

template <int N>
__device__ __forceinline__ void wait_vmcnt() {
    static_assert(N >= 0 && N <= 63, "vmcnt range");
    asm volatile("s_waitcnt vmcnt(%0)" ::"n"(N) : "memory");
}

// ------------------------------------------------------------------------ MFMA stage
struct StageArgs {
    const unsigned short* in;     // [N, H, W, CIN]
    unsigned short* out;          // [N, Ho, Wo, COUT]
    const i32x4* wfrag;           // [KC][CT][64] fragments of 8 x 16-bit
    const float* bn_mean;
    const float* bn_inv;
    const float* bn_beta;
    const unsigned short* skip;   // [N, Ss, Ss, COUT] (residual stages)
    const float* bn2_mean;
    const float* bn2_inv;
    const float* bn2_beta;
    const int32_t* rlo;           // legacy bilinear tables, [Ho]
    const int32_t* rhi;
    const float* rlerp;
    int H, W;                     // input rows / cols
    int Ho, Wo;                   // output rows / cols
    int Ss;                       // skip side
    int rows_per_band, n_bands, n_colblocks, n_ctg, npt;
    const float* ptab;            // folded BN tables [4][COUT]: scale1, shift1, scale2, shift2 (rw kernels)
    int skipcols;                 // skip-row columns staged in LDS per workgroup (rw residual kernels)
    int dbg_flags;                // timing experiments only: bit 0 = skip output stores, bit 1 = skip MFMAs
    unsigned long long* stamp_buf; // diagnostic build (-DRN_STAMPS) only: per-wave phase cycle sums
    float rscale;                 // residual resize scale = float(Ss) / float(Ho), fp32 as TF computes it
    // stage-0 fusion (8-channel rw variant only; s0_bgr == nullptr: the stage reads `in` as usual)
    const uint8_t* s0_bgr;        // [N, S, S, 3] uint8 image batch
    const i32x4* s0_wfrag;        // [3 (ky)][64 lanes] stage-0 A fragments (fp16, K = (kx < 4, c < 4))
    const float* s0_ptab;         // [2][8] stage-0 folded BN: scale (inv / 9 / 2^8), shift
    int s0_S;                     // image side
    int s0_private;               // stage-0 fusion: keep the wave-private rings (round-2 form; A/B arm)
    // column blocks of the row-blocked kernels (rn_stage4x / 5x / 6x): workgroup = image x band x block; block b owns output
    // columns [cb_xo0[b], cb_xo0[b] + cb_wo[b]) and reads the input columns under them (+ halo).  n_cb == 1: whole rows.
    int n_cb;
    int cb_xo0[4], cb_wo[4];
    // rn_stage5x.hip: 16-cout quarters whose convolution runs (0 or 4: all).  2: the channels of quarters 2 and 3 have a FROZEN
    // first BN (rn_fused_prepare proves fma(H, sc1', sh1') == sh1' in float32 for every input and relabels the channels so);
    // their waves skip the convolution and its pooling and interleave with the live quarters' waves on the SIMDs.
    // rn_stage4x.hip: 3 = the channels of the last quarter are constants in the handle's 16-bit store, filled once at rn_create
    // (rn_fused_post_alloc) and not computed (twelve-wave form).
    int live_q;
    // rn_stage5x.hip (with live_q == 2): non-null = the input channels 48..63 are such constants; wfrag = rn_stage5x_pack48's
    // fragments, cstart[cout] = what the constants add to every conv output (the accumulators' start value), and the waves of the
    // last quarter -- whose output channels are constants too -- do nothing.
    const float* cstart;
    // the 16 constant output channels (positions 48..63) of a stage that does not compute them, as stored 16-bit values: the
    // kernels write them next to the computed channels so that every 128-byte pixel of the output leaves as a FULL line (lines
    // with a quarter missing are read-modify-write cycles of the ECC memory: measured 10-20 % on both launches, box by box)
    const unsigned short* cvals;
    // dithered stores (bf16 handles, pack2_sr_bf16): 1 = the stage's output rows are stored with rn_dither_seed(row); plain_q = the
    // cout quarter (rn_stage4x / 5x) whose channels are constants of the handle and are stored with RN_SEED_PLAIN, or -1
    int dither;
    int plain_q;
};

// Column-block plan of a row-blocked kernel: the fewest blocks (<= 4) of equal width +-1 whose widths lie in [wo_min, wo_max].
inline bool rn_colblock_plan(int out_side, int wo_min, int wo_max, int* n_cb, int* xo0, int* wo) {
    for (int nb = 1; nb <= 4; ++nb) {
        const int hi = (out_side + nb - 1) / nb, lo = out_side / nb;
        if (hi > wo_max || lo < wo_min) continue;
        int x = 0;
        for (int b = 0; b < nb; ++b) {
            wo[b] = lo + (b < out_side % nb ? 1 : 0);
            xo0[b] = x;
            x += wo[b];
        }
        *n_cb = nb;
        return true;
    }
    return false;
}

// launch arguments of the cross-stage fused kernel (rn_stage23.hip): the last two steps of a depth-3 conv_block
struct Stage23Args {
    const unsigned short* in;     // [N, W, W, 32]: the block's first BN output (input of the pair AND skip tensor)
    unsigned short* out;          // [N, Wo, Wo, 32]
    const i32x4* wfrag2;          // [18][64] fragments of the first conv of the pair
    const i32x4* wfrag3;          // [18][64] fragments of the second conv
    const float* ptab;            // [5][32] folded BN: scale, shift of the first stage | scale', shift', scale2 of the second
    const int32_t* rlo;           // legacy bilinear tables W -> Wo, [Wo]
    const int32_t* rhi;
    const float* rlerp;
    float rscale;                 // float(W) / float(Wo), fp32 as TF computes it
    int W, Wo;                    // side of the input / of the output (= W - 10)
    int rows_per_band, n_bands;
    // column blocks: a workgroup owns output columns [cb_x0[b], cb_x0[b] + cb_wo[b]) of its band and reads input
    // columns cb_x0[b] .. cb_x0[b] + cb_wo[b] + 9 (whole rows when n_cblocks == 1: the 224 x 224 network)
    int n_cblocks;
    int cb_x0[4], cb_wo[4];
    unsigned long long* stamp_buf; // diagnostic build (-DRN_STAMPS) only: per-wave cycle sums
    // rn_stage23x.hip only: 1 = the first stage computes the first half of every 8-cout group only (B-ring channels 8 j .. 8 j + 3);
    // the other 16 channels of B are FROZEN -- constants whatever the input (rn_fused_prepare proves it per channel for the
    // handle's dtype and orders the channels accordingly) -- and are written from the table, bit for bit what the full
    // computation would store.  2 = every channel is computed.
    int producer_halves;
    int narrow_b;                 // (with producer_halves == 1) 1: the B ring holds the 16 computed channels only; wfrag3 = rn_stage23x_pack_narrow's
                                  // 10 fragments, ptab row 5 = the second conv's constant of the frozen channels.  2 (round 6): the ring holds
                                  // EIGHT channels (24 constant ones), wfrag3 = rn_stage23x_pack_narrow8's 6 fragments
    int dither;                   // bf16: the pair's OUTPUT rows are stored with rn_dither_seed(row) (its on-chip tensor never is)
};

// 64 -> 128 stage without pooling on 16x16x32 matrix tiles (rn_conv16.hip)
struct Conv16Args {
    const unsigned short* in;     // [N, H, W, 64]
    unsigned short* out;          // [N, Ho, Wo, 128]
    const i32x4* wfrag;           // [18 chunks][8 cout tiles][64 lanes] A-operand fragments (rn_conv16_pack)
    const float* ptab;            // folded BN: scale[128], shift[128]
    int H, W, Ho, Wo;
    int rows_per_band, n_bands, n_colblocks;
};

template <int CIN>
struct StageGeom {
    static constexpr int CP = CIN / 8;                               // 16-byte chunks per pixel
    static constexpr int K = 9 * CIN;
    static constexpr int KC = (K + 15) / 16;                         // 16-deep K chunks
    static constexpr int PIX_PER_BANKROW = CP >= 16 ? 1 : 16 / CP;   // pixels per 256-byte LDS bank row
    static constexpr int LPT = (34 * CP + 63) / 64;                  // ring-row chunks a thread prefetches
};

constexpr int NSLOT = 4;   // LDS ring: 3 live input rows + 1 being filled

__host__ __device__ constexpr int tile_nout(int pk, int ps) { return pk ? (32 - pk) / ps + 1 : 32; }
__host__ __device__ constexpr int tile_stride(int pk, int ps) { return pk ? tile_nout(pk, ps) * ps : 32; }

// chunk swizzle: XOR the 16-byte chunk index inside a pixel with a function of the pixel
// column so that 16 consecutive pixels reading the same chunk index hit 16 distinct
// 16-byte slots of the 256-byte LDS bank row.
template <int CP>
__device__ __forceinline__ int chunk_swz(int pix) {
    if constexpr (CP == 1)
        return 0;
    else if constexpr (CP >= 16)
        return pix & 15;
    else
        return (pix / (16 / CP)) & (CP - 1);
}


}  // namespace rnk
