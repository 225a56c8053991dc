// Device helpers and launch-argument block shared by the fused stage kernels.
#pragma once
#include "rn_internal.h"

namespace rnk {

using i32x4 = __attribute__((ext_vector_type(4))) int;
using f32x16 = __attribute__((ext_vector_type(16))) float;
using f32x4 = __attribute__((ext_vector_type(4))) float;
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
    using f32x2 = __attribute__((ext_vector_type(2))) float;
    const f32x2 v = {a, b};
    if constexpr (DT == RN_DTYPE_BF16) {
        using bf16x2 = __attribute__((ext_vector_type(2))) __bf16;
        return __builtin_bit_cast(unsigned, __builtin_convertvector(v, bf16x2));
    } else {
        using f16x2 = __attribute__((ext_vector_type(2))) _Float16;
        return __builtin_bit_cast(unsigned, __builtin_convertvector(v, f16x2));
    }
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

__device__ __forceinline__ float relu6f(float v) { return __builtin_amdgcn_fmed3f(v, 0.f, 6.f); }

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
