// Stages 0 + 1 in one launch on 16x16x32 matrix tiles, two rows per step (round 5; the successor of the S0F / S0SH forms of
// rn_stage_rw.hip for this pair of stages).  Reference: network.py:226 (conv_block(8, pool 3/1)) feeding the first step of
// network.py:227 (conv_block(32, pool 4/1, depth 3)); the input scaling of network.py:129 / :153 is folded into stage 0.
//
//   uint8 BGR image [N, S, S, 3]
//     -> conv3x3 3->8 (on the exact byte values) -> ReLU6 -> avg-pool 3/1 -> BN      = s0.bn (never leaves the CU: LDS ring)
//     -> conv3x3 8->32 -> ReLU6 -> avg-pool 4/1 -> BN                                  = out [N, S-9, S-9, 32]
//
// One workgroup = image x band of output rows x column block (<= 216 output columns: whole rows at 224, two blocks at 420,
// three at 600); 8 waves, two per SIMD, every wave runs BOTH stages for two adjacent 16-pixel tiles each (stage 0: columns
// 30 w .., stage 1: 29 w ..; a run of two tiles yields 30 / 29 pooled columns, the pooling window that crosses the border
// between the two tiles takes its right-hand columns from the neighbour tile's registers through a second band-matrix MFMA).
// A step handles TWO rows of every tensor; one workgroup barrier per step.
//
// Stage 0 (always fp16 operands: they are exact).  K = 32 of one MFMA = two image rows x (4 pixels x [R, G, B, 1]): lane
// (pixel m, g) of the A operand holds pixels m + 2 (g & 1) + {0, 1} of image row r + (g >> 1), taken from ONE unaligned
// 8-byte load and turned into fp16 numbers by four v_perm_b32 (byte x -> 0x3800 | x = 0.5 + x / 2048, exact; the affine map
// back to x lives in the weights and the per-cout constant).  The operand X(r) of rows (r, r + 1), r even, is built once and
// feeds four MFMAs: conv row r - 2 = X(r-2) [W0|W1] + X(r) [W2|0], conv row r - 1 = X(r-2) [0|W0] + X(r) [W1|W2].  The 16
// columns of the B operand are the 8 couts as fp16 hi + lo pairs (22 significand bits), summed by one DPP add per register;
// everything is scaled so that the accumulator is conv / 6: ReLU6 + fp16 pack is the [0, 1] clamp of v_cvt_pk_f16_f32.
// The 3 x 3 average runs on the matrix cores like every other pooling of the 16-bit path: vertical sums as packed fp16
// pair sums, horizontal window as an MFMA against a 0/1 band matrix whose A operand IS the packed accumulator.
// (Round 4 pooled this stage in fp32 on the VALU: the fp16 rounding of the ReLU6 values, 2^-11 relative, now enters the
//  3 x 3 mean -- below the rounding of the stage's 16-bit output.)
// Stage 1 (operand type = the handle's).  A conv row reads its three s0.bn rows from the LDS ring: lane (pixel m, g) reads the
// 16 bytes (8 channels) of ring pixel m + g -- ONE ds_read_b128 is the whole K = 32 operand of a tap row (g = 3: zero weights),
// conflict-free.  Pooling 4/1, BN and the 16-byte stores (a lane ends up with 8 consecutive couts) as in rn_stage23x.hip.
//
// Per wave and 2-row step: 2 global loads, 12 VALU for the stage-0 operands, 56 MFMAs (8 + 8 stage 0, 24 + 16 stage 1), 8 ring
// reads, 4 ring writes, 4 stores.
#include "rn_fused.h"
#include "rn_stage.h"

#include <atomic>
#include <cmath>
#include <utility>

using namespace rnk;

namespace {

constexpr int Z_NS = 8;                        // ring slots: s0.bn row r lives in slot r & 7
constexpr int Z_RINGPX = 224;                  // ring columns (8 waves x 30, the last run cut at 224)
constexpr int Z_ROW = Z_RINGPX * 16;           // bytes per ring row: 8 channels x 16 bit per pixel
constexpr int Z_DUMP_OFF = Z_NS * Z_ROW;       // where masked lanes write (their slot offset is added too: Z_NS rows of slack)
constexpr int Z_LDS = Z_DUMP_OFF + 8 * 512 + (Z_NS - 1) * Z_ROW + 1024;
constexpr int Z_WOMAX = 216;                   // output columns per block: 7 x 29 + 13
constexpr int OOB = 0x40000000;
#ifndef RN_S01_OUT_AUX
#define RN_S01_OUT_AUX 0
#endif
static_assert(Z_LDS <= 160 * 1024, "LDS budget");

template <int DT>
__device__ __forceinline__ f32x4 mfma16(i32x4 a, i32x4 b, f32x4 c) {
    if constexpr (DT == RN_DTYPE_BF16)
        return __builtin_amdgcn_mfma_f32_16x16x32_bf16(__builtin_bit_cast(bf16x8, a), __builtin_bit_cast(bf16x8, b), c, 0, 0, 0);
    else
        return __builtin_amdgcn_mfma_f32_16x16x32_f16(__builtin_bit_cast(f16x8, a), __builtin_bit_cast(f16x8, b), c, 0, 0, 0);
}

using i32x2 = __attribute__((ext_vector_type(2))) int;

// sum of the hi and lo columns of a stage-0 accumulator register: column n and n + 8 of the 16-lane row (row_ror:8)
__device__ __forceinline__ float hilo(float v) {
    return v + __int_as_float(__builtin_amdgcn_update_dpp(0, __float_as_int(v), 0x128, 0xf, 0xf, true));
}

// RT: the encoding of the s0.bn tensor in the LDS ring and of stage 1's conv operands -- an on-chip intermediate, not a stored
// tensor.  fp16 also on bf16 handles: three more significand bits than the bf16 the stage kernels would write to HBM, and the
// fp16 matrix instruction draws less power on these tensors than the bf16 one (NOTES.md round 4: +4.5 % for the whole pass).
#ifndef RN_S01_RING_DT
#define RN_S01_RING_DT RN_DTYPE_F16
#endif
template <int DT>
__global__ __launch_bounds__(512, 2) void stage01x_kernel(const Stage01Args a) {
    constexpr int RT = RN_S01_RING_DT == 0 ? DT : RN_S01_RING_DT;
    extern __shared__ __attribute__((aligned(64))) char smem[];
    const int tid = threadIdx.x;
    const int lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int m = lane & 15, g = lane >> 4;
    const int cb = blockIdx.x % a.n_cb, band = blockIdx.x / a.n_cb, n = blockIdx.y;
    const int S = a.S, Wo_full = a.Wo;
    const int x0 = a.cb_xo0[cb], Wob = a.cb_wo[cb];
    const int yo0 = band * a.rows_per_band;
    const int nrows = min(Wo_full, yo0 + a.rows_per_band) - yo0;
    const int nsteps = (nrows + 10) / 2 + 1;          // step t finishes output rows 2 t - 11 and 2 t - 10
    const f32x4 zero4 = {0.f, 0.f, 0.f, 0.f};

    for (int i = tid; i < Z_LDS / 16; i += 512) *reinterpret_cast<i32x4*>(smem + i * 16) = i32x4{0, 0, 0, 0};

    // ---- band matrices of the pooling MFMAs: B operand lane (xo = m, g), K element 8 g + e = pixel 4 g + (e & 3) of the first
    // (e < 4) or second (e >= 4) vertical term; 1.0 where the window of output column xo covers the pixel; the x-forms are the
    // same for the NEXT tile's registers (pixel 16 + ...)
    i32x4 pm3, pmx3, pm4, pmx4;
#pragma unroll
    for (int d = 0; d < 4; ++d) {
        unsigned w3 = 0, wx3 = 0, w4 = 0, wx4 = 0;
#pragma unroll
        for (int e2 = 0; e2 < 2; ++e2) {
            const int e = 2 * d + e2;
            const int p = 4 * g + (e & 3);
            w3 |= ((p >= m && p < m + 3) ? 0x3C00u : 0u) << (16 * e2);
            wx3 |= ((16 + p >= m && 16 + p < m + 3) ? 0x3C00u : 0u) << (16 * e2);
            w4 |= ((p >= m && p < m + 4) ? 0x3C00u : 0u) << (16 * e2);
            wx4 |= ((16 + p >= m && 16 + p < m + 4) ? 0x3C00u : 0u) << (16 * e2);
        }
        pm3[d] = static_cast<int>(w3);
        pmx3[d] = static_cast<int>(wx3);
        pm4[d] = static_cast<int>(w4);
        pmx4[d] = static_cast<int>(wx4);
    }
    asm volatile("" : "+v"(pm3), "+v"(pmx3), "+v"(pm4), "+v"(pmx4));

    // ---- weights and folded BN
    i32x4 w0[4], w1[6];
#pragma unroll
    for (int v = 0; v < 4; ++v) w0[v] = a.w0frag[v * 64 + lane];
#pragma unroll
    for (int f = 0; f < 6; ++f) w1[f] = a.w1frag[f * 64 + lane];
    const f32x4 sc0 = *reinterpret_cast<const f32x4*>(a.ptab0 + 4 * (g & 1));
    const f32x4 sh0 = *reinterpret_cast<const f32x4*>(a.ptab0 + 8 + 4 * (g & 1));
    f32x4 sc1[2], sh1[2];
#pragma unroll
    for (int h = 0; h < 2; ++h) {
        sc1[h] = *reinterpret_cast<const f32x4*>(a.ptab1 + 8 * g + 4 * h);
        sh1[h] = *reinterpret_cast<const f32x4*>(a.ptab1 + 32 + 8 * g + 4 * h);
    }

    // ---- stage 0: image bytes.  Lane (m, g) of tile k: pixels q, q + 1 (q = x0 + 30 w + 16 k + m + 2 (g & 1)) of image row
    // yo0 + 2 t + (g >> 1): 8 bytes from byte 3 q of the row -- pulled back where they would cross the end of the row (the last
    // row of the last image ends the buffer), the value shifted down instead
    const uint8_t* const img = a.bgr + static_cast<int64_t>(n) * S * S * 3;
    const int rowb = 3 * S;
    int boff[2];
    unsigned bsh[2];
#pragma unroll
    for (int k = 0; k < 2; ++k) {
        const int q = min(x0 + 30 * wave + 16 * k + m + 2 * (g & 1), S - 1);
        const int bo = 3 * q, bo2 = min(bo, rowb - 8);
        boff[k] = bo2;
        bsh[k] = static_cast<unsigned>(8 * (bo - bo2));
    }
    auto load64 = [&](int t, int k) __attribute__((always_inline)) -> unsigned long long {
        const int y = min(yo0 + 2 * t + (g >> 1), S - 1);
        unsigned long long v;
        __builtin_memcpy(&v, img + static_cast<int64_t>(y) * rowb + boff[k], 8);
        return v;
    };
    // s0.bn ring: write address of this lane's 4 couts of output column xo = m of tile k (masked lanes -> the dump rows)
    int wa0[2], rb1[2], voff[2];
#pragma unroll
    for (int k = 0; k < 2; ++k) {
        const int col0 = 30 * wave + 16 * k + m;
        const bool ok0 = g < 2 && (k == 0 || m < 14) && col0 < Z_RINGPX;
        wa0[k] = ok0 ? col0 * 16 + 8 * g : Z_DUMP_OFF + wave * 512 + (lane & 31) * 8 + k * 256;
        rb1[k] = (29 * wave + 16 * k + m + g) * 16;                         // stage 1: ring pixel m + g of the tile
        const int col1 = 29 * wave + 16 * k + m;
        const bool ok1 = (k == 0 || m < 13) && col1 < Wob;
        voff[k] = ok1 ? ((x0 + col1) * 32 + 8 * g) * 2 : OOB;
    }
    const int out_row_bytes = Wo_full * 64;
    const char* const out_img = reinterpret_cast<const char*>(a.out + static_cast<int64_t>(n) * Wo_full * Wo_full * 32);

    // ---- state carried from step to step
    unsigned long long pw[2][2];             // image bytes of the next two steps
    i32x4 Xp[2];                             // stage-0 operand of the previous step
    i32x2 V0h[2], S0h[2];                    // stage 0: packed ReLU6 row 2 t - 3, pair sum of rows 2 t - 4, 2 t - 3
    i32x2 V1h[2][2], T1h[2][2], S1h[2][2];   // stage 1: row 2 t - 9, pair sums (2 t - 11, 2 t - 10) and (2 t - 10, 2 t - 9)
#pragma unroll
    for (int k = 0; k < 2; ++k) {
        pw[k][0] = load64(0, k);
        pw[k][1] = load64(1, k);
        Xp[k] = i32x4{0, 0, 0, 0};
        V0h[k] = S0h[k] = i32x2{0, 0};
#pragma unroll
        for (int h = 0; h < 2; ++h) V1h[k][h] = T1h[k][h] = S1h[k][h] = i32x2{0, 0};
    }
#ifdef RN_S01_STAGGER          // experiment: de-phase the workgroups' store bursts (all images start together and run in lockstep)
    {
        const int ph = (blockIdx.y * 5 + blockIdx.x) & 7;
        for (int i = 0; i < ph; ++i) __builtin_amdgcn_s_sleep(8);
    }
#endif
    lds_barrier();

    auto packrow = [&](const f32x4& v) __attribute__((always_inline)) -> i32x2 {
        return i32x2{static_cast<int>(pack2_relu6_sixth(v[0], v[1])), static_cast<int>(pack2_relu6_sixth(v[2], v[3]))};
    };
    auto pk2 = [&](i32x2 x, i32x2 y) __attribute__((always_inline)) -> i32x2 { return i32x2{pk_add_f16(x[0], y[0]), pk_add_f16(x[1], y[1])}; };

    auto step = [&](auto PC, int t) __attribute__((always_inline)) {
        constexpr int P = decltype(PC)::value;           // t & 3: ring phase (two rows per step, eight slots)
        // ================================================================ stage 0: image rows 2 t, 2 t + 1
        i32x4 opa[2], opb[2];
#pragma unroll
        for (int k = 0; k < 2; ++k) {
            const unsigned long long v = pw[k][P & 1] >> bsh[k];
            pw[k][P & 1] = load64(t + 2, k);
            const unsigned lo = static_cast<unsigned>(v), hi = static_cast<unsigned>(v >> 32);
            const unsigned pb = __builtin_amdgcn_alignbyte(hi, lo, 3);               // bytes 3 .. 6: the second pixel
            // halves (R, G) and (B, 1.0) per pixel: byte x -> 0x38xx = 0.5 + x / 2048
            const i32x4 X = {static_cast<int>(__builtin_amdgcn_perm(0x3c003800u, lo, 0x05010502u)),
                             static_cast<int>(__builtin_amdgcn_perm(0x3c003800u, lo, 0x07060500u)),
                             static_cast<int>(__builtin_amdgcn_perm(0x3c003800u, pb, 0x05010502u)),
                             static_cast<int>(__builtin_amdgcn_perm(0x3c003800u, pb, 0x07060500u))};
            f32x4 ce = mfma16<RN_DTYPE_F16>(Xp[k], w0[0], zero4);                     // conv row 2 t - 2
            f32x4 co = mfma16<RN_DTYPE_F16>(Xp[k], w0[2], zero4);                     // conv row 2 t - 1
            ce = mfma16<RN_DTYPE_F16>(X, w0[1], ce);
            co = mfma16<RN_DTYPE_F16>(X, w0[3], co);
            Xp[k] = X;
            const f32x4 se = {hilo(ce[0]), hilo(ce[1]), hilo(ce[2]), hilo(ce[3])};
            const f32x4 so = {hilo(co[0]), hilo(co[1]), hilo(co[2]), hilo(co[3])};
            const i32x2 Ve = packrow(se), Vo = packrow(so);
            const i32x2 Sn = pk2(Ve, Vo);
            opa[k] = i32x4{S0h[k][0], S0h[k][1], Ve[0], Ve[1]};                       // s0.bn row 2 t - 4: conv rows 2t-4 .. 2t-2
            opb[k] = i32x4{V0h[k][0], V0h[k][1], Sn[0], Sn[1]};                       // s0.bn row 2 t - 3: conv rows 2t-3 .. 2t-1
            V0h[k] = Vo;
            S0h[k] = Sn;
        }
#pragma unroll
        for (int k = 0; k < 2; ++k) {
            f32x4 Ha = mfma16<RN_DTYPE_F16>(opa[k], pm3, zero4);
            f32x4 Hb = mfma16<RN_DTYPE_F16>(opb[k], pm3, zero4);
            if (k == 0) {
                Ha = mfma16<RN_DTYPE_F16>(opa[1], pmx3, Ha);
                Hb = mfma16<RN_DTYPE_F16>(opb[1], pmx3, Hb);
            }
            float ya[4], yb[4];
#pragma unroll
            for (int i = 0; i < 4; ++i) {
                ya[i] = __builtin_fmaf(Ha[i], sc0[i], sh0[i]);
                yb[i] = __builtin_fmaf(Hb[i], sc0[i], sh0[i]);
            }
            *reinterpret_cast<uint2*>(smem + wa0[k] + ((2 * P + 4) & 7) * Z_ROW) = pack4<RT>(ya[0], ya[1], ya[2], ya[3]);
            *reinterpret_cast<uint2*>(smem + wa0[k] + ((2 * P + 5) & 7) * Z_ROW) = pack4<RT>(yb[0], yb[1], yb[2], yb[3]);
        }
        // ================================================================ stage 1: conv rows 2 t - 8, 2 t - 7 from ring rows
        // 2 t - 8 .. 2 t - 5 (published by the barriers of the previous steps)
        i32x4 op1[2][2], op2[2][2];
#pragma unroll
        for (int k = 0; k < 2; ++k) {
            i32x4 fr[4];
#pragma unroll
            for (int j = 0; j < 4; ++j) fr[j] = *reinterpret_cast<const i32x4*>(smem + rb1[k] + ((2 * P + j) & 7) * Z_ROW);
#pragma unroll
            for (int h = 0; h < 2; ++h) {
                f32x4 ca = mfma16<RT>(fr[0], w1[h], zero4);
                f32x4 cb2 = mfma16<RT>(fr[1], w1[h], zero4);
                ca = mfma16<RT>(fr[1], w1[2 + h], ca);
                cb2 = mfma16<RT>(fr[2], w1[2 + h], cb2);
                ca = mfma16<RT>(fr[2], w1[4 + h], ca);
                cb2 = mfma16<RT>(fr[3], w1[4 + h], cb2);
                const i32x2 Va = packrow(ca), Vb = packrow(cb2);
                const i32x2 Tn = pk2(V1h[k][h], Va), Sn = pk2(Va, Vb);
                op1[k][h] = i32x4{T1h[k][h][0], T1h[k][h][1], Tn[0], Tn[1]};          // output row 2 t - 11
                op2[k][h] = i32x4{S1h[k][h][0], S1h[k][h][1], Sn[0], Sn[1]};          // output row 2 t - 10
                T1h[k][h] = Tn;
                S1h[k][h] = Sn;
                V1h[k][h] = Vb;
            }
        }
        const int o1 = 2 * t - 11, o2 = 2 * t - 10;
#ifdef RN_S01_NOSTORE          // timing experiment only (wrong results): what the launch costs without its output stream
        const int e1 = OOB, e2 = OOB;
#else
        const int e1 = (o1 >= 0 && o1 < nrows) ? 0 : OOB, e2 = (o2 >= 0 && o2 < nrows) ? 0 : OOB;
#endif
#ifdef RN_S01_SMALLOUT        // timing experiment only (wrong results): every row lands on the image's first two rows (cache-resident stream)
        const char* const r1p = out_img;
        const char* const r2p = out_img + out_row_bytes;
#else
        const char* const r1p = out_img + static_cast<int64_t>(yo0 + min(max(o1, 0), nrows - 1)) * out_row_bytes;
        const char* const r2p = out_img + static_cast<int64_t>(yo0 + min(max(o2, 0), nrows - 1)) * out_row_bytes;
#endif
        const __amdgpu_buffer_rsrc_t rs1 = __builtin_amdgcn_make_buffer_rsrc(const_cast<char*>(r1p), 0, out_row_bytes, 0x00020000);
        const __amdgpu_buffer_rsrc_t rs2 = __builtin_amdgcn_make_buffer_rsrc(const_cast<char*>(r2p), 0, out_row_bytes, 0x00020000);
#pragma unroll
        for (int k = 0; k < 2; ++k) {
            f32x4 H1[2], H2[2];
#pragma unroll
            for (int h = 0; h < 2; ++h) {
                H1[h] = mfma16<RN_DTYPE_F16>(op1[k][h], pm4, zero4);
                H2[h] = mfma16<RN_DTYPE_F16>(op2[k][h], pm4, zero4);
                if (k == 0) {
                    H1[h] = mfma16<RN_DTYPE_F16>(op1[1][h], pmx4, H1[h]);
                    H2[h] = mfma16<RN_DTYPE_F16>(op2[1][h], pmx4, H2[h]);
                }
            }
            float y1[8], y2[8];
#pragma unroll
            for (int h = 0; h < 2; ++h)
#pragma unroll
                for (int i = 0; i < 4; ++i) {
                    y1[4 * h + i] = __builtin_fmaf(H1[h][i], sc1[h][i], sh1[h][i]);
                    y2[4 * h + i] = __builtin_fmaf(H2[h][i], sc1[h][i], sh1[h][i]);
                }
            const i32x4 d1 = {static_cast<int>(pack2<DT>(y1[0], y1[1])), static_cast<int>(pack2<DT>(y1[2], y1[3])),
                              static_cast<int>(pack2<DT>(y1[4], y1[5])), static_cast<int>(pack2<DT>(y1[6], y1[7]))};
            const i32x4 d2 = {static_cast<int>(pack2<DT>(y2[0], y2[1])), static_cast<int>(pack2<DT>(y2[2], y2[3])),
                              static_cast<int>(pack2<DT>(y2[4], y2[5])), static_cast<int>(pack2<DT>(y2[6], y2[7]))};
            __builtin_amdgcn_raw_buffer_store_b128(d1, rs1, voff[k] | e1, 0, RN_S01_OUT_AUX);
            __builtin_amdgcn_raw_buffer_store_b128(d2, rs2, voff[k] | e2, 0, RN_S01_OUT_AUX);
        }
        lds_barrier();
    };
    int t = 0;
    for (; t + 3 < nsteps; t += 4) {
        step(IC<0>{}, t);
        step(IC<1>{}, t + 1);
        step(IC<2>{}, t + 2);
        step(IC<3>{}, t + 3);
    }
    const int rem = nsteps - t;
    if (rem > 0) step(IC<0>{}, t);
    if (rem > 1) step(IC<1>{}, t + 1);
    if (rem > 2) step(IC<2>{}, t + 2);
}

}  // namespace

bool rn_stage01x_supported(int cin0, int cout0, int pk0, int ps0, int cin1, int cout1, int pk1, int ps1, bool res1) {
    return cin0 == 3 && cout0 == 8 && pk0 == 3 && ps0 == 1 && cin1 == 8 && cout1 == 32 && pk1 == 4 && ps1 == 1 && !res1;
}

int rn_stage01x_ring_dtype(int dtype) { return RN_S01_RING_DT == 0 ? dtype : RN_S01_RING_DT; }

bool rn_stage01x_plan(int out_side, int* n_cb, int* xo0, int* wo) { return rn_colblock_plan(out_side, 1, Z_WOMAX, n_cb, xo0, wo); }

// Stage-0 B-operand fragments, four variants of [rows of the two image rows of an operand]: 0 = [W0 | W1], 1 = [W2 | 0],
// 2 = [0 | W0], 3 = [W1 | W2] (W_ky = kernel row ky).  frag[v][lane][j]: column n = lane % 16 (n < 8: fp16 hi part of cout n,
// n >= 8: lo part of cout n - 8), K element 8 g + j (g = lane / 16): image row g >> 1 of the operand, pixel kx = 2 (g & 1) +
// (j >> 2), channel c = j & 3 in the order R, G, B, one.  With the operand u = 0.5 + x / 2048 of a byte x:
//     conv(x') / 6 = sum w (2 x / 255 - 1) / 6 = sum (4096 w / 1530) u - sum w (2048 / 1530 + 1 / 6):
// the folded weight 4096 w / 1530 sits at (ky, kx < 3, c < 3), the constant in the `one` slot of (kx = 0) of the variant's FIRST
// non-empty kernel row -- once per conv row: variants 0 and 3.  Returns false when a folded value leaves the fp16 range.
bool rn_stage01x_pack0(const float* w_hwio, unsigned short (*cvt_f16)(float), float (*f16_f32)(unsigned short), std::vector<unsigned short>* out) {
    out->assign(static_cast<size_t>(4) * 64 * 8, 0);
    static const int kyof[4][2] = {{0, 1}, {2, -1}, {-1, 0}, {1, 2}};
    for (int v = 0; v < 4; ++v)
        for (int l = 0; l < 64; ++l)
            for (int j = 0; j < 8; ++j) {
                const int nn = l & 15, g = l >> 4, co = nn & 7, part = nn >> 3;
                const int ky = kyof[v][g >> 1], kx = 2 * (g & 1) + (j >> 2), c = j & 3;
                double val = 0.0;
                if (ky >= 0 && kx < 3 && c < 3) {
                    val = static_cast<double>(w_hwio[((ky * 3 + kx) * 3 + c) * 8 + co]) * (4096.0 / 1530.0);
                } else if (c == 3 && kx == 0 && (g >> 1) == 0 && (v == 0 || v == 3)) {
                    double sum = 0.0;
                    for (int tq = 0; tq < 27; ++tq) sum += static_cast<double>(w_hwio[tq * 8 + co]);
                    val = -sum * (2048.0 / 1530.0 + 1.0 / 6.0);
                } else {
                    continue;
                }
                if (!(std::fabs(val) < 65504.0)) return false;
                const unsigned short hi = cvt_f16(static_cast<float>(val));
                const unsigned short lo = cvt_f16(static_cast<float>(val - static_cast<double>(f16_f32(hi))));
                (*out)[(static_cast<size_t>(v) * 64 + l) * 8 + j] = part == 0 ? hi : lo;
            }
    return true;
}

// Stage-1 B-operand fragments: frag[ky][half][lane][j] = W[ky][kx = g][cin = j][cout(half, n)] / 6 for g = lane / 16 < 3 (g = 3: 0),
// cout(h, n) = 8 (n / 4) + 4 h + n % 4 as in rn_stage23x_pack (a lane's pooled rows of the two halves are 8 consecutive couts)
void rn_stage01x_pack1(const float* w_hwio, int dtype, unsigned short (*cvt_bf16)(float), unsigned short (*cvt_f16)(float),
                       std::vector<unsigned short>* out) {
    out->assign(static_cast<size_t>(6) * 64 * 8, 0);
    for (int ky = 0; ky < 3; ++ky)
        for (int h = 0; h < 2; ++h)
            for (int l = 0; l < 64; ++l)
                for (int j = 0; j < 8; ++j) {
                    const int nn = l & 15, g = l >> 4;
                    if (g >= 3) continue;
                    const int co = 8 * (nn >> 2) + 4 * h + (nn & 3);
                    const float v = w_hwio[((ky * 3 + g) * 8 + j) * 32 + co] / 6.0f;
                    (*out)[((static_cast<size_t>(ky) * 2 + h) * 64 + l) * 8 + j] = dtype == RN_DTYPE_BF16 ? cvt_bf16(v) : cvt_f16(v);
                }
}

int rn_stage01x_launch(int dtype, hipStream_t s, const Stage01Args& a, int n) {
    auto launch = [&](auto kern) -> int {
        static std::atomic<unsigned long long> attr_devices{0};     // per device and instantiation
        int dev = 0;
        RN_HIP(hipGetDevice(&dev));
        if (!(attr_devices.load(std::memory_order_acquire) >> (dev & 63) & 1ull)) {
            RN_HIP(hipFuncSetAttribute(reinterpret_cast<const void*>(kern), hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024));
            attr_devices.fetch_or(1ull << (dev & 63), std::memory_order_release);
        }
        hipLaunchKernelGGL(kern, dim3(a.n_bands * a.n_cb, n), dim3(512), Z_LDS, s, a);
        RN_CHECK_LAUNCH();
        return RN_OK;
    };
    if (dtype == RN_DTYPE_BF16) return launch(stage01x_kernel<RN_DTYPE_BF16>);
    return launch(stage01x_kernel<RN_DTYPE_F16>);
}
