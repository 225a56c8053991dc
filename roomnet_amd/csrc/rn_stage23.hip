// Cross-stage fused kernel for a depth-3 conv_block's last two steps (reference network.py:183-203):
//
//   A (block's first BN output = the residual's skip tensor, [N, W, W, 32])
//     -> conv3x3 32->32 -> ReLU6 -> avg-pool 4/1 -> BN            = B   (never leaves the CU: LDS ring)
//     -> conv3x3 32->32 -> ReLU6 -> avg-pool 4/1 -> BN -> + legacy-bilinear(A) -> BN   = out [N, W-10, W-10, 32]
//
// One launch instead of two: A is read from HBM once (it feeds the first conv AND the residual -- the second use
// re-reads recent rows, served by L2 / Infinity Cache), B's write and re-read disappear.  Stage-boundary model:
// 14.25 MB per image for the two stages (bf16, 224 input); this kernel moves 5.65 MB.
//
// Geometry: a workgroup = 4 waves (one per SIMD, whole register file) owns one image x one band of output rows and
// ALL columns (W <= 215: whole rows of A and B fit the LDS rings).  Per step t one row of A arrives by LDS-DMA and
// every wave runs four "jobs": the first conv's row t-2 for its two 32-column tiles (reads the A ring), then the
// second conv's row t-8 for its two tiles (reads the B ring rows the workgroup finished in earlier steps).  The
// epilogue of each job (ReLU6, pooling on the matrix cores, BN; residual MFMAs for the second stage) is cut into
// micro-ops that are placed between the MFMAs of the NEXT job's chain; the first stage's epilogue writes its B-row
// segment into the B ring (ds_write_b64), the second stage's epilogue stores to HBM.  One s_barrier per step.
// The residual's skip rows are fetched again (L2) into a small wave-private ring: each wave stages only the 64 skip
// columns its own two tiles interpolate from, so nothing but the B ring is shared between waves.
//
// The arithmetic of both stages is instruction-for-instruction that of stage_rw_kernel's POOLM variants
// (rn_stage_rw.hip): results are bit-identical to the two-launch path (tests/test_hip_fused.py).
#include "rn_fused.h"
#include "rn_stage.h"

#include <atomic>
#include <utility>

using namespace rnk;

namespace {

constexpr int F_NA = 4, F_NB = 4, F_NSK = 3;     // ring depths: A rows, B rows, private skip rows
constexpr int F_WMAX = 215, F_WMIN = 193;        // supported widths of A (the tail DMA piece needs W > 192)
constexpr int F_ROWA = F_WMAX * 64;              // bytes per A ring row (32 channels x 16 bit per pixel)
constexpr int F_BDUMMY = F_WMAX - 5;             // B ring column that invalid lanes write to (never read)
constexpr int F_ROWB = (F_BDUMMY + 1) * 64;
constexpr int F_SKROW = 64 * 64;                 // one private skip row: 64 columns
constexpr int F_NTAB = 5 * 32;                   // folded BN tables: sc2, sh2 | sc3', sh3', sc4
constexpr int F_RINGA_OFF = 1024;
constexpr int F_RINGB_OFF = F_RINGA_OFF + F_NA * F_ROWA;
constexpr int F_SKIP_OFF = F_RINGB_OFF + F_NB * F_ROWB;
constexpr int F_LDS = F_SKIP_OFF + 4 * F_NSK * F_SKROW;
constexpr int F_LAG = 11;                        // step t finishes output row t - F_LAG
static_assert(F_LDS <= 160 * 1024, "LDS budget");
static_assert(F_RINGA_OFF % 64 == 0 && F_ROWA % 64 == 0 && F_ROWB % 64 == 0, "B-write addresses are composed with OR/XOR");
static_assert((F_NA - 1) * F_ROWA < 65536 && (F_NB - 1) * F_ROWB < 65536, "slot offsets are DS immediates");

__device__ __forceinline__ unsigned lds_addr(const void* p) {
    return static_cast<unsigned>(reinterpret_cast<uintptr_t>((__attribute__((address_space(3))) const char*)p));
}

__device__ __forceinline__ int swz4(int pix) { return (pix >> 2) & 3; }   // chunk_swz<4>

using i32x2 = __attribute__((ext_vector_type(2))) int;

template <int DT>
__global__ __launch_bounds__(256, 1) void stage23_kernel(const Stage23Args a) {
    constexpr int KC = 18, BAHEAD = 4;
    extern __shared__ __attribute__((aligned(64))) char smem[];
    const int tid = threadIdx.x;
    const int lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int r = lane & 31, hh = lane >> 5;
    const int band = blockIdx.x, n = blockIdx.y;
    const int W = a.W, Wb = W - 5, Wo = a.Wo;
    const int yo0 = band * a.rows_per_band;
    const int nrows = min(Wo, yo0 + a.rows_per_band) - yo0;
    const int nsteps = nrows + F_LAG;

    float* const tab = reinterpret_cast<float*>(smem);
    for (int i = tid; i < F_NTAB; i += 256) tab[i] = a.ptab[i];

    // ---- both weight sets -> registers (lane-linear fragment order, see rn_fused_prepare)
    i32x4 w2[KC], w3[KC];
#pragma unroll
    for (int kc = 0; kc < KC; ++kc) {
        w2[kc] = a.wfrag2[kc * 64 + lane];
        w3[kc] = a.wfrag3[kc * 64 + lane];
        // pin the weights to the accumulator half of the register file (MFMA operands may be AGPRs; the VALU side
        // of the epilogues needs the architectural VGPRs)
        asm volatile("" : "+a"(w2[kc]), "+a"(w3[kc]));
    }

    // ---- A-row DMA: the workgroup's 256 lanes cover the 4 W chunks of a row in 3 full pieces + one tail piece that
    // gives every wave W - 192 chunks (so every piece of every wave has active lanes).  Chunk q = (pixel q / 4, slot
    // q % 4) holds source chunk (q % 4) ^ swz(pixel): the XOR swizzle is applied on the source address.
    const char* const in_img = reinterpret_cast<const char*>(a.in + static_cast<int64_t>(n) * W * W * 32);
    const int64_t in_row_bytes = static_cast<int64_t>(W) * 64;
    const int tailn = W - 192;
    const unsigned long long tail_mask = (1ull << tailn) - 1ull;
    unsigned ld_goff[4];
#pragma unroll
    for (int i = 0; i < 4; ++i) {
        const int q = i < 3 ? tid + 256 * i : 768 + tailn * wave + min(lane, tailn - 1);
        const int p = q >> 2, c = q & 3;
        ld_goff[i] = static_cast<unsigned>(min(p, W - 1) * 64 + ((c ^ swz4(p)) << 4));
    }
    char* const ringA = smem + F_RINGA_OFF;
    char* const ringB = smem + F_RINGB_OFF;
    auto issue_A_piece = [&](auto II, const char* row, int slot) __attribute__((always_inline)) {
        constexpr int i = decltype(II)::value;
        if constexpr (i < 3)
            dma16(row + ld_goff[i], ringA + slot * F_ROWA + (i * 256 + wave * 64) * 16);
        else
            dma16_masked(row + ld_goff[3], ringA + slot * F_ROWA + (768 + tailn * wave) * 16, tail_mask);
    };
    auto a_row_ptr = [&](int j) __attribute__((always_inline)) {       // local A row j (clamped to the band's last row)
        return in_img + static_cast<int64_t>(yo0 + min(j, nrows + 9)) * in_row_bytes;
    };

    // ---- private skip ring: the 64 columns of A this wave's two second-stage tiles interpolate from
    const int xs0 = a.rlo[min(58 * wave, Wo - 1)];
    unsigned sk_goff[4];
#pragma unroll
    for (int i = 0; i < 4; ++i) {
        const int q = lane + 64 * i;
        const int p = q >> 2, c = q & 3;
        sk_goff[i] = static_cast<unsigned>(min(xs0 + p, W - 1) * 64 + ((c ^ swz4(p)) << 4));
    }
    char* const skw = smem + F_SKIP_OFF + wave * (F_NSK * F_SKROW);
    auto issue_skip_row = [&](int y, int slot) __attribute__((always_inline)) {
        const char* row = in_img + static_cast<int64_t>(y) * in_row_bytes;
#pragma unroll
        for (int i = 0; i < 4; ++i) dma16(row + sk_goff[i], skw + slot * F_SKROW + i * 1024);
    };
    auto ylo_of = [&](int yo) __attribute__((always_inline)) {
        // TF-1.13 compute_interpolation_weights: src = yo * scale (fp32), lo = int(src)
        return static_cast<int>(mul_rounded(static_cast<float>(yo), a.rscale));
    };
    const int ylo_base = ylo_of(yo0);
    int sk_fetched = ylo_base - 1;                    // highest skip row whose DMA has been issued

    // ---- lane constants of this wave's two column tiles (tile gT covers conv columns 29 gT .. 29 gT + 31)
    unsigned baseA[2][3][2], baseB[2][3][2];          // LDS address of the (kx, channel-pair) fragment in ring slot 0
    unsigned wbB[2];                                  // B-ring write address of this lane's pixel (slot 0, chunk 0)
    int voff[2];                                      // byte offset of this lane's first 16-byte output chunk
    int a_off[2][4];                                  // transposed skip reads (residual), relative to a skip slot
    i32x4 bw[2][2];                                   // interpolation matrix (one 16-bit operand, see res_quant_lerp)
    constexpr int OOB = 0x40000000;
    const unsigned ringA_lds = lds_addr(ringA), ringB_lds = lds_addr(ringB), skw_lds = lds_addr(skw);
#pragma unroll
    for (int T = 0; T < 2; ++T) {
        const int x_t = 29 * (2 * wave + T);
#pragma unroll
        for (int kx = 0; kx < 3; ++kx)
#pragma unroll
            for (int cc = 0; cc < 2; ++cc) {
                const int ca = min(x_t + r + kx, W - 1), cb = min(x_t + r + kx, Wb - 1);
                baseA[T][kx][cc] = ringA_lds + static_cast<unsigned>(ca * 64 + (((cc * 2 + hh) ^ swz4(ca)) << 4));
                baseB[T][kx][cc] = ringB_lds + static_cast<unsigned>(cb * 64 + (((cc * 2 + hh) ^ swz4(cb)) << 4));
            }
        const int xo = x_t + r;
        const int colw = (r < 29 && xo < Wb) ? xo : F_BDUMMY;
        wbB[T] = ringB_lds + static_cast<unsigned>(colw * 64 + (swz4(colw) << 4) + hh * 8);
        voff[T] = (r < 29 && xo < Wo) ? (xo * 32 + 8 * hh) * 2 : OOB;
        // residual: R[cout][x_out] = Skip^T[cout][x_in] * Wx[x_in][x_out], K = 32 skip columns from the tile's first
        // source column (see rn_stage_rw.hip)
        const int xo_t0 = min(x_t, Wo - 1);
        const int xs_t = a.rlo[xo_t0] - xs0;
        const int xq = min(xo, Wo - 1);
        const int plo = a.rlo[xq] - xs0, phi = a.rhi[xq] - xs0;
        const float xl = a.rlerp[xq];
        const float xlq = res_quant_lerp<DT>(xl);
#pragma unroll
        for (int c = 0; c < 2; ++c) {
            unsigned short wh[8];
#pragma unroll
            for (int j = 0; j < 8; ++j) {
                const int xin = xs_t + 16 * c + 8 * hh + j;
                float w = 0.f;
                if (xin == plo) w += 1.0f - xlq;
                if (xin == phi) w += xlq;
                wh[j] = to16<DT>(w);
            }
#pragma unroll
            for (int d = 0; d < 4; ++d)
                bw[T][c][d] = static_cast<int>(static_cast<unsigned>(wh[2 * d]) | (static_cast<unsigned>(wh[2 * d + 1]) << 16));
        }
        const int grp = lane >> 4, li = lane & 15, q = li >> 2, pp = li & 3;
#pragma unroll
        for (int c = 0; c < 2; ++c)
#pragma unroll
            for (int t2 = 0; t2 < 2; ++t2) {
                const int pix = min(max(xs_t + 16 * c + 8 * (grp >> 1) + 4 * t2 + q, 0), 63);
                const int ch = 2 * (grp & 1) + (pp >> 1);
                a_off[T][2 * c + t2] = (pix * 4 + (ch ^ swz4(pix))) * 16 + (pp & 1) * 8;
            }
    }
    const int out_row_bytes = Wo * 64;
    const char* const out_img = reinterpret_cast<const char*>(a.out + static_cast<int64_t>(n) * Wo * Wo * 32);
    auto out_row_rsrc = [&](int yo) __attribute__((always_inline)) {
        return __builtin_amdgcn_make_buffer_rsrc(const_cast<char*>(out_img) + static_cast<int64_t>(yo) * out_row_bytes, 0,
                                                 out_row_bytes, 0x00020000);
    };

    // band matrix of the pool MFMA (see rn_stage_rw.hip, POOLM)
    i32x4 pmw[2];
#pragma unroll
    for (int c = 0; c < 2; ++c)
#pragma unroll
        for (int d = 0; d < 4; ++d) {
            unsigned w = 0;
#pragma unroll
            for (int e2 = 0; e2 < 2; ++e2) {
                const int j8 = 2 * d + e2;
                const int x = (j8 & 3) + 8 * (j8 >> 2) + 16 * c + 4 * hh;
                w |= ((x >= r && x < r + 4) ? 0x3C00u : 0u) << (16 * e2);      // fp16 1.0
            }
            pmw[c][d] = static_cast<int>(w);
        }

    // ---- pooling state of the four jobs (job = 2 * stage + tile)
    // (previous row's ReLU6 output as fp16 pairs + the pair-sum rows q_{j-1}, q_{j-2}: 24 registers per job)
    i32x4 hprev[4][2], qp0[4][2], qp1[4][2];
#pragma unroll
    for (int j = 0; j < 4; ++j)
        hprev[j][0] = hprev[j][1] = qp0[j][0] = qp0[j][1] = qp1[j][0] = qp1[j][1] = i32x4{0, 0, 0, 0};
    f32x16 acc[2];
#pragma unroll
    for (int g = 0; g < 16; ++g) acc[0][g] = acc[1][g] = 0.f;

    // ---- prologue: A row 0
    [&]<int... II>(std::integer_sequence<int, II...>) {
        (issue_A_piece(IC<II>{}, a_row_ptr(0), 0), ...);
    }(std::make_integer_sequence<int, 4>{});
    wait_vmcnt<0>();
    lds_barrier();

    // B fragment of K-chunk kc for job J in ring phase P (inline asm, BAHEAD chunks ahead, counted lgkmcnt)
    auto b_read = [&](auto JC, auto PC, auto KCC, float dep) __attribute__((always_inline)) -> i32x4 {
        constexpr int J = decltype(JC)::value, P = decltype(PC)::value, kc = decltype(KCC)::value;
        constexpr int S = J >> 1, T = J & 1;
        constexpr int tap = kc / 2, cc = kc % 2, ky = tap / 3, kx = tap % 3;
        // stage 1 of the pair reads A rows t-2 .. t (slot = row mod 4), stage 2 reads B rows t-8 .. t-6
        constexpr int slot = S == 0 ? (P + 2 + ky) % F_NA : (P + ky) % F_NB;
        constexpr int off = slot * (S == 0 ? F_ROWA : F_ROWB);
        const unsigned base = S == 0 ? baseA[T][kx][cc] : baseB[T][kx][cc];
        i32x4 v;
        asm volatile("ds_read_b128 %0, %1 offset:%2" : "=v"(v) : "v"(base), "n"(off), "v"(dep));
        return v;
    };
    auto mma_chain = [&](auto JC, auto PC, f32x16& accn, auto&& slotfn) __attribute__((always_inline)) {
        constexpr int J = decltype(JC)::value;
        const f32x16 zero = {0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f};
        i32x4 bq[KC];
        [&]<int... I>(std::integer_sequence<int, I...>) {
            ((bq[I] = b_read(JC, PC, IC<I>{}, 0.f)), ...);
        }(std::make_integer_sequence<int, BAHEAD>{});
        [&]<int... I>(std::integer_sequence<int, I...>) {
            (([&] {
                 if constexpr (I + BAHEAD < KC)
                     bq[I + BAHEAD] = b_read(JC, PC, IC<(I + BAHEAD < KC ? I + BAHEAD : 0)>{}, I == 0 ? 0.f : accn[0]);
                 constexpr int newer = (KC - 1 - I) < BAHEAD ? (KC - 1 - I) : BAHEAD;
                 asm volatile("s_waitcnt lgkmcnt(%1)" : "+v"(bq[I]) : "n"(newer));
                 if constexpr (J < 2)
                     accn = mfma32<DT>(bq[I], w2[I], I == 0 ? zero : accn);      // D'[pixel][cout]
                 else
                     accn = mfma32<DT>(bq[I], w3[I], I == 0 ? zero : accn);
                 slotfn(IC<I>{});
             }()),
             ...);
        }(std::make_integer_sequence<int, KC>{});
    };

    // ---- epilogue state (one epilogue is in flight at a time: that of the previous job)
    using TQ = i32x2[8];
    TQ tq;
    f32x16 H, r_lo, r_hi;
    i32x4 qp[2];
    uint2 pk[4];
    float yv[16];
    struct RowCtx {                 // per-step scalars of the second stage's epilogue
        float yl;
        unsigned sk_lo, sk_hi;      // LDS addresses of the two staged skip rows
        __amdgpu_buffer_rsrc_t rs;
        int emit_mask;              // 0: store, OOB: drop
    };
    auto tr_read = [&](unsigned addr) __attribute__((always_inline)) -> i32x2 {
        i32x2 v;
        asm volatile("ds_read_b64_tr_b16 %0, %1" : "=v"(v) : "v"(addr));
        return v;
    };
    // micro-op k of job J's epilogue; PR = parity of the job's conv row; DRAIN: no chain around it (explicit waits)
    constexpr int M_RES_RD = 0, M_FRONT = 1, NF = 8, M_RES_MM = M_FRONT + NF, M_POOL = M_RES_MM + 1, M_BN = M_POOL + 3,
                  M_STORE = M_BN + 8, NM = M_STORE + 1;
    auto mop = [&](auto JC, auto PRC, auto PWC, auto DRAINC, auto KK, const f32x16& acce, const RowCtx& cx)
                   __attribute__((always_inline)) {
        constexpr int J = decltype(JC)::value, PR = decltype(PRC)::value, PW = decltype(PWC)::value, k = decltype(KK)::value;
        constexpr bool DRAIN = decltype(DRAINC)::value != 0;
        constexpr int S = J >> 1, T = J & 1;
        constexpr bool RES = S == 1;
        if constexpr (k == M_RES_RD) {
            if constexpr (RES) {
#pragma unroll
                for (int qq = 0; qq < 4; ++qq) {
                    tq[qq] = tr_read(cx.sk_lo + static_cast<unsigned>(a_off[T][qq]));
                    tq[4 + qq] = tr_read(cx.sk_hi + static_cast<unsigned>(a_off[T][qq]));
                }
            }
        } else if constexpr (k >= M_FRONT && k < M_FRONT + NF) {
            constexpr int i2 = 2 * (k - M_FRONT);
            // ReLU6 -> fp16 pair -> vertical pair sum q_j = v_{j-1} + v_j (packed fp16 add)
            const int vp = static_cast<int>(pack2<RN_DTYPE_F16>(relu6f(acce[i2]), relu6f(acce[i2 + 1])));
            qp[i2 / 8][(i2 % 8) / 2] = pk_add_f16(hprev[J][i2 / 8][(i2 % 8) / 2], vp);
            hprev[J][i2 / 8][(i2 % 8) / 2] = vp;
        } else if constexpr (k == M_RES_MM) {
            if constexpr (RES) {
                if constexpr (DRAIN) asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
                asm volatile("" : "+v"(tq[0]), "+v"(tq[1]), "+v"(tq[2]), "+v"(tq[3]), "+v"(tq[4]), "+v"(tq[5]), "+v"(tq[6]), "+v"(tq[7]));
                const f32x16 zero = {0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f};
                const i32x4 al0 = {tq[0][0], tq[0][1], tq[1][0], tq[1][1]}, al1 = {tq[2][0], tq[2][1], tq[3][0], tq[3][1]};
                const i32x4 ah0 = {tq[4][0], tq[4][1], tq[5][0], tq[5][1]}, ah1 = {tq[6][0], tq[6][1], tq[7][0], tq[7][1]};
                r_lo = mfma32<DT>(al0, bw[T][0], zero);
                r_hi = mfma32<DT>(ah0, bw[T][0], zero);
                r_lo = mfma32<DT>(al1, bw[T][1], r_lo);
                r_hi = mfma32<DT>(ah1, bw[T][1], r_hi);
            }
        } else if constexpr (k == M_POOL) {
            const f32x16 zero = {0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f};
            i32x4(&qold)[2] = PR == 0 ? qp0[J] : qp1[J];
            H = mfma32<RN_DTYPE_F16>(qold[0], pmw[0], zero);
            H = mfma32<RN_DTYPE_F16>(qold[1], pmw[1], H);
            H = mfma32<RN_DTYPE_F16>(qp[0], pmw[0], H);
            H = mfma32<RN_DTYPE_F16>(qp[1], pmw[1], H);
            qold[0] = qp[0];
            qold[1] = qp[1];
        } else if constexpr (k >= M_BN && k < M_BN + 8) {
            constexpr int g = (k - M_BN) / 2, h2 = (k - M_BN) % 2;
            const float* pt_g = tab + (RES ? 64 : 0) + 4 * hh + 8 * g;
            const f32x4 sc1 = *reinterpret_cast<const f32x4*>(pt_g);
            const f32x4 sh1 = *reinterpret_cast<const f32x4*>(pt_g + 32);
            f32x4 sc2 = {0.f, 0.f, 0.f, 0.f};
            if constexpr (RES) sc2 = *reinterpret_cast<const f32x4*>(pt_g + 64);
#pragma unroll
            for (int jj = 2 * h2; jj < 2 * h2 + 2; ++jj) {
                float y = fmaf(H[4 * g + jj], sc1[jj], sh1[jj]);
                if constexpr (RES) {
                    const float lo = r_lo[4 * g + jj];
                    const float rs = lo + (r_hi[4 * g + jj] - lo) * cx.yl;
                    y = fmaf(rs, sc2[jj], y);
                }
                yv[4 * g + jj] = y;
            }
            if constexpr (h2 == 0)
                pk[g].x = pack2<DT>(yv[4 * g], yv[4 * g + 1]);
            else
                pk[g].y = pack2<DT>(yv[4 * g + 2], yv[4 * g + 3]);
        } else if constexpr (k == M_STORE) {
            if constexpr (!RES) {
                // B-row segment -> ring slot of B row t-5: this lane's pixel, channels 8g + 4hh .. +3 per write
                constexpr int off = ((PW + 3) % F_NB) * F_ROWB;
#pragma unroll
                for (int g = 0; g < 4; ++g) {
                    const i32x2 d = {static_cast<int>(pk[g].x), static_cast<int>(pk[g].y)};
                    asm volatile("ds_write_b64 %0, %1 offset:%2" ::"v"(wbB[T] ^ static_cast<unsigned>(g << 4)), "v"(d), "n"(off) : "memory");
                }
            } else {
                i32x4 vv[2];
#pragma unroll
                for (int kk = 0; kk < 4; kk += 2) {
                    const auto sx = __builtin_amdgcn_permlane32_swap(pk[kk].x, pk[kk + 1].x, false, false);
                    const auto sy = __builtin_amdgcn_permlane32_swap(pk[kk].y, pk[kk + 1].y, false, false);
                    vv[kk / 2][0] = static_cast<int>(sx[0]);
                    vv[kk / 2][1] = static_cast<int>(sy[0]);
                    vv[kk / 2][2] = static_cast<int>(sx[1]);
                    vv[kk / 2][3] = static_cast<int>(sy[1]);
                }
                const int vo = voff[T] | cx.emit_mask;
                __builtin_amdgcn_raw_buffer_store_b128(vv[0], cx.rs, vo, 0, 0);
                __builtin_amdgcn_raw_buffer_store_b128(vv[1], cx.rs, vo + 32, 0, 0);
            }
        }
    };

    RowCtx cx_cur{}, cx_prev{};
    cx_cur.emit_mask = cx_prev.emit_mask = OOB;
    cx_cur.sk_lo = cx_cur.sk_hi = cx_prev.sk_lo = cx_prev.sk_hi = skw_lds;
    cx_cur.rs = cx_prev.rs = out_row_rsrc(yo0);

    // One step (ring phase P = t mod 4).
    auto step = [&](auto PC, int t) __attribute__((always_inline)) {
        constexpr int P = decltype(PC)::value;
        constexpr int PR = P & 1;                     // parity of both stages' conv rows (t-2, t-8)
        // ---- scalars of this step's second-stage rows
        const int jo = t - F_LAG;
        const int yo = yo0 + min(max(jo, 0), nrows - 1);
        {
            const float src = mul_rounded(static_cast<float>(yo), a.rscale);
            const int ylo = static_cast<int>(src);
            const int yhi = min(ylo + 1, W - 1);
            cx_cur.yl = src - static_cast<float>(ylo);
            cx_cur.sk_lo = skw_lds + static_cast<unsigned>(((ylo - ylo_base) % F_NSK) * F_SKROW);
            cx_cur.sk_hi = skw_lds + static_cast<unsigned>(((yhi - ylo_base) % F_NSK) * F_SKROW);
            cx_cur.rs = out_row_rsrc(yo);
            cx_cur.emit_mask = jo >= 0 ? 0 : OOB;
        }
        const char* const a_next = a_row_ptr(t + 1);
        // chain of job J with the micro-ops of job JE's epilogue behind its MFMAs (JE = J - 1; job 0 carries the
        // deferred epilogue of the previous step's job 3, whose conv row has the other parity)
        auto run = [&](auto JC, auto JEC, auto PREC, auto PWEC, const RowCtx& cxe) __attribute__((always_inline)) {
            constexpr int J = decltype(JC)::value, JE = decltype(JEC)::value;
            f32x16& accn = acc[J & 1];
            const f32x16& acce = acc[JE & 1];
            auto slotfn = [&](auto II) __attribute__((always_inline)) {
                constexpr int I = decltype(II)::value;
                if constexpr (J == 0) {
                    // this step's A-row DMA (row t+1 -> slot (P+1) mod 4), one piece behind each of the first MFMAs
                    if constexpr (I < 4) issue_A_piece(IC<(I < 4 ? I : 0)>{}, a_next, (P + 1) % F_NA);
                }
                [&]<int... K>(std::integer_sequence<int, K...>) {
                    (([&] {
                         if constexpr (K * KC / NM == I) mop(JEC, PREC, PWEC, IC<0>{}, IC<K>{}, acce, cxe);
                     }()),
                     ...);
                }(std::make_integer_sequence<int, NM>{});
                __builtin_amdgcn_sched_barrier(0);
            };
            mma_chain(JC, PC, accn, slotfn);
        };
        run(IC<0>{}, IC<3>{}, IC<1 - PR>{}, IC<(P + 3) % 4>{}, cx_prev);
        // skip rows for the next step's residual: fetch while a slot is free (the deferred epilogue above was the
        // last reader of the previous step's pair)
        {
            const int yo_next = yo0 + min(max(jo + 1, 0), nrows - 1);
            const int need = min(ylo_of(yo_next) + 1, W - 1);
            const int lowest = ylo_of(yo);
#pragma unroll
            for (int it = 0; it < 2; ++it)
                if (sk_fetched < need && sk_fetched - 2 < lowest) {
                    ++sk_fetched;
                    issue_skip_row(sk_fetched, (sk_fetched - ylo_base) % F_NSK);
                }
        }
        run(IC<1>{}, IC<0>{}, IC<PR>{}, IC<P>{}, cx_cur);
        run(IC<2>{}, IC<1>{}, IC<PR>{}, IC<P>{}, cx_cur);
        // everything this wave's DMA engine was asked for in this step (A row t+1, skip rows) has landed before the
        // residual of this step reads the skip ring and before the barrier publishes the A row
        wait_vmcnt<0>();
        run(IC<3>{}, IC<2>{}, IC<PR>{}, IC<P>{}, cx_cur);
        cx_prev = cx_cur;
        asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
        raw_barrier();
    };

    int t = 0;
    for (; t + 3 < nsteps; t += 4) {
        step(IC<0>{}, t);
        step(IC<1>{}, t + 1);
        step(IC<2>{}, t + 2);
        step(IC<3>{}, t + 3);
    }
    const int rem = nsteps - t;     // 0..3 steps left, phases 0, 1, 2
    if (rem > 0) step(IC<0>{}, t);
    if (rem > 1) step(IC<1>{}, t + 1);
    if (rem > 2) step(IC<2>{}, t + 2);
    // drain: the epilogue of the last step's job 3 (its conv row has the parity of the last step)
    auto drain = [&](auto PRC) __attribute__((always_inline)) {
        [&]<int... K>(std::integer_sequence<int, K...>) {
            (mop(IC<3>{}, PRC, IC<0>{}, IC<1>{}, IC<K>{}, acc[1], cx_prev), ...);
        }(std::make_integer_sequence<int, NM>{});
    };
    if (((nsteps - 1) & 1) == 0)
        drain(IC<0>{});
    else
        drain(IC<1>{});
    wait_vmcnt<0>();
}

}  // namespace

bool rn_stage23_supported(int in_side) { return in_side >= F_WMIN && in_side <= F_WMAX; }

int rn_stage23_launch(int dtype, hipStream_t s, const Stage23Args& a, int n) {
    auto launch = [&](auto kern) -> int {
        static std::atomic<unsigned long long> attr_devices{0};     // per device and instantiation, see launch_rw
        int dev = 0;
        RN_HIP(hipGetDevice(&dev));
        if (!(attr_devices.load(std::memory_order_acquire) >> (dev & 63) & 1ull)) {
            RN_HIP(hipFuncSetAttribute(reinterpret_cast<const void*>(kern), hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024));
            attr_devices.fetch_or(1ull << (dev & 63), std::memory_order_release);
        }
        hipLaunchKernelGGL(kern, dim3(a.n_bands, n), dim3(256), F_LDS, s, a);
        RN_CHECK_LAUNCH();
        return RN_OK;
    };
    if (dtype == RN_DTYPE_BF16) return launch(stage23_kernel<RN_DTYPE_BF16>);
    return launch(stage23_kernel<RN_DTYPE_F16>);
}
