// 32 -> 64 stage with avg-pool 4/2 (reference network.py:228, first step of conv_block(64, pool 4/2, depth 2)) on 16x16x32
// matrix tiles with ROW-REGISTER BLOCKING -- the design of rn_stage5x.hip without the residual:
//
//   in [N, W, W, 32] -> conv3x3 VALID -> ReLU6 -> avg-pool 4x4 stride 2 -> BN  = out [N, Wo, Wo, 64]
//
//  * a wave owns 16 couts (36 weight registers: 9 taps x 4) for 7 or 6 ADJACENT 16-pixel tiles; eight waves = 4 cout
//    quarters x 2 pixel halves, two per SIMD (7 + 6 tiles): 13 tiles cover the 203 conv columns (wide 32-pixel tiles: 224);
//  * every operand fragment (16 pixels x all 32 channels of one tap column of the NEWEST input row) feeds three
//    accumulators (kernel row 0 of conv row s, row 1 of s-1, row 2 of s-2): 3 LDS reads per tile and step for 9 MFMAs, three
//    independent chains; the partial accumulators of all tiles live across steps (84 registers); the ring only ever holds
//    the newest row and the rows in flight (4 slots);
//  * pooling (stride 2) on the matrix cores as in rn_stage5x.hip (fp16 pair sums = A operand, one accumulation per tile pair).
// The most matrix-dense launch of the pass held the lowest clock at the board's power cap (1.64 GHz, profiles/r3_clock.txt):
// the 16x16x32 shape costs less energy per FLOP (profiles/r2_power_cap.txt), and one workgroup per CU x 256 images is one
// round of the chip (the 2-wave workgroups of the row-streaming kernel ran 3.5 rounds).
// NQ = 3 (round 6): the channels of the last cout quarter are CONSTANTS on this handle (rn_fused_prepare proves per channel that the
// 16-bit store of fma(H, sc, sh) is one number for every H in [0, 16], relabels the channels so and fills them into the output
// tensor once, at rn_create): twelve waves = 3 live quarters x 4 pixel runs (4 + 3 + 3 + 3 tiles = 31 + 23 + 23 + 23 pooled
// columns), rotated over the SIMDs so that each carries 10 (one: 9) tile rows per step instead of 13.
// One workgroup = one image x one band of output rows x one COLUMN BLOCK of 95..101 pooled columns (194..206 input columns:
// the whole row of the 224 x 224 network, two blocks at 420, three at 600; rn_colblock_plan).  Blocks do not overlap in the
// output; the 4 halo columns of a block's input are read by both neighbours.
#include "rn_fused.h"
#include "rn_stage.h"

#include <atomic>
#include <utility>

using namespace rnk;

namespace {

constexpr int U_NS = 4;                           // ring slots: the newest row + 3 in flight
constexpr int U_AHEAD = 3;
constexpr int U_RINGPX = 208;                     // pixels per ring row (13 tiles - 2 overlap columns + halo)
constexpr int U_ROW = U_RINGPX * 64;              // bytes per ring row (32 channels x 16 bit per pixel)
constexpr int U_TAB_BYTES = 4 * 64 * 4;
constexpr int U_RING_OFF = U_TAB_BYTES;
constexpr int U_LDS = U_RING_OFF + U_NS * U_ROW;
constexpr int U_WMIN = 193, U_WMAX = 206;

__device__ __forceinline__ unsigned lds_addr(const void* p) {
    return static_cast<unsigned>(reinterpret_cast<uintptr_t>((__attribute__((address_space(3))) const char*)p));
}
__device__ __forceinline__ int swz4x(int pix) { return (pix >> 1) & 3; }   // 4 chunks per 64-byte pixel (rn_stage23x.hip)

using i32x2 = __attribute__((ext_vector_type(2))) int;

template <int DT>
__device__ __forceinline__ f32x4 mfma16(i32x4 a, i32x4 b, f32x4 c) {
    if constexpr (DT == RN_DTYPE_BF16)
        return __builtin_amdgcn_mfma_f32_16x16x32_bf16(__builtin_bit_cast(bf16x8, a), __builtin_bit_cast(bf16x8, b), c, 0, 0, 0);
    else
        return __builtin_amdgcn_mfma_f32_16x16x32_f16(__builtin_bit_cast(f16x8, a), __builtin_bit_cast(f16x8, b), c, 0, 0, 0);
}

template <int DT, int NQ>
__global__ __launch_bounds__(NQ == 4 ? 512 : 768, NQ == 4 ? 2 : 3) void stage4x_kernel(const StageArgs a) {
    constexpr int U_NW = NQ == 4 ? 8 : 12;            // waves
    constexpr int U_NTH = U_NW * 64;
    constexpr int U_NT = NQ == 4 ? 7 : 4;             // tiles of the longest run
    constexpr int U_NU = (U_NT + 1) / 2;              // tile pairs (output stores per odd step) of the longest run
#ifdef RN_CLOCK
    unsigned long long ck_t0, ck_r0;
    clock_pair(ck_t0, ck_r0);
#endif
    extern __shared__ __attribute__((aligned(64))) char smem[];
    const int tid = threadIdx.x;
    const int lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    // cout quarter / pixel run.  NQ = 4: waves w and w + 4 share a SIMD (7 + 6 tiles).  NQ = 3: waves w, w + 4, w + 8 share a
    // SIMD and take the runs w, w + 1, w + 2 (mod 4) of the quarters 0, 1, 2
    const int cq = NQ == 4 ? (wave & 3) : (wave >> 2);
    const int run = NQ == 4 ? (wave >> 2) : (((wave & 3) + (wave >> 2)) & 3);
    const int px16 = lane & 15, g = lane >> 4;
    const int cb = blockIdx.x % a.n_cb, band = blockIdx.x / a.n_cb, n = blockIdx.y;
    const int Win = a.W, Wo_full = a.Wo, Ho = a.Ho;
    const int xo0 = a.cb_xo0[cb], Wo = a.cb_wo[cb];       // this block's pooled columns
    const int x0 = 2 * xo0;                               // its first input column
    const int W = min(2 * Wo + 4, Win - x0);              // input columns it reads
    const int yo0 = band * a.rows_per_band;
    const int nrows = min(Ho, yo0 + a.rows_per_band) - yo0;
    const int y0 = 2 * yo0;
    const int nconv = 2 * (nrows - 1) + 4;
    const int nin = nconv + 2;
    const f32x4 zero4 = {0.f, 0.f, 0.f, 0.f};
    constexpr int OOB = 0x40000000;

    char* const ring = smem + U_RING_OFF;
    const unsigned ring_lds = lds_addr(ring);
    // pixels W .. 207 of every slot are never written by the row DMA: zero them once
    for (int i = tid; i < U_NS * (U_RINGPX - U_WMIN) * 4; i += U_NTH) {
        const int slot = i / ((U_RINGPX - U_WMIN) * 4), rest = i % ((U_RINGPX - U_WMIN) * 4);
        const int p = U_WMIN + rest / 4, c = rest % 4;
        if (p >= W) *reinterpret_cast<i32x4*>(ring + slot * U_ROW + p * 64 + c * 16) = i32x4{0, 0, 0, 0};
    }

    // ---- this wave's tiles.  NQ = 4: run 0 = 7 tiles from column 0 (55 pooled columns), run 1 = 6 tiles from column 110 (47
    // pooled columns from 55).  NQ = 3: run 0 = 4 tiles (31 pooled columns), runs 1..3 = 3 tiles (23 each, from 31 / 54 / 77).
    // A run of t tiles pools 8 t - 1 columns; the next run starts 2 x that many conv columns further (2 columns of overlap).
    const bool has7 = run == 0;                           // (the longest run: U_NT tiles; the others one fewer)
    const int xo_run = NQ == 4 ? (run ? 55 : 0) : (run ? 8 + 23 * run : 0);
    const int xw = 2 * xo_run;
    const int nout_run = NQ == 4 ? (has7 ? 55 : 47) : (has7 ? 31 : 23);

    // ---- weights: fragment f = ky * 3 + kx (all 32 channels of the tap), B operand of D'[pixel][cout]
    i32x4 wf[9];
#pragma unroll
    for (int f = 0; f < 9; ++f) {
        const i32x4* src = a.wfrag + (f * 4 + cq) * 64 + lane;
        asm volatile("global_load_dwordx4 %0, %1, off" : "=v"(wf[f]) : "v"(src) : "memory");
    }

    // ---- input rows by LDS-DMA: piece 0 = chunks tid (pixels 0..127); piece 1 = the remaining (W - 128) x 4 chunks, dealt
    // ceil(/8) to each wave, lane-masked (every wave keeps at least one active lane for W >= 193)
    const char* const in_img = reinterpret_cast<const char*>(a.in + static_cast<int64_t>(n) * Win * Win * 32) + x0 * 64;
    const int row_bytes = Win * 64;
    const unsigned goff0 = static_cast<unsigned>((tid >> 2) * 64 + (((tid & 3) ^ swz4x(tid >> 2)) << 4));
    // (a wave whose share of a short tail is empty loads the row's last chunk once more: the same bytes to the same place)
    const int tail_total = W * 4 - U_NTH;
    const int tailn = (tail_total + U_NW - 1) / U_NW;
    const int tail_start = min(wave * tailn, tail_total - 1);
    const int tail_cnt = max(1, min(tail_total - wave * tailn, tailn));
    const unsigned long long tail_mask = (1ull << tail_cnt) - 1ull;
    unsigned goff1;
    {
        const int q = U_NTH + tail_start + min(lane, tail_cnt - 1);
        const int p = q >> 2, c = q & 3;
        goff1 = static_cast<unsigned>(p * 64 + ((c ^ swz4x(p)) << 4));
    }
    auto issue_row = [&](int y, int slot) __attribute__((always_inline)) {
        const char* row = in_img + static_cast<int64_t>(y0 + min(y, nin - 1)) * row_bytes;
        unsigned o0 = goff0, o1 = goff1;
        asm volatile("" : "+v"(o0), "+v"(o1));
        dma16(row + o0, ring + slot * U_ROW + wave * 1024);
        dma16_masked(row + o1, ring + slot * U_ROW + (U_NTH + tail_start) * 16, tail_mask);
    };

    // ---- operand read bases (slot 0): tap column kx; tile k adds 16 pixels = 1024 bytes (same swizzle)
    unsigned base[3];
#pragma unroll
    for (int kx = 0; kx < 3; ++kx) {
        const int p = xw + px16 + kx;
        base[kx] = ring_lds + static_cast<unsigned>(p * 64 + ((g ^ swz4x(p)) << 4));
    }

    // ---- pooling band matrices (stride 2): see rn_stage5x.hip
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

    // ---- output stores: tile pair u = 16 pooled columns (pair 3 of the longer run: its 7th tile alone, 7 columns)
    int voff[U_NU];
#pragma unroll
    for (int u = 0; u < U_NU; ++u) {
        const int xo = xo_run + 16 * u + px16;
        const bool valid = 16 * u + px16 < nout_run && xo < Wo;
        voff[u] = valid ? ((xo0 + xo) * 64 + 16 * cq + 4 * g) * 2 : OOB;
    }
    // NQ = 3: the waves of quarter 2 also store the constant quarter behind theirs (lane: channels 48 + 4 g .. + 3): full lines
    [[maybe_unused]] i32x2 cq3v = {0, 0};
    if constexpr (NQ == 3) cq3v = *reinterpret_cast<const i32x2*>(a.cvals + 4 * g);
    // folded BN of the lane's 4 couts (16 cq + 4 g + i): y = S * sc + sh
    const f32x4 sc = *reinterpret_cast<const f32x4*>(a.ptab + 16 * cq + 4 * g);
    const f32x4 sh = *reinterpret_cast<const f32x4*>(a.ptab + 64 + 16 * cq + 4 * g);

    // ---- state
    f32x4 acc[3][U_NT];           // partial accumulators of conv rows s, s-1, s-2 (index = conv row mod 3), per tile
    int hp[U_NT][2], pp2[U_NT][2];
#pragma unroll
    for (int k = 0; k < U_NT; ++k) {
        hp[k][0] = hp[k][1] = pp2[k][0] = pp2[k][1] = 0;
#pragma unroll
        for (int r3 = 0; r3 < 3; ++r3) acc[r3][k] = zero4;
    }
    const int out_row_bytes = Wo_full * 128;
    const char* const out_img = reinterpret_cast<const char*>(a.out + static_cast<int64_t>(n) * Ho * Wo_full * 64);

#pragma unroll
    for (int j = 0; j < U_AHEAD; ++j) issue_row(j, j);
    wait_vmcnt<0>();
#pragma unroll
    for (int f = 0; f < 9; ++f) asm volatile("" : "+v"(wf[f]));
    lds_barrier();

    int slot_cur = 0;
    // operand fragments double-buffered across tiles: tile k + 1's three reads go out before tile k's MFMAs (a tile's
    // chain is only 9 MFMAs long: the LDS round trip at its head was a bubble as long as the chain itself) -- and across STEPS:
    // the first tile's fragments of row s + 1 are read at the end of step s, in front of the barrier (round 4; the barrier of a
    // step publishes the row AFTER the one the step computes).  Read behind the barrier they left both waves of every SIMD
    // waiting for LDS at the same moment, ~350 cycles of a ~2 950-cycle step with an idle matrix pipe.
    i32x4 fq[2][3];
    auto reads_at = [&](auto KC, const unsigned (&bcr)[3]) __attribute__((always_inline)) {
        constexpr int k = decltype(KC)::value;
        auto& dst = fq[k & 1];
#pragma unroll
        for (int f = 0; f < 3; ++f) asm volatile("ds_read_b128 %0, %1 offset:%2" : "=v"(dst[f]) : "v"(bcr[f]), "n"(k * 1024));
    };
    {
        // row 0 is published by the prologue's barrier only after every wave's pieces have landed: wait_vmcnt<0> above
        unsigned bc0[3] = {base[0], base[1], base[2]};
        reads_at(IC<0>{}, bc0);
    }
    auto step = [&](auto RC, auto PARC, int s) __attribute__((always_inline)) {
        constexpr int R = decltype(RC)::value, PAR = decltype(PARC)::value;
        constexpr int iN = R, iM = (R + 2) % 3, iO = (R + 1) % 3;      // accumulators of conv rows s, s-1, s-2
        // row s + 1 has landed (row s was published by the previous step's barrier).  VM_CNT counts the output stores too and
        // retires in order: a step issues two DMA pieces behind its first MFMAs and, on odd rows, 3 (or 4: has7) stores at its
        // end.  Younger than the DMA of row s + 1 (issued at step s - 2): even s: DMA + stores of step s - 1 = 2 + 3; odd s:
        // stores of step s - 2, DMA of step s - 1 = 3 + 2 (one more with four stores: those waves wait for one piece more).
        // (NQ = 3: two stores per odd step)
        wait_vmcnt<(NQ == 4 ? 5 : 4)>();
        raw_barrier();
        const unsigned so = static_cast<unsigned>(slot_cur * U_ROW);
        unsigned bc[3];
#pragma unroll
        for (int kx = 0; kx < 3; ++kx) bc[kx] = base[kx] + so;
        i32x4 op[U_NT + 1];
        op[U_NT - 1] = op[U_NT] = i32x4{0, 0, 0, 0};
        auto reads = [&](auto KC) __attribute__((always_inline)) { reads_at(KC, bc); };
        auto tile = [&](auto KC, auto NEXTC) __attribute__((always_inline)) {
            constexpr int k = decltype(KC)::value;
            constexpr bool NEXT = decltype(NEXTC)::value != 0;        // tile k + 1's reads are issued here (3 more in flight)
            if constexpr (NEXT) reads(IC<k + 1>{});
            auto& cur = fq[k & 1];
            [&]<int... F>(std::integer_sequence<int, F...>) {
                (([&] {
                     asm volatile("s_waitcnt lgkmcnt(%1)" : "+v"(cur[F]) : "n"((NEXT ? 3 : 0) + 2 - F));
                     acc[iN][k] = mfma16<DT>(cur[F], wf[0 * 3 + F], F == 0 ? zero4 : acc[iN][k]);
                     acc[iM][k] = mfma16<DT>(cur[F], wf[1 * 3 + F], acc[iM][k]);
                     acc[iO][k] = mfma16<DT>(cur[F], wf[2 * 3 + F], acc[iO][k]);
                     if constexpr (k == 0 && F == 0) {
                         // the row DMA rides behind the step's first MFMAs: into the slot of row s - 1 (everybody is past it)
                         int sl = slot_cur + U_AHEAD;
                         sl = sl >= U_NS ? sl - U_NS : sl;
                         issue_row(s + U_AHEAD, sl);
                     }
                 }()),
                 ...);
            }(std::make_integer_sequence<int, 3>{});
            // conv row j = s - 2 of this tile is complete: ReLU6 -> fp16 pairs; even rows wait in hp, odd rows form the pair
            // sum and the pooling operand [previous pair sum | this pair sum]
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
        [&]<int... K>(std::integer_sequence<int, K...>) {
            (tile(IC<K>{}, IC<(K + 2 < U_NT ? 1 : 0)>{}), ...);
        }(std::make_integer_sequence<int, U_NT - 1>{});
        if (has7) {                       // (the last tile's reads stay behind the branch: its tile code exists once)
            reads(IC<U_NT - 1>{});
            tile(IC<U_NT - 1>{}, IC<0>{});
        }
        if constexpr (PAR == 1) {
            // odd conv row j = s - 2 >= 3 completes pooled row r = (j - 3) / 2
            const int r = (s - 5) >> 1;
            const bool emit = s >= 5 && r < nrows;
            const int rr = min(max(r, 0), nrows - 1);
            const char* orow = out_img + static_cast<int64_t>(yo0 + rr) * out_row_bytes;
            const __amdgpu_buffer_rsrc_t rs = __builtin_amdgcn_make_buffer_rsrc(const_cast<char*>(orow), 0, out_row_bytes, 0x00020000);
            const int emask = emit ? 0 : OOB;
            // bf16 handles: the row's stores are dithered by the output row (rn_stage.h); a constant quarter keeps the plain seed
            [[maybe_unused]] const unsigned seed = (!a.dither || cq == a.plain_q) ? RN_SEED_PLAIN : rn_dither_seed(yo0 + rr);
            auto out = [&](auto UC) __attribute__((always_inline)) {
                constexpr int u = decltype(UC)::value;
                f32x4 H = mfma16<RN_DTYPE_F16>(op[2 * u], pmA, zero4);
                H = mfma16<RN_DTYPE_F16>(op[2 * u + 1], pmB, H);
                if constexpr (2 * u + 2 < U_NT) H = mfma16<RN_DTYPE_F16>(op[2 * u + 2], pmC, H);
                float y[4];
#pragma unroll
                for (int i = 0; i < 4; ++i) y[i] = __builtin_fmaf(H[i], sc[i], sh[i]);
                i32x2 d;
                if constexpr (DT == RN_DTYPE_BF16)
                    d = i32x2{static_cast<int>(pack2_sr_bf16(y[0], y[1], seed)), static_cast<int>(pack2_sr_bf16(y[2], y[3], seed))};
                else
                    d = i32x2{static_cast<int>(pack2<DT>(y[0], y[1])), static_cast<int>(pack2<DT>(y[2], y[3]))};
                __builtin_amdgcn_raw_buffer_store_b64(d, rs, voff[u] | emask, 0, 0);
                if constexpr (NQ == 3)
                    if (cq == 2) __builtin_amdgcn_raw_buffer_store_b64(cq3v, rs, (voff[u] + 32) | emask, 0, 0);      // (wave-uniform; OOB stays OOB)
            };
            if constexpr (NQ == 4) {
                out(IC<0>{});
                out(IC<1>{});
                out(IC<2>{});
                if (has7) out(IC<3>{});
            } else {
                out(IC<0>{});
                out(IC<1>{});         // (the shorter runs' second pair is their third tile alone: op[3] = 0)
            }
        }
        slot_cur = slot_cur == U_NS - 1 ? 0 : slot_cur + 1;
        asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
        {
            // first tile of the next row (published by this step's barrier)
            const unsigned sn = static_cast<unsigned>(slot_cur * U_ROW);
            unsigned bn[3];
#pragma unroll
            for (int kx = 0; kx < 3; ++kx) bn[kx] = base[kx] + sn;
            reads_at(IC<0>{}, bn);
        }
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

// pooled columns per block: 2 wo + 4 input columns must lie in U_WMIN + 1 .. U_WMAX (the tail DMA piece needs > 192)
bool rn_stage4x_plan(int out_side, int* n_cb, int* xo0, int* wo) {
    return rn_colblock_plan(out_side, (U_WMIN + 1 - 4 + 1) / 2, (U_WMAX - 4) / 2, n_cb, xo0, wo);
}

bool rn_stage4x_supported(int cin, int cout, int pool_k, int pool_s, bool res, int in_side) {
    int ncb, xo0[4], wo[4];
    return cin == 32 && cout == 64 && pool_k == 4 && pool_s == 2 && !res && in_side >= U_WMIN &&
           rn_stage4x_plan((in_side - 6) / 2 + 1, &ncb, xo0, wo);
}

// B-operand fragments: frag[f = ky * 3 + kx][cout quarter q][lane][j] = W[tap f][channel 8 (lane / 16) + j][cout 16 q + lane % 16]
void rn_stage4x_pack(const float* w_hwio, int dtype, unsigned short (*cvt_bf16)(float), unsigned short (*cvt_f16)(float),
                     std::vector<unsigned short>* out) {
    out->assign(static_cast<size_t>(9) * 4 * 64 * 8, 0);
    for (int f = 0; f < 9; ++f)
        for (int q = 0; q < 4; ++q)
            for (int l = 0; l < 64; ++l)
                for (int j = 0; j < 8; ++j) {
                    const int k = f * 32 + 8 * (l >> 4) + j, co = 16 * q + (l & 15);
                    const float v = w_hwio[static_cast<size_t>(k) * 64 + co];
                    (*out)[((static_cast<size_t>(f) * 4 + q) * 64 + l) * 8 + j] = dtype == RN_DTYPE_BF16 ? cvt_bf16(v) : cvt_f16(v);
                }
}

int rn_stage4x_launch(int dtype, hipStream_t s, const StageArgs& a, int n) {
    int nthreads = 512;
    auto launch = [&](auto kern) -> int {
        static std::atomic<unsigned long long> attr_devices{0};
        int dev = 0;
        RN_HIP(hipGetDevice(&dev));
        if (!(attr_devices.load(std::memory_order_acquire) >> (dev & 63) & 1ull)) {
            RN_HIP(hipFuncSetAttribute(reinterpret_cast<const void*>(kern), hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024));
            attr_devices.fetch_or(1ull << (dev & 63), std::memory_order_release);
        }
        hipLaunchKernelGGL(kern, dim3(a.n_bands * a.n_cb, n), dim3(nthreads), U_LDS, s, a);
        RN_CHECK_LAUNCH();
        return RN_OK;
    };
    // three live quarters (StageArgs::live_q == 3): the twelve-wave form pools 100 columns per block at most
    bool q3 = a.live_q == 3;
    for (int b = 0; b < a.n_cb; ++b) q3 = q3 && a.cb_wo[b] <= 100;
    if (q3) {
        nthreads = 768;
        if (dtype == RN_DTYPE_BF16) return launch(stage4x_kernel<RN_DTYPE_BF16, 3>);
        return launch(stage4x_kernel<RN_DTYPE_F16, 3>);
    }
    if (dtype == RN_DTYPE_BF16) return launch(stage4x_kernel<RN_DTYPE_BF16, 4>);
    return launch(stage4x_kernel<RN_DTYPE_F16, 4>);
}
