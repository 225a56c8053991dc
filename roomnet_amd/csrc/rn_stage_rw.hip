// "rw" (register-weights) fused stage kernel: stages 1-7 of the 16-bit path
// (conv3x3 -> ReLU6 -> [avg-pool 4x4 / s] -> BN [-> + legacy-bilinear(skip) -> BN]), one launch per stage.
//
// Row-streaming implicit GEMM: a workgroup owns (image, band of output rows, block of columns) and walks
// down its band one conv row per step.  What the counters and the ISA of the earlier versions led to
// (profiles/r1_*):
//   1. input rows (and the residual's skip rows) arrive by LDS-DMA (global_load_lds_dwordx4: no VGPR
//      staging, no ds_write) into a ring of row slots, several rows in flight, retired by COUNTED
//      s_waitcnt vmcnt(N) and a bare s_barrier that does not drain the memory pipe; single-cout-tile
//      stages give every wave a private ring and need no barrier at all; the DMA pieces of a step are
//      spread over the step instead of issued in one burst;
//   2. one wave = one (32-column pixel tile, 32-channel cout tile); its weight fragments (K/16 x 4 VGPRs)
//      are loaded once into registers: one ds_read_b128 per MFMA, issued a few chunks ahead through
//      inline asm with counted lgkmcnt waits;
//   3. the row loop is unrolled by the ring depth, so every ring access is "lane-constant VGPR +
//      immediate offset"; the XOR chunk swizzle that keeps the fragment reads bank-conflict free is
//      applied on the DMA's per-lane SOURCE address (the LDS image of a DMA piece is lane-linear);
//   4. pooling: stride-1 stages run the horizontal window sums on the matrix cores (transposed conv tile,
//      fp16 pair sums x 0/1 band matrix), stride-2 stages use row-local DPP shifts on gapped tiles;
//      BN is one fma, the residual's bilinear resize is an MFMA against the interpolation matrix;
//   5. the epilogue of row s-1 is cut into micro-ops that are placed, in source order, behind the MFMAs of
//      row s; stores are buffer stores predicated by an out-of-range offset, so a step has no branches.
#include "rn_fused.h"
#include "rn_stage.h"

#include <algorithm>
#include <atomic>
#include <cstdio>
#include <cstdlib>
#include <type_traits>
#include <utility>

using namespace rnk;

namespace {

// Tile geometry of the rw kernels: 32 conv columns per tile.
//  pool 4/1: the 29 windows inside the tile, window i on lane i; the horizontal sums run on the matrix
//            cores (RwCfg::POOLM);
//  pool 4/2: "gapped" tiles -- the two 16-lane DPP rows of a half-wave hold conv columns 0..15 and 14..29,
//            so every window that starts at an even column of a row ends inside that row: 7 + 7 outputs per
//            tile from two row-local DPP shifts (row_shl:1, row_shl:2) on the vertically summed row.
//            (Measured on the same box: the MFMA pool is 5 % faster than DPP for the stride-1 residual stage,
//             7 % slower for the stride-2 stages, which pool only every other row.)
//  pool 4/2, "wide" (the 32->64 stage, one pixel tile per workgroup): plain lane -> column map, 15 windows per tile
//            (even columns 0..28); the horizontal sums cross the 16-lane DPP rows, so they use wave shifts
//            (wave_shl:1, three DPP operations per value instead of two).  203 conv columns = 7 tiles instead of 8.
constexpr int rw_tile_nout(int pk, int ps, bool wide = false) { return pk ? (ps == 2 ? (wide ? 15 : 14) : 32 - pk + 1) : 32; }
constexpr int rw_tile_stride(int pk, int ps, bool wide = false) { return pk ? rw_tile_nout(pk, ps, wide) * ps : 32; }
#ifndef RN_SPREAD_DMA
#define RN_SPREAD_DMA 1
#endif
constexpr int RW_SKIPBUF = 3;   // staged skip-row pairs (residual): 1 being read + 2 in flight

template <int DT, int CIN, int COUT, int PK, int PS, bool RES, int NPT, int KS = 1, bool S0F_ = false, bool WIDE_ = false, int S0SH_ = 0>
struct RwCfg {
    // S0F: stage 0 (uint8 image -> conv 3->8 -> ReLU6 -> pool 3/1 -> BN) is computed by the SAME wave, row by row,
    // straight into its private ring: the 8-channel tensor between stages 0 and 1 never reaches HBM (see s0_feed)
    static constexpr bool S0F = S0F_;
    // S0SH (with S0F): the stage-0 rows go to ONE ring shared by the workgroup -- wave w computes the 29
    // stage-0 columns [29 w, 29 w + 29) (one 32-column tile instead of two: the second tile of the private form exists
    // for 5 halo columns only and costs as much as the first) and reads its 34-column window once its neighbours are done.
    // The eight waves produce 232 stage-0 columns = the windows of 227 output columns: column blocks of this form are 227
    // wide (the eighth tile stores 24 of its 29 columns), one at 224 x 224, two at 420, three at 600.
    static constexpr bool S0SH = S0SH_ != 0;
    static constexpr int S0SH_BLKW = 227;
    static_assert(!S0SH || S0F, "shared stage-0 ring is a form of the stage-0 fusion");
    // S0SH_ == 2 (S0HW): NPT / 2 extra "helper" waves compute the stage-0 rows (two 29-column tiles each) and the NPT tile
    // waves run stage 1 only: 12 waves of <= 168 registers = three per SIMD instead of two, and the stage-0 row (a long
    // dependent sequence: image bytes -> operand -> 3 MFMAs -> DPP sums -> BN -> LDS) runs beside the stage-1 chains of the
    // SIMD's other waves instead of in front of them
    static constexpr bool S0HW = S0SH_ == 2;
    static constexpr int S0_TILES = S0HW ? 2 : (S0SH ? 1 : 2);
    static_assert(!S0F || (CIN == 8 && !RES && KS == 1 && COUT == 32), "stage-0 fusion feeds the 8-channel private-ring variant");
    // Ring depth: at step s the DMA for input row s + AHEAD is issued; NSLOT = AHEAD + 1 slots
    // (3 live rows + AHEAD - 2 in flight).  Rows of the 8-channel stage are only ~3.8 KB, so it
    // keeps 9 of them in flight to cover the HBM latency (Little's law), the others 3.
    // (7 instead of 5 for the private-ring 32-channel stage measured the same: it is not latency-bound)
    static constexpr int AHEAD = S0F ? 3 : (CIN == 8 ? 11 : 5);       // S0F: rows are produced in place, nothing in flight
    static constexpr int NSLOT = AHEAD + 1;
    static constexpr int CP = CIN / 8;
    static constexpr int KC = (9 * CIN + 15) / 16;
    static constexpr int CT = (COUT + 31) / 32;
    static constexpr int NG = COUT >= 32 ? 4 : COUT / 8;   // 4-channel groups per lane half-row
    static constexpr int CPO = COUT / 8;                       // 16-byte chunks per output/skip pixel
    static constexpr bool WIDE2 = WIDE_;                       // wide stride-2 tiles (see rw_tile_nout)
    static_assert(!WIDE_ || (PK == 4 && PS == 2 && KS == 1), "wide tiles: the stride-2 DPP pooling variants");
    static constexpr int TSTRIDE = rw_tile_stride(PK, PS, WIDE2);
    static constexpr int NOUT_T = rw_tile_nout(PK, PS, WIDE2);
    // POOLM: the 4-wide horizontal window sums run on the matrix cores.  The conv MFMA is issued with its
    // operands swapped (D'[pixel][cout]: pixel rows in the accumulator registers, one cout per lane), so a
    // lane's ReLU6'd values, rounded to fp16 pairs, ARE an A-operand fragment V[cout][x] of a second MFMA
    // H[cout][xo] = V * Pm with the 0/1 band matrix Pm[x][xo] = (PS*xo <= x < PS*xo + 4) as a lane-constant
    // B operand -- no cross-lane movement at all, and H comes out in the usual cout-in-register /
    // pixel-on-lane layout for BN, residual and the stores.  Vertical pooling = fp32 pair sums before the
    // rounding, so a pooled row costs 4 MFMAs (2 pair-sum rows x K = 32) instead of 48 DPP instructions.
    static constexpr bool POOLM = PK == 4 && PS == 1;
    static constexpr bool RES_SPLIT = !POOLM;                  // residual interpolation weights as hi + lo 16-bit operands
    static constexpr bool GAP = PK == 4 && PS == 2 && !WIDE2;  // gapped lane -> column map, DPP pooling (see rw_tile_nout)
    static constexpr bool DPP2 = PK == 4 && PS == 2;           // stride-2 pooling by DPP (gapped or wide)
    // KS = 3: the K dimension is split by kernel row over three waves per pixel tile (each keeps one
    // kernel row's weight fragments in registers); partial accumulators meet in LDS (K = 1152 stage)
    static constexpr int NTHREADS = 64 * NPT * CT * KS + (S0SH_ == 2 ? 64 * (NPT / 2) : 0);
    static constexpr int KCW = KC / KS;                        // K-chunks per wave
    // PRIV: every wave owns a private ring holding just its own 34-column input tile and fetches
    // it itself.  No wave ever reads another wave's LDS data, so the row loop needs NO workgroup
    // barrier: waves drift apart and the two waves sharing a SIMD overlap MFMA with epilogue
    // instead of colliding in lockstep (stamps: the barrier alone cost ~20 % of a step).  The 5
    // halo columns neighbouring tiles re-fetch come from L2, not HBM.  (Residual variants keep the
    // workgroup-shared ring: their staged skip rows are shared too.)
    // Only for single-cout-tile stages: with several cout tiles the waves of one pixel tile would each
    // fetch the same input, and once they drift apart the duplicates miss L2 (measured on the 32->64
    // stage: 1.65x the algorithmic HBM bytes); those stages keep the workgroup-shared ring.
    static constexpr bool PRIV = !RES && KS == 1 && CT == 1 && !S0SH;
    static constexpr int RINGCOLS = PRIV ? 34 : (NPT - 1) * TSTRIDE + 34;
    static constexpr int LOADERS = PRIV ? 64 : NTHREADS;       // lanes cooperating on one ring row
    static constexpr int NRINGS = PRIV ? NPT * CT : 1;
    static constexpr int LPT = (RINGCOLS * CP + LOADERS - 1) / LOADERS;     // DMA pieces per wave per row
    // shared ring: row stride padded to whole pieces; private ring: exact (the last piece is lane-masked)
    static constexpr int ROWB = PRIV ? RINGCOLS * CIN * 2 : LPT * NTHREADS * 16;
    static constexpr int SKIPCOLS_MAX = RINGCOLS + 8;          // residual scale <= ~1.1 (checked on the host)
    static constexpr int SLPT = RES ? (2 * SKIPCOLS_MAX * CPO + NTHREADS - 1) / NTHREADS : 0;
    static constexpr int SKIPBUFB = SLPT * NTHREADS * 16;      // one staged pair of skip rows (padded)
    static constexpr int PTAB_BYTES = 4 * COUT * 4;
    static constexpr int RING_OFF = PTAB_BYTES;
    static constexpr int SKIP_OFF = RING_OFF + NRINGS * NSLOT * ROWB;
    // output staging (one 32-channel cout tile only): each wave transposes its tile-row
    // [pixel][64 B] through LDS so that the global stores are lane-linear (1 KB contiguous per
    // instruction) instead of 16 B per lane at a 64-byte stride
    // (measured: +23 % on the 8->32 stage, +13 % on the 32->32 stage: 64 lanes x 16 B at a 64-byte
    //  stride are 64 partial-line write requests per instruction, lane-linear stores are 8 full lines;
    //  -10 % on the residual variant: its extra LDS round trip sits on the critical path of a one-wave-per-SIMD step)
    static constexpr bool STAGE_OUT = COUT == 32 && !RES;
    // folded-BN tables: persistent registers where the register file has room (one wave per SIMD, or the
    // small 8-channel stage); otherwise one batched LDS read at the start of every epilogue
    static constexpr bool PTAB_REGS = (NTHREADS <= 256 || CIN == 8 || (CIN == 32 && COUT == 64)) && !S0HW;   // (S0HW: 168 registers per wave)
    // ... except the 8-wave 32->32 variant, which sits at the 256-register cap: it reads each group's
    // table entries late (right before use) so they never pin registers across the MFMA chain
    static constexpr bool PTAB_LATE = !PTAB_REGS && CIN == 32 && COUT == 32;
    static constexpr int STAGE_WAVE_B = 32 * 64;
    static constexpr int STAGE_OFF = SKIP_OFF + (RES ? RW_SKIPBUF * SKIPBUFB : 0);
    static constexpr int PART_OFF = STAGE_OFF + (STAGE_OUT ? NPT * CT * STAGE_WAVE_B : 0);
    static constexpr int PART_B = 16 * 64 * 4;                 // one wave's partial accumulator tile
    static constexpr int LDS_BYTES = PART_OFF + (KS > 1 ? 2 * NPT * CT * (KS - 1) * PART_B : 0);
    // steady-state counted wait at the end of step s: everything up to input row s+3 and the
    // skip pair used by step s+1 has landed; what may stay in flight is what the wave issued
    // after them (the pieces of this step, plus one more row of input when there is no skip)
    // (residual with pool stride 2 issues a skip pair on even steps only: per-phase counts)
    static constexpr int vmcnt_steady(int phase) {
        if (!RES) return (AHEAD - 3) * LPT;
        if (PS == 1) return LPT + SLPT;
        return (phase & 1) == 0 ? 2 * LPT + 2 * SLPT : LPT;
    }
    static constexpr int VMCNT_STEADY = (RES ? 2 * LPT + 2 * SLPT : (AHEAD - 3) * LPT);   // upper bound (field-width check)
    static_assert(!RES || AHEAD == 5, "the residual wait counts are derived for AHEAD = 5");
    static_assert(COUT % 32 == 0 || COUT == 16, "cout must be whole 32-channel tiles (or one half tile)");
    static_assert(PK == 0 || PK == 4, "pool window 4 or none");
    static_assert(KCW * 4 <= 80 || NTHREADS <= 256 || KS > 1, "weights need the whole register file: <= 1 wave per SIMD");
    static_assert(KS == 1 || (KS == 3 && CIN >= 16 && KC % 3 == 0 && !RES), "K split = one kernel row per wave");
    static_assert(LDS_BYTES <= 160 * 1024, "LDS budget");
    static_assert(NSLOT % 2 == 0, "pool-ring parity is tied to the unroll");
    static_assert(!PRIV || (LPT - 1) * 64 < RINGCOLS * CP, "every DMA piece must have at least one active lane");
    static_assert(VMCNT_STEADY <= 63, "vmcnt field");
};

#ifdef RN_STAMPS
// In-kernel stamps (diagnostic build only; never in the shipped library): s_memtime + its own wait
// in ONE asm statement, fenced by sched_barrier so the segments hold what they are named for.
__device__ __forceinline__ unsigned long long stamp() {
    unsigned long long t;
    __builtin_amdgcn_sched_barrier(0);
    asm volatile("s_memtime %0\n\ts_waitcnt lgkmcnt(0)" : "=s"(t)::"memory");
    __builtin_amdgcn_sched_barrier(0);
    return t;
}
#endif

template <int DT, int CIN, int COUT, int PK, int PS, bool RES, int NPT, int KS, bool S0F, bool WIDE, int S0SH>
__global__ __launch_bounds__((RwCfg<DT, CIN, COUT, PK, PS, RES, NPT, KS, S0F, WIDE, S0SH>::NTHREADS), 1) void stage_rw_kernel(const StageArgs a) {
    using C = RwCfg<DT, CIN, COUT, PK, PS, RES, NPT, KS, S0F, WIDE, S0SH>;
    constexpr int CP = C::CP, KC = C::KC, CT = C::CT, CPO = C::CPO, TSTRIDE = C::TSTRIDE, NOUT_T = C::NOUT_T;
    constexpr int RINGCOLS = C::RINGCOLS, ROWB = C::ROWB, NTHREADS = C::NTHREADS, LPT = C::LPT, SLPT = C::SLPT;
    constexpr int PIXB = CIN * 2, NG = C::NG;
    constexpr int RW_NSLOT = C::NSLOT, RW_AHEAD = C::AHEAD;

    extern __shared__ __attribute__((aligned(16))) char smem[];
#ifdef RN_STAMPS
    const unsigned long long st_entry = stamp();
#endif
#ifdef RN_CLOCK
    unsigned long long ck_t0, ck_r0;
    clock_pair(ck_t0, ck_r0);
#endif
    const int tid = threadIdx.x;
    const int lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int pt = wave % NPT, ct = (wave / NPT) % CT;
    const bool helper = C::S0HW && wave >= NPT;      // stage-0 helper wave (wave-uniform); tiles 2 hq, 2 hq + 1
    const int hq = wave - NPT;
    const int ks = __builtin_amdgcn_readfirstlane(wave / (NPT * CT));     // kernel row of this wave (K split)
    constexpr int KCW = C::KCW;
    const int r = lane & 31, hh = lane >> 5;

    int bid = blockIdx.x;
    const int cb = bid % a.n_colblocks;
    const int band = bid / a.n_colblocks;
    const int n = blockIdx.y;

    float* const ptab = reinterpret_cast<float*>(smem);
    char* const ring = smem + C::RING_OFF + (C::PRIV ? wave * (C::NSLOT * C::ROWB) : 0);
    char* const skipb = smem + C::SKIP_OFF;
    const int skipbytes = a.skipcols * COUT * 2;        // bytes per staged skip row

    const int yo0 = band * a.rows_per_band;
    const int yo1 = min(a.Ho, yo0 + a.rows_per_band);
    const int nout_rows = yo1 - yo0;
    const int yc0 = PK ? yo0 * PS : yo0;
    const int nconv = PK ? (nout_rows - 1) * PS + 4 : nout_rows;
    const int nin = nconv + 2;
    constexpr int BLKC = C::S0SH ? C::S0SH_BLKW : NPT * TSTRIDE;           // conv columns per column block
    constexpr int BLKO = C::S0SH ? C::S0SH_BLKW : NPT * NOUT_T;            // output columns per column block
    const int x0c = cb * BLKC;
    const int xo_blk0 = PK ? x0c / PS : x0c;

    // ---- folded BN tables -> LDS
    for (int i = tid; i < 4 * COUT; i += NTHREADS) ptab[i] = a.ptab[i];
    // Private-ring variants: a wave whose whole tile lies right of the image has nothing to do and shares nothing with
    // the others (no barrier in their row loop; a finished wave is not waited for by the one barrier of the prologue):
    // it leaves after its share of the table copy.  (600 x 600: 21 tiles in 3 blocks of 8 -- three such waves.)
    if constexpr (C::PRIV) {
        if ((PK ? (x0c + pt * TSTRIDE) / PS : x0c + pt * TSTRIDE) >= a.Wo) return;
    }

    // ---- this wave's weight fragments -> registers (lane-linear, coalesced)
    i32x4 wreg[KCW];
#pragma unroll
    for (int kc = 0; kc < KCW; ++kc) wreg[kc] = a.wfrag[((ks * KCW + kc) * CT + ct) * 64 + lane];

    // ---- input-row DMA: piece i of a row covers ring chunks [i*NTHREADS, (i+1)*NTHREADS); this
    // lane fills chunk q = tid + i*NTHREADS = (pixel p, slot c') and therefore fetches source
    // chunk c' ^ swz(p) of pixel p.  Pieces past the row end land in the row's padding; columns
    // past the image edge are clamped (they only feed discarded lanes).
    // (addresses are "uniform row base + lane-constant unsigned byte offset": the global_load_lds
    //  saddr form, no per-row VALU address arithmetic)
    const char* const in_img = reinterpret_cast<const char*>(a.in + static_cast<int64_t>(n) * a.H * a.W * CIN);
    const int64_t in_row_bytes = static_cast<int64_t>(a.W) * CIN * 2;
    unsigned ld_goff[LPT];
    const int x_ring0 = C::PRIV ? x0c + pt * TSTRIDE : x0c;            // image column of ring column 0
#pragma unroll
    for (int i = 0; i < LPT; ++i) {
        const int q = (C::PRIV ? lane : tid) + i * C::LOADERS;
        const int p = q / CP, c8 = q % CP;
        const int pc = min(x_ring0 + min(p, RINGCOLS - 1), a.W - 1);
        ld_goff[i] = static_cast<unsigned>((pc * CIN + (c8 ^ chunk_swz<CP>(p)) * 8) * 2);
    }
    const int piece_base = C::PRIV ? 0 : wave * 64 * 16;   // LDS byte offset of this wave inside a piece
    // private ring: active lanes of the last (partial) piece of a row
    constexpr int TAIL_LANES = C::PRIV ? RINGCOLS * CP - (LPT - 1) * 64 : 64;
    constexpr unsigned long long tail_mask = TAIL_LANES >= 64 ? ~0ull : ((1ull << TAIL_LANES) - 1ull);
    // piece i of input row yc0 + j -> ring slot
    auto issue_row_piece_at = [&](auto II, const char* row, int slot) __attribute__((always_inline)) {
        constexpr int i = decltype(II)::value;
        if constexpr (C::PRIV) {
            if constexpr ((i + 1) * 64 <= RINGCOLS * CP)
                dma16(row + ld_goff[i], ring + slot * ROWB + i * 64 * 16);
            else
                dma16_masked(row + ld_goff[i], ring + slot * ROWB + i * 64 * 16, tail_mask);
        } else {
            dma16(row + ld_goff[i], ring + slot * ROWB + i * NTHREADS * 16 + piece_base);
        }
    };
    auto issue_row = [&](int j, int slot) __attribute__((always_inline)) {
        const char* row = in_img + static_cast<int64_t>(yc0 + j) * in_row_bytes;
        [&]<int... II>(std::integer_sequence<int, II...>) {
            (issue_row_piece_at(IC<II>{}, row, slot), ...);
        }(std::make_integer_sequence<int, LPT>{});
    };

    // ---- skip-row DMA (residual stages): the pair of rows lo/hi of one output row
    int xs0 = 0;
    unsigned sk_goff[SLPT > 0 ? SLPT : 1];     // byte offset inside a skip row
    bool sk_hi[SLPT > 0 ? SLPT : 1];           // this lane's chunk of piece i belongs to the hi row
    const char* skip_img = nullptr;
    const int64_t skip_row_bytes = static_cast<int64_t>(a.Ss) * COUT * 2;
    if constexpr (RES) {
        xs0 = a.rlo[min(xo_blk0, a.Wo - 1)];
        skip_img = reinterpret_cast<const char*>(a.skip + static_cast<int64_t>(n) * a.Ss * a.Ss * COUT);
        const int per_row = a.skipcols * CPO;
#pragma unroll
        for (int i = 0; i < SLPT; ++i) {
            const int q = tid + i * NTHREADS;
            const int row = q >= per_row ? 1 : 0;
            const int qq = min(q - row * per_row, per_row - 1);
            const int p = qq / CPO, c8 = qq % CPO;
            const int pc = min(xs0 + p, a.Ss - 1);
            sk_goff[i] = static_cast<unsigned>((pc * COUT + (c8 ^ chunk_swz<CPO>(p)) * 8) * 2);
            sk_hi[i] = row != 0;
        }
    }
    // piece i of the skip-row pair of local output row e -> buffer
    // (row base and lo->hi distance are computed once per step by skip_rows(), not per piece)
    struct SkipRows {
        const char* r0;
        unsigned hi_delta;
    };
    auto skip_rows = [&](int e) __attribute__((always_inline)) -> SkipRows {
        SkipRows sr{nullptr, 0u};
        if constexpr (RES) {
            const int yo = yo0 + min(max(e, 0), nout_rows - 1);
            // TF-1.13 compute_interpolation_weights: src = yo * scale (fp32), lo = int(src), hi = min(lo+1, in-1)
            const float src = mul_rounded(static_cast<float>(yo), a.rscale);
            const int ylo = static_cast<int>(src);
            const int yhi = min(ylo + 1, a.Ss - 1);
            sr.r0 = skip_img + static_cast<int64_t>(ylo) * skip_row_bytes;
            sr.hi_delta = static_cast<unsigned>((yhi - ylo) * skip_row_bytes);
        }
        return sr;
    };
    auto issue_skip_piece = [&](auto II, const SkipRows& sr, int buf) __attribute__((always_inline)) {
        constexpr int i = decltype(II)::value;
        if constexpr (RES)
            dma16(sr.r0 + (sk_goff[i] + (sk_hi[i] ? sr.hi_delta : 0u)), skipb + buf * C::SKIPBUFB + i * NTHREADS * 16 + piece_base);
    };
    auto issue_skip = [&](int e, int buf) __attribute__((always_inline)) {               // skip rows of local output row e -> buffer
        if constexpr (RES) {
            const int yo = yo0 + min(max(e, 0), nout_rows - 1);
            // TF-1.13 compute_interpolation_weights: src = yo * scale (fp32), lo = int(src), hi = min(lo+1, in-1)
            const float src = mul_rounded(static_cast<float>(yo), a.rscale);
            const int ylo = static_cast<int>(src);
            const int yhi = min(ylo + 1, a.Ss - 1);
            // uniform base = lo row; lanes of the hi row add the (uniform) row distance
            const char* r0 = skip_img + static_cast<int64_t>(ylo) * skip_row_bytes;
            const unsigned hi_delta = static_cast<unsigned>((yhi - ylo) * skip_row_bytes);
#pragma unroll
            for (int i = 0; i < SLPT; ++i)
                dma16(r0 + (sk_goff[i] + (sk_hi[i] ? hi_delta : 0u)),
                      skipb + buf * C::SKIPBUFB + i * NTHREADS * 16 + piece_base);
        }
    };

    // ---- stage-0 fusion (S0F): this wave computes stage 0 for its own 34 ring columns, one row per call, on the matrix
    // cores with the im2col in registers exactly like stage0_kernel (rn_fused.hip; same instruction sequence, so the
    // ring holds bit for bit what the two-launch path reads back from HBM).  34 output columns = 36 conv columns = two
    // 32-column MFMA tiles: tile 0 at the ring origin gives output columns 0..28 (conv column 31 of a tile is not
    // computable: its right neighbour pixel sits in the other half-wave), tile 1 at +29 gives 29..33.
    // s0_feed(i, slot): takes image row yc0 + i; from i = 2 a conv row (i - 2) is complete, from i = 4 output row
    // i - 4 = ReLU6 -> 3x3 sums -> BN, written as 8-byte lane pieces into ring slot `slot`.
    constexpr int S0_AHEAD = 4;            // image rows prefetched per tile: a circular queue indexed by row mod 4, which
                                           // is a compile-time phase inside the row loop (unrolled by the ring depth 4)
    i32x4 s0_w[3];
    f32x4 s0_scale = {0.f, 0.f, 0.f, 0.f}, s0_shift = {0.f, 0.f, 0.f, 0.f};
    const uint8_t* s0_src[2] = {nullptr, nullptr};
    int s0_sh[2] = {0, 0};
    unsigned s0_pw[2][S0_AHEAD];
    i32x4 s0_bfr[2][3];
    float s0_h1[2][4], s0_h2[2][4];
    int s0_wr[2] = {0, 0};                 // byte offset of this lane's 8-byte piece inside a ring row, or -1
    const int s0_rows = nin + 4;           // image rows this band reads
    if (C::S0F && (!C::S0HW || helper)) {
#pragma unroll
        for (int ky = 0; ky < 3; ++ky) s0_w[ky] = a.s0_wfrag[ky * 64 + lane];
        s0_scale = *reinterpret_cast<const f32x4*>(a.s0_ptab + 4 * hh);
        s0_shift = *reinterpret_cast<const f32x4*>(a.s0_ptab + 8 + 4 * hh);
#pragma unroll
        for (int u = 0; u < C::S0_TILES; ++u) {
            const int tix = C::S0HW ? 2 * hq + u : pt;                   // shared ring: the 29-column tile this (wave, u) produces
            const int xt0 = x0c + (C::S0HW ? tix * 29 : pt * TSTRIDE + 29 * u);   // first conv / image column of the tile
            const int px = min(xt0 + r + 2 * hh, a.s0_S - 1);           // this lane's image column (clamped at the edge)
            s0_sh[u] = px == a.s0_S - 1 ? 8 : 0;                        // last column: load one byte early and shift
            s0_src[u] = a.s0_bgr + (static_cast<int64_t>(n) * a.s0_S * a.s0_S + static_cast<int64_t>(yc0) * a.s0_S + px) * 3 - (s0_sh[u] >> 3);
            const int oc = C::S0HW ? tix * 29 + r : (C::S0SH ? pt * TSTRIDE : 0) + 29 * u + r;   // ring column of this lane's output pixel
            s0_wr[u] = (r < (u == 0 || C::S0HW ? 29 : 5) && oc < RINGCOLS) ? oc * PIXB + 8 * hh : -1;
#pragma unroll
            for (int j = 0; j < 4; ++j) s0_h1[u][j] = s0_h2[u][j] = 0.f;
#pragma unroll
            for (int i = 0; i < 3; ++i) s0_bfr[u][i] = i32x4{0, 0, 0, 0};
        }
    }
    auto s0_load = [&](int u, int j) __attribute__((always_inline)) -> unsigned {
        unsigned w;
        __builtin_memcpy(&w, s0_src[u] + static_cast<int64_t>(min(j, s0_rows - 1)) * (a.s0_S * 3), 4);   // unaligned dword
        return w;
    };
    auto s0_feed = [&](auto STEADYC, auto QC, int i, int slot) __attribute__((always_inline)) {
        constexpr bool STEADY = decltype(STEADYC)::value != 0;          // i >= 4 is known: no start-up branches
        constexpr int Q = decltype(QC)::value;                          // i mod S0_AHEAD
        if constexpr (C::S0F) {
            const unsigned nb_mask = hh ? 0u : 0xffffffffu;
            const f32x16 zero = {0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f};
#pragma unroll
            for (int u = 0; u < C::S0_TILES; ++u) {
                // the byte values as fp16 numbers, exact; ((x / 255.) * 2) - 1 of network.py:129 lives in the weights
                const unsigned w = s0_pw[u][Q] >> s0_sh[u];              // bytes: B, G, R
                int d0, d1;
                s0_pixel_halves(w, d0, d1);
                s0_pw[u][Q] = s0_load(u, i + S0_AHEAD);
                i32x4 f;
                f[0] = d0;
                f[1] = d1;
                f[2] = static_cast<int>(static_cast<unsigned>(__builtin_amdgcn_update_dpp(0, d0, 0x130, 0xf, 0xf, true)) & nb_mask);
                f[3] = static_cast<int>(static_cast<unsigned>(__builtin_amdgcn_update_dpp(0, d1, 0x130, 0xf, 0xf, true)) & nb_mask);
                s0_bfr[u][2] = f;
                if (STEADY || i >= 2) {
                    f32x16 acc = mfma32<RN_DTYPE_F16>(s0_w[0], s0_bfr[u][0], zero);
                    acc = mfma32<RN_DTYPE_F16>(s0_w[1], s0_bfr[u][1], acc);
                    acc = mfma32<RN_DTYPE_F16>(s0_w[2], s0_bfr[u][2], acc);
                    float y[4];
#pragma unroll
                    for (int j = 0; j < 4; ++j) {
                        const float v = s0_relu6(acc, j);
                        const float v1 = lane_next(v);
                        const float hs = (v + v1) + lane_next(v1);
                        y[j] = fmaf((s0_h2[u][j] + s0_h1[u][j]) + hs, s0_scale[j], s0_shift[j]);
                        s0_h2[u][j] = s0_h1[u][j];
                        s0_h1[u][j] = hs;
                    }
                    if ((STEADY || i >= 4) && s0_wr[u] >= 0)
                        *reinterpret_cast<uint2*>(ring + slot * ROWB + s0_wr[u]) = pack4<DT>(y[0], y[1], y[2], y[3]);
                }
                if (STEADY || i >= 1) {
                    s0_bfr[u][0] = s0_bfr[u][1];
                    s0_bfr[u][1] = s0_bfr[u][2];
                } else {
                    s0_bfr[u][1] = s0_bfr[u][2];
                }
            }
        }
    };
    // The same row cut into pieces for the steady state: the front (operand + the three MFMAs) runs at the top of a step, the
    // rest rides in the chain slots of stage 1's conv row (s0_part<0,1>: ReLU6 + 3x3 sums of two channels each, <2>: BN +
    // write) -- one wave's stage-0 row is a long dependent sequence (MFMA chain -> DPP sums -> BN -> LDS), and both waves
    // of a SIMD ran it at the same time (stamps: 755 cycles a row for ~260 cycles of issue)
    f32x16 s0_acc[2];
    float s0_y[2][4];
    auto s0_front = [&](auto QC, int i) __attribute__((always_inline)) {
        constexpr int Q = decltype(QC)::value;
        if constexpr (C::S0F) {
            const unsigned nb_mask = hh ? 0u : 0xffffffffu;
            const f32x16 zero = {0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f};
#pragma unroll
            for (int u = 0; u < C::S0_TILES; ++u) {
                const unsigned w = s0_pw[u][Q] >> s0_sh[u];              // bytes: B, G, R
                int d0, d1;
                s0_pixel_halves(w, d0, d1);
                s0_pw[u][Q] = s0_load(u, i + S0_AHEAD);
                i32x4 f;
                f[0] = d0;
                f[1] = d1;
                f[2] = static_cast<int>(static_cast<unsigned>(__builtin_amdgcn_update_dpp(0, d0, 0x130, 0xf, 0xf, true)) & nb_mask);
                f[3] = static_cast<int>(static_cast<unsigned>(__builtin_amdgcn_update_dpp(0, d1, 0x130, 0xf, 0xf, true)) & nb_mask);
                f32x16 acc = mfma32<RN_DTYPE_F16>(s0_w[0], s0_bfr[u][0], zero);
                acc = mfma32<RN_DTYPE_F16>(s0_w[1], s0_bfr[u][1], acc);
                s0_acc[u] = mfma32<RN_DTYPE_F16>(s0_w[2], f, acc);
                s0_bfr[u][0] = s0_bfr[u][1];
                s0_bfr[u][1] = f;
            }
        }
    };
    auto s0_part = [&](auto KC_, int slot) __attribute__((always_inline)) {
        constexpr int k = decltype(KC_)::value;
        if constexpr (C::S0F) {
#pragma unroll
            for (int u = 0; u < C::S0_TILES; ++u) {
                if constexpr (k < 2) {
#pragma unroll
                    for (int j = 2 * k; j < 2 * k + 2; ++j) {
                        const float v = s0_relu6(s0_acc[u], j);
                        const float v1 = lane_next(v);
                        const float hs = (v + v1) + lane_next(v1);
                        s0_y[u][j] = fmaf((s0_h2[u][j] + s0_h1[u][j]) + hs, s0_scale[j], s0_shift[j]);
                        s0_h2[u][j] = s0_h1[u][j];
                        s0_h1[u][j] = hs;
                    }
                } else {
                    if (s0_wr[u] >= 0)
                        *reinterpret_cast<uint2*>(ring + slot * ROWB + s0_wr[u]) = pack4<DT>(s0_y[u][0], s0_y[u][1], s0_y[u][2], s0_y[u][3]);
                }
            }
        }
    };
    // ---- prologue: rows 0 .. RW_AHEAD-1 in flight (clamped: a no-pool band can be shorter)
    if constexpr (C::S0F) {
        if constexpr (C::S0SH) {
            // ring columns 232 .. of every slot are written by nobody and feed discarded lanes only: finite values
            static_assert(!C::S0SH || (TSTRIDE == 29 && NPT == 8), "shared stage-0 ring: 8 tiles of 29 columns");
            constexpr int Z0 = NPT * 29 * PIXB, ZN = (ROWB - Z0) / 16;
            for (int i = tid; i < RW_NSLOT * ZN; i += NTHREADS)
                *reinterpret_cast<i32x4*>(ring + (i / ZN) * ROWB + Z0 + (i % ZN) * 16) = i32x4{0, 0, 0, 0};
        }
        static_assert(!C::S0F || (RW_NSLOT == 4 && S0_AHEAD == 4 && RW_AHEAD == 3), "queue phase = ring phase");
        if (!C::S0HW || helper) {
#pragma unroll
            for (int u = 0; u < C::S0_TILES; ++u)
#pragma unroll
                for (int k = 0; k < S0_AHEAD; ++k) s0_pw[u][k] = s0_load(u, k);
            s0_feed(IC<0>{}, IC<0>{}, 0, 0);
            s0_feed(IC<0>{}, IC<1>{}, 1, 0);
            s0_feed(IC<0>{}, IC<2>{}, 2, 0);
            s0_feed(IC<0>{}, IC<3>{}, 3, 0);
            s0_feed(IC<1>{}, IC<0>{}, 4, 0);               // emits ring rows 0 .. RW_AHEAD-1
            s0_feed(IC<1>{}, IC<1>{}, 5, 1);
            s0_feed(IC<1>{}, IC<2>{}, 6, 2);
        }
    } else {
#pragma unroll
        for (int j = 0; j < RW_AHEAD; ++j) issue_row(min(j, nin - 1), j);
    }

    // ---- lane constants of this wave's pixel tile
    const int pm = C::GAP ? r - 2 * (r >> 4) : r;            // conv column of this lane inside the tile
    const int xrel0 = (C::PRIV ? 0 : pt * TSTRIDE) + pm;     // ring column of conv column (tap kx = 0)
    int boff[3][CIN >= 16 ? CIN / 16 : 1];
    int b8_ky[CIN >= 16 ? 1 : KC], b8_off[CIN >= 16 ? 1 : KC];   // CIN == 8: per K-chunk tap row / offset
    if constexpr (CIN >= 16) {
#pragma unroll
        for (int kx = 0; kx < 3; ++kx)
#pragma unroll
            for (int cc = 0; cc < CIN / 16; ++cc)
                boff[kx][cc] = (xrel0 + kx) * PIXB + (((cc * 2 + hh) ^ chunk_swz<CP>(xrel0 + kx)) << 4);
    } else {
#pragma unroll
        for (int kc = 0; kc < KC; ++kc) {
            int tap = 2 * kc + hh;
            tap = tap > 8 ? 8 : tap;          // K padded 72 -> 80, zero weights there
            const int ky = tap / 3, kx = tap - 3 * ky;
            b8_ky[kc] = ky;
            b8_off[kc] = (xrel0 + kx) * PIXB;
        }
    }
    // 8-channel stage: K-chunk kc holds taps 2 kc (lower half-wave) and 2 kc + 1 (upper): both lie in ONE kernel row except
    // for chunk 1 (taps 2 | 3 = rows 0 | 1), so the ring slot of a chunk is a compile-time DS offset on a lane-constant base
    // (chunk 1: one precomputed address per ring phase) -- no per-row address arithmetic, and the reads are inline asm so
    // they go out where they are written (stage-0 fusion: all five at the top of the step, behind them the stage-0 row)
    constexpr bool B8_ASM = CIN == 8 && RW_NSLOT <= 4;
    unsigned b8_base[CIN >= 16 ? 1 : KC], b8_k1[B8_ASM ? RW_NSLOT : 1];
    if constexpr (B8_ASM) {
        static_assert(!B8_ASM || (RW_NSLOT - 1) * ROWB + 16 <= 65535, "ring slot as a DS immediate offset");
        const unsigned ring_lds0 = static_cast<unsigned>(reinterpret_cast<uintptr_t>((__attribute__((address_space(3))) char*)ring));
#pragma unroll
        for (int kc = 0; kc < KC; ++kc) b8_base[kc] = ring_lds0 + static_cast<unsigned>(b8_off[kc]);
#pragma unroll
        for (int p = 0; p < RW_NSLOT; ++p) b8_k1[p] = b8_base[1] + static_cast<unsigned>(((p + hh) % RW_NSLOT) * ROWB);
    }
    i32x4 b8_pre[B8_ASM ? KC : 1];
    // output column of this lane: MFMA-pooled tiles deliver window i of the tile on lane i (see POOLM), gapped
    // tiles the window that starts at the lane's own conv column
    const int xo = C::DPP2 ? (x0c + pt * TSTRIDE + pm) / PS : (PK ? (x0c + pt * TSTRIDE) / PS : x0c + pt * TSTRIDE) + r;
    // a window of the tile ends up on this lane (gapped / wide tiles: the window that starts at the lane's own, even, column)
    const bool lane_win = C::GAP ? ((r & 1) == 0 && (r & 15) <= 12) : (C::WIDE2 ? ((r & 1) == 0 && r <= 28) : r < NOUT_T);
    const bool lane_out = lane_win && xo < a.Wo && (xo - xo_blk0) < BLKO;
    const float* const ptab_lane = ptab + ct * 32 + 4 * hh;                                // + 8*g (+ table*COUT)
    f32x4 sc1r[NG], sh1r[NG], sc2r[NG], sh2r[NG];
    if constexpr (C::PTAB_REGS) {
#pragma unroll
        for (int g = 0; g < NG; ++g) {
            const float* gp = a.ptab + ct * 32 + 4 * hh + 8 * g;
            sc1r[g] = *reinterpret_cast<const f32x4*>(gp);
            sh1r[g] = *reinterpret_cast<const f32x4*>(gp + COUT);
            if constexpr (RES) {
                sc2r[g] = *reinterpret_cast<const f32x4*>(gp + 2 * COUT);
                sh2r[g] = *reinterpret_cast<const f32x4*>(gp + 3 * COUT);
            }
        }
    }
    // staging: this lane's output pixel index inside the tile, number of valid pixels of the tile
    const int pr = r;
    const int xo_t0s = PK ? (x0c + pt * TSTRIDE) / PS : (x0c + pt * TSTRIDE);     // first output column of the tile
    const int nvalid = max(0, min(NOUT_T, min(a.Wo, xo_blk0 + BLKO) - xo_t0s)); // wave-uniform
    // LDS byte addresses (as 32-bit LDS offsets, used by inline-asm DS ops):
    //   write: after the half-wave swap this lane holds 16-byte chunk (2k + hh) of pixel pr, stored at
    //          pixel*64 + ((chunk ^ swz) << 4); the k = 1 address is the k = 0 address ^ 32
    //   read : lane-linear chunk g = lane + 64k of the tile-row; the k = 1 address is + 1024
    unsigned st_w0 = 0, st_r0 = 0;
    if constexpr (C::STAGE_OUT) {
        const unsigned stage_base = static_cast<unsigned>(reinterpret_cast<uintptr_t>(
            (__attribute__((address_space(3))) char*)(smem + C::STAGE_OFF + wave * C::STAGE_WAVE_B)));
        st_w0 = stage_base + pr * 64 + ((hh ^ ((pr >> 1) & 3)) << 4);
        const int gp = lane >> 2, gc = lane & 3;
        st_r0 = stage_base + gp * 64 + ((gc ^ ((gp >> 1) & 3)) << 4);
    }
    // Output stores go through a raw buffer resource over ONE output row (base = row, num_records =
    // row bytes, voffset = lane): a lane that must not store passes an out-of-range voffset and the
    // hardware drops it -- predication without EXEC changes or branches, so a row step stays one
    // basic block.  soffset stays the literal 0 on purpose: with an SGPR soffset hipcc pads no wait
    // state between a 128-bit buffer store and a VALU write of its data registers (LLVM assumes the
    // hazard does not exist in that form); on gfx950 the next v_cndmask then corrupted the first data
    // dword of the lanes read last (found as NaNs in 4 pixels per tile-row).
    constexpr int OOB = 0x40000000;
    const int out_row_bytes = a.Wo * COUT * 2;
    const char* const out_img = reinterpret_cast<const char*>(a.out + static_cast<int64_t>(n) * a.Ho * a.Wo * COUT);
    auto out_row_rsrc = [&](int yo) __attribute__((always_inline)) {
        return __builtin_amdgcn_make_buffer_rsrc(const_cast<char*>(out_img) + static_cast<int64_t>(yo) * out_row_bytes, 0,
                                                 out_row_bytes, 0x00020000);
    };
#ifdef RN_DIAG
    const bool stores_on = !(a.dbg_flags & 1);     // diagnostic builds only: timing without the output stores
#else
    constexpr bool stores_on = true;
#endif
    // direct stores: this lane's 16-byte chunk of pixel xo (second chunk at +32 bytes)
    const int voff_lane = (lane_out && stores_on) ? (xo * COUT + ct * 32 + 8 * hh) * 2 : OOB;
    // staged stores: lane-linear 16-byte chunks of the tile-row (second instruction + 1024 bytes)
    const int voff_st0 = (stores_on && lane < 4 * nvalid) ? xo_t0s * COUT * 2 + lane * 16 : OOB;
    const int voff_st1 = (stores_on && lane + 64 < 4 * nvalid) ? xo_t0s * COUT * 2 + lane * 16 + 1024 : OOB;

    // ---- residual on the matrix cores: R[cout][x_out] = Skip^T[cout][x_in] * Wx[x_in][x_out]
    // (Wx = the legacy-bilinear interpolation matrix of this tile: two non-zeros per column).
    // K = 32 skip columns starting at the tile's first source column; A fragments come from
    // the staged [x_in][cout] skip rows with the transposed read ds_read_b64_tr_b16; the
    // weights are lane constants, split hi + lo into two 16-bit operands so the interpolation
    // keeps ~16 bits (one bf16 operand would quantise the lerp weights to 8 bits).
    int a_off[4];                // byte offsets of the 4 transposed reads (2 K-chunks x 2 halves)
    i32x4 bw_h[2], bw_l[2];
    if constexpr (RES) {
        const int xo_t0 = min((x0c + pt * TSTRIDE) / PS, a.Wo - 1);
        const int xs_t = a.rlo[xo_t0] - xs0;                 // K origin inside the staged block
        const int xq = min(xo, a.Wo - 1);
        const int plo = a.rlo[xq] - xs0, phi = a.rhi[xq] - xs0;
        // stride-1 residual (scale ~1.05, shared with the cross-stage fused kernel rn_stage23.hip): ONE 16-bit operand
        // whose two weights are exact (res_quant_lerp); the stride-2 residual keeps the hi + lo split
        const float xl = C::RES_SPLIT ? a.rlerp[xq] : res_quant_lerp<DT>(a.rlerp[xq]);
#pragma unroll
        for (int c = 0; c < 2; ++c) {
            unsigned short wh[8], wl[8];
#pragma unroll
            for (int j = 0; j < 8; ++j) {
                const int xin = xs_t + 16 * c + 8 * hh + j;
                float w = 0.f;
                if (xin == plo) w += 1.0f - xl;
                if (xin == phi) w += xl;
                wh[j] = to16<DT>(w);
                wl[j] = to16<DT>(w - from16<DT>(wh[j]));
            }
#pragma unroll
            for (int d = 0; d < 4; ++d) {
                bw_h[c][d] = static_cast<int>(static_cast<unsigned>(wh[2 * d]) | (static_cast<unsigned>(wh[2 * d + 1]) << 16));
                bw_l[c][d] = static_cast<int>(static_cast<unsigned>(wl[2 * d]) | (static_cast<unsigned>(wl[2 * d + 1]) << 16));
            }
        }
        // transposed read: lane 4q+p of a 16-lane group addresses row q, columns 4p..4p+3 of a
        // 4-row x 16-column block and receives column (lane & 15) of the 4 rows
        const int grp = lane >> 4, li = lane & 15, q = li >> 2, pp = li & 3;
#pragma unroll
        for (int c = 0; c < 2; ++c)
#pragma unroll
            for (int t = 0; t < 2; ++t) {
                const int pix = min(max(xs_t + 16 * c + 8 * (grp >> 1) + 4 * t + q, 0), a.skipcols - 1);
                const int ch = ct * 4 + 2 * (grp & 1) + (pp >> 1);
                a_off[2 * c + t] = (pix * CPO + (ch ^ chunk_swz<CPO>(pix))) * 16 + (pp & 1) * 8;
            }
    }
    // The transposed reads go through inline asm on purpose: for an LDS load with a typed
    // address-space-3 pointer hipcc's waitcnt pass cannot prove the read does not alias the
    // in-flight LDS-DMA writes and inserts s_waitcnt vmcnt(0) in front of it, draining the whole
    // prefetch queue every row.  The skip pair read here was retired by the counted wait of the
    // previous step.  The asm loads are waited for by skip_wait() (names every destination).
    using i32x2 = __attribute__((ext_vector_type(2))) int;
    auto tr_read = [&](const char* p) __attribute__((always_inline)) -> i32x2 {
        i32x2 v;
        const unsigned addr = static_cast<unsigned>(reinterpret_cast<uintptr_t>((__attribute__((address_space(3))) const char*)p));
        asm volatile("ds_read_b64_tr_b16 %0, %1" : "=v"(v) : "v"(addr));
        return v;
    };

    // ---- pooling state
    // vertical pair sums: q_j = v_{j-1} + v_j in fp32 (hprev = previous row), rounded to fp16 pairs;
    // a 4-row window is q_{j-2} + q_j (stride 1: q ring by row parity; stride 2: pairs of even/odd rows)
    float hprev[16], q0f[16];    // q0f: fp32 pair-sum ring of the DPP (stride 2) variant
#pragma unroll
    for (int g = 0; g < 16; ++g) hprev[g] = q0f[g] = 0.f;
    i32x4 qp0[2] = {{0, 0, 0, 0}, {0, 0, 0, 0}}, qp1[2] = {{0, 0, 0, 0}, {0, 0, 0, 0}};
    i32x4 hprevp[2] = {{0, 0, 0, 0}, {0, 0, 0, 0}};    // POOLM: previous row's ReLU6 output as fp16 pairs
    // band matrix Pm[x][xo] as the pool MFMA's B operand: lane (xo = r, k-group hh), K slot (chunk c, j) is the
    // conv column held by accumulator register 8c + j of the transposed conv tile: x = (j & 3) + 8 (j >> 2) + 16 c + 4 hh
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
                const bool in = PK && x >= PS * r && x < PS * r + 4;
                w |= (in ? 0x3C00u : 0u) << (16 * e2);      // fp16 1.0
            }
            pmw[c][d] = static_cast<int>(w);
        }
    f32x16 acc0, acc1;
#pragma unroll
    for (int g = 0; g < 16; ++g) acc0[g] = acc1[g] = 0.f;

    wait_vmcnt<0>();
    lds_barrier();

    if constexpr (C::S0HW) {
        if (helper) {
            // stage-0 helper: one ring row per step, in step with the tile waves' barriers (nconv + 1 of them: the first
            // step, nconv - 1 full steps, the drain step); step k produces ring row k + RW_AHEAD, read from step k + 1 on
            auto hstep = [&](auto PC, int k) __attribute__((always_inline)) {
                constexpr int P = decltype(PC)::value;
                s0_front(IC<(P + RW_AHEAD + 4) % 4>{}, k + RW_AHEAD + 4);
                s0_part(IC<0>{}, (P + RW_AHEAD) % RW_NSLOT);
                s0_part(IC<1>{}, (P + RW_AHEAD) % RW_NSLOT);
                s0_part(IC<2>{}, (P + RW_AHEAD) % RW_NSLOT);
                asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
                raw_barrier();
            };
            int k = 0;
            for (; k + 3 < nconv; k += 4) {
                hstep(IC<0>{}, k);
                hstep(IC<1>{}, k + 1);
                hstep(IC<2>{}, k + 2);
                hstep(IC<3>{}, k + 3);
            }
            if (k < nconv) hstep(IC<0>{}, k++);
            if (k < nconv) hstep(IC<1>{}, k++);
            if (k < nconv) hstep(IC<2>{}, k++);
            raw_barrier();          // the tile waves' drain step
            return;
        }
    }

    // B fragment of K-chunk kc for the conv row whose first input row sits in ring slot P
    // (compiler-visible load; used by the 8-channel variant whose tap row is lane dependent)
    auto b_frag = [&](auto PC, auto KCC) __attribute__((always_inline)) -> i32x4 {
        constexpr int P = decltype(PC)::value, kc = decltype(KCC)::value;
        if constexpr (CIN >= 16) {
            constexpr int tap = kc / (CIN / 16), cc = kc % (CIN / 16);
            constexpr int ky = tap / 3, kx = tap % 3;
            return *reinterpret_cast<const i32x4*>(ring + ((P + ky) % RW_NSLOT) * ROWB + boff[kx][cc]);
        } else {
            int slot = P + b8_ky[kc];
            slot = slot >= RW_NSLOT ? slot - RW_NSLOT : slot;
            return *reinterpret_cast<const i32x4*>(ring + slot * ROWB + b8_off[kc]);
        }
    };
    // Explicit B-fragment pipeline (CIN >= 16): hipcc keeps these LDS reads only ONE K-chunk ahead
    // of their MFMA and waits lgkmcnt(0) in front of every MFMA, which exposes the LDS latency 18
    // times per row.  Here the reads are inline asm, BAHEAD chunks ahead, retired by counted
    // s_waitcnt lgkmcnt(N) (LDS returns in order; any compiler-issued DS op in between only makes
    // the count conservative).  Ring offsets beyond the 16-bit DS immediate use a second base.
    constexpr int BAHEAD = KC >= 6 ? 4 : 1;
    constexpr int SLOT_SPLIT = 65535 / ROWB >= RW_NSLOT ? RW_NSLOT : 65535 / ROWB;   // slots reachable from base 0
    unsigned bbase0[3][CIN >= 16 ? CIN / 16 : 1], bbase1[3][CIN >= 16 ? CIN / 16 : 1];
    if constexpr (CIN >= 16) {
        const unsigned ring_lds = static_cast<unsigned>(reinterpret_cast<uintptr_t>((__attribute__((address_space(3))) char*)ring));
#pragma unroll
        for (int kx = 0; kx < 3; ++kx)
#pragma unroll
            for (int cc = 0; cc < CIN / 16; ++cc) {
                bbase0[kx][cc] = ring_lds + static_cast<unsigned>(boff[kx][cc]);
                bbase1[kx][cc] = bbase0[kx][cc] + static_cast<unsigned>(SLOT_SPLIT * ROWB);
            }
    }
    // `dep` is a scheduling tie only (the instruction does not read it): naming the accumulator of the
    // MFMA BAHEAD chunks back as an input keeps hipcc from hoisting every read of the row to the top
    // of the step, which made all KC fragments live at once (72-144 VGPRs, spilled to AGPRs).
    auto b_read_asm = [&](auto PC, auto KCC, float dep) __attribute__((always_inline)) -> i32x4 {
        constexpr int P = decltype(PC)::value, kc = decltype(KCC)::value;   // kc: global K-chunk
        constexpr int tap = kc / (CIN / 16), cc = kc % (CIN / 16);
        constexpr int ky = tap / 3, kx = tap % 3;
        constexpr int slot = (P + ky) % RW_NSLOT;
        i32x4 v;
        if constexpr (slot < SLOT_SPLIT)
            asm volatile("ds_read_b128 %0, %1 offset:%2" : "=v"(v) : "v"(bbase0[kx][cc]), "n"(slot * ROWB), "v"(dep));
        else
            asm volatile("ds_read_b128 %0, %1 offset:%2"
                         : "=v"(v)
                         : "v"(bbase1[kx][cc]), "n"((slot - SLOT_SPLIT) * ROWB), "v"(dep));
        return v;
    };
    auto b8_issue = [&](auto PC) __attribute__((always_inline)) {
        constexpr int P = decltype(PC)::value;
        if constexpr (B8_ASM) {
            static_assert(!B8_ASM || KC == 5, "taps 0..8 in five chunks");
            [&]<int... KCI>(std::integer_sequence<int, KCI...>) {
                (([&] {
                     auto& pre = b8_pre;            // (asm operands of a nested lambda must name its own locals)
                     auto& base = b8_base;
                     auto& k1 = b8_k1;
                     if constexpr (KCI == 1) {
                         asm volatile("ds_read_b128 %0, %1" : "=v"(pre[1]) : "v"(k1[P]));
                     } else {
                         constexpr int ky = (2 * KCI > 8 ? 8 : 2 * KCI) / 3;
                         asm volatile("ds_read_b128 %0, %1 offset:%2" : "=v"(pre[KCI]) : "v"(base[KCI]), "n"(((P + ky) % RW_NSLOT) * ROWB));
                     }
                 }()),
                 ...);
            }(std::make_integer_sequence<int, KC>{});
        }
    };
    // MFMA chain of one conv row over the K-chunks [KB, KB + KCW) (KB = 0 unless K is split)
    // `slot(IC<I>)` runs right behind MFMA I: the epilogue micro-ops of that chain slot (see step())
    auto mma_chain = [&](auto PC, auto KBC, f32x16& acc, auto&& slot) __attribute__((always_inline)) {
        constexpr int KB = decltype(KBC)::value;
        const f32x16 zero = {0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f};
        i32x4 bq[KCW];
        if constexpr (CIN >= 16) {
            [&]<int... I>(std::integer_sequence<int, I...>) {
                ((bq[I] = b_read_asm(PC, IC<KB + I>{}, 0.f)), ...);
            }(std::make_integer_sequence<int, BAHEAD>{});
            [&]<int... I>(std::integer_sequence<int, I...>) {
                (([&] {
                     if constexpr (I + BAHEAD < KCW)
                         bq[I + BAHEAD] = b_read_asm(PC, IC<KB + (I + BAHEAD < KCW ? I + BAHEAD : 0)>{}, I == 0 ? 0.f : acc[0]);
                     constexpr int newer = (KCW - 1 - I) < BAHEAD ? (KCW - 1 - I) : BAHEAD;   // my reads issued after chunk I
                     asm volatile("s_waitcnt lgkmcnt(%1)" : "+v"(bq[I]) : "n"(newer));
                     if constexpr (C::POOLM)
                         acc = mfma32<DT>(bq[I], wreg[I], I == 0 ? zero : acc);   // D'[pixel][cout]
                     else
                         acc = mfma32<DT>(wreg[I], bq[I], I == 0 ? zero : acc);   // D[cout][pixel]
                     slot(IC<I>{});
                 }()),
                 ...);
            }(std::make_integer_sequence<int, KCW>{});
        } else if constexpr (B8_ASM) {
            // fragments were issued by b8_issue() at the top of the step
            [&]<int... I>(std::integer_sequence<int, I...>) {
                (([&] {
                     auto& pre = b8_pre;
                     if constexpr (I == 0)
                         asm volatile("s_waitcnt lgkmcnt(0)" : "+v"(pre[0]), "+v"(pre[1]), "+v"(pre[2]), "+v"(pre[3]), "+v"(pre[4]));
                     acc = C::POOLM ? mfma32<DT>(pre[I], wreg[I], I == 0 ? zero : acc) : mfma32<DT>(wreg[I], pre[I], I == 0 ? zero : acc);
                     slot(IC<I>{});
                 }()),
                 ...);
            }(std::make_integer_sequence<int, KC>{});
        } else {
            bq[0] = b_frag(PC, IC<0>{});
            [&]<int... I>(std::integer_sequence<int, I...>) {
                ((bq[I + 1 < KC ? I + 1 : 0] = (I + 1 < KC ? b_frag(PC, IC<(I + 1 < KC ? I + 1 : 0)>{}) : bq[0]),
                  acc = C::POOLM ? mfma32<DT>(bq[I], wreg[I], I == 0 ? zero : acc) : mfma32<DT>(wreg[I], bq[I], I == 0 ? zero : acc),
                  slot(IC<I>{})),
                 ...);
            }(std::make_integer_sequence<int, KC>{});
        }
    };
    auto mma_row = [&](auto PC, f32x16& acc, auto&& slot) __attribute__((always_inline)) {
        if constexpr (KS == 1) {
            mma_chain(PC, IC<0>{}, acc, slot);
        } else {
            // wave-uniform dispatch on the kernel row: ring slot and tap offsets stay compile-time
            if (ks == 0)
                mma_chain(PC, IC<0>{}, acc, slot);
            else if (ks == 1)
                mma_chain(PC, IC<KCW>{}, acc, slot);
            else
                mma_chain(PC, IC<2 * KCW>{}, acc, slot);
        }
    };
    // K split: LDS exchange of partial accumulators, double-buffered by step parity
    const unsigned part_lds = static_cast<unsigned>(reinterpret_cast<uintptr_t>(
        (__attribute__((address_space(3))) char*)(smem + C::PART_OFF))) + lane * 16;
    auto part_addr = [&](int parity, int k1) __attribute__((always_inline)) -> unsigned {   // k1 = ks - 1 of the writer
        return part_lds + static_cast<unsigned>(((parity * NPT * CT + (ct * NPT + pt)) * (KS - 1) + k1) * C::PART_B);
    };
    auto part_write = [&](int parity, const f32x16& acc_in) __attribute__((always_inline)) {
        const unsigned ad = part_addr(parity, ks - 1);
        // The DS writes below are inline asm and read registers the MFMA chain has just written:
        // hipcc inserts no XDL-write -> DS-read wait states around asm, so spend them explicitly
        // (16-pass MFMA: 18 states; two s_nop 15 are 32).  The "+v" operand ties the statement to the
        // accumulator registers -- a memory clobber alone does not keep the MFMAs in front of it.
        f32x16 acc = acc_in;
        asm volatile("s_nop 15\n\ts_nop 15" : "+v"(acc));
#pragma unroll
        for (int g = 0; g < 4; ++g) {
            const i32x4 v = {__float_as_int(acc[4 * g]), __float_as_int(acc[4 * g + 1]), __float_as_int(acc[4 * g + 2]),
                             __float_as_int(acc[4 * g + 3])};
            switch (g) {
                case 0: asm volatile("ds_write_b128 %0, %1\n\ts_nop 1" ::"v"(ad), "v"(v) : "memory"); break;
                case 1: asm volatile("ds_write_b128 %0, %1 offset:1024\n\ts_nop 1" ::"v"(ad), "v"(v) : "memory"); break;
                case 2: asm volatile("ds_write_b128 %0, %1 offset:2048\n\ts_nop 1" ::"v"(ad), "v"(v) : "memory"); break;
                default: asm volatile("ds_write_b128 %0, %1 offset:3072\n\ts_nop 1" ::"v"(ad), "v"(v) : "memory"); break;
            }
        }
        asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
    };
    auto part_add = [&](int parity, f32x16& acc) __attribute__((always_inline)) {
#pragma unroll
        for (int k1 = 0; k1 < KS - 1; ++k1) {
            const unsigned ad = part_addr(parity, k1);
            i32x4 v0, v1, v2, v3;
            asm volatile("ds_read_b128 %0, %1" : "=v"(v0) : "v"(ad) : "memory");
            asm volatile("ds_read_b128 %0, %1 offset:1024" : "=v"(v1) : "v"(ad) : "memory");
            asm volatile("ds_read_b128 %0, %1 offset:2048" : "=v"(v2) : "v"(ad) : "memory");
            asm volatile("ds_read_b128 %0, %1 offset:3072" : "=v"(v3) : "v"(ad) : "memory");
            asm volatile("s_waitcnt lgkmcnt(0)" : "+v"(v0), "+v"(v1), "+v"(v2), "+v"(v3));
#pragma unroll
            for (int j = 0; j < 4; ++j) {
                acc[j] += __int_as_float(v0[j]);
                acc[4 + j] += __int_as_float(v1[j]);
                acc[8 + j] += __int_as_float(v2[j]);
                acc[12 + j] += __int_as_float(v3[j]);
            }
        }
    };

    // transposed LDS reads of the staged skip pair (residual), then R_lo / R_hi = Skip_lo/hi^T * Wx
    using TQ = i32x2[8];
    auto res_issue = [&](int skip_buf, TQ& t) __attribute__((always_inline)) {
        const char* sk0 = skipb + skip_buf * C::SKIPBUFB;
        const char* sk1 = sk0 + skipbytes;
#pragma unroll
        for (int q = 0; q < 4; ++q) {
            t[q] = tr_read(sk0 + a_off[q]);
            t[4 + q] = tr_read(sk1 + a_off[q]);
        }
    };
    auto res_mfma = [&](const TQ& t, f32x16& r_lo, f32x16& r_hi) __attribute__((always_inline)) {
        const f32x16 zero = {0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f};
        const i32x4 al0 = {t[0][0], t[0][1], t[1][0], t[1][1]}, al1 = {t[2][0], t[2][1], t[3][0], t[3][1]};
        const i32x4 ah0 = {t[4][0], t[4][1], t[5][0], t[5][1]}, ah1 = {t[6][0], t[6][1], t[7][0], t[7][1]};
        r_lo = mfma32<DT>(al0, bw_h[0], zero);
        r_hi = mfma32<DT>(ah0, bw_h[0], zero);
        if constexpr (C::RES_SPLIT) {
            r_lo = mfma32<DT>(al0, bw_l[0], r_lo);
            r_hi = mfma32<DT>(ah0, bw_l[0], r_hi);
        }
        r_lo = mfma32<DT>(al1, bw_h[1], r_lo);
        r_hi = mfma32<DT>(ah1, bw_h[1], r_hi);
        if constexpr (C::RES_SPLIT) {
            r_lo = mfma32<DT>(al1, bw_l[1], r_lo);
            r_hi = mfma32<DT>(ah1, bw_l[1], r_hi);
        }
    };

    // skip-pair bookkeeping: the pair for local output row e lives in buffer (e + 3k) mod 3
    // PS = 1: first issue (step 0, conv row 1, e = -2) -> buffer 1; first read (row 0, e = -3) -> 0
    // PS = 2: first issue (step 0, conv row 1, e = -1) -> buffer 2; first emit-phase read (row 1) -> 2
    int sbuf_issue = PS == 1 ? 1 : 2;
    int sbuf_read = PS == 1 ? 0 : 2;

#ifdef RN_STAMPS
    unsigned long long st_work = 0, st_dma = 0, st_bar = 0, st_chain = 0;
    const unsigned long long st_loop0 = stamp();
#endif
    // One pipeline step s (ring phase P = s mod RW_NSLOT): DMA for row s+RW_AHEAD, MFMA chain of conv row s,
    // epilogue of conv row s-1, counted wait, barrier.
    //
    // The epilogue is cut into micro-ops and micro-op k runs right behind chain MFMA floor(k * KCW / NM) -- in
    // SOURCE order, pinned by sched_barrier(0).  Explicit placement instead of scheduler hints because the
    // epilogue has MFMAs of its own (pooling, residual) in the middle of its dependency chain
    // (ReLU/pair sums -> pool MFMAs -> BN -> stores): left to sched_group_barrier they ended up behind the
    // whole conv chain, and everything after them ran un-overlapped.  A wave issues one instruction per ~4-5
    // cycles whatever its kind (tools/ubench/mfma_valu_overlap.hip), so each slot between two dependent
    // 32-cycle MFMAs holds ~6 of them (12 with two waves per SIMD taking turns on the matrix pipe).
    auto step = [&](auto PC, auto MMAC, auto EPIC, int s) __attribute__((always_inline)) {
        constexpr int P = decltype(PC)::value;
#ifdef RN_STAMPS
        const unsigned long long ts0 = stamp();
#endif
        constexpr bool MMA = decltype(MMAC)::value != 0, EPI = decltype(EPIC)::value != 0;
        constexpr int JP = (P + RW_NSLOT - 1) % RW_NSLOT;                     // phase of conv row s-1 (the epilogue's row)
        constexpr bool emit_phase = PK == 0 || PS == 1 || (JP & 1) == 1;
        constexpr bool RESW = RES && EPI && emit_phase;                        // this step adds a residual row
        f32x16& acc_new = (P & 1) == 0 ? acc0 : acc1;
        f32x16& acc_old = (P & 1) == 0 ? acc1 : acc0;
        const int j = s - 1;                         // conv row of the epilogue (local)
        const bool emit = PK ? j >= 3 : true;
        const int yo = yo0 + (PK ? (j - 3) / PS : j);   // emitted output row
        // bf16 instantiations of the stages whose output can be dithered (32+ channels in and out, residual or stride-2 pooling: the
        // last step of the 32-channel block and the two steps of the 64-channel block) store through v_cvt_sr_bf16_f32 -- with the row's
        // dither seed (StageArgs::dither) or the plain one (round half up); no branch, one instruction per value (rn_stage.h).  Every
        // other instantiation rounds to nearest even with one v_cvt_pk per pair: the 1.5 M values per image of the first block's first
        // step cost the stage 0 + 1 launch 0.02 ms as SR stores (round 6: 0.198-0.21 -> 0.22-0.236 ms), more than their dither returned
        constexpr bool SRP = DT == RN_DTYPE_BF16 && CIN >= 32 && COUT >= 32 && (RES || PS == 2);
        [[maybe_unused]] const unsigned seed_out = a.dither ? rn_dither_seed(max(yo, 0)) : RN_SEED_PLAIN;
        float yl = 0.f;
        if constexpr (RESW) {
            const float src = mul_rounded(static_cast<float>(yo), a.rscale);
            yl = src - static_cast<float>(static_cast<int>(src));
        }
        // ---- state of this row's epilogue, shared by its micro-ops
        TQ tq;
        f32x16 H, r_lo, r_hi;
        i32x4 qp[2];
        uint2 pk[4];
        // micro-op list: residual reads | 8 x front (2 accumulator registers each) | residual MFMAs | pool MFMAs |
        // 2 idle (MFMA latency) | 2 x NG BN halves (2 channels each) | swaps + stores
        constexpr int M_RES_RD = 0, M_FRONT = 1, NF = 8, M_RES_MM = M_FRONT + NF, M_POOL = M_RES_MM + 1, M_BN = M_POOL + 3,
                      M_STORE = M_BN + 2 * NG, NM = M_STORE + 1;
        float yv[16];
        const bool epi_wave = KS == 1 || ks == 0;        // K split: only the wave of kernel row 0 owns the epilogue
        // explicit placement where the epilogue has MFMAs of its own; the plain VALU epilogue of the stride-2
        // non-residual stage does better under hipcc's own interleave (sched_group_barrier hints): 0.375 vs 0.39 ms
        constexpr bool SLICED = KS == 1 && !(C::DPP2 && !RES);
        auto mop = [&](auto KK) __attribute__((always_inline)) {
            constexpr int k = decltype(KK)::value;
            if constexpr (k == M_RES_RD) {
                if constexpr (KS > 1) part_add((P + 1) & 1, acc_old);
                if constexpr (RESW) res_issue(sbuf_read, tq);
            } else if constexpr (k >= M_FRONT && k < M_FRONT + NF) {
                if constexpr (C::DPP2) {
                    // stride 2, DPP variant: windows start at even conv rows and end at odd rows j = 2e + 3; the odd
                    // rows sum vertically first, then run the horizontal half on the 4-row sums
                    constexpr int i2 = 2 * (k - M_FRONT);
                    if constexpr ((JP & 1) == 1) {
                        float t[2], u[2];
#pragma unroll
                        for (int jj = 0; jj < 2; ++jj) {
                            const float pq = hprev[i2 + jj] + relu6f(acc_old[i2 + jj]);
                            t[jj] = q0f[i2 + jj] + pq;
                            q0f[i2 + jj] = pq;
                        }
#pragma unroll
                        for (int jj = 0; jj < 2; ++jj) u[jj] = t[jj] + (C::WIDE2 ? lane_next(t[jj]) : row_next<1>(t[jj]));
#pragma unroll
                        for (int jj = 0; jj < 2; ++jj) H[i2 + jj] = u[jj] + (C::WIDE2 ? lane_next(lane_next(u[jj])) : row_next<2>(u[jj]));
                    } else {
#pragma unroll
                        for (int i = i2; i < i2 + 2; ++i) hprev[i] = relu6f(acc_old[i]);
                    }
                }
                // ReLU6 + vertical pair sums of accumulator registers 4i .. 4i+3, rounded to two fp16 pairs
                if constexpr (C::POOLM) {
                    // ReLU6 -> fp16 pair -> vertical pair sum q_j = v_{j-1} + v_j as one packed fp16 add (the previous
                    // row is kept as fp16 pairs: 8 registers instead of 16)
                    constexpr int i2 = 2 * (k - M_FRONT);
                    const int vp = static_cast<int>(pack2_relu6_sixth(acc_old[i2], acc_old[i2 + 1]));
                    qp[i2 / 8][(i2 % 8) / 2] = pk_add_f16(hprevp[i2 / 8][(i2 % 8) / 2], vp);
                    hprevp[i2 / 8][(i2 % 8) / 2] = vp;
                }
            } else if constexpr (k == M_RES_MM) {
                if constexpr (RESW) {
                    // the transposed reads were issued a few chain waits ago: LDS returns in order, they have landed
                    if constexpr (!MMA) asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
                    asm volatile("" : "+v"(tq[0]), "+v"(tq[1]), "+v"(tq[2]), "+v"(tq[3]), "+v"(tq[4]), "+v"(tq[5]), "+v"(tq[6]), "+v"(tq[7]));
                    res_mfma(tq, r_lo, r_hi);
                }
            } else if constexpr (k == M_POOL) {
                if constexpr (C::POOLM && (PS == 1 || (JP & 1) == 1)) {
                    // stride 1: window rows j-3..j = q_{j-2} + q_j, ring by parity; stride 2 (odd rows only): rows
                    // 2e..2e+3 = (pair of the previous odd row) + (this pair)
                    const f32x16 zero = {0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f};
                    i32x4(&qold)[2] = (PS == 2 || (JP & 1) == 0) ? qp0 : qp1;
                    H = mfma32<RN_DTYPE_F16>(qold[0], pmw[0], zero);
                    H = mfma32<RN_DTYPE_F16>(qold[1], pmw[1], H);
                    H = mfma32<RN_DTYPE_F16>(qp[0], pmw[0], H);
                    H = mfma32<RN_DTYPE_F16>(qp[1], pmw[1], H);
                    qold[0] = qp[0];
                    qold[1] = qp[1];
                }
            } else if constexpr (k >= M_BN && k < M_BN + 2 * NG) {
                if constexpr (emit_phase) {
                    constexpr int g = (k - M_BN) / 2, h2 = (k - M_BN) % 2;    // channel group, half (channels 2 h2, 2 h2 + 1)
                    [[maybe_unused]] f32x4 sc1, sh1, sc2, sh2;
                    if constexpr (C::PTAB_REGS) {
                        sc1 = sc1r[g];
                        sh1 = sh1r[g];
                        if constexpr (RES) {
                            sc2 = sc2r[g];
                            sh2 = sh2r[g];
                        }
                    } else {
                        const float* pt_g = ptab_lane + 8 * g;
                        sc1 = *reinterpret_cast<const f32x4*>(pt_g);
                        sh1 = *reinterpret_cast<const f32x4*>(pt_g + COUT);
                        if constexpr (RES) {
                            sc2 = *reinterpret_cast<const f32x4*>(pt_g + 2 * COUT);
                            sh2 = *reinterpret_cast<const f32x4*>(pt_g + 3 * COUT);
                        }
                    }
#pragma unroll
                    for (int jj = 2 * h2; jj < 2 * h2 + 2; ++jj) {
                        const float S = PK ? H[4 * g + jj] : relu6f(acc_old[4 * g + jj]);
                        float y = fmaf(S, sc1[jj], sh1[jj]);
                        if constexpr (RES) {
                            const float lo = r_lo[4 * g + jj];
                            const float rs = lo + (r_hi[4 * g + jj] - lo) * yl;
                            y = fmaf(rs, sc2[jj], y);      // tables 0/1 already carry the second BN (see rn_fused_prepare)
                        }
                        yv[4 * g + jj] = y;
                    }
                    if constexpr (h2 == 0)
                        pk[g].x = SRP ? pack2_sr_bf16(yv[4 * g], yv[4 * g + 1], seed_out) : pack2<DT>(yv[4 * g], yv[4 * g + 1]);
                    else
                        pk[g].y = SRP ? pack2_sr_bf16(yv[4 * g + 2], yv[4 * g + 3], seed_out) : pack2<DT>(yv[4 * g + 2], yv[4 * g + 3]);
                }
            } else if constexpr (k == M_STORE) {
                if constexpr (emit_phase) {
                    // half-wave swap: lower half-wave gets channels 8k..8k+7, upper 8(k+1)..8(k+1)+7
                    i32x4 vv[2];
#pragma unroll
                    for (int kk = 0; kk < NG; kk += 2) {
                        const auto sx = __builtin_amdgcn_permlane32_swap(pk[kk].x, pk[kk + 1].x, false, false);
                        const auto sy = __builtin_amdgcn_permlane32_swap(pk[kk].y, pk[kk + 1].y, false, false);
                        vv[kk / 2][0] = static_cast<int>(sx[0]);
                        vv[kk / 2][1] = static_cast<int>(sy[0]);
                        vv[kk / 2][2] = static_cast<int>(sx[1]);
                        vv[kk / 2][3] = static_cast<int>(sy[1]);
                    }
                    const __amdgpu_buffer_rsrc_t rs = out_row_rsrc(yo);
                    if constexpr (C::STAGE_OUT) {
                        // transpose through the wave's staging tile, then lane-linear 16-byte stores.
                        // LDS operations of one wave execute in order: no barrier needed.  Inline asm keeps
                        // hipcc from guarding these DS ops with vmcnt(0) against the in-flight LDS-DMA.
                        // (lanes without an output write a pixel slot nobody reads: no predicate needed)
                        // (leading s_nop: the data registers were just written by v_permlane32_swap, and hipcc pads
                        //  no hazards between its own instructions and the inside of an asm string)
                        asm volatile("s_nop 1\n\tds_write_b128 %0, %1\n\ts_nop 1" ::"v"(st_w0), "v"(vv[0]) : "memory");
                        asm volatile("s_nop 1\n\tds_write_b128 %0, %1\n\ts_nop 1" ::"v"(st_w0 ^ 32u), "v"(vv[1]) : "memory");
                        i32x4 o0, o1;
                        asm volatile("ds_read_b128 %0, %1" : "=v"(o0) : "v"(st_r0) : "memory");
                        asm volatile("ds_read_b128 %0, %1 offset:1024" : "=v"(o1) : "v"(st_r0) : "memory");
                        asm volatile("s_waitcnt lgkmcnt(0)" : "+v"(o0), "+v"(o1));
                        __builtin_amdgcn_raw_buffer_store_b128(o0, rs, emit ? voff_st0 : OOB, 0, 0);
                        __builtin_amdgcn_raw_buffer_store_b128(o1, rs, emit ? voff_st1 : OOB, 0, 0);
                    } else {
                        const int vo = emit ? voff_lane : OOB;
#pragma unroll
                        for (int kk = 0; kk < NG; kk += 2)
                            __builtin_amdgcn_raw_buffer_store_b128(vv[kk / 2], rs, vo + 16 * kk, 0, 0);
                    }
                }
            }
        };
        // micro-ops of chain slot I (the ones with floor(k * KCW / NM) == I)
        // DMA pieces of this step (input row s + AHEAD, then the skip pair), spread over the chain slots: issued in
        // one burst at the top of the step -- every wave of the CU at once, next to the burst of fragment reads --
        // each global_load_lds held its wave for 100+ cycles
        constexpr bool SKIPW = RES && MMA && (PS == 1 || (P & 1) == 0);         // this step issues a skip pair
        constexpr int NPIECE = LPT + (SKIPW ? SLPT : 0);
        constexpr int DSTEP = (KCW - 1) / NPIECE >= 1 ? (KCW - 1) / NPIECE : 1;
        const int sbuf_now = sbuf_issue;
        const SkipRows skr = skip_rows((s - 2) / PS);
        const char* const in_row_next = in_img + static_cast<int64_t>(yc0 + min(s + RW_AHEAD, nin - 1)) * in_row_bytes;
        auto slot = [&](auto II) __attribute__((always_inline)) {
            constexpr int I = decltype(II)::value;
            if constexpr (MMA && RN_SPREAD_DMA && !C::S0F) {
                [&]<int... PI>(std::integer_sequence<int, PI...>) {
                    (([&] {
                         constexpr int at = PI * DSTEP < KCW ? PI * DSTEP : KCW - 1;
                         if constexpr (at == I) {
                             if constexpr (PI < LPT)
                                 issue_row_piece_at(IC<PI>{}, in_row_next, (P + RW_AHEAD) % RW_NSLOT);
                             else
                                 issue_skip_piece(IC<(PI < LPT ? 0 : PI - LPT)>{}, skr, sbuf_now);
                         }
                     }()),
                     ...);
                }(std::make_integer_sequence<int, NPIECE>{});
            }
            if constexpr (MMA && C::S0F && !C::S0HW && I < 3) s0_part(IC<(I < 3 ? I : 0)>{}, (P + RW_AHEAD) % RW_NSLOT);
            if constexpr (EPI && SLICED) {
                [&]<int... K>(std::integer_sequence<int, K...>) {
                    (([&] {
                         if constexpr (K * KCW / NM == I) mop(IC<K>{});
                     }()),
                     ...);
                }(std::make_integer_sequence<int, NM>{});
            }
            if constexpr (SLICED) __builtin_amdgcn_sched_barrier(0);
        };
        if constexpr (MMA) b8_issue(PC);
        if constexpr (MMA) {
            // (past the end of the band the last row is fetched again into a free slot: the number of
            //  DMA pieces per step stays constant, so the counted waits and the code path do too)
            if constexpr (!RN_SPREAD_DMA && !C::S0F) issue_row(min(s + RW_AHEAD, nin - 1), (P + RW_AHEAD) % RW_NSLOT);
            // S0F: image row s + RW_AHEAD + 4 completes stage-0 output row s + RW_AHEAD -> the slot the DMA would fill
            if constexpr (C::S0F && !C::S0HW) s0_front(IC<(P + RW_AHEAD + 4) % 4>{}, s + RW_AHEAD + 4);      // the rest: chain slots 0..2
            if constexpr (RES && (PS == 1 || (P & 1) == 0)) {
                // pair for the epilogue of conv row s+1 (runs in step s+2): e = (s + 1 - 3) / PS
                // (stride 2: only odd conv rows emit, so pairs are issued on even steps)
                static_assert(!RES || PK == 4, "residual variant pools");
                if constexpr (!RN_SPREAD_DMA) issue_skip((s - 2) / PS, sbuf_issue);
                sbuf_issue = sbuf_issue == RW_SKIPBUF - 1 ? 0 : sbuf_issue + 1;
            }
#if defined(RN_STAMPS) && defined(RN_STAMP_CHAIN)
            const unsigned long long tc0 = stamp();
#endif
            mma_row(PC, acc_new, slot);
#if defined(RN_STAMPS) && defined(RN_STAMP_CHAIN)
            st_chain += stamp() - tc0;
#endif
            if constexpr (KS > 1) {
                // waves of kernel rows 1, 2 publish their partial sums for the epilogue of the next step
                if (ks > 0) part_write(P & 1, acc_new);
            }
        }
        if constexpr (EPI && (!MMA || !SLICED)) {
            // drain step (no chain to hang the micro-ops on), or the K-split variant (three chain bodies, only
            // the wave of kernel row 0 finishes rows): the whole epilogue in one piece
            if (epi_wave) {
                [&]<int... K>(std::integer_sequence<int, K...>) { (mop(IC<K>{}), ...); }(std::make_integer_sequence<int, NM>{});
            }
        }
        if constexpr (MMA && EPI && !SLICED && KS == 1) {
            // software pipeline by hint: spread the VALU epilogue of row s-1 through the MFMA chain of row s
#pragma unroll
            for (int i = 0; i < KCW; ++i) {
                __builtin_amdgcn_sched_group_barrier(0x008, 1, 0);   // one MFMA
                __builtin_amdgcn_sched_group_barrier(0x002, 8, 0);   // VALU
            }
        }
        // the epilogue of this step handled conv row s-1: rotate the skip buffers after an emit phase
        if constexpr (RESW) sbuf_read = sbuf_read == RW_SKIPBUF - 1 ? 0 : sbuf_read + 1;
#ifdef RN_STAMPS
        const unsigned long long ts1 = stamp();
#endif
        if constexpr (MMA && !C::S0F) {
            // retire the DMA of input row s+3 (and of the skip pair the next epilogue reads)
            wait_vmcnt<C::vmcnt_steady(P)>();
        }
#ifdef RN_STAMPS
        const unsigned long long ts2 = stamp();
#endif
        if constexpr (C::S0SH) asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");     // this step's stage-0 row is in LDS
        if constexpr (!C::PRIV) raw_barrier();
#ifdef RN_STAMPS
        const unsigned long long ts3 = stamp();
        st_work += ts1 - ts0;
        st_dma += ts2 - ts1;
        st_bar += ts3 - ts2;
#endif
    };

    using T = IC<1>;
    using F = IC<0>;
    step(IC<0>{}, T{}, F{}, 0);
    int s = 1;
    // steady state: NSLOT steps per trip, ring phases 1, 2, ..., NSLOT-1, 0
    for (; s + RW_NSLOT - 1 < nconv; s += RW_NSLOT) {
        [&]<int... I>(std::integer_sequence<int, I...>) {
            (step(IC<(1 + I) % RW_NSLOT>{}, T{}, T{}, s + I), ...);
        }(std::make_integer_sequence<int, RW_NSLOT>{});
    }
    // remainder (s = 1 mod NSLOT here): phases 1, 2, ... while rows are left
    [&]<int... I>(std::integer_sequence<int, I...>) {
        ((s < nconv ? (step(IC<1 + I>{}, T{}, T{}, s), ++s, 0) : 0), ...);
    }(std::make_integer_sequence<int, RW_NSLOT - 1>{});
    // drain: epilogue of the last conv row
    [&]<int... I>(std::integer_sequence<int, I...>) {
        ((s % RW_NSLOT == I ? (step(IC<I>{}, F{}, T{}, s), 0) : 0), ...);
    }(std::make_integer_sequence<int, RW_NSLOT>{});
    wait_vmcnt<0>();   // the clamped re-fetches of the last steps are still in flight
#ifdef RN_CLOCK
    if (a.stamp_buf && tid == 0) {
        unsigned long long t1, r1;
        clock_pair(t1, r1);
        const int64_t wg = static_cast<int64_t>(blockIdx.y) * gridDim.x + blockIdx.x;
        a.stamp_buf[wg * 2 + 0] = t1 - ck_t0;
        a.stamp_buf[wg * 2 + 1] = r1 - ck_r0;
    }
#endif
#ifdef RN_STAMPS
    if (a.stamp_buf && lane == 0) {
        const int64_t w = (static_cast<int64_t>(blockIdx.y) * gridDim.x + blockIdx.x) * (NTHREADS / 64) + wave;
        a.stamp_buf[w * 4 + 0] = st_work;
        a.stamp_buf[w * 4 + 1] = st_dma;
        a.stamp_buf[w * 4 + 2] = st_bar | (st_chain << 32);
        // prologue (entry -> first step) in the upper half of slot 3, whole lifetime in the upper half of slot 1
        a.stamp_buf[w * 4 + 3] = static_cast<unsigned long long>(nconv) | ((st_loop0 - st_entry) << 32);
        a.stamp_buf[w * 4 + 1] = (st_dma & 0xffffffffull) | ((stamp() - st_entry) << 32);
#ifdef RN_STAMP_HWID
        // residency experiment: where and when did this wave run?  (HW_ID: wave slot, SIMD, CU, SH, SE, ...)
        unsigned hwid;
        asm volatile("s_getreg_b32 %0, hwreg(HW_REG_HW_ID)" : "=s"(hwid));
        unsigned xcc;
        asm volatile("s_getreg_b32 %0, hwreg(HW_REG_XCC_ID)" : "=s"(xcc));
        a.stamp_buf[w * 4 + 0] = st_entry;
        a.stamp_buf[w * 4 + 2] = stamp();
        a.stamp_buf[w * 4 + 3] = static_cast<unsigned long long>(nconv) | (static_cast<unsigned long long>(hwid) << 32);
        a.stamp_buf[w * 4 + 1] = xcc;
#endif
    }
#endif
}

template <int DT, int CIN, int COUT, int PK, int PS, bool RES, int NPT, int KS = 1, bool S0F = false, bool WIDE = false, int S0SH = 0>
int launch_rw(hipStream_t s, const StageArgs& a, dim3 grid) {
    using C = RwCfg<DT, CIN, COUT, PK, PS, RES, NPT, KS, S0F, WIDE, S0SH>;
    auto kern = stage_rw_kernel<DT, CIN, COUT, PK, PS, RES, NPT, KS, S0F, WIDE, S0SH>;
    // the attribute is per device: remember which devices of this process have it (one handle per GPU per process
    // is the normal deployment, several handles on several GPUs / threads in one process must work too)
    static std::atomic<unsigned long long> attr_devices{0};
    int dev = 0;
    RN_HIP(hipGetDevice(&dev));
    if (!(attr_devices.load(std::memory_order_acquire) >> (dev & 63) & 1ull)) {
        RN_HIP(hipFuncSetAttribute(reinterpret_cast<const void*>(kern), hipFuncAttributeMaxDynamicSharedMemorySize,
                                   160 * 1024));
        attr_devices.fetch_or(1ull << (dev & 63), std::memory_order_release);
    }
#ifdef RN_DIAG
    if (getenv("RN_DEBUG_OCC")) {
        int nb = -1;
        (void)hipOccupancyMaxActiveBlocksPerMultiprocessor(&nb, reinterpret_cast<const void*>(kern), C::NTHREADS, C::LDS_BYTES);
        hipFuncAttributes fa{};
        (void)hipFuncGetAttributes(&fa, reinterpret_cast<const void*>(kern));
        fprintf(stderr, "[occ] CIN %d COUT %d RES %d NPT %d: threads %d, LDS %d B, regs %d, static LDS %zu, max blocks per CU %d\n", CIN, COUT,
                int(RES), NPT, C::NTHREADS, C::LDS_BYTES, fa.numRegs, fa.sharedSizeBytes, nb);
    }
#endif
    hipLaunchKernelGGL(kern, grid, dim3(C::NTHREADS), C::LDS_BYTES, s, a);
    RN_CHECK_LAUNCH();
    return RN_OK;
}

template <int CIN, int COUT, int PK, int PS, bool RES, int NPT, int KS = 1, bool S0F = false, bool WIDE = false, int S0SH = 0>
int launch_rw_dt(int dtype, hipStream_t s, const StageArgs& a, dim3 grid) {
    if (dtype == RN_DTYPE_BF16) return launch_rw<RN_DTYPE_BF16, CIN, COUT, PK, PS, RES, NPT, KS, S0F, WIDE, S0SH>(s, a, grid);
    return launch_rw<RN_DTYPE_F16, CIN, COUT, PK, PS, RES, NPT, KS, S0F, WIDE, S0SH>(s, a, grid);
}

}  // namespace

// Geometry query for the rw kernels.  Returns false when no rw variant covers the stage, so
// the caller can use the generic kernel.
static bool rw_plan(int cin, int cout, int pool_k, int pool_s, bool res, int out_side, int skip_side, bool wide,
                    RwPlan* plan) {
    if (pool_k != 4 && pool_k != 0) return false;
    if (pool_k == 4 && pool_s != 1 && pool_s != 2) return false;
    const int ps = pool_k ? pool_s : 1;
    int variant = -1, npt = 0;
    const int nout_t = rw_tile_nout(pool_k, ps, wide), tstride = rw_tile_stride(pool_k, ps, wide);
    const int tiles = (out_side + nout_t - 1) / nout_t;
    // waves per workgroup = npt x cout tiles: 8 waves (two per SIMD, <= 256 registers) where the
    // weights are small, 4 or 2 waves (one per SIMD, whole register file) for K >= 576 / residual
    if (cin == 8 && cout == 32 && pool_k == 4 && ps == 1 && !res) variant = 0, npt = tiles > 4 ? 8 : 4;
    if (cin == 32 && cout == 32 && pool_k == 4 && ps == 1 && !res) variant = 1, npt = tiles > 4 ? 8 : 4;
    if (cin == 32 && cout == 32 && pool_k == 4 && ps == 1 && res) variant = 2, npt = 4;
    // 32->64: one "wide" pixel tile x two cout tiles per workgroup (2 waves; four workgroups per CU): 203 conv columns
    // are 7 wide tiles instead of 8 gapped ones -- measured 0.375 -> 0.360 ms at batch 256 (the 4-tile workgroups of
    // gapped tiles measured the same as 1- and 2-tile ones, so the decomposition itself is free)
    if (cin == 32 && cout == 64 && pool_k == 4 && ps == 2 && !res) variant = 3, npt = 1;
    if (cin == 64 && cout == 64 && pool_k == 4 && ps == 2 && res) variant = 4, npt = 2;
    if (cin == 64 && cout == 128 && pool_k == 0 && !res) variant = 5, npt = 1;
    if (cin == 128 && cout == 16 && pool_k == 4 && ps == 2 && !res) variant = 6, npt = 2;   // K split over 3 waves
    if (variant < 0) return false;
    plan->variant = variant;
    plan->npt = npt;
    plan->wide = wide;
    plan->wgs_per_cu = variant == 3 ? 4 : 1;        // 2-wave workgroups at <= 256 registers, 25 KB of LDS each
    plan->n_colblocks = (tiles + npt - 1) / npt;
    const int ringcols = (npt - 1) * tstride + 34;
    plan->skipcols = 0;
    plan->lds_bytes = 0;   // the kernel's LDS size is a compile-time constant of the variant
    if (res) {
        // columns of the skip tensor a column block touches: rlo[first] .. rhi[last]
        const float scale = static_cast<float>(skip_side) / static_cast<float>(out_side);
        const int cols_out = npt * nout_t;
        int skipcols = static_cast<int>(cols_out * scale) + 3;
        if (skipcols > skip_side) skipcols = skip_side;
        if (skipcols > ringcols + 8) return false;      // SKIPCOLS_MAX of the kernel
        plan->skipcols = skipcols;
        // the MFMA residual covers K = 32 source columns per tile: every valid lane's hi column
        // must lie within 31 of the tile's first lo column (same fp32 arithmetic as the tables)
        auto lo_of = [&](int i) { return static_cast<int>(static_cast<float>(i) * scale); };
        for (int t = 0; t < tiles; ++t) {
            const int first = t * nout_t;
            const int last = std::min(out_side - 1, first + nout_t - 1);
            const int hi_last = std::min(lo_of(last) + 1, skip_side - 1);
            if (hi_last - lo_of(first) > 31) return false;
        }
    }
    return true;
}

// Geometry query for the rw kernels (see above).  Stride-2 pooling stages can use gapped (14 windows) or wide (15 windows
// per tile, one DPP operation more per value) tiles: the 32->64 stage always runs wide, the 64->64 residual stage where
// that saves a column block and the residual's K = 32 window still covers a tile (600 x 600: 142 output columns = 11
// gapped tiles = 6 blocks of 2, or 10 wide tiles = 5 blocks; 224 x 224: 4 tiles either way -> gapped).
bool rn_rw_supported(int cin, int cout, int pool_k, int pool_s, bool res, int out_side, int skip_side,
                     RwPlan* plan) {
    const bool s2 = pool_k == 4 && pool_s == 2;
    if (s2 && !res && cin == 32 && cout == 64) return rw_plan(cin, cout, pool_k, pool_s, res, out_side, skip_side, true, plan);
    if (s2 && res && cin == 64 && cout == 64 && ((out_side + 14) / 15 + 1) / 2 < ((out_side + 13) / 14 + 1) / 2) {
        RwPlan w;
        if (rw_plan(cin, cout, pool_k, pool_s, res, out_side, skip_side, true, &w)) {
            *plan = w;
            return true;
        }
    }
    return rw_plan(cin, cout, pool_k, pool_s, res, out_side, skip_side, false, plan);
}

// column blocks of the shared-ring form of the fused stages 0 + 1 (variant 0, 8 tiles) for `out_side` output columns
int rn_rw_s0sh_colblocks(int out_side) { return (out_side + 226) / 227; }

// 1: shared stage-0 ring, every wave computes its own stage-0 tile (0.216-0.22 ms at batch 256).
// 2: + four stage-0 helper waves (12-wave workgroups, three waves per SIMD): bit-identical, but the stage-1 waves need ~200
//    registers and get 168 -- 46 spilled -- 0.227-0.234 ms.  Kept as a build switch: the direction needs a stage-1 wave
//    designed for 168 registers (NOTES.md).
#ifndef RN_S0_MODE
#define RN_S0_MODE 1
#endif
int rn_rw_launch(const RwPlan& p, int dtype, hipStream_t s, const StageArgs& a, dim3 grid) {
    switch (p.variant * 16 + p.npt) {
        case 0 * 16 + 4:
            if (a.s0_bgr) return launch_rw_dt<8, 32, 4, 1, false, 4, 1, true>(dtype, s, a, grid);
            return launch_rw_dt<8, 32, 4, 1, false, 4>(dtype, s, a, grid);
        case 0 * 16 + 8:
            // shared stage-0 ring: column blocks of 227 output columns (the caller sized n_colblocks for that, rn_rw_s0sh_colblocks)
            if (a.s0_bgr && !a.s0_private)
                return launch_rw_dt<8, 32, 4, 1, false, 8, 1, true, false, RN_S0_MODE>(dtype, s, a, grid);
            if (a.s0_bgr) return launch_rw_dt<8, 32, 4, 1, false, 8, 1, true>(dtype, s, a, grid);
            return launch_rw_dt<8, 32, 4, 1, false, 8>(dtype, s, a, grid);
        case 1 * 16 + 4: return launch_rw_dt<32, 32, 4, 1, false, 4>(dtype, s, a, grid);
        case 1 * 16 + 8: return launch_rw_dt<32, 32, 4, 1, false, 8>(dtype, s, a, grid);
        case 2 * 16 + 4: return launch_rw_dt<32, 32, 4, 1, true, 4>(dtype, s, a, grid);
        case 3 * 16 + 1: return launch_rw_dt<32, 64, 4, 2, false, 1, 1, false, true>(dtype, s, a, grid);
        case 4 * 16 + 2:
            if (p.wide) return launch_rw_dt<64, 64, 4, 2, true, 2, 1, false, true>(dtype, s, a, grid);
            return launch_rw_dt<64, 64, 4, 2, true, 2>(dtype, s, a, grid);
        case 5 * 16 + 1: return launch_rw_dt<64, 128, 0, 1, false, 1>(dtype, s, a, grid);
        case 6 * 16 + 2: return launch_rw_dt<128, 16, 4, 2, false, 2, 3>(dtype, s, a, grid);
        default:
            rn_set_error("rw kernel: no instantiation for variant %d npt %d", p.variant, p.npt);
            return RN_E_INVALID;
    }
}
