// Cross-stage fused kernel for a depth-3 conv_block's last two steps (reference network.py:183-203) on 16x16x32 matrix
// tiles -- the round-3 successor of stage23pc_kernel (rn_stage23.hip, 32x32x16 tiles), same data flow:
//
//   A (block's first BN output = the residual's skip tensor, [N, W, W, 32])
//     -> conv3x3 32->32 -> ReLU6 -> avg-pool 4/1 -> BN            = B   (never leaves the CU: LDS ring)
//     -> conv3x3 32->32 -> ReLU6 -> avg-pool 4/1 -> BN -> + legacy-bilinear(A) -> BN   = out [N, W-10, W-10, 32]
//
// Same workgroup geometry (image x band of rows x column block, 8 waves: 4 producers run the first stage, 4 consumers the
// second, wave w and w + 4 share a SIMD), same rings (A 4 rows, B 4 rows, wave-private 3-row skip rings), same step
// schedule (one A row in, B row t-5 finished, output row t-11 finished, one s_barrier) as rn_stage23.hip.  What changes:
//
//  * Pixels are tiled in 16s (v_mfma_f32_16x16x32_{bf16,f16}: one MFMA = 16 pixels x 16 couts x all 32 channels of one
//    tap).  A wave owns 3 or 4 ADJACENT tiles; the pooling window that crosses a tile border takes its right-hand columns
//    from the next tile's registers through a second band-matrix MFMA, so tiles inside a wave sit at stride 16 and only
//    the border between two waves costs the 3 overlap columns of the old stride-29 tiles.  Stage 2's 213 conv columns are
//    14 tiles (4 + 4 + 3 + 3 over the producers), stage 3's 208 are 14 (3 + 3 + 4 + 4 over the consumers): every SIMD hosts
//    7 tiles = 112 pixel columns per step where the 32-pixel form hosted 128.
//  * The two 16-cout halves of a tile are independent accumulators (two MFMA chains interleaved: no dependent-issue
//    stall), every epilogue object is half the size (a pooled tile is 8 registers, a residual pair 16), and their MFMA
//    latencies are half as long: the per-wave serial time of a step, which bounds the old kernel (each wave ~95 % busy
//    at 55 % matrix-pipe duty), shrinks with them.
//  * Cout order inside a tile is permuted on the host (fragment column n of half h = cout 8 (n / 4) + 4 h + n % 4), so a
//    lane ends up with 8 CONSECUTIVE couts of one pixel: one 16-byte ds_write / buffer_store per lane and tile, 1 KB
//    contiguous per wave instruction, no permlane swaps.
//  * LDS chunk swizzle (pixel >> 1) & 3: conflict-free for the 16x16x32 operand read at every tile alignment
//    (enumerated over ds_read_b128's lane groups; the 32-pixel form's (pixel >> 2) & 3 is 2-way conflicted here).
//
// Arithmetic: the same operations in the same order per output element as the 32x32x16 kernels; the matrix cores add the
// 288 products of one convolution output tap by tap, 32 channels at a time, where the 32x32x16 form adds them 16 at a
// time.  On gfx950 the two orders give the SAME bits: every test that compares the fused pair with one launch per stage
// bit for bit (224, 420 and 600 inputs, 1-4 column blocks, 1-73 bands) passes unchanged with this kernel.
#include "rn_fused.h"
#include "rn_stage.h"

#include <atomic>
#include <utility>

using namespace rnk;

namespace {

// the on-chip tensor's 16-bit store: never dithered, round to nearest even (one v_cvt_pk per pair) -- like the stage-launch arm's
// store of the same tensor (rn_stage_rw.hip: only the instantiations whose output can be dithered use the SR conversion)
template <int DT>
__device__ __forceinline__ unsigned pack2p(float a, float b) {
    return pack2<DT>(a, b);
}

// Cache policy (aux operand of the buffer store / LDS-DMA; 2 = nt): the output rows and the residual's second read of an A row are
// streamed -- neither is read again by this launch -- so that the L2 keeps the A rows between their first read (the conv operand)
// and their second (the skip row, 0-10 row steps later).  FETCH_SIZE of the launch 1 315 -> 928 MB per 256 images (757 = every A row
// once); the rate does not move (the pass is not bound by HBM; gpurun_out/r4/s17_nt.txt).
constexpr int RN_NT_OUT = 2, RN_NT_SKIP = 2;
constexpr int X_NA = 4, X_NB = 4, X_NSK = 3;     // ring depths: A rows, B rows, private skip rows
constexpr int X_WMIN = 193;
constexpr int X_WMAX = 215;                      // widest A row (column blocks: 193..215, rn_stage23_plan; the tail DMA piece needs W > 192)
constexpr int X_ROWA = X_WMAX * 64;              // bytes per A ring row (32 channels x 16 bit per pixel)
constexpr int X_BDUMMY = X_WMAX - 5;             // B ring column that invalid lanes write to (never read for a valid output)
constexpr int X_ROWB = (X_BDUMMY + 1) * 64;
constexpr int X_SKROW = 64 * 64;                 // one private skip row: 64 columns
constexpr int X_ROWB_N = (X_BDUMMY + 1) * 32;    // ... of the narrow B ring (16 channels per pixel)
constexpr int X_ROWB_8 = (X_BDUMMY + 1) * 16;    // ... of the eight-channel B ring (round 6: 24 constant channels)
constexpr int X_NTAB = 6 * 32;                   // folded BN tables: sc2, sh2 | sc3', sh3', sc4 | narrow form: the second conv's per-cout constant
constexpr int X_RINGA_OFF = 1024;
constexpr int X_RINGB_OFF = X_RINGA_OFF + X_NA * X_ROWA;
constexpr int X_SKIP_OFF = X_RINGB_OFF + X_NB * X_ROWB;
constexpr int X_LDS = X_SKIP_OFF + 4 * X_NSK * X_SKROW;
constexpr int X_LAG = 11;                        // step t finishes output row t - X_LAG
constexpr int X_KT = 9;                          // taps = K chunks of 32 (all channels of one tap)
static_assert(X_LDS <= 160 * 1024, "LDS budget");

// first conv column of each wave's tile run and its length in tiles; consecutive waves overlap by 3 columns (the last
// tile of a wave has no right neighbour in registers and yields 13 pooled columns)
__device__ __forceinline__ int xp_start(int w) { return w == 0 ? 0 : w == 1 ? 61 : w == 2 ? 122 : 167; }   // producers: 4 4 3 3 tiles
__device__ __forceinline__ int xc_start(int w) { return w == 0 ? 0 : w == 1 ? 45 : w == 2 ? 90 : 148; }    // consumers: 3 3 4 4 tiles
// (consumer 2 ends at column 147, not 150: a consumer interpolates from a 64-column private skip ring, enough for 58
//  output columns at the block's 215 -> 205 scale)

__device__ __forceinline__ unsigned lds_addr(const void* p) {
    return static_cast<unsigned>(reinterpret_cast<uintptr_t>((__attribute__((address_space(3))) const char*)p));
}

__device__ __forceinline__ int swzx(int pix) { return (pix >> 1) & 3; }   // A and B rings (16x16x32 operand reads)

using i32x2 = __attribute__((ext_vector_type(2))) int;

template <int DT>
__device__ __forceinline__ f32x4 mfma16(i32x4 a, i32x4 b, f32x4 c) {
    if constexpr (DT == RN_DTYPE_BF16)
        return __builtin_amdgcn_mfma_f32_16x16x32_bf16(__builtin_bit_cast(bf16x8, a), __builtin_bit_cast(bf16x8, b), c, 0, 0, 0);
    else
        return __builtin_amdgcn_mfma_f32_16x16x32_f16(__builtin_bit_cast(f16x8, a), __builtin_bit_cast(f16x8, b), c, 0, 0, 0);
}

// PH = halves of every 8-cout group the PRODUCER computes: 2 = all 32 channels of B; 1 = the first half only -- the other 16
// channels are frozen (Stage23Args::producer_halves: constants for every input, proven per channel by rn_fused_prepare) and are
// written from the table, the same bits the full computation stores.  Half of the first conv's matrix instructions (9 + 2 of
// 18 + 4 per tile) and half of its epilogue go; two adjacent tiles then share a chain so that it still alternates two
// independent accumulators.
// NB (with PH = 1) = the NARROW form of the B tensor: the ring holds only the 16 computed channels (32 bytes per pixel, ring
// channel 4 g + i = B position 8 g + i); the second conv contracts them two kernel columns at a time (K = 32 = [kx 0 | kx 1],
// then [kx 2 | zero weights]: 6 operand reads and 12 MFMAs per tile where the 32-channel ring takes 9 and 18) and starts its
// accumulators from the per-cout constant of the 16 frozen channels (Stage23Args::ptab row 5: sum over taps and frozen
// channels of weight x stored 16-bit value -- a VALID convolution sees every tap of every pixel, so the constant is exact).
// N8 (with NB; round 6) = EIGHT computed channels: rn_fused_prepare proves 24 channels of B constant in the handle's 16-bit store
// (the shipped checkpoint: 26 in bf16, 25 in fp16) and puts the live ones at B positions 0..3 and 8..11 -- the lane groups 0, 1 of
// the producer's half -- so the ring pixel is ONE 16-byte chunk and a K = 32 operand of the second conv is FOUR TAPS x 8 channels:
// lane group g of chunk c reads tap 4 c + g (each from its own ring row and column: the offsets sit in the lane's base), three
// chunks for the nine taps (the last three groups of chunk 2 carry zero weights) -- 3 operand reads and 6 MFMAs per tile where the
// 16-channel ring takes 5 and 10.
template <int DT, int PH, bool NB, bool N8 = false>
__global__ __launch_bounds__(512, 2) void stage23x_kernel(const Stage23Args a) {
    static_assert(!NB || PH == 1, "the narrow B ring holds the computed half only");
    static_assert(!N8 || NB, "the eight-channel ring is a narrow ring");
    constexpr int ROWB = N8 ? X_ROWB_8 : NB ? X_ROWB_N : X_ROWB;
// Registers: the kernel sits at 256 VGPRs with six spilled dwords, two of them reloaded inside the consumer's loop.  A build that
// keeps four of the consumers' weight fragments in LDS instead has no scratch access at all and is 3 % SLOWER (0.490 against
// 0.475 ms, one session, round 4; two fragments: equal); the reloads are not what bounds the kernel (NOTES.md).
    constexpr int AHEAD = 3;                              // operand reads in flight ahead of their MFMA pair
    extern __shared__ __attribute__((aligned(64))) char smem[];
    const int tid = threadIdx.x;
    const int lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int wq = wave & 3;
    const int px16 = lane & 15, g = lane >> 4;           // tile column / K group (conv operand), pixel / cout group (pooled tiles)
#ifdef RN_CLOCK
    unsigned long long ck_t0, ck_r0;
    clock_pair(ck_t0, ck_r0);
#endif
    const int cblk = blockIdx.x % a.n_cblocks, band = blockIdx.x / a.n_cblocks, n = blockIdx.y;
    const int Win = a.W, Hout = a.Wo, x0 = a.cb_x0[cblk];
    const int Wo = a.cb_wo[cblk], W = Wo + 10, Wb = W - 5;
    const int yo0 = band * a.rows_per_band;
    const int nrows = min(Hout, yo0 + a.rows_per_band) - yo0;
    const int nsteps = nrows + X_LAG;

    float* const tab = reinterpret_cast<float*>(smem);
    for (int i = tid; i < X_NTAB; i += 512) tab[i] = a.ptab[i];
    char* const ringA = smem + X_RINGA_OFF;
    char* const ringB = smem + X_RINGB_OFF;
    const unsigned ringA_lds = lds_addr(ringA), ringB_lds = lds_addr(ringB);
    const char* const in_img = reinterpret_cast<const char*>(a.in + static_cast<int64_t>(n) * Win * Win * 32);
    const char* const in_blk = in_img + x0 * 64;
    constexpr int OOB = 0x40000000;
    const f32x4 zero4 = {0.f, 0.f, 0.f, 0.f};

    // band matrices of the pooling MFMA.  H[cout][xo] = sum_k V[cout][k] * Pm[k][xo]: the A operand of lane (cout, g) is the
    // lane's own packed history -- K elements 8 g + e are pixel 4 g + (e & 3) of the vertical pair sums two rows back
    // (e < 4) and of the current row (e >= 4) -- and Pm[k][xo] = 1 where xo <= pixel < xo + 4.  pmx is the same for the
    // NEXT tile's registers (pixel 16 + 4 g + (e & 3)): the windows of columns 13..15 end there.
    i32x4 pm, pmx;
#pragma unroll
    for (int d = 0; d < 4; ++d) {
        unsigned w0 = 0, w1 = 0;
#pragma unroll
        for (int e2 = 0; e2 < 2; ++e2) {
            const int e = 2 * d + e2;
            const int p = 4 * g + (e & 3);
            w0 |= ((p >= px16 && p < px16 + 4) ? 0x3C00u : 0u) << (16 * e2);
            w1 |= ((16 + p >= px16 && 16 + p < px16 + 4) ? 0x3C00u : 0u) << (16 * e2);
        }
        pm[d] = static_cast<int>(w0);
        pmx[d] = static_cast<int>(w1);
    }
    asm volatile("" : "+v"(pm), "+v"(pmx));

    // two interleaved accumulation chains (the tile's two 16-cout halves) over the nine taps; the tap's operand is ONE
    // ds_read_b128 (16 pixels x 32 channels), issued AHEAD taps early and retired by a counted wait
    // `hook(IC<tap>)` runs behind the tap's MFMA pair: DMA issue and scalar bookkeeping ride in the MFMAs' shadow
    auto chain = [&](auto S0C, auto ROWC, auto KC, const unsigned (&base)[3], const i32x4 (&wr)[2 * X_KT], f32x4 (&acc)[2], auto&& hook) __attribute__((always_inline)) {
        constexpr int S0 = decltype(S0C)::value, ROW = decltype(ROWC)::value, k = decltype(KC)::value;
        i32x4 fq[X_KT];
        auto rd = [&](auto TC, float dep) __attribute__((always_inline)) -> i32x4 {
            constexpr int tap = decltype(TC)::value, ky = tap / 3, kx = tap % 3;
            constexpr int off = ((S0 + ky) % 4) * ROW + k * 1024;
            i32x4 v;
            asm volatile("ds_read_b128 %0, %1 offset:%2" : "=v"(v) : "v"(base[kx]), "n"(off), "v"(dep));
            return v;
        };
        [&]<int... I>(std::integer_sequence<int, I...>) { ((fq[I] = rd(IC<I>{}, 0.f)), ...); }(std::make_integer_sequence<int, AHEAD>{});
        [&]<int... I>(std::integer_sequence<int, I...>) {
            (([&] {
                 if constexpr (I + AHEAD < X_KT) fq[I + AHEAD] = rd(IC<(I + AHEAD < X_KT ? I + AHEAD : 0)>{}, I == 0 ? 0.f : acc[0][0]);
                 // one counted wait per tap (one per PAIR of taps measured slower, 0.48-0.50 against 0.475 ms: a wait that
                 // stalls costs more than its issue slot)
                 constexpr int newer = (X_KT - 1 - I) < AHEAD ? (X_KT - 1 - I) : AHEAD;
                 asm volatile("s_waitcnt lgkmcnt(%1)" : "+v"(fq[I]) : "n"(newer));
                 acc[0] = mfma16<DT>(fq[I], wr[2 * I], I == 0 ? zero4 : acc[0]);          // D'[pixel][cout], couts of half 0
                 acc[1] = mfma16<DT>(fq[I], wr[2 * I + 1], I == 0 ? zero4 : acc[1]);
                 hook(IC<I>{});
             }()),
             ...);
        }(std::make_integer_sequence<int, X_KT>{});
    };
    // ReLU6 -> fp16 pairs -> vertical pair sums; the pooling operand of the tile half = [pair sums two rows back | current]
    // Cut into six slices (per 16-cout half: pack rows 0-1, pack rows 2-3, pair sums + operand), so that the slices of tile k
    // can ride behind the MFMA pairs of tile k + 1's chain (the wave's other accumulator pair): with_finish() below.
    auto finish_part = [&](auto PARTC, auto PRC, const f32x4 (&acc)[2], int (&hp)[2][2], auto& q0, auto& q1, i32x4 (&op)[2],
                           int (&vt)[2][2]) __attribute__((always_inline)) {
        constexpr int part = decltype(PARTC)::value, PQ = decltype(PRC)::value, PR = PQ & 1, HI = (PQ >> 1) & 1;      // PRC = row mod 4
        constexpr int h = part / 3, sub = part % 3;
        if constexpr (sub == 0) vt[h][0] = static_cast<int>(pack2_relu6_sixth(acc[h][0], acc[h][1]));
        if constexpr (sub == 1) vt[h][1] = static_cast<int>(pack2_relu6_sixth(acc[h][2], acc[h][3]));
        if constexpr (sub == 2) {
            const int n0 = pk_add_f16(hp[h][0], vt[h][0]), n1 = pk_add_f16(hp[h][1], vt[h][1]);
            hp[h][0] = vt[h][0];
            hp[h][1] = vt[h][1];
            if constexpr (NB) {
                // the operand quad of this row parity lives in place: the pooling matrix weighs its two K halves alike (pm), so
                // the new pair sums overwrite the half that held the pair sums of four rows ago and nothing is moved (the
                // products are exact in fp32: the order of the halves does not change a bit)
                auto& Q = PR == 0 ? q0[h] : q1[h];
                Q[2 * HI] = n0;
                Q[2 * HI + 1] = n1;
                op[h] = Q;
            } else {
                auto& qold = PR == 0 ? q0[h] : q1[h];
                op[h] = i32x4{qold[0], qold[1], n0, n1};
                qold[0] = n0;
                qold[1] = n1;
            }
        }
    };
    auto finish = [&](auto PRC, const f32x4 (&acc)[2], int (&hp)[2][2], auto& q0, auto& q1, i32x4 (&op)[2]) __attribute__((always_inline)) {
        int vt[2][2];
        [&]<int... I>(std::integer_sequence<int, I...>) { (finish_part(IC<I>{}, PRC, acc, hp, q0, q1, op, vt), ...); }(std::make_integer_sequence<int, 6>{});
    };
    // chain hook = `base` (DMA / bookkeeping) + slice i - 1 of the previous tile's finish behind MFMA pair i = 1 .. 6
    auto with_finish = [&](auto&& base, auto PRC, const f32x4 (&acc)[2], int (&hp)[2][2], auto& q0, auto& q1, i32x4 (&op)[2],
                           int (&vt)[2][2]) __attribute__((always_inline)) {
        return [&, PRC](auto IC_) __attribute__((always_inline)) {
            constexpr int i = decltype(IC_)::value;
            base(IC_);
            if constexpr (i >= 1 && i <= 6) finish_part(IC<(i >= 1 && i <= 6 ? i - 1 : 0)>{}, PRC, acc, hp, q0, q1, op, vt);
        };
    };

    // ---- residual skip rows: schedule as in rn_stage23.hip (the partner producer fetches the regular new row during the
    // step before it is needed, the consumer itself the rare second one after the lo row jumped by 2)
    const int xs0 = a.rlo[x0 + min(xc_start(wq), Wo - 1)];     // first skip column of consumer wq's ring (full-width index)
    // piece i of a skip row = pixels xs0 + lane / 4 + 16 i: one lane offset, 1024 bytes per piece in the scalar base; pixels
    // right of the image read on into the next row / the slack behind the tensor and only feed columns that are not stored
    // (no chunk swizzle in the skip rings: a transposed read takes one 8-byte half of every 16-byte chunk it touches, the
    //  two 16-lane groups of a half-wave are 2-way conflicted whatever the chunk order -- 8 LDS cycles per tile)
    const unsigned sk_goff0 = static_cast<unsigned>((xs0 + (lane >> 2)) * 64 + ((lane & 3) << 4));
    char* const skw = smem + X_SKIP_OFF + wq * (X_NSK * X_SKROW);
    const unsigned skw_lds = lds_addr(skw);
    auto issue_skip_piece = [&](auto IC_, int y, int slot) __attribute__((always_inline)) {
        constexpr int i = decltype(IC_)::value;
        const char* row = in_img + static_cast<int64_t>(y) * static_cast<int64_t>(Win * 64);
        unsigned off = sk_goff0;
        asm volatile("" : "+v"(off));
        __builtin_amdgcn_global_load_lds(row + i * 1024 + off, (lds_void_ptr)(skw + slot * X_SKROW + i * 1024), 16, 0, RN_NT_SKIP);
    };
    auto issue_skip_row = [&](int y, int slot) __attribute__((always_inline)) {
        issue_skip_piece(IC<0>{}, y, slot);
        issue_skip_piece(IC<1>{}, y, slot);
        issue_skip_piece(IC<2>{}, y, slot);
        issue_skip_piece(IC<3>{}, y, slot);
    };
    auto no_hook = [](auto) __attribute__((always_inline)) {};
    struct VLerp {
        int ylo;
        float yl;
    };
    auto vlerp_of = [&](int yo) __attribute__((always_inline)) -> VLerp {
        const float src = mul_rounded(static_cast<float>(yo), a.rscale);
        const int ylo = static_cast<int>(src);
        VLerp v;
        v.ylo = __builtin_amdgcn_readfirstlane(ylo);
        v.yl = __int_as_float(__builtin_amdgcn_readfirstlane(__float_as_int(src - static_cast<float>(ylo))));
        return v;
    };
    int sk_f = vlerp_of(yo0).ylo - 1;                 // highest skip row fetched so far (by either role)
    int sk_slot = X_NSK - 1;                          // its ring slot (row y lives in slot (y - ylo(yo0)) mod 3)
    auto ylo_step = [&](int t) __attribute__((always_inline)) {
        return vlerp_of(yo0 + min(max(t - X_LAG, 0), nrows - 1)).ylo;
    };

    if (wave < 4) {
        // =============================================================== producer: first stage, A ring -> B ring
        const bool has4 = wq < 2;                                   // tiles of this wave: 4 4 3 3
        const int xw = xp_start(wq);
        i32x4 w2[2 * X_KT];
#pragma unroll
        for (int f = 0; f < 2 * X_KT; ++f) {
            if (PH == 1 && (f & 1)) continue;             // (fragment 2 tap + half: the second halves are not computed)
            const i32x4* src = a.wfrag2 + f * 64 + lane;
            asm volatile("global_load_dwordx4 %0, %1, off" : "=v"(w2[f]) : "v"(src) : "memory");
        }
        // ---- A rows: four DMA pieces per producer wave and row (the fourth is the masked W - 192 tail)
        const int ptid = wq * 64 + lane;
        const int tailn = W - 192;
        const unsigned long long tail_mask = (1ull << tailn) - 1ull;
        const unsigned ld_goff = static_cast<unsigned>((ptid >> 2) * 64 + (((ptid & 3) ^ swzx(ptid >> 2)) << 4));
        unsigned ld_goff_tail;
        {
            const int q = 768 + tailn * wq + min(lane, tailn - 1);
            const int p = q >> 2, c = q & 3;
            ld_goff_tail = static_cast<unsigned>(min(p, W - 1) * 64 + ((c ^ swzx(p)) << 4));
        }
        auto issue_A_piece = [&](auto IC_, const char* row, int slot) __attribute__((always_inline)) {
            constexpr int i = decltype(IC_)::value;
            if constexpr (i < 3) {
                unsigned off = ld_goff;
                asm volatile("" : "+v"(off));
                dma16(row + i * 4096 + off, ringA + slot * X_ROWA + (i * 256 + wq * 64) * 16);
            } else {
                unsigned off = ld_goff_tail;
                asm volatile("" : "+v"(off));
                dma16_masked(row + off, ringA + slot * X_ROWA + (768 + tailn * wq) * 16, tail_mask);
            }
        };
        auto issue_A_pieces = [&](const char* row, int slot) __attribute__((always_inline)) {
            issue_A_piece(IC<0>{}, row, slot);
            issue_A_piece(IC<1>{}, row, slot);
            issue_A_piece(IC<2>{}, row, slot);
            issue_A_piece(IC<3>{}, row, slot);
        };
        const char* a_next = in_blk + static_cast<int64_t>(yo0) * (Win * 64);
        unsigned baseA[3], wbB[4];
#pragma unroll
        for (int kx = 0; kx < 3; ++kx) {
            const int p = xw + px16 + kx;                 // (tile k adds 16 pixels = 1024 bytes: same swizzle; no clamp: columns right
            baseA[kx] = ringA_lds + static_cast<unsigned>(p * 64 + ((g ^ swzx(p)) << 4));   //  of the row only feed columns that are dropped)
        }
#pragma unroll
        for (int k = 0; k < 4; ++k) {
            const int xo = xw + 16 * k + px16;
            const bool last = k == (has4 ? 3 : 2);
            const int col = (xo < Wb && !(last && px16 >= 13) && (k < 3 || has4)) ? xo : X_BDUMMY;
            if constexpr (N8)       // lane groups 0, 1 hold the 8 live couts: 8 bytes at 8 g of the 16-byte pixel; groups 2, 3 write the dummy pixel
                wbB[k] = ringB_lds + static_cast<unsigned>((g < 2 ? col : X_BDUMMY) * 16 + 8 * (g & 1));
            else if constexpr (NB)  // 8 bytes (the lane's 4 computed couts) at 8 g of the 32-byte pixel (no swizzle: see baseN0)
                wbB[k] = ringB_lds + static_cast<unsigned>(col * 32 + 8 * g);
            else
                wbB[k] = ringB_lds + static_cast<unsigned>(col * 64 + ((g ^ swzx(col)) << 4));
        }
        // folded BN of the lane's 8 couts (8 g .. 8 g + 7)
        f32x4 scv[2], shv[2];
#pragma unroll
        for (int h = 0; h < 2; ++h) {
            scv[h] = *reinterpret_cast<const f32x4*>(a.ptab + 8 * g + 4 * h);
            shv[h] = *reinterpret_cast<const f32x4*>(a.ptab + 32 + 8 * g + 4 * h);
        }
        int hp[4][2][2];
        std::conditional_t<NB, i32x4[4][2], int[4][2][2]> q0, q1;      // pair sums of the last rows of either parity (NB: the operand quads)
#pragma unroll
        for (int k = 0; k < 4; ++k)
#pragma unroll
            for (int h = 0; h < 2; ++h)
#pragma unroll
                for (int j = 0; j < 2; ++j) {
                    hp[k][h][j] = 0;
                    if constexpr (NB) {
                        q0[k][h][2 * j] = q0[k][h][2 * j + 1] = q1[k][h][2 * j] = q1[k][h][2 * j + 1] = 0;
                    } else {
                        q0[k][h][j] = q1[k][h][j] = 0;
                    }
                }
        issue_A_pieces(a_next, 0);
        int ylo_nxt = ylo_step(0);                       // lo skip row of the output row the coming step finishes
        wait_vmcnt<0>();
#pragma unroll
        for (int f = 0; f < 2 * X_KT; ++f)
            if (PH == 2 || !(f & 1)) asm volatile("" : "+v"(w2[f]));
        lds_barrier();
        // PH == 1: the frozen channels of the lane's group (positions 8 g + 4 .. 8 g + 7) as the two dwords every store carries
        [[maybe_unused]] const int cdead0 = static_cast<int>(pack2p<DT>(shv[1][0], shv[1][1])), cdead1 = static_cast<int>(pack2p<DT>(shv[1][2], shv[1][3]));

        // ---- PH == 1: two adjacent tiles share a chain (one accumulator each: two independent MFMA chains, as the two halves of a
        // tile are in the full form); a tap costs two operand reads and two MFMAs
        [[maybe_unused]] auto chain2 = [&](auto S0C, auto K0C, auto K1C, f32x4& acc0, f32x4& acc1, auto&& hook) __attribute__((always_inline)) {
            constexpr int S0 = decltype(S0C)::value, k0 = decltype(K0C)::value, k1 = decltype(K1C)::value;
            i32x4 fa[X_KT], fb[X_KT];
            auto rd = [&](auto TC, auto KC_, float dep) __attribute__((always_inline)) -> i32x4 {
                constexpr int tap = decltype(TC)::value, ky = tap / 3, kx = tap % 3, k = decltype(KC_)::value;
                constexpr int off = ((S0 + ky) % 4) * X_ROWA + k * 1024;
                i32x4 v;
                asm volatile("ds_read_b128 %0, %1 offset:%2" : "=v"(v) : "v"(baseA[kx]), "n"(off), "v"(dep));
                return v;
            };
            [&]<int... I>(std::integer_sequence<int, I...>) { ((fa[I] = rd(IC<I>{}, IC<k0>{}, 0.f), fb[I] = rd(IC<I>{}, IC<k1>{}, 0.f)), ...); }(std::make_integer_sequence<int, AHEAD>{});
            [&]<int... I>(std::integer_sequence<int, I...>) {
                (([&] {
                     if constexpr (I + AHEAD < X_KT) {
                         fa[I + AHEAD] = rd(IC<(I + AHEAD < X_KT ? I + AHEAD : 0)>{}, IC<k0>{}, I == 0 ? 0.f : acc0[0]);
                         fb[I + AHEAD] = rd(IC<(I + AHEAD < X_KT ? I + AHEAD : 0)>{}, IC<k1>{}, I == 0 ? 0.f : acc1[0]);
                     }
                     constexpr int newer = (X_KT - 1 - I) < AHEAD ? (X_KT - 1 - I) : AHEAD;
                     asm volatile("s_waitcnt lgkmcnt(%2)" : "+v"(fa[I]), "+v"(fb[I]) : "n"(2 * newer));
                     acc0 = mfma16<DT>(fa[I], w2[2 * I], I == 0 ? zero4 : acc0);
                     acc1 = mfma16<DT>(fb[I], w2[2 * I], I == 0 ? zero4 : acc1);
                     hook(IC<I>{});
                 }()),
                 ...);
            }(std::make_integer_sequence<int, X_KT>{});
        };
        // ... and a single tile (the third tile of a three-tile wave): one dependent chain
        [[maybe_unused]] auto chain1 = [&](auto S0C, auto KC_, f32x4& acc0, auto&& hook) __attribute__((always_inline)) {
            constexpr int S0 = decltype(S0C)::value, k = decltype(KC_)::value;
            i32x4 fa[X_KT];
            auto rd = [&](auto TC, float dep) __attribute__((always_inline)) -> i32x4 {
                constexpr int tap = decltype(TC)::value, ky = tap / 3, kx = tap % 3;
                constexpr int off = ((S0 + ky) % 4) * X_ROWA + k * 1024;
                i32x4 v;
                asm volatile("ds_read_b128 %0, %1 offset:%2" : "=v"(v) : "v"(baseA[kx]), "n"(off), "v"(dep));
                return v;
            };
            [&]<int... I>(std::integer_sequence<int, I...>) { ((fa[I] = rd(IC<I>{}, 0.f)), ...); }(std::make_integer_sequence<int, AHEAD>{});
            [&]<int... I>(std::integer_sequence<int, I...>) {
                (([&] {
                     if constexpr (I + AHEAD < X_KT) fa[I + AHEAD] = rd(IC<(I + AHEAD < X_KT ? I + AHEAD : 0)>{}, I == 0 ? 0.f : acc0[0]);
                     constexpr int newer = (X_KT - 1 - I) < AHEAD ? (X_KT - 1 - I) : AHEAD;
                     asm volatile("s_waitcnt lgkmcnt(%1)" : "+v"(fa[I]) : "n"(newer));
                     acc0 = mfma16<DT>(fa[I], w2[2 * I], I == 0 ? zero4 : acc0);
                     hook(IC<I>{});
                 }()),
                 ...);
            }(std::make_integer_sequence<int, X_KT>{});
        };
        // the first-half finish of two tiles behind MFMA pairs 1 .. 6 of the next chain
        [[maybe_unused]] auto with_finish2 = [&](auto&& base, auto PRC, const f32x4 (&accx)[2], int (&hpx)[2][2], auto& q0x, auto& q1x, i32x4 (&opx)[2],
                                const f32x4 (&accy)[2], int (&hpy)[2][2], auto& q0y, auto& q1y, i32x4 (&opy)[2], int (&vt)[2][2]) __attribute__((always_inline)) {
            return [&, PRC](auto IC_) __attribute__((always_inline)) {
                constexpr int i = decltype(IC_)::value;
                base(IC_);
                if constexpr (i >= 1 && i <= 3) finish_part(IC<(i >= 1 && i <= 3 ? i - 1 : 0)>{}, PRC, accx, hpx, q0x, q1x, opx, vt);
                if constexpr (i >= 4 && i <= 6) finish_part(IC<(i >= 4 && i <= 6 ? i - 4 : 0)>{}, PRC, accy, hpy, q0y, q1y, opy, vt);
            };
        };
        [[maybe_unused]] auto finish1 = [&](auto PRC, const f32x4 (&acc)[2], int (&hpx)[2][2], auto& q0x, auto& q1x, i32x4 (&opx)[2]) __attribute__((always_inline)) {
            int vt[2][2];
            finish_part(IC<0>{}, PRC, acc, hpx, q0x, q1x, opx, vt);
            finish_part(IC<1>{}, PRC, acc, hpx, q0x, q1x, opx, vt);
            finish_part(IC<2>{}, PRC, acc, hpx, q0x, q1x, opx, vt);
        };
        [[maybe_unused]] auto out1 = [&](auto KC, auto PC, const i32x4 (&opk)[2], const i32x4 (&opn)[2], bool cross) __attribute__((always_inline)) {
            constexpr int k = decltype(KC)::value, P = decltype(PC)::value;
            constexpr int off = ((P + 3) % X_NB) * ROWB;         // B row t-5
            f32x4 H = mfma16<RN_DTYPE_F16>(opk[0], pm, zero4);
            if (cross) H = mfma16<RN_DTYPE_F16>(opn[0], pmx, H);
            float y[4];
#pragma unroll
            for (int i = 0; i < 4; ++i) y[i] = __builtin_fmaf(H[i], scv[0][i], shv[0][i]);
            auto& wb = wbB;
            if constexpr (NB) {
                const i32x2 d = {static_cast<int>(pack2p<DT>(y[0], y[1])), static_cast<int>(pack2p<DT>(y[2], y[3]))};
                asm volatile("ds_write_b64 %0, %1 offset:%2" ::"v"(wb[k]), "v"(d), "n"(off) : "memory");
            } else {
                const i32x4 d = {static_cast<int>(pack2p<DT>(y[0], y[1])), static_cast<int>(pack2p<DT>(y[2], y[3])), cdead0, cdead1};
                asm volatile("ds_write_b128 %0, %1 offset:%2" ::"v"(wb[k]), "v"(d), "n"(off) : "memory");
            }
        };

        auto out = [&](auto KC, auto PC, const i32x4 (&opk)[2], const i32x4 (&opn)[2], bool cross) __attribute__((always_inline)) {
            constexpr int k = decltype(KC)::value, P = decltype(PC)::value;
            constexpr int off = ((P + 3) % X_NB) * X_ROWB;       // B row t-5
            f32x4 H[2];
#pragma unroll
            for (int h = 0; h < 2; ++h) H[h] = mfma16<RN_DTYPE_F16>(opk[h], pm, zero4);
            if (cross) {
#pragma unroll
                for (int h = 0; h < 2; ++h) H[h] = mfma16<RN_DTYPE_F16>(opn[h], pmx, H[h]);
            }
            float y[8];
#pragma unroll
            for (int h = 0; h < 2; ++h)
#pragma unroll
                for (int i = 0; i < 4; ++i) y[4 * h + i] = __builtin_fmaf(H[h][i], scv[h][i], shv[h][i]);
            const i32x4 d = {static_cast<int>(pack2p<DT>(y[0], y[1])), static_cast<int>(pack2p<DT>(y[2], y[3])),
                             static_cast<int>(pack2p<DT>(y[4], y[5])), static_cast<int>(pack2p<DT>(y[6], y[7]))};
            auto& wb = wbB;
            asm volatile("ds_write_b128 %0, %1 offset:%2" ::"v"(wb[k]), "v"(d), "n"(off) : "memory");
        };
        auto step = [&](auto PC, int t) __attribute__((always_inline)) {
            constexpr int P = decltype(PC)::value;
            [[maybe_unused]] constexpr int PR = P & 1;
            if (t < nrows + 9) a_next += Win * 64;
            // A row t+1: two DMA pieces behind MFMA pairs of the first tile's chain, two behind the second's.  Skip rows of the
            // partner consumer: decided behind the first chain's MFMAs, fetched behind the third tile's; when no new row is
            // due the newest one is fetched again into its own slot (same bytes), so there is no branch around a DMA
            auto hook0 = [&](auto IC_) __attribute__((always_inline)) {
                constexpr int i = decltype(IC_)::value;
                if constexpr (i == 1) issue_A_piece(IC<0>{}, a_next, (P + 1) % X_NA);
                if constexpr (i == 5) issue_A_piece(IC<1>{}, a_next, (P + 1) % X_NA);
                if constexpr (i == 3) {
                    const int need_cur = min(ylo_nxt + 1, Win - 1);              // ylo_nxt: lo row of this step's output row
                    if (sk_f < need_cur) {                  // the consumer fetches this one itself (lo row jumped by 2)
                        ++sk_f;
                        sk_slot = sk_slot == X_NSK - 1 ? 0 : sk_slot + 1;
                    }
                }
                if constexpr (i == 7) {
                    ylo_nxt = ylo_step(t + 1);
                    const int need_next = min(ylo_nxt + 1, Win - 1);
                    if (sk_f < need_next) {
                        ++sk_f;
                        sk_slot = sk_slot == X_NSK - 1 ? 0 : sk_slot + 1;
                    }
                }
            };
            auto hook1 = [&](auto IC_) __attribute__((always_inline)) {
                constexpr int i = decltype(IC_)::value;
                if constexpr (i == 1) issue_A_piece(IC<2>{}, a_next, (P + 1) % X_NA);
                if constexpr (i == 5) issue_A_piece(IC<3>{}, a_next, (P + 1) % X_NA);
            };
            auto hook2 = [&](auto IC_) __attribute__((always_inline)) {
                constexpr int i = decltype(IC_)::value;
                if constexpr (i == 0) issue_skip_piece(IC<0>{}, sk_f, sk_slot);
                if constexpr (i == 2) issue_skip_piece(IC<1>{}, sk_f, sk_slot);
                if constexpr (i == 4) issue_skip_piece(IC<2>{}, sk_f, sk_slot);
                if constexpr (i == 6) issue_skip_piece(IC<3>{}, sk_f, sk_slot);
            };
            if constexpr (PH == 1) {
                // the DMA pieces of A row t + 1 and the skip bookkeeping behind the first chain's MFMA pairs, the partner
                // consumer's skip pieces and the two tiles' finish slices behind the second's
                auto hookA = [&](auto IC_) __attribute__((always_inline)) {
                    constexpr int i = decltype(IC_)::value;
                    if constexpr (i == 0) issue_A_piece(IC<0>{}, a_next, (P + 1) % X_NA);
                    if constexpr (i == 2) issue_A_piece(IC<1>{}, a_next, (P + 1) % X_NA);
                    if constexpr (i == 4) issue_A_piece(IC<2>{}, a_next, (P + 1) % X_NA);
                    if constexpr (i == 6) issue_A_piece(IC<3>{}, a_next, (P + 1) % X_NA);
                    if constexpr (i == 3 || i == 7) hook0(IC_);              // (hook0's bookkeeping slots)
                };
                f32x4 ac0[2], ac1[2], ac2[2], ac3[2];
                int vt[2][2];
                i32x4 op[4][2];
                op[3][0] = op[3][1] = i32x4{0, 0, 0, 0};
                chain2(IC<(P + 2) % 4>{}, IC<0>{}, IC<1>{}, ac0[0], ac1[0], hookA);
                if (has4) {
                    chain2(IC<(P + 2) % 4>{}, IC<2>{}, IC<3>{}, ac2[0], ac3[0],
                           with_finish2(hook2, IC<(P & 3)>{}, ac0, hp[0], q0[0], q1[0], op[0], ac1, hp[1], q0[1], q1[1], op[1], vt));
                    out1(IC<0>{}, PC, op[0], op[1], true);
                    finish1(IC<(P & 3)>{}, ac2, hp[2], q0[2], q1[2], op[2]);
                    out1(IC<1>{}, PC, op[1], op[2], true);
                    finish1(IC<(P & 3)>{}, ac3, hp[3], q0[3], q1[3], op[3]);
                    out1(IC<2>{}, PC, op[2], op[3], true);
                    out1(IC<3>{}, PC, op[3], op[3], false);
                } else {
                    chain1(IC<(P + 2) % 4>{}, IC<2>{}, ac2[0],
                           with_finish2(hook2, IC<(P & 3)>{}, ac0, hp[0], q0[0], q1[0], op[0], ac1, hp[1], q0[1], q1[1], op[1], vt));
                    out1(IC<0>{}, PC, op[0], op[1], true);
                    finish1(IC<(P & 3)>{}, ac2, hp[2], q0[2], q1[2], op[2]);
                    out1(IC<1>{}, PC, op[1], op[2], true);
                    out1(IC<2>{}, PC, op[2], op[3], true);         // op[3] = 0, and columns 13..15 of the third tile go to the dummy column
                }
                wait_vmcnt<0>();
                asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
                raw_barrier();
                return;
            }
            // tile k's ReLU6 / pack / pair sums ride behind the MFMAs of tile k + 1's chain (two accumulator pairs alternate)
            f32x4 accA[2], accB[2];
            int vt[2][2];
            i32x4 op[4][2];
            op[3][0] = op[3][1] = i32x4{0, 0, 0, 0};
            chain(IC<(P + 2) % 4>{}, IC<X_ROWA>{}, IC<0>{}, baseA, w2, accA, hook0);      // conv row t-2: A rows t-2 .. t
            chain(IC<(P + 2) % 4>{}, IC<X_ROWA>{}, IC<1>{}, baseA, w2, accB, with_finish(hook1, IC<(P & 3)>{}, accA, hp[0], q0[0], q1[0], op[0], vt));
            chain(IC<(P + 2) % 4>{}, IC<X_ROWA>{}, IC<2>{}, baseA, w2, accA, with_finish(hook2, IC<(P & 3)>{}, accB, hp[1], q0[1], q1[1], op[1], vt));
            out(IC<0>{}, PC, op[0], op[1], true);
            if (has4) {
                chain(IC<(P + 2) % 4>{}, IC<X_ROWA>{}, IC<3>{}, baseA, w2, accB, with_finish(no_hook, IC<(P & 3)>{}, accA, hp[2], q0[2], q1[2], op[2], vt));
                out(IC<1>{}, PC, op[1], op[2], true);
                finish(IC<(P & 3)>{}, accB, hp[3], q0[3], q1[3], op[3]);
                out(IC<2>{}, PC, op[2], op[3], true);
                out(IC<3>{}, PC, op[3], op[3], false);
            } else {
                finish(IC<(P & 3)>{}, accA, hp[2], q0[2], q1[2], op[2]);
                out(IC<1>{}, PC, op[1], op[2], true);
                out(IC<2>{}, PC, op[2], op[3], true);         // op[3] = 0, and columns 13..15 of the third tile go to the dummy column
            }
            wait_vmcnt<0>();
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
        const int rem = nsteps - t;
        if (rem > 0) step(IC<0>{}, t);
        if (rem > 1) step(IC<1>{}, t + 1);
        if (rem > 2) step(IC<2>{}, t + 2);
        wait_vmcnt<0>();
        return;
    }

    // =================================================================== consumer: second stage, B ring -> HBM
    __builtin_amdgcn_s_setprio(1);                                  // (measured: consumers high 0.51 ms, none 0.55, producers high 0.55)
    const bool has4 = wq >= 2;                                      // tiles of this wave: 3 3 4 4
    const int xw = xc_start(wq);
    const int xend = wq == 3 ? Wo : min(xc_start(wq + 1), Wo);      // this wave stores output columns [xw, xend)
    constexpr int NW3 = N8 ? 6 : NB ? 10 : 2 * X_KT;                 // narrow: fragment 2 * chunk + half
    i32x4 w3[2 * X_KT];
#pragma unroll
    for (int f = 0; f < NW3; ++f) {
        const i32x4* src = a.wfrag3 + f * 64 + lane;
        asm volatile("global_load_dwordx4 %0, %1, off" : "=v"(w3[f]) : "v"(src) : "memory");
    }
    unsigned baseB[3], a_off[4];
    i32x4 wx[4];
#pragma unroll
    for (int kx = 0; kx < 3; ++kx) {
        const int p = xw + px16 + kx;
        baseB[kx] = ringB_lds + static_cast<unsigned>(p * 64 + ((g ^ swzx(p)) << 4));
    }
    // narrow form: the 9 taps x 16 channels are contracted as five K = 32 chunks of two taps, tap 2 c + (g >> 1) of chunk c (tap 9 has
    // zero weights): (ky0: kx0 | kx1) (ky0: kx2 | ky1: kx0) (ky1: kx1 | kx2) (ky2: kx0 | kx1) (ky2: kx2 | -); a lane reads channels
    // 8 (g & 1) .. + 7 of its tap's pixel.  Chunk 1 takes its two taps from two ring rows: the row step sits in the lane's base
    // (baseMix; baseMixW for the phase whose second row wraps to ring slot 0)
    [[maybe_unused]] unsigned baseN0 = 0, baseMix = 0, baseMixW = 0;
    [[maybe_unused]] f32x4 cinit[2] = {zero4, zero4};
    if constexpr (NB) {
        // plain layout: ds_read_b128's four lane groups ({0-3, 12-15, 20-27}, ...: MI355X_MICROARCH.md) take the even 16-byte
        // slots of eight pixels from the g = 0 / 2 lanes and the odd slots of the other eight from g = 1 / 3 -- 16 distinct
        // slots of the 256-byte bank row at every alignment (an XOR of the chunk with bit 3 of the pixel made it 2-way: 30 %
        // conflict cycles, profiles/r5_b_sq_summary.txt)
        if constexpr (!N8) {
        const unsigned lane0 = ringB_lds + static_cast<unsigned>((xw + px16) * 32 + ((g & 1) << 4));
        baseN0 = lane0 + static_cast<unsigned>((g >> 1) * 32);
        baseMix = lane0 + static_cast<unsigned>((g >> 1) ? X_ROWB_N : 64);
        baseMixW = lane0 + static_cast<unsigned>((g >> 1) ? 0 : 3 * X_ROWB_N + 64);
        }
#pragma unroll
        for (int h = 0; h < 2; ++h) {
            const float c = a.ptab[160 + 8 * (px16 >> 2) + 4 * h + (px16 & 3)];       // D'[pixel][cout]: column px16 of half h
            cinit[h] = f32x4{c, c, c, c};
        }
        // (the ring's unwritten columns feed zero-weight K slots: they must be finite.  Consumer waves only: threads 256 .. 511)
        for (int i = tid - 256; i < X_NB * ROWB / 16; i += 256) *reinterpret_cast<i32x4*>(ringB + i * 16) = i32x4{0, 0, 0, 0};
    }
    // N8: tap 4 c + g of chunk c for lane group g (taps 9..11 of chunk 2: zero weights, they read tap 8's pixel): its kernel row and
    // the lane's column address inside a ring row
    [[maybe_unused]] int ky8[3] = {0, 0, 0};
    [[maybe_unused]] unsigned col8[3] = {0, 0, 0};
    if constexpr (N8) {
#pragma unroll
        for (int c = 0; c < 3; ++c) {
            const int tap = min(4 * c + g, 8);
            ky8[c] = tap / 3;
            col8[c] = ringB_lds + static_cast<unsigned>((xw + px16 + tap % 3) * 16);
        }
    }
    // output stores: tile k = 1024 bytes further (immediate); a lane stores while its column lies left of the wave's limit
    // for that tile: lim(k) = columns of tile k this wave owns (13 in its last tile, cut at the next wave's start / the row end)
    const int voff0 = ((x0 + xw + px16) * 32 + 8 * g) * 2;
    const int ntile = has4 ? 4 : 3;
    auto lim = [&](int k) __attribute__((always_inline)) { return min(xend - xw - 16 * k, k == ntile - 1 ? 13 : 16); };
#pragma unroll
    for (int k = 0; k < 4; ++k) {
        const int xo = xw + 16 * k + px16;
        // interpolation window of the tile: 32 source columns starting at xs_k (ring-relative, kept inside the 64-column ring)
        const int xs_k = min(a.rlo[x0 + min(xw + 16 * k, Wo - 1)] - xs0, 32);
        const int xq = x0 + min(xo, Wo - 1);
        const int plo = a.rlo[xq] - xs0, phi = a.rhi[xq] - xs0;
        const float xlq = res_quant_lerp<DT>(a.rlerp[xq]);
        unsigned short wh[8];
#pragma unroll
        for (int j = 0; j < 8; ++j) {
            const int xin = xs_k + 8 * g + j;
            float w = 0.f;
            if (xin == plo) w += 1.0f - xlq;
            if (xin == phi) w += xlq;
            wh[j] = to16<DT>(w);
        }
#pragma unroll
        for (int d = 0; d < 4; ++d) wx[k][d] = static_cast<int>(static_cast<unsigned>(wh[2 * d]) | (static_cast<unsigned>(wh[2 * d + 1]) << 16));
        // transposed reads of the skip rows: lane 4 q + p of a 16-lane group supplies the address of pixel row q, couts
        // 8 p + 4 h .. + 3 (chunk p, byte 8 h: the permuted cout order of fragment column 4 p + j); the second block of 4
        // pixels lies 256 bytes further
        const int q = px16 >> 2, p = px16 & 3;
        a_off[k] = skw_lds + static_cast<unsigned>((xs_k + 8 * g + q) * 64 + (p << 4));
    }
    asm volatile("" : "+v"(wx[0]), "+v"(wx[1]), "+v"(wx[2]), "+v"(wx[3]));
    // folded BN of the lane's 8 couts (sc3', sh3', sc4): read from the LDS table per tile (24 registers otherwise)
    const unsigned tabl_lds = lds_addr(tab + 64 + 8 * g);
    const int out_row_bytes = Hout * 64;
    const char* out_row = reinterpret_cast<const char*>(a.out + static_cast<int64_t>(n) * Hout * Hout * 32) + static_cast<int64_t>(yo0) * out_row_bytes;
    int hp[4][2][2];
    std::conditional_t<NB, i32x4[4][2], int[4][2][2]> q0, q1;
#pragma unroll
    for (int k = 0; k < 4; ++k)
#pragma unroll
        for (int h = 0; h < 2; ++h)
#pragma unroll
            for (int j = 0; j < 2; ++j) {
                    hp[k][h][j] = 0;
                    if constexpr (NB) {
                        q0[k][h][2 * j] = q0[k][h][2 * j + 1] = q1[k][h][2 * j] = q1[k][h][2 * j + 1] = 0;
                    } else {
                        q0[k][h][j] = q1[k][h][j] = 0;
                    }
                }
    struct RowCtx {
        float yl;
        unsigned sk_lo, sk_hi;
        const char* row;          // (the buffer resource is built at the store: four more scalars held across the step spilled)
        int emit_mask;
        unsigned seed;            // dither seed of the output row (bf16 handles)
    };
    VLerp vl_cur = vlerp_of(yo0);
    VLerp vl_pre = vl_cur;                                // interpolation of the NEXT step's output row, one step ahead
    int slot_cur = 0;
    RowCtx cx{};
    wait_vmcnt<0>();                                      // the weight fragments have landed
#pragma unroll
    for (int f = 0; f < NW3; ++f) asm volatile("" : "+v"(w3[f]));
    lds_barrier();
    // the narrow second conv: five two-tap chunks, one accumulator pair started from the frozen channels' constant
    [[maybe_unused]] auto chainN = [&](auto S0C, auto KC, f32x4 (&acc)[2], auto&& hook) __attribute__((always_inline)) {
        constexpr int S0 = decltype(S0C)::value, k = decltype(KC)::value;
        i32x4 fq[5];
        auto rd = [&](auto CC, float dep) __attribute__((always_inline)) -> i32x4 {
            constexpr int c = decltype(CC)::value;
            constexpr int row0 = S0 % 4, row1 = (S0 + 1) % 4, row2 = (S0 + 2) % 4;
            i32x4 v;
            if constexpr (c == 1) {
                if constexpr (row0 == 3)
                    asm volatile("ds_read_b128 %0, %1 offset:%2" : "=v"(v) : "v"(baseMixW), "n"(k * 512), "v"(dep));
                else
                    asm volatile("ds_read_b128 %0, %1 offset:%2" : "=v"(v) : "v"(baseMix), "n"(row0 * X_ROWB_N + k * 512), "v"(dep));
            } else {
                constexpr int off = (c == 0 ? row0 * X_ROWB_N : c == 2 ? row1 * X_ROWB_N + 32 : c == 3 ? row2 * X_ROWB_N : row2 * X_ROWB_N + 64) + k * 512;
                asm volatile("ds_read_b128 %0, %1 offset:%2" : "=v"(v) : "v"(baseN0), "n"(off), "v"(dep));
            }
            return v;
        };
        [&]<int... I>(std::integer_sequence<int, I...>) { ((fq[I] = rd(IC<I>{}, 0.f)), ...); }(std::make_integer_sequence<int, AHEAD>{});
        [&]<int... I>(std::integer_sequence<int, I...>) {
            (([&] {
                 if constexpr (I + AHEAD < 5) fq[I + AHEAD] = rd(IC<(I + AHEAD < 5 ? I + AHEAD : 0)>{}, I == 0 ? 0.f : acc[0][0]);
                 constexpr int newer = (4 - I) < AHEAD ? (4 - I) : AHEAD;
                 asm volatile("s_waitcnt lgkmcnt(%1)" : "+v"(fq[I]) : "n"(newer));
                 acc[0] = mfma16<DT>(fq[I], w3[2 * I], I == 0 ? cinit[0] : acc[0]);
                 acc[1] = mfma16<DT>(fq[I], w3[2 * I + 1], I == 0 ? cinit[1] : acc[1]);
                 hook(IC<I + 1>{});                         // (the hooks count MFMA pairs 1 .. 6 of a nine-tap chain:
                 if constexpr (I == 4) hook(IC<6>{});       //  the last pair of this one carries two of their slices)
             }()),
             ...);
        }(std::make_integer_sequence<int, 5>{});
        hook(IC<7>{});
    };
    // the eight-channel second conv: three four-tap chunks; the lane's three operand addresses of this step's ring phase are formed
    // once per chain (ring row (S0 + ky) mod 4 of the lane's tap)
    [[maybe_unused]] auto chainN8 = [&](auto S0C, auto KC, f32x4 (&acc)[2], auto&& hook) __attribute__((always_inline)) {
        constexpr int S0 = decltype(S0C)::value, k = decltype(KC)::value;
        unsigned ad[3];
#pragma unroll
        for (int c = 0; c < 3; ++c) ad[c] = col8[c] + static_cast<unsigned>(((S0 + ky8[c]) & 3) * X_ROWB_8);
        i32x4 fq[3];
#pragma unroll
        for (int c = 0; c < 3; ++c) asm volatile("ds_read_b128 %0, %1 offset:%2" : "=v"(fq[c]) : "v"(ad[c]), "n"(k * 256));
        [&]<int... I>(std::integer_sequence<int, I...>) {
            (([&] {
                 asm volatile("s_waitcnt lgkmcnt(%1)" : "+v"(fq[I]) : "n"(2 - I));
                 acc[0] = mfma16<DT>(fq[I], w3[2 * I], I == 0 ? cinit[0] : acc[0]);
                 acc[1] = mfma16<DT>(fq[I], w3[2 * I + 1], I == 0 ? cinit[1] : acc[1]);
                 hook(IC<2 * I + 1>{});                     // (the hooks count MFMA pairs 1 .. 7 of a nine-tap chain: two of
                 hook(IC<2 * I + 2>{});                     //  their slices behind every pair of this one)
             }()),
             ...);
        }(std::make_integer_sequence<int, 3>{});
        hook(IC<7>{});
    };
    auto cchain = [&](auto PC_, auto KC, f32x4 (&acc)[2], auto&& hook) __attribute__((always_inline)) {
        if constexpr (N8)
            chainN8(PC_, KC, acc, hook);
        else if constexpr (NB)
            chainN(PC_, KC, acc, hook);
        else
            chain(PC_, IC<X_ROWB>{}, KC, baseB, w3, acc, hook);
    };

    // The LDS reads of a tile's epilogue (8 transposed reads of the skip rows, 6 table reads) are issued in FRONT of the next
    // tile's ReLU6 / pack work (out_reads), their consumers run behind it (out_rest): the pack's ~25 VALU instructions cover
    // the LDS round trip that used to stand exposed between the reads and the residual MFMAs.
    struct OutRegs {
        i32x2 tq[2][2][2];                      // [lo / hi][half][block]
        f32x4 tsc1[2], tsh1[2], tsc2[2];
    };
    auto out_reads = [&](auto KC, OutRegs& R) __attribute__((always_inline)) {
        constexpr int k = decltype(KC)::value;
        auto& tq = R.tq;
        // residual: R_lo / R_hi [cout][xo] = Skip^T [cout][32 source columns] * Wx on the matrix cores
        const unsigned alo = a_off[k] + cx.sk_lo, ahi = a_off[k] + cx.sk_hi;
        asm volatile("ds_read_b64_tr_b16 %0, %1" : "=v"(tq[0][0][0]) : "v"(alo));
        asm volatile("ds_read_b64_tr_b16 %0, %1 offset:256" : "=v"(tq[0][0][1]) : "v"(alo));
        asm volatile("ds_read_b64_tr_b16 %0, %1 offset:8" : "=v"(tq[0][1][0]) : "v"(alo));
        asm volatile("ds_read_b64_tr_b16 %0, %1 offset:264" : "=v"(tq[0][1][1]) : "v"(alo));
        asm volatile("ds_read_b64_tr_b16 %0, %1" : "=v"(tq[1][0][0]) : "v"(ahi));
        asm volatile("ds_read_b64_tr_b16 %0, %1 offset:256" : "=v"(tq[1][0][1]) : "v"(ahi));
        asm volatile("ds_read_b64_tr_b16 %0, %1 offset:8" : "=v"(tq[1][1][0]) : "v"(ahi));
        asm volatile("ds_read_b64_tr_b16 %0, %1 offset:264" : "=v"(tq[1][1][1]) : "v"(ahi));
    };
    auto out_rest = [&](auto KC, OutRegs& R, const i32x4 (&opk)[2], const i32x4 (&opn)[2], bool cross) __attribute__((always_inline)) {
        constexpr int k = decltype(KC)::value;
        auto& tq = R.tq;
        auto& tsc1 = R.tsc1;
        auto& tsh1 = R.tsh1;
        auto& tsc2 = R.tsc2;
        // BN tables of the lane's 8 couts: read here, behind the skip reads and in front of the eight pooling / residual MFMAs
        // (a chain earlier they cost 24 registers across the chain)
        {
            const unsigned ta = tabl_lds;
            auto& t1 = R.tsc1;
            auto& t2 = R.tsh1;
            auto& t3 = R.tsc2;
#pragma unroll
            for (int h = 0; h < 2; ++h) {
                asm volatile("ds_read_b128 %0, %1 offset:%2" : "=v"(t1[h]) : "v"(ta), "n"(16 * h));
                asm volatile("ds_read_b128 %0, %1 offset:%2" : "=v"(t2[h]) : "v"(ta), "n"(128 + 16 * h));
                asm volatile("ds_read_b128 %0, %1 offset:%2" : "=v"(t3[h]) : "v"(ta), "n"(256 + 16 * h));
            }
        }
        f32x4 H[2];
#pragma unroll
        for (int h = 0; h < 2; ++h) H[h] = mfma16<RN_DTYPE_F16>(opk[h], pm, zero4);
        if (cross) {
#pragma unroll
            for (int h = 0; h < 2; ++h) H[h] = mfma16<RN_DTYPE_F16>(opn[h], pmx, H[h]);
        }
        asm volatile("s_waitcnt lgkmcnt(0)"
                     : "+v"(tq[0][0][0]), "+v"(tq[0][0][1]), "+v"(tq[0][1][0]), "+v"(tq[0][1][1]), "+v"(tq[1][0][0]), "+v"(tq[1][0][1]),
                       "+v"(tq[1][1][0]), "+v"(tq[1][1][1]), "+v"(tsc1[0]), "+v"(tsc1[1]), "+v"(tsh1[0]), "+v"(tsh1[1]), "+v"(tsc2[0]), "+v"(tsc2[1]));
        float y[8];
#pragma unroll
        for (int h = 0; h < 2; ++h) {
            const i32x4 al = {tq[0][h][0][0], tq[0][h][0][1], tq[0][h][1][0], tq[0][h][1][1]};
            const i32x4 ah = {tq[1][h][0][0], tq[1][h][0][1], tq[1][h][1][0], tq[1][h][1][1]};
            const f32x4 r_lo = mfma16<DT>(al, wx[k], zero4);
            const f32x4 r_hi = mfma16<DT>(ah, wx[k], zero4);
#pragma unroll
            for (int i = 0; i < 4; ++i) {
                const float y1 = __builtin_fmaf(H[h][i], tsc1[h][i], tsh1[h][i]);
                const float lo = r_lo[i];
                const float rs = __builtin_fmaf(r_hi[i] - lo, cx.yl, lo);
                y[4 * h + i] = __builtin_fmaf(rs, tsc2[h][i], y1);
            }

        }
        i32x4 d;
        if constexpr (DT == RN_DTYPE_BF16)          // (bf16: v_cvt_sr_bf16_f32 with the output row's dither seed or the plain one, rn_stage.h)
            d = i32x4{static_cast<int>(pack2_sr_bf16(y[0], y[1], cx.seed)), static_cast<int>(pack2_sr_bf16(y[2], y[3], cx.seed)),
                      static_cast<int>(pack2_sr_bf16(y[4], y[5], cx.seed)), static_cast<int>(pack2_sr_bf16(y[6], y[7], cx.seed))};
        else
            d = i32x4{static_cast<int>(pack2<DT>(y[0], y[1])), static_cast<int>(pack2<DT>(y[2], y[3])),
                      static_cast<int>(pack2<DT>(y[4], y[5])), static_cast<int>(pack2<DT>(y[6], y[7]))};
        const __amdgpu_buffer_rsrc_t rs = __builtin_amdgcn_make_buffer_rsrc(const_cast<char*>(cx.row), 0, out_row_bytes, 0x00020000);
        const int vo = (px16 < lim(k) ? voff0 + 1024 * k : OOB) | cx.emit_mask;
        __builtin_amdgcn_raw_buffer_store_b128(d, rs, vo, 0, RN_NT_OUT);
    };
    auto step = [&](auto PC, int t) __attribute__((always_inline)) {
        constexpr int P = decltype(PC)::value;
        [[maybe_unused]] constexpr int PR = P & 1;
        const int jo = t - X_LAG;
        cx.yl = vl_cur.yl;
        cx.sk_lo = static_cast<unsigned>(slot_cur * X_SKROW);
        cx.sk_hi = static_cast<unsigned>((vl_cur.ylo + 1 > Win - 1 ? slot_cur : (slot_cur == X_NSK - 1 ? 0 : slot_cur + 1)) * X_SKROW);
        cx.row = out_row;
        cx.emit_mask = jo >= 0 ? 0 : OOB;
        cx.seed = a.dither ? rn_dither_seed(yo0 + max(jo, 0)) : RN_SEED_PLAIN;
        if (jo >= 0 && jo < nrows - 1) out_row += out_row_bytes;
        const VLerp vl_next = vl_pre;                  // (computed behind the previous step's third chain)
        {
            const int need_cur = min(vl_cur.ylo + 1, Win - 1);
            if (sk_f < need_cur) {
                ++sk_f;
                sk_slot = sk_slot == X_NSK - 1 ? 0 : sk_slot + 1;
                issue_skip_row(sk_f, sk_slot);
            }
            const int need_next = min(vl_next.ylo + 1, Win - 1);
            if (sk_f < need_next) {
                ++sk_f;
                sk_slot = sk_slot == X_NSK - 1 ? 0 : sk_slot + 1;
            }
        }
        // Tile k's ReLU6 / pack / pair sums ride behind the MFMAs of tile k + 1's chain; the LDS reads of tile k's epilogue go
        // out in front of tile k + 2's chain and are consumed behind it (they return in order, ahead of the chain's first
        // fragment: ~50 cycles of LDS array time at the head of the chain instead of an exposed round trip behind it)
        f32x4 accA[2], accB[2];
        int vt[2][2];
        i32x4 op[4][2];
        op[3][0] = op[3][1] = i32x4{0, 0, 0, 0};
        OutRegs R;
        auto hook_c2 = [&](auto IC_) __attribute__((always_inline)) {
            if constexpr (decltype(IC_)::value == 7) vl_pre = vlerp_of(yo0 + min(max(jo + 2, 0), nrows - 1));
        };
        cchain(IC<P>{}, IC<0>{}, accA, no_hook);           // conv row t-8: B rows t-8 .. t-6
        cchain(IC<P>{}, IC<1>{}, accB, with_finish(no_hook, IC<(P & 3)>{}, accA, hp[0], q0[0], q1[0], op[0], vt));
        wait_vmcnt<0>();                                       // a skip row fetched at the top of this step has landed
        out_reads(IC<0>{}, R);
        cchain(IC<P>{}, IC<2>{}, accA, with_finish(hook_c2, IC<(P & 3)>{}, accB, hp[1], q0[1], q1[1], op[1], vt));
        out_rest(IC<0>{}, R, op[0], op[1], true);
        if (has4) {
            out_reads(IC<1>{}, R);
            cchain(IC<P>{}, IC<3>{}, accB, with_finish(no_hook, IC<(P & 3)>{}, accA, hp[2], q0[2], q1[2], op[2], vt));
            out_rest(IC<1>{}, R, op[1], op[2], true);
            out_reads(IC<2>{}, R);
            finish(IC<(P & 3)>{}, accB, hp[3], q0[3], q1[3], op[3]);
            __builtin_amdgcn_sched_barrier(0);
            out_rest(IC<2>{}, R, op[2], op[3], true);
            out_reads(IC<3>{}, R);
            out_rest(IC<3>{}, R, op[3], op[3], false);
        } else {
            out_reads(IC<1>{}, R);
            finish(IC<(P & 3)>{}, accA, hp[2], q0[2], q1[2], op[2]);
            __builtin_amdgcn_sched_barrier(0);
            out_rest(IC<1>{}, R, op[1], op[2], true);
            out_reads(IC<2>{}, R);
            out_rest(IC<2>{}, R, op[2], op[3], true);        // op[3] = 0: columns 13..15 of the third tile are not stored
        }
        {
            int sl = slot_cur + (vl_next.ylo - vl_cur.ylo);
            slot_cur = sl >= X_NSK ? sl - X_NSK : sl;
            vl_cur = vl_next;
        }
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
    const int rem = nsteps - t;
    if (rem > 0) step(IC<0>{}, t);
    if (rem > 1) step(IC<1>{}, t + 1);
    if (rem > 2) step(IC<2>{}, t + 2);
    wait_vmcnt<0>();
#ifdef RN_CLOCK
    if (a.stamp_buf && tid == 256) {
        unsigned long long t1, r1;
        clock_pair(t1, r1);
        const int64_t wg = static_cast<int64_t>(blockIdx.y) * gridDim.x + blockIdx.x;
        a.stamp_buf[wg * 2 + 0] = t1 - ck_t0;
        a.stamp_buf[wg * 2 + 1] = r1 - ck_r0;
    }
#endif
}

}  // namespace

// Column blocks of the output side (= in_side - 10): as few as fit the rings, of equal width +-1.  Every block's input
// width (its output width + 10) must lie in [X_WMIN, X_WMAX].  224 x 224: one block of 205; 600 x 600: 194 + 194 + 193.
bool rn_stage23_plan(int in_side, int* n_cblocks, int* x0, int* wo) {
    const int out = in_side - 10;
    if (out < X_WMIN - 10) return false;
    const int nb = (out + (X_WMAX - 10) - 1) / (X_WMAX - 10);
    if (nb > 4) return false;
    const int base = out / nb, rem = out % nb;
    if (base + 10 < X_WMIN) return false;
    int x = 0;
    for (int b = 0; b < nb; ++b) {
        const int w = base + (b < rem ? 1 : 0);
        if (x0) x0[b] = x;
        if (wo) wo[b] = w;
        x += w;
    }
    if (n_cblocks) *n_cblocks = nb;
    return true;
}

bool rn_stage23_supported(int in_side) { return rn_stage23_plan(in_side, nullptr, nullptr, nullptr); }

// B-operand fragments of the 16x16x32 form: frag[tap][half][lane][j] = W[k = 32 tap + 8 (lane / 16) + j][cout(half, lane % 16)]
// with cout(h, n) = 8 (n / 4) + 4 h + n % 4 (so that a lane's pooled rows 4 g + i of the two halves are couts 8 g .. 8 g + 7)
void rn_stage23x_pack(const float* w_hwio, int dtype, unsigned short (*cvt_bf16)(float), unsigned short (*cvt_f16)(float),
                      std::vector<unsigned short>* out) {
    out->assign(static_cast<size_t>(2 * X_KT) * 64 * 8, 0);
    for (int tap = 0; tap < X_KT; ++tap)
        for (int h = 0; h < 2; ++h)
            for (int l = 0; l < 64; ++l)
                for (int j = 0; j < 8; ++j) {
                    const int k = 32 * tap + 8 * (l >> 4) + j, nn = l & 15;
                    const int co = 8 * (nn >> 2) + 4 * h + (nn & 3);
                    const float v = w_hwio[static_cast<size_t>(k) * 32 + co];
                    (*out)[((static_cast<size_t>(tap) * 2 + h) * 64 + l) * 8 + j] = dtype == RN_DTYPE_BF16 ? cvt_bf16(v) : cvt_f16(v);
                }
}

// Narrow second conv (NB): frag[(2 ky + j)][half][lane][jj] = W[ky][kx = 2 j + (lane / 32)][cin = ring_cin[8 ((lane / 16) & 1) + jj]][cout(half, lane % 16)]
// (kx = 3: zero); ring_cin[r] = the stage-2 channel at ring channel r (16 entries).  `w_hwio` is the UNpermuted [tap][cin][cout] kernel.
void rn_stage23x_pack_narrow(const float* w_hwio, const int* ring_cin, int dtype, unsigned short (*cvt_bf16)(float), unsigned short (*cvt_f16)(float),
                             std::vector<unsigned short>* out) {
    out->assign(static_cast<size_t>(10) * 64 * 8, 0);
    for (int c = 0; c < 5; ++c)
        for (int h = 0; h < 2; ++h)
            for (int l = 0; l < 64; ++l)
                for (int jj = 0; jj < 8; ++jj) {
                    const int tap = 2 * c + (l >> 5), nn = l & 15;          // chunk c = taps 2 c, 2 c + 1 (tap = 3 ky + kx; tap 9: zeros)
                    if (tap > 8) continue;
                    const int cin = ring_cin[8 * ((l >> 4) & 1) + jj];
                    const int co = 8 * (nn >> 2) + 4 * h + (nn & 3);
                    const float v = w_hwio[(static_cast<size_t>(tap) * 32 + cin) * 32 + co];
                    (*out)[((static_cast<size_t>(c) * 2 + h) * 64 + l) * 8 + jj] = dtype == RN_DTYPE_BF16 ? cvt_bf16(v) : cvt_f16(v);
                }
}

// Eight-channel second conv (N8): frag[c][half][lane][jj] = W[tap 4 c + lane / 16][cin = ring_cin[jj]][cout(half, lane % 16)] (taps > 8: zero);
// ring_cin[r] = the stage-2 channel at ring channel r (8 entries).  `w_hwio` is the UNpermuted [tap][cin][cout] kernel.
void rn_stage23x_pack_narrow8(const float* w_hwio, const int* ring_cin, int dtype, unsigned short (*cvt_bf16)(float), unsigned short (*cvt_f16)(float),
                              std::vector<unsigned short>* out) {
    out->assign(static_cast<size_t>(6) * 64 * 8, 0);
    for (int c = 0; c < 3; ++c)
        for (int h = 0; h < 2; ++h)
            for (int l = 0; l < 64; ++l)
                for (int jj = 0; jj < 8; ++jj) {
                    const int tap = 4 * c + (l >> 4), nn = l & 15;
                    if (tap > 8) continue;
                    const int cin = ring_cin[jj];
                    const int co = 8 * (nn >> 2) + 4 * h + (nn & 3);
                    const float v = w_hwio[(static_cast<size_t>(tap) * 32 + cin) * 32 + co];
                    (*out)[((static_cast<size_t>(c) * 2 + h) * 64 + l) * 8 + jj] = dtype == RN_DTYPE_BF16 ? cvt_bf16(v) : cvt_f16(v);
                }
}

int rn_stage23x_launch(int dtype, hipStream_t s, const Stage23Args& a, int n) {
    auto launch = [&](auto kern) -> int {
        static std::atomic<unsigned long long> attr_devices{0};     // per device and instantiation
        int dev = 0;
        RN_HIP(hipGetDevice(&dev));
        if (!(attr_devices.load(std::memory_order_acquire) >> (dev & 63) & 1ull)) {
            RN_HIP(hipFuncSetAttribute(reinterpret_cast<const void*>(kern), hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024));
            attr_devices.fetch_or(1ull << (dev & 63), std::memory_order_release);
        }
        hipLaunchKernelGGL(kern, dim3(a.n_bands * a.n_cblocks, n), dim3(512), X_LDS, s, a);
        RN_CHECK_LAUNCH();
        return RN_OK;
    };
    if (a.producer_halves == 1 && a.narrow_b == 2) {
        if (dtype == RN_DTYPE_BF16) return launch(stage23x_kernel<RN_DTYPE_BF16, 1, true, true>);
        return launch(stage23x_kernel<RN_DTYPE_F16, 1, true, true>);
    }
    if (a.producer_halves == 1 && a.narrow_b) {
        if (dtype == RN_DTYPE_BF16) return launch(stage23x_kernel<RN_DTYPE_BF16, 1, true>);
        return launch(stage23x_kernel<RN_DTYPE_F16, 1, true>);
    }
    if (a.producer_halves == 1) {
        rn_set_error("stage23x: producer_halves == 1 runs in the narrow form only");
        return RN_E_STATE;
    }
    if (dtype == RN_DTYPE_BF16) return launch(stage23x_kernel<RN_DTYPE_BF16, 2, false>);
    return launch(stage23x_kernel<RN_DTYPE_F16, 2, false>);
}
