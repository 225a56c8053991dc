// Un-pooled 64 -> 128 stage (reference network.py:231, conv_block(128, pooling=False)) on 16x16x32 matrix tiles:
//
//   in [N, H, W, 64] -> conv3x3 VALID -> ReLU6 -> BN(inference)  = out [N, H-2, W-2, 128]
//
// Why a kernel of its own: the row-streaming template (rn_stage_rw.hip) tiles pixels in 32s; this stage's rows are 46
// pixels at 224 x 224, i.e. two 32-pixel tiles of which the second is 44 % full.  With v_mfma_f32_16x16x32_{bf16,f16} the
// pixel dimension is tiled in 16s: 46 pixels = three tiles (96 % full), a quarter less matrix work, and the chip
// sustains 1.12 x the FLOP rate with that shape at its power cap (profiles/r2_power_cap.txt).  The stage has no pooling
// and no residual, so every output pixel is independent and its epilogue is four fmas per lane and tile.
//
// Geometry: workgroup = image x band of output rows x block of 48 output columns; 4 waves, wave w computes couts
// 32 w .. 32 w + 31 (two 16-cout tiles) for the block's three 16-pixel tiles: six 16x16 accumulators (24 registers).
// Its 2 x 18 weight fragments (K = 9 taps x 64 channels = 18 chunks of 32) stay in registers (144).  Input rows arrive by
// LDS-DMA into a ring of 6 row slots shared by the four waves (3 live rows + 3 in flight), one s_barrier per row.
// Fragment layout (both operands): lane (i = lane % 16, kg = lane / 16) holds K elements 8 kg .. 8 kg + 7 of row / column
// i, i.e. 8 consecutive channels of one tap -- one 16-byte chunk of the NHWC pixel, read with one ds_read_b128; the
// chunk index is XOR-swizzled with pixel & 7 on the DMA's source address so the reads are bank-conflict free.
#include "rn_fused.h"
#include "rn_stage.h"

#include <atomic>
#include <utility>
#include <vector>

using namespace rnk;

namespace {

constexpr int C16_CIN = 64, C16_COUT = 128;
constexpr int C16_BLKW = 48;                      // output columns per workgroup (three 16-pixel tiles)
constexpr int C16_RINGW = C16_BLKW + 2;           // input columns a block reads
constexpr int C16_NSLOT = 6, C16_AHEAD = 5;       // ring rows; row s + AHEAD is fetched while row s is computed
constexpr int C16_ROWB = 2 * 256 * 16;            // bytes per ring row: two DMA pieces of 256 lanes x 16 B (50 px = 400 chunks + padding)
constexpr int C16_KC = 18;                        // K chunks of 32: (tap, channel half)
constexpr int C16_TAB_OFF = C16_NSLOT * C16_ROWB;     // folded BN tables behind the ring: scale[128], shift[128]
constexpr int C16_LDS = C16_TAB_OFF + 2 * C16_COUT * 4;

using f32x4v = __attribute__((ext_vector_type(4))) float;
using i32x2v = __attribute__((ext_vector_type(2))) int;

template <int DT>
__device__ __forceinline__ f32x4v mfma16(i32x4 a, i32x4 b, f32x4v c) {
    if constexpr (DT == RN_DTYPE_BF16)
        return __builtin_amdgcn_mfma_f32_16x16x32_bf16(__builtin_bit_cast(bf16x8, a), __builtin_bit_cast(bf16x8, b), c, 0, 0, 0);
    else
        return __builtin_amdgcn_mfma_f32_16x16x32_f16(__builtin_bit_cast(f16x8, a), __builtin_bit_cast(f16x8, b), c, 0, 0, 0);
}

// conflict-free for all three tap columns and both channel halves under ds_read_b128's lane grouping (checked by enumeration;
// (pix >> 1) & 7, the rings' swizzle for 32-pixel tiles, is 2-way conflicted here for kx = 1, 2)
__device__ __forceinline__ int c16_swz(int pix) { return pix & 7; }

template <int DT>
__global__ __launch_bounds__(256, 2) void conv16_kernel(const Conv16Args a) {
    extern __shared__ __attribute__((aligned(64))) char smem[];
    const int tid = threadIdx.x;
    const int lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int i16 = lane & 15, kg = lane >> 4;
    const int cb = blockIdx.x % a.n_colblocks, band = blockIdx.x / a.n_colblocks, n = blockIdx.y;
    const int x0 = cb * C16_BLKW;
    const int yo0 = band * a.rows_per_band;
    const int nrows = min(a.Ho, yo0 + a.rows_per_band) - yo0;
    const int nin = nrows + 2;

    // ---- weights -> registers: fragment (chunk c, cout tile 2 wave + ct), lane-linear 16-byte loads
    // Loaded by inline asm straight INTO the accumulator file (a VMEM load may target AGPRs): no VGPR staging, all 36 in
    // flight together.  (Compiler-visible loads pinned with an empty "+a" asm waited vmcnt(0) behind every single load --
    // 36 serial L2 round trips per wave -- or, batched, needed more VGPRs at once than the kernel has.)  hipcc does not
    // see these loads: the wait_vmcnt<0>() in front of the row loop retires them, and the empty asm behind it orders every
    // use after that wait.
    i32x4 wr[C16_KC][2];
    {
        const i32x4* wp = a.wfrag + 2 * wave * 64 + lane;
#pragma unroll
        for (int c = 0; c < C16_KC; ++c)
#pragma unroll
            for (int ct = 0; ct < 2; ++ct) {
                const i32x4* src = wp + (c * 8 + ct) * 64;
                if (2 * c + ct < 32)       // 32 fragments fill the 128 accumulator registers a 2-waves-per-SIMD wave gets
                    asm volatile("global_load_dwordx4 %0, %1, off" : "=a"(wr[c][ct]) : "v"(src) : "memory");
                else
                    asm volatile("global_load_dwordx4 %0, %1, off" : "=v"(wr[c][ct]) : "v"(src) : "memory");
            }
    }

    // ---- input-row DMA: chunk q = tid + 256 i of a ring row = (pixel p = q / 8, slot c = q % 8) <- source chunk c ^ swz(p)
    const char* const in_img = reinterpret_cast<const char*>(a.in + static_cast<int64_t>(n) * a.H * a.W * C16_CIN);
    const int64_t in_row_bytes = static_cast<int64_t>(a.W) * C16_CIN * 2;
    unsigned goff[2];
#pragma unroll
    for (int i = 0; i < 2; ++i) {
        const int q = tid + 256 * i;
        const int p = min(q >> 3, C16_RINGW - 1), c = q & 7;          // chunks past the 50th pixel land in the row's padding
        const int pc = min(x0 + p, a.W - 1);
        goff[i] = static_cast<unsigned>((pc * C16_CIN + ((c ^ c16_swz(p)) << 3)) * 2);
    }
    auto issue_row = [&](int j, int slot) __attribute__((always_inline)) {
        const char* row = in_img + static_cast<int64_t>(yo0 + min(j, nin - 1)) * in_row_bytes;
#pragma unroll
        for (int i = 0; i < 2; ++i) {
            unsigned off = goff[i];
            asm volatile("" : "+v"(off));
            dma16(row + off, smem + slot * C16_ROWB + i * 4096 + wave * 1024);
        }
    };
#pragma unroll
    for (int j = 0; j < C16_AHEAD; ++j) issue_row(j, j);

    // ---- lane constants: fragment read offsets (tap column kx, channel half h), BN tables, store offsets
    const unsigned ring_lds = static_cast<unsigned>(reinterpret_cast<uintptr_t>((__attribute__((address_space(3))) char*)smem));
    unsigned boff[3][2];
#pragma unroll
    for (int kx = 0; kx < 3; ++kx)
#pragma unroll
        for (int h = 0; h < 2; ++h) {
            const int p = i16 + kx;                                    // (pixel tiles add 16 pixels = 2048 bytes: same swizzle)
            boff[kx][h] = ring_lds + static_cast<unsigned>(p * 128 + (((4 * h + kg) ^ c16_swz(p)) << 4));
        }
    // BN tables in LDS (read per row in the epilogue: keeping 16 more registers live through the MFMA loop spills)
    float* const tab = reinterpret_cast<float*>(smem + C16_TAB_OFF);
    tab[tid] = a.ptab[tid];                                            // 256 threads, 2 x 128 entries
    const float* const tab_lane = tab + 32 * wave + 4 * kg;
    constexpr int OOB = 0x40000000;
    int voff[3];
#pragma unroll
    for (int pt = 0; pt < 3; ++pt) {
        const int xo = x0 + 16 * pt + i16;
        voff[pt] = xo < a.Wo ? (xo * C16_COUT + 32 * wave + 4 * kg) * 2 : OOB;
    }
    const int out_row_bytes = a.Wo * C16_COUT * 2;
    const char* out_row = reinterpret_cast<const char*>(a.out + static_cast<int64_t>(n) * a.Ho * a.Wo * C16_COUT) +
                          static_cast<int64_t>(yo0) * out_row_bytes;

    wait_vmcnt<0>();                                                   // rows 0 .. AHEAD-1 and the weights have landed
#pragma unroll
    for (int c = 0; c < C16_KC; ++c)
#pragma unroll
        for (int ct = 0; ct < 2; ++ct) {
            if (2 * c + ct < 32)
                asm volatile("" : "+a"(wr[c][ct]));
            else
                asm volatile("" : "+v"(wr[c][ct]));
        }
    lds_barrier();                                                     // the BN table written above is visible to every wave
    // One step = one output row s (ring phase P = s mod NSLOT, compile time).  Order: counted wait for row s + 2 ->
    // barrier (every wave's pieces of it have landed, and every wave is done reading row s - 1) -> DMA of row s + AHEAD
    // into the slot of row s - 1 -> 108 MFMAs -> epilogue + 6 stores.  The wait counts only the LOADS issued behind the
    // pieces of row s + 2 (rows s + 3, s + 4: four pieces), like the row-streaming template: loads return in order among
    // themselves, a count that included the stores in between would lean on load / store ordering as well.
    auto step = [&](auto PC, int s) __attribute__((always_inline)) {
        constexpr int P = decltype(PC)::value;
        wait_vmcnt<4>();
        raw_barrier();
        issue_row(s + C16_AHEAD, (P + C16_AHEAD) % C16_NSLOT);
        f32x4v acc[3][2];
        // fragment double buffer: the three reads of chunk C + 1 are issued before the six MFMAs of chunk C and retired by a
        // counted wait (LDS returns in order), so their latency hides behind 96 cycles of matrix work
        i32x4 bq[2][3];
        auto rd = [&](auto CC, i32x4 (&b)[3]) __attribute__((always_inline)) {
            constexpr int C = decltype(CC)::value;
            constexpr int tap = C / 2, h = C % 2, ky = tap / 3, kx = tap % 3;
            constexpr int slot_off = ((P + ky) % C16_NSLOT) * C16_ROWB;
            const unsigned ad = boff[kx][h];
            asm volatile("ds_read_b128 %0, %1 offset:%2" : "=v"(b[0]) : "v"(ad), "n"(slot_off));
            asm volatile("ds_read_b128 %0, %1 offset:%2" : "=v"(b[1]) : "v"(ad), "n"(slot_off + 2048));
            asm volatile("ds_read_b128 %0, %1 offset:%2" : "=v"(b[2]) : "v"(ad), "n"(slot_off + 4096));
        };
        rd(IC<0>{}, bq[0]);
        [&]<int... C>(std::integer_sequence<int, C...>) {
            (([&] {
                 const f32x4v zero = {0.f, 0.f, 0.f, 0.f};
                 i32x4(&b)[3] = bq[C & 1];
                 if constexpr (C + 1 < C16_KC) {
                     rd(IC<(C + 1 < C16_KC ? C + 1 : 0)>{}, bq[(C + 1) & 1]);
                     asm volatile("s_waitcnt lgkmcnt(3)" : "+v"(b[0]), "+v"(b[1]), "+v"(b[2]));
                 } else {
                     asm volatile("s_waitcnt lgkmcnt(0)" : "+v"(b[0]), "+v"(b[1]), "+v"(b[2]));
                 }
#pragma unroll
                 for (int pt = 0; pt < 3; ++pt)
#pragma unroll
                     for (int ct = 0; ct < 2; ++ct) acc[pt][ct] = mfma16<DT>(wr[C][ct], b[pt], C == 0 ? zero : acc[pt][ct]);
             }()),
             ...);
        }(std::make_integer_sequence<int, C16_KC>{});
        // epilogue: ReLU6 -> BN -> 4 couts of one pixel = 8 bytes per lane and tile
        const __amdgpu_buffer_rsrc_t rs = __builtin_amdgcn_make_buffer_rsrc(const_cast<char*>(out_row), 0, out_row_bytes, 0x00020000);
#pragma unroll
        for (int pt = 0; pt < 3; ++pt)
#pragma unroll
            for (int ct = 0; ct < 2; ++ct) {
                const f32x4v sc = *reinterpret_cast<const f32x4v*>(tab_lane + 16 * ct);
                const f32x4v sh = *reinterpret_cast<const f32x4v*>(tab_lane + 16 * ct + C16_COUT);
                float y[4];
#pragma unroll
                for (int j = 0; j < 4; ++j) y[j] = fmaf(relu6f(acc[pt][ct][j]), sc[j], sh[j]);
                const i32x2v d = {static_cast<int>(pack2<DT>(y[0], y[1])), static_cast<int>(pack2<DT>(y[2], y[3]))};
                __builtin_amdgcn_raw_buffer_store_b64(d, rs, voff[pt] + 32 * ct, 0, 0);
            }
        out_row += out_row_bytes;
    };
    int s = 0;
    for (; s + C16_NSLOT - 1 < nrows; s += C16_NSLOT) {
        [&]<int... I>(std::integer_sequence<int, I...>) { (step(IC<I>{}, s + I), ...); }(std::make_integer_sequence<int, C16_NSLOT>{});
    }
    [&]<int... I>(std::integer_sequence<int, I...>) {
        ((s + I < nrows ? (step(IC<I>{}, s + I), 0) : 0), ...);
    }(std::make_integer_sequence<int, C16_NSLOT - 1>{});
    wait_vmcnt<0>();
}


// ------------------------------------------------------------------------------------------------------------------
// 128 -> 16 stage with avg-pool 4/2 (reference network.py:232, conv_block(16)) on the same 16x16x32 tiles:
//
//   in [N, H, W, 128] -> conv3x3 VALID -> ReLU6 -> avg-pool 4x4 stride 2 -> BN  = out [N, Ho, Wo, 16]
//
// The 32x32x16 template spends half of every MFMA on cout rows that do not exist (16 of 32) and splits K = 1152 over
// three waves that meet in LDS every row.  Here M = 16 is the stage's cout count exactly: one wave = one 16-pixel tile x
// all 16 couts, its 36 weight fragments (K = 9 taps x 4 channel quarters) in registers, no K split.  A 16-pixel tile is
// one 16-lane DPP row, so the horizontal pool is the two row-local shifts of the gapped 32-pixel tiles: windows start
// at the even columns 0 .. 12, 7 outputs per tile, tile stride 14 conv columns -- 44 conv columns (224 x 224) are
// exactly 3 tiles.  Workgroup = image x band x block of 3 tiles (21 output columns), 3 waves sharing the input ring.
constexpr int P16_CIN = 128, P16_COUT = 16;
// NT = pixel tiles = waves per workgroup: 3 (21 output columns: the whole row of the 224 x 224 network) or 5 (35: wider inputs run
// in two column blocks instead of four -- the launch is bound by re-reading stage 6's output, and a block re-reads 4 halo
// columns: 600 x 600 read 184 columns for 138 with three-tile blocks, 148 with five-tile ones)
template <int NT>
struct P16 {
    static constexpr int BLKO = 7 * NT;               // output columns per block
    static constexpr int RINGW = 14 * (NT - 1) + 18;  // input columns a block reads (46 / 74)
    static constexpr int PIECES = 4;                  // DMA pieces of 64 NT lanes x 16 B per row (46 x 16 = 736 <= 768; 74 x 16 = 1184 <= 1280)
    static constexpr int ROWB = PIECES * 64 * NT * 16;
    static constexpr int LDS = C16_NSLOT * ROWB;
    static_assert(RINGW * 16 <= PIECES * 64 * NT, "a ring row fits its DMA pieces");
};
constexpr int P16_KC = 36;
constexpr int P16_RD = 4;                          // operand fragments in flight + 1 (2 / 4 / 6 / 8 measured the same 30 us at batch
                                                   // 256: the launch is bound by re-reading stage 6's output from HBM, 4.6 TB/s)

// 16 chunks per 256-byte pixel = one whole bank row per pixel: chunk ^ ((pixel & 7) << 1) is conflict-free for every tap
// column and channel quarter under ds_read_b128's lane grouping (enumerated; pixel & 15 and (pixel >> 1) & 15 are not)
__device__ __forceinline__ int p16_swz(int pix) { return (pix & 7) << 1; }

template <int DT, int NT>
__global__ __launch_bounds__(64 * NT, NT == 3 ? 2 : 1) void conv16p_kernel(const Conv16Args a) {
    constexpr int P16_NT = NT, P16_BLKO = P16<NT>::BLKO, P16_RINGW = P16<NT>::RINGW, P16_PIECES = P16<NT>::PIECES, P16_ROWB = P16<NT>::ROWB;
    extern __shared__ __attribute__((aligned(64))) char smem[];
    const int tid = threadIdx.x;
    const int lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int i16 = lane & 15, kg = lane >> 4;
    const int cb = blockIdx.x % a.n_colblocks, band = blockIdx.x / a.n_colblocks, n = blockIdx.y;
    const int xc0 = 2 * P16_BLKO * cb;                       // first conv / input column of the block
    const int yo0 = band * a.rows_per_band;
    const int nout = min(a.Ho, yo0 + a.rows_per_band) - yo0;
    const int nconv = 2 * nout + 2, nin = nconv + 2;

    // weights: 36 fragments, 32 of them straight into accumulator registers (see conv16_kernel)
    i32x4 wr[P16_KC];
#pragma unroll
    for (int c = 0; c < P16_KC; ++c) {
        const i32x4* src = a.wfrag + c * 64 + lane;
        if (c < 32)
            asm volatile("global_load_dwordx4 %0, %1, off" : "=a"(wr[c]) : "v"(src) : "memory");
        else
            asm volatile("global_load_dwordx4 %0, %1, off" : "=v"(wr[c]) : "v"(src) : "memory");
    }

    const char* const in_img = reinterpret_cast<const char*>(a.in + static_cast<int64_t>(n) * a.H * a.W * P16_CIN);
    const int64_t in_row_bytes = static_cast<int64_t>(a.W) * P16_CIN * 2;
    unsigned goff[P16_PIECES];
#pragma unroll
    for (int i = 0; i < P16_PIECES; ++i) {
        const int q = tid + 64 * P16_NT * i;
        const int p = min(q >> 4, P16_RINGW - 1), c = q & 15;
        const int pc = min(xc0 + p, a.W - 1);
        goff[i] = static_cast<unsigned>((pc * P16_CIN + ((c ^ p16_swz(p)) << 3)) * 2);
    }
    auto issue_row = [&](int j, int slot) __attribute__((always_inline)) {
        const char* row = in_img + static_cast<int64_t>(2 * yo0 + min(j, nin - 1)) * in_row_bytes;
#pragma unroll
        for (int i = 0; i < P16_PIECES; ++i) {
            unsigned off = goff[i];
            asm volatile("" : "+v"(off));
            dma16(row + off, smem + slot * P16_ROWB + i * (64 * P16_NT * 16) + wave * 1024);
        }
    };
#pragma unroll
    for (int j = 0; j < C16_AHEAD; ++j) issue_row(j, j);

    const unsigned ring_lds = static_cast<unsigned>(reinterpret_cast<uintptr_t>((__attribute__((address_space(3))) char*)smem));
    unsigned boff[3][4];
#pragma unroll
    for (int kx = 0; kx < 3; ++kx)
#pragma unroll
        for (int q = 0; q < 4; ++q) {
            const int p = 14 * wave + i16 + kx;
            boff[kx][q] = ring_lds + static_cast<unsigned>(p * 256 + (((4 * q + kg) ^ p16_swz(p)) << 4));
        }
    const f32x4v sc = *reinterpret_cast<const f32x4v*>(a.ptab + 4 * kg);
    const f32x4v sh = *reinterpret_cast<const f32x4v*>(a.ptab + P16_COUT + 4 * kg);
    constexpr int OOB = 0x40000000;
    const int xo = P16_BLKO * cb + 7 * wave + (i16 >> 1);
    const int voff_lane = ((i16 & 1) == 0 && i16 <= 12 && xo < a.Wo) ? (xo * P16_COUT + 4 * kg) * 2 : OOB;
    const int out_row_bytes = a.Wo * P16_COUT * 2;
    const char* out_row = reinterpret_cast<const char*>(a.out + static_cast<int64_t>(n) * a.Ho * a.Wo * P16_COUT) +
                          static_cast<int64_t>(yo0) * out_row_bytes;
    float hprev[4] = {0.f, 0.f, 0.f, 0.f}, q0[4] = {0.f, 0.f, 0.f, 0.f};

    wait_vmcnt<0>();
#pragma unroll
    for (int c = 0; c < P16_KC; ++c) {
        if (c < 32)
            asm volatile("" : "+a"(wr[c]));
        else
            asm volatile("" : "+v"(wr[c]));
    }
    // One step = one conv row s.  vmcnt: the loads behind the 4 pieces of input row s + 2 are the 8 pieces of rows s + 3
    // and s + 4 (stores are not counted: see conv16_kernel).
    auto step = [&](auto PC, int s) __attribute__((always_inline)) {
        constexpr int P = decltype(PC)::value;
        wait_vmcnt<2 * P16_PIECES>();
        raw_barrier();
        issue_row(s + C16_AHEAD, (P + C16_AHEAD) % C16_NSLOT);
        f32x4v acc[3];                                   // one chain per kernel row, summed (ky 0 + ky 1) + ky 2: the order the
                                                         // one-launch back end (rn_backend.hip) adds its per-kernel-row partials in
        // operand reads run P16_RD - 1 chunks ahead of their MFMA behind counted waits
        constexpr int RD = P16_RD;
        i32x4 bq[RD];
        auto rd = [&](auto CC, i32x4& b) __attribute__((always_inline)) {
            constexpr int C = decltype(CC)::value;
            constexpr int tap = C / 4, q = C % 4, ky = tap / 3, kx = tap % 3;
            constexpr int slot_off = ((P + ky) % C16_NSLOT) * P16_ROWB;
            // (the instruction's offset field is 16 bits: the upper ring slots of the five-tile form lie beyond it)
            constexpr int imm = slot_off <= 65535 ? slot_off : 0;
            const unsigned ad = boff[kx][q] + static_cast<unsigned>(slot_off - imm);      // (named outside the asm: implicit capture)
            asm volatile("ds_read_b128 %0, %1 offset:%2" : "=v"(b) : "v"(ad), "n"(imm));
        };
        [&]<int... C>(std::integer_sequence<int, C...>) { (rd(IC<C>{}, bq[C]), ...); }(std::make_integer_sequence<int, RD - 1>{});
        [&]<int... C>(std::integer_sequence<int, C...>) {
            (([&] {
                 const f32x4v zero = {0.f, 0.f, 0.f, 0.f};
                 if constexpr (C + RD - 1 < P16_KC) rd(IC<(C + RD - 1 < P16_KC ? C + RD - 1 : 0)>{}, bq[(C + RD - 1) % RD]);
                 constexpr int newer = (P16_KC - 1 - C) < RD - 1 ? (P16_KC - 1 - C) : RD - 1;
                 asm volatile("s_waitcnt lgkmcnt(%1)" : "+v"(bq[C % RD]) : "n"(newer));
                 acc[C / 12] = mfma16<DT>(wr[C], bq[C % RD], C % 12 == 0 ? zero : acc[C / 12]);
             }()),
             ...);
        }(std::make_integer_sequence<int, P16_KC>{});
        // ReLU6 -> vertical 4-row sums on the odd rows -> horizontal 4-column sums inside the 16-lane DPP row -> BN
        float v[4];
#pragma unroll
        for (int j = 0; j < 4; ++j) v[j] = relu6f((acc[0][j] + acc[1][j]) + acc[2][j]);
        int vo = OOB;
        float y[4] = {0.f, 0.f, 0.f, 0.f};
        if constexpr ((P & 1) == 1) {
#pragma unroll
            for (int j = 0; j < 4; ++j) {
                const float pq = hprev[j] + v[j];
                const float t = q0[j] + pq;
                q0[j] = pq;
                const float u = t + row_next<1>(t);
                const float H = u + row_next<2>(u);
                y[j] = fmaf(H, sc[j], sh[j]);
            }
            if (s >= 3) vo = voff_lane;
        } else {
#pragma unroll
            for (int j = 0; j < 4; ++j) hprev[j] = v[j];
        }
        const __amdgpu_buffer_rsrc_t rs = __builtin_amdgcn_make_buffer_rsrc(const_cast<char*>(out_row), 0, out_row_bytes, 0x00020000);
        const i32x2v d = {static_cast<int>(pack2<DT>(y[0], y[1])), static_cast<int>(pack2<DT>(y[2], y[3]))};
        __builtin_amdgcn_raw_buffer_store_b64(d, rs, vo, 0, 0);
        if ((P & 1) == 1 && s >= 3) out_row += out_row_bytes;
    };
    int s = 0;
    for (; s + C16_NSLOT - 1 < nconv; s += C16_NSLOT) {
        [&]<int... I>(std::integer_sequence<int, I...>) { (step(IC<I>{}, s + I), ...); }(std::make_integer_sequence<int, C16_NSLOT>{});
    }
    [&]<int... I>(std::integer_sequence<int, I...>) {
        ((s + I < nconv ? (step(IC<I>{}, s + I), 0) : 0), ...);
    }(std::make_integer_sequence<int, C16_NSLOT - 1>{});
    wait_vmcnt<0>();
}

}  // namespace

bool rn_conv16_supported(int cin, int cout, int pool_k, bool res) { return cin == C16_CIN && cout == C16_COUT && pool_k == 0 && !res; }

int rn_conv16_colblocks(int out_side) { return (out_side + C16_BLKW - 1) / C16_BLKW; }

// A-operand fragments: frag[chunk c][cout tile t][lane][j] = W[k = 32 c + 8 (lane / 16) + j][cout = 16 t + lane % 16]
// (k = tap * 64 + channel, the HWIO order of the checkpoint)
void rn_conv16_pack(const float* w_hwio, int dtype, unsigned short (*cvt_bf16)(float), unsigned short (*cvt_f16)(float),
                    std::vector<unsigned short>* out) {
    out->assign(static_cast<size_t>(C16_KC) * 8 * 64 * 8, 0);
    for (int c = 0; c < C16_KC; ++c)
        for (int t = 0; t < 8; ++t)
            for (int l = 0; l < 64; ++l)
                for (int j = 0; j < 8; ++j) {
                    const int k = 32 * c + 8 * (l >> 4) + j, co = 16 * t + (l & 15);
                    const float v = w_hwio[static_cast<size_t>(k) * C16_COUT + co];
                    (*out)[((static_cast<size_t>(c) * 8 + t) * 64 + l) * 8 + j] = dtype == RN_DTYPE_BF16 ? cvt_bf16(v) : cvt_f16(v);
                }
}

int rn_conv16_launch(int dtype, hipStream_t s, const Conv16Args& a, int n) {
    auto launch = [&](auto kern) -> int {
        hipLaunchKernelGGL(kern, dim3(a.n_bands * a.n_colblocks, n), dim3(256), C16_LDS, s, a);
        RN_CHECK_LAUNCH();
        return RN_OK;
    };
    if (dtype == RN_DTYPE_BF16) return launch(conv16_kernel<RN_DTYPE_BF16>);
    return launch(conv16_kernel<RN_DTYPE_F16>);
}

bool rn_conv16p_supported(int cin, int cout, int pool_k, int pool_s, bool res) {
    return cin == P16_CIN && cout == P16_COUT && pool_k == 4 && pool_s == 2 && !res;
}

// tiles per workgroup for an output row of `out_side` columns, and the column blocks that gives
static int conv16p_nt(int out_side) { return out_side <= P16<3>::BLKO ? 3 : 5; }
int rn_conv16p_wgs_per_cu(int out_side) { return conv16p_nt(out_side) == 3 ? 2 : 1; }      // 72 KB / 123 KB of LDS per workgroup
int rn_conv16p_colblocks(int out_side) {
    const int blko = 7 * conv16p_nt(out_side);
    return (out_side + blko - 1) / blko;
}

// A-operand fragments: frag[chunk c][lane][j] = W[k = 32 c + 8 (lane / 16) + j][cout = lane % 16]
void rn_conv16p_pack(const float* w_hwio, int dtype, unsigned short (*cvt_bf16)(float), unsigned short (*cvt_f16)(float),
                     std::vector<unsigned short>* out) {
    out->assign(static_cast<size_t>(P16_KC) * 64 * 8, 0);
    for (int c = 0; c < P16_KC; ++c)
        for (int l = 0; l < 64; ++l)
            for (int j = 0; j < 8; ++j) {
                const int k = 32 * c + 8 * (l >> 4) + j, co = l & 15;
                const float v = w_hwio[static_cast<size_t>(k) * P16_COUT + co];
                (*out)[(static_cast<size_t>(c) * 64 + l) * 8 + j] = dtype == RN_DTYPE_BF16 ? cvt_bf16(v) : cvt_f16(v);
            }
}

int rn_conv16p_launch(int dtype, hipStream_t s, const Conv16Args& a, int n) {
    const int nt = conv16p_nt(a.Wo);
    auto launch = [&](auto kern) -> int {
        static std::atomic<unsigned long long> attr_devices{0};     // 72 KB of dynamic LDS: per device and instantiation
        int dev = 0;
        RN_HIP(hipGetDevice(&dev));
        if (!(attr_devices.load(std::memory_order_acquire) >> (dev & 63) & 1ull)) {
            RN_HIP(hipFuncSetAttribute(reinterpret_cast<const void*>(kern), hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024));
            attr_devices.fetch_or(1ull << (dev & 63), std::memory_order_release);
        }
        hipLaunchKernelGGL(kern, dim3(a.n_bands * a.n_colblocks, n), dim3(64 * nt), static_cast<size_t>(nt == 3 ? P16<3>::LDS : P16<5>::LDS), s, a);
        RN_CHECK_LAUNCH();
        return RN_OK;
    };
    if (nt == 3) {
        if (dtype == RN_DTYPE_BF16) return launch(conv16p_kernel<RN_DTYPE_BF16, 3>);
        return launch(conv16p_kernel<RN_DTYPE_F16, 3>);
    }
    if (dtype == RN_DTYPE_BF16) return launch(conv16p_kernel<RN_DTYPE_BF16, 5>);
    return launch(conv16p_kernel<RN_DTYPE_F16, 5>);
}
