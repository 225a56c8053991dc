// The back end of the 16-bit path in ONE launch: stage 6 (64 -> 128, no pool) -> stage 7 (128 -> 16, pool 4/2) -> stages 8 + 9
// (16 -> 16, pool 4/2, the second with the block's residual) -> flatten -> dense head -> softmax / argmax
// (reference network.py:229-237, :44-45).  One workgroup per image; neither s6.bn nor s7.bn reaches HBM at default flags.
//
// Before (round 3): three launches -- stage6x_kernel 72 us, conv16p_kernel 30 us, tail_kernel 26 us at batch 256 -- each with one
// workgroup (or two) per image: stage 7 was bound by re-reading stage 6's output from HBM (138 MB in 30 us = 4.6 TB/s), the
// tail by launch and staging latency.
//
// Workgroup = 11 waves:
//   waves 0-7   stage 6 exactly as rn_stage6x.hip (row-register blocking: wave = 16 couts x the row's three 16-pixel tiles, the
//               newest input row feeds three live accumulator sets), but a finished conv row (ReLU6 + BN, 16-bit) goes into a
//               four-row LDS ring ("mid") instead of HBM, in the 256-byte-pixel layout and chunk swizzle conv16p reads;
//   waves 8-10  stage 7 with conv16p_kernel's arithmetic (16-pixel tiles x all 16 couts, 36 K-chunks, DPP pooling), split by KERNEL
//               ROW: wave 8 + ky holds the 12 weight fragments of kernel row ky in registers (all 36 are 144 per lane: three waves
//               per SIMD leave 168) and turns the NEWEST mid row r into the partial sums of conv row r - ky for the three tiles
//               (36 MFMAs); the partials meet in LDS, and one step later wave 8 + t adds the three partials of tile t in the
//               fixed order (P0 + P1) + P2 -- conv16p_kernel sums the same way -- pools them and writes the pooled row into an
//               LDS image.  Only the newest mid row is ever read: two ring rows.  (First form: one wave per tile reading all
//               36 weight fragments from LDS every row: 216 KB of LDS reads per step next to stage 6's 144 slowed the loop
//               from 73 to 100 us.)
//   waves 0-3   then run the tail (rn_tail_body.h) on that image.
// One s_barrier per step; stage 7 adds a third wave to three of the four SIMDs whose matrix pipe stage 6 left 45 % idle.
// Same arithmetic, operand values and summation order as the three launches: bit-identical (tests/test_hip_fused.py).
#include "rn_tail_body.h"

#include <atomic>
#include <utility>

using namespace rnk;

namespace {

constexpr int B_NS = 4, B_AHEAD = 3;             // stage-6 input ring: newest row + 3 in flight
constexpr int B_ROW6 = 52 * 128;                 // bytes per input ring row (64 channels x 16 bit, 52 pixels)
constexpr int B_WMIN = 35;                       // narrowest stage-6 input row (rn_stage6x.hip: 35 .. 50 columns)
constexpr int B_NM = 2;                          // mid ring rows: the one stage 7 reads + the one being written
constexpr int B_ROWM = 48 * 256;                 // bytes per mid row (128 channels x 16 bit, 48 pixels)
constexpr int B_NP = 4;                          // partial-sum slots (conv rows in flight between their first partial and their sum)
constexpr int B_LAG = 6;                         // step s finishes stage-7 conv row s - B_LAG
constexpr int B_OFF_RING6 = 0;
constexpr int B_OFF_MID = B_OFF_RING6 + B_NS * B_ROW6;
constexpr int B_OFF_PART = B_OFF_MID + B_NM * B_ROWM;         // partial sums [B_NP conv rows][3 kernel rows][3 tiles] x 1 KB (fp32 16x16)
constexpr int B_OFF_X7 = B_OFF_PART + B_NP * 9 * 1024;        // stage-7 output image [So7][So7][16], So7 <= 22
constexpr int B_OFF_XA = B_OFF_X7 + 16384;                    // tail: second step's output [16][16][16]
constexpr int B_OFF_XB = B_OFF_XA + 16 * 16 * 16 * 2;         // third step's output [8][8][16]
constexpr int B_OFF_BUF0 = B_OFF_XB + 8 * 8 * 16 * 2;
constexpr int B_OFF_SMALL = B_OFF_BUF0 + 8 * 8 * 16 * 4;
constexpr int B_OFF_WL = B_OFF_SMALL + 2 * 64 * 4;
constexpr int B_OFF_TTAB = B_OFF_WL + T_W_LDS * 4;
constexpr int B_LDS = B_OFF_TTAB + 144 * 4;
static_assert(B_LDS <= 160 * 1024, "LDS budget");

struct BackendArgs {
    // stage 6
    const unsigned short* in;     // [N, W, W, 64]
    const i32x4* wfrag6;          // rn_stage6x_pack
    const float* ptab6;           // folded BN: scale[128], shift[128]
    const float* cstart6;         // K48 (null otherwise): wfrag6 = rn_stage6x_pack48's fragments, cstart6[cout] = the constant input channels' sum
    int W, Wo;                    // stage-6 input / output side
    // stage 7
    const i32x4* wfrag7;          // rn_conv16p_pack: [36][64 lanes]
    const float* ptab7;           // folded BN: scale[16] (inv / 16), shift[16]
    int So7;
    TailArgs tail;
};

__device__ __forceinline__ unsigned lds_addr(const void* p) {
    return static_cast<unsigned>(reinterpret_cast<uintptr_t>((__attribute__((address_space(3))) const char*)p));
}
__device__ __forceinline__ int swz8(int pix) { return pix & 7; }                   // stage-6 input ring (rn_stage6x.hip)
__device__ __forceinline__ int swz16(int pix) { return (pix & 7) << 1; }           // mid ring (conv16p: 16 chunks per pixel)

using i32x2 = __attribute__((ext_vector_type(2))) int;

template <int DT>
__device__ __forceinline__ f32x4 mfma16(i32x4 a, i32x4 b, f32x4 c) {
    if constexpr (DT == RN_DTYPE_BF16)
        return __builtin_amdgcn_mfma_f32_16x16x32_bf16(__builtin_bit_cast(bf16x8, a), __builtin_bit_cast(bf16x8, b), c, 0, 0, 0);
    else
        return __builtin_amdgcn_mfma_f32_16x16x32_f16(__builtin_bit_cast(f16x8, a), __builtin_bit_cast(f16x8, b), c, 0, 0, 0);
}

template <int DT, bool K48>
__global__ __launch_bounds__(704) void backend_kernel(const BackendArgs a) {
    constexpr int NF = K48 ? 5 : 6;                 // stage-6 operand fragments per tile and row step (rn_stage6x.hip: K48)
    extern __shared__ __attribute__((aligned(64))) char smem[];
    const int tid = threadIdx.x;
    const int lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int px16 = lane & 15, g = lane >> 4;
    const int n = blockIdx.x;
    const int W = a.W, Wo = a.Wo;
    const int nin = Wo + 2;                         // stage-6 input rows = steps of its loop
    const int nconv7 = Wo - 2;                      // stage-7 conv rows
    const int nsteps = max(nin, nconv7 + B_LAG);
    const f32x4 zero4 = {0.f, 0.f, 0.f, 0.f};

    char* const ring6 = smem + B_OFF_RING6;
    char* const mid = smem + B_OFF_MID;
    const unsigned mid_lds = lds_addr(mid);
    // ---- prologue work shared by all waves: zero the input-ring pixels the row DMA never writes, stage the dense kernels in LDS
    for (int i = tid; i < B_NS * (52 - B_WMIN) * 8; i += 704) {
        const int slot = i / ((52 - B_WMIN) * 8), rest = i % ((52 - B_WMIN) * 8);
        const int p = B_WMIN + rest / 8, c = rest % 8;
        if (p >= W) *reinterpret_cast<i32x4*>(ring6 + slot * B_ROW6 + p * 128 + c * 16) = i32x4{0, 0, 0, 0};
    }
    int w_off[RN_MAX_DENSE];
    tail_stage_dense_dma(a.tail.head, reinterpret_cast<float*>(smem + B_OFF_WL), w_off, wave, lane, 11);

    if (wave < 8) {
        // =============================================================== stage 6 (rn_stage6x.hip), output into the mid ring
        const unsigned ring_lds = lds_addr(ring6);
        i32x4 wf[3 * NF];
#pragma unroll
        for (int f = 0; f < 3 * NF; ++f) {
            const i32x4* src = a.wfrag6 + (f * 8 + wave) * 64 + lane;
            asm volatile("global_load_dwordx4 %0, %1, off" : "=v"(wf[f]) : "v"(src) : "memory");
        }
        const char* const in_img = reinterpret_cast<const char*>(a.in + static_cast<int64_t>(n) * W * W * 64);
        const int row_bytes = W * 128;
        const unsigned long long dma_mask = (1ull << W) - 1ull;
        unsigned goff;
        {
            const int q = wave * W + min(lane, W - 1);
            const int p = q >> 3, c = q & 7;
            goff = static_cast<unsigned>(p * 128 + ((c ^ swz8(p)) << 4));
        }
        auto issue_row = [&](int y, int slot) __attribute__((always_inline)) {
            const char* row = in_img + static_cast<int64_t>(min(y, nin - 1)) * row_bytes;
            unsigned o = goff;
            asm volatile("" : "+v"(o));
            dma16_masked(row + o, ring6 + slot * B_ROW6 + wave * W * 16, dma_mask);
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
            for (int j = 0; j < 2; ++j) {
                const int p = px16 + (j == 0 ? (g >> 1) : 2);
                base[j][1] = ring_lds + static_cast<unsigned>(p * 128 + (((4 + (g & 1)) ^ swz8(p)) << 4));
            }
            cst4 = *reinterpret_cast<const f32x4*>(a.cstart6 + 16 * wave + 4 * g);
        }
        // a lane holds couts 16 wave + 4 g .. + 3 of pixel 16 k + px16: 8 bytes = half (g & 1) of chunk 2 wave + g / 2 of the pixel
        unsigned moff[3];
#pragma unroll
        for (int k = 0; k < 3; ++k) {
            const int xo = 16 * k + px16;
            moff[k] = mid_lds + static_cast<unsigned>(xo * 256 + (((2 * wave + (g >> 1)) ^ swz16(xo)) << 4) + (g & 1) * 8);
        }
        const f32x4 sc = *reinterpret_cast<const f32x4*>(a.ptab6 + 16 * wave + 4 * g);
        const f32x4 sh = *reinterpret_cast<const f32x4*>(a.ptab6 + 128 + 16 * wave + 4 * g);
        f32x4 acc[3][3];
#pragma unroll
        for (int r3 = 0; r3 < 3; ++r3)
#pragma unroll
            for (int k = 0; k < 3; ++k) acc[r3][k] = zero4;
#pragma unroll
        for (int j = 0; j < B_AHEAD; ++j) issue_row(j, j);
        wait_vmcnt<0>();
#pragma unroll
        for (int f = 0; f < 3 * NF; ++f) asm volatile("" : "+v"(wf[f]));
        lds_barrier();

        int slot_cur = 0;
        auto step = [&](auto RC, int s) __attribute__((always_inline)) {
            constexpr int R = decltype(RC)::value;
            constexpr int iN = R, iM = (R + 2) % 3, iO = (R + 1) % 3;
            // (vmcnt: one DMA piece per step and nothing else -- no stores: the rows go to LDS -- so the pieces of rows s + 1 and
            //  s + 2 are what may still be in flight behind row s's)
            wait_vmcnt<B_AHEAD - 1>();
            raw_barrier();
            {
                int sl = slot_cur + B_AHEAD;
                sl = sl >= B_NS ? sl - B_NS : sl;
                issue_row(s + B_AHEAD, sl);
            }
            const unsigned so = static_cast<unsigned>(slot_cur * B_ROW6);
            unsigned bc[3][2];
#pragma unroll
            for (int kx = 0; kx < 3; ++kx)
#pragma unroll
                for (int ch = 0; ch < 2; ++ch) bc[kx][ch] = base[kx][ch] + so;
            const int j = s - 2;                                            // conv row completed by this step
            const int jr = min(max(j, 0), Wo - 1);
            const unsigned mslot = static_cast<unsigned>((jr & (B_NM - 1)) * B_ROWM);       // (rows j < 0 write row 0's slot early: harmless)
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
                const unsigned ma = moff[k] + mslot;
                asm volatile("ds_write_b64 %0, %1" ::"v"(ma), "v"(d) : "memory");
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
            if (s < nin) {
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
            }
            slot_cur = slot_cur == B_NS - 1 ? 0 : slot_cur + 1;
            asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
        };
        int s = 0;
        for (; s + 2 < nsteps; s += 3) {
            step(IC<0>{}, s);
            step(IC<1>{}, s + 1);
            step(IC<2>{}, s + 2);
        }
        const int rem = nsteps - s;
        if (rem > 0) step(IC<0>{}, s);
        if (rem > 1) step(IC<1>{}, s + 1);
        wait_vmcnt<0>();
    } else {
        // =============================================================== stage 7 (conv16p_kernel's arithmetic), split by kernel row
        const int t = wave - 8;                                  // kernel row of the MFMA phase, pixel tile of the pooling phase
        const int i16 = px16, kg = g;
        i32x4 w7[12];                                            // fragments of kernel row t: chunk c = kx * 4 + q
#pragma unroll
        for (int c = 0; c < 12; ++c) {
            const i32x4* src = a.wfrag7 + (t * 12 + c) * 64 + lane;
            asm volatile("global_load_dwordx4 %0, %1, off" : "=v"(w7[c]) : "v"(src) : "memory");
        }
        unsigned boff[3][3][4];                                  // [tile][kx][channel quarter]
#pragma unroll
        for (int k = 0; k < 3; ++k)
#pragma unroll
            for (int kx = 0; kx < 3; ++kx)
#pragma unroll
                for (int q = 0; q < 4; ++q) {
                    const int p = min(14 * k + i16 + kx, 47);
                    boff[k][kx][q] = mid_lds + static_cast<unsigned>(p * 256 + (((4 * q + kg) ^ swz16(p)) << 4));
                }
        char* const part = smem + B_OFF_PART;
        const unsigned part_lds = lds_addr(part) + static_cast<unsigned>(lane) * 16u;
        const f32x4 sc = *reinterpret_cast<const f32x4*>(a.ptab7 + 4 * kg);
        const f32x4 sh = *reinterpret_cast<const f32x4*>(a.ptab7 + 16 + 4 * kg);
        const int So7 = a.So7;
        const int xo = 7 * t + (i16 >> 1);
        const bool lane_out = (i16 & 1) == 0 && i16 <= 12 && xo < So7;
        char* const x7 = smem + B_OFF_X7;
        float hprev[4] = {0.f, 0.f, 0.f, 0.f}, q0[4] = {0.f, 0.f, 0.f, 0.f};
        wait_vmcnt<0>();
#pragma unroll
        for (int c = 0; c < 12; ++c) asm volatile("" : "+v"(w7[c]));
        lds_barrier();                                           // (the prologue barrier of the stage-6 waves)
        for (int s = 0; s < nsteps; ++s) {
            raw_barrier();
            // ---- pooling phase first (its reads are of partials written in earlier steps): conv row je is complete
            const int je = s - B_LAG;
            if (je >= 0 && je < nconv7) {
                const unsigned pa = part_lds + static_cast<unsigned>(((je & (B_NP - 1)) * 9 + t) * 1024);
                f32x4 p0, p1, p2;
                asm volatile("ds_read_b128 %0, %1" : "=v"(p0) : "v"(pa));
                asm volatile("ds_read_b128 %0, %1 offset:3072" : "=v"(p1) : "v"(pa));
                asm volatile("ds_read_b128 %0, %1 offset:6144" : "=v"(p2) : "v"(pa));
                asm volatile("s_waitcnt lgkmcnt(0)" : "+v"(p0), "+v"(p1), "+v"(p2));
                float v[4];
#pragma unroll
                for (int j = 0; j < 4; ++j) v[j] = relu6f((p0[j] + p1[j]) + p2[j]);
                if (je & 1) {
                    float y[4];
#pragma unroll
                    for (int j = 0; j < 4; ++j) {
                        const float pq = hprev[j] + v[j];
                        const float tt = q0[j] + pq;
                        q0[j] = pq;
                        const float u = tt + row_next<1>(tt);
                        const float H = u + row_next<2>(u);
                        y[j] = fmaf(H, sc[j], sh[j]);
                    }
                    if (je >= 3) {
                        const int r = (je - 3) >> 1;                 // pooled row
                        const i32x2 d = {static_cast<int>(pack2<DT>(y[0], y[1])), static_cast<int>(pack2<DT>(y[2], y[3]))};
                        if (lane_out) *reinterpret_cast<i32x2*>(x7 + ((r * So7 + xo) * 16 + 4 * kg) * 2) = d;
                    }
                } else {
#pragma unroll
                    for (int j = 0; j < 4; ++j) hprev[j] = v[j];
                }
            }
            // ---- MFMA phase: the newest visible mid row r (stage 6 wrote it during the previous step) as kernel row t of conv
            // row r - t, for the three tiles
            const int r = s - 3;
            const int jc = r - t;
            if (r >= 0 && r < Wo && jc >= 0 && jc < nconv7) {
                const unsigned so = static_cast<unsigned>((r & (B_NM - 1)) * B_ROWM);
                const int pw_off = ((jc & (B_NP - 1)) * 9 + 3 * t) * 1024 + lane * 16;
                auto tile = [&](auto KC) __attribute__((always_inline)) {
                    constexpr int k = decltype(KC)::value;
                    i32x4 bq[12];
                    auto rdb = [&](auto CC) __attribute__((always_inline)) {
                        constexpr int C = decltype(CC)::value;
                        auto& dst = bq[C];                                  // (named outside the asm: implicit capture)
                        const unsigned ad = boff[k][C / 4][C % 4] + so;
                        asm volatile("ds_read_b128 %0, %1" : "=v"(dst) : "v"(ad));
                    };
                    [&]<int... C>(std::integer_sequence<int, C...>) { (rdb(IC<C>{}), ...); }(std::make_integer_sequence<int, 12>{});
                    f32x4 acc = zero4;
                    auto mm = [&](auto CC) __attribute__((always_inline)) {
                        constexpr int C = decltype(CC)::value;
                        auto& src = bq[C];
                        asm volatile("s_waitcnt lgkmcnt(%1)" : "+v"(src) : "n"(11 - C));
                        acc = mfma16<DT>(w7[C], src, acc);                  // chunks of the kernel row in order: kx * 4 + q
                    };
                    [&]<int... C>(std::integer_sequence<int, C...>) { (mm(IC<C>{}), ...); }(std::make_integer_sequence<int, 12>{});
                    // (a compiler-visible store: the matrix core's result registers need wait states before an LDS instruction may
                    //  read them, and hipcc pads no hazards between its own instructions and the inside of an asm string)
                    *reinterpret_cast<f32x4*>(part + pw_off + k * 1024) = acc;
                };
                tile(IC<0>{});
                tile(IC<1>{});
                tile(IC<2>{});
            }
            asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
        }
        wait_vmcnt<0>();
    }
    // =================================================================== tail on the first four waves
    __syncthreads();
    if (wave >= 4) return;
    tail_phase<DT>(a.tail, reinterpret_cast<const unsigned short*>(smem + B_OFF_X7), reinterpret_cast<unsigned short*>(smem + B_OFF_XA),
                   reinterpret_cast<unsigned short*>(smem + B_OFF_XB), reinterpret_cast<float*>(smem + B_OFF_BUF0),
                   reinterpret_cast<float(*)[64]>(smem + B_OFF_SMALL), reinterpret_cast<const float*>(smem + B_OFF_WL), w_off, reinterpret_cast<float*>(smem + B_OFF_TTAB), n, tid);
}

}  // namespace

// the fused back end covers: stage 6 as rn_stage6x (one column block), stage 7 as conv16p in one column block (<= 3 tiles), a
// tail rn_tail_supported accepts
bool rn_backend_supported(const rn_handle* h) {
    const size_t ns = h->stages.size();
    if (ns < 5 || !rn_tail_supported(h)) return false;
    const StagePlan& s6 = h->stages[ns - 4];
    const StagePlan& s7 = h->stages[ns - 3];
    int ncb, xo0[4], wo[4];
    if (!rn_stage6x_supported(s6.cin, s6.cout, s6.pool_k, s6.skip_stage >= 0, s6.in_side) || !rn_stage6x_plan(s6.out_side, &ncb, xo0, wo) ||
        ncb != 1)
        return false;
    // (stage 7 runs on three 16-pixel tiles = 21 pooled columns here)
    if (!rn_conv16p_supported(s7.cin, s7.cout, s7.pool_k, s7.pool_s, s7.skip_stage >= 0) || s7.out_side > 21) return false;
    if (s7.in_side != s6.out_side || s6.node_bn2 >= 0 || s7.node_bn2 >= 0) return false;
    for (size_t k = ns - 2; k < ns; ++k)
        if (h->stages[k].skip_stage == static_cast<int>(ns) - 4) return false;      // stage 6's output has no other consumer
    return true;
}

int rn_backend_launch(rn_handle* h, const i32x4* wfrag6, const float* ptab6, const float* cstart6, const i32x4* wfrag7, const float* ptab7,
                      const i32x4* wfrag_a, const i32x4* wfrag_b, const HeadArgs& head, int n, float* d_probs, int64_t* d_ids) {
    const size_t ns = h->stages.size();
    const StagePlan& s5 = h->stages[ns - 5];
    const StagePlan& s6 = h->stages[ns - 4];
    const StagePlan& s7 = h->stages[ns - 3];
    BackendArgs a{};
    a.in = static_cast<const unsigned short*>(h->nodes[s5.node_bn2 >= 0 ? s5.node_bn2 : s5.node_bn].ptr);
    a.wfrag6 = wfrag6;
    a.ptab6 = ptab6;
    a.cstart6 = cstart6;
    a.W = s6.in_side;
    a.Wo = s6.out_side;
    a.wfrag7 = wfrag7;
    a.ptab7 = ptab7;
    a.So7 = s7.out_side;
    rn_tail_fill_args(h, wfrag_a, wfrag_b, head, d_probs, d_ids, &a.tail);
    auto launch = [&](auto kern) -> int {
        static std::atomic<unsigned long long> attr_devices{0};
        int dev = 0;
        RN_HIP(hipGetDevice(&dev));
        if (!(attr_devices.load(std::memory_order_acquire) >> (dev & 63) & 1ull)) {
            RN_HIP(hipFuncSetAttribute(reinterpret_cast<const void*>(kern), hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024));
            attr_devices.fetch_or(1ull << (dev & 63), std::memory_order_release);
        }
        hipLaunchKernelGGL(kern, dim3(n), dim3(704), B_LDS, h->stream, a);
        RN_CHECK_LAUNCH();
        return RN_OK;
    };
    if (cstart6) {
        if (h->dtype == RN_DTYPE_BF16) return launch(backend_kernel<RN_DTYPE_BF16, true>);
        return launch(backend_kernel<RN_DTYPE_F16, true>);
    }
    if (h->dtype == RN_DTYPE_BF16) return launch(backend_kernel<RN_DTYPE_BF16, false>);
    return launch(backend_kernel<RN_DTYPE_F16, false>);
}
