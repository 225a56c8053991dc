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
// Geometry: a workgroup owns one image x one band of output rows x one column block.  W <= 215 (the 224 x 224 network):
// the block is the whole row -- 4 x A row + 4 x B row + the skip rows = 155 KB of LDS.  Wider inputs are cut into column
// blocks of 193..215 input columns (rn_stage23_plan; 600 x 600: three), each re-reading its 10 halo columns.  Per step t
// one row of A arrives by LDS-DMA, the first stage finishes B row t-5 for all eight 32-column tiles (written to the B
// ring with ds_write_b64), the second stage finishes output row t-11 from the B rows of earlier steps.  One s_barrier
// per step.  The residual's skip rows are fetched again (L2) into small wave-private rings: each second-stage wave
// stages only the 64 skip columns its own two tiles interpolate from, so nothing but the A and B rings is shared.
//
// The arithmetic of both stages is instruction-for-instruction that of stage_rw_kernel's POOLM variants
// (rn_stage_rw.hip): results are bit-identical to the two-launch path
// (tests/test_hip_fused.py::test_cross_stage_fusion_is_bit_identical_to_stage_launches).
#include "rn_fused.h"
#include "rn_stage.h"

#include <atomic>
#include <utility>

using namespace rnk;

// Register class of the lane-constant MFMA operands (weight fragments, pooling band matrix, interpolation weights).
// With NO "a" constraint anywhere in the kernel LLVM's attributor marks it amdgpu-no-agpr and the allocator gets one flat
// file of 256 VGPRs per wave; with any "a" constraint it splits the wave's 256 registers 128 + 128 between architectural
// and accumulator registers, and everything the VALU touches has to fit the 128 (spilled to AGPRs and copied back:
// 300-500 v_accvgpr moves in this kernel).
#define RN_WREG_OUT(x) "=v"(x)
#define RN_WREG_IO(x) "+v"(x)

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

#ifdef RN_STAMPS
// diagnostic build only (tools/build_stamps.sh): s_memtime + its wait in one statement, fenced
__device__ __forceinline__ unsigned long long stamp23() {
    unsigned long long t;
    __builtin_amdgcn_sched_barrier(0);
    asm volatile("s_memtime %0\n\ts_waitcnt lgkmcnt(0)" : "=s"(t)::"memory");
    __builtin_amdgcn_sched_barrier(0);
    return t;
}
#endif

// Producer / consumer workgroup: 8 waves, two per SIMD.  Waves 0-3 ("producers") run the first stage of the pair for
// two column tiles each, fetch the A rows and write their B-row segments into the B ring; waves 4-7 ("consumers") run
// the second stage for two tiles each on the B rows of earlier steps.  Wave w and wave w + 4 share a SIMD, so every
// SIMD hosts one producer and one consumer: while one of them is in a VALU-heavy epilogue the other one's MFMA chain
// keeps the matrix pipe busy.  (Measured against a 4-wave form -- one wave per SIMD doing all four jobs with the
// epilogue micro-ops placed between its own MFMAs, in the history: 5800 cycles per step against 5200 here.)  The
// consumers run at s_setprio 1.  Work is balanced between the roles by letting the producer issue the partner
// consumer's regular skip-row DMA (see the schedule below): barrier wait 130 (producer) / 520 (consumer) cycles per
// step, profiles/r2_c_stamps.txt.  Both roles use 254 of 256 registers: check tools/spills.sh after ANY edit -- a
// handful of spilled registers costs 25 % and hipcc says nothing.
template <int DT>
__global__ __launch_bounds__(512, 2) void stage23pc_kernel(const Stage23Args a) {
    constexpr int KC = 18, BAHEAD = 4;
    extern __shared__ __attribute__((aligned(64))) char smem[];
    const int tid = threadIdx.x;
    const int lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int wq = wave & 3;                             // tile pair of this wave
    const int r = lane & 31, hh = lane >> 5;
    const int cblk = blockIdx.x % a.n_cblocks, band = blockIdx.x / a.n_cblocks, n = blockIdx.y;
    // block-local widths (what the rings hold) and the full sides (row pitches, vertical clamps, residual tables)
    const int Win = a.W, Hout = a.Wo, x0 = a.cb_x0[cblk];
    const int Wo = a.cb_wo[cblk], W = Wo + 10, Wb = W - 5;
    const int yo0 = band * a.rows_per_band;
    const int nrows = min(Hout, yo0 + a.rows_per_band) - yo0;
    const int nsteps = nrows + F_LAG;

    float* const tab = reinterpret_cast<float*>(smem);
    for (int i = tid; i < F_NTAB; i += 512) tab[i] = a.ptab[i];
    char* const ringA = smem + F_RINGA_OFF;
    char* const ringB = smem + F_RINGB_OFF;
    const unsigned ringA_lds = lds_addr(ringA), ringB_lds = lds_addr(ringB);
    const char* const in_img = reinterpret_cast<const char*>(a.in + static_cast<int64_t>(n) * Win * Win * 32);
    const char* const in_blk = in_img + x0 * 64;          // first input column of this column block
    constexpr int OOB = 0x40000000;

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
                w |= ((x >= r && x < r + 4) ? 0x3C00u : 0u) << (16 * e2);
            }
            pmw[c][d] = static_cast<int>(w);
        }
    asm volatile("" : RN_WREG_IO(pmw[0]), RN_WREG_IO(pmw[1]));

    // fragment read of K-chunk kc, conv row whose first input row sits in ring slot S0 (inline asm, counted lgkmcnt)
    // `hook(IC<I>)` runs right behind chain MFMA I: DMA issue and scalar bookkeeping ride in the MFMAs' shadow there instead
    // of standing in front of the chain (an LDS-DMA piece costs ~60-180 cycles to issue, a float -> readfirstlane round trip ~40)
    auto chain = [&](auto S0C, auto ROWBC, const unsigned (&base)[3][2], const i32x4 (&wr)[KC], f32x16& accn, auto&& hook) __attribute__((always_inline)) {
        constexpr int S0 = decltype(S0C)::value, ROWB_ = decltype(ROWBC)::value;
        const f32x16 zero = {0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f};
        i32x4 bq[KC];
        auto rd = [&](auto KCC, float dep) __attribute__((always_inline)) -> i32x4 {
            constexpr int kc = decltype(KCC)::value;
            constexpr int tap = kc / 2, cc = kc % 2, ky = tap / 3, kx = tap % 3;
            constexpr int off = ((S0 + ky) % 4) * ROWB_;
            i32x4 v;
            asm volatile("ds_read_b128 %0, %1 offset:%2" : "=v"(v) : "v"(base[kx][cc]), "n"(off), "v"(dep));
            return v;
        };
        [&]<int... I>(std::integer_sequence<int, I...>) { ((bq[I] = rd(IC<I>{}, 0.f)), ...); }(std::make_integer_sequence<int, BAHEAD>{});
        [&]<int... I>(std::integer_sequence<int, I...>) {
            (([&] {
                 if constexpr (I + BAHEAD < KC) bq[I + BAHEAD] = rd(IC<(I + BAHEAD < KC ? I + BAHEAD : 0)>{}, I == 0 ? 0.f : accn[0]);
                 constexpr int newer = (KC - 1 - I) < BAHEAD ? (KC - 1 - I) : BAHEAD;
                 asm volatile("s_waitcnt lgkmcnt(%1)" : "+v"(bq[I]) : "n"(newer));
                 accn = mfma32<DT>(bq[I], wr[I], I == 0 ? zero : accn);      // D'[pixel][cout]
                 hook(IC<I>{});
             }()),
             ...);
        }(std::make_integer_sequence<int, KC>{});
    };
    // ReLU6 -> fp16 pairs -> vertical pair sums -> pooled sums H on the matrix cores (state: hp, q0, q1 of the tile)
    auto pool = [&](auto PRC, const f32x16& acce, i32x4 (&hp)[2], i32x4 (&q0)[2], i32x4 (&q1)[2]) __attribute__((always_inline)) -> f32x16 {
        constexpr int PR = decltype(PRC)::value;
        i32x4 qp[2];
#pragma unroll
        for (int i2 = 0; i2 < 16; i2 += 2) {
            const int vp = static_cast<int>(pack2_relu6_sixth(acce[i2], acce[i2 + 1]));
            qp[i2 / 8][(i2 % 8) / 2] = pk_add_f16(hp[i2 / 8][(i2 % 8) / 2], vp);
            hp[i2 / 8][(i2 % 8) / 2] = vp;
        }
        const f32x16 zero = {0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f};
        i32x4(&qold)[2] = PR == 0 ? q0 : q1;
        f32x16 H = mfma32<RN_DTYPE_F16>(qold[0], pmw[0], zero);
        H = mfma32<RN_DTYPE_F16>(qold[1], pmw[1], H);
        H = mfma32<RN_DTYPE_F16>(qp[0], pmw[0], H);
        H = mfma32<RN_DTYPE_F16>(qp[1], pmw[1], H);
        qold[0] = qp[0];
        qold[1] = qp[1];
        return H;
    };

    // ---- residual skip rows (shared by both roles): consumer wave wq + 4 owns a private 3-row ring with the 64 columns of
    // A its two tiles interpolate from.  Schedule (both roles run the same arithmetic, so they agree without talking): at
    // the start of step t the ring holds rows ylo(t), ylo(t)+1 of that step's output row.  During step t the PRODUCER of
    // the same tile pair fetches the one new row step t+1 needs (into the slot of row ylo(t)-1, which nobody reads any
    // more; it lands before the producer's end-of-step wait and the barrier publishes it).  When the lo row jumps by 2
    // (1 step in 20) a second new row is needed whose slot is still being read: the CONSUMER fetches that one itself at
    // the start of step t+1 and waits for it before its first epilogue.
    const int xs0 = a.rlo[x0 + min(58 * wq, Wo - 1)];     // first skip column of this wave's tile pair (full-width index)
    // piece i of a skip row = pixels xs0 + lane / 4 + 16 i (the chunk swizzle has period 16 pixels): one lane offset, the
    // 1024 bytes per piece go into the scalar row base.  Pixels right of the image are NOT clamped: they read on into the
    // next row (the last row of the last image: into the 8 KB of slack behind every activation tensor, rn_api.hip) and
    // feed only output columns right of the image, which are never stored.
    const unsigned sk_goff0 = static_cast<unsigned>((xs0 + (lane >> 2)) * 64 + (((lane & 3) ^ swz4(lane >> 2)) << 4));
    char* const skw = smem + F_SKIP_OFF + wq * (F_NSK * F_SKROW);
    const unsigned skw_lds = lds_addr(skw);
    auto issue_skip_piece = [&](auto IC_, int y, int slot) __attribute__((always_inline)) {
        constexpr int i = decltype(IC_)::value;
        const char* row = in_img + static_cast<int64_t>(y) * static_cast<int64_t>(Win * 64);
        unsigned off = sk_goff0;
        asm volatile("" : "+v"(off));
        dma16(row + i * 1024 + off, skw + slot * F_SKROW + i * 1024);
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
    int sk_slot = F_NSK - 1;                          // its ring slot (row y lives in slot (y - ylo(yo0)) mod 3)
    auto ylo_step = [&](int t) __attribute__((always_inline)) {      // lo skip row of the output row step t finishes
        return vlerp_of(yo0 + min(max(t - F_LAG, 0), nrows - 1)).ylo;
    };

    if (wave < 4) {
        // =============================================================== producer: first stage, A ring -> B ring
        // weight fragments: loaded by inline asm straight into accumulator registers, all 18 in flight together (a
        // compiler-visible load pinned with "+a" got an s_waitcnt vmcnt(0) of its own: 18 serial L2 round trips per wave
        // in front of every band); retired by the wait_vmcnt<0>() below, ordered by the empty asm behind it
        i32x4 w2[KC];
#pragma unroll
        for (int kc = 0; kc < KC; ++kc) {
            const i32x4* src = a.wfrag2 + kc * 64 + lane;
            asm volatile("global_load_dwordx4 %0, %1, off" : RN_WREG_OUT(w2[kc]) : "v"(src) : "memory");
        }
        // ---- A rows: four DMA pieces per producer wave and row (a piece is 64 lanes x 16 B, the fourth is the masked W-192 tail)
        const int ptid = wq * 64 + lane;
        const int tailn = W - 192;
        const unsigned long long tail_mask = (1ull << tailn) - 1ull;
        // pieces 0..2 cover pixels ptid/4 + 64 i (< 192 <= W - 1: never clamped), and the chunk swizzle has period 16 pixels:
        // their source offsets differ by exactly 4096 bytes, which goes into the scalar row base -- one offset register
        const unsigned ld_goff = static_cast<unsigned>((ptid >> 2) * 64 + (((ptid & 3) ^ swz4(ptid >> 2)) << 4));
        unsigned ld_goff_tail;
        {
            const int q = 768 + tailn * wq + min(lane, tailn - 1);
            const int p = q >> 2, c = q & 3;
            ld_goff_tail = static_cast<unsigned>(min(p, W - 1) * 64 + ((c ^ swz4(p)) << 4));
        }
        auto issue_A_pieces = [&](auto I0, auto I1, const char* row, int slot) __attribute__((always_inline)) {
    #pragma unroll
            for (int i = decltype(I0)::value; i < decltype(I1)::value; ++i) {
                if (i < 3) {
                    unsigned off = ld_goff;
                    asm volatile("" : "+v"(off));
                    dma16(row + i * 4096 + off, ringA + slot * F_ROWA + (i * 256 + wq * 64) * 16);
                } else {
                    unsigned off = ld_goff_tail;
                    asm volatile("" : "+v"(off));
                    dma16_masked(row + off, ringA + slot * F_ROWA + (768 + tailn * wq) * 16, tail_mask);
                }
            }
        };
        const char* a_next = in_blk + static_cast<int64_t>(yo0) * (Win * 64);
        unsigned baseA[2][3][2], wbB[2];
#pragma unroll
        for (int T = 0; T < 2; ++T) {
            const int x_t = 29 * (2 * wq + T);
#pragma unroll
            for (int kx = 0; kx < 3; ++kx)
#pragma unroll
                for (int cc = 0; cc < 2; ++cc) {
                    const int ca = min(x_t + r + kx, W - 1);
                    baseA[T][kx][cc] = ringA_lds + static_cast<unsigned>(ca * 64 + (((cc * 2 + hh) ^ swz4(ca)) << 4));
                }
            const int xo = x_t + r;
            const int colw = (r < 29 && xo < Wb) ? xo : F_BDUMMY;
            wbB[T] = ringB_lds + static_cast<unsigned>(colw * 64 + (swz4(colw) << 4) + hh * 8);
        }
        i32x4 hp[2][2], q0[2][2], q1[2][2];
#pragma unroll
        for (int T = 0; T < 2; ++T) hp[T][0] = hp[T][1] = q0[T][0] = q0[T][1] = q1[T][0] = q1[T][1] = i32x4{0, 0, 0, 0};
        issue_A_pieces(IC<0>{}, IC<4>{}, a_next, 0);
        int ylo_nxt = ylo_step(0);                       // lo skip row of the output row the coming step finishes
        wait_vmcnt<0>();
#pragma unroll
        for (int kc = 0; kc < KC; ++kc) asm volatile("" : RN_WREG_IO(w2[kc]));
        lds_barrier();

        const unsigned tabl_lds = lds_addr(tab + 4 * hh);
        auto epi = [&](auto TC, auto PC, const f32x16& acce) __attribute__((always_inline)) {
            constexpr int T = decltype(TC)::value, P = decltype(PC)::value;
            // folded-BN table entries of the lane's 16 channels: all eight reads go out in front of the pooling (32 VALU + 4
            // dependent MFMAs cover their latency); compiler-visible loads sat behind the pooling, two per channel group, each
            // pair waited for with lgkmcnt(0) right before its use: four exposed LDS round trips per epilogue
            f32x4 tsc[4], tsh[4];
            {
                const unsigned ta = tabl_lds;                    // (named outside the asm: implicit capture)
                auto& t1 = tsc;
                auto& t2 = tsh;
#pragma unroll
                for (int g = 0; g < 4; ++g) {
                    asm volatile("ds_read_b128 %0, %1 offset:%2" : "=v"(t1[g]) : "v"(ta), "n"(32 * g));
                    asm volatile("ds_read_b128 %0, %1 offset:%2" : "=v"(t2[g]) : "v"(ta), "n"(128 + 32 * g));
                }
            }
            const f32x16 H = pool(IC<(P & 1)>{}, acce, hp[T], q0[T], q1[T]);
            asm volatile("s_waitcnt lgkmcnt(0)" : "+v"(tsc[0]), "+v"(tsc[1]), "+v"(tsc[2]), "+v"(tsc[3]), "+v"(tsh[0]), "+v"(tsh[1]), "+v"(tsh[2]), "+v"(tsh[3]));
            constexpr int off = ((P + 3) % F_NB) * F_ROWB;       // B row t-5
            auto& wb = wbB;                                      // (named here: implicit capture of an asm operand)
#pragma unroll
            for (int g = 0; g < 4; ++g) {
                const f32x4 sc = tsc[g], sh = tsh[g];
                const f32x2 y0 = pk_fma(f32x2{H[4 * g], H[4 * g + 1]}, f32x2{sc[0], sc[1]}, f32x2{sh[0], sh[1]});
                const f32x2 y1 = pk_fma(f32x2{H[4 * g + 2], H[4 * g + 3]}, f32x2{sc[2], sc[3]}, f32x2{sh[2], sh[3]});
                const i32x2 d = {static_cast<int>(pack2<DT>(y0[0], y0[1])), static_cast<int>(pack2<DT>(y1[0], y1[1]))};
                asm volatile("ds_write_b64 %0, %1 offset:%2" ::"v"(wb[T] ^ static_cast<unsigned>(g << 4)), "v"(d), "n"(off) : "memory");
            }
        };
#ifdef RN_STAMPS
        unsigned long long st_bar = 0, st_seg[6] = {0, 0, 0, 0, 0, 0};
#endif
        auto step = [&](auto PC, int t) __attribute__((always_inline)) {
            constexpr int P = decltype(PC)::value;
            if (t < nrows + 9) a_next += Win * 64;
            // A row t+1: one DMA piece behind every fourth MFMA of the first chain.  Skip rows of the partner consumer (see the
            // schedule above): decided behind the first MFMAs, fetched behind those of the second chain; when no new row is
            // due the newest one is fetched again into its own slot (same bytes), so the step has no branch around a DMA
            auto hook0 = [&](auto IC_) __attribute__((always_inline)) {
                constexpr int i = decltype(IC_)::value;
                if constexpr (i == 0) issue_A_pieces(IC<0>{}, IC<1>{}, a_next, (P + 1) % F_NA);
                if constexpr (i == 4) issue_A_pieces(IC<1>{}, IC<2>{}, a_next, (P + 1) % F_NA);
                if constexpr (i == 8) issue_A_pieces(IC<2>{}, IC<3>{}, a_next, (P + 1) % F_NA);
                if constexpr (i == 12) issue_A_pieces(IC<3>{}, IC<4>{}, a_next, (P + 1) % F_NA);
                if constexpr (i == 2) {
                    const int need_cur = min(ylo_nxt + 1, Win - 1);              // ylo_nxt: lo row of this step's output row
                    if (sk_f < need_cur) {                  // the consumer fetches this one itself (lo row jumped by 2)
                        ++sk_f;
                        sk_slot = sk_slot == F_NSK - 1 ? 0 : sk_slot + 1;
                    }
                }
                if constexpr (i == 6) {
                    ylo_nxt = ylo_step(t + 1);
                    const int need_next = min(ylo_nxt + 1, Win - 1);
                    if (sk_f < need_next) {
                        ++sk_f;
                        sk_slot = sk_slot == F_NSK - 1 ? 0 : sk_slot + 1;
                    }
                }
            };
            auto hook1 = [&](auto IC_) __attribute__((always_inline)) {
                constexpr int i = decltype(IC_)::value;
                if constexpr (i == 1) issue_skip_piece(IC<0>{}, sk_f, sk_slot);
                if constexpr (i == 5) issue_skip_piece(IC<1>{}, sk_f, sk_slot);
                if constexpr (i == 9) issue_skip_piece(IC<2>{}, sk_f, sk_slot);
                if constexpr (i == 13) issue_skip_piece(IC<3>{}, sk_f, sk_slot);
            };
            f32x16 acc;
#ifdef RN_STAMPS
            const unsigned long long tp0 = stamp23();
#endif
            chain(IC<(P + 2) % 4>{}, IC<F_ROWA>{}, baseA[0], w2, acc, hook0);     // conv row t-2: A rows t-2 .. t
#ifdef RN_STAMPS
            const unsigned long long tp1 = stamp23();
#endif
            epi(IC<0>{}, PC, acc);
#ifdef RN_STAMPS
            const unsigned long long tp2 = stamp23();
#endif
            chain(IC<(P + 2) % 4>{}, IC<F_ROWA>{}, baseA[1], w2, acc, hook1);
#ifdef RN_STAMPS
            const unsigned long long tp3 = stamp23();
#endif
            epi(IC<1>{}, PC, acc);
#ifdef RN_STAMPS
            const unsigned long long tp4 = stamp23();
            st_seg[0] += tp1 - tp0;
            st_seg[1] += tp2 - tp1;
            st_seg[2] += tp3 - tp2;
            st_seg[3] += tp4 - tp3;
#endif
#ifdef RN_STAMPS
            const unsigned long long tb0 = stamp23();
#endif
            wait_vmcnt<0>();
            asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
            raw_barrier();
#ifdef RN_STAMPS
            st_bar += stamp23() - tb0;
#endif
        };
#ifdef RN_STAMPS
        const unsigned long long st_loop0 = stamp23();
#endif
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
#ifdef RN_STAMPS
        if (a.stamp_buf && lane == 0) {
            const int64_t w = (static_cast<int64_t>(blockIdx.y) * gridDim.x + blockIdx.x) * 8 + wave;
            a.stamp_buf[w * 4 + 0] = stamp23() - st_loop0;
            a.stamp_buf[w * 4 + 1] = 0;
            a.stamp_buf[w * 4 + 2] = st_bar;
            a.stamp_buf[w * 4 + 3] = static_cast<unsigned long long>(nsteps);
            unsigned long long* sg = a.stamp_buf + static_cast<int64_t>(gridDim.x) * gridDim.y * 32 + w * 8;
            for (int k = 0; k < 6; ++k) sg[k] = st_seg[k];
        }
#endif
        return;
    }

    // =================================================================== consumer: second stage, B ring -> HBM
    __builtin_amdgcn_s_setprio(1);
    i32x4 w3[KC];                                         // (loaded like the producer's: see there)
#pragma unroll
    for (int kc = 0; kc < KC; ++kc) {
        const i32x4* src = a.wfrag3 + kc * 64 + lane;
        asm volatile("global_load_dwordx4 %0, %1, off" : RN_WREG_OUT(w3[kc]) : "v"(src) : "memory");
    }
    unsigned baseB[2][3][2], a_off[2][2];
    int voff[2];
    i32x4 bw[2][2];
#pragma unroll
    for (int T = 0; T < 2; ++T) {
        const int x_t = 29 * (2 * wq + T);
#pragma unroll
        for (int kx = 0; kx < 3; ++kx)
#pragma unroll
            for (int cc = 0; cc < 2; ++cc) {
                const int cb = min(x_t + r + kx, Wb - 1);
                baseB[T][kx][cc] = ringB_lds + static_cast<unsigned>(cb * 64 + (((cc * 2 + hh) ^ swz4(cb)) << 4));
            }
        const int xo = x_t + r;
        voff[T] = (r < 29 && xo < Wo) ? ((x0 + xo) * 32 + 8 * hh) * 2 : OOB;
        const int xo_t0 = x0 + min(x_t, Wo - 1);
        const int xs_t = a.rlo[xo_t0] - xs0;
        const int xq = x0 + min(xo, Wo - 1);
        const int plo = a.rlo[xq] - xs0, phi = a.rhi[xq] - xs0;
        const float xlq = res_quant_lerp<DT>(a.rlerp[xq]);
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
        for (int t2 = 0; t2 < 2; ++t2) {
            const int pix = xs_t + 8 * (grp >> 1) + 4 * t2 + q;
            const int ch = 2 * (grp & 1) + (pp >> 1);
            a_off[T][t2] = skw_lds + static_cast<unsigned>((pix * 4 + (ch ^ swz4(pix))) * 16 + (pp & 1) * 8);
        }
    }
    asm volatile("" : RN_WREG_IO(bw[0][0]), RN_WREG_IO(bw[0][1]), RN_WREG_IO(bw[1][0]), RN_WREG_IO(bw[1][1]));
    const int out_row_bytes = Hout * 64;
    const char* out_row = reinterpret_cast<const char*>(a.out + static_cast<int64_t>(n) * Hout * Hout * 32) + static_cast<int64_t>(yo0) * out_row_bytes;
    i32x4 hp[2][2], q0[2][2], q1[2][2];
#pragma unroll
    for (int T = 0; T < 2; ++T) hp[T][0] = hp[T][1] = q0[T][0] = q0[T][1] = q1[T][0] = q1[T][1] = i32x4{0, 0, 0, 0};
    struct RowCtx {
        float yl;
        unsigned sk_lo, sk_hi;
        __amdgpu_buffer_rsrc_t rs;
        int emit_mask;
    };
    VLerp vl_cur = vlerp_of(yo0);
    VLerp vl_pre = vl_cur;                                // interpolation of the NEXT step's output row, one step ahead
    int slot_cur = 0;
    RowCtx cx_cur{};
    cx_cur.emit_mask = OOB;
    cx_cur.rs = __builtin_amdgcn_make_buffer_rsrc(const_cast<char*>(out_row), 0, out_row_bytes, 0x00020000);
    f32x16 acc1;
#pragma unroll
    for (int g = 0; g < 16; ++g) acc1[g] = 0.f;
    wait_vmcnt<0>();                                      // the weight fragments have landed
#pragma unroll
    for (int kc = 0; kc < KC; ++kc) asm volatile("" : RN_WREG_IO(w3[kc]));
    lds_barrier();

    const unsigned tabl_lds = lds_addr(tab + 64 + 4 * hh);
    auto epi = [&](auto TC, auto PRC, const f32x16& acce, const RowCtx& cx) __attribute__((always_inline)) {
        constexpr int T = decltype(TC)::value;
        // residual: transposed reads of the staged skip pair, R_lo / R_hi = Skip^T * Wx on the matrix cores
        i32x2 tq[8];
        auto& ao = a_off;
#pragma unroll
        for (int t2 = 0; t2 < 2; ++t2) {
            const unsigned alo = ao[T][t2] + cx.sk_lo, ahi = ao[T][t2] + cx.sk_hi;
            asm volatile("ds_read_b64_tr_b16 %0, %1" : "=v"(tq[t2]) : "v"(alo));
            asm volatile("ds_read_b64_tr_b16 %0, %1 offset:1024" : "=v"(tq[2 + t2]) : "v"(alo));
            asm volatile("ds_read_b64_tr_b16 %0, %1" : "=v"(tq[4 + t2]) : "v"(ahi));
            asm volatile("ds_read_b64_tr_b16 %0, %1 offset:1024" : "=v"(tq[6 + t2]) : "v"(ahi));
        }
        f32x4 tsc1[4], tsh1[4], tsc2[4];
        auto tab_issue = [&](auto GC) __attribute__((always_inline)) {
            constexpr int g = decltype(GC)::value;
            auto& t1 = tsc1;
            auto& t2 = tsh1;
            auto& t3 = tsc2;
            const unsigned ta = tabl_lds;        // (named outside the asm: implicit capture)
            asm volatile("ds_read_b128 %0, %1 offset:%2" : "=v"(t1[g]) : "v"(ta), "n"(32 * g));
            asm volatile("ds_read_b128 %0, %1 offset:%2" : "=v"(t2[g]) : "v"(ta), "n"(128 + 32 * g));
            asm volatile("ds_read_b128 %0, %1 offset:%2" : "=v"(t3[g]) : "v"(ta), "n"(256 + 32 * g));
        };
        // Pooling and residual MFMAs in the order their operands become available, so that the matrix pipe works while the
        // VALU packs: the two pooling MFMAs that only take the history (q of two rows ago) go first, the four residual MFMAs
        // follow as soon as the transposed reads are back, the two pooling MFMAs of the new row close the sequence.  (Same
        // accumulation order of H as pool(): bit-identical.)
        constexpr int PR = decltype(PRC)::value;
        const f32x16 zero = {0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f};
        i32x4(&qold)[2] = PR == 0 ? q0[T] : q1[T];
        f32x16 H = mfma32<RN_DTYPE_F16>(qold[0], pmw[0], zero);
        H = mfma32<RN_DTYPE_F16>(qold[1], pmw[1], H);
        i32x4 qp[2];
        auto pack_half = [&](auto HC) __attribute__((always_inline)) {
            constexpr int hf = decltype(HC)::value;
            auto& hpT = hp[T];
#pragma unroll
            for (int i2 = 8 * hf; i2 < 8 * hf + 8; i2 += 2) {
                const int vp = static_cast<int>(pack2_relu6_sixth(acce[i2], acce[i2 + 1]));
                qp[hf][(i2 % 8) / 2] = pk_add_f16(hpT[hf][(i2 % 8) / 2], vp);
                hpT[hf][(i2 % 8) / 2] = vp;
            }
        };
        pack_half(IC<0>{});
        asm volatile("s_waitcnt lgkmcnt(0)" : "+v"(tq[0]), "+v"(tq[1]), "+v"(tq[2]), "+v"(tq[3]), "+v"(tq[4]), "+v"(tq[5]), "+v"(tq[6]), "+v"(tq[7]));
        const i32x4 al0 = {tq[0][0], tq[0][1], tq[1][0], tq[1][1]}, al1 = {tq[2][0], tq[2][1], tq[3][0], tq[3][1]};
        const i32x4 ah0 = {tq[4][0], tq[4][1], tq[5][0], tq[5][1]}, ah1 = {tq[6][0], tq[6][1], tq[7][0], tq[7][1]};
        f32x16 r_lo = mfma32<DT>(al0, bw[T][0], zero);
        f32x16 r_hi = mfma32<DT>(ah0, bw[T][0], zero);
        r_lo = mfma32<DT>(al1, bw[T][1], r_lo);
        r_hi = mfma32<DT>(ah1, bw[T][1], r_hi);
        pack_half(IC<1>{});
        H = mfma32<RN_DTYPE_F16>(qp[0], pmw[0], H);
        H = mfma32<RN_DTYPE_F16>(qp[1], pmw[1], H);
        qold[0] = qp[0];
        qold[1] = qp[1];
        uint2 pk[4];
        tab_issue(IC<0>{});
        // folded-BN table entries of channel group g: read by inline asm one group ahead of their use and retired by a
        // counted wait (compiler-visible LDS loads are waited for with lgkmcnt(0) right before use: four exposed LDS
        // round trips per epilogue on the wave the whole step waits for)
        [&]<int... G>(std::integer_sequence<int, G...>) {
            (([&] {
                 constexpr int g = G;
                 auto& t1 = tsc1;
                 auto& t2 = tsh1;
                 auto& t3 = tsc2;
                 if constexpr (g + 1 < 4) {
                     tab_issue(IC<(g + 1 < 4 ? g + 1 : 0)>{});
                     asm volatile("s_waitcnt lgkmcnt(3)" : "+v"(t1[g]), "+v"(t2[g]), "+v"(t3[g]));
                 } else {
                     asm volatile("s_waitcnt lgkmcnt(0)" : "+v"(t1[g]), "+v"(t2[g]), "+v"(t3[g]));
                 }
             }(),
             [&] {
            constexpr int g = G;
            const f32x4 sc1 = tsc1[g], sh1 = tsh1[g], sc2 = tsc2[g];
            f32x2 y[2];
            const f32x2 ylv = pk_splat(cx.yl);
#pragma unroll
            for (int jj = 0; jj < 4; jj += 2) {
                const f32x2 h2 = {H[4 * g + jj], H[4 * g + jj + 1]};
                const f32x2 lo = {r_lo[4 * g + jj], r_lo[4 * g + jj + 1]};
                const f32x2 hi = {r_hi[4 * g + jj], r_hi[4 * g + jj + 1]};
                const f32x2 y1 = pk_fma(h2, f32x2{sc1[jj], sc1[jj + 1]}, f32x2{sh1[jj], sh1[jj + 1]});
                const f32x2 rs = pk_fma(pk_sub(hi, lo), ylv, lo);
                y[jj / 2] = pk_fma(rs, f32x2{sc2[jj], sc2[jj + 1]}, y1);
            }
            pk[g].x = pack2<DT>(y[0][0], y[0][1]);
            pk[g].y = pack2<DT>(y[1][0], y[1][1]);
             }()),
             ...);
        }(std::make_integer_sequence<int, 4>{});
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
    };
#ifdef RN_STAMPS
    unsigned long long st_bar = 0, st_seg[6] = {0, 0, 0, 0, 0, 0};
#endif
    auto step = [&](auto PC, int t) __attribute__((always_inline)) {
        constexpr int P = decltype(PC)::value;
        constexpr int PR = P & 1;
        const int jo = t - F_LAG;
        cx_cur.yl = vl_cur.yl;
        cx_cur.sk_lo = static_cast<unsigned>(slot_cur * F_SKROW);
        cx_cur.sk_hi = static_cast<unsigned>((vl_cur.ylo + 1 > Win - 1 ? slot_cur : (slot_cur == F_NSK - 1 ? 0 : slot_cur + 1)) * F_SKROW);
        cx_cur.rs = __builtin_amdgcn_make_buffer_rsrc(const_cast<char*>(out_row), 0, out_row_bytes, 0x00020000);
        cx_cur.emit_mask = jo >= 0 ? 0 : OOB;
        if (jo >= 0 && jo < nrows - 1) out_row += out_row_bytes;
#ifdef RN_STAMPS
        const unsigned long long tc0 = stamp23();
#endif
        // skip rows: the rare second new row of a step whose lo row jumped by 2 is fetched here; the regular one for the
        // next step is fetched by the partner producer during this step (mirrored in sk_f / sk_slot)
        const VLerp vl_next = vl_pre;                  // (computed behind the previous step's second chain)
        {
            const int need_cur = min(vl_cur.ylo + 1, Win - 1);
            if (sk_f < need_cur) {
                ++sk_f;
                sk_slot = sk_slot == F_NSK - 1 ? 0 : sk_slot + 1;
                issue_skip_row(sk_f, sk_slot);
            }
            const int need_next = min(vl_next.ylo + 1, Win - 1);
            if (sk_f < need_next) {
                ++sk_f;
                sk_slot = sk_slot == F_NSK - 1 ? 0 : sk_slot + 1;
            }
        }
#ifdef RN_STAMPS
        const unsigned long long tc1 = stamp23();
        st_seg[0] += tc1 - tc0;
        const unsigned long long tc2 = tc1;
#endif
        f32x16 acc0;
        chain(IC<P>{}, IC<F_ROWB>{}, baseB[0], w3, acc0, no_hook);       // conv row t-8: B rows t-8 .. t-6
#ifdef RN_STAMPS
        const unsigned long long tc3 = stamp23();
        st_seg[2] += tc3 - tc2;
#endif
        wait_vmcnt<0>();                                       // a skip row fetched at the top of this step has landed
#ifdef RN_STAMPS
        const unsigned long long tc4 = stamp23();
        st_seg[3] += tc4 - tc3;
#endif
        epi(IC<0>{}, IC<PR>{}, acc0, cx_cur);
#ifdef RN_STAMPS
        const unsigned long long tc5 = stamp23();
        st_seg[4] += tc5 - tc4;
#endif
        // the next step's vertical interpolation (float multiply -> floor -> two readfirstlanes) rides behind this chain's MFMAs
        auto hook_c1 = [&](auto IC_) __attribute__((always_inline)) {
            if constexpr (decltype(IC_)::value == 2) vl_pre = vlerp_of(yo0 + min(max(jo + 2, 0), nrows - 1));
        };
        chain(IC<P>{}, IC<F_ROWB>{}, baseB[1], w3, acc1, hook_c1);
        epi(IC<1>{}, IC<PR>{}, acc1, cx_cur);
#ifdef RN_STAMPS
        st_seg[5] += stamp23() - tc5;
#endif
        {
            int sl = slot_cur + (vl_next.ylo - vl_cur.ylo);
            slot_cur = sl >= F_NSK ? sl - F_NSK : sl;
            vl_cur = vl_next;
        }
#ifdef RN_STAMPS
        const unsigned long long tb0 = stamp23();
#endif
        asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
        raw_barrier();
#ifdef RN_STAMPS
        st_bar += stamp23() - tb0;
#endif
    };
#ifdef RN_STAMPS
    const unsigned long long st_loop0 = stamp23();
#endif
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
#ifdef RN_STAMPS
    if (a.stamp_buf && lane == 0) {
        const int64_t w = (static_cast<int64_t>(blockIdx.y) * gridDim.x + blockIdx.x) * 8 + wave;
        a.stamp_buf[w * 4 + 0] = stamp23() - st_loop0;
        a.stamp_buf[w * 4 + 1] = 0;
        a.stamp_buf[w * 4 + 2] = st_bar;
        a.stamp_buf[w * 4 + 3] = static_cast<unsigned long long>(nsteps);
        unsigned long long* sg = a.stamp_buf + static_cast<int64_t>(gridDim.x) * gridDim.y * 32 + w * 8;
        for (int k = 0; k < 6; ++k) sg[k] = st_seg[k];
    }
#endif
}

}  // namespace

int rn_stage23_launch(int dtype, hipStream_t s, const Stage23Args& a, int n) {
    auto launch = [&](auto kern) -> int {
        static std::atomic<unsigned long long> attr_devices{0};     // per device and instantiation, see launch_rw
        int dev = 0;
        RN_HIP(hipGetDevice(&dev));
        if (!(attr_devices.load(std::memory_order_acquire) >> (dev & 63) & 1ull)) {
            RN_HIP(hipFuncSetAttribute(reinterpret_cast<const void*>(kern), hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024));
            attr_devices.fetch_or(1ull << (dev & 63), std::memory_order_release);
        }
        hipLaunchKernelGGL(kern, dim3(a.n_bands * a.n_cblocks, n), dim3(512), F_LDS, s, a);
        RN_CHECK_LAUNCH();
        return RN_OK;
    };
    if (dtype == RN_DTYPE_BF16) return launch(stage23pc_kernel<RN_DTYPE_BF16>);
    return launch(stage23pc_kernel<RN_DTYPE_F16>);
}
