// The tail of the 16-bit path in ONE launch: the last two conv stages of the depth-3 block network.py:230 builds
// (16->16, pool 4/2 each; the second one adds the legacy-bilinear resize of the block's first BN output and a second
// BN, network.py:198-203), the flatten (:231-233), the four dense blocks (:210-223, :234-237), softmax and argmax
// (:44-45).  At 224 input these are 21x21x16 -> 8x8x16 -> 2x2x16 -> 64 -> 32 -> 16 -> 8 -> 6 per image: three
// launches of pure latency (0.044 ms of a 1.7 ms step) before this kernel.
//
// One 256-thread workgroup per image.  The stage input (and residual skip tensor) is staged in LDS once; each wave
// streams the conv rows of its share of pooled rows through the matrix cores exactly like the generic stage kernel
// (stage_mfma_kernel, rn_fused.hip: same MFMA sequence, same DPP / register-ring pooling order, same un-folded BN
// expression), so the results are those of the three-launch path; the stage outputs are still written to HBM (a few KB:
// rn_tap keeps working) and kept in LDS for the next phase; wave 0 finishes with the dense chain of head_kernel
// (rn_kernels_f32.hip) on dense kernels that were staged in LDS while the conv phases ran.
#include "rn_tail_body.h"

#include <atomic>

using namespace rnk;

namespace {

template <int DT>
__global__ __launch_bounds__(256) void tail_kernel(const TailArgs a) {
    __shared__ __attribute__((aligned(16))) unsigned short xin[T_MAXSIDE * T_MAXSIDE * 16];   // first step's output
    __shared__ __attribute__((aligned(16))) unsigned short xa[16 * 16 * 16];                  // second step's output
    __shared__ __attribute__((aligned(16))) unsigned short xb[8 * 8 * 16];                    // third step's output = flatten
    __shared__ float buf0[8 * 8 * 16];
    __shared__ float small[2][64];
    __shared__ float wl[T_W_LDS];
    __shared__ __attribute__((aligned(16))) float ttab[144];
    const int tid = threadIdx.x;
    const int img = blockIdx.x;
    const int S = a.a.Si;
    // ---- stage the block's first output (input and skip tensor) and the dense kernels
    {
        const i32x4* src = reinterpret_cast<const i32x4*>(a.in + static_cast<int64_t>(img) * S * S * 16);
        const int nchunk = S * S * 2;
        for (int i = tid; i < nchunk; i += 256) reinterpret_cast<i32x4*>(xin)[i] = src[i];
    }
    int w_off[RN_MAX_DENSE];
    tail_stage_dense_dma(a.head, wl, w_off, __builtin_amdgcn_readfirstlane(tid >> 6), tid & 63, 4);      // lands while the conv phases run; tail_phase retires it
    tail_phase<DT>(a, xin, xa, xb, buf0, small, wl, w_off, ttab, img, tid);
}

}  // namespace

// The tail kernel covers: two 16->16 stages with pool 4/2 (the second with the residual) whose inputs fit its LDS images,
// a flatten that fits buf0 and dense layers no wider than 64.
bool rn_tail_supported(const rn_handle* h) {
    const size_t n = h->stages.size();
    if (n < 3) return false;
    const StagePlan& s0 = h->stages[n - 3];
    const StagePlan& sa = h->stages[n - 2];
    const StagePlan& sb = h->stages[n - 1];
    auto ok = [](const StagePlan& s) { return s.cin == 16 && s.cout == 16 && s.pool_k == 4 && s.pool_s == 2; };
    if (!ok(sa) || !ok(sb) || s0.cout != 16 || s0.node_bn2 >= 0) return false;
    if (sa.skip_stage >= 0 || sb.skip_stage != static_cast<int>(n) - 3) return false;
    if (sa.in_side > T_MAXSIDE || sa.out_side > 16 || sb.out_side > 8 || sb.out_side < 1) return false;
    if (h->dense.empty() || h->dense[0].nin != sb.out_side * sb.out_side * 16 || h->dense[0].nin > 8 * 8 * 16) return false;
    for (size_t d = 0; d < h->dense.size(); ++d)
        if (h->dense[d].nout > 64 || (d > 0 && h->dense[d].nin > 64)) return false;
    return true;
}

void rn_tail_fill_args(rn_handle* h, const i32x4* wfrag_a, const i32x4* wfrag_b, const HeadArgs& head, float* d_probs, int64_t* d_ids,
                       TailArgs* out) {
    const size_t ns = h->stages.size();
    const StagePlan& s0 = h->stages[ns - 3];
    const StagePlan& sa = h->stages[ns - 2];
    const StagePlan& sb = h->stages[ns - 1];
    TailArgs a{};
    a.in = static_cast<const unsigned short*>(h->nodes[s0.node_bn].ptr);
    a.a = TailStage{wfrag_a, sa.bn.mean, sa.bn.inv, sa.bn.beta, static_cast<unsigned short*>(h->nodes[sa.node_bn].ptr), sa.in_side,
                    sa.out_side};
    a.b = TailStage{wfrag_b, sb.bn.mean, sb.bn.inv, sb.bn.beta, static_cast<unsigned short*>(h->nodes[sb.node_bn2].ptr), sb.in_side,
                    sb.out_side};
    a.mean2 = sb.bn2.mean;
    a.inv2 = sb.bn2.inv;
    a.beta2 = sb.bn2.beta;
    a.rlo = sb.rt.lo;
    a.rhi = sb.rt.hi;
    a.rlerp = sb.rt.lerp;
    a.head = head;
    a.probs = d_probs;
    a.ids = d_ids;
    *out = a;
}

int rn_tail_launch(rn_handle* h, const i32x4* wfrag_a, const i32x4* wfrag_b, const HeadArgs& head, int n, float* d_probs,
                   int64_t* d_ids) {
    TailArgs a;
    rn_tail_fill_args(h, wfrag_a, wfrag_b, head, d_probs, d_ids, &a);
    if (h->dtype == RN_DTYPE_BF16)
        hipLaunchKernelGGL(tail_kernel<RN_DTYPE_BF16>, dim3(n), dim3(256), 0, h->stream, a);
    else
        hipLaunchKernelGGL(tail_kernel<RN_DTYPE_F16>, dim3(n), dim3(256), 0, h->stream, a);
    RN_CHECK_LAUNCH();
    return RN_OK;
}
