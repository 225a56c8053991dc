// Fused 16-bit (bf16 / fp16 storage, fp32 accumulate) MFMA execution path.
#pragma once
#include <vector>
#include "rn_internal.h"
#include "rn_stage.h"

// Pack weights into MFMA fragment order, choose per-stage launch geometry.
int rn_fused_prepare(rn_handle* h, const rn_weights* w);
// Enqueue the fused forward pass.  Exactly one of d_bgr / d_rgb is non-null.
int rn_fused_forward(rn_handle* h, const uint8_t* d_bgr, const float* d_rgb, int n, float* d_probs,
                     int64_t* d_ids);
void rn_fused_release(rn_handle* h);
// the stage under which the launch that computes `stage` reports its time (== stage for a launch of its own)
int rn_fused_launch_rep(const rn_handle* h, int stage);
// the stage's output tensor is never written to HBM on this handle (it only exists in LDS inside a fused launch)
bool rn_fused_stage_elided(const rn_handle* h, int stage);
// channel relabelling of a node's stored tensor on this handle (position p holds the reference's channel perm[p]), or null
const int* rn_fused_node_perm(const rn_handle* h, int node_id);
void rn_fused_frozen_info(const rn_handle* h, int info[4]);
// constant channels nobody computes (round 6): info = {stage whose last cout quarter is constant in the handle's 16-bit store or -1,
// channels of it proven constant, channels folded (16), input channels the stage behind it still contracts (48)}
void rn_fused_const_info(const rn_handle* h, int info[4]);
// after the activation buffers exist: fill the constant channels once
int rn_fused_post_alloc(rn_handle* h);
// head launcher shared with the unfused path (defined in rn_api.hip)
int rn_run_head(rn_handle* h, int n, float* d_probs, int64_t* d_ids);
void rn_record_event(rn_handle* h, int idx);
void rn_fill_head_args(rn_handle* h, HeadArgs* a);

// ---- the last two conv stages + flatten + dense head + softmax/argmax in one launch (rn_tail.hip)
bool rn_tail_supported(const rn_handle* h);
int rn_tail_launch(rn_handle* h, const rnk::i32x4* wfrag_a, const rnk::i32x4* wfrag_b, const HeadArgs& head, int n, float* d_probs,
                   int64_t* d_ids);

// ---- register-weights stage kernels (rn_stage_rw.hip)
struct RwPlan {
    int variant = -1;
    int npt = 0;
    int n_colblocks = 0;
    int skipcols = 0;
    size_t lds_bytes = 0;
    bool wide = false;       // stride-2 pooling on wide tiles (15 windows per tile) instead of gapped ones (14)
    int wgs_per_cu = 1;      // workgroups of this variant resident on one CU (register / LDS budget of the instantiation)
};
bool rn_rw_supported(int cin, int cout, int pool_k, int pool_s, bool res, int out_side, int skip_side,
                     RwPlan* plan);
int rn_rw_launch(const RwPlan& p, int dtype, hipStream_t s, const rnk::StageArgs& a, dim3 grid);
int rn_rw_s0sh_colblocks(int out_side);      // column blocks (227 wide) of the shared-ring stage 0 + 1 kernel

// ---- cross-stage fused pair: conv-pool-BN -> conv-pool-BN + residual of a depth-3 block (rn_stage23.hip)
bool rn_stage23_supported(int in_side);
// rn_conv16.hip: the un-pooled 64 -> 128 stage on 16x16x32 matrix tiles
bool rn_conv16_supported(int cin, int cout, int pool_k, bool res);
int rn_conv16_colblocks(int out_side);
void rn_conv16_pack(const float* w_hwio, int dtype, unsigned short (*cvt_bf16)(float), unsigned short (*cvt_f16)(float),
                    std::vector<unsigned short>* out);
int rn_conv16_launch(int dtype, hipStream_t s, const rnk::Conv16Args& a, int n);
// ... and the 128 -> 16 stage with avg-pool 4/2 (one wave = one 16-pixel tile x all 16 couts, no K split)
bool rn_conv16p_supported(int cin, int cout, int pool_k, int pool_s, bool res);
int rn_conv16p_colblocks(int out_side);
int rn_conv16p_wgs_per_cu(int out_side);
void rn_conv16p_pack(const float* w_hwio, int dtype, unsigned short (*cvt_bf16)(float), unsigned short (*cvt_f16)(float),
                     std::vector<unsigned short>* out);
int rn_conv16p_launch(int dtype, hipStream_t s, const rnk::Conv16Args& a, int n);
bool rn_stage23_plan(int in_side, int* n_cblocks, int* x0, int* wo);   // column blocks (x0, wo: 4 entries)
int rn_stage23_launch(int dtype, hipStream_t s, const rnk::Stage23Args& a, int n);
// the un-pooled 64 -> 128 stage with row-register blocking (rn_stage6x.hip)
bool rn_stage6x_supported(int cin, int cout, int pool_k, bool res, int in_side);
bool rn_stage6x_plan(int out_side, int* n_cb, int* xo0, int* wo);       // column blocks (xo0, wo: 4 entries)
void rn_stage6x_pack(const float* w_hwio, int dtype, unsigned short (*cvt_bf16)(float), unsigned short (*cvt_f16)(float),
                     std::vector<unsigned short>* out);
int rn_stage6x_launch(int dtype, hipStream_t s, const rnk::StageArgs& a, int n);
// ... without its input channels 48..63 (constants of the handle): StageArgs::cstart carries their sum
void rn_stage6x_pack48(const float* w_hwio, int dtype, unsigned short (*cvt_bf16)(float), unsigned short (*cvt_f16)(float),
                       std::vector<unsigned short>* out);
// the 32 -> 64 stage with pool 4/2 on 16x16x32 tiles with row-register blocking (rn_stage4x.hip)
bool rn_stage4x_supported(int cin, int cout, int pool_k, int pool_s, bool res, int in_side);
bool rn_stage4x_plan(int out_side, int* n_cb, int* xo0, int* wo);
void rn_stage4x_pack(const float* w_hwio, int dtype, unsigned short (*cvt_bf16)(float), unsigned short (*cvt_f16)(float),
                     std::vector<unsigned short>* out);
int rn_stage4x_launch(int dtype, hipStream_t s, const rnk::StageArgs& a, int n);
// the 64 -> 64 residual stage with pool 4/2 on 16x16x32 tiles with row-register blocking (rn_stage5x.hip)
bool rn_stage5x_supported(int cin, int cout, int pool_k, int pool_s, bool res, int in_side, int skip_side);
bool rn_stage5x_plan(int out_side, int* n_cb, int* xo0, int* wo);
void rn_stage5x_pack(const float* w_hwio, int dtype, unsigned short (*cvt_bf16)(float), unsigned short (*cvt_f16)(float),
                     std::vector<unsigned short>* out);
int rn_stage5x_launch(int dtype, hipStream_t s, const rnk::StageArgs& a, int n);
// ... without its input channels 48..63 (constants on the handle): 15 fragments per cout quarter (StageArgs::cstart carries their sum)
void rn_stage5x_pack48(const float* w_hwio, int dtype, unsigned short (*cvt_bf16)(float), unsigned short (*cvt_f16)(float),
                       std::vector<unsigned short>* out);
// the same pair on 16x16x32 tiles (rn_stage23x.hip): own weight fragment order, same launch arguments and column blocks
void rn_stage23x_pack(const float* w_hwio, int dtype, unsigned short (*cvt_bf16)(float), unsigned short (*cvt_f16)(float),
                      std::vector<unsigned short>* out);
int rn_stage23x_launch(int dtype, hipStream_t s, const rnk::Stage23Args& a, int n);
void rn_stage23x_pack_narrow(const float* w_hwio, const int* ring_cin, int dtype, unsigned short (*cvt_bf16)(float), unsigned short (*cvt_f16)(float),
                             std::vector<unsigned short>* out);
void rn_stage23x_pack_narrow8(const float* w_hwio, const int* ring_cin, int dtype, unsigned short (*cvt_bf16)(float), unsigned short (*cvt_f16)(float),
                              std::vector<unsigned short>* out);

// ---- float32 conv stages on the matrix cores (rn_stage_f32m.hip): the throughput path of RN_DTYPE_F32 handles without RN_FLAG_TAPS
int rn_f32m_prepare(rn_handle* h, const rn_weights* w);
void rn_f32m_release(rn_handle* h);
bool rn_f32m_covers(const rn_handle* h, int stage);
int rn_f32m_launch(rn_handle* h, int stage, const float* in, int n);      // writes the stage's output node

// ---- the back end in one launch: stage 6 -> 7 -> 8 -> 9 -> head (rn_backend.hip)
bool rn_backend_supported(const rn_handle* h);
int rn_backend_launch(rn_handle* h, const rnk::i32x4* wfrag6, const float* ptab6, const float* cstart6, const rnk::i32x4* wfrag7,
                      const float* ptab7, const rnk::i32x4* wfrag_a, const rnk::i32x4* wfrag_b, const HeadArgs& head, int n, float* d_probs,
                      int64_t* d_ids);
