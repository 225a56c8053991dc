// Fused 16-bit (bf16 / fp16 storage, fp32 accumulate) MFMA execution path.
#pragma once
#include "rn_internal.h"

// Pack weights into MFMA fragment order, choose per-stage launch geometry.
int rn_fused_prepare(rn_handle* h, const rn_weights* w);
// Enqueue the fused forward pass.  Exactly one of d_bgr / d_rgb is non-null.
int rn_fused_forward(rn_handle* h, const uint8_t* d_bgr, const float* d_rgb, int n, float* d_probs,
                     int64_t* d_ids);
void rn_fused_release(rn_handle* h);
// head launcher shared with the unfused path (defined in rn_api.hip)
int rn_run_head(rn_handle* h, int n, float* d_probs, int64_t* d_ids);
void rn_record_event(rn_handle* h, int idx);
