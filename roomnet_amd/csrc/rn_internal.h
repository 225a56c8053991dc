// Internal declarations shared by the translation units of libroomnet_hip.so.
#pragma once
#include <hip/hip_runtime.h>
#include <cstdint>
#include <cstdio>
#include <cstdarg>
#include <map>
#include <string>
#include <vector>

#include "roomnet_hip.h"

#define RN_VERSION_STRING "roomnet_hip 0.1 (gfx950)"

void rn_set_error(const char* fmt, ...);

// (a failing runtime call also leaves its status in the thread's sticky "last error", which RN_CHECK_LAUNCH reads: a failure
//  that was reported here must not come back as the "launch failure" of the next, innocent kernel -- consume it)
#define RN_HIP(expr)                                                                          \
    do {                                                                                      \
        hipError_t _e = (expr);                                                               \
        if (_e != hipSuccess) {                                                               \
            rn_set_error("%s failed: %s (%s:%d)", #expr, hipGetErrorString(_e), __FILE__,     \
                         __LINE__);                                                           \
            (void)hipGetLastError();                                                          \
            return RN_E_HIP;                                                                  \
        }                                                                                     \
    } while (0)

#define RN_CHECK_LAUNCH()                                                                     \
    do {                                                                                      \
        hipError_t _e = hipGetLastError();                                                    \
        if (_e != hipSuccess) {                                                               \
            rn_set_error("kernel launch failed: %s (%s:%d)", hipGetErrorString(_e), __FILE__, \
                         __LINE__);                                                           \
            return RN_E_HIP;                                                                  \
        }                                                                                     \
    } while (0)

// per-channel affine form of an inference BN: y = (x - mean) * inv + beta
struct BnDev {
    float* mean = nullptr;   // [c]
    float* inv = nullptr;    // [c]  rsqrt(var + eps) * gamma, computed on the host in fp32
    float* beta = nullptr;   // [c]
};

struct ResizeTab {           // legacy TF-1.13 bilinear tables (host-computed, fp32)
    int32_t* lo = nullptr;   // [out]
    int32_t* hi = nullptr;   // [out]
    float* lerp = nullptr;   // [out]
};

struct StagePlan {
    int cin, cout, in_side, conv_side, pool_k, pool_s, out_side;
    int skip_stage, skip_side;
    // device weights
    float* w_f32 = nullptr;      // HWIO fp32 (unfused path and stage 0)
    void* w_frag = nullptr;      // MFMA fragment-packed 16-bit weights (fused path)
    int kchunks = 0;             // number of 16-deep K chunks in w_frag
    BnDev bn, bn2;
    ResizeTab rt;                // skip_side -> out_side
    // node ids (-1 when absent)
    int node_conv = -1, node_pool = -1, node_bn = -1, node_add = -1, node_bn2 = -1;
};

struct DensePlan {
    int nin, nout;
    float* w = nullptr;          // [nin, nout]
    float* bias = nullptr;       // [nout] or null
    float* inv = nullptr;        // BN: x*inv + shift   (null: no BN)
    float* shift = nullptr;
    int node_mm = -1, node_relu = -1, node_bn = -1;
};

struct NodeBuf {
    rn_node_info info;
    void* ptr = nullptr;         // device buffer [max_batch, h, w, c] (null: not materialised)
    int dtype = RN_DTYPE_F32;    // storage type of ptr
};

// one image of a batched crop + resize (rn_imageops.hip): crop window at `src`, scales and mode as cv::resize computes them
struct rn_resize_item {
    const uint8_t* src;
    int src_h, src_w;
    int64_t src_row_bytes;
    double scale_x, scale_y;
    int mode;
};

struct rn_handle {
    int device = 0;
    int dtype = RN_DTYPE_F32;
    unsigned flags = 0;
    int max_batch = 0;
    int im_side = 0, num_classes = 0;
    float bn_eps = 1e-3f;
    int n_cu = 0;                // compute units of the device (launch geometry is sized in rounds of the chip)
    hipStream_t own_stream = nullptr;
    hipStream_t stream = nullptr;
    std::vector<StagePlan> stages;
    std::vector<DensePlan> dense;
    std::vector<NodeBuf> nodes;
    int node_input = -1, node_flat = -1, node_softmax = -1;
    float* lut = nullptr;        // 256-entry uint8 -> float32 table
    // staging for host-buffer calls
    uint8_t* d_in_u8 = nullptr;
    uint8_t* d_raw = nullptr;      // staging for raw (un-resized) images, rn_classify_images_u8
    rn_resize_item* d_items = nullptr;           // [max_batch] crop windows of a batched resize (device) ...
    std::vector<rn_resize_item> items_host;      // ... and their host copy (kept until the next call: the upload is asynchronous)
    size_t raw_cap = 0;
    float* d_probs = nullptr;
    int64_t* d_ids = nullptr;
    // two-slot host pipeline (rn_submit_u8 / rn_collect): allocated by the first submit
    struct HostSlot {
        uint8_t* d_in = nullptr;
        float* d_probs = nullptr;
        int64_t* d_ids = nullptr;
        float* h_probs = nullptr;    // pinned
        int64_t* h_ids = nullptr;    // pinned
        hipEvent_t uploaded = nullptr, done = nullptr;
        int n = 0;
        bool busy = false;
    };
    HostSlot slots[2];
    hipStream_t copy_stream = nullptr;
    std::vector<void*> allocs;   // everything to hipFree on destroy
    void* fused = nullptr;       // plan of the fused 16-bit path (rn_fused.hip)
    void* f32m = nullptr;        // plan of the float32 matrix-core stage kernels (rn_stage_f32m.hip)
    // float32 handles: frozen first-BN channels of the 64 -> 64 residual stage folded (rn_create): the stage's index (or -1), the
    // couts whose convolution still runs, and the relabelling of the tensors it touches (node id -> position p holds channel perm[p])
    int f32_fold_stage = -1;
    int f32_fold_live = 0;
    // ... and frozen INPUT channels of a stage (its producer's BN freezes them; relabelled to the end): the stage's index (or -1),
    // the input channels it still contracts (a multiple of 8), how many channels were proven constant
    int f32_kfold_stage = -1;
    int f32_kfold_live = 0;
    int f32_kfold_proven = 0;
    std::map<int, std::vector<int>> node_perm;
    // profiling
    bool profiling = false;
    std::vector<hipEvent_t> events;   // [0]=start, [1]=after preprocess, [2+i]=after stage i, last=after head
    bool timing_valid = false;
    int last_n = 0;
};

// ---- launchers (rn_kernels_f32.hip) -------------------------------------------------
int rn_launch_preprocess_u8(hipStream_t s, const uint8_t* bgr, float* rgb, const float* lut, int64_t npix);
int rn_launch_conv3x3_relu6_f32(hipStream_t s, const float* in, const float* w, float* out, int n, int h,
                                int wd, int cin, int cout);
// stage 0 (3 -> 8, pool 3/1) as one launch (float32 handles without taps): same bits as the four per-node launches
int rn_launch_stage0_fused_f32(hipStream_t s, const float* in, const float* w, float* out, int n, int side, const BnDev& bn);
int rn_launch_avgpool_f32(hipStream_t s, const float* in, float* out, int n, int h, int w, int c, int k, int st);
int rn_launch_bn_f32(hipStream_t s, const float* in, float* out, int64_t npix, int c, const BnDev& bn);
int rn_launch_resize_add_f32(hipStream_t s, const float* x, const float* skip, float* out, int n, int side,
                             int skip_side, int c, const ResizeTab& rt);
// head: flatten + dense chain + softmax + argmax.  tap pointers may be null.
struct HeadArgs {
    int n_dense;
    int nin[RN_MAX_DENSE], nout[RN_MAX_DENSE];
    const float* w[RN_MAX_DENSE];
    const float* bias[RN_MAX_DENSE];
    const float* inv[RN_MAX_DENSE];
    const float* shift[RN_MAX_DENSE];
    float* tap_mm[RN_MAX_DENSE];
    float* tap_relu[RN_MAX_DENSE];
    float* tap_bn[RN_MAX_DENSE];
};
void rn_resize_item_fill(rn_resize_item* it, const uint8_t* d_src, int src_h, int src_w, int64_t src_row_bytes, int S);
int rn_launch_resize_batch_u8(hipStream_t s, const rn_resize_item* d_items, int n, uint8_t* d_dst_base, int S);
int rn_launch_resize_u8(hipStream_t s, const uint8_t* d_src, int src_h, int src_w, int64_t src_row_bytes, uint8_t* d_dst,
                        int dst_h, int dst_w);
int rn_launch_head(hipStream_t s, const void* flat, int flat_dtype, int n, const HeadArgs& a, float* probs,
                   int64_t* ids);
int rn_launch_convert_to_f32(hipStream_t s, const void* in, int dtype, float* out, int64_t n);
