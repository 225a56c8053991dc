// Host side of libroomnet_hip.so: execution plan, weight upload, forward orchestration,
// taps and timing.  See include/roomnet_hip.h for the contract of every entry point and
// the reference interface (file:line) it replaces.
#include "rn_internal.h"
#include "rn_fused.h"

#include <algorithm>
#include <cmath>
#include <cstring>
#include <new>

// ------------------------------------------------------------------------- errors
static thread_local char g_err[512] = "";

void rn_set_error(const char* fmt, ...) {
    va_list ap;
    va_start(ap, fmt);
    vsnprintf(g_err, sizeof(g_err), fmt, ap);
    va_end(ap);
}

extern "C" const char* rn_last_error(void) { return g_err; }
extern "C" const char* rn_version(void) { return RN_VERSION_STRING; }

extern "C" int rn_device_count(void) {
    int n = 0;
    if (hipGetDeviceCount(&n) != hipSuccess) return 0;
    return n;
}

// ------------------------------------------------------------------------ helpers
namespace {

struct DeviceGuard {
    int prev = -1;
    bool ok = false;
    explicit DeviceGuard(int dev) {
        if (hipGetDevice(&prev) != hipSuccess) prev = -1;
        ok = hipSetDevice(dev) == hipSuccess;
    }
    ~DeviceGuard() {
        if (prev >= 0) (void)hipSetDevice(prev);
    }
};

int dev_alloc(rn_handle* h, size_t bytes, void** out) {
    void* p = nullptr;
    if (bytes == 0) bytes = 16;
    hipError_t e = hipMalloc(&p, bytes);
    if (e != hipSuccess) {
        rn_set_error("hipMalloc(%zu bytes) failed: %s", bytes, hipGetErrorString(e));
        return RN_E_NOMEM;
    }
    h->allocs.push_back(p);
    *out = p;
    return RN_OK;
}

template <typename T>
int upload(rn_handle* h, const T* src, size_t count, T** out) {
    void* p = nullptr;
    int rc = dev_alloc(h, count * sizeof(T), &p);
    if (rc != RN_OK) return rc;
    RN_HIP(hipMemcpy(p, src, count * sizeof(T), hipMemcpyHostToDevice));
    *out = static_cast<T*>(p);
    return RN_OK;
}

int upload_bn(rn_handle* h, int c, const float* gamma, const float* beta, const float* mean, const float* var,
              float eps, BnDev* out) {
    std::vector<float> inv(c);
    for (int i = 0; i < c; ++i) inv[i] = (1.0f / sqrtf(var[i] + eps)) * gamma[i];
    int rc;
    if ((rc = upload(h, mean, c, &out->mean)) != RN_OK) return rc;
    if ((rc = upload(h, inv.data(), c, &out->inv)) != RN_OK) return rc;
    if ((rc = upload(h, beta, c, &out->beta)) != RN_OK) return rc;
    return RN_OK;
}

// TF-1.13 compute_interpolation_weights, align_corners=False, no half-pixel centres
int upload_resize_tab(rn_handle* h, int in_size, int out_size, ResizeTab* rt) {
    std::vector<int32_t> lo(out_size), hi(out_size);
    std::vector<float> lerp(out_size);
    const float scale = static_cast<float>(in_size) / static_cast<float>(out_size);
    for (int i = out_size - 1; i >= 0; --i) {
        const float src = static_cast<float>(i) * scale;
        lo[i] = static_cast<int32_t>(src);
        hi[i] = lo[i] + 1 < in_size - 1 ? lo[i] + 1 : in_size - 1;
        lerp[i] = src - static_cast<float>(lo[i]);
    }
    int rc;
    if ((rc = upload(h, lo.data(), out_size, &rt->lo)) != RN_OK) return rc;
    if ((rc = upload(h, hi.data(), out_size, &rt->hi)) != RN_OK) return rc;
    if ((rc = upload(h, lerp.data(), out_size, &rt->lerp)) != RN_OK) return rc;
    return RN_OK;
}

int add_node(rn_handle* h, const char* name, int hh, int ww, int cc) {
    NodeBuf nb;
    std::memset(&nb.info, 0, sizeof(nb.info));
    std::snprintf(nb.info.name, RN_NAME_LEN, "%s", name);
    nb.info.h = hh;
    nb.info.w = ww;
    nb.info.c = cc;
    h->nodes.push_back(nb);
    return static_cast<int>(h->nodes.size()) - 1;
}

size_t node_elems(const NodeBuf& nb) { return static_cast<size_t>(nb.info.h) * nb.info.w * nb.info.c; }

size_t dtype_size(int dtype) { return dtype == RN_DTYPE_F32 ? 4 : 2; }

bool fused_mode(const rn_handle* h) { return h->dtype != RN_DTYPE_F32; }

int validate(const rn_weights* w, int dtype, int max_batch, unsigned flags) {
    if (!w || !w->stages || !w->dense) {
        rn_set_error("rn_create: null weights");
        return RN_E_INVALID;
    }
    if (w->n_stages < 1 || w->n_stages > RN_MAX_STAGES || w->n_dense < 1 || w->n_dense > RN_MAX_DENSE) {
        rn_set_error("rn_create: unsupported stage/dense count (%d/%d)", w->n_stages, w->n_dense);
        return RN_E_INVALID;
    }
    if (w->im_side < 8 || w->num_classes < 1 || w->num_classes > 64) {
        rn_set_error("rn_create: bad im_side %d / num_classes %d", w->im_side, w->num_classes);
        return RN_E_INVALID;
    }
    if (dtype != RN_DTYPE_F32 && dtype != RN_DTYPE_BF16 && dtype != RN_DTYPE_F16) {
        rn_set_error("rn_create: unknown dtype %d", dtype);
        return RN_E_INVALID;
    }
    if (flags & ~(RN_FLAG_TAPS | RN_FLAG_STAGE_LAUNCHES | RN_FLAG_GENERIC_KERNELS | RN_FLAG_PAIR_32X32 | RN_FLAG_COMPUTE_FROZEN | RN_FLAG_NO_DITHER)) {
        rn_set_error("rn_create: unknown flag bits 0x%x", flags);
        return RN_E_INVALID;
    }
#ifndef RN_ROUND2_ARMS
    if (flags & RN_FLAG_PAIR_32X32) {
        rn_set_error("rn_create: RN_FLAG_PAIR_32X32 selects the round-2 comparison kernels, which are built into libroomnet_hip_ab.so "
                     "(tests and A/B runs; roomnet_amd/csrc/build.sh) and not into this library");
        return RN_E_INVALID;
    }
#endif
    if ((flags & RN_FLAG_TAPS) && dtype != RN_DTYPE_F32) {
        rn_set_error("rn_create: RN_FLAG_TAPS needs RN_DTYPE_F32 (the unfused per-node path)");
        return RN_E_INVALID;
    }
    if (max_batch < 1) {
        rn_set_error("rn_create: max_batch must be >= 1");
        return RN_E_INVALID;
    }
    if (w->stages[0].cin != 3) {
        rn_set_error("rn_create: first stage must take 3 input channels");
        return RN_E_INVALID;
    }
    return RN_OK;
}

int build_plan(rn_handle* h, const rn_weights* w) {
    int side = w->im_side, ch = 3, rc;
    h->node_input = add_node(h, "input", side, side, 3);
    char name[RN_NAME_LEN];
    for (int i = 0; i < w->n_stages; ++i) {
        const rn_conv_stage& s = w->stages[i];
        if (s.cin != ch || s.cout < 1 || s.cout > 512 || !s.kernel || !s.gamma || !s.beta || !s.mean ||
            !s.variance) {
            rn_set_error("stage %d: inconsistent description (cin %d, expected %d)", i, s.cin, ch);
            return RN_E_INVALID;
        }
        StagePlan p{};
        p.cin = s.cin;
        p.cout = s.cout;
        p.in_side = side;
        p.conv_side = side - 2;
        p.pool_k = s.pool_k;
        p.pool_s = s.pool_k ? s.pool_s : 1;
        if (p.conv_side < 1 || (p.pool_k && (p.conv_side < p.pool_k || p.pool_s < 1))) {
            rn_set_error("stage %d: im_side too small for this graph", i);
            return RN_E_INVALID;
        }
        p.out_side = p.pool_k ? (p.conv_side - p.pool_k) / p.pool_s + 1 : p.conv_side;
        p.skip_stage = s.skip_stage;
        if (s.skip_stage >= 0) {
            if (s.skip_stage >= i || h->stages[s.skip_stage].cout != s.cout || !s.gamma2 || !s.beta2 ||
                !s.mean2 || !s.variance2) {
                rn_set_error("stage %d: bad residual description", i);
                return RN_E_INVALID;
            }
            p.skip_side = h->stages[s.skip_stage].out_side;
        }
        if ((rc = upload(h, s.kernel, static_cast<size_t>(9) * s.cin * s.cout, &p.w_f32)) != RN_OK) return rc;
        if ((rc = upload_bn(h, s.cout, s.gamma, s.beta, s.mean, s.variance, w->bn_epsilon, &p.bn)) != RN_OK)
            return rc;
        if (s.skip_stage >= 0) {
            if ((rc = upload_bn(h, s.cout, s.gamma2, s.beta2, s.mean2, s.variance2, w->bn_epsilon, &p.bn2)) !=
                RN_OK)
                return rc;
            if ((rc = upload_resize_tab(h, p.skip_side, p.out_side, &p.rt)) != RN_OK) return rc;
        }
        std::snprintf(name, sizeof(name), "s%d.conv", i);
        p.node_conv = add_node(h, name, p.conv_side, p.conv_side, p.cout);
        if (p.pool_k) {
            std::snprintf(name, sizeof(name), "s%d.pool", i);
            p.node_pool = add_node(h, name, p.out_side, p.out_side, p.cout);
        }
        std::snprintf(name, sizeof(name), "s%d.bn", i);
        p.node_bn = add_node(h, name, p.out_side, p.out_side, p.cout);
        if (s.skip_stage >= 0) {
            std::snprintf(name, sizeof(name), "s%d.add", i);
            p.node_add = add_node(h, name, p.out_side, p.out_side, p.cout);
            std::snprintf(name, sizeof(name), "s%d.bn2", i);
            p.node_bn2 = add_node(h, name, p.out_side, p.out_side, p.cout);
        }
        h->stages.push_back(p);
        side = p.out_side;
        ch = p.cout;
    }
    const int flat_len = side * side * ch;
    h->node_flat = add_node(h, "flat", 1, 1, flat_len);
    int nin = flat_len;
    for (int d = 0; d < w->n_dense; ++d) {
        const rn_dense_layer& l = w->dense[d];
        if (l.nin != nin || l.nout < 1 || l.nout > 64 || !l.kernel) {
            rn_set_error("dense %d: expected %d inputs, got %d (is the checkpoint for this im_side?)", d, nin,
                         l.nin);
            return RN_E_INVALID;
        }
        DensePlan p{};
        p.nin = l.nin;
        p.nout = l.nout;
        if ((rc = upload(h, l.kernel, static_cast<size_t>(l.nin) * l.nout, &p.w)) != RN_OK) return rc;
        if (l.bias && (rc = upload(h, l.bias, l.nout, &p.bias)) != RN_OK) return rc;
        if (l.gamma) {
            if (!l.beta || !l.mean || !l.variance) {
                rn_set_error("dense %d: incomplete BN parameters", d);
                return RN_E_INVALID;
            }
            // tf.nn.batch_normalization: inv = rsqrt(var+eps)*gamma; shift = beta - mean*inv
            std::vector<float> inv(l.nout), shift(l.nout);
            for (int j = 0; j < l.nout; ++j) {
                inv[j] = (1.0f / sqrtf(l.variance[j] + w->bn_epsilon)) * l.gamma[j];
                shift[j] = l.beta[j] - l.mean[j] * inv[j];
            }
            if ((rc = upload(h, inv.data(), l.nout, &p.inv)) != RN_OK) return rc;
            if ((rc = upload(h, shift.data(), l.nout, &p.shift)) != RN_OK) return rc;
        }
        std::snprintf(name, sizeof(name), "d%d.mm", d);
        p.node_mm = add_node(h, name, 1, 1, l.nout);
        std::snprintf(name, sizeof(name), "d%d.relu", d);
        p.node_relu = add_node(h, name, 1, 1, l.nout);
        if (l.gamma) {
            std::snprintf(name, sizeof(name), "d%d.bn", d);
            p.node_bn = add_node(h, name, 1, 1, l.nout);
        }
        h->dense.push_back(p);
        nin = l.nout;
    }
    if (nin != w->num_classes) {
        rn_set_error("last dense layer has %d outputs, num_classes is %d", nin, w->num_classes);
        return RN_E_INVALID;
    }
    h->node_softmax = add_node(h, "softmax", 1, 1, w->num_classes);
    return RN_OK;
}

// Allocate activation buffers.  Stage outputs always get their own buffer (they are
// skip sources and are always tappable).  In the unfused path the conv/pool/add
// intermediates share two scratch buffers unless RN_FLAG_TAPS asks for all of them.
int alloc_buffers(rn_handle* h) {
    const size_t nb = static_cast<size_t>(h->max_batch);
    int rc;
    const bool taps = (h->flags & RN_FLAG_TAPS) != 0;
    const bool fused = fused_mode(h);
    void* p = nullptr;
    // input (fp32 RGB): unfused path only; the fused stage 0 reads uint8 directly
    if (!fused) {
        if ((rc = dev_alloc(h, nb * node_elems(h->nodes[h->node_input]) * 4, &p)) != RN_OK) return rc;
        h->nodes[h->node_input].ptr = p;
    }
    size_t scratch_elems = 0;
    for (auto& s : h->stages) {
        const int out_node = s.node_bn2 >= 0 ? s.node_bn2 : s.node_bn;
        for (int id : {s.node_conv, s.node_pool, s.node_bn, s.node_add, s.node_bn2}) {
            if (id < 0) continue;
            NodeBuf& n = h->nodes[id];
            const bool is_out = (id == out_node) || (id == s.node_bn);
            if (fused) {
                if (id == out_node) {
                    n.dtype = h->dtype;
                    if ((rc = dev_alloc(h, nb * node_elems(n) * dtype_size(h->dtype) + 8192, &p)) != RN_OK) return rc;
                    n.ptr = p;
                }
                continue;
            }
            if (taps || is_out) {
                if ((rc = dev_alloc(h, nb * node_elems(n) * 4, &p)) != RN_OK) return rc;
                n.ptr = p;
            } else {
                scratch_elems = std::max(scratch_elems, node_elems(n));
            }
        }
    }
    if (!fused && !taps) {
        void* s0 = nullptr;
        void* s1 = nullptr;
        if ((rc = dev_alloc(h, nb * scratch_elems * 4, &s0)) != RN_OK) return rc;
        if ((rc = dev_alloc(h, nb * scratch_elems * 4, &s1)) != RN_OK) return rc;
        for (auto& s : h->stages) {
            if (s.node_conv >= 0 && !h->nodes[s.node_conv].ptr) h->nodes[s.node_conv].ptr = s0;
            if (s.node_pool >= 0 && !h->nodes[s.node_pool].ptr) h->nodes[s.node_pool].ptr = s1;
            if (s.node_add >= 0 && !h->nodes[s.node_add].ptr) h->nodes[s.node_add].ptr = s0;
        }
    }
    // flat aliases the last stage output
    {
        const StagePlan& last = h->stages.back();
        const NodeBuf& src = h->nodes[last.node_bn2 >= 0 ? last.node_bn2 : last.node_bn];
        h->nodes[h->node_flat].ptr = src.ptr;
        h->nodes[h->node_flat].dtype = src.dtype;
    }
    for (auto& d : h->dense)
        for (int id : {d.node_mm, d.node_relu, d.node_bn}) {
            if (id < 0) continue;
            if ((rc = dev_alloc(h, nb * node_elems(h->nodes[id]) * 4, &p)) != RN_OK) return rc;
            h->nodes[id].ptr = p;
        }
    // staging for the host-buffer entry points + softmax node
    if ((rc = dev_alloc(h, nb * h->im_side * h->im_side * 3, &p)) != RN_OK) return rc;
    h->d_in_u8 = static_cast<uint8_t*>(p);
    if ((rc = dev_alloc(h, nb * h->num_classes * 4, &p)) != RN_OK) return rc;
    h->d_probs = static_cast<float*>(p);
    h->nodes[h->node_softmax].ptr = p;
    if ((rc = dev_alloc(h, nb * 8, &p)) != RN_OK) return rc;
    h->d_ids = static_cast<int64_t*>(p);
    return RN_OK;
}

void record(rn_handle* h, int idx) {
    if (h->profiling && idx < static_cast<int>(h->events.size())) (void)hipEventRecord(h->events[idx], h->stream);
}

void fill_head_args(rn_handle* h, HeadArgs& a) {
    a = HeadArgs{};
    a.n_dense = static_cast<int>(h->dense.size());
    for (int d = 0; d < a.n_dense; ++d) {
        const DensePlan& p = h->dense[d];
        a.nin[d] = p.nin;
        a.nout[d] = p.nout;
        a.w[d] = p.w;
        a.bias[d] = p.bias;
        a.inv[d] = p.inv;
        a.shift[d] = p.shift;
        a.tap_mm[d] = static_cast<float*>(h->nodes[p.node_mm].ptr);
        a.tap_relu[d] = static_cast<float*>(h->nodes[p.node_relu].ptr);
        a.tap_bn[d] = p.node_bn >= 0 ? static_cast<float*>(h->nodes[p.node_bn].ptr) : nullptr;
    }
}

int run_head(rn_handle* h, int n, float* d_probs, int64_t* d_ids) {
    HeadArgs a;
    fill_head_args(h, a);
    const NodeBuf& flat = h->nodes[h->node_flat];
    return rn_launch_head(h->stream, flat.ptr, flat.dtype, n, a, d_probs, d_ids);
}

// Float32 forward: one launch per graph node; without RN_FLAG_TAPS the conv stages rn_stage_f32m.hip covers run as one
// matrix-core launch each.
int forward_unfused(rn_handle* h, const float* d_rgb, int n, float* d_probs, int64_t* d_ids) {
    int rc;
    const float* cur = d_rgb;
    for (size_t i = 0; i < h->stages.size(); ++i) {
        StagePlan& s = h->stages[i];
        if (rn_f32m_covers(h, static_cast<int>(i))) {
            // the whole stage in one launch on the matrix cores; only its output node is written
            if ((rc = rn_f32m_launch(h, static_cast<int>(i), cur, n)) != RN_OK) return rc;
            cur = static_cast<const float*>(h->nodes[s.node_bn2 >= 0 ? s.node_bn2 : s.node_bn].ptr);
            record(h, 2 + static_cast<int>(i));
            continue;
        }
        if (i == 0 && h->f32m && s.cin == 3 && s.cout == 8 && s.pool_k == 3 && s.pool_s == 1 && s.skip_stage < 0) {
            // (handles without taps) stage 0 in one launch, bit-identical to its per-node launches
            float* bn0 = static_cast<float*>(h->nodes[s.node_bn].ptr);
            if ((rc = rn_launch_stage0_fused_f32(h->stream, cur, s.w_f32, bn0, n, s.in_side, s.bn)) != RN_OK) return rc;
            cur = bn0;
            record(h, 2 + static_cast<int>(i));
            continue;
        }
        float* conv = static_cast<float*>(h->nodes[s.node_conv].ptr);
        if ((rc = rn_launch_conv3x3_relu6_f32(h->stream, cur, s.w_f32, conv, n, s.in_side, s.in_side, s.cin,
                                              s.cout)) != RN_OK)
            return rc;
        const float* pooled = conv;
        if (s.pool_k) {
            float* pl = static_cast<float*>(h->nodes[s.node_pool].ptr);
            if ((rc = rn_launch_avgpool_f32(h->stream, conv, pl, n, s.conv_side, s.conv_side, s.cout, s.pool_k,
                                            s.pool_s)) != RN_OK)
                return rc;
            pooled = pl;
        }
        float* bn = static_cast<float*>(h->nodes[s.node_bn].ptr);
        const int64_t npix = static_cast<int64_t>(n) * s.out_side * s.out_side;
        if ((rc = rn_launch_bn_f32(h->stream, pooled, bn, npix, s.cout, s.bn)) != RN_OK) return rc;
        cur = bn;
        if (s.skip_stage >= 0) {
            const StagePlan& sk = h->stages[s.skip_stage];
            const float* skip = static_cast<const float*>(h->nodes[sk.node_bn].ptr);
            float* add = static_cast<float*>(h->nodes[s.node_add].ptr);
            if ((rc = rn_launch_resize_add_f32(h->stream, bn, skip, add, n, s.out_side, s.skip_side, s.cout,
                                               s.rt)) != RN_OK)
                return rc;
            float* bn2 = static_cast<float*>(h->nodes[s.node_bn2].ptr);
            if ((rc = rn_launch_bn_f32(h->stream, add, bn2, npix, s.cout, s.bn2)) != RN_OK) return rc;
            cur = bn2;
        }
        record(h, 2 + static_cast<int>(i));
    }
    if ((rc = run_head(h, n, d_probs, d_ids)) != RN_OK) return rc;
    record(h, 2 + static_cast<int>(h->stages.size()));
    return RN_OK;
}

int check_call(rn_handle* h, int n, const void* a, const void* b, const void* c) {
    if (!h) {
        rn_set_error("null handle");
        return RN_E_INVALID;
    }
    if (!a || !b || !c) {
        rn_set_error("null buffer");
        return RN_E_INVALID;
    }
    if (n < 1 || n > h->max_batch) {
        rn_set_error("batch %d out of range (max_batch %d)", n, h->max_batch);
        return RN_E_RANGE;
    }
    return RN_OK;
}

}  // namespace

int rn_run_head(rn_handle* h, int n, float* d_probs, int64_t* d_ids) { return run_head(h, n, d_probs, d_ids); }
void rn_fill_head_args(rn_handle* h, HeadArgs* a) { fill_head_args(h, *a); }
void rn_record_event(rn_handle* h, int idx) { record(h, idx); }

// ---------------------------------------------------------------------------- API
extern "C" int rn_create(const rn_weights* w, int device, int dtype, int max_batch, unsigned flags,
                         rn_handle** out) {
    if (!out) {
        rn_set_error("rn_create: null out pointer");
        return RN_E_INVALID;
    }
    *out = nullptr;
    int rc = validate(w, dtype, max_batch, flags);
    if (rc != RN_OK) return rc;
    int ndev = 0;
    if (hipGetDeviceCount(&ndev) != hipSuccess || ndev < 1) {
        rn_set_error("rn_create: no HIP device available");
        return RN_E_HIP;
    }
    if (device < 0 || device >= ndev) {
        rn_set_error("rn_create: device %d out of range (%d devices)", device, ndev);
        return RN_E_INVALID;
    }
    DeviceGuard guard(device);
    if (!guard.ok) {
        rn_set_error("rn_create: hipSetDevice(%d) failed", device);
        return RN_E_HIP;
    }
    rn_handle* h = new (std::nothrow) rn_handle();
    if (!h) {
        rn_set_error("rn_create: out of host memory");
        return RN_E_NOMEM;
    }
    {
        // the kernels are gfx950 code sized for its 160 KiB of LDS per CU: refuse anything else up front
        hipDeviceProp_t prop{};
        if (hipGetDeviceProperties(&prop, device) != hipSuccess) {
            rn_set_error("rn_create: hipGetDeviceProperties(%d) failed", device);
            delete h;
            return RN_E_HIP;
        }
        if (strncmp(prop.gcnArchName, "gfx950", 6) != 0 || prop.maxSharedMemoryPerMultiProcessor < 160 * 1024) {
            rn_set_error("rn_create: device %d is %s with %zu bytes of LDS per CU; this library is built for gfx950 (MI355X, 160 KiB)",
                         device, prop.gcnArchName, static_cast<size_t>(prop.maxSharedMemoryPerMultiProcessor));
            delete h;
            return RN_E_INVALID;
        }
        h->n_cu = prop.multiProcessorCount;
    }
    h->device = device;
    h->dtype = dtype;
    h->flags = flags;
    h->max_batch = max_batch;
    h->im_side = w->im_side;
    h->num_classes = w->num_classes;
    h->bn_eps = w->bn_epsilon;
    auto fail = [&](int code) {
        rn_destroy(h);
        return code;
    };
    if (hipStreamCreateWithFlags(&h->own_stream, hipStreamNonBlocking) != hipSuccess) {
        rn_set_error("rn_create: hipStreamCreate failed");
        return fail(RN_E_HIP);
    }
    h->stream = h->own_stream;
    // ---- float32 handles on the matrix-core path: frozen first-BN channels of a 64 -> 64 residual stage (stage 5).  The stage
    // kernels form y1 = ((x * 1/16 - mean) * inv + beta) un-contracted: where |inv| * max(|mean|, |6 - mean|) < 2^-25 |beta| the
    // product vanishes against beta and y1 IS beta for every input (the reference's float32 computes the same expression).  With
    // >= 32 such channels the stage's channels are relabelled on a copy of the weights (stage 4's couts, stage 5's cins + couts,
    // stage 6's cins: the residual pairs channel c of stage 4's output with channel c of stage 5's) so that the second 32-cout
    // tile is all frozen: it is not convolved (rn_f32m_launch runs the residual for it alone).  rn_tap un-relabels.
    std::vector<rn_conv_stage> stg_copy;
    std::vector<std::vector<float>> owned;
    rn_weights wp;
    std::vector<int> fold_pi, kfold_pi;
    int fold_r = -1, kfold_r = -1, kfold_live = 0, kfold_proven = 0;
    if (!fused_mode(h) && !(flags & (RN_FLAG_TAPS | RN_FLAG_COMPUTE_FROZEN))) {
        for (int r = 2; r + 1 < w->n_stages && fold_r < 0; ++r) {
            const rn_conv_stage& s5 = w->stages[r];
            const rn_conv_stage& s4 = w->stages[r - 1];
            const rn_conv_stage& s6 = w->stages[r + 1];
            if (!(s5.cin == 64 && s5.cout == 64 && s5.pool_k == 4 && s5.pool_s == 2 && s5.skip_stage == r - 1 && s5.gamma2 && s4.cout == 64 &&
                  s4.skip_stage < 0 && s6.cin == 64 && s6.skip_stage < 0 && s5.gamma && s5.beta && s5.mean && s5.variance))
                continue;
            bool other_use = false;
            for (int k = 0; k < w->n_stages; ++k) other_use |= (k != r && w->stages[k].skip_stage == r - 1) || w->stages[k].skip_stage == r;
            if (other_use) continue;
            std::vector<int> frozen, live;
            for (int c = 0; c < 64; ++c) {
                const float inv = (1.0f / sqrtf(s5.variance[c] + w->bn_epsilon)) * s5.gamma[c];
                const double reach = std::max(std::fabs(static_cast<double>(s5.mean[c])), std::fabs(6.0 - static_cast<double>(s5.mean[c])));
                const bool fz = std::fabs(static_cast<double>(inv)) * reach * (1.0 + 1e-6) < std::fabs(static_cast<double>(s5.beta[c])) * 2.98023223876953125e-8;
                (fz ? frozen : live).push_back(c);
            }
            if (frozen.size() < 32) continue;
            while (frozen.size() > 32) {
                live.push_back(frozen.back());
                frozen.pop_back();
            }
            std::sort(live.begin(), live.end());
            fold_pi.resize(64);
            for (int p = 0; p < 32; ++p) fold_pi[p] = live[p];
            for (int p = 0; p < 32; ++p) fold_pi[32 + p] = frozen[p];
            fold_r = r;
        }
        auto ensure_copy = [&]() {
            if (!stg_copy.empty()) return;
            stg_copy.assign(w->stages, w->stages + w->n_stages);
            wp = *w;
            wp.stages = stg_copy.data();
            w = &wp;
        };
        auto perm_vec = [&](const std::vector<int>& pi, const float* src) -> const float* {
            owned.emplace_back(pi.size());
            for (size_t p = 0; p < pi.size(); ++p) owned.back()[p] = src[pi[p]];
            return owned.back().data();
        };
        auto perm_kernel = [&](const std::vector<int>& pi, const float* src, int cin, int cout, bool pin, bool pout) -> const float* {
            owned.emplace_back(static_cast<size_t>(9) * cin * cout);
            std::vector<float>& dst = owned.back();
            for (int tap = 0; tap < 9; ++tap)
                for (int ci = 0; ci < cin; ++ci)
                    for (int co = 0; co < cout; ++co)
                        dst[(static_cast<size_t>(tap) * cin + ci) * cout + co] = src[(static_cast<size_t>(tap) * cin + (pin ? pi[ci] : ci)) * cout + (pout ? pi[co] : co)];
            return dst.data();
        };
        auto bn1_frozen = [&](const rn_conv_stage& st, int c) {
            const float inv = (1.0f / sqrtf(st.variance[c] + w->bn_epsilon)) * st.gamma[c];
            const double reach = std::max(std::fabs(static_cast<double>(st.mean[c])), std::fabs(6.0 - static_cast<double>(st.mean[c])));
            return std::fabs(static_cast<double>(inv)) * reach * (1.0 + 1e-6) < std::fabs(static_cast<double>(st.beta[c])) * 2.98023223876953125e-8;
        };
        if (fold_r >= 0) {
            const std::vector<int>& pi = fold_pi;
            ensure_copy();
            const int r = fold_r;
            const rn_conv_stage s4 = stg_copy[r - 1], s5 = stg_copy[r], s6 = stg_copy[r + 1];
            stg_copy[r - 1].kernel = perm_kernel(pi, s4.kernel, s4.cin, 64, false, true);
            stg_copy[r - 1].gamma = perm_vec(pi, s4.gamma);
            stg_copy[r - 1].beta = perm_vec(pi, s4.beta);
            stg_copy[r - 1].mean = perm_vec(pi, s4.mean);
            stg_copy[r - 1].variance = perm_vec(pi, s4.variance);
            stg_copy[r].kernel = perm_kernel(pi, s5.kernel, 64, 64, true, true);
            stg_copy[r].gamma = perm_vec(pi, s5.gamma);
            stg_copy[r].beta = perm_vec(pi, s5.beta);
            stg_copy[r].mean = perm_vec(pi, s5.mean);
            stg_copy[r].variance = perm_vec(pi, s5.variance);
            stg_copy[r].gamma2 = perm_vec(pi, s5.gamma2);
            stg_copy[r].beta2 = perm_vec(pi, s5.beta2);
            stg_copy[r].mean2 = perm_vec(pi, s5.mean2);
            stg_copy[r].variance2 = perm_vec(pi, s5.variance2);
            stg_copy[r + 1].kernel = perm_kernel(pi, s6.kernel, 64, s6.cout, true, false);
        }
        // ---- ... and frozen INPUT channels: a pooled stage p without a second BN whose output only feeds the convolution of stage
        // p + 1 (nobody's residual) and whose BN freezes >= 8 channels (same inequality).  Its couts / the consumer's cins are
        // relabelled so that the last 8 k channels are all frozen; the consumer contracts the others and starts its accumulators
        // from the frozen ones' contribution (rn_f32m_prepare: a variant of the stage kernel must exist for that channel count)
        for (int r = 1; r < w->n_stages && kfold_r < 0; ++r) {
            const int p = r - 1;
            const rn_conv_stage& sp = w->stages[p];
            const rn_conv_stage& sc = w->stages[r];
            if (fold_r >= 0 && (p == fold_r - 1 || p == fold_r)) continue;          // (those channels are relabelled already)
            if (sp.skip_stage >= 0 || sp.gamma2 || sp.pool_k != 4 || !sp.gamma || !sp.beta || !sp.mean || !sp.variance) continue;
            if (sc.cin != sp.cout || sp.cout % 8 != 0 || sp.cout > 64) continue;
            bool other_use = false;
            for (int k = 0; k < w->n_stages; ++k) other_use |= w->stages[k].skip_stage == p;
            if (other_use) continue;
            std::vector<int> frozen, live;
            for (int c = 0; c < sp.cout; ++c) (bn1_frozen(sp, c) ? frozen : live).push_back(c);
            const int proven = static_cast<int>(frozen.size()), nf = proven / 8 * 8;
            if (nf < 8 || nf >= sp.cout) continue;
            while (static_cast<int>(frozen.size()) > nf) {
                live.push_back(frozen.back());
                frozen.pop_back();
            }
            std::sort(live.begin(), live.end());
            kfold_pi = live;
            kfold_pi.insert(kfold_pi.end(), frozen.begin(), frozen.end());
            kfold_r = r;
            kfold_live = sp.cout - nf;
            kfold_proven = proven;
        }
        if (kfold_r >= 0) {
            const std::vector<int>& pi = kfold_pi;
            ensure_copy();
            const int r = kfold_r;
            const rn_conv_stage sp = stg_copy[r - 1], sc = stg_copy[r];
            stg_copy[r - 1].kernel = perm_kernel(pi, sp.kernel, sp.cin, sp.cout, false, true);
            stg_copy[r - 1].gamma = perm_vec(pi, sp.gamma);
            stg_copy[r - 1].beta = perm_vec(pi, sp.beta);
            stg_copy[r - 1].mean = perm_vec(pi, sp.mean);
            stg_copy[r - 1].variance = perm_vec(pi, sp.variance);
            stg_copy[r].kernel = perm_kernel(pi, sc.kernel, sc.cin, sc.cout, true, false);
        }
    }
    if ((rc = build_plan(h, w)) != RN_OK) return fail(rc);
    if (fold_r >= 0) {
        h->f32_fold_stage = fold_r;
        h->f32_fold_live = 32;
        h->node_perm[h->stages[fold_r - 1].node_bn] = fold_pi;
        h->node_perm[h->stages[fold_r].node_bn2] = fold_pi;
    }
    if (kfold_r >= 0) {
        h->f32_kfold_stage = kfold_r;           // (rn_f32m_prepare takes it back if no kernel variant contracts that channel count)
        h->f32_kfold_live = kfold_live;
        h->f32_kfold_proven = kfold_proven;
        h->node_perm[h->stages[kfold_r - 1].node_bn] = kfold_pi;
    }
    // uint8 -> float32 table, evaluated in float64 like the reference's NumPy expression
    {
        float lut[256];
        for (int v = 0; v < 256; ++v) lut[v] = static_cast<float>(((static_cast<double>(v) / 255.) * 2) - 1);
        if ((rc = upload(h, lut, 256, &h->lut)) != RN_OK) return fail(rc);
    }
    if (fused_mode(h) && (rc = rn_fused_prepare(h, w)) != RN_OK) return fail(rc);
    // float32 handles without per-node taps run their conv stages on the matrix cores (rn_stage_f32m.hip)
    if (!fused_mode(h) && !(h->flags & RN_FLAG_TAPS) && (rc = rn_f32m_prepare(h, w)) != RN_OK) return fail(rc);
    if ((rc = alloc_buffers(h)) != RN_OK) return fail(rc);
    if (fused_mode(h) && (rc = rn_fused_post_alloc(h)) != RN_OK) return fail(rc);
    h->events.resize(3 + h->stages.size());
    for (auto& e : h->events)
        if (hipEventCreate(&e) != hipSuccess) {
            rn_set_error("rn_create: hipEventCreate failed");
            return fail(RN_E_HIP);
        }
    if (hipDeviceSynchronize() != hipSuccess) {
        rn_set_error("rn_create: device synchronize failed");
        return fail(RN_E_HIP);
    }
    *out = h;
    return RN_OK;
}

extern "C" void rn_destroy(rn_handle* h) {
    if (!h) return;
    DeviceGuard guard(h->device);
    (void)hipDeviceSynchronize();
    for (auto e : h->events)
        if (e) (void)hipEventDestroy(e);
    for (void* p : h->allocs) (void)hipFree(p);
    if (h->d_raw) (void)hipFree(h->d_raw);
    if (h->d_items) (void)hipFree(h->d_items);
    for (auto& sl : h->slots) {
        if (sl.d_in) (void)hipFree(sl.d_in);
        if (sl.d_probs) (void)hipFree(sl.d_probs);
        if (sl.d_ids) (void)hipFree(sl.d_ids);
        if (sl.h_probs) (void)hipHostFree(sl.h_probs);
        if (sl.h_ids) (void)hipHostFree(sl.h_ids);
        if (sl.uploaded) (void)hipEventDestroy(sl.uploaded);
        if (sl.done) (void)hipEventDestroy(sl.done);
    }
    if (h->copy_stream) (void)hipStreamDestroy(h->copy_stream);
    rn_fused_release(h);
    rn_f32m_release(h);
    if (h->own_stream) (void)hipStreamDestroy(h->own_stream);
    delete h;
}

extern "C" int rn_set_stream(rn_handle* h, void* hip_stream) {
    if (!h) {
        rn_set_error("null handle");
        return RN_E_INVALID;
    }
    h->stream = hip_stream ? static_cast<hipStream_t>(hip_stream) : h->own_stream;
    return RN_OK;
}

extern "C" int rn_set_stream_null(rn_handle* h) {
    if (!h) {
        rn_set_error("null handle");
        return RN_E_INVALID;
    }
    h->stream = nullptr;          // hipStream_t(0): the null stream
    return RN_OK;
}

extern "C" int rn_sync(rn_handle* h) {
    if (!h) {
        rn_set_error("null handle");
        return RN_E_INVALID;
    }
    DeviceGuard guard(h->device);
    RN_HIP(hipStreamSynchronize(h->stream));
    return RN_OK;
}

extern "C" int rn_forward_f32_device(rn_handle* h, const float* d_rgb, int n, float* d_probs, int64_t* d_ids) {
    int rc = check_call(h, n, d_rgb, d_probs, d_ids);
    if (rc != RN_OK) return rc;
    DeviceGuard guard(h->device);
    (void)hipGetLastError();      // a stale status of the CALLER's own runtime calls on this thread is not this pass's launch failure
    h->timing_valid = false;
    record(h, 0);
    record(h, 1);
    if (fused_mode(h))
        rc = rn_fused_forward(h, nullptr, d_rgb, n, d_probs, d_ids);
    else {
        // keep the "input" node readable through rn_tap
        float* in_node = static_cast<float*>(h->nodes[h->node_input].ptr);
        if (in_node != d_rgb)
            RN_HIP(hipMemcpyAsync(in_node, d_rgb, static_cast<size_t>(n) * h->im_side * h->im_side * 3 * 4,
                                  hipMemcpyDeviceToDevice, h->stream));
        rc = forward_unfused(h, in_node, n, d_probs, d_ids);
    }
    if (rc != RN_OK) return rc;
    h->last_n = n;
    h->timing_valid = h->profiling;
    return RN_OK;
}

extern "C" int rn_forward_u8_device(rn_handle* h, const uint8_t* d_bgr, int n, float* d_probs, int64_t* d_ids) {
    int rc = check_call(h, n, d_bgr, d_probs, d_ids);
    if (rc != RN_OK) return rc;
    DeviceGuard guard(h->device);
    (void)hipGetLastError();      // (see rn_forward_f32_device)
    h->timing_valid = false;
    record(h, 0);
    if (fused_mode(h)) {
        record(h, 1);
        rc = rn_fused_forward(h, d_bgr, nullptr, n, d_probs, d_ids);
    } else {
        float* in_node = static_cast<float*>(h->nodes[h->node_input].ptr);
        rc = rn_launch_preprocess_u8(h->stream, d_bgr, in_node, h->lut,
                                     static_cast<int64_t>(n) * h->im_side * h->im_side);
        if (rc != RN_OK) return rc;
        record(h, 1);
        rc = forward_unfused(h, in_node, n, d_probs, d_ids);
    }
    if (rc != RN_OK) return rc;
    h->last_n = n;
    h->timing_valid = h->profiling;
    return RN_OK;
}

extern "C" int rn_forward_u8(rn_handle* h, const uint8_t* bgr, int n, float* probs, int64_t* ids) {
    int rc = check_call(h, n, bgr, probs, ids);
    if (rc != RN_OK) return rc;
    DeviceGuard guard(h->device);
    const size_t in_bytes = static_cast<size_t>(n) * h->im_side * h->im_side * 3;
    RN_HIP(hipMemcpyAsync(h->d_in_u8, bgr, in_bytes, hipMemcpyHostToDevice, h->stream));
    if ((rc = rn_forward_u8_device(h, h->d_in_u8, n, h->d_probs, h->d_ids)) != RN_OK) return rc;
    RN_HIP(hipMemcpyAsync(probs, h->d_probs, static_cast<size_t>(n) * h->num_classes * 4, hipMemcpyDeviceToHost,
                          h->stream));
    RN_HIP(hipMemcpyAsync(ids, h->d_ids, static_cast<size_t>(n) * 8, hipMemcpyDeviceToHost, h->stream));
    RN_HIP(hipStreamSynchronize(h->stream));
    return RN_OK;
}

// ---- two-slot host pipeline: the upload of batch k+1 overlaps the forward pass of batch k
extern "C" int rn_submit_u8(rn_handle* h, const uint8_t* bgr, int n, int slot) {
    if (!h || !bgr || slot < 0 || slot > 1) {
        rn_set_error("rn_submit_u8: bad argument");
        return RN_E_INVALID;
    }
    if (n < 1 || n > h->max_batch) {
        rn_set_error("rn_submit_u8: n = %d out of range (1..%d)", n, h->max_batch);
        return RN_E_RANGE;
    }
    rn_handle::HostSlot& sl = h->slots[slot];
    if (sl.busy) {
        rn_set_error("rn_submit_u8: slot %d holds results that were not collected", slot);
        return RN_E_STATE;
    }
    DeviceGuard guard(h->device);
    if (!h->copy_stream) RN_HIP(hipStreamCreateWithFlags(&h->copy_stream, hipStreamNonBlocking));
    if (!sl.d_in) {
        const size_t in_bytes = static_cast<size_t>(h->max_batch) * h->im_side * h->im_side * 3;
        RN_HIP(hipMalloc(reinterpret_cast<void**>(&sl.d_in), in_bytes));
        RN_HIP(hipMalloc(reinterpret_cast<void**>(&sl.d_probs), static_cast<size_t>(h->max_batch) * h->num_classes * 4));
        RN_HIP(hipMalloc(reinterpret_cast<void**>(&sl.d_ids), static_cast<size_t>(h->max_batch) * 8));
        RN_HIP(hipHostMalloc(reinterpret_cast<void**>(&sl.h_probs), static_cast<size_t>(h->max_batch) * h->num_classes * 4, 0));
        RN_HIP(hipHostMalloc(reinterpret_cast<void**>(&sl.h_ids), static_cast<size_t>(h->max_batch) * 8, 0));
        RN_HIP(hipEventCreateWithFlags(&sl.uploaded, hipEventDisableTiming));
        RN_HIP(hipEventCreateWithFlags(&sl.done, hipEventDisableTiming));
    }
    const size_t in_bytes = static_cast<size_t>(n) * h->im_side * h->im_side * 3;
    // from pageable memory this call returns once the bytes are staged: the previous batch's kernels, enqueued earlier on
    // the compute stream, run meanwhile; from pinned memory it is asynchronous outright
    RN_HIP(hipMemcpyAsync(sl.d_in, bgr, in_bytes, hipMemcpyHostToDevice, h->copy_stream));
    RN_HIP(hipEventRecord(sl.uploaded, h->copy_stream));
    RN_HIP(hipStreamWaitEvent(h->stream, sl.uploaded, 0));
    int rc = rn_forward_u8_device(h, sl.d_in, n, sl.d_probs, sl.d_ids);
    if (rc != RN_OK) return rc;
    RN_HIP(hipMemcpyAsync(sl.h_probs, sl.d_probs, static_cast<size_t>(n) * h->num_classes * 4, hipMemcpyDeviceToHost, h->stream));
    RN_HIP(hipMemcpyAsync(sl.h_ids, sl.d_ids, static_cast<size_t>(n) * 8, hipMemcpyDeviceToHost, h->stream));
    RN_HIP(hipEventRecord(sl.done, h->stream));
    sl.n = n;
    sl.busy = true;
    return RN_OK;
}

extern "C" int rn_collect(rn_handle* h, int slot, float* probs, int64_t* ids) {
    if (!h || !probs || !ids || slot < 0 || slot > 1) {
        rn_set_error("rn_collect: bad argument");
        return RN_E_INVALID;
    }
    rn_handle::HostSlot& sl = h->slots[slot];
    if (!sl.busy) {
        rn_set_error("rn_collect: nothing was submitted to slot %d", slot);
        return RN_E_STATE;
    }
    DeviceGuard guard(h->device);
    RN_HIP(hipEventSynchronize(sl.done));
    std::memcpy(probs, sl.h_probs, static_cast<size_t>(sl.n) * h->num_classes * 4);
    std::memcpy(ids, sl.h_ids, static_cast<size_t>(sl.n) * 8);
    sl.busy = false;
    return RN_OK;
}

// ---- caller-side image pipeline on the device (network.py:137-156)
namespace {
// network.py:137-146: the centred square window of an h x w image (abs((w - h) // 2) with Python floor division)
void center_crop_window(int hh, int ww, int* x0, int* y0, int* side) {
    if (hh == ww) {
        *x0 = *y0 = 0;
        *side = hh;
    } else if (ww > hh) {
        *x0 = (ww - hh) / 2;
        *y0 = 0;
        *side = hh;
    } else {
        // abs((w - h) // 2): floor division of a negative number rounds away from zero before abs()
        *x0 = 0;
        *y0 = (hh - ww + 1) / 2;
        *side = ww;
    }
}
}  // namespace

extern "C" int rn_crop_resize_u8_device(rn_handle* h, const uint8_t* d_src, int src_h, int src_w, uint8_t* d_dst_batch,
                                        int index) {
    if (!h || !d_src || !d_dst_batch || src_h < 1 || src_w < 1 || index < 0 || index >= h->max_batch) {
        rn_set_error("rn_crop_resize_u8_device: bad argument (image %dx%d, slot %d of %d)", src_w, src_h, index,
                     h ? h->max_batch : 0);
        return RN_E_INVALID;
    }
    DeviceGuard guard(h->device);
    int x0, y0, side;
    center_crop_window(src_h, src_w, &x0, &y0, &side);
    const int S = h->im_side;
    return rn_launch_resize_u8(h->stream, d_src + (static_cast<int64_t>(y0) * src_w + x0) * 3, side, side,
                               static_cast<int64_t>(src_w) * 3, d_dst_batch + static_cast<int64_t>(index) * S * S * 3, S, S);
}

// the device table of a batched resize (max_batch entries), allocated at the first use
static int ensure_items(rn_handle* h) {
    if (h->d_items) return RN_OK;
    void* p = nullptr;
    hipError_t e = hipMalloc(&p, static_cast<size_t>(h->max_batch) * sizeof(rn_resize_item));
    if (e != hipSuccess) {
        rn_set_error("hipMalloc(resize table) failed: %s", hipGetErrorString(e));
        return RN_E_NOMEM;
    }
    h->d_items = static_cast<rn_resize_item*>(p);
    h->items_host.resize(static_cast<size_t>(h->max_batch));
    return RN_OK;
}

extern "C" int rn_crop_resize_batch_u8_device(rn_handle* h, const uint8_t* const* d_srcs, const int* heights, const int* widths, int n,
                                              uint8_t* d_dst_batch) {
    if (!h || !d_srcs || !heights || !widths || !d_dst_batch || n < 1 || n > h->max_batch) {
        rn_set_error("rn_crop_resize_batch_u8_device: bad argument (%d images, max_batch %d)", n, h ? h->max_batch : 0);
        return RN_E_INVALID;
    }
    DeviceGuard guard(h->device);
    int rc = ensure_items(h);
    if (rc != RN_OK) return rc;
    // (the host copy of the table may be rewritten at once: a hipMemcpyAsync out of pageable memory returns when its source has been
    //  read into the runtime's staging buffer -- the property rn_group_forward_u8's upload threads rely on too)
    for (int i = 0; i < n; ++i) {
        if (!d_srcs[i] || heights[i] < 1 || widths[i] < 1) {
            rn_set_error("rn_crop_resize_batch_u8_device: image %d is empty", i);
            return RN_E_INVALID;
        }
        int x0, y0, side;
        center_crop_window(heights[i], widths[i], &x0, &y0, &side);
        rn_resize_item_fill(&h->items_host[i], d_srcs[i] + (static_cast<int64_t>(y0) * widths[i] + x0) * 3, side, side,
                            static_cast<int64_t>(widths[i]) * 3, h->im_side);
    }
    RN_HIP(hipMemcpyAsync(h->d_items, h->items_host.data(), static_cast<size_t>(n) * sizeof(rn_resize_item), hipMemcpyHostToDevice, h->stream));
    return rn_launch_resize_batch_u8(h->stream, h->d_items, n, d_dst_batch, h->im_side);
}

extern "C" int rn_classify_images_u8(rn_handle* h, const uint8_t* const* images, const int* heights, const int* widths, int n,
                                     float* probs, int64_t* ids) {
    int rc = check_call(h, n, images, probs, ids);
    if (rc != RN_OK) return rc;
    if (!heights || !widths) {
        rn_set_error("rn_classify_images_u8: heights / widths missing");
        return RN_E_INVALID;
    }
    DeviceGuard guard(h->device);
    // one staging buffer for the raw images, grown to the batch's total size: uploads and resize launches are queued
    // on the handle's stream back to back
    // (only the centred square window of each image crosses PCIe: a 1920x1080 frame uploads 1080x1080)
    size_t total = 0;
    for (int i = 0; i < n; ++i) {
        if (!images[i] || heights[i] < 1 || widths[i] < 1) {
            rn_set_error("rn_classify_images_u8: image %d is empty", i);
            return RN_E_INVALID;
        }
        const size_t side = static_cast<size_t>(heights[i] < widths[i] ? heights[i] : widths[i]);
        total += side * side * 3;
    }
    if (total > h->raw_cap) {
        if (h->d_raw) (void)hipFree(h->d_raw);
        h->d_raw = nullptr;
        h->raw_cap = 0;
        void* p = nullptr;
        const size_t cap = total + total / 4;
        hipError_t e = hipMalloc(&p, cap);
        if (e != hipSuccess) {
            rn_set_error("hipMalloc(%zu bytes of image staging) failed: %s", cap, hipGetErrorString(e));
            return RN_E_NOMEM;
        }
        h->d_raw = static_cast<uint8_t*>(p);
        h->raw_cap = cap;
    }
    size_t off = 0;
    const int S = h->im_side;
    if ((rc = ensure_items(h)) != RN_OK) return rc;
    for (int i = 0; i < n; ++i) {
        int x0, y0, side;
        center_crop_window(heights[i], widths[i], &x0, &y0, &side);
        const size_t row = static_cast<size_t>(side) * 3, bytes = row * side;
        const uint8_t* win = images[i] + (static_cast<size_t>(y0) * widths[i] + x0) * 3;
        if (side == widths[i])
            RN_HIP(hipMemcpyAsync(h->d_raw + off, win, bytes, hipMemcpyHostToDevice, h->stream));
        else
            RN_HIP(hipMemcpy2DAsync(h->d_raw + off, row, win, static_cast<size_t>(widths[i]) * 3, row, side,
                                    hipMemcpyHostToDevice, h->stream));
        rn_resize_item_fill(&h->items_host[i], h->d_raw + off, side, side, static_cast<int64_t>(row), S);
        off += bytes;
    }
    // ONE resize launch for the batch (the crop windows' table goes up behind the images)
    RN_HIP(hipMemcpyAsync(h->d_items, h->items_host.data(), static_cast<size_t>(n) * sizeof(rn_resize_item), hipMemcpyHostToDevice, h->stream));
    if ((rc = rn_launch_resize_batch_u8(h->stream, h->d_items, n, h->d_in_u8, S)) != RN_OK) return rc;
    if ((rc = rn_forward_u8_device(h, h->d_in_u8, n, h->d_probs, h->d_ids)) != RN_OK) return rc;
    RN_HIP(hipMemcpyAsync(probs, h->d_probs, static_cast<size_t>(n) * h->num_classes * 4, hipMemcpyDeviceToHost,
                          h->stream));
    RN_HIP(hipMemcpyAsync(ids, h->d_ids, static_cast<size_t>(n) * 8, hipMemcpyDeviceToHost, h->stream));
    RN_HIP(hipStreamSynchronize(h->stream));
    return RN_OK;
}

extern "C" int rn_forward_f32(rn_handle* h, const float* rgb, int n, float* probs, int64_t* ids) {
    int rc = check_call(h, n, rgb, probs, ids);
    if (rc != RN_OK) return rc;
    if (fused_mode(h)) {
        rn_set_error("rn_forward_f32: 16-bit handles take uint8 input (use rn_forward_u8)");
        return RN_E_STATE;
    }
    DeviceGuard guard(h->device);
    float* in_node = static_cast<float*>(h->nodes[h->node_input].ptr);
    const size_t in_bytes = static_cast<size_t>(n) * h->im_side * h->im_side * 3 * 4;
    RN_HIP(hipMemcpyAsync(in_node, rgb, in_bytes, hipMemcpyHostToDevice, h->stream));
    if ((rc = rn_forward_f32_device(h, in_node, n, h->d_probs, h->d_ids)) != RN_OK) return rc;
    RN_HIP(hipMemcpyAsync(probs, h->d_probs, static_cast<size_t>(n) * h->num_classes * 4, hipMemcpyDeviceToHost,
                          h->stream));
    RN_HIP(hipMemcpyAsync(ids, h->d_ids, static_cast<size_t>(n) * 8, hipMemcpyDeviceToHost, h->stream));
    RN_HIP(hipStreamSynchronize(h->stream));
    return RN_OK;
}

extern "C" int rn_node_count(const rn_handle* h) { return h ? static_cast<int>(h->nodes.size()) : 0; }

extern "C" int rn_node_info_get(const rn_handle* h, int node_id, rn_node_info* out) {
    if (!h || !out || node_id < 0 || node_id >= static_cast<int>(h->nodes.size())) {
        rn_set_error("rn_node_info_get: bad argument");
        return RN_E_RANGE;
    }
    *out = h->nodes[node_id].info;
    return RN_OK;
}

extern "C" int rn_tap(rn_handle* h, int node_id, float* out, size_t cap_elems, size_t* n_elems) {
    if (!h || !out || node_id < 0 || node_id >= static_cast<int>(h->nodes.size())) {
        rn_set_error("rn_tap: bad argument");
        return RN_E_RANGE;
    }
    if (h->last_n < 1) {
        rn_set_error("rn_tap: no forward pass has run on this handle");
        return RN_E_STATE;
    }
    const NodeBuf& nb = h->nodes[node_id];
    if (!nb.ptr) {
        rn_set_error("rn_tap: node %s is not materialised on this handle (create with RN_FLAG_TAPS)", nb.info.name);
        return RN_E_STATE;
    }
    if (fused_mode(h)) {
        bool fused_away = false;
        for (size_t i = 0; i < h->stages.size(); ++i)
            fused_away |= node_id == h->stages[i].node_bn && rn_fused_stage_elided(h, static_cast<int>(i));
        if (fused_away) {
            rn_set_error("rn_tap: node %s is fused into its successor's kernel on this handle and never written "
                         "(create with RN_FLAG_STAGE_LAUNCHES)", nb.info.name);
            return RN_E_STATE;
        }
    }
    const bool shared_scratch = !(h->flags & RN_FLAG_TAPS) && !fused_mode(h);
    if (shared_scratch) {
        // a residual stage that runs as ONE matrix-core launch (rn_stage_f32m.hip) writes its output node (sK.bn2) only: the
        // first BN output never leaves the kernel
        for (size_t i = 0; i < h->stages.size(); ++i) {
            const StagePlan& s = h->stages[i];
            if (node_id == s.node_bn && s.node_bn2 >= 0 && rn_f32m_covers(h, static_cast<int>(i))) {
                rn_set_error("rn_tap: node %s stays inside the stage's matrix-core launch on this handle and is never written "
                             "(create with RN_FLAG_TAPS)", nb.info.name);
                return RN_E_STATE;
            }
        }
        const char* nm = nb.info.name;
        const size_t len = std::strlen(nm);
        const bool inter = (len > 5 && (!std::strcmp(nm + len - 5, ".conv") || !std::strcmp(nm + len - 5, ".pool"))) ||
                           (len > 4 && !std::strcmp(nm + len - 4, ".add"));
        if (inter) {
            rn_set_error("rn_tap: node %s shares scratch memory on this handle (create with RN_FLAG_TAPS)", nm);
            return RN_E_STATE;
        }
    }
    const size_t total = static_cast<size_t>(h->last_n) * node_elems(nb);
    if (n_elems) *n_elems = total;
    if (cap_elems < total) {
        rn_set_error("rn_tap: buffer too small (%zu < %zu elements)", cap_elems, total);
        return RN_E_RANGE;
    }
    DeviceGuard guard(h->device);
    RN_HIP(hipStreamSynchronize(h->stream));
    if (nb.dtype == RN_DTYPE_F32) {
        RN_HIP(hipMemcpy(out, nb.ptr, total * 4, hipMemcpyDeviceToHost));
    } else {
        float* tmp = nullptr;
        RN_HIP(hipMalloc(reinterpret_cast<void**>(&tmp), total * 4));
        int rc = rn_launch_convert_to_f32(h->stream, nb.ptr, nb.dtype, tmp, static_cast<int64_t>(total));
        if (rc == RN_OK && hipStreamSynchronize(h->stream) != hipSuccess) rc = RN_E_HIP;
        if (rc == RN_OK && hipMemcpy(out, tmp, total * 4, hipMemcpyDeviceToHost) != hipSuccess) rc = RN_E_HIP;
        (void)hipFree(tmp);
        if (rc != RN_OK) {
            if (rc == RN_E_HIP) rn_set_error("rn_tap: device copy failed");
            return rc;
        }
    }
    // a handle may store this tensor with its channels relabelled (frozen-channel folding, rn_fused_prepare): hand it out in the
    // reference's channel order
    {
        const int* perm = fused_mode(h) ? rn_fused_node_perm(h, node_id) : nullptr;
        if (!perm) {
            auto it = h->node_perm.find(node_id);
            if (it != h->node_perm.end()) perm = it->second.data();
        }
        if (perm) {
            const int c = nb.info.c;
            std::vector<float> px(static_cast<size_t>(c));
            for (size_t q = 0; q < total / static_cast<size_t>(c); ++q) {
                float* pix = out + q * c;
                for (int p = 0; p < c; ++p) px[perm[p]] = pix[p];
                std::memcpy(pix, px.data(), static_cast<size_t>(c) * 4);
            }
        }
    }
    return RN_OK;
}

extern "C" int rn_set_profiling(rn_handle* h, int enable) {
    if (!h) {
        rn_set_error("null handle");
        return RN_E_INVALID;
    }
    h->profiling = enable != 0;
    h->timing_valid = false;
    return RN_OK;
}

extern "C" int rn_timing(rn_handle* h, rn_stage_ms* out) {
    if (!h || !out) {
        rn_set_error("rn_timing: bad argument");
        return RN_E_INVALID;
    }
    if (!h->timing_valid) {
        rn_set_error("rn_timing: no profiled forward pass (call rn_set_profiling(h,1) first)");
        return RN_E_STATE;
    }
    DeviceGuard guard(h->device);
    const int ns = static_cast<int>(h->stages.size());
    RN_HIP(hipEventSynchronize(h->events[2 + ns]));
    std::memset(out, 0, sizeof(*out));
    out->n_stages = ns;
    RN_HIP(hipEventElapsedTime(&out->preprocess_ms, h->events[0], h->events[1]));
    for (int i = 0; i < ns; ++i) RN_HIP(hipEventElapsedTime(&out->stage_ms[i], h->events[1 + i], h->events[2 + i]));
    RN_HIP(hipEventElapsedTime(&out->head_ms, h->events[1 + ns], h->events[2 + ns]));
    RN_HIP(hipEventElapsedTime(&out->total_ms, h->events[0], h->events[2 + ns]));
    return RN_OK;
}

extern "C" int rn_dominant_stage(const rn_handle* h) {
    if (!h) return -1;
    int best = 0;
    double bestf = -1;
    for (size_t i = 0; i < h->stages.size(); ++i) {
        const StagePlan& s = h->stages[i];
        const double f = 2.0 * s.conv_side * s.conv_side * 9.0 * s.cin * s.cout;
        if (f > bestf) {
            bestf = f;
            best = static_cast<int>(i);
        }
    }
    return best;
}

extern "C" int rn_stage_launch(const rn_handle* h, int stage) {
    if (!h || stage < 0 || stage >= static_cast<int>(h->stages.size())) {
        rn_set_error("rn_stage_launch: bad argument");
        return RN_E_RANGE;
    }
    return fused_mode(h) ? rn_fused_launch_rep(h, stage) : stage;
}

extern "C" int rn_device_malloc(rn_handle* h, size_t bytes, void** d_ptr) {
    if (!h || !d_ptr) {
        rn_set_error("rn_device_malloc: bad argument");
        return RN_E_INVALID;
    }
    DeviceGuard guard(h->device);
    hipError_t e = hipMalloc(d_ptr, bytes ? bytes : 16);
    if (e != hipSuccess) {
        rn_set_error("hipMalloc(%zu) failed: %s", bytes, hipGetErrorString(e));
        return RN_E_NOMEM;
    }
    return RN_OK;
}

extern "C" int rn_device_free(rn_handle* h, void* d_ptr) {
    if (!h) {
        rn_set_error("null handle");
        return RN_E_INVALID;
    }
    DeviceGuard guard(h->device);
    RN_HIP(hipFree(d_ptr));
    return RN_OK;
}

extern "C" int rn_memcpy_h2d(rn_handle* h, void* d_dst, const void* src, size_t bytes) {
    if (!h || !d_dst || !src) {
        rn_set_error("rn_memcpy_h2d: bad argument");
        return RN_E_INVALID;
    }
    DeviceGuard guard(h->device);
    RN_HIP(hipMemcpyAsync(d_dst, src, bytes, hipMemcpyHostToDevice, h->stream));
    RN_HIP(hipStreamSynchronize(h->stream));
    return RN_OK;
}

extern "C" int rn_memcpy_d2h(rn_handle* h, void* dst, const void* d_src, size_t bytes) {
    if (!h || !dst || !d_src) {
        rn_set_error("rn_memcpy_d2h: bad argument");
        return RN_E_INVALID;
    }
    DeviceGuard guard(h->device);
    RN_HIP(hipMemcpyAsync(dst, d_src, bytes, hipMemcpyDeviceToHost, h->stream));
    RN_HIP(hipStreamSynchronize(h->stream));
    return RN_OK;
}

// ---- pinned host memory: what makes the host-buffer entries asynchronous (rn_submit_u8, rn_group_forward_u8)
extern "C" int rn_host_alloc(size_t bytes, void** ptr) {
    if (!ptr || bytes == 0) {
        rn_set_error("rn_host_alloc: bad argument");
        return RN_E_INVALID;
    }
    *ptr = nullptr;
    // portable: usable as a copy source for every device of the process (a group's shards come out of one buffer)
    hipError_t e = hipHostMalloc(ptr, bytes, hipHostMallocPortable);
    if (e != hipSuccess) {
        rn_set_error("rn_host_alloc: hipHostMalloc(%zu bytes) failed: %s", bytes, hipGetErrorString(e));
        (void)hipGetLastError();       // reported: it must not surface again as the next kernel's "launch failure"
        *ptr = nullptr;
        return RN_E_NOMEM;
    }
    return RN_OK;
}

extern "C" int rn_host_free(void* ptr) {
    if (!ptr) return RN_OK;
    RN_HIP(hipHostFree(ptr));
    return RN_OK;
}

extern "C" int rn_const_info(const rn_handle* h, int info[4]) {
    if (!h || !info) {
        rn_set_error("rn_const_info: bad argument");
        return RN_E_INVALID;
    }
    info[0] = -1;
    info[1] = info[2] = info[3] = 0;
    if (fused_mode(h)) rn_fused_const_info(h, info);
    return RN_OK;
}

extern "C" int rn_frozen_info(const rn_handle* h, int info[4]) {
    if (!h || !info) {
        rn_set_error("rn_frozen_info: bad argument");
        return RN_E_INVALID;
    }
    info[0] = info[1] = 0;
    info[2] = -1;
    info[3] = 4;
    if (fused_mode(h)) {
        rn_fused_frozen_info(h, info);
    } else {
        if (h->f32_fold_stage >= 0 && rn_f32m_covers(h, h->f32_fold_stage)) {
            info[2] = h->f32_fold_stage;
            info[3] = h->f32_fold_live / 16;
        }
        if (h->f32_kfold_stage >= 0 && rn_f32m_covers(h, h->f32_kfold_stage)) {
            info[0] = h->stages[h->f32_kfold_stage].cin - h->f32_kfold_live;
            info[1] = h->f32_kfold_proven;
        }
    }
    return RN_OK;
}
