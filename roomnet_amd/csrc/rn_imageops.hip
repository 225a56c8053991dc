// GPU side of the caller's image pipeline: the crop + cv2.resize(INTER_LINEAR) of
// RoomNet.infer_optimized (reference network.py:137-156) for uint8 HWC images, so that a directory of
// arbitrarily sized photographs reaches the forward pass without a host resize.
//
// Integer-exact restatement of the published OpenCV algorithm (imgproc/resize.cpp), the same one
// roomnet_amd/imageops.py restates on the host and the parity tests compare against bit for bit:
// half-pixel-centre source coordinates in float32, 11-bit fixed-point coefficients, horizontal pass
// into 32-bit values, vertical pass (((b0 * (S0 >> 4)) >> 16) + ((b1 * (S1 >> 4)) >> 16) + 2) >> 2, and the
// special case that an exact 2x2 down-scale is a 2x2 box average.  HBM-bound byte work: one thread
// per destination pixel, 12 source bytes in, 3 bytes out; no LDS, no MFMA.  A batch of images is ONE launch
// (resize_batch_u8_kernel: blockIdx.z = image, a device table of crop windows); profiles/r6_imageops.*.
#include "rn_internal.h"

#include <cmath>
#include <cstdint>
#include <limits>

namespace {

struct ResizeArgs {
    const uint8_t* src;     // first pixel of the crop window
    uint8_t* dst;           // [dst_h, dst_w, 3]
    int src_h, src_w;       // crop window size
    int64_t src_row_bytes;  // bytes between source rows (full image width * 3)
    int dst_h, dst_w;
    double scale_x, scale_y;   // src / dst, computed on the host in double like cv2 does
    int mode;                  // 0 = bilinear, 1 = copy (same size), 2 = 2x2 box
};

__device__ __forceinline__ void lin_coeff(int d, double scale, int ssize, bool clamp_frac, int& s, int& c0, int& c1) {
    float f = static_cast<float>((static_cast<double>(d) + 0.5) * scale - 0.5);
    int si = static_cast<int>(floorf(f));
    f -= static_cast<float>(si);
    if (clamp_frac) {
        if (si < 0) { f = 0.f; si = 0; }
        if (si >= ssize - 1) { f = 0.f; si = ssize - 1; }
    }
    // cvRound = round-half-to-even of the float product, saturated to short
    int a0 = __float2int_rn((1.0f - f) * 2048.0f), a1 = __float2int_rn(f * 2048.0f);
    a0 = min(max(a0, -32768), 32767);
    a1 = min(max(a1, -32768), 32767);
    s = si; c0 = a0; c1 = a1;
}

__device__ __forceinline__ void resize_pixel(const ResizeArgs& a, int x, int y) {
    uint8_t* o = a.dst + (static_cast<int64_t>(y) * a.dst_w + x) * 3;
    if (a.mode == 1) {
        const uint8_t* p = a.src + static_cast<int64_t>(y) * a.src_row_bytes + x * 3;
        o[0] = p[0]; o[1] = p[1]; o[2] = p[2];
        return;
    }
    if (a.mode == 2) {
        const uint8_t* p0 = a.src + static_cast<int64_t>(2 * y) * a.src_row_bytes + 2 * x * 3;
        const uint8_t* p1 = p0 + a.src_row_bytes;
#pragma unroll
        for (int c = 0; c < 3; ++c) o[c] = static_cast<uint8_t>((p0[c] + p0[3 + c] + p1[c] + p1[3 + c] + 2) >> 2);
        return;
    }
    int sx, a0, a1, sy, b0, b1;
    lin_coeff(x, a.scale_x, a.src_w, true, sx, a0, a1);
    lin_coeff(y, a.scale_y, a.src_h, false, sy, b0, b1);
    const int sx1 = min(sx + 1, a.src_w - 1);
    const int y0 = min(max(sy, 0), a.src_h - 1), y1 = min(max(sy + 1, 0), a.src_h - 1);
    const uint8_t* r0 = a.src + static_cast<int64_t>(y0) * a.src_row_bytes;
    const uint8_t* r1 = a.src + static_cast<int64_t>(y1) * a.src_row_bytes;
#pragma unroll
    for (int c = 0; c < 3; ++c) {
        const int h0 = r0[sx * 3 + c] * a0 + r0[sx1 * 3 + c] * a1;     // horizontal pass, 32-bit
        const int h1 = r1[sx * 3 + c] * a0 + r1[sx1 * 3 + c] * a1;
        // arithmetic shifts on signed values, as the reference's int arithmetic (coefficients can be negative
        // only through the vertical extrapolation at the borders)
        const int v = (((b0 * (h0 >> 4)) >> 16) + ((b1 * (h1 >> 4)) >> 16) + 2) >> 2;
        o[c] = static_cast<uint8_t>(min(max(v, 0), 255));
    }
}

__global__ __launch_bounds__(256) void resize_linear_u8_kernel(const ResizeArgs a) {
    const int x = blockIdx.x * 64 + (threadIdx.x & 63);
    const int y = blockIdx.y * 4 + (threadIdx.x >> 6);
    if (x >= a.dst_w || y >= a.dst_h) return;
    resize_pixel(a, x, y);
}

// The same per-pixel arithmetic for a BATCH of images in one launch (blockIdx.z = image): sources of any size, one
// destination slot of side S each.  One launch per image left a 224 x 224 output on 196 workgroups -- less than the chip --
// and paid a launch per image: 256 images took 256 launches in front of a 1.1 ms forward pass.
__global__ __launch_bounds__(256) void resize_batch_u8_kernel(const rn_resize_item* items, uint8_t* dst_base, int S) {
    const int x = blockIdx.x * 64 + (threadIdx.x & 63);
    const int y = blockIdx.y * 4 + (threadIdx.x >> 6);
    if (x >= S || y >= S) return;
    const rn_resize_item it = items[blockIdx.z];
    ResizeArgs a;
    a.src = it.src;
    a.dst = dst_base + static_cast<int64_t>(blockIdx.z) * S * S * 3;
    a.src_h = it.src_h;
    a.src_w = it.src_w;
    a.src_row_bytes = it.src_row_bytes;
    a.dst_h = a.dst_w = S;
    a.scale_x = it.scale_x;
    a.scale_y = it.scale_y;
    a.mode = it.mode;
    resize_pixel(a, x, y);
}

}  // namespace

// host side of one batch item: the crop window [src_h, src_w] at d_src (rows src_row_bytes apart) -> S x S
void rn_resize_item_fill(rn_resize_item* it, const uint8_t* d_src, int src_h, int src_w, int64_t src_row_bytes, int S) {
    it->src = d_src;
    it->src_h = src_h;
    it->src_w = src_w;
    it->src_row_bytes = src_row_bytes;
    it->scale_x = 1.0 / (static_cast<double>(S) / static_cast<double>(src_w));
    it->scale_y = 1.0 / (static_cast<double>(S) / static_cast<double>(src_h));
    const double eps = std::numeric_limits<double>::epsilon();
    it->mode = (src_h == S && src_w == S) ? 1 : ((std::fabs(it->scale_x - 2.0) < eps && std::fabs(it->scale_y - 2.0) < eps) ? 2 : 0);
}

int rn_launch_resize_batch_u8(hipStream_t s, const rn_resize_item* d_items, int n, uint8_t* d_dst_base, int S) {
    if (n < 1 || S < 1) {
        rn_set_error("resize: empty batch (%d images -> %dx%d)", n, S, S);
        return RN_E_INVALID;
    }
    dim3 grid((S + 63) / 64, (S + 3) / 4, n);
    hipLaunchKernelGGL(resize_batch_u8_kernel, grid, dim3(256), 0, s, d_items, d_dst_base, S);
    RN_CHECK_LAUNCH();
    return RN_OK;
}

int rn_launch_resize_u8(hipStream_t s, const uint8_t* d_src, int src_h, int src_w, int64_t src_row_bytes, uint8_t* d_dst,
                        int dst_h, int dst_w) {
    if (src_h < 1 || src_w < 1 || dst_h < 1 || dst_w < 1) {
        rn_set_error("resize: empty image (%dx%d -> %dx%d)", src_w, src_h, dst_w, dst_h);
        return RN_E_INVALID;
    }
    ResizeArgs a{};
    a.src = d_src;
    a.dst = d_dst;
    a.src_h = src_h;
    a.src_w = src_w;
    a.src_row_bytes = src_row_bytes;
    a.dst_h = dst_h;
    a.dst_w = dst_w;
    // inv_scale = dsize / ssize, scale = 1 / inv_scale, in double (cv::resize)
    a.scale_x = 1.0 / (static_cast<double>(dst_w) / static_cast<double>(src_w));
    a.scale_y = 1.0 / (static_cast<double>(dst_h) / static_cast<double>(src_h));
    const double eps = std::numeric_limits<double>::epsilon();
    a.mode = (src_h == dst_h && src_w == dst_w) ? 1 : ((std::fabs(a.scale_x - 2.0) < eps && std::fabs(a.scale_y - 2.0) < eps) ? 2 : 0);
    dim3 grid((dst_w + 63) / 64, (dst_h + 3) / 4);
    hipLaunchKernelGGL(resize_linear_u8_kernel, grid, dim3(256), 0, s, a);
    RN_CHECK_LAUNCH();
    return RN_OK;
}
