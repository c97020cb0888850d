// tef_resize.hip — bilinear up-sampling by an integer factor, align_corners=False, forward and backward.
// Reference: torch.nn.functional.interpolate(mode="bilinear", align_corners=False) as called by
// models/submodules.py:264 (decoder features, x2) and models/model.py:79 (flow heads, x8/x4/x2/x1, times a scalar).
// ATen semantics (UpSample.h area_pixel_compute_source_index): src = max((dst + 0.5) / s - 0.5, 0),
// i0 = floor(src), i1 = i0 + (i0 < n - 1), weights (1 - l, l) with l = src - i0.
#include <hip/hip_runtime.h>
#include <stdint.h>

#include "tef.h"
#include "tef_common.h"

namespace {

__device__ __forceinline__ void src_index(int o, float inv, int n, int &i0, int &i1, float &l0, float &l1)
{
    float s = inv * ((float)o + 0.5f) - 0.5f;
    s = s < 0.0f ? 0.0f : s;
    i0 = (int)s;
    i1 = i0 + (i0 < n - 1 ? 1 : 0);
    l1 = s - (float)i0;
    l0 = 1.0f - l1;
}

// y [planes][Ho - ct][Wo - cl] = mul * upsample(x + x2)[ct:, cl:]   (x2 may be null; ct / cl = rows / columns cropped at
// the top / left: RecEVFlowNet pads its input there, models/model_util.py:52-65, and crops the flows again, model.py:83)
__global__ __launch_bounds__(256) void upsample_fwd_kernel(const float *__restrict__ x, const float *__restrict__ x2,
                                                           int planes, int H, int W, int sh, int sw, float mul, int ct,
                                                           int cl, float *__restrict__ y)
{
    const int Hc = H * sh - ct, Wc = W * sw - cl;
    size_t idx = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (idx >= (size_t)planes * Hc * Wc) return;
    int ox = (int)(idx % Wc) + cl;
    size_t t = idx / Wc;
    int oy = (int)(t % Hc) + ct, pl = (int)(t / Hc);
    int y0, y1, x0, x1;
    float ly0, ly1, lx0, lx1;
    src_index(oy, 1.0f / (float)sh, H, y0, y1, ly0, ly1);
    src_index(ox, 1.0f / (float)sw, W, x0, x1, lx0, lx1);
    const float *p = x + (size_t)pl * H * W;
    float v00 = p[y0 * W + x0], v01 = p[y0 * W + x1], v10 = p[y1 * W + x0], v11 = p[y1 * W + x1];
    if (x2) {
        const float *q = x2 + (size_t)pl * H * W;
        v00 += q[y0 * W + x0]; v01 += q[y0 * W + x1]; v10 += q[y1 * W + x0]; v11 += q[y1 * W + x1];
    }
    float v = ly0 * (lx0 * v00 + lx1 * v01) + ly1 * (lx0 * v10 + lx1 * v11);
    y[idx] = mul * v;
}

// sh = sw = 2 without crop, W a multiple of 2 (the decoder levels): one thread per 4 consecutive outputs of a row — they
// read input columns 2j - 1 .. 2j + 2 of two input rows (8 values instead of 16) and leave as one 16-byte store.  The
// per-output expression is the one above (same weights from src_index, same order of operations).
__device__ __forceinline__ void upsample2x_fwd4_item(const float *__restrict__ x, const float *__restrict__ x2, size_t idx, int H,
                                                      int W, float mul, float *__restrict__ y)
{
    const int Ho = 2 * H, Wo = 2 * W, W4 = Wo >> 2;
    const int j = (int)(idx % W4);
    const size_t t = idx / W4;
    const int oy = (int)(t % Ho), pl = (int)(t / Ho);
    int y0, y1;
    float ly0, ly1;
    src_index(oy, 0.5f, H, y0, y1, ly0, ly1);
    const float *p = x + (size_t)pl * H * W, *q = x2 ? x2 + (size_t)pl * H * W : nullptr;
    // input columns the four outputs can touch: 2j - 1 .. 2j + 2, clamped like src_index does
    float r0[4], r1[4];
#pragma unroll
    for (int c = 0; c < 4; ++c) {
        const int xi = min(max(2 * j - 1 + c, 0), W - 1);
        r0[c] = p[y0 * W + xi];
        r1[c] = p[y1 * W + xi];
        if (q) {
            r0[c] += q[y0 * W + xi];
            r1[c] += q[y1 * W + xi];
        }
    }
    float o[4];
#pragma unroll
    for (int e = 0; e < 4; ++e) {
        int x0, x1;
        float lx0, lx1;
        src_index(4 * j + e, 0.5f, W, x0, x1, lx0, lx1);
        const int a = x0 - (2 * j - 1), b = x1 - (2 * j - 1);          // positions in r0 / r1 (0 .. 3)
        const float v00 = a == 0 ? r0[0] : (a == 1 ? r0[1] : (a == 2 ? r0[2] : r0[3]));
        const float v01 = b == 0 ? r0[0] : (b == 1 ? r0[1] : (b == 2 ? r0[2] : r0[3]));
        const float v10 = a == 0 ? r1[0] : (a == 1 ? r1[1] : (a == 2 ? r1[2] : r1[3]));
        const float v11 = b == 0 ? r1[0] : (b == 1 ? r1[1] : (b == 2 ? r1[2] : r1[3]));
        o[e] = mul * (ly0 * (lx0 * v00 + lx1 * v01) + ly1 * (lx0 * v10 + lx1 * v11));
    }
    *reinterpret_cast<float4 *>(y + ((size_t)pl * Ho + oy) * Wo + 4 * j) = make_float4(o[0], o[1], o[2], o[3]);
}
__global__ __launch_bounds__(256) void upsample2x_fwd4_kernel(const float *__restrict__ x, const float *__restrict__ x2,
                                                              int planes, int H, int W, float mul, float *__restrict__ y)
{
    const size_t idx = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (idx >= (size_t)planes * (2 * H) * (W >> 1)) return;
    upsample2x_fwd4_item(x, x2, idx, H, W, mul, y);
}
// two tensors of one geometry in one launch (a decoder level of RecEVFlowNet: features + skip, and the previous prediction)
__global__ __launch_bounds__(256) void upsample2x_fwd4_pair_kernel(const float *__restrict__ xa, const float *__restrict__ xa2,
                                                                   int planes_a, float *__restrict__ ya,
                                                                   const float *__restrict__ xb, int planes_b,
                                                                   float *__restrict__ yb, int H, int W)
{
    const size_t idx = (size_t)blockIdx.x * blockDim.x + threadIdx.x, na = (size_t)planes_a * (2 * H) * (W >> 1);
    if (idx < na) upsample2x_fwd4_item(xa, xa2, idx, H, W, 1.0f, ya);
    else if (idx - na < (size_t)planes_b * (2 * H) * (W >> 1)) upsample2x_fwd4_item(xb, nullptr, idx - na, H, W, 1.0f, yb);
}

// Exact adjoint, separable: dX = Ry^T dY Rx with the forward's 1-D weights (same expressions as the forward, so the pair
// is an exact adjoint).  One workgroup per (plane, input row): first the column sums over the <= 3*sh output rows that
// touch the input row (T[ox], kept in LDS), then every input pixel of the row collects its <= 3*sw columns of T.
__device__ __forceinline__ float tap_weight(int o, float inv, int n, int i)
{
    int i0, i1;
    float l0, l1;
    src_index(o, inv, n, i0, i1, l0, l1);
    return (i0 == i ? l0 : 0.0f) + (i1 == i ? l1 : 0.0f);
}

// dy is the cropped tensor [planes][Ho - ct][Wo - cl]: cropped rows / columns carry no gradient
__global__ __launch_bounds__(128) void upsample_bwd_kernel(const float *__restrict__ dy, int planes, int H, int W,
                                                           int sh, int sw, float mul, int ct, int cl,
                                                           float *__restrict__ dx)
{
    extern __shared__ float trow[];                 // [Wo]
    const int Ho = H * sh, Wo = W * sw, Wc = Wo - cl;
    const int iy = blockIdx.x % H, pl = blockIdx.x / H;
    const float *g = dy + (size_t)pl * (Ho - ct) * Wc;
    const float invh = 1.0f / (float)sh, invw = 1.0f / (float)sw;
    const int oy_lo = max(ct, (iy - 1) * sh), oy_hi = min(Ho, (iy + 2) * sh);
    for (int ox = threadIdx.x; ox < Wo; ox += blockDim.x) {
        float acc = 0.0f;
        if (ox >= cl)
            for (int oy = oy_lo; oy < oy_hi; ++oy) {
                float wy = tap_weight(oy, invh, H, iy);     // uniform across the workgroup
                if (wy != 0.0f) acc += wy * g[(size_t)(oy - ct) * Wc + (ox - cl)];
            }
        trow[ox] = acc;
    }
    __syncthreads();
    for (int ix = threadIdx.x; ix < W; ix += blockDim.x) {
        const int ox_lo = max(0, (ix - 1) * sw), ox_hi = min(Wo, (ix + 2) * sw);
        float acc = 0.0f;
        for (int ox = ox_lo; ox < ox_hi; ++ox) {
            float wx = tap_weight(ox, invw, W, ix);
            if (wx != 0.0f) acc += wx * trow[ox];
        }
        dx[((size_t)pl * H + iy) * W + ix] = mul * acc;
    }
}

// The same adjoint for sh = sw = 2 (the four decoder levels, 60 of the 61 MB a pass up-samples), one thread per INPUT
// pixel: only the output rows / columns 2i - 1 .. 2i + 2 can carry a weight for input index i, so a thread sums its 4 x 4
// neighbourhood — vertical sums first, then the horizontal one, in ascending order: the same additions in the same
// order as the row kernel above.  (That one launches a 128-thread workgroup per input row: at the deep levels, rows of
// 8 and 16 pixels, 22 us for 1 - 2 MB.)
__device__ __forceinline__ void upsample2x_bwd_item(const float *__restrict__ dy, size_t idx, int H, int W, float mul, int ct, int cl,
                                                     float *__restrict__ dx)
{
    const int Ho = 2 * H, Wo = 2 * W, Wc = Wo - cl;
    const int ix = (int)(idx % W);
    const size_t t = idx / W;
    const int iy = (int)(t % H), pl = (int)(t / H);
    const float *g = dy + (size_t)pl * (Ho - ct) * Wc;
    float wy[4];
#pragma unroll
    for (int a = 0; a < 4; ++a) {
        const int oy = 2 * iy - 1 + a;
        wy[a] = (oy >= ct && oy < Ho) ? tap_weight(oy, 0.5f, H, iy) : 0.0f;
    }
    float acc = 0.0f;
#pragma unroll
    for (int b = 0; b < 4; ++b) {
        const int ox = 2 * ix - 1 + b;
        if (ox < cl || ox >= Wo) continue;
        const float wx = tap_weight(ox, 0.5f, W, ix);
        if (wx == 0.0f) continue;
        float col = 0.0f;
#pragma unroll
        for (int a = 0; a < 4; ++a)
            if (wy[a] != 0.0f) col += wy[a] * g[(size_t)(2 * iy - 1 + a - ct) * Wc + (ox - cl)];
        acc += wx * col;
    }
    dx[idx] = mul * acc;
}
__global__ __launch_bounds__(256) void upsample2x_bwd_kernel(const float *__restrict__ dy, int planes, int H, int W,
                                                             float mul, int ct, int cl, float *__restrict__ dx)
{
    const size_t idx = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (idx >= (size_t)planes * H * W) return;
    upsample2x_bwd_item(dy, idx, H, W, mul, ct, cl, dx);
}
__global__ __launch_bounds__(256) void upsample2x_bwd_pair_kernel(const float *__restrict__ dya, int planes_a, float *__restrict__ dxa,
                                                                  const float *__restrict__ dyb, int planes_b,
                                                                  float *__restrict__ dxb, int H, int W)
{
    const size_t idx = (size_t)blockIdx.x * blockDim.x + threadIdx.x, na = (size_t)planes_a * H * W;
    if (idx < na) upsample2x_bwd_item(dya, idx, H, W, 1.0f, 0, 0, dxa);
    else if (idx - na < (size_t)planes_b * H * W) upsample2x_bwd_item(dyb, idx - na, H, W, 1.0f, 0, 0, dxb);
}

}  // namespace

extern "C" {

int tef_upsample_bilinear_crop(const float *x, const float *x2, int planes, int H, int W, int scale_h, int scale_w,
                               float mul, int crop_top, int crop_left, float *y, void *stream)
{
    if (!x || !y || planes < 1 || H < 1 || W < 1 || scale_h < 1 || scale_w < 1 || crop_top < 0 || crop_left < 0 ||
        crop_top >= H * scale_h || crop_left >= W * scale_w)
        return tef::fail("tef_upsample_bilinear: bad arguments"), TEF_ERR_INVALID;
    if (scale_h == 2 && scale_w == 2 && crop_top == 0 && crop_left == 0 && (W & 1) == 0 && W >= 2) {
        const size_t n4 = (size_t)planes * (2 * H) * (W >> 1);
        hipLaunchKernelGGL(upsample2x_fwd4_kernel, dim3((unsigned)((n4 + 255) / 256)), dim3(256), 0, (hipStream_t)stream, x, x2,
                           planes, H, W, mul, y);
        return tef::check_launch("upsample2x_fwd4_kernel");
    }
    size_t n = (size_t)planes * (H * scale_h - crop_top) * (W * scale_w - crop_left);
    hipLaunchKernelGGL(upsample_fwd_kernel, dim3((unsigned)((n + 255) / 256)), dim3(256), 0, (hipStream_t)stream, x, x2,
                       planes, H, W, scale_h, scale_w, mul, crop_top, crop_left, y);
    return tef::check_launch("upsample_fwd_kernel");
}

int tef_upsample_bilinear_crop_backward(const float *dy, int planes, int H, int W, int scale_h, int scale_w, float mul,
                                        int crop_top, int crop_left, float *dx, void *stream)
{
    if (!dy || !dx || planes < 1 || H < 1 || W < 1 || scale_h < 1 || scale_w < 1 || crop_top < 0 || crop_left < 0 ||
        crop_top >= H * scale_h || crop_left >= W * scale_w)
        return tef::fail("tef_upsample_bilinear_backward: bad arguments"), TEF_ERR_INVALID;
    if (scale_h == 2 && scale_w == 2) {
        const size_t n = (size_t)planes * H * W;
        hipLaunchKernelGGL(upsample2x_bwd_kernel, dim3((unsigned)((n + 255) / 256)), dim3(256), 0, (hipStream_t)stream, dy,
                           planes, H, W, mul, crop_top, crop_left, dx);
        return tef::check_launch("upsample2x_bwd_kernel");
    }
    size_t lds = (size_t)W * scale_w * sizeof(float);
    if (lds > 64 * 1024) return tef::fail("tef_upsample_bilinear_backward: output row too wide"), TEF_ERR_INVALID;
    hipLaunchKernelGGL(upsample_bwd_kernel, dim3((unsigned)((size_t)planes * H)), dim3(128), lds, (hipStream_t)stream, dy,
                       planes, H, W, scale_h, scale_w, mul, crop_top, crop_left, dx);
    return tef::check_launch("upsample_bwd_kernel");
}

int tef_upsample2x_pair(const float *xa, const float *xa2, int planes_a, float *ya, const float *xb, int planes_b, float *yb,
                        int H, int W, void *stream)
{
    if (!xa || !ya || !xb || !yb || planes_a < 1 || planes_b < 1 || H < 1 || W < 2 || (W & 1))
        return tef::fail("tef_upsample2x_pair: bad arguments (even width)"), TEF_ERR_INVALID;
    const size_t n4 = (size_t)(planes_a + planes_b) * (2 * H) * (W >> 1);
    hipLaunchKernelGGL(upsample2x_fwd4_pair_kernel, dim3((unsigned)((n4 + 255) / 256)), dim3(256), 0, (hipStream_t)stream, xa, xa2,
                       planes_a, ya, xb, planes_b, yb, H, W);
    return tef::check_launch("upsample2x_fwd4_pair_kernel");
}

int tef_upsample2x_pair_backward(const float *dya, int planes_a, float *dxa, const float *dyb, int planes_b, float *dxb, int H,
                                 int W, void *stream)
{
    if (!dya || !dxa || !dyb || !dxb || planes_a < 1 || planes_b < 1 || H < 1 || W < 1)
        return tef::fail("tef_upsample2x_pair_backward: bad arguments"), TEF_ERR_INVALID;
    const size_t n = (size_t)(planes_a + planes_b) * H * W;
    hipLaunchKernelGGL(upsample2x_bwd_pair_kernel, dim3((unsigned)((n + 255) / 256)), dim3(256), 0, (hipStream_t)stream, dya,
                       planes_a, dxa, dyb, planes_b, dxb, H, W);
    return tef::check_launch("upsample2x_bwd_pair_kernel");
}

int tef_upsample_bilinear(const float *x, int planes, int H, int W, int scale_h, int scale_w, float mul, float *y,
                          void *stream)
{
    return tef_upsample_bilinear_crop(x, nullptr, planes, H, W, scale_h, scale_w, mul, 0, 0, y, stream);
}

int tef_upsample_bilinear_backward(const float *dy, int planes, int H, int W, int scale_h, int scale_w, float mul,
                                   float *dx, void *stream)
{
    return tef_upsample_bilinear_crop_backward(dy, planes, H, W, scale_h, scale_w, mul, 0, 0, dx, stream);
}

}
