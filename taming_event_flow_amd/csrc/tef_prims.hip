// tef_prims.hip — the reference's warping / IWE primitives (utils/iwe.py:17-40 get_event_flow, :63-113 get_interpolation,
// :116-136 interpolate) as stand-alone DIFFERENTIABLE operators for callers that build their own loss from them: batched
// forward of the flow lookup and the backward of all three.  (The training loss does not come through here: loss.flow
// runs the fused kernels of tef_loss.hip.)  One thread per event, global float atomics for the two scatters — the same
// formulation as ATen's grid_sampler_2d / scatter_add backward, which is what the reference's autograd runs.
#include <hip/hip_runtime.h>
#include <stdint.h>

#include "tef.h"
#include "tef_common.h"

namespace {

// taps of grid_sample(bilinear, align_corners=True, zeros) at (y, x): utils/iwe.py:30-35 + ATen grid_sampler_2d
struct Taps {
    int i00, i01, i10, i11;        // linear index of nw, ne, sw, se; -1 outside the frame
    float s, n, e, w;              // 1-D factors: rows (1 - n, n), columns (1 - w, w)
};

__device__ __forceinline__ float unnormalize(float v, int size)
{
    float nn = (2.0f * v) / (float)(size - 1) - 1.0f;
    return (nn + 1.0f) * ((float)(size - 1) / 2.0f);
}

__device__ __forceinline__ Taps make_taps(float y, float x, int H, int W)
{
    Taps t;
    const float iy = unnormalize(y, H), ix = unnormalize(x, W);
    const float fy0 = floorf(iy), fx0 = floorf(ix);
    t.n = iy - fy0; t.w = ix - fx0; t.s = 1.0f - t.n; t.e = 1.0f - t.w;
    // NaN / huge locations: every comparison below is false, the taps are all outside
    const int y0 = (fy0 >= -2.0f && fy0 <= (float)H) ? (int)fy0 : -2, x0 = (fx0 >= -2.0f && fx0 <= (float)W) ? (int)fx0 : -2;
    const int y1 = y0 + 1, x1 = x0 + 1;
    const bool vy0 = (y0 >= 0) & (y0 < H), vy1 = (y1 >= 0) & (y1 < H), vx0 = (x0 >= 0) & (x0 < W), vx1 = (x1 >= 0) & (x1 < W);
    t.i00 = (vy0 && vx0) ? y0 * W + x0 : -1;
    t.i01 = (vy0 && vx1) ? y0 * W + x1 : -1;
    t.i10 = (vy1 && vx0) ? y1 * W + x0 : -1;
    t.i11 = (vy1 && vx1) ? y1 * W + x1 : -1;
    return t;
}

__device__ __forceinline__ float tap(const float *__restrict__ m, int i) { return i >= 0 ? m[i] : 0.0f; }

__global__ __launch_bounds__(256) void event_flow_kernel(const float *__restrict__ fx, const float *__restrict__ fy, int B,
                                                         int H, int W, const float *__restrict__ loc, int N,
                                                         float *__restrict__ out)
{
    const size_t e = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (e >= (size_t)B * N) return;
    const size_t b = e / N;
    const float *mx = fx + b * H * W, *my = fy + b * H * W;
    const Taps t = make_taps(loc[2 * e], loc[2 * e + 1], H, W);
    const float w00 = t.s * t.e, w01 = t.s * t.w, w10 = t.n * t.e, w11 = t.n * t.w;
    // one product + three fused multiply-adds, like ATen's (contracted) CPU kernel: see tef_loss.hip::quad_value
    out[2 * e] = __builtin_fmaf(tap(my, t.i11), w11, __builtin_fmaf(tap(my, t.i10), w10, __builtin_fmaf(tap(my, t.i01), w01, tap(my, t.i00) * w00)));
    out[2 * e + 1] = __builtin_fmaf(tap(mx, t.i11), w11, __builtin_fmaf(tap(mx, t.i10), w10, __builtin_fmaf(tap(mx, t.i01), w01, tap(mx, t.i00) * w00)));
}

// gout [B][N][2] = (d / d f_y, d / d f_x).  dfx / dfy += bilinear weights x gout (zeroed by the caller);
// dloc = gout . Jacobian of the lookup (taps outside the frame read as 0, like the forward).
__global__ __launch_bounds__(256) void event_flow_bwd_kernel(const float *__restrict__ fx, const float *__restrict__ fy, int B,
                                                             int H, int W, const float *__restrict__ loc, int N,
                                                             const float *__restrict__ gout, float *__restrict__ dfx,
                                                             float *__restrict__ dfy, float *__restrict__ dloc)
{
    const size_t e = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (e >= (size_t)B * N) return;
    const size_t b = e / N;
    const Taps t = make_taps(loc[2 * e], loc[2 * e + 1], H, W);
    const float gy = gout[2 * e], gx = gout[2 * e + 1];
    if (dfx) {
        float *dx = dfx + b * H * W, *dy = dfy + b * H * W;
        const float w00 = t.s * t.e, w01 = t.s * t.w, w10 = t.n * t.e, w11 = t.n * t.w;
        if (t.i00 >= 0) { atomicAdd(dy + t.i00, gy * w00); atomicAdd(dx + t.i00, gx * w00); }
        if (t.i01 >= 0) { atomicAdd(dy + t.i01, gy * w01); atomicAdd(dx + t.i01, gx * w01); }
        if (t.i10 >= 0) { atomicAdd(dy + t.i10, gy * w10); atomicAdd(dx + t.i10, gx * w10); }
        if (t.i11 >= 0) { atomicAdd(dy + t.i11, gy * w11); atomicAdd(dx + t.i11, gx * w11); }
    }
    if (dloc) {
        const float *mx = fx + b * H * W, *my = fy + b * H * W;
        const float y00 = tap(my, t.i00), y01 = tap(my, t.i01), y10 = tap(my, t.i10), y11 = tap(my, t.i11);
        const float x00 = tap(mx, t.i00), x01 = tap(mx, t.i01), x10 = tap(mx, t.i10), x11 = tap(mx, t.i11);
        const float jyy = (y10 - y00) * t.e + (y11 - y01) * t.w, jyx = (y01 - y00) * t.s + (y11 - y10) * t.n;   // d f_y / d (y, x)
        const float jxy = (x10 - x00) * t.e + (x11 - x01) * t.w, jxx = (x01 - x00) * t.s + (x11 - x10) * t.n;   // d f_x / d (y, x)
        dloc[2 * e] = gy * jyy + gx * jxy;
        dloc[2 * e + 1] = gy * jyx + gx * jxx;
    }
}

// autograd through torch.max(zeros, 1 - |d|): slope -sign(d) inside the hat, half of it exactly at the tie, abs'(0) = 0
__device__ __forceinline__ float hat(float d, float &slope)
{
    const float v = 1.0f - fabsf(d);
    const float sg = d > 0.0f ? 1.0f : (d < 0.0f ? -1.0f : 0.0f);
    slope = v > 0.0f ? -sg : (v == 0.0f ? -0.5f * sg : 0.0f);
    return fmaxf(v, 0.0f);
}

// gw [B][4n] (corner blocks TL, TR, BL, BR) -> dloc [B][n][2]: weights = prod of the two hats, zero for corners outside
__global__ __launch_bounds__(256) void interp_corners_bwd_kernel(const float *__restrict__ loc, int B, int n, int H, int W,
                                                                 const float *__restrict__ gw, float *__restrict__ dloc)
{
    const size_t e = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (e >= (size_t)B * n) return;
    const size_t b = e / n, k = e - b * n;
    const float y = loc[2 * e], x = loc[2 * e + 1];
    const float cy[2] = {floorf(y), floorf(y + 1.0f)}, cx[2] = {floorf(x), floorf(x + 1.0f)};
    float gy = 0.0f, gx = 0.0f;
#pragma unroll
    for (int c = 0; c < 4; ++c) {
        const float iy = cy[c >> 1], ix = cx[c & 1];
        if (!(iy >= 0.0f && iy < (float)H && ix >= 0.0f && ix < (float)W)) continue;
        float sy, sx;
        const float wy = hat(y - iy, sy), wx = hat(x - ix, sx);
        const float r = gw[(b * 4 + c) * n + k];
        gy += r * (sy * wx);
        gx += r * (wy * sx);
    }
    dloc[2 * e] = gy;
    dloc[2 * e + 1] = gx;
}

// gimg [B][HW] -> dweights = gimg[idx] * mask, dmask = gimg[idx] * weights (either may be null)
__global__ __launch_bounds__(256) void scatter_add_bwd_kernel(const float *__restrict__ idx, const float *__restrict__ wgt,
                                                              const float *__restrict__ mask, int B, int n, int HW,
                                                              const float *__restrict__ gimg, float *__restrict__ dw,
                                                              float *__restrict__ dmask)
{
    const size_t e = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (e >= (size_t)B * n) return;
    const size_t b = e / n;
    const long p = (long)idx[e];
    const float g = (p >= 0 && p < HW) ? gimg[b * HW + p] : 0.0f;
    if (dw) dw[e] = mask ? g * mask[e] : g;
    if (dmask) dmask[e] = g * wgt[e];
}

inline unsigned nblk(size_t n) { return (unsigned)((n + 255) / 256); }

}  // namespace

extern "C" {

int tef_event_flow(const float *fx, const float *fy, int B, int H, int W, const float *loc, int N, float *out, void *stream)
{
    if (!fx || !fy || !loc || !out || B < 1 || H < 2 || W < 2 || N < 0)
        return tef::fail("tef_event_flow: bad arguments"), TEF_ERR_INVALID;
    if (N == 0) return 0;
    hipLaunchKernelGGL(event_flow_kernel, dim3(nblk((size_t)B * N)), dim3(256), 0, (hipStream_t)stream, fx, fy, B, H, W, loc, N, out);
    return tef::check_launch("event_flow_kernel");
}

int tef_event_flow_backward(const float *fx, const float *fy, int B, int H, int W, const float *loc, int N, const float *gout,
                            float *dfx, float *dfy, float *dloc, void *stream)
{
    if (!fx || !fy || !loc || !gout || B < 1 || H < 2 || W < 2 || N < 0 || ((dfx == nullptr) != (dfy == nullptr)))
        return tef::fail("tef_event_flow_backward: bad arguments (dfx and dfy go together)"), TEF_ERR_INVALID;
    if (N == 0 || (!dfx && !dloc)) return 0;
    hipLaunchKernelGGL(event_flow_bwd_kernel, dim3(nblk((size_t)B * N)), dim3(256), 0, (hipStream_t)stream, fx, fy, B, H, W, loc, N,
                       gout, dfx, dfy, dloc);
    return tef::check_launch("event_flow_bwd_kernel");
}

int tef_interp_corners_backward(const float *loc, int B, int n, int H, int W, const float *gweights, float *dloc, void *stream)
{
    if (!loc || !gweights || !dloc || B < 1 || n < 0 || H < 1 || W < 1)
        return tef::fail("tef_interp_corners_backward: bad arguments"), TEF_ERR_INVALID;
    if (n == 0) return 0;
    hipLaunchKernelGGL(interp_corners_bwd_kernel, dim3(nblk((size_t)B * n)), dim3(256), 0, (hipStream_t)stream, loc, B, n, H, W,
                       gweights, dloc);
    return tef::check_launch("interp_corners_bwd_kernel");
}

int tef_scatter_add_backward(const float *idx, const float *weights, const float *mask, int B, int n, int HW, const float *gimg,
                             float *dweights, float *dmask, void *stream)
{
    if (!idx || !gimg || B < 1 || n < 0 || HW < 1 || (dmask && !weights))
        return tef::fail("tef_scatter_add_backward: bad arguments"), TEF_ERR_INVALID;
    if (n == 0 || (!dweights && !dmask)) return 0;
    hipLaunchKernelGGL(scatter_add_bwd_kernel, dim3(nblk((size_t)B * n)), dim3(256), 0, (hipStream_t)stream, idx, weights, mask, B, n,
                       HW, gimg, dweights, dmask);
    return tef::check_launch("scatter_add_bwd_kernel");
}

}  // extern "C"
