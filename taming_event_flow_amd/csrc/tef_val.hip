// tef_val.hip — validation metrics of the reference's loss/flow_val.py (FWL, RSAT, AEE, windowed event / IWE / flow
// images, forward-propagated and accumulated flow).  Evaluation only (batch 1, no gradients): plain one-thread-per-
// element kernels with global float atomics for the scatters; the training hot path lives in tef_loss.hip.
#include <hip/hip_runtime.h>
#include <stdint.h>

#include "tef.h"
#include "tef_common.h"

namespace {

// ---- bilinear flow lookup on planar maps: utils/iwe.py:17-40 + ATen grid_sampler_2d (align_corners=True, zeros) ----
__device__ __forceinline__ float unnormalize(float v, int size)
{
    float nn = (2.0f * v) / (float)(size - 1) - 1.0f;
    return (nn + 1.0f) * ((float)(size - 1) / 2.0f);
}

__device__ __forceinline__ void sample_flow(const float *__restrict__ fx, const float *__restrict__ fy, int H, int W,
                                            float y, float x, float &oy, float &ox)
{
    float iy = unnormalize(y, H), ix = unnormalize(x, W);
    float fy0 = floorf(iy), fx0 = floorf(ix);
    float n = iy - fy0, w = ix - fx0, s = 1.0f - n, e = 1.0f - w;
    int y0 = (int)fy0, x0 = (int)fx0, y1 = y0 + 1, x1 = x0 + 1;
    bool vy0 = (y0 >= 0) & (y0 < H), vy1 = (y1 >= 0) & (y1 < H), vx0 = (x0 >= 0) & (x0 < W), vx1 = (x1 >= 0) & (x1 < W);
    float v00y = 0, v01y = 0, v10y = 0, v11y = 0, v00x = 0, v01x = 0, v10x = 0, v11x = 0;
    if (vy0 && vx0) { v00y = fy[y0 * W + x0]; v00x = fx[y0 * W + x0]; }
    if (vy0 && vx1) { v01y = fy[y0 * W + x1]; v01x = fx[y0 * W + x1]; }
    if (vy1 && vx0) { v10y = fy[y1 * W + x0]; v10x = fx[y1 * W + x0]; }
    if (vy1 && vx1) { v11y = fy[y1 * W + x1]; v11x = fx[y1 * W + x1]; }
    // one product + three fused multiply-adds, like ATen's (contracted) CPU kernel: see tef_loss.hip::quad_value
    oy = __builtin_fmaf(v11y, n * w, __builtin_fmaf(v10y, n * e, __builtin_fmaf(v01y, s * w, v00y * (s * e))));
    ox = __builtin_fmaf(v11x, n * w, __builtin_fmaf(v10x, n * e, __builtin_fmaf(v01x, s * w, v00x * (s * e))));
}

__device__ __forceinline__ bool inbounds(float y, float x, int H, int W)   // utils/iwe.py:52-57
{
    return (y >= 0.0f) & (y <= (float)H - 1.0f) & (x >= 0.0f) & (x <= (float)W - 1.0f);
}

// One warping step of an event list (loss/flow_val.py:337-342 sampling only; :492-517 forward step; :528-556 backward
// steps): flow lookup at loc, loc += (tref - ts) * flow, purge (loc and mask zeroed when out of bounds), ts = tref.
__global__ __launch_bounds__(256) void val_event_step_kernel(const float *__restrict__ fx, const float *__restrict__ fy,
                                                             int H, int W, float *__restrict__ loc,
                                                             float *__restrict__ ts, float *__restrict__ mask, int N,
                                                             float tref, int do_warp, float *__restrict__ flow_out)
{
    int e = blockIdx.x * blockDim.x + threadIdx.x;
    if (e >= N) return;
    float y = loc[2 * e], x = loc[2 * e + 1], f_y, f_x;
    sample_flow(fx, fy, H, W, y, x, f_y, f_x);
    if (flow_out) { flow_out[2 * e] = f_y; flow_out[2 * e + 1] = f_x; }
    if (!do_warp) return;
    float dt = tref - ts[e];
    y = y + dt * f_y;
    x = x + dt * f_x;
    float m = inbounds(y, x, H, W) ? 1.0f : 0.0f;
    loc[2 * e] = y * m;
    loc[2 * e + 1] = x * m;
    mask[2 * e] *= m;
    mask[2 * e + 1] *= m;
    ts[e] = tref;
}

// Image of a (warped) event list: utils/iwe.py:63-136 with round_idx (nearest pixel, torch.round = half-to-even,
// weight 1; metrics) or bilinear (4 corners; visualisation images).  cnt[2][HW] += w * mask_c; tsum += (w * ts) * mask_c.
__global__ __launch_bounds__(256) void val_splat_kernel(const float *__restrict__ loc, const float *__restrict__ mask,
                                                        const float *__restrict__ ts, int N, int H, int W, int round_idx,
                                                        float *__restrict__ cnt, float *__restrict__ tsum)
{
    int e = blockIdx.x * blockDim.x + threadIdx.x;
    if (e >= N) return;
    float y = loc[2 * e], x = loc[2 * e + 1];
    float m0 = mask[2 * e], m1 = mask[2 * e + 1];
    float t = ts ? ts[e] : 0.0f;
    const int HW = H * W;
    if (round_idx) {
        float ry = rintf(y), rx = rintf(x);
        if (ry >= 0.0f && ry < (float)H && rx >= 0.0f && rx < (float)W) {
            int p = (int)ry * W + (int)rx;
            if (m0 != 0.0f) { atomicAdd(cnt + p, m0); if (tsum) atomicAdd(tsum + p, t * m0); }
            if (m1 != 0.0f) { atomicAdd(cnt + HW + p, m1); if (tsum) atomicAdd(tsum + HW + p, t * m1); }
        }
        return;
    }
    float cy[2] = {floorf(y), floorf(y + 1.0f)}, cx[2] = {floorf(x), floorf(x + 1.0f)};
#pragma unroll
    for (int k = 0; k < 4; ++k) {
        float iy = cy[k >> 1], ix = cx[k & 1];
        if (!(iy >= 0.0f && iy < (float)H && ix >= 0.0f && ix < (float)W)) continue;
        float w = fmaxf(0.0f, 1.0f - fabsf(y - iy)) * fmaxf(0.0f, 1.0f - fabsf(x - ix));
        if (w == 0.0f) continue;
        int p = (int)iy * W + (int)ix;
        if (m0 != 0.0f) { atomicAdd(cnt + p, w * m0); if (tsum) atomicAdd(tsum + p, (w * t) * m0); }
        if (m1 != 0.0f) { atomicAdd(cnt + HW + p, w * m1); if (tsum) atomicAdd(tsum + HW + p, (w * t) * m1); }
    }
}

__device__ __forceinline__ double block_sum(double v, double *sh)
{
    sh[threadIdx.x] = v;
    __syncthreads();
    for (int s = blockDim.x >> 1; s > 0; s >>= 1) {
        if ((int)threadIdx.x < s) sh[threadIdx.x] += sh[threadIdx.x + s];
        __syncthreads();
    }
    double r = sh[0];
    __syncthreads();
    return r;
}

// FWL (loss/flow_val.py:189-212) = var(fw count image) / var(zero count image) (unbiased, over all pixels);
// RSAT (:214-274) = [sum (T/(C+eps)/passes)^2 / #{C_pos+C_neg > 0}]_fw / [same]_zero.     out = (fwl, rsat)
// Two launches: every workgroup leaves its 8 partial sums (fp64) in `part`, one workgroup adds them in block order.
constexpr int kMetricThreads = 256, kMetricBlocksMax = 256;

__global__ __launch_bounds__(kMetricThreads) void val_metrics_partial_kernel(const float *__restrict__ cf,
                                                                             const float *__restrict__ tf,
                                                                             const float *__restrict__ cz,
                                                                             const float *__restrict__ tz, int HW,
                                                                             float passes, double *__restrict__ part)
{
    __shared__ double sh[kMetricThreads];
    double s1[2] = {0, 0}, s2[2] = {0, 0}, sq[2] = {0, 0}, nz[2] = {0, 0};
    for (int p = blockIdx.x * blockDim.x + threadIdx.x; p < HW; p += gridDim.x * blockDim.x) {
#pragma unroll
        for (int k = 0; k < 2; ++k) {
            const float *c = k ? cz : cf, *t = k ? tz : tf;
            float c0 = c[p], c1 = c[HW + p];
            float img = c0 + c1;
            s1[k] += img;
            s2[k] += (double)img * img;
            float a0 = t[p] / (c0 + 1e-9f) / passes, a1 = t[HW + p] / (c1 + 1e-9f) / passes;
            sq[k] += (double)(a0 * a0) + (double)(a1 * a1);
            nz[k] += (img > 0.0f) ? 1.0 : 0.0;
        }
    }
    for (int k = 0; k < 2; ++k) {
        double r0 = block_sum(s1[k], sh), r1 = block_sum(s2[k], sh), r2 = block_sum(sq[k], sh), r3 = block_sum(nz[k], sh);
        if (threadIdx.x == 0) {
            double *o = part + (size_t)blockIdx.x * 8;
            o[k] = r0; o[2 + k] = r1; o[4 + k] = r2; o[6 + k] = r3;
        }
    }
}

__global__ __launch_bounds__(64) void val_metrics_final_kernel(const double *__restrict__ part, int nblocks, int HW,
                                                                float *__restrict__ out)
{
    if (threadIdx.x != 0) return;
    double r[8] = {0, 0, 0, 0, 0, 0, 0, 0};
    for (int b = 0; b < nblocks; ++b)
        for (int q = 0; q < 8; ++q) r[q] += part[(size_t)b * 8 + q];
    double n = (double)HW;
    double var_f = (r[2] - r[0] * r[0] / n) / (n - 1.0), var_z = (r[3] - r[1] * r[1] / n) / (n - 1.0);
    out[0] = (float)(var_f / var_z);
    out[1] = (float)((r[4] / r[6]) / (r[5] / r[7]));
}

// Forward propagation of a flow map (loss/flow_val.py:43-74): every pixel carries its flow vector to
// pixel + dt * flow and splats it bilinearly; acc = (weight, weight * f_y, weight * f_x) planes [3][HW].
__global__ __launch_bounds__(256) void val_prop_splat_kernel(const float *__restrict__ fx, const float *__restrict__ fy,
                                                             int H, int W, float dt, float *__restrict__ acc)
{
    int p = blockIdx.x * blockDim.x + threadIdx.x;
    const int HW = H * W;
    if (p >= HW) return;
    float py = (float)(p / W), px = (float)(p % W), f_y, f_x;
    sample_flow(fx, fy, H, W, py, px, f_y, f_x);
    float y = py + dt * f_y, x = px + dt * f_x;
    if (!inbounds(y, x, H, W)) return;        // purged: weight 0 (the zeroed location gets nothing either)
    float cy[2] = {floorf(y), floorf(y + 1.0f)}, cx[2] = {floorf(x), floorf(x + 1.0f)};
#pragma unroll
    for (int k = 0; k < 4; ++k) {
        float iy = cy[k >> 1], ix = cx[k & 1];
        if (!(iy >= 0.0f && iy < (float)H && ix >= 0.0f && ix < (float)W)) continue;
        float w = fmaxf(0.0f, 1.0f - fabsf(y - iy)) * fmaxf(0.0f, 1.0f - fabsf(x - ix));
        if (w == 0.0f) continue;
        int q = (int)iy * W + (int)ix;
        atomicAdd(acc + q, w);
        atomicAdd(acc + HW + q, w * f_y);
        atomicAdd(acc + 2 * HW + q, w * f_x);
    }
}

__global__ __launch_bounds__(256) void val_prop_divide_kernel(const float *__restrict__ acc, int HW,
                                                              float *__restrict__ out_x, float *__restrict__ out_y)
{
    int p = blockIdx.x * blockDim.x + threadIdx.x;
    if (p >= HW) return;
    float w = acc[p] + 1e-9f;
    out_y[p] = acc[HW + p] / w;
    out_x[p] = acc[2 * HW + p] / w;
}

// Accumulated backward flow (loss/flow_val.py:582-604): indices (y, x planes) walk along the newest flow.
__global__ __launch_bounds__(256) void val_accum_flow_kernel(const float *__restrict__ fx, const float *__restrict__ fy,
                                                             int H, int W, float *__restrict__ idx,
                                                             float *__restrict__ out_mask, float *__restrict__ acc_x,
                                                             float *__restrict__ acc_y)
{
    int p = blockIdx.x * blockDim.x + threadIdx.x;
    const int HW = H * W;
    if (p >= HW) return;
    float y = idx[p], x = idx[HW + p], f_y, f_x;
    float valid = inbounds(y, x, H, W) ? 1.0f : 0.0f;
    out_mask[p] += valid;
    sample_flow(fx, fy, H, W, y, x, f_y, f_x);
    y = y + f_y * valid;
    x = x + f_x * valid;
    idx[p] = y;
    idx[HW + p] = x;
    acc_y[p] = y - (float)(p / W);
    acc_x[p] = x - (float)(p % W);
}

// Per-pixel average of P flow maps over the passes where the flow is non-zero (loss/flow_val.py:145-172).
__global__ __launch_bounds__(256) void val_avg_flow_kernel(const float *__restrict__ mx, const float *__restrict__ my,
                                                           int P, int HW, const float *__restrict__ div,
                                                           const float *__restrict__ event_mask, int PM,
                                                           float *__restrict__ out)
{
    int p = blockIdx.x * blockDim.x + threadIdx.x;
    if (p >= HW) return;
    float sx = 0.0f, sy = 0.0f, cnt = 0.0f;
    for (int i = 0; i < P; ++i) {
        float vx = mx[(size_t)i * HW + p], vy = my[(size_t)i * HW + p];
        if (div) { vx /= div[p]; vy /= div[p]; }
        sx += vx;
        sy += vy;
        cnt += (vx != 0.0f || vy != 0.0f) ? 1.0f : 0.0f;
    }
    if (event_mask) {
        float m = 0.0f;
        for (int i = 0; i < PM; ++i) m += event_mask[(size_t)i * HW + p];
        float keep = m > 0.0f ? 1.0f : 0.0f;
        sx *= keep;
        sy *= keep;
    }
    out[p] = sx / (cnt + 1e-9f);
    out[HW + p] = sy / (cnt + 1e-9f);
}

// Average endpoint error (loss/flow_val.py:276-314) over pixels with valid ground truth (and input events).
__global__ __launch_bounds__(1024) void val_aee_kernel(const float *__restrict__ pred, const float *__restrict__ gt,
                                                       const float *__restrict__ mask, int PM, int HW,
                                                       float *__restrict__ out)
{
    __shared__ double sh[1024];
    double s = 0.0, n = 0.0;
    for (int p = threadIdx.x; p < HW; p += blockDim.x) {
        float gx = gt[p], gy = gt[HW + p];
        bool ok = !(gx == 0.0f && gy == 0.0f);
        if (ok && mask) {
            float m = 0.0f;
            for (int i = 0; i < PM; ++i) m += mask[(size_t)i * HW + p];
            ok = m > 0.0f;
        }
        if (!ok) continue;
        float dx = pred[p] - gx, dy = pred[HW + p] - gy;
        s += (double)sqrtf(dx * dx + dy * dy);
        n += 1.0;
    }
    double ts = block_sum(s, sh), tn = block_sum(n, sh);
    if (threadIdx.x == 0) out[0] = (float)(ts / tn);
}

// One-shot per-polarity IWE for visualisation: utils/iwe.py:139-224 deblur_events / :227-257 compute_pol_iwe.
// The flow lookup here is the reference's own gather (nearest by truncation, or 4-corner bilinear with hat weights),
// NOT grid_sample; events are warped to tref = 1 and splatted (nearest or bilinear).  out [B][2][H][W].
__global__ __launch_bounds__(256) void pol_iwe_kernel(const float *__restrict__ flow, const float *__restrict__ ev,
                                                      const float *__restrict__ pm, int B, int N, int H, int W,
                                                      int round_idx, int round_flow, float *__restrict__ out)
{
    int e = blockIdx.x * blockDim.x + threadIdx.x;
    int b = blockIdx.y;
    if (e >= N) return;
    const int HW = H * W;
    const float *evp = ev + ((size_t)b * N + e) * 4;
    float ts = evp[0], y = evp[1], x = evp[2];
    if (!(y >= 0.0f && y < (float)H && x >= 0.0f && x < (float)W)) return;     // mask_unfeasible zeroes the weights
    const float *fxm = flow + (size_t)b * 2 * HW, *fym = fxm + HW;               // channel 0 = x, 1 = y
    float f_y, f_x;
    if (round_flow) {
        int p = (int)(y * (float)W + x);         // (y * W + x).long()
        f_y = fym[p];
        f_x = fxm[p];
    } else {
        float cy[2] = {floorf(y), floorf(y + 1.0f)}, cx[2] = {floorf(x), floorf(x + 1.0f)};
        f_y = 0.0f;
        f_x = 0.0f;
#pragma unroll
        for (int k = 0; k < 4; ++k) {            // corner order TL, TR, BL, BR as in the reference's summation
            float iy = cy[k >> 1], ix = cx[k & 1];
            bool ok = iy >= 0.0f && iy < (float)H && ix >= 0.0f && ix < (float)W;
            float w = ok ? fmaxf(0.0f, 1.0f - fabsf(y - iy)) * fmaxf(0.0f, 1.0f - fabsf(x - ix)) : 0.0f;
            int p = ok ? (int)(iy * (float)W + ix) : 0;
            f_y += w * fym[p];
            f_x += w * fxm[p];
        }
    }
    float wy = y + (1.0f - ts) * f_y, wx = x + (1.0f - ts) * f_x;
    float m0 = pm[((size_t)b * N + e) * 2], m1 = pm[((size_t)b * N + e) * 2 + 1];
    float *o = out + (size_t)b * 2 * HW;
    if (round_idx) {
        float ry = rintf(wy), rx = rintf(wx);
        if (ry >= 0.0f && ry < (float)H && rx >= 0.0f && rx < (float)W) {
            int p = (int)ry * W + (int)rx;
            if (m0 != 0.0f) atomicAdd(o + p, m0);
            if (m1 != 0.0f) atomicAdd(o + HW + p, m1);
        }
        return;
    }
    float cy[2] = {floorf(wy), floorf(wy + 1.0f)}, cx[2] = {floorf(wx), floorf(wx + 1.0f)};
#pragma unroll
    for (int k = 0; k < 4; ++k) {
        float iy = cy[k >> 1], ix = cx[k & 1];
        if (!(iy >= 0.0f && iy < (float)H && ix >= 0.0f && ix < (float)W)) continue;
        float w = fmaxf(0.0f, 1.0f - fabsf(wy - iy)) * fmaxf(0.0f, 1.0f - fabsf(wx - ix));
        if (w == 0.0f) continue;
        int p = (int)iy * W + (int)ix;
        if (m0 != 0.0f) atomicAdd(o + p, w * m0);
        if (m1 != 0.0f) atomicAdd(o + HW + p, w * m1);
    }
}

inline unsigned nblk(size_t n) { return (unsigned)((n + 255) / 256); }

}  // namespace

// ---- stand-alone forms of the reference's interpolation primitives (utils/iwe.py:63-136), forward only -----------------
// get_interpolation: corners TL, TR, BL, BR as four blocks of n along the event axis; linear index iy * W + ix and bilinear
// weight per corner, both zeroed for corners outside the frame; round_idx: nearest pixel (torch.round), weight 1.
__global__ __launch_bounds__(256) void interp_corners_kernel(const float *__restrict__ loc, int B, int n, int H, int W,
                                                             int round_idx, float *__restrict__ idx, float *__restrict__ wgt)
{
    int e = blockIdx.x * blockDim.x + threadIdx.x;
    if (e >= B * n) return;
    const int b = e / n, k = e - b * n;
    const float y = loc[2 * (size_t)e], x = loc[2 * (size_t)e + 1];
    if (round_idx) {
        const float ry = rintf(y), rx = rintf(x);
        const float m = (ry >= 0.0f && ry < (float)H && rx >= 0.0f && rx < (float)W) ? 1.0f : 0.0f;
        idx[e] = (ry * m) * (float)W + rx * m;
        wgt[e] = m;                                       // prod(ones) * mask
        return;
    }
    const float cy[2] = {floorf(y), floorf(y + 1.0f)}, cx[2] = {floorf(x), floorf(x + 1.0f)};
#pragma unroll
    for (int c = 0; c < 4; ++c) {
        const float iy = cy[c >> 1], ix = cx[c & 1];
        const float m = (iy >= 0.0f && iy < (float)H && ix >= 0.0f && ix < (float)W) ? 1.0f : 0.0f;
        const float w = fmaxf(0.0f, 1.0f - fabsf(y - iy)) * fmaxf(0.0f, 1.0f - fabsf(x - ix));
        const size_t o = ((size_t)b * 4 + c) * n + k;
        idx[o] = (iy * m) * (float)W + ix * m;
        wgt[o] = w * m;
    }
}

// interpolate: out[b][idx] += w (* mask); out zeroed by the caller
__global__ __launch_bounds__(256) void scatter_add_kernel(const float *__restrict__ idx, const float *__restrict__ wgt,
                                                          const float *__restrict__ mask, int B, int n, int HW,
                                                          float *__restrict__ out)
{
    int e = blockIdx.x * blockDim.x + threadIdx.x;
    if (e >= B * n) return;
    const int b = e / n;
    float w = wgt[e];
    if (mask) w *= mask[e];
    const long p = (long)idx[e];
    if (p >= 0 && p < HW) atomicAdd(out + (size_t)b * HW + p, w);
}

extern "C" {

int tef_val_event_step(const float *fx, const float *fy, int H, int W, float *loc, float *ts, float *mask, int N,
                       float tref, int do_warp, float *flow_out, void *stream)
{
    if (!fx || !fy || H < 2 || W < 2 || N < 0 || (N > 0 && !loc) || (do_warp && N > 0 && (!ts || !mask)))
        return tef::fail("tef_val_event_step: bad arguments"), TEF_ERR_INVALID;
    if (N == 0) return 0;
    hipLaunchKernelGGL(val_event_step_kernel, dim3(nblk(N)), dim3(256), 0, (hipStream_t)stream, fx, fy, H, W, loc, ts, mask, N,
                       tref, do_warp, flow_out);
    return tef::check_launch("val_event_step_kernel");
}

int tef_val_event_image(const float *loc, const float *mask, const float *ts, int N, int H, int W, int round_idx,
                        float *cnt, float *tsum, void *stream)
{
    if (H < 1 || W < 1 || N < 0 || !cnt || (N > 0 && (!loc || !mask)))
        return tef::fail("tef_val_event_image: bad arguments"), TEF_ERR_INVALID;
    hipStream_t st = (hipStream_t)stream;
    if (hipMemsetAsync(cnt, 0, sizeof(float) * 2 * H * W, st) != hipSuccess) return tef::fail("memset failed"), TEF_ERR_LAUNCH;
    if (tsum && hipMemsetAsync(tsum, 0, sizeof(float) * 2 * H * W, st) != hipSuccess) return tef::fail("memset failed"), TEF_ERR_LAUNCH;
    if (N == 0) return 0;
    hipLaunchKernelGGL(val_splat_kernel, dim3(nblk(N)), dim3(256), 0, st, loc, mask, ts, N, H, W, round_idx, cnt, tsum);
    return tef::check_launch("val_splat_kernel");
}

static int metric_blocks(int HW)
{
    int nb = (HW + kMetricThreads * 4 - 1) / (kMetricThreads * 4);
    return nb < 1 ? 1 : (nb > kMetricBlocksMax ? kMetricBlocksMax : nb);
}

size_t tef_val_metrics_scratch_bytes(int H, int W)
{
    if (H < 1 || W < 1) return 0;
    return (size_t)metric_blocks(H * W) * 8 * sizeof(double);
}

int tef_val_metrics(const float *cnt_fw, const float *ts_fw, const float *cnt_zero, const float *ts_zero, int H, int W,
                    float passes, float *out2, void *scratch, size_t scratch_bytes, void *stream)
{
    if (!cnt_fw || !ts_fw || !cnt_zero || !ts_zero || !out2 || !scratch || H * W < 2)
        return tef::fail("tef_val_metrics: bad arguments"), TEF_ERR_INVALID;
    if (scratch_bytes < tef_val_metrics_scratch_bytes(H, W)) return tef::fail("tef_val_metrics: scratch too small"), TEF_ERR_WORKSPACE;
    const int nb = metric_blocks(H * W);
    hipStream_t st = (hipStream_t)stream;
    hipLaunchKernelGGL(val_metrics_partial_kernel, dim3(nb), dim3(kMetricThreads), 0, st, cnt_fw, ts_fw, cnt_zero, ts_zero,
                       H * W, passes, (double *)scratch);
    if (int rc = tef::check_launch("val_metrics_partial_kernel")) return rc;
    hipLaunchKernelGGL(val_metrics_final_kernel, dim3(1), dim3(64), 0, st, (const double *)scratch, nb, H * W, out2);
    return tef::check_launch("val_metrics_final_kernel");
}

int tef_val_forward_prop_flow(const float *fx, const float *fy, int H, int W, float dt, float *scratch3, float *out_x,
                              float *out_y, void *stream)
{
    if (!fx || !fy || !scratch3 || !out_x || !out_y || H < 2 || W < 2)
        return tef::fail("tef_val_forward_prop_flow: bad arguments"), TEF_ERR_INVALID;
    hipStream_t st = (hipStream_t)stream;
    const int HW = H * W;
    if (hipMemsetAsync(scratch3, 0, sizeof(float) * 3 * HW, st) != hipSuccess) return tef::fail("memset failed"), TEF_ERR_LAUNCH;
    hipLaunchKernelGGL(val_prop_splat_kernel, dim3(nblk(HW)), dim3(256), 0, st, fx, fy, H, W, dt, scratch3);
    if (int rc = tef::check_launch("val_prop_splat_kernel")) return rc;
    hipLaunchKernelGGL(val_prop_divide_kernel, dim3(nblk(HW)), dim3(256), 0, st, scratch3, HW, out_x, out_y);
    return tef::check_launch("val_prop_divide_kernel");
}

int tef_val_accum_flow(const float *fx, const float *fy, int H, int W, float *indices, float *out_mask, float *acc_x,
                       float *acc_y, void *stream)
{
    if (!fx || !fy || !indices || !out_mask || !acc_x || !acc_y || H < 2 || W < 2)
        return tef::fail("tef_val_accum_flow: bad arguments"), TEF_ERR_INVALID;
    hipLaunchKernelGGL(val_accum_flow_kernel, dim3(nblk((size_t)H * W)), dim3(256), 0, (hipStream_t)stream, fx, fy, H, W,
                       indices, out_mask, acc_x, acc_y);
    return tef::check_launch("val_accum_flow_kernel");
}

int tef_val_average_flow(const float *maps_x, const float *maps_y, int P, int H, int W, const float *divisor,
                         const float *event_mask, int mask_passes, float *out, void *stream)
{
    if (!maps_x || !maps_y || !out || P < 1 || H < 1 || W < 1)
        return tef::fail("tef_val_average_flow: bad arguments"), TEF_ERR_INVALID;
    hipLaunchKernelGGL(val_avg_flow_kernel, dim3(nblk((size_t)H * W)), dim3(256), 0, (hipStream_t)stream, maps_x, maps_y, P,
                       H * W, divisor, event_mask, mask_passes, out);
    return tef::check_launch("val_avg_flow_kernel");
}

int tef_pol_iwe(const float *flow, const float *event_list, const float *pol_mask, int B, int N, int H, int W,
                int round_idx, int round_flow, float *out, void *stream)
{
    if (!flow || !out || B < 1 || N < 0 || H < 1 || W < 1 || (N > 0 && (!event_list || !pol_mask)))
        return tef::fail("tef_pol_iwe: bad arguments"), TEF_ERR_INVALID;
    hipStream_t st = (hipStream_t)stream;
    if (hipMemsetAsync(out, 0, sizeof(float) * (size_t)B * 2 * H * W, st) != hipSuccess) return tef::fail("memset failed"), TEF_ERR_LAUNCH;
    if (N == 0) return 0;
    hipLaunchKernelGGL(pol_iwe_kernel, dim3(nblk(N), B), dim3(256), 0, st, flow, event_list, pol_mask, B, N, H, W, round_idx,
                       round_flow, out);
    return tef::check_launch("pol_iwe_kernel");
}

int tef_val_aee(const float *pred, const float *gt, const float *event_mask, int mask_passes, int H, int W, float *out,
                void *stream)
{
    if (!pred || !gt || !out || H < 1 || W < 1) return tef::fail("tef_val_aee: bad arguments"), TEF_ERR_INVALID;
    hipLaunchKernelGGL(val_aee_kernel, dim3(1), dim3(1024), 0, (hipStream_t)stream, pred, gt, event_mask, mask_passes, H * W,
                       out);
    return tef::check_launch("val_aee_kernel");
}


int tef_interp_corners(const float *loc, int B, int n, int H, int W, int round_idx, float *idx, float *weights, void *stream)
{
    if (B < 1 || n < 0 || H < 1 || W < 1 || !idx || !weights || (n > 0 && !loc))
        return tef::fail("tef_interp_corners: bad arguments"), TEF_ERR_INVALID;
    if (n == 0) return 0;
    hipLaunchKernelGGL(interp_corners_kernel, dim3(nblk(B * n)), dim3(256), 0, (hipStream_t)stream, loc, B, n, H, W, round_idx,
                       idx, weights);
    return tef::check_launch("interp_corners_kernel");
}

int tef_scatter_add(const float *idx, const float *weights, const float *mask, int B, int n, int HW, float *out, void *stream)
{
    if (B < 1 || n < 0 || HW < 1 || !out || (n > 0 && (!idx || !weights)))
        return tef::fail("tef_scatter_add: bad arguments"), TEF_ERR_INVALID;
    hipStream_t st = (hipStream_t)stream;
    if (hipMemsetAsync(out, 0, sizeof(float) * (size_t)B * HW, st) != hipSuccess) return tef::fail("memset failed"), TEF_ERR_LAUNCH;
    if (n == 0) return 0;
    hipLaunchKernelGGL(scatter_add_kernel, dim3(nblk(B * n)), dim3(256), 0, st, idx, weights, mask, B, n, HW, out);
    return tef::check_launch("scatter_add_kernel");
}

}  // extern "C"
