// tef_loss.hip — event warping + image-of-warped-events (IWE) scatter + timestamp contrast loss,
// forward and backward, for gfx950 (MI355X).  Replaces the tensor-of-temporaries formulation of the
// reference (loss/flow.py:415-746 Iterative, :216-412 Linear, utils/iwe.py:5-136) by a per-event
// trajectory formulation:
//
//   forward   K1 warp      one thread per (head, sample, event): walks the event through the flow maps,
//                          stores its position at every reference time (trajectory planes) and the
//                          border-compensation flags (loss/flow.py:671-681)
//             K2 splat     one workgroup per (image, head, sample, polarity[, row band]): the IWE pair
//                          (count, sum of weighted timestamps) lives in LDS (128 KiB for 128x128),
//                          events stream in coalesced, bilinear corners go in with LDS float atomics
//             K3 stats     per image: sum of squared mean timestamps, number of active pixels
//             K4 reduce    deterministic sum of the per-image terms -> scalar loss
//   backward  K6 chain     one thread per (head, sample, grad event): gathers d loss / d position at
//                          every reference time from the stored IWEs, reverse sweep along the
//                          trajectory (grid_sample backward w.r.t. grid), emits one flow-gradient
//                          vector per (event, flow map)
//             K7 dflow     one workgroup per (flow map, head, sample[, row band]): bilinear splat of
//                          those vectors into an LDS-resident gradient map, plain coalesced write-out
//                          (no global atomics, no zero-fill pass)
//
// Arithmetic is fp32 and follows the reference / ATen op order where it matters for parity
// (coordinate normalisation + un-normalisation of grid_sample, floor(y + 1) corners, true division
// for the timestamp normalisation); compiled with -ffp-contract=off.
#include <hip/hip_runtime.h>
#include <stdint.h>
#include <stdio.h>
#include <string.h>

#include "tef.h"
#include "tef_common.h"

namespace {

constexpr float kEps = 1e-9f;
constexpr int kSplatThreads = 1024;
constexpr size_t kLdsBudget = 128 * 1024;   // per-workgroup image budget (160 KiB LDS per CU on gfx950)

// ---------------------------------------------------------------------------------------------
// Window descriptor handed to kernels by value (kernarg segment).
// ---------------------------------------------------------------------------------------------
struct Win {
    int kind, B, H, W, P, F, S, mode_div, M, Md, Mt, nplanes, nimg;
    int img_base[TEF_MAX_SCALES + 1];
    int off[TEF_MAX_PASSES + 1];
    int doff[TEF_MAX_PASSES + 1];
};

struct Events {
    const float *ts, *y, *x, *mp, *mn;
    const uint8_t *bin;
    int cap;
};

struct Img {
    int s, plane, le, he, lo, hi;
    float tref, delta, coef;
};

__host__ __device__ inline int images_of_scale(const Win &w, int s)
{
    int scale = w.P >> s;
    return (w.kind == TEF_KIND_ITERATIVE) ? (1 << s) * (scale + 1) : (1 << s) * 2;
}

// image index -> (scale, window, reference time, bin range, normalisation)   loss/flow.py:657-731 / :309-397
__device__ inline Img decode_image(const Win &w, int j)
{
    Img im;
    int s = 0;
    while (s + 1 < w.S && j >= w.img_base[s + 1]) ++s;
    int r = j - w.img_base[s];
    int scale = w.P >> s;
    im.s = s;
    if (w.kind == TEF_KIND_ITERATIVE) {
        int per = scale + 1, wi = r / per, q = r - wi * per;
        int delta = scale / w.mode_div;
        im.lo = wi * scale;
        im.hi = im.lo + scale;
        int tref = im.lo + q;
        im.le = max(im.lo, tref - delta);
        im.he = min(im.hi, tref + delta);
        im.plane = tref;
        im.tref = (float)tref;
        im.delta = (float)delta;
        im.coef = 1.0f / ((float)(1 << s) * (float)(2 * delta + 1) * (float)w.S * (float)w.F);
    } else {
        int wi = r >> 1, e = r & 1;
        im.lo = wi * scale;
        im.hi = im.lo + scale;
        im.le = im.lo;
        im.he = im.hi;
        im.plane = 2 * s + e;
        im.tref = (float)(e ? im.lo : im.hi);
        im.delta = (float)scale;
        im.coef = 1.0f / ((float)(1 << s) * 2.0f * (float)w.S * (float)w.F);
    }
    return im;
}

// ---------------------------------------------------------------------------------------------
// Bilinear flow lookup: utils/iwe.py:17-40 + ATen grid_sampler_2d (bilinear, align_corners=True, zeros).
// ---------------------------------------------------------------------------------------------
struct Taps {
    int i00, i01, i10, i11;   // -1 when outside
    float s, n, e, w;         // (1-fy), fy, (1-fx), fx
};

__device__ __forceinline__ float unnormalize(float v, int size)
{
    float nn = (2.0f * v) / (float)(size - 1) - 1.0f;       // utils/iwe.py:30-31
    return (nn + 1.0f) * ((float)(size - 1) / 2.0f);         // ATen ComputeLocation<align_corners=true>
}

__device__ __forceinline__ Taps make_taps(float y, float x, int H, int W)
{
    Taps t;
    float iy = unnormalize(y, H), ix = unnormalize(x, W);
    float fy = floorf(iy), fx = floorf(ix);
    t.n = iy - fy;
    t.w = ix - fx;
    t.s = 1.0f - t.n;
    t.e = 1.0f - t.w;
    int y0 = (int)fy, x0 = (int)fx, y1 = y0 + 1, x1 = x0 + 1;
    bool vy0 = (y0 >= 0) & (y0 < H), vy1 = (y1 >= 0) & (y1 < H);
    bool vx0 = (x0 >= 0) & (x0 < W), vx1 = (x1 >= 0) & (x1 < W);
    t.i00 = (vy0 && vx0) ? y0 * W + x0 : -1;
    t.i01 = (vy0 && vx1) ? y0 * W + x1 : -1;
    t.i10 = (vy1 && vx0) ? y1 * W + x0 : -1;
    t.i11 = (vy1 && vx1) ? y1 * W + x1 : -1;
    return t;
}

struct Quad { float v00, v01, v10, v11; };

__device__ __forceinline__ Quad load_quad(const float *map, const Taps &t)
{
    Quad q;
    q.v00 = t.i00 >= 0 ? map[t.i00] : 0.0f;
    q.v01 = t.i01 >= 0 ? map[t.i01] : 0.0f;
    q.v10 = t.i10 >= 0 ? map[t.i10] : 0.0f;
    q.v11 = t.i11 >= 0 ? map[t.i11] : 0.0f;
    return q;
}

__device__ __forceinline__ float quad_value(const Quad &q, const Taps &t)
{
    return q.v00 * (t.s * t.e) + q.v01 * (t.s * t.w) + q.v10 * (t.n * t.e) + q.v11 * (t.n * t.w);
}

// d value / d(y, x)
__device__ __forceinline__ void quad_jacobian(const Quad &q, const Taps &t, float &dy, float &dx)
{
    dx = (q.v01 - q.v00) * t.s + (q.v11 - q.v10) * t.n;
    dy = (q.v10 - q.v00) * t.e + (q.v11 - q.v01) * t.w;
}

__device__ __forceinline__ bool inbounds(float y, float x, int H, int W)   // utils/iwe.py:52-57 (closed interval)
{
    return (y >= 0.0f) & (y <= (float)H - 1.0f) & (x >= 0.0f) & (x <= (float)W - 1.0f);
}

__device__ __forceinline__ const float *flow_map(const Win &w, const float *flows, int t, int i, int b, int c)
{
    return flows + ((((size_t)t * w.F + i) * w.B + b) * 2 + c) * (size_t)(w.H * w.W);
}

// ---------------------------------------------------------------------------------------------
// Splat geometry: utils/iwe.py:63-113 get_interpolation, corners TL, TR, BL, BR.
// ---------------------------------------------------------------------------------------------
struct Splat {
    int iy[2], ix[2];
    float wy[2], wx[2];     // hat weights per axis (top/bottom, left/right)
    float sy[2], sx[2];     // their derivatives (autograd of max(0, 1 - |d|): ties split 0.5, abs'(0) = 0)
};

__device__ __forceinline__ float hat(float d, float &slope)
{
    float v = 1.0f - fabsf(d);
    float sg = (d > 0.0f) ? 1.0f : ((d < 0.0f) ? -1.0f : 0.0f);
    slope = (v > 0.0f) ? -sg : ((v == 0.0f) ? -0.5f * sg : 0.0f);
    return fmaxf(v, 0.0f);
}

__device__ __forceinline__ Splat make_splat(float y, float x)
{
    Splat s;
    float cy[2] = {floorf(y), floorf(y + 1.0f)};
    float cx[2] = {floorf(x), floorf(x + 1.0f)};
#pragma unroll
    for (int k = 0; k < 2; ++k) {
        s.iy[k] = (int)cy[k];
        s.ix[k] = (int)cx[k];
        s.wy[k] = hat(y - cy[k], s.sy[k]);
        s.wx[k] = hat(x - cx[k], s.sx[k]);
    }
    return s;
}

__device__ __forceinline__ uint32_t pack_meta(uint32_t bits, int kb, int kf)
{
    return bits | ((uint32_t)(kb + 1) << 8) | ((uint32_t)kf << 16);
}

// =============================================================================================
// K1 (Iterative): iterative warping of every event to every reference time.
// loss/flow.py:521-586 event_warping, :492-519 update_warping_indices, :599-654.
// traj plane k (k = 0..P) holds the event position at tref = k; meta packs the per-scale
// border-compensation bits (:671-681), kb (last out-of-bounds tref going backward, -1 if none)
// and kf (first out-of-bounds tref going forward, P+1 if none).
// =============================================================================================
__global__ __launch_bounds__(256) void iter_warp_kernel(Win w, const float *__restrict__ flows, Events g, Events d,
                                                        float2 *__restrict__ traj, uint32_t *__restrict__ meta)
{
    int u = blockIdx.x * blockDim.x + threadIdx.x;
    if (u >= w.Mt) return;
    int ib = blockIdx.y, i = ib / w.B, b = ib - i * w.B;
    bool isd = u >= w.M;
    int sl = isd ? u - w.M : u;
    const Events &E = isd ? d : g;
    size_t o = (size_t)b * E.cap + sl;
    uint32_t *mo = meta + (size_t)ib * w.Mt + u;
    if (E.mp[o] == 0.0f && E.mn[o] == 0.0f) {   // collate padding (dataloader/base.py:414-421): contributes nothing
        *mo = 0u;
        return;
    }
    const int H = w.H, W = w.W, P = w.P;
    float ts = E.ts[o], y0 = E.y[o], x0 = E.x[o];
    int t = E.bin[sl];
    float2 *tr = traj + (size_t)ib * w.nplanes * w.Mt + u;

    // flow at the original location, shared by the first forward and the first backward step
    Taps tp = make_taps(y0, x0, H, W);
    float fy0 = quad_value(load_quad(flow_map(w, flows, t, i, b, 1), tp), tp);
    float fx0 = quad_value(load_quad(flow_map(w, flows, t, i, b, 0), tp), tp);

    int kf = P + 1, kb = -1;
    {   // forward: maps t .. P-1, positions at tref = t+1 .. P
        float y = y0, x = x0, fy = fy0, fx = fx0;
        float dt = (float)(t + 1) - ts;             // utils/iwe.py:14 (tref - ts)
        for (int k = t; k < P; ++k) {
            if (k > t) {
                Taps q = make_taps(y, x, H, W);
                fy = quad_value(load_quad(flow_map(w, flows, k, i, b, 1), q), q);
                fx = quad_value(load_quad(flow_map(w, flows, k, i, b, 0), q), q);
                dt = 1.0f;
            }
            y = y + dt * fy;
            x = x + dt * fx;
            tr[(size_t)(k + 1) * w.Mt] = make_float2(y, x);
            if (!inbounds(y, x, H, W)) { kf = k + 1; break; }       // cumulative purge, loss/flow.py:575
        }
    }
    {   // backward: maps t .. 0, positions at tref = t .. 0
        float y = y0, x = x0, fy = fy0, fx = fx0;
        float dt = (float)t - ts;
        for (int k = t; k >= 0; --k) {
            if (k < t) {
                Taps q = make_taps(y, x, H, W);
                fy = quad_value(load_quad(flow_map(w, flows, k, i, b, 1), q), q);
                fx = quad_value(load_quad(flow_map(w, flows, k, i, b, 0), q), q);
                dt = -1.0f;
            }
            y = y + dt * fy;
            x = x + dt * fx;
            tr[(size_t)k * w.Mt] = make_float2(y, x);
            if (!inbounds(y, x, H, W)) { kb = k; break; }
        }
    }
    uint32_t bits = 0;
    for (int s = 0; s < w.S; ++s) {
        int scale = P >> s, wi = t / scale;
        if (wi >= (1 << s)) continue;
        int lo = wi * scale, hi = lo + scale;
        if (kb < lo && kf > hi) bits |= 1u << s;
    }
    *mo = pack_meta(bits, kb, kf);
}

// =============================================================================================
// K1 (Linear): one flow sample per event, linear warp to both ends of each window.
// loss/flow.py:268-283 (sample), :337-343 (warp + shared purge).  Plane 2s = forward (tref = hi),
// plane 2s+1 = backward (tref = lo).
// =============================================================================================
__global__ __launch_bounds__(256) void linear_warp_kernel(Win w, const float *__restrict__ flows, Events g, Events d,
                                                          float2 *__restrict__ traj, uint32_t *__restrict__ meta)
{
    int u = blockIdx.x * blockDim.x + threadIdx.x;
    if (u >= w.Mt) return;
    int ib = blockIdx.y, i = ib / w.B, b = ib - i * w.B;
    bool isd = u >= w.M;
    int sl = isd ? u - w.M : u;
    const Events &E = isd ? d : g;
    size_t o = (size_t)b * E.cap + sl;
    uint32_t *mo = meta + (size_t)ib * w.Mt + u;
    if (E.mp[o] == 0.0f && E.mn[o] == 0.0f) {
        *mo = 0u;
        return;
    }
    const int H = w.H, W = w.W;
    float ts = E.ts[o], y0 = E.y[o], x0 = E.x[o];
    int t = E.bin[sl];
    Taps tp = make_taps(y0, x0, H, W);
    float fy = quad_value(load_quad(flow_map(w, flows, t, i, b, 1), tp), tp);
    float fx = quad_value(load_quad(flow_map(w, flows, t, i, b, 0), tp), tp);
    float2 *tr = traj + (size_t)ib * w.nplanes * w.Mt + u;
    uint32_t bits = 0;
    for (int s = 0; s < w.S; ++s) {
        int scale = w.P >> s, wi = t / scale;
        if (wi >= (1 << s)) continue;
        int lo = wi * scale, hi = lo + scale;
        float dtf = (float)hi - ts, dtb = (float)lo - ts;
        float yf = y0 + dtf * fy, xf = x0 + dtf * fx;
        float yb = y0 + dtb * fy, xb = x0 + dtb * fx;
        tr[(size_t)(2 * s) * w.Mt] = make_float2(yf, xf);
        tr[(size_t)(2 * s + 1) * w.Mt] = make_float2(yb, xb);
        if (inbounds(yf, xf, H, W) && inbounds(yb, xb, H, W)) bits |= 1u << s;
    }
    *mo = pack_meta(bits, -1, 0);
}

// =============================================================================================
// K2: image of warped events.  loss/flow.py:81-110 iwe_formatting = utils/iwe.py:63-136
// get_interpolation + 4x interpolate (scatter_add_).  One workgroup owns the (count, timestamp)
// pair of ONE polarity of one image (or a row band of it) in LDS.
//   iwe [(j * F*B + ib) * 2 + c][H*W] float2 = (C, T) summed over grad AND detached events (:725-726).
// =============================================================================================
__global__ __launch_bounds__(kSplatThreads) void splat_kernel(Win w, Events g, Events d,
                                                              const float2 *__restrict__ traj,
                                                              const uint32_t *__restrict__ meta,
                                                              float2 *__restrict__ iwe, int rows_per_band, int nbands)
{
    extern __shared__ float2 img[];
    int bid = blockIdx.x;
    int band = bid % nbands;
    bid /= nbands;
    int c = bid & 1;
    bid >>= 1;
    const int FB = w.F * w.B;
    int ib = bid % FB, j = bid / FB;
    int b = ib % w.B;
    const int H = w.H, W = w.W;
    int r0 = band * rows_per_band, r1 = min(H, r0 + rows_per_band);
    int npx = (r1 - r0) * W;
    for (int p = threadIdx.x; p < npx; p += blockDim.x) img[p] = make_float2(0.0f, 0.0f);
    __syncthreads();

    Img im = decode_image(w, j);
    const float2 *pl = traj + ((size_t)ib * w.nplanes + im.plane) * w.Mt;
    const uint32_t *mt = meta + (size_t)ib * w.Mt;
#pragma unroll 1
    for (int list = 0; list < 2; ++list) {
        const Events &E = list ? d : g;
        int base = list ? w.M : 0;
        int s0 = list ? w.doff[im.le] : w.off[im.le];
        int s1 = list ? w.doff[im.he] : w.off[im.he];
        const float *mask = (c ? E.mn : E.mp) + (size_t)b * E.cap;
        const float *tsp = E.ts + (size_t)b * E.cap;
        for (int sl = s0 + threadIdx.x; sl < s1; sl += blockDim.x) {
            int u = base + sl;
            if (!((mt[u] >> im.s) & 1u)) continue;          // shared border mask (:671-681)
            float m = mask[sl];
            if (m == 0.0f) continue;
            float2 p = pl[u];
            float tau = 1.0f - fabsf(im.tref - tsp[sl]) / im.delta;     // :94-95
            Splat sp = make_splat(p.x, p.y);                 // traj stores (y, x) in (.x, .y)
#pragma unroll
            for (int k = 0; k < 4; ++k) {
                int iy = sp.iy[k >> 1], ix = sp.ix[k & 1];
                float wgt = sp.wy[k >> 1] * sp.wx[k & 1];
                if (wgt == 0.0f || iy < r0 || iy >= r1 || ix < 0 || ix >= W) continue;
                float2 *px = img + (iy - r0) * W + ix;
                atomicAdd(&px->x, wgt * m);
                atomicAdd(&px->y, (wgt * tau) * m);
            }
        }
    }
    __syncthreads();
    float2 *out = iwe + (((size_t)j * FB + ib) * 2 + c) * (size_t)(H * W) + (size_t)r0 * W;
    for (int p = threadIdx.x; p < npx; p += blockDim.x) out[p] = img[p];
}

// =============================================================================================
// K3: per-image focus loss terms.  loss/flow.py:112-129 focus_loss on A = T / (C + 1e-9) (:727).
//   stats[(j*FB + ib)*2 + 0] = sum_px (A_pos^2 + A_neg^2) / n,   [+1] = n = #{C_pos + C_neg != 0} + 1e-9
// =============================================================================================
__global__ __launch_bounds__(256) void image_stats_kernel(Win w, const float2 *__restrict__ iwe,
                                                          float *__restrict__ stats)
{
    __shared__ double ssum[256];
    __shared__ int scnt[256];
    const int HW = w.H * w.W;
    const float2 *pos = iwe + (size_t)blockIdx.x * 2 * HW;
    const float2 *neg = pos + HW;
    float acc = 0.0f;
    int nnz = 0;
    for (int p = threadIdx.x; p < HW; p += blockDim.x) {
        float2 a = pos[p], bq = neg[p];
        float a0 = a.y / (a.x + kEps), a1 = bq.y / (bq.x + kEps);
        acc += a0 * a0 + a1 * a1;
        nnz += ((a.x + bq.x) != 0.0f);
    }
    ssum[threadIdx.x] = (double)acc;
    scnt[threadIdx.x] = nnz;
    __syncthreads();
    for (int s = blockDim.x >> 1; s > 0; s >>= 1) {
        if ((int)threadIdx.x < s) {
            ssum[threadIdx.x] += ssum[threadIdx.x + s];
            scnt[threadIdx.x] += scnt[threadIdx.x + s];
        }
        __syncthreads();
    }
    if (threadIdx.x == 0) {
        float n = (float)scnt[0] + kEps;
        stats[(size_t)blockIdx.x * 2] = (float)ssum[0] / n;
        stats[(size_t)blockIdx.x * 2 + 1] = n;
    }
}

// K4: loss = sum_images coef * (sum over samples of the per-sample term); fixed summation order.
__global__ __launch_bounds__(256) void loss_reduce_kernel(Win w, const float *__restrict__ stats,
                                                          float *__restrict__ loss_out)
{
    __shared__ double ssum[256];
    const int FB = w.F * w.B;
    double acc = 0.0;
    for (int q = threadIdx.x; q < w.nimg * FB; q += blockDim.x) {
        Img im = decode_image(w, q / FB);
        acc += (double)stats[(size_t)q * 2] * (double)im.coef;
    }
    ssum[threadIdx.x] = acc;
    __syncthreads();
    for (int s = blockDim.x >> 1; s > 0; s >>= 1) {
        if ((int)threadIdx.x < s) ssum[threadIdx.x] += ssum[threadIdx.x + s];
        __syncthreads();
    }
    if (threadIdx.x == 0) loss_out[0] = (float)ssum[0];
}

// ---------------------------------------------------------------------------------------------
// d(coef * image loss)/d position of one event at one image:
//   dl/dw_k = sum_c m_c * K * 2 A_c (tau - A_c) / (C_c + eps),  K = grad_out * coef / n
// followed by the derivative of the bilinear hat weights.
// ---------------------------------------------------------------------------------------------
__device__ __forceinline__ float2 image_grad(const Win &w, const float2 *__restrict__ iwe,
                                             const float *__restrict__ stats, int ib, int j, float kscale, float tref,
                                             float delta, float2 p, float ts, float mp, float mn)
{
    const int HW = w.H * w.W;
    const int FB = w.F * w.B;
    size_t q = (size_t)j * FB + ib;
    float kimg = kscale / stats[q * 2 + 1];
    const float2 *pos = iwe + q * 2 * HW;
    const float2 *neg = pos + HW;
    float tau = 1.0f - fabsf(tref - ts) / delta;
    Splat sp = make_splat(p.x, p.y);
    float gy = 0.0f, gx = 0.0f;
#pragma unroll
    for (int k = 0; k < 4; ++k) {
        int iy = sp.iy[k >> 1], ix = sp.ix[k & 1];
        if (iy < 0 || iy >= w.H || ix < 0 || ix >= w.W) continue;
        int px = iy * w.W + ix;
        float dw = 0.0f;
        if (mp != 0.0f) {
            float2 ct = pos[px];
            float A = ct.y / (ct.x + kEps), R = 1.0f / (ct.x + kEps);
            dw += mp * (2.0f * A * (tau - A) * R);
        }
        if (mn != 0.0f) {
            float2 ct = neg[px];
            float A = ct.y / (ct.x + kEps), R = 1.0f / (ct.x + kEps);
            dw += mn * (2.0f * A * (tau - A) * R);
        }
        dw *= kimg;
        gy += dw * (sp.sy[k >> 1] * sp.wx[k & 1]);
        gx += dw * (sp.wy[k >> 1] * sp.sx[k & 1]);
    }
    return make_float2(gy, gx);
}

// gradient w.r.t. the event position at tref = k, summed over the temporal scales that use it (Iterative)
__device__ __forceinline__ float2 iter_position_grad(const Win &w, const float2 *__restrict__ iwe,
                                                     const float *__restrict__ stats, int ib, uint32_t bits, int t,
                                                     int k, float gout, float2 p, float ts, float mp, float mn)
{
    float2 g = make_float2(0.0f, 0.0f);
    for (int s = 0; s < w.S; ++s) {
        if (!((bits >> s) & 1u)) continue;
        int scale = w.P >> s, wi = t / scale;
        int lo = wi * scale, hi = lo + scale;
        if (k < lo || k > hi) continue;
        int delta = scale / w.mode_div;
        int le = max(lo, k - delta), he = min(hi, k + delta);
        if (t < le || t >= he) continue;
        int j = w.img_base[s] + wi * (scale + 1) + (k - lo);
        float coef = 1.0f / ((float)(1 << s) * (float)(2 * delta + 1) * (float)w.S * (float)w.F);
        float2 a = image_grad(w, iwe, stats, ib, j, gout * coef, (float)k, (float)delta, p, ts, mp, mn);
        g.x += a.x;
        g.y += a.y;
    }
    return g;
}

// =============================================================================================
// K6 (Iterative): reverse sweep along each grad event's trajectory.
// Autograd counterpart of loss/flow.py:555-584: p' = p + dt * f(p) with f = bilinear lookup, so the
// adjoint picks up (I + dt * J^T) per step, and every step leaves dt * adjoint as the gradient of the
// sampled flow vector.  contrib[(ib*P + k)*M + sl] = (d/d f_y, d/d f_x) of this event's sample of map k.
// =============================================================================================
__global__ __launch_bounds__(256) void iter_chain_bwd_kernel(Win w, const float *__restrict__ flows, Events g,
                                                             const float2 *__restrict__ traj,
                                                             const uint32_t *__restrict__ meta,
                                                             const float2 *__restrict__ iwe,
                                                             const float *__restrict__ stats,
                                                             const float *__restrict__ grad_out,
                                                             float2 *__restrict__ contrib)
{
    int sl = blockIdx.x * blockDim.x + threadIdx.x;
    if (sl >= w.M) return;
    int ib = blockIdx.y, i = ib / w.B, b = ib - i * w.B;
    const int H = w.H, W = w.W, P = w.P, M = w.M;
    float2 *co = contrib + (size_t)ib * P * M + sl;
    uint32_t mv = meta[(size_t)ib * w.Mt + sl];
    uint32_t bits = mv & 0xffu;
    if (bits == 0u) {
        for (int k = 0; k < P; ++k) co[(size_t)k * M] = make_float2(0.0f, 0.0f);
        return;
    }
    int kb = (int)((mv >> 8) & 0xffu) - 1, kf = (int)((mv >> 16) & 0xffu);
    size_t o = (size_t)b * g.cap + sl;
    float ts = g.ts[o], mp = g.mp[o], mn = g.mn[o];
    int t = g.bin[sl];
    float gout = grad_out[0];
    const float2 *tr = traj + (size_t)ib * w.nplanes * w.Mt + sl;
    float c0y = 0.0f, c0x = 0.0f;

    float ay = 0.0f, ax = 0.0f;
    for (int k = P; k > t; --k) {        // forward chain, newest first
        if (k >= kf) {
            if (k - 1 > t) co[(size_t)(k - 1) * M] = make_float2(0.0f, 0.0f);
            continue;
        }
        float2 pk = tr[(size_t)k * w.Mt];
        float2 gk = iter_position_grad(w, iwe, stats, ib, bits, t, k, gout, pk, ts, mp, mn);
        ay += gk.x;
        ax += gk.y;
        if (k - 1 == t) {
            float c = (float)(t + 1) - ts;
            c0y += c * ay;
            c0x += c * ax;
        } else {
            co[(size_t)(k - 1) * M] = make_float2(ay, ax);
            float2 q = tr[(size_t)(k - 1) * w.Mt];
            Taps tp = make_taps(q.x, q.y, H, W);
            float jyy, jyx, jxy, jxx;
            quad_jacobian(load_quad(flow_map(w, flows, k - 1, i, b, 1), tp), tp, jyy, jyx);
            quad_jacobian(load_quad(flow_map(w, flows, k - 1, i, b, 0), tp), tp, jxy, jxx);
            float ny = ay + (ay * jyy + ax * jxy), nx = ax + (ay * jyx + ax * jxx);
            ay = ny;
            ax = nx;
        }
    }
    ay = 0.0f;
    ax = 0.0f;
    for (int k = 0; k <= t; ++k) {       // backward chain, oldest first
        if (k <= kb) {
            if (k < t) co[(size_t)k * M] = make_float2(0.0f, 0.0f);
            continue;
        }
        float2 pk = tr[(size_t)k * w.Mt];
        float2 gk = iter_position_grad(w, iwe, stats, ib, bits, t, k, gout, pk, ts, mp, mn);
        ay += gk.x;
        ax += gk.y;
        if (k == t) {
            float c = (float)t - ts;
            c0y += c * ay;
            c0x += c * ax;
        } else {
            co[(size_t)k * M] = make_float2(-ay, -ax);
            float2 q = tr[(size_t)(k + 1) * w.Mt];
            Taps tp = make_taps(q.x, q.y, H, W);
            float jyy, jyx, jxy, jxx;
            quad_jacobian(load_quad(flow_map(w, flows, k, i, b, 1), tp), tp, jyy, jyx);
            quad_jacobian(load_quad(flow_map(w, flows, k, i, b, 0), tp), tp, jxy, jxx);
            float ny = ay - (ay * jyy + ax * jxy), nx = ax - (ay * jyx + ax * jxx);
            ay = ny;
            ax = nx;
        }
    }
    co[(size_t)t * M] = make_float2(c0y, c0x);
}

// K6 (Linear): d/d(sampled flow) = sum over scales and both window ends of (tref - ts) * d/d position.
__global__ __launch_bounds__(256) void linear_bwd_kernel(Win w, Events g, const float2 *__restrict__ traj,
                                                         const uint32_t *__restrict__ meta,
                                                         const float2 *__restrict__ iwe,
                                                         const float *__restrict__ stats,
                                                         const float *__restrict__ grad_out,
                                                         float2 *__restrict__ contrib)
{
    int sl = blockIdx.x * blockDim.x + threadIdx.x;
    if (sl >= w.M) return;
    int ib = blockIdx.y, b = ib % w.B;
    uint32_t bits = meta[(size_t)ib * w.Mt + sl] & 0xffu;
    float cy = 0.0f, cx = 0.0f;
    if (bits) {
        size_t o = (size_t)b * g.cap + sl;
        float ts = g.ts[o], mp = g.mp[o], mn = g.mn[o];
        int t = g.bin[sl];
        float gout = grad_out[0];
        const float2 *tr = traj + (size_t)ib * w.nplanes * w.Mt + sl;
        for (int s = 0; s < w.S; ++s) {
            if (!((bits >> s) & 1u)) continue;
            int scale = w.P >> s, wi = t / scale;
            int lo = wi * scale, hi = lo + scale;
            float coef = 1.0f / ((float)(1 << s) * 2.0f * (float)w.S * (float)w.F);
            for (int e = 0; e < 2; ++e) {
                float tref = (float)(e ? lo : hi);
                int j = w.img_base[s] + wi * 2 + e;
                float2 p = tr[(size_t)(2 * s + e) * w.Mt];
                float2 gp = image_grad(w, iwe, stats, ib, j, gout * coef, tref, (float)scale, p, ts, mp, mn);
                cy += (tref - ts) * gp.x;
                cx += (tref - ts) * gp.y;
            }
        }
    }
    contrib[(size_t)ib * w.M + sl] = make_float2(cy, cx);
}

// =============================================================================================
// K7: flow-map gradient = bilinear splat (grid_sample backward w.r.t. input) of the per-event vectors.
// One workgroup per (pass k, head, sample[, band]); LDS holds the (x, y) gradient planes.
// Sample position of event (bin t) on map k: t < k -> trajectory plane k, t > k -> plane k+1,
// t == k -> original location.  Linear: only the events of pass k sample map k.
//   dflows [P][F][B][2][H][W] is fully overwritten.
// =============================================================================================
__global__ __launch_bounds__(kSplatThreads) void dflow_splat_kernel(Win w, Events g, const float2 *__restrict__ traj,
                                                                    const float2 *__restrict__ contrib,
                                                                    float *__restrict__ dflows, int rows_per_band,
                                                                    int nbands)
{
    extern __shared__ float2 img[];      // (.x = d/d flow_x, .y = d/d flow_y)
    int bid = blockIdx.x;
    int band = bid % nbands;
    bid /= nbands;
    const int FB = w.F * w.B;
    int ib = bid % FB, k = bid / FB;
    int i = ib / w.B, b = ib - i * w.B;
    const int H = w.H, W = w.W, M = w.M;
    int r0 = band * rows_per_band, r1 = min(H, r0 + rows_per_band);
    int npx = (r1 - r0) * W;
    for (int p = threadIdx.x; p < npx; p += blockDim.x) img[p] = make_float2(0.0f, 0.0f);
    __syncthreads();

    const bool iter = (w.kind == TEF_KIND_ITERATIVE);
    const float2 *co = iter ? contrib + ((size_t)ib * w.P + k) * M : contrib + (size_t)ib * M;
    const float2 *tr = traj + (size_t)ib * w.nplanes * w.Mt;
    int s0 = iter ? 0 : w.off[k], s1 = iter ? M : w.off[k + 1];
    for (int sl = s0 + threadIdx.x; sl < s1; sl += blockDim.x) {
        float2 cv = co[sl];
        if (cv.x == 0.0f && cv.y == 0.0f) continue;
        int t = g.bin[sl];
        float y, x;
        if (t == k) {
            size_t o = (size_t)b * g.cap + sl;
            y = g.y[o];
            x = g.x[o];
        } else {
            float2 p = tr[(size_t)(t < k ? k : k + 1) * w.Mt + sl];
            y = p.x;
            x = p.y;
        }
        Taps tp = make_taps(y, x, H, W);
        const int idx[4] = {tp.i00, tp.i01, tp.i10, tp.i11};
        const float wt[4] = {tp.s * tp.e, tp.s * tp.w, tp.n * tp.e, tp.n * tp.w};
#pragma unroll
        for (int q = 0; q < 4; ++q) {
            if (idx[q] < 0) continue;
            int iy = idx[q] / W;
            if (iy < r0 || iy >= r1) continue;
            float2 *px = img + (idx[q] - r0 * W);
            atomicAdd(&px->x, cv.y * wt[q]);     // cv = (d/d f_y, d/d f_x)
            atomicAdd(&px->y, cv.x * wt[q]);
        }
    }
    __syncthreads();
    float *ox = dflows + ((((size_t)k * w.F + i) * w.B + b) * 2) * (size_t)(H * W) + (size_t)r0 * W;
    float *oy = ox + (size_t)H * W;
    for (int p = threadIdx.x; p < npx; p += blockDim.x) {
        float2 v = img[p];
        ox[p] = v.x;
        oy[p] = v.y;
    }
}

// K0: AoS -> SoA packing of one pass (Iterative.update / Linear.update bookkeeping, loss/flow.py:457-473).
__global__ __launch_bounds__(256) void pack_events_kernel(float *__restrict__ ev, const float *__restrict__ pm, int B,
                                                          int N, float ts_shift, float ts_override, int pass_idx,
                                                          int slot0, int cap, float *__restrict__ ts,
                                                          float *__restrict__ y, float *__restrict__ x,
                                                          float *__restrict__ mp, float *__restrict__ mn,
                                                          uint8_t *__restrict__ bin)
{
    int e = blockIdx.x * blockDim.x + threadIdx.x;
    if (e >= N) return;
    int b = blockIdx.y;
    float4 v = reinterpret_cast<const float4 *>(ev)[(size_t)b * N + e];
    float2 m = reinterpret_cast<const float2 *>(pm)[(size_t)b * N + e];
    float t = v.x + ts_shift;
    ev[((size_t)b * N + e) * 4] = t;                       // in-place shift of the caller's list (:457-458)
    size_t o = (size_t)b * cap + slot0 + e;
    ts[o] = (ts_override >= 0.0f) ? ts_override : t;
    y[o] = v.y;
    x[o] = v.z;
    mp[o] = m.x;
    mn[o] = m.y;
    if (b == 0) bin[slot0 + e] = (uint8_t)pass_idx;
}

// ---------------------------------------------------------------------------------------------
// Host side
// ---------------------------------------------------------------------------------------------
struct Layout {
    size_t traj, meta, iwe, stats, contrib, total;
};

inline size_t align_up(size_t v) { return (v + 255) & ~(size_t)255; }

bool make_win(const tef_loss_cfg *c, Win *w)
{
    if (!c) return tef::fail("null config");
    if (c->kind != TEF_KIND_ITERATIVE && c->kind != TEF_KIND_LINEAR) return tef::fail("unknown loss kind");
    if (c->B < 1 || c->H < 2 || c->W < 2 || c->F < 1) return tef::fail("bad B/H/W/F");
    if (c->P < 1 || c->P > TEF_MAX_PASSES) return tef::fail("passes_loss out of range [1, 64]");
    if (c->S < 1 || c->S > TEF_MAX_SCALES) return tef::fail("scales_loss out of range [1, 6]");
    if ((size_t)c->W * sizeof(float2) > kLdsBudget) return tef::fail("image row does not fit the LDS band");
    if (c->kind == TEF_KIND_ITERATIVE) {
        // iterative_mode "four" raises TypeError in the reference itself (loss/flow.py:666-692); only one/two exist here
        if (c->mode_div != 1 && c->mode_div != 2) return tef::fail("iterative_mode must be 'one' or 'two'");
        if (c->P < 2) return tef::fail("Iterative needs passes_loss >= 2 (the reference fails in torch.cat for 1)");
    }
    memset(w, 0, sizeof(*w));
    w->kind = c->kind; w->B = c->B; w->H = c->H; w->W = c->W; w->P = c->P; w->F = c->F; w->S = c->S;
    w->mode_div = c->mode_div; w->M = c->M; w->Md = c->Md; w->Mt = c->M + c->Md;
    w->nplanes = (c->kind == TEF_KIND_ITERATIVE) ? c->P + 1 : 2 * c->S;
    if (c->M < 0 || c->Md < 0 || c->off[0] != 0 || c->doff[0] != 0 || c->off[c->P] != c->M || c->doff[c->P] != c->Md)
        return tef::fail("inconsistent slot offsets");
    for (int t = 0; t <= c->P; ++t) {
        w->off[t] = c->off[t];
        w->doff[t] = c->doff[t];
        if (t && (c->off[t] < c->off[t - 1] || c->doff[t] < c->doff[t - 1])) return tef::fail("offsets not monotone");
    }
    int n = 0;
    for (int s = 0; s < c->S; ++s) {
        int scale = c->P >> s;
        if (scale < 1) return tef::fail("passes_loss // 2**scale is zero");
        if (c->kind == TEF_KIND_ITERATIVE && scale / c->mode_div < 1)
            return tef::fail("delta_passes is zero for a temporal scale (the reference divides by it)");
        w->img_base[s] = n;
        n += images_of_scale(*w, s);
    }
    w->img_base[c->S] = n;
    w->nimg = n;
    return true;
}

Layout make_layout(const Win &w)
{
    Layout L;
    const size_t FB = (size_t)w.F * w.B, HW = (size_t)w.H * w.W;
    size_t o = 0;
    L.traj = o;    o += align_up(FB * w.nplanes * (size_t)w.Mt * sizeof(float2));
    L.meta = o;    o += align_up(FB * (size_t)w.Mt * sizeof(uint32_t));
    L.iwe = o;     o += align_up((size_t)w.nimg * FB * 2 * HW * sizeof(float2));
    L.stats = o;   o += align_up((size_t)w.nimg * FB * 2 * sizeof(float));
    L.contrib = o; o += align_up(FB * (size_t)(w.kind == TEF_KIND_ITERATIVE ? w.P : 1) * (size_t)w.M * sizeof(float2));
    L.total = o;
    return L;
}

inline Events to_events(const tef_events *e)
{
    Events r;
    if (e) { r.ts = e->ts; r.y = e->y; r.x = e->x; r.mp = e->mp; r.mn = e->mn; r.bin = e->bin; r.cap = e->cap; }
    else { r.ts = r.y = r.x = r.mp = r.mn = nullptr; r.bin = nullptr; r.cap = 0; }
    return r;
}

inline void band_geometry(const Win &w, int *rows_per_band, int *nbands, size_t *lds)
{
    int rows = (int)(kLdsBudget / ((size_t)w.W * sizeof(float2)));
    if (rows > w.H) rows = w.H;
    *rows_per_band = rows;
    *nbands = (w.H + rows - 1) / rows;
    *lds = (size_t)rows * w.W * sizeof(float2);
}

bool g_attr_done = false;

bool ensure_attrs()
{
    if (g_attr_done) return true;
    // opt in to > 64 KiB dynamic LDS for the two LDS-resident splat kernels
    hipError_t e1 = hipFuncSetAttribute((const void *)splat_kernel, hipFuncAttributeMaxDynamicSharedMemorySize,
                                        (int)kLdsBudget);
    hipError_t e2 = hipFuncSetAttribute((const void *)dflow_splat_kernel, hipFuncAttributeMaxDynamicSharedMemorySize,
                                        (int)kLdsBudget);
    if (e1 != hipSuccess || e2 != hipSuccess) return tef::fail_hip("hipFuncSetAttribute", e1 != hipSuccess ? e1 : e2);
    g_attr_done = true;
    return true;
}

}  // namespace

extern "C" {

size_t tef_loss_workspace_bytes(const tef_loss_cfg *cfg)
{
    Win w;
    if (!make_win(cfg, &w)) return 0;
    return make_layout(w).total;
}

int tef_pack_events(float *ev, const float *pm, int B, int N, float ts_shift, float ts_override, int pass_idx,
                    int slot0, int cap, float *ts, float *y, float *x, float *mp, float *mn, uint8_t *bin,
                    void *stream)
{
    if (B < 1 || N < 0 || slot0 < 0 || slot0 + N > cap || pass_idx < 0 || pass_idx >= TEF_MAX_PASSES)
        return tef::fail("tef_pack_events: bad sizes"), TEF_ERR_INVALID;
    if (N == 0) return 0;
    dim3 grid((N + 255) / 256, B);
    { tef::ProfScope ps(tef::PROF_PACK, (hipStream_t)stream); hipLaunchKernelGGL(pack_events_kernel, grid, dim3(256), 0, (hipStream_t)stream, ev, pm, B, N, ts_shift, ts_override,
                       pass_idx, slot0, cap, ts, y, x, mp, mn, bin); }
    return tef::check_launch("pack_events_kernel");
}

int tef_loss_forward(const tef_loss_cfg *cfg, const float *flows, const tef_events *grad, const tef_events *det,
                     void *workspace, size_t workspace_bytes, float *loss_out, void *stream)
{
    Win w;
    if (!make_win(cfg, &w)) return TEF_ERR_INVALID;
    if (!flows || !grad || !workspace || !loss_out) return tef::fail("null pointer"), TEF_ERR_INVALID;
    if (w.Md > 0 && !det) return tef::fail("detached events missing"), TEF_ERR_INVALID;
    if (grad->cap < w.M || (det && w.Md > 0 && det->cap < w.Md)) return tef::fail("event capacity < slots"), TEF_ERR_INVALID;
    Layout L = make_layout(w);
    if (workspace_bytes < L.total) return tef::fail("workspace too small"), TEF_ERR_WORKSPACE;
    if (!ensure_attrs()) return TEF_ERR_LAUNCH;
    hipStream_t st = (hipStream_t)stream;
    char *ws = (char *)workspace;
    float2 *traj = (float2 *)(ws + L.traj);
    uint32_t *meta = (uint32_t *)(ws + L.meta);
    float2 *iwe = (float2 *)(ws + L.iwe);
    float *stats = (float *)(ws + L.stats);
    Events g = to_events(grad), d = to_events(w.Md > 0 ? det : nullptr);
    const int FB = w.F * w.B;

    if (w.Mt > 0) {
        dim3 grid((w.Mt + 255) / 256, FB);
        if (w.kind == TEF_KIND_ITERATIVE)
            { tef::ProfScope ps(tef::PROF_WARP, st); hipLaunchKernelGGL(iter_warp_kernel, grid, dim3(256), 0, st, w, flows, g, d, traj, meta); }
        else
            { tef::ProfScope ps(tef::PROF_WARP, st); hipLaunchKernelGGL(linear_warp_kernel, grid, dim3(256), 0, st, w, flows, g, d, traj, meta); }
        if (int rc = tef::check_launch("warp_kernel")) return rc;
    }
    int rows, nbands;
    size_t lds;
    band_geometry(w, &rows, &nbands, &lds);
    { tef::ProfScope ps(tef::PROF_SPLAT, st); hipLaunchKernelGGL(splat_kernel, dim3((unsigned)(w.nimg * FB * 2 * nbands)), dim3(kSplatThreads), lds, st, w, g, d,
                       traj, meta, iwe, rows, nbands); }
    if (int rc = tef::check_launch("splat_kernel")) return rc;
    { tef::ProfScope ps(tef::PROF_STATS, st); hipLaunchKernelGGL(image_stats_kernel, dim3((unsigned)(w.nimg * FB)), dim3(256), 0, st, w, iwe, stats); }
    if (int rc = tef::check_launch("image_stats_kernel")) return rc;
    { tef::ProfScope ps(tef::PROF_REDUCE, st); hipLaunchKernelGGL(loss_reduce_kernel, dim3(1), dim3(256), 0, st, w, stats, loss_out); }
    return tef::check_launch("loss_reduce_kernel");
}

int tef_loss_backward(const tef_loss_cfg *cfg, const float *flows, const tef_events *grad, const tef_events *det,
                      void *workspace, size_t workspace_bytes, const float *grad_out, float *dflows, void *stream)
{
    (void)det;
    Win w;
    if (!make_win(cfg, &w)) return TEF_ERR_INVALID;
    if (!flows || !grad || !workspace || !grad_out || !dflows) return tef::fail("null pointer"), TEF_ERR_INVALID;
    Layout L = make_layout(w);
    if (workspace_bytes < L.total) return tef::fail("workspace too small"), TEF_ERR_WORKSPACE;
    if (!ensure_attrs()) return TEF_ERR_LAUNCH;
    hipStream_t st = (hipStream_t)stream;
    char *ws = (char *)workspace;
    float2 *traj = (float2 *)(ws + L.traj);
    uint32_t *meta = (uint32_t *)(ws + L.meta);
    float2 *iwe = (float2 *)(ws + L.iwe);
    float *stats = (float *)(ws + L.stats);
    float2 *contrib = (float2 *)(ws + L.contrib);
    Events g = to_events(grad);
    const int FB = w.F * w.B;
    if (w.M > 0) {
        dim3 grid((w.M + 255) / 256, FB);
        if (w.kind == TEF_KIND_ITERATIVE)
            { tef::ProfScope ps(tef::PROF_CHAIN_BWD, st); hipLaunchKernelGGL(iter_chain_bwd_kernel, grid, dim3(256), 0, st, w, flows, g, traj, meta, iwe, stats,
                               grad_out, contrib); }
        else
            { tef::ProfScope ps(tef::PROF_CHAIN_BWD, st); hipLaunchKernelGGL(linear_bwd_kernel, grid, dim3(256), 0, st, w, g, traj, meta, iwe, stats, grad_out,
                               contrib); }
        if (int rc = tef::check_launch("chain_bwd_kernel")) return rc;
    }
    int rows, nbands;
    size_t lds;
    band_geometry(w, &rows, &nbands, &lds);
    { tef::ProfScope ps(tef::PROF_DFLOW, st); hipLaunchKernelGGL(dflow_splat_kernel, dim3((unsigned)(w.P * FB * nbands)), dim3(kSplatThreads), lds, st, w, g, traj,
                       contrib, dflows, rows, nbands); }
    return tef::check_launch("dflow_splat_kernel");
}

}  // extern "C"
