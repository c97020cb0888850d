// tef_smooth.hip — optional Charbonnier smoothness priors on the flow maps of a loss window.
// Reference: loss/flow.py:170-209 flow_spatial_smoothing, :131-168 flow_temporal_smoothing (both off in
// configs/train_flow.yml:21-22).  Dense stencils over [P][F][B][2][H][W]; HBM-trivial next to the CM loss.
#include <hip/hip_runtime.h>
#include <stdint.h>

#include "tef.h"
#include "tef_common.h"

namespace {

constexpr float kEps = 1e-9f;

struct Dims {
    int B, H, W, P, F;
};

__device__ __forceinline__ const float *fmap(const Dims &d, const float *flows, int t, int i, int b, int c)
{
    return flows + ((((size_t)t * d.F + i) * d.B + b) * 2 + c) * (size_t)(d.H * d.W);
}

// the four difference families of loss/flow.py:180-187 as (dy, dx) from the first to the second pixel
__device__ __constant__ int kOffY[4] = {0, 1, 1, -1};
__device__ __constant__ int kOffX[4] = {1, 0, 1, 1};

__device__ __forceinline__ bool pair_ok(int y, int x, int dy, int dx, int H, int W)
{
    int y2 = y + dy, x2 = x + dx;
    return (y >= 0) & (y < H) & (x >= 0) & (x < W) & (y2 >= 0) & (y2 < H) & (x2 >= 0) & (x2 < W);
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

// ---- spatial ---------------------------------------------------------------------------------
// partial[block] = sum over the block's pixels of sum_fam sum_c charb(d) / (ny*nx)
__global__ __launch_bounds__(256) void spatial_fwd_kernel(Dims d, const float *__restrict__ flows,
                                                          double *__restrict__ partial)
{
    __shared__ double sh[256];
    const int HW = d.H * d.W;
    int map = blockIdx.y;   // (t*F + i)*B + b
    int p = blockIdx.x * blockDim.x + threadIdx.x;
    double acc = 0.0;
    if (p < HW) {
        int y = p / d.W, x = p - y * d.W;
        const float *mx = flows + (size_t)map * 2 * HW, *my = mx + HW;
        for (int f = 0; f < 4; ++f) {
            int dy = kOffY[f], dx = kOffX[f];
            if (!pair_ok(y, x, dy, dx, d.H, d.W)) continue;
            int q = (y + dy) * d.W + x + dx;
            float cnt = (float)((d.H - (dy != 0)) * (d.W - (dx != 0)));
            float a = mx[p] - mx[q], b = my[p] - my[q];
            acc += (double)((sqrtf(a * a + 1e-6f) + sqrtf(b * b + 1e-6f)) / cnt);
        }
    }
    double tot = block_sum(acc, sh);
    if (threadIdx.x == 0) partial[(size_t)blockIdx.y * gridDim.x + blockIdx.x] = tot;
}

__global__ __launch_bounds__(256) void spatial_bwd_kernel(Dims d, const float *__restrict__ flows, float weight,
                                                          const float *__restrict__ grad_out,
                                                          float *__restrict__ dflows)
{
    const int HW = d.H * d.W;
    int map = blockIdx.y;
    int p = blockIdx.x * blockDim.x + threadIdx.x;
    if (p >= HW) return;
    int y = p / d.W, x = p - y * d.W;
    float k = grad_out[0] * weight / (4.0f * (float)d.F * (float)d.P);
    for (int c = 0; c < 2; ++c) {
        const float *m = flows + ((size_t)map * 2 + c) * HW;
        float g = 0.0f;
        for (int f = 0; f < 4; ++f) {
            int dy = kOffY[f], dx = kOffX[f];
            float cnt = (float)((d.H - (dy != 0)) * (d.W - (dx != 0)));
            if (pair_ok(y, x, dy, dx, d.H, d.W)) {            // this pixel is the first of the pair
                float a = m[p] - m[(y + dy) * d.W + x + dx];
                g += (a / sqrtf(a * a + 1e-6f)) / cnt;
            }
            if (pair_ok(y - dy, x - dx, dy, dx, d.H, d.W)) {  // this pixel is the second of the pair
                float a = m[(y - dy) * d.W + x - dx] - m[p];
                g -= (a / sqrtf(a * a + 1e-6f)) / cnt;
            }
        }
        dflows[((size_t)map * 2 + c) * HW + p] += k * g;
    }
}

// ---- temporal --------------------------------------------------------------------------------
struct Taps {
    int i00, i01, i10, i11;
    float s, n, e, w;
};

__device__ __forceinline__ float unnormalize(float v, int size)
{
    float nn = (2.0f * v) / (float)(size - 1) - 1.0f;
    return (nn + 1.0f) * ((float)(size - 1) / 2.0f);
}

__device__ __forceinline__ Taps make_taps(float y, float x, int H, int W)
{
    Taps t;
    float iy = unnormalize(y, H), ix = unnormalize(x, W);
    float fy = floorf(iy), fx = floorf(ix);
    t.n = iy - fy; t.w = ix - fx; t.s = 1.0f - t.n; t.e = 1.0f - t.w;
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

__device__ __forceinline__ Quad load_quad(const float *m, const Taps &t)
{
    Quad q;
    q.v00 = t.i00 >= 0 ? m[t.i00] : 0.0f; q.v01 = t.i01 >= 0 ? m[t.i01] : 0.0f;
    q.v10 = t.i10 >= 0 ? m[t.i10] : 0.0f; q.v11 = t.i11 >= 0 ? m[t.i11] : 0.0f;
    return q;
}

__device__ __forceinline__ float quad_value(const Quad &q, const Taps &t)
{
    return q.v00 * (t.s * t.e) + q.v01 * (t.s * t.w) + q.v10 * (t.n * t.e) + q.v11 * (t.n * t.w);
}

__device__ __forceinline__ bool inbounds(float y, float x, int H, int W)
{
    return (y >= 0.0f) & (y <= (float)H - 1.0f) & (x >= 0.0f) & (x <= (float)W - 1.0f);
}

// One block per (head i, pair j, sample b): term = sum(mask * charb) / (sum(mask) + eps)   (:161-163)
// terms[(i*(P-1)+j)*B + b] = {term, denom}
__global__ __launch_bounds__(256) void temporal_fwd_kernel(Dims d, const float *__restrict__ flows,
                                                           double *__restrict__ terms)
{
    __shared__ double sh[256];
    const int HW = d.H * d.W;
    int q = blockIdx.x;
    int b = q % d.B, j = (q / d.B) % (d.P - 1), i = q / (d.B * (d.P - 1));
    const float *fx0 = fmap(d, flows, j, i, b, 0), *fy0 = fmap(d, flows, j, i, b, 1);
    const float *fx1 = fmap(d, flows, j + 1, i, b, 0), *fy1 = fmap(d, flows, j + 1, i, b, 1);
    double acc = 0.0, cnt = 0.0;
    for (int p = threadIdx.x; p < HW; p += blockDim.x) {
        int y = p / d.W, x = p - y * d.W;
        float wy = (float)y + fy0[p], wx = (float)x + fx0[p];          // :143
        if (!inbounds(wy, wx, d.H, d.W)) continue;                      // :147-152
        Taps t = make_taps(wy, wx, d.H, d.W);
        float sy = quad_value(load_quad(fy1, t), t), sx = quad_value(load_quad(fx1, t), t);
        float a = fy0[p] - sy, c = fx0[p] - sx;
        acc += (double)(sqrtf(a * a + 1e-9f) + sqrtf(c * c + 1e-9f));   // :161-162
        cnt += 1.0;
    }
    double ta = block_sum(acc, sh), tc = block_sum(cnt, sh);
    if (threadIdx.x == 0) {
        float denom = (float)tc + kEps;
        terms[(size_t)q * 2] = (double)((float)ta / denom);
        terms[(size_t)q * 2 + 1] = (double)denom;
    }
}

__global__ __launch_bounds__(256) void temporal_bwd_kernel(Dims d, const float *__restrict__ flows, float weight,
                                                           const double *__restrict__ terms,
                                                           const float *__restrict__ grad_out,
                                                           float *__restrict__ dflows)
{
    const int HW = d.H * d.W;
    int q = blockIdx.y;
    int b = q % d.B, j = (q / d.B) % (d.P - 1), i = q / (d.B * (d.P - 1));
    int p = blockIdx.x * blockDim.x + threadIdx.x;
    if (p >= HW) return;
    const float *fx0 = fmap(d, flows, j, i, b, 0), *fy0 = fmap(d, flows, j, i, b, 1);
    const float *fx1 = fmap(d, flows, j + 1, i, b, 0), *fy1 = fmap(d, flows, j + 1, i, b, 1);
    int y = p / d.W, x = p - y * d.W;
    float wy = (float)y + fy0[p], wx = (float)x + fx0[p];
    if (!inbounds(wy, wx, d.H, d.W)) return;
    float scale = grad_out[0] * weight / ((float)d.F * (float)(d.P - 1) * (float)terms[(size_t)q * 2 + 1]);
    Taps t = make_taps(wy, wx, d.H, d.W);
    Quad qy = load_quad(fy1, t), qx = load_quad(fx1, t);
    float a = fy0[p] - quad_value(qy, t), c = fx0[p] - quad_value(qx, t);
    float gy = scale * (a / sqrtf(a * a + 1e-9f)), gx = scale * (c / sqrtf(c * c + 1e-9f));
    // Jacobians of the sampled next flow w.r.t. the sampling location
    float jyx = (qy.v01 - qy.v00) * t.s + (qy.v11 - qy.v10) * t.n, jyy = (qy.v10 - qy.v00) * t.e + (qy.v11 - qy.v01) * t.w;
    float jxx = (qx.v01 - qx.v00) * t.s + (qx.v11 - qx.v10) * t.n, jxy = (qx.v10 - qx.v00) * t.e + (qx.v11 - qx.v01) * t.w;
    float *dx0 = dflows + (fx0 - flows), *dy0 = dflows + (fy0 - flows);
    float *dx1 = dflows + (fx1 - flows), *dy1 = dflows + (fy1 - flows);
    atomicAdd(dy0 + p, gy - (gy * jyy + gx * jxy));
    atomicAdd(dx0 + p, gx - (gy * jyx + gx * jxx));
    const int idx[4] = {t.i00, t.i01, t.i10, t.i11};
    const float wt[4] = {t.s * t.e, t.s * t.w, t.n * t.e, t.n * t.w};
#pragma unroll
    for (int k = 0; k < 4; ++k) {
        if (idx[k] < 0) continue;
        atomicAdd(dy1 + idx[k], -gy * wt[k]);
        atomicAdd(dx1 + idx[k], -gx * wt[k]);
    }
}

// loss_out += weight_s/(4 F P) * sum(spatial partials) + weight_t/(F (P-1)) * sum(temporal terms)
__global__ __launch_bounds__(256) void smoothing_reduce_kernel(Dims d, const double *__restrict__ spart, int ns,
                                                               float ws, const double *__restrict__ terms, int nt,
                                                               float wt, float *__restrict__ loss_out)
{
    __shared__ double sh[256];
    double a = 0.0, b = 0.0;
    for (int q = threadIdx.x; q < ns; q += blockDim.x) a += spart[q];
    for (int q = threadIdx.x; q < nt; q += blockDim.x) b += terms[(size_t)q * 2];
    double ta = block_sum(a, sh), tb = block_sum(b, sh);
    if (threadIdx.x == 0) {
        float add = 0.0f;
        if (ns > 0) add += ws * (float)(ta / (4.0 * d.F * d.P));
        if (nt > 0) add += wt * (float)(tb / ((double)d.F * (d.P - 1)));
        loss_out[0] += add;
    }
}

struct Plan {
    Dims d;
    int sblocks, nmaps, ns, nt;
    size_t off_terms, total;
};

bool make_plan(const tef_loss_cfg *c, Plan *pl)
{
    if (!c || c->B < 1 || c->H < 2 || c->W < 2 || c->P < 1 || c->F < 1) return tef::fail("tef_smoothing: bad config");
    pl->d = Dims{c->B, c->H, c->W, c->P, c->F};
    pl->sblocks = (c->H * c->W + 255) / 256;
    pl->nmaps = c->P * c->F * c->B;
    pl->ns = pl->sblocks * pl->nmaps;
    pl->nt = c->P > 1 ? c->F * (c->P - 1) * c->B : 0;
    pl->off_terms = ((size_t)pl->ns * sizeof(double) + 255) & ~(size_t)255;
    pl->total = pl->off_terms + (((size_t)pl->nt * 2 * sizeof(double) + 255) & ~(size_t)255);
    return true;
}

}  // namespace

extern "C" {

size_t tef_smoothing_scratch_bytes(const tef_loss_cfg *cfg)
{
    Plan pl;
    if (!make_plan(cfg, &pl)) return 0;
    return pl.total;
}

int tef_smoothing_forward(const tef_loss_cfg *cfg, const float *flows, float spat_weight, float temp_weight,
                          void *scratch, float *loss_out, void *stream)
{
    Plan pl;
    if (!make_plan(cfg, &pl)) return TEF_ERR_INVALID;
    if (!flows || !scratch || !loss_out) return tef::fail("tef_smoothing_forward: null pointer"), TEF_ERR_INVALID;
    hipStream_t st = (hipStream_t)stream;
    double *spart = (double *)scratch;
    double *terms = (double *)((char *)scratch + pl.off_terms);
    bool do_s = spat_weight >= 0.0f, do_t = temp_weight >= 0.0f && pl.nt > 0;
    if (do_s) {
        { tef::ProfScope ps(tef::PROF_SMOOTH_FWD, st); hipLaunchKernelGGL(spatial_fwd_kernel, dim3(pl.sblocks, pl.nmaps), dim3(256), 0, st, pl.d, flows, spart); }
        if (int rc = tef::check_launch("spatial_fwd_kernel")) return rc;
    }
    if (do_t) {
        { tef::ProfScope ps(tef::PROF_SMOOTH_FWD, st); hipLaunchKernelGGL(temporal_fwd_kernel, dim3(pl.nt), dim3(256), 0, st, pl.d, flows, terms); }
        if (int rc = tef::check_launch("temporal_fwd_kernel")) return rc;
    }
    if (do_s || do_t) {
        { tef::ProfScope ps(tef::PROF_SMOOTH_FWD, st); hipLaunchKernelGGL(smoothing_reduce_kernel, dim3(1), dim3(256), 0, st, pl.d, spart, do_s ? pl.ns : 0,
                           spat_weight, terms, do_t ? pl.nt : 0, temp_weight, loss_out); }
        if (int rc = tef::check_launch("smoothing_reduce_kernel")) return rc;
    }
    return 0;
}

int tef_smoothing_backward(const tef_loss_cfg *cfg, const float *flows, float spat_weight, float temp_weight,
                           void *scratch, const float *grad_out, float *dflows, void *stream)
{
    Plan pl;
    if (!make_plan(cfg, &pl)) return TEF_ERR_INVALID;
    if (!flows || !scratch || !grad_out || !dflows)
        return tef::fail("tef_smoothing_backward: null pointer"), TEF_ERR_INVALID;
    hipStream_t st = (hipStream_t)stream;
    double *terms = (double *)((char *)scratch + pl.off_terms);
    if (spat_weight >= 0.0f) {
        { tef::ProfScope ps(tef::PROF_SMOOTH_BWD, st); hipLaunchKernelGGL(spatial_bwd_kernel, dim3(pl.sblocks, pl.nmaps), dim3(256), 0, st, pl.d, flows, spat_weight,
                           grad_out, dflows); }
        if (int rc = tef::check_launch("spatial_bwd_kernel")) return rc;
    }
    if (temp_weight >= 0.0f && pl.nt > 0) {
        { tef::ProfScope ps(tef::PROF_SMOOTH_BWD, st); hipLaunchKernelGGL(temporal_bwd_kernel, dim3(pl.sblocks, pl.nt), dim3(256), 0, st, pl.d, flows, temp_weight,
                           terms, grad_out, dflows); }
        if (int rc = tef::check_launch("temporal_bwd_kernel")) return rc;
    }
    return 0;
}

}  // extern "C"
