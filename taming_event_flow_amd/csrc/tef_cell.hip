// tef_cell.hip — the fused ConvGRU cell of RecEVFlowNet (reference models/submodules.py:111-152) and the element-wise
// pieces of the network's hand-written backward pass.
//
//   forward   update | reset = sigmoid(conv([x, h]))            one GEMM with 2C output rows, two output tensors
//             out_inputs     = tanh(conv([x, h * reset]))       the product is formed inside the patch loads
//             new_state      = h * (1 - update) + out_inputs * update      in the epilogue of that same GEMM
//
//   backward  every gradient that has several consumers is SUMMED WHERE IT IS CONSUMED (the kernels below take up to four
//             addend pointers), never accumulated by a separate pass, and the gate pre-activation gradients are formed
//             together with the bias gradients in the same sweep:
//     A  dhn = sum of sources; g_o = dhn * u * (1 - o^2); g_u = dhn * (o - h) * u (1 - u); dh = dhn * (1 - u)
//        input-gradient GEMM of the out gate on g_o        -> dx_a, dxg (gradient of h * r)
//     B  g_r = dxg * h * r (1 - r); dh += dxg * r
//        input-gradient GEMM of the update | reset gates on [g_u | g_r]  -> dx_b, dh_c
//     C  dx = dx_a + dx_b; dh += dh_c
//   g_o and [g_u | g_r] stay in caller-owned buffers: the weight gradients are either formed here or, in a BPTT window,
//   by one tef_conv_wgrad_parts reduction per layer over all passes.
#include <hip/hip_runtime.h>
#include <stdint.h>

#include <algorithm>

#include "tef.h"
#include "tef_common.h"

namespace {

constexpr int kMaxSrc = 4;
struct Sources {
    const float *p[kMaxSrc];
    int n;
};

__device__ __forceinline__ float4 ld4(const float *p) { return *reinterpret_cast<const float4 *>(p); }
__device__ __forceinline__ void st4(float *p, float4 v) { *reinterpret_cast<float4 *>(p) = v; }
__device__ __forceinline__ float4 add4(float4 a, float4 b) { return make_float4(a.x + b.x, a.y + b.y, a.z + b.z, a.w + b.w); }

__device__ __forceinline__ float4 sum_sources(const Sources &s, size_t i)
{
    float4 v = ld4(s.p[0] + i);
    for (int k = 1; k < s.n; ++k) v = add4(v, ld4(s.p[k] + i));
    return v;
}
__device__ __forceinline__ float sum_sources1(const Sources &s, size_t i)
{
    float v = s.p[0][i];
    for (int k = 1; k < s.n; ++k) v += s.p[k][i];
    return v;
}

// workgroup sum of `local` (256 threads) added to *dst by one atomic
__device__ __forceinline__ void block_add(float local, float *dst, float *red)
{
    for (int s = 32; s > 0; s >>= 1) local += __shfl_down(local, s, 64);
    if ((threadIdx.x & 63) == 0) red[threadIdx.x >> 6] = local;
    __syncthreads();
    if (threadIdx.x == 0 && dst) atomicAdd(dst, (red[0] + red[1]) + (red[2] + red[3]));
    __syncthreads();
}

__device__ __forceinline__ float act_grad(float g, float y, int act)
{
    if (act == TEF_ACT_RELU) return y > 0.0f ? g : 0.0f;
    if (act == TEF_ACT_TANH) return g * (1.0f - y * y);
    if (act == TEF_ACT_SIGMOID) return g * (y * (1.0f - y));
    return g;
}

// Tensors are [B][C][HW]; one block row (blockIdx.y) per channel, so the bias gradient is a per-row sum.
// Element loop of channel n: q walks the B * HW elements of the channel, 4 at a time when HW % 4 == 0.
template <class F4, class F1>
__device__ __forceinline__ void channel_loop(int B, int HW, F4 body4, F1 body1)
{
    if ((HW & 3) == 0) {
        const int Q = (B * HW) >> 2;
        for (int q = blockIdx.x * blockDim.x + threadIdx.x; q < Q; q += gridDim.x * blockDim.x) {
            const int m = q << 2, b = m / HW;
            body4(b, m - b * HW);
        }
    } else {
        const int M = B * HW;
        for (int m = blockIdx.x * blockDim.x + threadIdx.x; m < M; m += gridDim.x * blockDim.x) {
            const int b = m / HW;
            body1(b, m - b * HW);
        }
    }
}

// g = (sum of dy sources) * act'(out);  dbias[n] += sum g        (any conv layer's pre-activation gradient)
__global__ __launch_bounds__(256) void grad_act_kernel(Sources dy, const float *__restrict__ out, int act, int B, int C,
                                                       int HW, float *__restrict__ g, float *__restrict__ dbias)
{
    __shared__ float red[4];
    const int n = blockIdx.y;
    float local = 0.0f;
    channel_loop(B, HW,
        [&](int b, int p) {
            const size_t i = ((size_t)b * C + n) * HW + p;
            float4 v = sum_sources(dy, i);
            if (act != TEF_ACT_NONE) {
                float4 y = ld4(out + i);
                v = make_float4(act_grad(v.x, y.x, act), act_grad(v.y, y.y, act), act_grad(v.z, y.z, act), act_grad(v.w, y.w, act));
            }
            st4(g + i, v);
            local += (v.x + v.y) + (v.z + v.w);
        },
        [&](int b, int p) {
            const size_t i = ((size_t)b * C + n) * HW + p;
            float v = sum_sources1(dy, i);
            if (act != TEF_ACT_NONE) v = act_grad(v, out[i], act);
            g[i] = v;
            local += v;
        });
    if (dbias) block_add(local, dbias + n, red);
}

// cell backward, step A
__global__ __launch_bounds__(256) void cell_bwd_a_kernel(Sources dhn, const float *__restrict__ h, const float *__restrict__ u,
                                                         const float *__restrict__ o, int B, int C, int HW,
                                                         float *__restrict__ g_ur, float *__restrict__ g_o,
                                                         float *__restrict__ dh, float *__restrict__ db_u,
                                                         float *__restrict__ db_o)
{
    __shared__ float red[4];
    const int n = blockIdx.y;
    float su = 0.0f, so = 0.0f;
    channel_loop(B, HW,
        [&](int b, int p) {
            const size_t i = ((size_t)b * C + n) * HW + p, iu = ((size_t)b * 2 * C + n) * HW + p;
            float4 d = sum_sources(dhn, i), hv = ld4(h + i), uv = ld4(u + i), ov = ld4(o + i);
            float4 go = make_float4((d.x * uv.x) * (1.0f - ov.x * ov.x), (d.y * uv.y) * (1.0f - ov.y * ov.y),
                                    (d.z * uv.z) * (1.0f - ov.z * ov.z), (d.w * uv.w) * (1.0f - ov.w * ov.w));
            float4 gu = make_float4((d.x * (ov.x - hv.x)) * (uv.x * (1.0f - uv.x)), (d.y * (ov.y - hv.y)) * (uv.y * (1.0f - uv.y)),
                                    (d.z * (ov.z - hv.z)) * (uv.z * (1.0f - uv.z)), (d.w * (ov.w - hv.w)) * (uv.w * (1.0f - uv.w)));
            st4(g_o + i, go);
            st4(g_ur + iu, gu);
            st4(dh + i, make_float4(d.x * (1.0f - uv.x), d.y * (1.0f - uv.y), d.z * (1.0f - uv.z), d.w * (1.0f - uv.w)));
            so += (go.x + go.y) + (go.z + go.w);
            su += (gu.x + gu.y) + (gu.z + gu.w);
        },
        [&](int b, int p) {
            const size_t i = ((size_t)b * C + n) * HW + p, iu = ((size_t)b * 2 * C + n) * HW + p;
            float d = sum_sources1(dhn, i), hv = h[i], uv = u[i], ov = o[i];
            float go = (d * uv) * (1.0f - ov * ov), gu = (d * (ov - hv)) * (uv * (1.0f - uv));
            g_o[i] = go;
            g_ur[iu] = gu;
            dh[i] = d * (1.0f - uv);
            so += go;
            su += gu;
        });
    block_add(so, db_o ? db_o + n : nullptr, red);
    block_add(su, db_u ? db_u + n : nullptr, red);
}

// cell backward, step B: dxg = gradient of (h * r)
__global__ __launch_bounds__(256) void cell_bwd_b_kernel(const float *__restrict__ dxg, const float *__restrict__ h,
                                                         const float *__restrict__ r, int B, int C, int HW,
                                                         float *__restrict__ g_ur, float *__restrict__ dh,
                                                         float *__restrict__ db_r)
{
    __shared__ float red[4];
    const int n = blockIdx.y;
    float sr = 0.0f;
    channel_loop(B, HW,
        [&](int b, int p) {
            const size_t i = ((size_t)b * C + n) * HW + p, ir = ((size_t)b * 2 * C + C + n) * HW + p;
            float4 d = ld4(dxg + i), hv = ld4(h + i), rv = ld4(r + i), a = ld4(dh + i);
            float4 gr = make_float4((d.x * hv.x) * (rv.x * (1.0f - rv.x)), (d.y * hv.y) * (rv.y * (1.0f - rv.y)),
                                    (d.z * hv.z) * (rv.z * (1.0f - rv.z)), (d.w * hv.w) * (rv.w * (1.0f - rv.w)));
            st4(g_ur + ir, gr);
            st4(dh + i, make_float4(a.x + d.x * rv.x, a.y + d.y * rv.y, a.z + d.z * rv.z, a.w + d.w * rv.w));
            sr += (gr.x + gr.y) + (gr.z + gr.w);
        },
        [&](int b, int p) {
            const size_t i = ((size_t)b * C + n) * HW + p, ir = ((size_t)b * 2 * C + C + n) * HW + p;
            float d = dxg[i], hv = h[i], rv = r[i];
            float gr = (d * hv) * (rv * (1.0f - rv));
            g_ur[ir] = gr;
            dh[i] += d * rv;
            sr += gr;
        });
    block_add(sr, db_r ? db_r + n : nullptr, red);
}

// step C: dx = dx_a + dx_b and dh += dh_c in one launch (blockIdx.y selects the pair)
__global__ __launch_bounds__(256) void cell_bwd_c_kernel(const float *__restrict__ dx_a, const float *__restrict__ dx_b,
                                                         float *__restrict__ dx, const float *__restrict__ dh_c,
                                                         float *__restrict__ dh, size_t n)
{
    const float *a = blockIdx.y ? dh : dx_a, *b = blockIdx.y ? dh_c : dx_b;
    float *out = blockIdx.y ? dh : dx;
    size_t i = ((size_t)blockIdx.x * blockDim.x + threadIdx.x) * 4;
    if (i + 3 < n) {
        st4(out + i, add4(ld4(a + i), ld4(b + i)));
    } else {
        for (; i < n; ++i) out[i] = a[i] + b[i];
    }
}

// step C with the producer of x folded in (round 6): instead of dx = dx_a + dx_b, the pre-activation gradient of the layer that
// made x — g_x = act'(x) * (dx_a + dx_b), db_x[n] += sum g_x (grad_act_kernel's operations on the same values) — and
// dh += dh_c, in one channel-wise sweep.  In RecEVFlowNet x is the relu output of the level's strided head convolution.
__global__ __launch_bounds__(256) void cell_bwd_c_head_kernel(const float *__restrict__ dx_a, const float *__restrict__ dx_b,
                                                              const float *__restrict__ x, int x_act, float *__restrict__ g_x,
                                                              float *__restrict__ db_x, const float *__restrict__ dh_c,
                                                              float *__restrict__ dh, int B, int C, int HW)
{
    __shared__ float red[4];
    const int n = blockIdx.y;
    float local = 0.0f;
    channel_loop(B, HW,
        [&](int b, int p) {
            const size_t i = ((size_t)b * C + n) * HW + p;
            float4 v = add4(ld4(dx_a + i), ld4(dx_b + i));
            if (x_act != TEF_ACT_NONE) {
                const float4 y = ld4(x + i);
                v = make_float4(act_grad(v.x, y.x, x_act), act_grad(v.y, y.y, x_act), act_grad(v.z, y.z, x_act), act_grad(v.w, y.w, x_act));
            }
            st4(g_x + i, v);
            local += (v.x + v.y) + (v.z + v.w);
            st4(dh + i, add4(ld4(dh + i), ld4(dh_c + i)));
        },
        [&](int b, int p) {
            const size_t i = ((size_t)b * C + n) * HW + p;
            float v = dx_a[i] + dx_b[i];
            if (x_act != TEF_ACT_NONE) v = act_grad(v, x[i], x_act);
            g_x[i] = v;
            local += v;
            dh[i] = dh[i] + dh_c[i];
        });
    if (db_x) block_add(local, db_x + n, red);
}

// out = act(a + b): the residual connection of ResidualBlock (models/submodules.py:219-226)
__global__ __launch_bounds__(256) void add_act_kernel(const float *__restrict__ a, const float *__restrict__ b, int act,
                                                      size_t n, float *__restrict__ out)
{
    size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n) return;
    float v = a[i] + b[i];
    if (act == TEF_ACT_RELU) v = fmaxf(v, 0.0f);
    else if (act == TEF_ACT_TANH) v = tanhf(v);
    else if (act == TEF_ACT_SIGMOID) v = 1.0f / (1.0f + expf(-v));
    out[i] = v;
}

inline tef_conv_desc gate_desc(const tef_gru_desc *d, int rows, int act)
{
    tef_conv_desc c;
    c.B = d->B; c.C0 = d->C; c.C1 = d->C; c.H = d->H; c.W = d->W; c.N = rows; c.ksize = 3; c.stride = 1; c.act = act;
    return c;
}

inline bool gru_ok(const tef_gru_desc *d)
{
    if (!d || d->B < 1 || d->C < 1 || d->H < 1 || d->W < 1) return tef::fail("tef_convgru: bad descriptor");
    return true;
}

inline size_t align256(size_t v) { return (v + 255) & ~(size_t)255; }

inline bool fill_sources(const float *const *ptrs, int n, Sources *s, const char *what)
{
    if (!ptrs || n < 1 || n > kMaxSrc) return tef::fail(what);
    s->n = n;
    for (int k = 0; k < kMaxSrc; ++k) s->p[k] = k < n ? ptrs[k] : nullptr;
    for (int k = 0; k < n; ++k)
        if (!ptrs[k]) return tef::fail(what);
    return true;
}

inline dim3 channel_grid(int B, int C, int HW)
{
    return dim3((unsigned)std::min<size_t>(32, ((size_t)B * HW / 4 + 255) / 256 + 1), (unsigned)C);
}

}  // namespace

extern "C" {

size_t tef_convgru_workspace_bytes(const tef_gru_desc *d)
{
    if (!gru_ok(d)) return 0;
    tef_conv_desc ur = gate_desc(d, 2 * d->C, TEF_ACT_SIGMOID), og = gate_desc(d, d->C, TEF_ACT_TANH);
    size_t act = align256((size_t)d->B * d->C * d->H * d->W * sizeof(float));
    return 4 * act + std::max(tef_conv_workspace_bytes(&ur), tef_conv_workspace_bytes(&og));
}

int tef_convgru_cell_fwd(const tef_gru_desc *d, const float *x, const float *h, const float *wp_ur, const float *wp_o,
                         const float *bias_ur, const float *bias_o, float *u, float *r, float *o, float *hn,
                         void *workspace, size_t workspace_bytes, void *stream)
{
    if (!gru_ok(d)) return TEF_ERR_INVALID;
    if (!x || !h || !wp_ur || !wp_o || !u || !r || !o || !hn || !workspace) return tef::fail("tef_convgru_cell_fwd: null pointer"), TEF_ERR_INVALID;
    if (workspace_bytes < tef_convgru_workspace_bytes(d)) return tef::fail("tef_convgru_cell_fwd: workspace too small"), TEF_ERR_WORKSPACE;
    tef_conv_desc ur = gate_desc(d, 2 * d->C, TEF_ACT_SIGMOID), og = gate_desc(d, d->C, TEF_ACT_TANH);
    // update | reset share their input (submodules.py:146-148): one GEMM, two output tensors
    if (int rc = tef_conv_forward_split(&ur, x, h, nullptr, wp_ur, bias_ur, u, r, d->C, workspace, workspace_bytes, stream)) return rc;
    // out gate on [x, h * r] (the product is formed in the patch loads); its epilogue also writes the new state (:150)
    return tef_conv_forward_blend(&og, x, h, r, wp_o, bias_o, o, nullptr, d->C, h, u, hn, workspace, workspace_bytes, stream);
}

int tef_convgru_cell_bwd(const tef_gru_desc *d, const float *x, const float *h, const float *u, const float *r,
                         const float *o, const float *const *dhn, int ndhn, const float *w2_ur, const float *w2_o,
                         float *g_ur, float *g_o, float *dx, float *dh, float *dw_u, float *dw_r, float *dw_o,
                         float *db_u, float *db_r, float *db_o, void *workspace, size_t workspace_bytes, void *stream)
{
    return tef_convgru_cell_bwd_head(d, x, h, u, r, o, dhn, ndhn, w2_ur, w2_o, g_ur, g_o, dx, dh, dw_u, dw_r, dw_o, db_u, db_r, db_o,
                                     TEF_ACT_NONE, nullptr, nullptr, workspace, workspace_bytes, stream);
}

int tef_convgru_cell_bwd_head(const tef_gru_desc *d, const float *x, const float *h, const float *u, const float *r,
                              const float *o, const float *const *dhn, int ndhn, const float *w2_ur, const float *w2_o,
                              float *g_ur, float *g_o, float *dx, float *dh, float *dw_u, float *dw_r, float *dw_o,
                              float *db_u, float *db_r, float *db_o, int x_act, float *g_x, float *db_x, void *workspace,
                              size_t workspace_bytes, void *stream)
{
    if (!gru_ok(d)) return TEF_ERR_INVALID;
    if (!x || !h || !u || !r || !o || !w2_ur || !w2_o || !g_ur || !g_o || (!dx && !g_x) || !dh || !workspace)
        return tef::fail("tef_convgru_cell_bwd: null pointer"), TEF_ERR_INVALID;
    if ((dw_u != nullptr) != (dw_r != nullptr)) return tef::fail("tef_convgru_cell_bwd: dw_u and dw_r go together"), TEF_ERR_INVALID;
    if (workspace_bytes < tef_convgru_workspace_bytes(d)) return tef::fail("tef_convgru_cell_bwd: workspace too small"), TEF_ERR_WORKSPACE;
    Sources src;
    if (!fill_sources(dhn, ndhn, &src, "tef_convgru_cell_bwd: 1..4 state-gradient sources")) return TEF_ERR_INVALID;
    hipStream_t st = (hipStream_t)stream;
    const int B = d->B, C = d->C, HW = d->H * d->W;
    const size_t n = (size_t)B * C * HW, act = align256(n * sizeof(float));
    char *ws = (char *)workspace;
    float *dx_a = (float *)ws, *dxg = (float *)(ws + act), *dx_b = (float *)(ws + 2 * act), *dh_c = (float *)(ws + 3 * act);
    void *cws = ws + 4 * act;
    const size_t cws_bytes = workspace_bytes - 4 * act;
    tef_conv_desc ur = gate_desc(d, 2 * C, TEF_ACT_NONE), og = gate_desc(d, C, TEF_ACT_NONE);     // gradients arrive pre-formed
    const dim3 grid = channel_grid(B, C, HW);

    hipLaunchKernelGGL(cell_bwd_a_kernel, grid, dim3(256), 0, st, src, h, u, o, B, C, HW, g_ur, g_o, dh, db_u, db_o);
    if (int rc = tef::check_launch("cell_bwd_a_kernel")) return rc;
    if (int rc = tef_conv_backward_keep(&og, x, h, r, w2_o, nullptr, nullptr, g_o, nullptr, C, dx_a, dxg, dw_o, nullptr, nullptr,
                                        nullptr, C, nullptr, cws, cws_bytes, stream)) return rc;
    hipLaunchKernelGGL(cell_bwd_b_kernel, grid, dim3(256), 0, st, dxg, h, r, B, C, HW, g_ur, dh, db_r);
    if (int rc = tef::check_launch("cell_bwd_b_kernel")) return rc;
    if (int rc = tef_conv_backward_keep(&ur, x, h, nullptr, w2_ur, nullptr, nullptr, g_ur, nullptr, 2 * C, dx_b, dh_c, dw_u, dw_r,
                                        nullptr, nullptr, C, nullptr, cws, cws_bytes, stream)) return rc;
    if (g_x) {      // the producer of x folded into the last sweep: its pre-activation gradient instead of dx
        hipLaunchKernelGGL(cell_bwd_c_head_kernel, grid, dim3(256), 0, st, dx_a, dx_b, x, x_act, g_x, db_x, dh_c, dh, B, C, HW);
        return tef::check_launch("cell_bwd_c_head_kernel");
    }
    hipLaunchKernelGGL(cell_bwd_c_kernel, dim3((unsigned)((n / 4 + 255) / 256 + 1), 2), dim3(256), 0, st, dx_a, dx_b, dx, dh_c, dh, n);
    return tef::check_launch("cell_bwd_c_kernel");
}

int tef_grad_act(const float *const *dy, int ndy, const float *out, int act, int B, int C, int HW, float *g, float *dbias,
                 void *stream)
{
    Sources src;
    if (!fill_sources(dy, ndy, &src, "tef_grad_act: 1..4 gradient sources")) return TEF_ERR_INVALID;
    if (!g || B < 1 || C < 1 || HW < 1 || (act != TEF_ACT_NONE && !out)) return tef::fail("tef_grad_act: bad arguments"), TEF_ERR_INVALID;
    hipLaunchKernelGGL(grad_act_kernel, channel_grid(B, C, HW), dim3(256), 0, (hipStream_t)stream, src, out, act, B, C, HW, g, dbias);
    return tef::check_launch("grad_act_kernel");
}

int tef_add_act(const float *a, const float *b, int act, size_t n, float *out, void *stream)
{
    if (!a || !b || !out) return tef::fail("tef_add_act: null pointer"), TEF_ERR_INVALID;
    if (n == 0) return 0;
    hipLaunchKernelGGL(add_act_kernel, dim3((unsigned)((n + 255) / 256)), dim3(256), 0, (hipStream_t)stream, a, b, act, n, out);
    return tef::check_launch("add_act_kernel");
}

}  // extern "C"
