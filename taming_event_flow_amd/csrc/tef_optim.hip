// tef_optim.hip — the optimiser step of the training window on the FLAT parameter / gradient buffers: global-norm
// clipping + Adam + zero_grad in three launches (reference train_flow.py:127-131: clip_grad_norm_, optimizer.step(),
// optimizer.zero_grad() — ~12 ATen launches over 60 tensors, 1 ms of a 38 ms window).
#include <hip/hip_runtime.h>
#include <stdint.h>

#include "tef.h"
#include "tef_common.h"

namespace {

constexpr int kNormBlocks = 1024;

__global__ __launch_bounds__(256) void sumsq_partial_kernel(const float *__restrict__ x, size_t n, double *__restrict__ part)
{
    __shared__ double red[4];
    double acc = 0.0;
    const size_t n4 = n >> 2, stride = (size_t)gridDim.x * blockDim.x;
    const float4 *x4 = reinterpret_cast<const float4 *>(x);
    for (size_t k = (size_t)blockIdx.x * blockDim.x + threadIdx.x; k < n4; k += stride) {
        const float4 v = x4[k];
        acc += (double)v.x * v.x + (double)v.y * v.y + (double)v.z * v.z + (double)v.w * v.w;
    }
    if (blockIdx.x == 0 && threadIdx.x < (n & 3)) {
        const float v = x[(n4 << 2) + threadIdx.x];
        acc += (double)v * v;
    }
    for (int sft = 32; sft > 0; sft >>= 1) acc += __shfl_down(acc, sft, 64);
    if ((threadIdx.x & 63) == 0) red[threadIdx.x >> 6] = acc;
    __syncthreads();
    if (threadIdx.x == 0) part[blockIdx.x] = (red[0] + red[1]) + (red[2] + red[3]);
}

// ||x||_2 from the partial sums (fixed order), and the step counter of the update that follows
__global__ __launch_bounds__(256) void norm_final_kernel(const double *__restrict__ part, int nparts, float *__restrict__ norm,
                                                         float *__restrict__ step)
{
    __shared__ double red[256];
    double acc = 0.0;
    for (int k = threadIdx.x; k < nparts; k += blockDim.x) acc += part[k];
    red[threadIdx.x] = acc;
    __syncthreads();
    for (int s = blockDim.x >> 1; s > 0; s >>= 1) {
        if ((int)threadIdx.x < s) red[threadIdx.x] += red[threadIdx.x + s];
        __syncthreads();
    }
    if (threadIdx.x == 0) {
        norm[0] = (float)sqrt(red[0]);
        if (step) step[0] += 1.0f;
    }
}

// torch.optim.Adam (amsgrad=False, weight_decay=0, maximize=False) on flat buffers, gradient clipped by `scale` first and
// cleared afterwards.  Op order of torch's _single_tensor_adam: lerp, mul + addcmul, sqrt / sqrt(bc2) + eps, addcdiv.
// Hyper-parameters: by value (hp == nullptr), or read from DEVICE memory, hp = {lr, beta1, beta2, eps, max_norm} as doubles:
// a window captured in a hipGraph then follows a learning-rate schedule without being captured again (the host refreshes
// the five numbers before a replay).
__global__ __launch_bounds__(256) void adam_clip_kernel(float *__restrict__ p, float *__restrict__ g, float *__restrict__ m,
                                                        float *__restrict__ v, size_t n, const float *__restrict__ norm,
                                                        float max_norm, double lr, double b1d, double b2d, double epsd,
                                                        const double *__restrict__ hp, const float *__restrict__ step)
{
    if (hp) {
        lr = hp[0]; b1d = hp[1]; b2d = hp[2]; epsd = hp[3];
        max_norm = (float)hp[4];
    }
    const float t = step[0];
    float scale = 1.0f;
    if (max_norm > 0.0f) {
        // clip_grad_norm_: clamp(max_norm / (norm + 1e-6), max=1).  torch's clamp PROPAGATES a NaN norm (every parameter is
        // poisoned and the failure is loud); fminf would return 1 and update with the finite elements of a broken gradient
        const float s = max_norm / (norm[0] + 1e-6f);
        scale = (s < 1.0f || s != s) ? s : 1.0f;
    }
    // bias corrections as torch's _single_tensor_adam forms them for a host step count: Python floats (double) —
    // 1 - beta ** step, lr / bc1, sqrt(bc2) — rounded to fp32 only where they enter the tensor arithmetic
    const double bc1 = 1.0 - pow(b1d, (double)t), bc2 = 1.0 - pow(b2d, (double)t);
    const float step_size = (float)(lr / bc1), bc2_sqrt = (float)sqrt(bc2);
    const float w1 = (float)(1.0 - b1d), b2 = (float)b2d, w2 = (float)(1.0 - b2d), eps = (float)epsd;
    const size_t n4 = n >> 2, stride = (size_t)gridDim.x * blockDim.x;
    auto one = [&](float &pp, float &gg, float &mm, float &vv) {
        const float gr = gg * scale;
        mm = mm + w1 * (gr - mm);                                // exp_avg.lerp_(grad, 1 - beta1)
        vv = vv * b2 + (w2 * gr) * gr;                           // exp_avg_sq.mul_(beta2).addcmul_(grad, grad, value=1 - beta2): (value * t1) * t2
        const float denom = sqrtf(vv) / bc2_sqrt + eps;
        pp = pp - step_size * (mm / denom);                      // param.addcdiv_(exp_avg, denom, value=-step_size)
        gg = 0.0f;                                               // zero_grad
    };
    float4 *p4 = reinterpret_cast<float4 *>(p), *g4 = reinterpret_cast<float4 *>(g), *m4 = reinterpret_cast<float4 *>(m),
           *v4 = reinterpret_cast<float4 *>(v);
    for (size_t k = (size_t)blockIdx.x * blockDim.x + threadIdx.x; k < n4; k += stride) {
        float4 a = p4[k], b = g4[k], c = m4[k], d = v4[k];
        one(a.x, b.x, c.x, d.x); one(a.y, b.y, c.y, d.y); one(a.z, b.z, c.z, d.z); one(a.w, b.w, c.w, d.w);
        p4[k] = a; g4[k] = b; m4[k] = c; v4[k] = d;
    }
    if (blockIdx.x == 0 && threadIdx.x < (n & 3)) {
        const size_t k = (n4 << 2) + threadIdx.x;
        one(p[k], g[k], m[k], v[k]);
    }
}

}  // namespace

extern "C" {

size_t tef_l2_norm_scratch_bytes(void) { return kNormBlocks * sizeof(double); }

int tef_l2_norm(const float *x, size_t n, void *scratch, float *out, float *step, void *stream)
{
    if (!x || !scratch || !out) return tef::fail("tef_l2_norm: null pointer"), TEF_ERR_INVALID;
    if ((uintptr_t)x & 15) return tef::fail("tef_l2_norm: buffer must be 16-byte aligned"), TEF_ERR_INVALID;
    hipStream_t st = (hipStream_t)stream;
    hipLaunchKernelGGL(sumsq_partial_kernel, dim3(kNormBlocks), dim3(256), 0, st, x, n, (double *)scratch);
    hipLaunchKernelGGL(norm_final_kernel, dim3(1), dim3(256), 0, st, (const double *)scratch, kNormBlocks, out, step);
    return tef::check_launch("norm kernels");
}

int tef_adam_clip_step(float *p, float *g, float *m, float *v, size_t n, const float *norm, float max_norm, double lr,
                       double beta1, double beta2, double eps, const float *step, void *stream)
{
    if (!p || !g || !m || !v || !norm || !step) return tef::fail("tef_adam_clip_step: null pointer"), TEF_ERR_INVALID;
    if (((uintptr_t)p | (uintptr_t)g | (uintptr_t)m | (uintptr_t)v) & 15)
        return tef::fail("tef_adam_clip_step: buffers must be 16-byte aligned"), TEF_ERR_INVALID;
    const size_t n4 = n >> 2;
    unsigned blocks = (unsigned)((n4 + 255) / 256);
    if (blocks > 4096) blocks = 4096;
    if (blocks < 1) blocks = 1;
    hipLaunchKernelGGL(adam_clip_kernel, dim3(blocks), dim3(256), 0, (hipStream_t)stream, p, g, m, v, n, norm, max_norm, lr, beta1,
                       beta2, eps, (const double *)nullptr, step);
    return tef::check_launch("adam_clip_kernel");
}

int tef_adam_clip_step_hp(float *p, float *g, float *m, float *v, size_t n, const float *norm, const double *hp,
                          const float *step, void *stream)
{
    if (!p || !g || !m || !v || !norm || !step || !hp) return tef::fail("tef_adam_clip_step_hp: null pointer"), TEF_ERR_INVALID;
    if (((uintptr_t)p | (uintptr_t)g | (uintptr_t)m | (uintptr_t)v) & 15)
        return tef::fail("tef_adam_clip_step_hp: buffers must be 16-byte aligned"), TEF_ERR_INVALID;
    const size_t n4 = n >> 2;
    unsigned blocks = (unsigned)((n4 + 255) / 256);
    if (blocks > 4096) blocks = 4096;
    if (blocks < 1) blocks = 1;
    hipLaunchKernelGGL(adam_clip_kernel, dim3(blocks), dim3(256), 0, (hipStream_t)stream, p, g, m, v, n, norm, 0.0f, 0.0, 0.0, 0.0,
                       0.0, hp, step);
    return tef::check_launch("adam_clip_kernel");
}

}  // extern "C"
