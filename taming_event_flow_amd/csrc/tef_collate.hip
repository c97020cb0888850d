// tef_collate.hip — the loader's per-sample event formatting + grad/detached split + collate, for a whole batch.
//
// Replaces, for B ragged samples at once (reference: one __getitem__ per sample on the host, then custom_collate):
//   dataloader/h5.py:340-345      samples with <= 10 events become empty
//   dataloader/base.py:153-177    event_formatting: ps*2-1, ts = (ts - ts[0]) / (ts[-1] - ts[0])
//   dataloader/base.py:192-222    augment_events: Horizontal / Vertical / Polarity flips
//   dataloader/base.py:252-278    create_list_encoding (ts,y,x,p), create_polarity_mask (p>0, p<0)
//   dataloader/base.py:348-377    split_event_list: sampled indices -> gradient list (in sampled order), the rest ->
//                                 detached list (in stream order)
//   dataloader/base.py:392-434    custom_collate: zero-pad to the longest list of the batch, [B, N, 4] / [B, N, 2]
// Three launches: mark the sampled events, count the unsampled ones per 1024-event chunk, then every chunk places its
// events (prefix over the chunk counts + an in-workgroup scan give the stream-ordered detached slot).  The random
// choice itself (`probs.multinomial`) stays with the caller: indices come in, so results are reproducible bit for bit.
#include <hip/hip_runtime.h>
#include <stdint.h>

#include "tef.h"
#include "tef_common.h"

namespace {

constexpr int kChunk = 1024;     // events per workgroup (= threads)

struct Batch {
    int off[TEF_MAX_BATCH + 1];  // raw event range of sample b: [off[b], off[b+1])
    int flags[TEF_MAX_BATCH];    // TEF_AUG_* bits
    int B, G, N, Nd, H, W, nchunk;
};

// sample b: n raw events, ng of them to the gradient list (ng < n  <=>  split)
__device__ __forceinline__ void counts(const Batch &bt, int b, int &n, int &ng)
{
    n = bt.off[b + 1] - bt.off[b];
    if (n <= 10) n = 0;                                  // h5.py:340-345
    ng = (bt.G > 0 && n > bt.G) ? bt.G : n;              // base.py:362
}

__global__ void mark_kernel(Batch bt, const int *__restrict__ sampled, int *__restrict__ slot)
{
    int b = blockIdx.y, j = blockIdx.x * blockDim.x + threadIdx.x;
    int n, ng;
    counts(bt, b, n, ng);
    if (ng == n || j >= ng) return;
    int i = sampled[(size_t)b * bt.G + j];
    if (i >= 0 && i < n) slot[bt.off[b] + i] = j + 1;    // event i goes to gradient slot j
}

__global__ __launch_bounds__(kChunk) void count_kernel(Batch bt, const int *__restrict__ slot, int *__restrict__ chunk_cnt)
{
    int b = blockIdx.y, c = blockIdx.x;
    int n, ng;
    counts(bt, b, n, ng);
    int i = c * kChunk + threadIdx.x;
    int un = (ng < n && i < n && slot[bt.off[b] + i] == 0) ? 1 : 0;
    unsigned long long m = __ballot(un);
    __shared__ int wsum[kChunk / 64];
    if ((threadIdx.x & 63) == 0) wsum[threadIdx.x >> 6] = __popcll(m);
    __syncthreads();
    if (threadIdx.x == 0) {
        int s = 0;
        for (int w = 0; w < kChunk / 64; ++w) s += wsum[w];
        chunk_cnt[(size_t)b * bt.nchunk + c] = s;
    }
}

__global__ __launch_bounds__(kChunk) void place_kernel(Batch bt, const float *__restrict__ xs, const float *__restrict__ ys,
                                                       const float *__restrict__ ts, const float *__restrict__ ps,
                                                       const int *__restrict__ slot, const int *__restrict__ chunk_cnt,
                                                       float4 *__restrict__ ev, float2 *__restrict__ pm,
                                                       float4 *__restrict__ dev, float2 *__restrict__ dpm)
{
    int b = blockIdx.y, c = blockIdx.x;
    int n, ng;
    counts(bt, b, n, ng);
    if (c * kChunk >= n) return;
    bool split = ng < n;
    int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    int i = c * kChunk + threadIdx.x;
    int base = bt.off[b];
    int s = (split && i < n) ? slot[base + i] : 0;
    int un = (split && i < n && s == 0) ? 1 : 0;
    // detached slot = unsampled events before this one: earlier chunks + earlier waves + earlier lanes
    __shared__ int wsum[kChunk / 64];
    __shared__ int before;
    unsigned long long m = __ballot(un);
    if (lane == 0) wsum[wave] = __popcll(m);
    if (threadIdx.x == 0) before = 0;
    __syncthreads();
    if (split) {
        int part = 0;
        for (int k = threadIdx.x; k < c; k += kChunk) part += chunk_cnt[(size_t)b * bt.nchunk + k];
        if (part) atomicAdd(&before, part);
    }
    __syncthreads();
    if (i >= n) return;
    int dslot = before + __popcll(m & ((1ull << lane) - 1ull));
    for (int w = 0; w < wave; ++w) dslot += wsum[w];

    // event_formatting + augment_events, in the reference's fp32 op order
    float t0 = ts[base], t1 = ts[base + (bt.off[b + 1] - bt.off[b]) - 1];
    float t = (ts[base + i] - t0) / (t1 - t0);                    // base.py:175
    float p = ps[base + i] * 2.0f - 1.0f;                          // :173
    float x = xs[base + i], y = ys[base + i];
    int fl = bt.flags[b];
    if (fl & TEF_AUG_HORIZONTAL) x = (float)(bt.W - 1) - x;       // :208
    if (fl & TEF_AUG_VERTICAL) y = (float)(bt.H - 1) - y;         // :214
    if (fl & TEF_AUG_POLARITY) p = -p;                            // :220
    float4 e = make_float4(t, y, x, p);                           // :263 (ts, ys, xs, ps)
    float2 k = make_float2(p > 0.0f ? 1.0f : 0.0f, p < 0.0f ? 1.0f : 0.0f);   // :272-278
    if (!split) {
        ev[(size_t)b * bt.N + i] = e;
        pm[(size_t)b * bt.N + i] = k;
    } else if (s > 0) {
        ev[(size_t)b * bt.N + (s - 1)] = e;
        pm[(size_t)b * bt.N + (s - 1)] = k;
    } else {
        dev[(size_t)b * bt.Nd + dslot] = e;
        dpm[(size_t)b * bt.Nd + dslot] = k;
    }
}

}  // namespace

extern "C" int tef_collate_counts(const int *offsets, int B, int max_grad, int *n_grad, int *n_detached)
{
    if (!offsets || !n_grad || !n_detached || B < 1 || B > TEF_MAX_BATCH || max_grad < 0)
        return tef::fail("tef_collate_counts: bad arguments"), TEF_ERR_INVALID;
    int N = 0, Nd = 0;
    for (int b = 0; b < B; ++b) {
        int n = offsets[b + 1] - offsets[b];
        if (n < 0) return tef::fail("tef_collate_counts: offsets must not decrease"), TEF_ERR_INVALID;
        if (n <= 10) n = 0;
        int ng = (max_grad > 0 && n > max_grad) ? max_grad : n;
        if (ng > N) N = ng;
        if (n - ng > Nd) Nd = n - ng;
    }
    *n_grad = N;
    *n_detached = Nd;
    return 0;
}

extern "C" size_t tef_collate_workspace_bytes(const int *offsets, int B)
{
    if (!offsets || B < 1 || B > TEF_MAX_BATCH) return 0;
    size_t total = (size_t)(offsets[B] - offsets[0]);
    int longest = 0;
    for (int b = 0; b < B; ++b) longest = offsets[b + 1] - offsets[b] > longest ? offsets[b + 1] - offsets[b] : longest;
    size_t nchunk = (size_t)(longest + kChunk - 1) / kChunk;
    return (total + (size_t)B * nchunk + 16) * sizeof(int);
}

extern "C" int tef_collate_events(const float *xs, const float *ys, const float *ts, const float *ps,
                                  const int *offsets, const int *sampled, const int *flags, int B, int max_grad, int N,
                                  int Nd, int H, int W, void *workspace, size_t workspace_bytes, float *event_list,
                                  float *pol_mask, float *d_event_list, float *d_pol_mask, void *stream)
{
    if (!offsets || B < 1 || B > TEF_MAX_BATCH || max_grad < 0 || N < 0 || Nd < 0 || H < 1 || W < 1)
        return tef::fail("tef_collate_events: bad arguments"), TEF_ERR_INVALID;
    int needN = 0, needNd = 0;
    if (tef_collate_counts(offsets, B, max_grad, &needN, &needNd) != 0) return TEF_ERR_INVALID;
    if (N < needN || Nd < needNd) return tef::fail("tef_collate_events: N / Nd smaller than tef_collate_counts"), TEF_ERR_INVALID;
    if (offsets[0] != 0) return tef::fail("tef_collate_events: offsets[0] must be 0"), TEF_ERR_INVALID;
    int total = offsets[B];
    if ((N > 0 && (!event_list || !pol_mask)) || (Nd > 0 && (!d_event_list || !d_pol_mask)) ||
        (total > 0 && (!xs || !ys || !ts || !ps)) || (needNd > 0 && !sampled))
        return tef::fail("tef_collate_events: null buffer"), TEF_ERR_INVALID;
    if (workspace_bytes < tef_collate_workspace_bytes(offsets, B) || (total > 0 && !workspace))
        return tef::fail("tef_collate_events: workspace too small"), TEF_ERR_INVALID;
    hipStream_t st = (hipStream_t)stream;
    Batch bt;
    int longest = 0;
    for (int b = 0; b < B; ++b) {
        bt.off[b] = offsets[b];
        bt.flags[b] = flags ? flags[b] : 0;
        int n = offsets[b + 1] - offsets[b];
        if (n > longest) longest = n;
    }
    bt.off[B] = offsets[B];
    bt.B = B; bt.G = max_grad; bt.N = N; bt.Nd = Nd; bt.H = H; bt.W = W;
    bt.nchunk = (longest + kChunk - 1) / kChunk;
    hipError_t e = hipSuccess;
    // zero padding of custom_collate (base.py:418-420)
    if (N > 0) {
        e = hipMemsetAsync(event_list, 0, (size_t)B * N * 4 * sizeof(float), st);
        if (e == hipSuccess) e = hipMemsetAsync(pol_mask, 0, (size_t)B * N * 2 * sizeof(float), st);
    }
    if (e == hipSuccess && Nd > 0) {
        e = hipMemsetAsync(d_event_list, 0, (size_t)B * Nd * 4 * sizeof(float), st);
        if (e == hipSuccess) e = hipMemsetAsync(d_pol_mask, 0, (size_t)B * Nd * 2 * sizeof(float), st);
    }
    if (e != hipSuccess) return tef::fail_hip("hipMemsetAsync", e), TEF_ERR_LAUNCH;
    if (total == 0 || bt.nchunk == 0) return 0;
    int *slot = (int *)workspace;
    int *chunk_cnt = slot + total;
    dim3 grid((unsigned)bt.nchunk, (unsigned)B);
    if (needNd > 0) {
        e = hipMemsetAsync(slot, 0, (size_t)total * sizeof(int), st);
        if (e != hipSuccess) return tef::fail_hip("hipMemsetAsync", e), TEF_ERR_LAUNCH;
        hipLaunchKernelGGL(mark_kernel, dim3((unsigned)((max_grad + 255) / 256), (unsigned)B), dim3(256), 0, st, bt,
                           sampled, slot);
        hipLaunchKernelGGL(count_kernel, grid, dim3(kChunk), 0, st, bt, (const int *)slot, chunk_cnt);
    }
    hipLaunchKernelGGL(place_kernel, grid, dim3(kChunk), 0, st, bt, xs, ys, ts, ps, (const int *)slot,
                       (const int *)chunk_cnt, (float4 *)event_list, (float2 *)pol_mask, (float4 *)d_event_list,
                       (float2 *)d_pol_mask);
    return tef::check_launch("collate kernels");
}
