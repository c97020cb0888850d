// tef_encode.hip — event-count / voxel-grid input representations (dataloader/encodings.py).
//   events_to_image    :8-29   img[y, x] += p                         (index_put_, accumulate=True)
//   events_to_channels :59-81  per-polarity counts (both positive)      -> [2, H, W]
//   events_to_voxel    :32-56  temporal-bilinear voxel grid (signed)    -> [bins, H, W]
// One workgroup per (sample, output channel[, row band]) keeps its channel in LDS as fp64 (ds_add_f64, the fast
// LDS float atomic on gfx950) and streams the event list once; counts are exact, voxel sums are independent of
// the event order to far below fp32 resolution.  Works on one sample (the reference's call shape) or on a
// zero-padded batch [B, N, 4] straight from the collate (the batched form the DP loop uses).
#include <hip/hip_runtime.h>
#include <stdint.h>

#include "tef.h"
#include "tef_common.h"

namespace {

constexpr size_t kLdsBudget = 144 * 1024;
constexpr int kThreads = 1024;

// xs/ys/ts/ps: element e of sample b at base[b * bs + e * es]; an optional second AoS list [B][N2][4] (ts,y,x,p)
// follows the first (the detached events of the collated batch)
__global__ __launch_bounds__(kThreads) void encode_kernel(const float *__restrict__ xs, const float *__restrict__ ys,
                                                          const float *__restrict__ ts, const float *__restrict__ ps,
                                                          long bs, int es, int N, const float *__restrict__ list2, int N2,
                                                          int mode, int C, int H, int W, int rows_per_band, int nbands,
                                                          float *__restrict__ out)
{
    extern __shared__ double img[];
    int bid = blockIdx.x;
    int band = bid % nbands;
    bid /= nbands;
    int c = bid % C, b = bid / C;
    int r0 = band * rows_per_band, r1 = min(H, r0 + rows_per_band);
    int npx = (r1 - r0) * W;
    for (int p = threadIdx.x; p < npx; p += blockDim.x) img[p] = 0.0;
    __syncthreads();
    // One event's contribution to this (channel, band).
    auto add = [&](float p, float t01, float y, float x) {
        float v;
        if (mode == TEF_ENCODE_IMAGE) {
            v = p;
        } else if (mode == TEF_ENCODE_CHANNELS) {
            // mask_pos = {p<0: 0, p>0: 1, else p}; mask_neg = {p>0: 0, p<0: -1, else p}   (encodings.py:72-80)
            float mpos = p > 0.0f ? 1.0f : (p < 0.0f ? 0.0f : p);
            float mneg = p < 0.0f ? -1.0f : (p > 0.0f ? 0.0f : p);
            v = p * (c == 0 ? mpos : mneg);
        } else {
            float t = t01 * (float)(C - 1);                              // encodings.py:47
            v = p * fmaxf(0.0f, 1.0f - fabsf(t - (float)c));            // :52
        }
        if (v == 0.0f) return;
        int iy = (int)y, ix = (int)x;                                    // .long() truncation (:24-27)
        if (iy < 0) iy += H;                                             // python-style negative index
        if (ix < 0) ix += W;
        if (iy < r0 || iy >= r1 || ix < 0 || ix >= W) return;
        atomicAdd(img + (iy - r0) * W + ix, (double)v);
    };
    // A workgroup streams every event of its sample.  The stream is bound by the address rate of the CU's load unit:
    // (ts, y, x, p) lists are read as one 16-byte load per event instead of four strided 4-byte loads.
    auto stream_aos = [&](const float4 *l, int n) {
        for (int e = threadIdx.x; e < n; e += blockDim.x) {
            float4 v = l[e];
            add(v.w, v.x, v.y, v.z);
        }
    };
    if (N > 0) {
        if (es == 4 && ts && ys == ts + 1 && xs == ts + 2 && ps == ts + 3) {
            stream_aos(reinterpret_cast<const float4 *>(ts + (size_t)b * bs), N);
        } else {
            const float *bx = xs + (size_t)b * bs, *by = ys + (size_t)b * bs, *bp = ps + (size_t)b * bs;
            const float *bt = ts ? ts + (size_t)b * bs : bp;
            for (int e = threadIdx.x; e < N; e += blockDim.x) {
                size_t o = (size_t)e * es;
                add(bp[o], bt[o], by[o], bx[o]);
            }
        }
    }
    if (N2 > 0) stream_aos(reinterpret_cast<const float4 *>(list2 + (size_t)b * N2 * 4), N2);     // (ts, y, x, p)
    __syncthreads();
    float *o = out + ((size_t)b * C + c) * (size_t)(H * W) + (size_t)r0 * W;
    for (int p = threadIdx.x; p < npx; p += blockDim.x) o[p] = (float)img[p];
}

}  // namespace

static int encode(const float *xs, const float *ys, const float *ts, const float *ps, int B, long batch_stride,
                  int elem_stride, int N, const float *list2, int N2, int mode, int channels, int H, int W, float *out,
                  void *stream)
{
    if ((N > 0 && (!xs || !ys || !ps)) || (N2 > 0 && !list2) || !out || B < 1 || N < 0 || N2 < 0 || H < 1 || W < 1 ||
        elem_stride < 1)
        return tef::fail("tef_encode_events: bad arguments"), TEF_ERR_INVALID;
    int C;
    if (mode == TEF_ENCODE_IMAGE) C = 1;
    else if (mode == TEF_ENCODE_CHANNELS) C = 2;
    else if (mode == TEF_ENCODE_VOXEL) {
        C = channels;
        if ((N > 0 && !ts) || C < 1) return tef::fail("tef_encode_events: voxel needs ts and bins >= 1"), TEF_ERR_INVALID;
    } else return tef::fail("tef_encode_events: unknown mode"), TEF_ERR_INVALID;
    if ((size_t)W * sizeof(double) > kLdsBudget) return tef::fail("tef_encode_events: row too wide"), TEF_ERR_INVALID;
    static const hipError_t attr = hipFuncSetAttribute((const void *)encode_kernel,
                                                       hipFuncAttributeMaxDynamicSharedMemorySize, (int)kLdsBudget);
    if (attr != hipSuccess) return tef::fail_hip("hipFuncSetAttribute", attr), TEF_ERR_LAUNCH;
    int rows = (int)(kLdsBudget / ((size_t)W * sizeof(double)));
    if (rows > H) rows = H;
    int nbands = (H + rows - 1) / rows;
    size_t lds = (size_t)rows * W * sizeof(double);
    hipStream_t st = (hipStream_t)stream;
    {
        tef::ProfScope ps_(tef::PROF_ENCODE, st);
        hipLaunchKernelGGL(encode_kernel, dim3((unsigned)(B * C * nbands)), dim3(kThreads), lds, st, xs, ys, ts, ps,
                           batch_stride, elem_stride, N, list2, N2, mode, C, H, W, rows, nbands, out);
    }
    return tef::check_launch("encode_kernel");
}

extern "C" int tef_encode_events(const float *xs, const float *ys, const float *ts, const float *ps, int B,
                                 long batch_stride, int elem_stride, int N, int mode, int channels, int H, int W,
                                 float *out, void *stream)
{
    return encode(xs, ys, ts, ps, B, batch_stride, elem_stride, N, nullptr, 0, mode, channels, H, W, out, stream);
}

extern "C" int tef_encode_event_lists(const float *event_list, int N, const float *d_event_list, int Nd, int B, int mode,
                                      int channels, int H, int W, float *out, void *stream)
{
    const float *l = event_list;
    return encode(l ? l + 2 : nullptr, l ? l + 1 : nullptr, l, l ? l + 3 : nullptr, B, (long)N * 4, 4, N, d_event_list, Nd,
                  mode, channels, H, W, out, stream);
}
