// Micro-benchmark: LDS float-atomic throughput on gfx950 with random (bilinear-splat-like) addresses.
// Build: hipcc --offload-arch=gfx950 -O3 tools/lds_atomic_bench.hip -o /tmp/lds_atomic_bench
#include <hip/hip_runtime.h>
#include <stdio.h>
#include <stdlib.h>
#include <vector>

constexpr int W = 128, H = 128;

template <int MODE>
__global__ __launch_bounds__(1024) void k(const float2 *__restrict__ pos, int n_per_block, float2 *out)
{
    extern __shared__ float2 img[];
    for (int p = threadIdx.x; p < H * W; p += blockDim.x) img[p] = make_float2(0.f, 0.f);
    __syncthreads();
    const float2 *pp = pos + (size_t)blockIdx.x * n_per_block;
    for (int e = threadIdx.x; e < n_per_block; e += blockDim.x) {
        float2 p = pp[e];
        float fy = floorf(p.x), fx = floorf(p.y);
        int iy = (int)fy, ix = (int)fx;
        float wy1 = p.x - fy, wx1 = p.y - fx, wy0 = 1.f - wy1, wx0 = 1.f - wx1;
        float2 *b = img + iy * W + ix;
        if (MODE == 0) {          // 8 x ds_add_f32 (C and T channels)
            atomicAdd(&b[0].x, wy0 * wx0);     atomicAdd(&b[0].y, wy0 * wx0 * 0.5f);
            atomicAdd(&b[1].x, wy0 * wx1);     atomicAdd(&b[1].y, wy0 * wx1 * 0.5f);
            atomicAdd(&b[W].x, wy1 * wx0);     atomicAdd(&b[W].y, wy1 * wx0 * 0.5f);
            atomicAdd(&b[W + 1].x, wy1 * wx1); atomicAdd(&b[W + 1].y, wy1 * wx1 * 0.5f);
        } else if (MODE == 1) {   // 4 x ds_add_f32 (one channel)
            atomicAdd(&b[0].x, wy0 * wx0);
            atomicAdd(&b[1].x, wy0 * wx1);
            atomicAdd(&b[W].x, wy1 * wx0);
            atomicAdd(&b[W + 1].x, wy1 * wx1);
        } else if (MODE == 2) {   // non-atomic read-modify-write (racy; upper bound on LDS traffic)
            b[0].x += wy0 * wx0; b[0].y += wy0 * wx0 * .5f;
            b[1].x += wy0 * wx1; b[1].y += wy0 * wx1 * .5f;
            b[W].x += wy1 * wx0; b[W].y += wy1 * wx0 * .5f;
            b[W + 1].x += wy1 * wx1; b[W + 1].y += wy1 * wx1 * .5f;
        } else if (MODE == 3) {   // 4 x 64-bit integer atomic (fixed point pair)
            unsigned long long *q = (unsigned long long *)b;
            unsigned long long v = ((unsigned long long)(unsigned)(wy0 * wx0 * 4194304.f) << 32) | (unsigned)(wy0 * wx0 * 2097152.f);
            atomicAdd(&q[0], v); atomicAdd(&q[1], v); atomicAdd(&q[W], v); atomicAdd(&q[W + 1], v);
        } else if (MODE == 5) {   // 4 x ds_add_f64
            double *q = (double *)b;
            atomicAdd(&q[0], (double)(wy0 * wx0)); atomicAdd(&q[1], (double)(wy0 * wx1));
            atomicAdd(&q[W], (double)(wy1 * wx0)); atomicAdd(&q[W + 1], (double)(wy1 * wx1));
        } else if (MODE == 6) {   // 8 x ds_add_u32
            unsigned *q = (unsigned *)b;
            unsigned v = (unsigned)(wy0 * wx0 * 4194304.f);
            atomicAdd(&q[0], v); atomicAdd(&q[1], v); atomicAdd(&q[2], v); atomicAdd(&q[3], v);
            atomicAdd(&q[2 * W], v); atomicAdd(&q[2 * W + 1], v); atomicAdd(&q[2 * W + 2], v); atomicAdd(&q[2 * W + 3], v);
        } else if (MODE == 7) {   // 8 x ds_add_u64 (two 64-bit quantities per corner; image = 64 rows)
            unsigned long long *q = (unsigned long long *)img + (size_t)((iy & 63) * W + ix) * 2;
            unsigned long long v = (unsigned long long)(wy0 * wx0 * 1099511627776.f);
            atomicAdd(&q[0], v); atomicAdd(&q[1], v); atomicAdd(&q[2], v); atomicAdd(&q[3], v);
            atomicAdd(&q[2 * W], v); atomicAdd(&q[2 * W + 1], v); atomicAdd(&q[2 * W + 2], v); atomicAdd(&q[2 * W + 3], v);
        } else if (MODE == 4) {   // stream only (no LDS work)
            if (p.x < -1.f) b[0].x = 1.f;
        }
    }
    __syncthreads();
    if (out) for (int p = threadIdx.x; p < H * W; p += blockDim.x) out[(size_t)blockIdx.x * H * W + p] = img[p];
}

int main()
{
    const int nblocks = 1024, n_per_block = 65536;
    size_t n = (size_t)nblocks * n_per_block;
    std::vector<float2> h(n);
    for (size_t i = 0; i < n; ++i) h[i] = make_float2((H - 1.001f) * (rand() / (float)RAND_MAX), (W - 1.001f) * (rand() / (float)RAND_MAX));
    float2 *d, *o;
    hipMalloc(&d, n * sizeof(float2));
    hipMalloc(&o, (size_t)nblocks * H * W * sizeof(float2));
    hipMemcpy(d, h.data(), n * sizeof(float2), hipMemcpyHostToDevice);
    size_t lds = H * W * sizeof(float2);
    auto run = [&](auto kern, const char *name, int per_event) {
        hipFuncSetAttribute((const void *)kern, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
        hipEvent_t a, b; hipEventCreate(&a); hipEventCreate(&b);
        for (int it = 0; it < 2; ++it) hipLaunchKernelGGL(kern, dim3(nblocks), dim3(1024), lds, 0, d, n_per_block, (float2 *)nullptr);
        hipEventRecord(a);
        for (int it = 0; it < 5; ++it) hipLaunchKernelGGL(kern, dim3(nblocks), dim3(1024), lds, 0, d, n_per_block, (float2 *)nullptr);
        hipEventRecord(b); hipEventSynchronize(b);
        float ms; hipEventElapsedTime(&ms, a, b); ms /= 5;
        printf("%-28s %8.3f ms  %7.1f Gevents/s  %7.1f G LDS-ops/s  stream %6.1f GB/s\n", name, ms, n / ms / 1e6,
               (double)n * per_event / ms / 1e6, n * 8.0 / ms / 1e6);
    };
    run(k<0>, "8x ds_add_f32 (C,T)", 8);
    run(k<1>, "4x ds_add_f32", 4);
    run(k<2>, "8x non-atomic RMW", 8);
    run(k<3>, "4x ds_add_u64", 4);
    run(k<5>, "4x ds_add_f64", 4);
    run(k<6>, "8x ds_add_u32", 8);
    run(k<7>, "8x ds_add_u64", 8);
    run(k<4>, "stream only", 0);
    return 0;
}
