// Micro-benchmark: what sets the ds_add_f64 / ds_add_f32 / ds_add_u64 rate on gfx950 — the LDS atomic unit or bank
// conflicts?  Each lane issues K atomics per iteration to addresses drawn from one of several patterns.
// Build: hipcc --offload-arch=gfx950 -O3 -munsafe-fp-atomics tools/lds_atomic_patterns.hip -o gpurun_out/lds_atomic_patterns
#include <hip/hip_runtime.h>
#include <stdio.h>
#include <stdlib.h>
#include <vector>

constexpr int PLANE = 16384;      // elements in the LDS plane (doubles: 128 KiB)

// TYPE 0 = f64, 1 = f32, 2 = u64, 3 = non-atomic f64 RMW
template <int TYPE>
__global__ __launch_bounds__(1024) void k(const int *__restrict__ idx, int n_per_block, int iters, double *out)
{
    extern __shared__ double img[];
    for (int p = threadIdx.x; p < PLANE; p += blockDim.x) img[p] = 0.0;
    __syncthreads();
    // 8 addresses per lane, loaded ONCE; every iteration re-uses them shifted by a wave-uniform amount (same conflict
    // structure, no global traffic in the loop)
    const int *pp = idx + ((size_t)blockIdx.x * 1024 + threadIdx.x) * 8;
    int a[8];
    for (int q = 0; q < 8; ++q) a[q] = pp[q];
    for (int it = 0; it < iters; ++it) {
        int sh = (it * 72) & (PLANE - 1);
#pragma unroll
        for (int q = 0; q < 8; ++q) {
            int ad = (a[q] + sh) & (PLANE - 1);
            if (TYPE == 0) atomicAdd(img + ad, 1.0);
            else if (TYPE == 1) atomicAdd((float *)img + ad, 1.0f);
            else if (TYPE == 2) atomicAdd((unsigned long long *)img + ad, 1ull);
            else img[ad] += 1.0;
        }
    }
    __syncthreads();
    if (out) for (int p = threadIdx.x; p < PLANE; p += blockDim.x) out[(size_t)blockIdx.x * PLANE + p] = img[p];
}

int main()
{
    const int nblocks = 1024, n_per_block = 8192, iters = 64;
    size_t n = (size_t)nblocks * n_per_block;
    std::vector<int> h(n);
    int *d;
    hipMalloc(&d, n * sizeof(int));
    const char *pat_names[] = {"random over plane", "lane-linear (conflict-free)", "same address per wave",
                               "random in 16x16 window (row stride 136)", "2x2 footprint of sorted 8x8 tile (stride 136)",
                               "lane-linear stride 2 (2-way)", "lane-linear stride 4 (4-way)", "random in 64 consecutive"};
    for (int pat = 0; pat < 8; ++pat) {
        for (size_t i = 0; i < n; ++i) {
            int q_ = (int)(i & 7); size_t th_ = i >> 3; int lane = (int)(th_ & 63), wave = (int)(th_ >> 6) * 8 + q_;
            int v;
            switch (pat) {
            case 0: v = rand() % PLANE; break;
            case 1: v = (wave * 64 + lane) % PLANE; break;
            case 2: v = (wave * 97) % PLANE; break;
            case 3: { int base = (wave * 131) % (PLANE - 16 * 136); v = base + (rand() % 16) * 136 + rand() % 16; } break;
            case 4: { int base = (wave * 131) % (PLANE - 12 * 136); v = base + (rand() % 10) * 136 + rand() % 10; } break;
            case 5: v = ((wave * 64 + lane) * 2) % PLANE; break;
            case 6: v = ((wave * 64 + lane) * 4) % PLANE; break;
            default: v = (wave * 64) % (PLANE - 64) + rand() % 64; break;
            }
            h[i] = v;
        }
        hipMemcpy(d, h.data(), n * sizeof(int), hipMemcpyHostToDevice);
        auto run = [&](auto kern, const char *name) {
            size_t lds = PLANE * sizeof(double);
            hipFuncSetAttribute((const void *)kern, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
            hipEvent_t a, b; hipEventCreate(&a); hipEventCreate(&b);
            hipLaunchKernelGGL(kern, dim3(nblocks), dim3(1024), lds, 0, d, n_per_block, iters, (double *)nullptr);
            hipEventRecord(a);
            for (int r = 0; r < 3; ++r) hipLaunchKernelGGL(kern, dim3(nblocks), dim3(1024), lds, 0, d, n_per_block, iters, (double *)nullptr);
            hipEventRecord(b); hipEventSynchronize(b);
            float ms; hipEventElapsedTime(&ms, a, b); ms /= 3;
            double rate = (double)n * iters / ms / 1e9;      // T atomics / s
            printf("  %-10s %8.3f ms  %6.3f T ops/s  (%5.1f cycles per wave-instruction per CU at 2.4 GHz)\n", name, ms, rate,
                   64.0 * 256 * 2.4e9 / (rate * 1e12));
        };
        printf("pattern %d: %s\n", pat, pat_names[pat]);
        run(k<0>, "ds_add_f64");
        run(k<1>, "ds_add_f32");
        run(k<2>, "ds_add_u64");
        run(k<3>, "rmw f64");
    }
    return 0;
}
