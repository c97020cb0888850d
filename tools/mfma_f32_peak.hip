// Calibration: sustained rate of v_mfma_f32_32x32x2_f32 on every SIMD of the chip, and the shader clock it runs at.
//   hipcc --offload-arch=gfx950 -O3 -o mfma_f32_peak tools/mfma_f32_peak.hip && ./mfma_f32_peak
#include <hip/hip_runtime.h>
#include <stdio.h>
#include <stdlib.h>

typedef float f32x16 __attribute__((ext_vector_type(16)));

template <int NACC>
__global__ __launch_bounds__(256) void mfma_loop(int iters, float *out, unsigned long long *cyc)
{
    f32x16 acc[NACC];
    for (int i = 0; i < NACC; ++i)
        for (int e = 0; e < 16; ++e) acc[i][e] = 0.0f;
    float a = (float)(threadIdx.x & 7) * 0.25f, b = (float)(threadIdx.x & 3) * 0.5f;
    unsigned long long t0 = __builtin_readcyclecounter();
    unsigned long long m0 = __builtin_amdgcn_s_memtime();
    for (int it = 0; it < iters; ++it) {
#pragma unroll
        for (int r = 0; r < 4; ++r)
#pragma unroll
            for (int i = 0; i < NACC; ++i) acc[i] = __builtin_amdgcn_mfma_f32_32x32x2f32(a, b, acc[i], 0, 0, 0);
    }
    unsigned long long m1 = __builtin_amdgcn_s_memtime();
    unsigned long long t1 = __builtin_readcyclecounter();
    float s = 0.0f;
    for (int i = 0; i < NACC; ++i)
        for (int e = 0; e < 16; ++e) s += acc[i][e];
    out[blockIdx.x * blockDim.x + threadIdx.x] = s;
    if (threadIdx.x == 0) { cyc[2 * blockIdx.x] = t1 - t0; cyc[2 * blockIdx.x + 1] = m1 - m0; }
}

template <int NACC>
void run(int wgs, int iters, const char *name)
{
    float *out;
    unsigned long long *cyc, *h = (unsigned long long *)malloc(sizeof(unsigned long long) * 2 * wgs);
    hipMalloc(&out, sizeof(float) * wgs * 256);
    hipMalloc(&cyc, sizeof(unsigned long long) * 2 * wgs);
    hipEvent_t e0, e1;
    hipEventCreate(&e0); hipEventCreate(&e1);
    mfma_loop<NACC><<<wgs, 256>>>(iters / 10, out, cyc);
    hipDeviceSynchronize();
    hipEventRecord(e0);
    mfma_loop<NACC><<<wgs, 256>>>(iters, out, cyc);
    hipEventRecord(e1);
    hipEventSynchronize(e1);
    float ms;
    hipEventElapsedTime(&ms, e0, e1);
    hipMemcpy(h, cyc, sizeof(unsigned long long) * 2 * wgs, hipMemcpyDeviceToHost);
    double flops = (double)wgs * 4 * iters * 4 * NACC * 4096.0;
    double c = 0, m = 0;
    for (int i = 0; i < wgs; ++i) { c += h[2 * i]; m += h[2 * i + 1]; }
    c /= wgs; m /= wgs;
    printf("%-28s wgs %5d  %8.3f ms  %7.1f TFLOP/s  readcyclecounter %.0f (%.0f MHz)  s_memtime %.0f (%.0f MHz)  cycles/MFMA/SIMD %.1f\n",
           name, wgs, ms, flops / ms / 1e9, c, c / ms / 1e3, m, m / ms / 1e3, m / ((double)iters * 4 * NACC) / ((wgs + 255) / 256));
    hipFree(out); hipFree(cyc); free(h);
}

int main()
{
    run<4>(256, 4000, "1 wave/SIMD, 4 acc");
    run<1>(256, 16000, "1 wave/SIMD, 1 acc");
    run<4>(512, 4000, "2 waves/SIMD, 4 acc");
    run<4>(1024, 2000, "4 waves/SIMD, 4 acc");
    run<4>(256, 40000, "1 wave/SIMD, 4 acc, 10x longer");
    return 0;
}
