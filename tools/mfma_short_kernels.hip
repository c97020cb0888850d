// Calibration: what a SHORT MFMA-only kernel reaches (the convolution launches of one pass are 40-100 us each).
// 256 workgroups x 512 threads (two waves per SIMD), n MFMAs (v_mfma_f32_32x32x2_f32) per wave, 20 launches back to back.
//   hipcc --offload-arch=gfx950 -O3 -o /tmp/mfma_short tools/mfma_short_kernels.hip && /tmp/mfma_short
#include <hip/hip_runtime.h>
#include <stdio.h>

typedef float f32x16 __attribute__((ext_vector_type(16)));

__global__ __launch_bounds__(512) void mfma_only(int iters, float *out)
{
    f32x16 acc[2];
    for (int i = 0; i < 2; ++i)
        for (int e = 0; e < 16; ++e) acc[i][e] = 0.0f;
    float a = (float)(threadIdx.x & 7) * 0.25f, b = (float)(threadIdx.x & 3) * 0.5f;
    for (int it = 0; it < iters; ++it) {
#pragma unroll
        for (int r = 0; r < 4; ++r)
#pragma unroll
            for (int i = 0; i < 2; ++i) acc[i] = __builtin_amdgcn_mfma_f32_32x32x2f32(a, b, acc[i], 0, 0, 0);
    }
    float s = 0.0f;
    for (int i = 0; i < 2; ++i)
        for (int e = 0; e < 16; ++e) s += acc[i][e];
    if (s == 12345.678f) out[blockIdx.x * blockDim.x + threadIdx.x] = s;
}

int main()
{
    float *out;
    hipMalloc(&out, sizeof(float) * 1024 * 512);
    hipEvent_t e0, e1;
    hipEventCreate(&e0); hipEventCreate(&e1);
    const int reps = 20;
    for (int wgs : {256, 512}) {
        for (int mfmas : {144, 288, 576, 1152, 2304, 9216, 36864}) {
            int iters = mfmas / 8;
            for (int i = 0; i < 3; ++i) hipLaunchKernelGGL(mfma_only, dim3(wgs), dim3(512), 0, 0, iters, out);
            hipDeviceSynchronize();
            hipEventRecord(e0);
            for (int i = 0; i < reps; ++i) hipLaunchKernelGGL(mfma_only, dim3(wgs), dim3(512), 0, 0, iters, out);
            hipEventRecord(e1);
            hipEventSynchronize(e1);
            float ms;
            hipEventElapsedTime(&ms, e0, e1);
            double us = ms * 1e3 / reps, flops = (double)wgs * 8 * mfmas * 4096.0;
            double ideal = (double)(wgs / 256) * 2 * mfmas * 64 / 2400.0;       // us at 2.4 GHz, 2 waves per SIMD per 256 wgs
            printf("wgs %4d  %6d MFMA/wave  %8.1f us/launch  ideal %8.1f us  %6.1f TFLOP/s  overhead %6.1f us\n", wgs, mfmas, us,
                   ideal, flops / us / 1e6, us - ideal);
        }
    }
    return 0;
}
