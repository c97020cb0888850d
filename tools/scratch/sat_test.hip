#include <hip/hip_runtime.h>
#include <stdio.h>
__global__ void k(int *q, int *o)
{
    int v = 1;
    asm volatile("s_atomic_add %0, %1, 0x0 glc\n\ts_waitcnt lgkmcnt(0)" : "+s"(v) : "s"(q) : "memory");
    if ((threadIdx.x & 63) == 0) o[blockIdx.x * 4 + (threadIdx.x >> 6)] = v;
}
int main()
{
    int *q, *o, h[64], hq;
    hipMalloc(&q, 4); hipMalloc(&o, 64 * 4);
    hipMemset(q, 0, 4); hipMemset(o, 0xff, 64 * 4);
    hipLaunchKernelGGL(k, dim3(16), dim3(256), 0, 0, q, o);
    hipError_t e = hipDeviceSynchronize();
    hipMemcpy(h, o, 64 * 4, hipMemcpyDeviceToHost); hipMemcpy(&hq, q, 4, hipMemcpyDeviceToHost);
    printf("err=%d counter=%d returned:", (int)e, hq);
    long s = 0; for (int i = 0; i < 64; ++i) { s += h[i]; if (i < 8) printf(" %d", h[i]); }
    printf(" sum=%ld (expect counter=64, sum=2016)\n", s);
    return 0;
}
