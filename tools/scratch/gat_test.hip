#include <hip/hip_runtime.h>
#include <stdio.h>
__global__ void k(int *q, int *o, int xcd)
{
    int v = 0;
    if (threadIdx.x == 0) asm volatile("global_atomic_add %0, %1, %2, %3 sc0" : "=v"(v) : "v"(0), "v"(1), "s"(q + xcd) : "memory");
    __syncthreads();
    if (threadIdx.x == 0) {
        asm volatile("s_waitcnt vmcnt(0)" : "+v"(v) : : "memory");
        o[blockIdx.x] = v;
    }
}
int main()
{
    int *q, *o, h[64], hq[8];
    (void)hipMalloc(&q, 32); (void)hipMalloc(&o, 64 * 4);
    (void)hipMemset(q, 0, 32); (void)hipMemset(o, 0xff, 64 * 4);
    hipLaunchKernelGGL(k, dim3(64), dim3(256), 0, 0, q, o, 3);
    hipError_t e = hipDeviceSynchronize();
    (void)hipMemcpy(h, o, 64 * 4, hipMemcpyDeviceToHost); (void)hipMemcpy(hq, q, 32, hipMemcpyDeviceToHost);
    printf("err=%d counter[3]=%d returned:", (int)e, hq[3]);
    long s = 0; for (int i = 0; i < 64; ++i) { s += h[i]; if (i < 8) printf(" %d", h[i]); }
    printf(" sum=%ld (expect 64, 2016)\n", s);
    return 0;
}
