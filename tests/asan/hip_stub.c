/* Dry-run HIP runtime for the host-side sanitizer run of libtef_hip.so (tests/asan/run_host_asan.sh; test build only, never
 * part of the product): every entry point the library imports, as a no-op that reports success.  Kernel launches do nothing;
 * hipMemsetAsync and hipMemcpyAsync really write — the "device" buffers of the dry run are host tensors, so a memset past the end of one is an
 * AddressSanitizer report.  Preloaded in front of libamdhip64.so, whose versioned symbols these unversioned ones satisfy. */
#include <stddef.h>
#include <string.h>

typedef struct { unsigned x, y, z; } dim3_;
static int g_calls;

void **__hipRegisterFatBinary(const void *data) { static void *h; (void)data; return &h; }
void __hipRegisterFunction(void **m, const void *host, char *dev, const char *name, unsigned tl, void *tid, void *bid,
                           void *bd, void *gd, int *ws) { (void)m; (void)host; (void)dev; (void)name; (void)tl; (void)tid; (void)bid; (void)bd; (void)gd; (void)ws; }
void __hipUnregisterFatBinary(void **m) { (void)m; }
int __hipPushCallConfiguration(dim3_ g, dim3_ b, size_t shmem, void *stream) { (void)g; (void)b; (void)shmem; (void)stream; return 0; }
int __hipPopCallConfiguration(dim3_ *g, dim3_ *b, size_t *shmem, void **stream)
{
    if (g) { g->x = g->y = g->z = 1; }
    if (b) { b->x = b->y = b->z = 1; }
    if (shmem) *shmem = 0;
    if (stream) *stream = 0;
    return 0;
}
int hipLaunchKernel(const void *f, dim3_ g, dim3_ b, void **args, size_t shmem, void *stream)
{ (void)f; (void)g; (void)b; (void)args; (void)shmem; (void)stream; ++g_calls; return 0; }
int hipExtLaunchKernel(const void *f, dim3_ g, dim3_ b, void **args, size_t shmem, void *stream, void *e0, void *e1, int flags)
{ (void)f; (void)g; (void)b; (void)args; (void)shmem; (void)stream; (void)e0; (void)e1; (void)flags; ++g_calls; return 0; }
int hipGetLastError(void) { return 0; }
const char *hipGetErrorString(int e) { (void)e; return "dry run"; }
int hipFuncSetAttribute(const void *f, int attr, int v) { (void)f; (void)attr; (void)v; return 0; }
int hipGetDevice(int *d) { if (d) *d = 0; return 0; }
int hipDeviceGetAttribute(int *v, int attr, int dev) { (void)attr; (void)dev; if (v) *v = 256; return 0; }
int hipMemsetAsync(void *p, int value, size_t n, void *stream) { (void)stream; memset(p, value, n); return 0; }
/* (a real copy too: the window's copied weight-gradient batches, tef_net_window_wgrads with plan.copy_batch) */
int hipMemcpyAsync(void *dst, const void *src, size_t n, int kind, void *stream) { (void)kind; (void)stream; memcpy(dst, src, n); return 0; }
int hipEventCreate(void **e) { static int dummy; if (e) *e = &dummy; return 0; }
int hipEventRecord(void *e, void *s) { (void)e; (void)s; return 0; }
int hipEventSynchronize(void *e) { (void)e; return 0; }
int hipEventElapsedTime(float *ms, void *a, void *b) { (void)a; (void)b; if (ms) *ms = 0.0f; return 0; }
int tef_dry_run_launches(void) { return g_calls; }
