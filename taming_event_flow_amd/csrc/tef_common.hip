// tef_common.hip — version + error reporting of the C ABI (include/tef.h).
#include <stdio.h>
#include <string.h>

#include "tef_common.h"

namespace {
thread_local char g_err[512] = "";
}

namespace tef {

bool fail(const char *msg)
{
    snprintf(g_err, sizeof(g_err), "%s", msg);
    return false;
}

bool fail_hip(const char *what, hipError_t e)
{
    snprintf(g_err, sizeof(g_err), "%s: %s", what, hipGetErrorString(e));
    return false;
}

int check_launch(const char *kernel)
{
    hipError_t e = hipGetLastError();
    if (e == hipSuccess) return 0;
    fail_hip(kernel, e);
    return TEF_ERR_LAUNCH;
}

}  // namespace tef

extern "C" {

int tef_version(void) { return TEF_VERSION; }

const char *tef_last_error(void) { return g_err; }

}
