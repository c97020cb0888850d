// tef_common.h — error plumbing shared by the translation units of libtef_hip.so.
#ifndef TEF_COMMON_H
#define TEF_COMMON_H

#include <hip/hip_runtime.h>

#include "tef.h"

namespace tef {

// Sets the calling thread's last-error text; returns false so callers can `return fail(...)`.
bool fail(const char *msg);
bool fail_hip(const char *what, hipError_t e);
// hipGetLastError() after a launch: 0 or TEF_ERR_LAUNCH (message recorded).
int check_launch(const char *kernel);

}  // namespace tef

#endif
