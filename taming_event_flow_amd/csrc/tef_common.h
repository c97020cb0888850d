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

// Opt-in per-kernel timing (tef_profile_* in include/tef.h): HIP events recorded on the launch stream
// around each kernel while enabled; zero cost when disabled.
enum ProfSlot {
    PROF_PACK = 0, PROF_WARP, PROF_SPLAT, PROF_STATS, PROF_REDUCE, PROF_CHAIN_BWD, PROF_DFLOW,
    PROF_SMOOTH_FWD, PROF_SMOOTH_BWD, PROF_ENCODE, PROF_CONV_FWD, PROF_CONV_DGRAD, PROF_CONV_WGRAD, PROF_COUNT, PROF_CHAIN_BWD_REST,
    PROF_NSLOTS
};
// A (start, stop) event pair to hand to hipExtLaunchKernelGGL: the timestamps then come from the kernel's own dispatch
// signal and no marker packets enter the stream.  Both are null while profiling is off (= a plain launch).
void prof_events(int slot, hipEvent_t *start, hipEvent_t *stop);
void prof_begin(int slot, hipStream_t st);
void prof_end(int slot, hipStream_t st);
struct ProfScope {
    int slot;
    hipStream_t st;
    ProfScope(int s, hipStream_t stream) : slot(s), st(stream) { prof_begin(slot, st); }
    ~ProfScope() { prof_end(slot, st); }
};

// Per-LAYER attribution (round 6): a named scope around everything a network layer enqueues in one direction (its
// convolution launches AND the reduce / activation launches that belong to it); collected with the slots above and read
// back as text through tef_profile_layers.  Only meaningful on ONE stream (an event pair measures what the stream did in
// between).  `label` must outlive the collection (static storage).
void layer_begin(const char *label, hipStream_t st);
void layer_end(hipStream_t st);
struct LayerScope {
    hipStream_t st;
    LayerScope(const char *label, hipStream_t stream) : st(stream) { layer_begin(label, st); }
    ~LayerScope() { layer_end(st); }
};

}  // namespace tef

#endif
