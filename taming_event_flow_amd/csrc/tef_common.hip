// tef_common.hip — version + error reporting of the C ABI (include/tef.h).
#include <stdio.h>
#include <string.h>

#include "tef_common.h"

#include <algorithm>
#include <map>
#include <string>
#include <vector>

namespace {
thread_local char g_err[512] = "";

const char *kSlotNames[tef::PROF_NSLOTS] = {
    "pack_events", "warp", "iwe_splat", "mag_reduce", "loss_reduce", "chain_bwd", "dflow_splat",
    "smoothing_fwd", "smoothing_bwd", "encode", "conv_fwd_gemm", "conv_dgrad_gemm", "conv_wgrad_gemm", "image_count",
    "chain_bwd_rest",
};
struct Pending { int slot; hipEvent_t a, b; };
bool g_prof_on = false;
double g_ms[tef::PROF_NSLOTS];
long g_calls[tef::PROF_NSLOTS];
std::vector<Pending> g_pending;
std::vector<hipEvent_t> g_pool;
hipEvent_t g_open[tef::PROF_NSLOTS];
struct LayerPending { const char *label; hipEvent_t a, b; };
std::vector<LayerPending> g_layer_pending;
const char *g_layer_open = nullptr;
hipEvent_t g_layer_open_ev = nullptr;
std::map<std::string, std::pair<long, double>> g_layers;      // label -> (scopes, ms)

hipEvent_t get_event()
{
    if (!g_pool.empty()) {
        hipEvent_t e = g_pool.back();
        g_pool.pop_back();
        return e;
    }
    hipEvent_t e = nullptr;
    (void)hipEventCreate(&e);
    return e;
}
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

void prof_events(int slot, hipEvent_t *start, hipEvent_t *stop)
{
    *start = *stop = nullptr;
    if (!g_prof_on) return;
    *start = get_event();
    *stop = get_event();
    g_pending.push_back(Pending{slot, *start, *stop});
}

void prof_begin(int slot, hipStream_t st)
{
    if (!g_prof_on) return;
    hipEvent_t e = get_event();
    (void)hipEventRecord(e, st);
    g_open[slot] = e;
}

void prof_end(int slot, hipStream_t st)
{
    if (!g_prof_on) return;
    hipEvent_t e = get_event();
    (void)hipEventRecord(e, st);
    g_pending.push_back(Pending{slot, g_open[slot], e});
}

void layer_begin(const char *label, hipStream_t st)
{
    if (!g_prof_on) return;
    g_layer_open = label;
    g_layer_open_ev = get_event();
    (void)hipEventRecord(g_layer_open_ev, st);
}

void layer_end(hipStream_t st)
{
    if (!g_prof_on || !g_layer_open) return;
    hipEvent_t e = get_event();
    (void)hipEventRecord(e, st);
    g_layer_pending.push_back(LayerPending{g_layer_open, g_layer_open_ev, e});
    g_layer_open = nullptr;
}

}  // namespace tef

extern "C" {

int tef_profile_enable(int on)
{
    g_prof_on = on != 0;
    for (auto &p : g_layer_pending) { g_pool.push_back(p.a); g_pool.push_back(p.b); }
    g_layer_pending.clear();
    g_layers.clear();
    g_layer_open = nullptr;
    for (int i = 0; i < tef::PROF_NSLOTS; ++i) { g_ms[i] = 0.0; g_calls[i] = 0; }
    for (auto &p : g_pending) { g_pool.push_back(p.a); g_pool.push_back(p.b); }
    g_pending.clear();
    return 0;
}

int tef_profile_pause(int paused)
{
    g_prof_on = paused == 0;
    return 0;
}

int tef_profile_collect(void)
{
    for (auto &p : g_pending) {
        if (hipEventSynchronize(p.b) != hipSuccess) return tef::fail("tef_profile_collect: event sync failed"), TEF_ERR_LAUNCH;
        float ms = 0.0f;
        if (hipEventElapsedTime(&ms, p.a, p.b) == hipSuccess) { g_ms[p.slot] += ms; g_calls[p.slot] += 1; }
        g_pool.push_back(p.a);
        g_pool.push_back(p.b);
    }
    g_pending.clear();
    for (auto &p : g_layer_pending) {
        if (hipEventSynchronize(p.b) != hipSuccess) return tef::fail("tef_profile_collect: event sync failed"), TEF_ERR_LAUNCH;
        float ms = 0.0f;
        if (hipEventElapsedTime(&ms, p.a, p.b) == hipSuccess) {
            auto &e = g_layers[p.label];
            e.first += 1;
            e.second += ms;
        }
        g_pool.push_back(p.a);
        g_pool.push_back(p.b);
    }
    g_layer_pending.clear();
    return 0;
}

long tef_profile_layers(char *buf, size_t nbytes)
{
    std::string out;
    char line[160];
    for (auto &kv : g_layers) {
        snprintf(line, sizeof(line), "%s,%ld,%.6f\n", kv.first.c_str(), kv.second.first, kv.second.second);
        out += line;
    }
    if (buf && nbytes) {
        const size_t n = std::min(out.size(), nbytes - 1);
        memcpy(buf, out.data(), n);
        buf[n] = 0;
    }
    return (long)out.size();
}

int tef_profile_slots(void) { return tef::PROF_NSLOTS; }
const char *tef_profile_name(int slot) { return (slot >= 0 && slot < tef::PROF_NSLOTS) ? kSlotNames[slot] : ""; }
double tef_profile_ms(int slot) { return (slot >= 0 && slot < tef::PROF_NSLOTS) ? g_ms[slot] : 0.0; }
long tef_profile_calls(int slot) { return (slot >= 0 && slot < tef::PROF_NSLOTS) ? g_calls[slot] : 0; }

int tef_version(void) { return TEF_VERSION; }

const char *tef_last_error(void) { return g_err; }

}
