// tef_net.hip — RecEVFlowNet: one recurrent pass, forward or backward, as ONE call of the C ABI (include/tef.h
// tef_net_*), and the window's deferred weight gradients as one more.
//
// The reference runs a pass as ~60 autograd nodes (models/arch.py:217-242, models/model.py:65-85).  Round 2 made it one
// autograd node whose Python body issued ~100 ctypes calls and ~60 tensor allocations per pass: correct, but an eager
// caller — what a drop-in train_flow.py loop is — spent 49 ms of host time on a window the GPU finishes in 38.  Here the
// layer table is walked in C: the caller hands over ONE activation arena per pass (the "tape": everything the backward
// reads, plus the pass's outputs) and ONE gradient arena per backward call, both laid out by this file, and every launch
// of the pass is enqueued from a single call.  Host code only: the kernels are the ones of tef_conv.hip / tef_cell.hip /
// tef_resize.hip, reached through their own C entry points.
//
//   forward   [4 x (strided head conv, fused ConvGRU cell)] -> [nres residual blocks] -> [4 x (bilinear x2 of
//             (features + encoder skip) [+ bilinear x2 of the previous prediction], conv over the two sources, 1x1 head,
//             bilinear to the input size x 2^level x flow_scale, cropped)]
//   backward  the same table in reverse; a gradient with several producers is summed where it is consumed
//             (tef_grad_act / tef_convgru_cell_bwd take up to four addends); weight gradients go into the accumulators of
//             the plan or, for layers marked `defer`, stay as pre-activation gradients in the gradient arena until
//             tef_net_window_wgrads reduces them over all passes of the window.
#include <stdint.h>
#include <string.h>

#include <stdio.h>

#include <algorithm>
#include <map>
#include <string>

#include "tef.h"
#include "tef_common.h"

namespace {

constexpr int L = TEF_NET_MAX_LEVELS, R = TEF_NET_MAX_RES;

struct Geo {
    int lv, nres, B, top;
    int C[L], h[L], w[L];
    size_t n[L];                    // elements of a level's state
    int src[L], cin0[L], cin1[L], out[L], hs[L], ws[L], lvl[L], s[L];      // decoder k
    size_t fl;                      // elements of one full-resolution flow
};

bool make_geo(const tef_net_plan *p, Geo *g)
{
    if (!p) return tef::fail("tef_net: null plan");
    if (p->levels < 1 || p->levels > L || p->nres < 0 || p->nres > R) return tef::fail("tef_net: levels / residual blocks out of range");
    if (p->B < 1 || p->bins < 1 || p->nout < 1) return tef::fail("tef_net: bad sizes");
    if (p->H % (1 << p->levels) || p->W % (1 << p->levels) || p->H < (1 << p->levels) || p->W < (1 << p->levels))
        return tef::fail("tef_net: the (padded) input sides must be multiples of 2^levels");
    if (p->crop_top < 0 || p->crop_left < 0 || p->crop_top >= p->H || p->crop_left >= p->W) return tef::fail("tef_net: bad crop");
    g->lv = p->levels; g->nres = p->nres; g->B = p->B;
    for (int i = 0; i < g->lv; ++i) {
        g->C[i] = p->width[i];
        g->h[i] = p->H >> (i + 1);
        g->w[i] = p->W >> (i + 1);
        if (g->C[i] < 1) return tef::fail("tef_net: bad level width");
        g->n[i] = (size_t)p->B * g->C[i] * g->h[i] * g->w[i];
    }
    g->top = g->lv - 1;
    for (int k = 0; k < g->lv; ++k) {
        const int lvl = g->lv - 1 - k;
        g->lvl[k] = lvl;
        g->src[k] = k == 0 ? g->C[g->top] : p->dec_out[k - 1];
        g->out[k] = p->dec_out[k];
        if (g->out[k] < 1) return tef::fail("tef_net: bad decoder width");
        g->cin0[k] = k == 0 ? g->src[k] : p->nout;       // conv sources: (up-sampled prediction, up-sampled features) or features alone
        g->cin1[k] = k == 0 ? 0 : g->src[k];
        g->hs[k] = g->h[lvl] * 2;
        g->ws[k] = g->w[lvl] * 2;
        g->s[k] = 1 << lvl;
    }
    g->fl = (size_t)p->B * p->nout * (p->H - p->crop_top) * (p->W - p->crop_left);
    return true;
}

// activation arena of one pass (floats)
struct Tape {
    size_t e[L], u[L], r[L], o[L], hn[L];
    size_t mid[R], y[R], lin;
    size_t upx[L], upp[L], d[L], p[L], flow[L];
    size_t total;
};
// gradient arena of one backward call (floats)
struct GTape {
    size_t g_e[L], g_ur[L], g_o[L], de[L], dh[L], dxin[L];
    size_t gy[R], gmid[R], dmid[R], dres[R];
    size_t gp[L], gd[L], fup[L], dd[L], dx0[L], dx1[L], skip[L], dprev[L];
    size_t dtop;        // decoder half run alone: the deepest state's gradient addends summed (residual chain + skip)
    size_t total;
};

inline size_t take(size_t &o, size_t n)
{
    const size_t at = o;
    o += (n + 63) & ~(size_t)63;         // 256-byte aligned pieces
    return at;
}

Tape make_tape(const tef_net_plan *p, const Geo &g)
{
    Tape t{};
    size_t o = 0;
    for (int i = 0; i < g.lv && !p->dec_only; ++i) {      // (a decoder-only arena reads the states through plan.hn_ext)
        t.e[i] = take(o, g.n[i]); t.u[i] = take(o, g.n[i]); t.r[i] = take(o, g.n[i]); t.o[i] = take(o, g.n[i]);
        t.hn[i] = take(o, g.n[i]);
    }
    for (int j = 0; j < g.nres; ++j) { t.mid[j] = take(o, g.n[g.top]); t.y[j] = take(o, g.n[g.top]); }
    t.lin = take(o, g.n[g.top]);
    for (int k = 0; k < g.lv; ++k) {
        const size_t px = (size_t)g.B * g.hs[k] * g.ws[k];
        t.upx[k] = take(o, px * g.src[k]);
        t.upp[k] = take(o, k ? px * p->nout : 0);
        t.d[k] = take(o, px * g.out[k]);
        t.p[k] = take(o, px * p->nout);
        t.flow[k] = take(o, g.fl);
    }
    t.total = o;
    return t;
}

GTape make_gtape(const tef_net_plan *p, const Geo &g)
{
    GTape t{};
    size_t o = 0;
    for (int i = 0; i < g.lv && !p->dec_only; ++i) {
        t.g_e[i] = take(o, g.n[i]); t.g_ur[i] = take(o, 2 * g.n[i]); t.g_o[i] = take(o, g.n[i]);
        t.de[i] = take(o, g.n[i]); t.dh[i] = take(o, g.n[i]);
        const size_t nin = i ? g.n[i - 1] : (size_t)p->B * p->bins * p->H * p->W;
        t.dxin[i] = take(o, nin);
    }
    for (int j = 0; j < g.nres; ++j) {
        t.gy[j] = take(o, g.n[g.top]); t.gmid[j] = take(o, g.n[g.top]); t.dmid[j] = take(o, g.n[g.top]);
        t.dres[j] = take(o, g.n[g.top]);
    }
    for (int k = 0; k < g.lv; ++k) {
        const size_t px = (size_t)g.B * g.hs[k] * g.ws[k], pl = (size_t)g.B * g.h[g.lvl[k]] * g.w[g.lvl[k]];
        t.gp[k] = take(o, px * p->nout);
        t.gd[k] = take(o, px * g.out[k]);
        t.fup[k] = take(o, px * p->nout);
        t.dd[k] = take(o, px * g.out[k]);
        t.dx0[k] = take(o, px * g.cin0[k]);
        t.dx1[k] = take(o, px * g.cin1[k]);
        t.skip[k] = take(o, pl * g.src[k]);
        t.dprev[k] = take(o, k ? pl * p->nout : 0);
    }
    t.dtop = take(o, g.n[g.top]);
    t.total = o;
    return t;
}

inline tef_conv_desc cdesc(int B, int C0, int C1, int H, int W, int N, int k, int stride, int act)
{
    tef_conv_desc d;
    d.B = B; d.C0 = C0; d.C1 = C1; d.H = H; d.W = W; d.N = N; d.ksize = k; d.stride = stride; d.act = act;
    return d;
}

struct Descs {
    tef_conv_desc head[L], ur[L], og[L], res, dec[L], pred[L];
    tef_gru_desc gru[L];
};

Descs make_descs(const tef_net_plan *p, const Geo &g, int act_mode /* 1: forward activations, 0: TEF_ACT_NONE */)
{
    Descs D;
    for (int i = 0; i < g.lv; ++i) {
        const int cin = i ? g.C[i - 1] : p->bins, hin = i ? g.h[i - 1] : p->H, win = i ? g.w[i - 1] : p->W;
        D.head[i] = cdesc(g.B, cin, 0, hin, win, g.C[i], 3, 2, act_mode ? TEF_ACT_RELU : TEF_ACT_NONE);
        D.ur[i] = cdesc(g.B, g.C[i], g.C[i], g.h[i], g.w[i], 2 * g.C[i], 3, 1, act_mode ? TEF_ACT_SIGMOID : TEF_ACT_NONE);
        D.og[i] = cdesc(g.B, g.C[i], g.C[i], g.h[i], g.w[i], g.C[i], 3, 1, act_mode ? TEF_ACT_TANH : TEF_ACT_NONE);
        D.gru[i].B = g.B; D.gru[i].C = g.C[i]; D.gru[i].H = g.h[i]; D.gru[i].W = g.w[i];
    }
    D.res = cdesc(g.B, g.C[g.top], 0, g.h[g.top], g.w[g.top], g.C[g.top], 3, 1, TEF_ACT_NONE);
    for (int k = 0; k < g.lv; ++k) {
        D.dec[k] = cdesc(g.B, g.cin0[k], g.cin1[k], g.hs[k], g.ws[k], g.out[k], 3, 1, act_mode ? TEF_ACT_RELU : TEF_ACT_NONE);
        D.pred[k] = cdesc(g.B, g.out[k], 0, g.hs[k], g.ws[k], p->nout, 1, 1, act_mode ? p->final_act : TEF_ACT_NONE);
    }
    return D;
}

size_t workspace_need(const tef_net_plan *p, const Geo &g)
{
    const Descs D = make_descs(p, g, 1);
    size_t m = tef_conv_workspace_bytes(&D.res);
    for (int i = 0; i < g.lv; ++i) {
        if (!p->dec_only) {
            m = std::max(m, tef_conv_workspace_bytes(&D.head[i]));
            m = std::max(m, tef_conv_workspace_bytes(&D.ur[i]));
            m = std::max(m, tef_conv_workspace_bytes(&D.og[i]));
            m = std::max(m, tef_convgru_workspace_bytes(&D.gru[i]));
        }
        m = std::max(m, tef_conv_workspace_bytes(&D.dec[i]));
        m = std::max(m, tef_conv_workspace_bytes(&D.pred[i]));
    }
    return m;
}

// the new state of level i of the pass(es) a plan describes: in the pass's own arena, or (decoder half of a whole window as
// one batch) the caller's stack of the passes' states
inline const float *state_of(const tef_net_plan *p, const Tape &t, const float *tape, int i)
{
    return p->hn_ext[i] ? p->hn_ext[i] : tape + t.hn[i];
}

// layer bits of the "ran" mask a backward call reports (which pre-activation gradients of the arena are valid)
inline uint64_t bit_head(int i) { return 1ull << i; }
inline uint64_t bit_ur(int i) { return 1ull << (8 + i); }
inline uint64_t bit_og(int i) { return 1ull << (16 + i); }
inline uint64_t bit_res1(int j) { return 1ull << (24 + j); }
inline uint64_t bit_res2(int j) { return 1ull << (28 + j); }
inline uint64_t bit_dec(int k) { return 1ull << (32 + k); }
inline uint64_t bit_pred(int k) { return 1ull << (40 + k); }

// static label strings of the per-layer profiling scopes (tef::LayerScope): "enc1.head fwd", "dec2 dgrad", ...
const char *lbl(const char *fmt, int idx)
{
    static std::map<std::string, std::string> table;      // (node addresses are stable)
    char buf[64];
    snprintf(buf, sizeof(buf), fmt, idx);
    return table.emplace(buf, buf).first->second.c_str();
}
#define TEF_LAYER(fmt, idx) tef::LayerScope layer_scope_(lbl(fmt, idx), (hipStream_t)stream)

#define TEF_TRY(call)                 \
    do {                              \
        const int rc_ = (call);       \
        if (rc_ != 0) return rc_;     \
    } while (0)

// input gradients of a convolution whose pre-activation gradient g is formed; the weight gradient is accumulated now
// unless the layer defers it (then g stays in the arena for tef_net_window_wgrads)
// 3x3 single-source layers outside the multi-part weight-gradient kernels: deferred all the same when the caller has promised
// a workspace for tef_net_window_wgrads (plan.copy_batch): their passes are copied side by side and reduced as one batch
inline bool copy_batch_ok(const tef_net_plan *p, const tef_conv_desc &d)
{
    return p->copy_batch && d.ksize == 3 && d.C1 == 0 && !tef_conv_wgrad_parts_supported(&d);
}

int conv_bwd(const tef_net_plan *p, const tef_conv_desc &d, const tef_net_conv &c, const float *g, const float *x0, const float *x1, float *dx0,
             float *dx1, void *ws, size_t ws_bytes, void *stream)
{
    const bool defer = c.defer && (tef_conv_wgrad_parts_supported(&d) || copy_batch_ok(p, d));
    return tef_conv_backward_keep(&d, x0, x1, nullptr, c.w2, nullptr, nullptr, g, nullptr, d.N, dx0, dx1,
                                  defer ? nullptr : c.dw, nullptr, nullptr, nullptr, d.N, nullptr, ws, ws_bytes, stream);
}

// the same, delivering the pre-activation gradient of the layer that produced the input (tef_conv_backward_post)
int conv_bwd_post(const tef_net_plan *p, const tef_conv_desc &d, const tef_net_conv &c, const float *g, const float *x0, const tef_conv_post &post, void *ws,
                  size_t ws_bytes, void *stream)
{
    const bool defer = c.defer && (tef_conv_wgrad_parts_supported(&d) || copy_batch_ok(p, d));
    return tef_conv_backward_post(&d, x0, c.w2, g, defer ? nullptr : c.dw, &post, ws, ws_bytes, stream);
}

}  // namespace

extern "C" {

size_t tef_net_tape_floats(const tef_net_plan *p)
{
    Geo g;
    if (!make_geo(p, &g)) return 0;
    return make_tape(p, g).total;
}

size_t tef_net_gtape_floats(const tef_net_plan *p)
{
    Geo g;
    if (!make_geo(p, &g)) return 0;
    return make_gtape(p, g).total;
}

size_t tef_net_workspace_bytes(const tef_net_plan *p)
{
    Geo g;
    if (!make_geo(p, &g)) return 0;
    return workspace_need(p, g);
}

int tef_net_layout(const tef_net_plan *p, size_t *flow_off, size_t *state_off, size_t *dstate_off, size_t *dx_off)
{
    Geo g;
    if (!make_geo(p, &g)) return TEF_ERR_INVALID;
    const Tape t = make_tape(p, g);
    const GTape gt = make_gtape(p, g);
    for (int k = 0; k < g.lv; ++k) {
        if (flow_off) flow_off[k] = t.flow[k];
        if (state_off) state_off[k] = t.hn[k];
        if (dstate_off) dstate_off[k] = gt.dh[k];
    }
    if (dx_off) *dx_off = gt.dxin[0];
    return 0;
}

int tef_net_pass_forward(const tef_net_plan *p, const float *x, const float *const *states_in, float *tape, void *ws,
                         size_t ws_bytes, void *stream)
{
    return tef_net_pass_forward_part(p, TEF_NET_ENCODERS | TEF_NET_DECODERS, x, states_in, tape, ws, ws_bytes, stream);
}

namespace {
int pass_forward_impl(const tef_net_plan *p, int part, int lo, int hi, const float *x, const float *const *states_in, float *tape,
                      void *ws, size_t ws_bytes, void *stream);
int pass_backward_impl(const tef_net_plan *p, int part, int lo, int hi, int above_valid, const float *x, const float *const *states_in,
                       const float *tape, const float *const *dflows, const float *const *dstates, const float *const *dstates2,
                       int want_dx, float *gtape, unsigned long long *ran_out, long long *dstate_off, int *dx_valid, void *ws,
                       size_t ws_bytes, void *stream);
}  // namespace

int tef_net_pass_forward_part(const tef_net_plan *p, int part, const float *x, const float *const *states_in, float *tape,
                              void *ws, size_t ws_bytes, void *stream)
{
    return pass_forward_impl(p, part, 0, TEF_NET_MAX_LEVELS, x, states_in, tape, ws, ws_bytes, stream);
}

int tef_net_pass_forward_levels(const tef_net_plan *p, int lo, int hi, const float *x, const float *const *states_in, float *tape,
                                void *ws, size_t ws_bytes, void *stream)
{
    if (lo < 0 || hi <= lo) return tef::fail("tef_net_pass_forward_levels: empty level range"), TEF_ERR_INVALID;
    return pass_forward_impl(p, TEF_NET_ENCODERS, lo, hi, x, states_in, tape, ws, ws_bytes, stream);
}

namespace {
int pass_forward_impl(const tef_net_plan *p, int part, int lo, int hi, const float *x, const float *const *states_in, float *tape,
                      void *ws, size_t ws_bytes, void *stream)
{
    Geo g;
    if (!make_geo(p, &g)) return TEF_ERR_INVALID;
    hi = std::min(hi, g.lv);
    if (lo >= hi && (part & TEF_NET_ENCODERS)) return tef::fail("tef_net_pass_forward: level range outside the network"), TEF_ERR_INVALID;
    if (!(part & (TEF_NET_ENCODERS | TEF_NET_DECODERS))) return tef::fail("tef_net_pass_forward: no part selected"), TEF_ERR_INVALID;
    if (!tape || !ws || ((part & TEF_NET_ENCODERS) && ((lo == 0 && !x) || !states_in)))
        return tef::fail("tef_net_pass_forward: null pointer"), TEF_ERR_INVALID;
    if (ws_bytes < workspace_need(p, g)) return tef::fail("tef_net_pass_forward: workspace too small"), TEF_ERR_WORKSPACE;
    const Tape t = make_tape(p, g);
    const Descs D = make_descs(p, g, 1);
    if (part & TEF_NET_ENCODERS) {
        const float *cur = lo ? tape + t.hn[lo - 1] : x;      // (levels from lo on: the state below was written by an earlier call)
        for (int i = lo; i < hi; ++i) {
            if (!states_in[i]) return tef::fail("tef_net_pass_forward: null state (pass zeros for a fresh sequence)"), TEF_ERR_INVALID;
            {
                TEF_LAYER("enc%d.head fwd", i);
                TEF_TRY(tef_conv_forward(&D.head[i], cur, nullptr, nullptr, p->head[i].wp, p->head[i].bias, tape + t.e[i], ws, ws_bytes, stream));
            }
            {
                TEF_LAYER("enc%d.gru fwd", i);
                TEF_TRY(tef_convgru_cell_fwd(&D.gru[i], tape + t.e[i], states_in[i], p->gate_ur[i].wp, p->gate_o[i].wp, p->gate_ur[i].bias,
                                             p->gate_o[i].bias, tape + t.u[i], tape + t.r[i], tape + t.o[i], tape + t.hn[i], ws, ws_bytes, stream));
            }
            cur = tape + t.hn[i];
        }
    }
    if (!(part & TEF_NET_DECODERS)) return 0;
    const float *cur = state_of(p, t, tape, g.top);  // (the new states of this pass: written by the encoder half)
    tef_conv_desc dres = D.res;
    for (int j = 0; j < g.nres; ++j) {
        dres.act = TEF_ACT_RELU;
        {
            TEF_LAYER("res%d.conv1 fwd", j);
            TEF_TRY(tef_conv_forward(&dres, cur, nullptr, nullptr, p->res1[j].wp, p->res1[j].bias, tape + t.mid[j], ws, ws_bytes, stream));
        }
        dres.act = TEF_ACT_NONE;
        TEF_LAYER("res%d.conv2 fwd", j);
        TEF_TRY(tef_conv_forward(&dres, tape + t.mid[j], nullptr, nullptr, p->res2[j].wp, p->res2[j].bias, tape + t.lin, ws, ws_bytes, stream));
        TEF_TRY(tef_add_act(tape + t.lin, cur, TEF_ACT_RELU, g.n[g.top], tape + t.y[j], stream));      // submodules.py:219-226
        cur = tape + t.y[j];
    }
    const float *pred = nullptr;
    for (int k = 0; k < g.lv; ++k) {
        const int lvl = g.lvl[k];
        // features + encoder skip, x2 (arch.py:236 "sum" skip + UpsampleConvLayer's interpolate), previous prediction x2
        {
            TEF_LAYER("dec%d.up fwd", k);
            if (pred && !(g.w[lvl] & 1))      // both in one launch
                TEF_TRY(tef_upsample2x_pair(cur, state_of(p, t, tape, lvl), g.B * g.src[k], tape + t.upx[k], pred, g.B * p->nout, tape + t.upp[k],
                                            g.h[lvl], g.w[lvl], stream));
            else {
                TEF_TRY(tef_upsample_bilinear_crop(cur, state_of(p, t, tape, lvl), g.B * g.src[k], g.h[lvl], g.w[lvl], 2, 2, 1.0f, 0, 0, tape + t.upx[k], stream));
                if (pred) TEF_TRY(tef_upsample_bilinear_crop(pred, nullptr, g.B * p->nout, g.h[lvl], g.w[lvl], 2, 2, 1.0f, 0, 0, tape + t.upp[k], stream));
            }
        }
        const float *x0 = pred ? tape + t.upp[k] : tape + t.upx[k], *x1 = pred ? tape + t.upx[k] : nullptr;
        {
            TEF_LAYER("dec%d fwd", k);
            TEF_TRY(tef_conv_forward(&D.dec[k], x0, x1, nullptr, p->dec[k].wp, p->dec[k].bias, tape + t.d[k], ws, ws_bytes, stream));
        }
        TEF_LAYER("pred%d fwd", k);
        TEF_TRY(tef_conv_forward(&D.pred[k], tape + t.d[k], nullptr, nullptr, p->pred[k].wp, p->pred[k].bias, tape + t.p[k], ws, ws_bytes, stream));
        // to the input size, x 2^level (model.py:76-81) x the caller's flow scaling, top / left padding cropped (:83)
        TEF_TRY(tef_upsample_bilinear_crop(tape + t.p[k], nullptr, g.B * p->nout, g.hs[k], g.ws[k], g.s[k], g.s[k],
                                           (float)g.s[k] * p->flow_scale, p->crop_top, p->crop_left, tape + t.flow[k], stream));
        cur = tape + t.d[k];
        pred = tape + t.p[k];
    }
    return 0;
}
}  // namespace

int tef_net_pass_backward(const tef_net_plan *p, const float *x, const float *const *states_in, const float *tape,
                          const float *const *dflows, const float *const *dstates, int want_dx, float *gtape,
                          unsigned long long *ran_out, int *dstate_valid, int *dx_valid, void *ws, size_t ws_bytes,
                          void *stream)
{
    long long off[L];
    if (!dstate_valid) return tef::fail("tef_net_pass_backward: null pointer"), TEF_ERR_INVALID;
    const int rc = tef_net_pass_backward_part(p, TEF_NET_ENCODERS | TEF_NET_DECODERS, x, states_in, tape, dflows, dstates, want_dx,
                                              gtape, ran_out, off, dx_valid, ws, ws_bytes, stream);
    if (rc == 0)
        for (int i = 0; i < p->levels; ++i) dstate_valid[i] = off[i] >= 0;
    return rc;
}

int tef_net_pass_backward_part(const tef_net_plan *p, int part, const float *x, const float *const *states_in, const float *tape,
                               const float *const *dflows, const float *const *dstates, int want_dx, float *gtape,
                               unsigned long long *ran_out, long long *dstate_off, int *dx_valid, void *ws, size_t ws_bytes,
                               void *stream)
{
    return tef_net_pass_backward_part2(p, part, x, states_in, tape, dflows, dstates, nullptr, want_dx, gtape, ran_out, dstate_off,
                                       dx_valid, ws, ws_bytes, stream);
}

int tef_net_pass_backward_part2(const tef_net_plan *p, int part, const float *x, const float *const *states_in, const float *tape,
                                const float *const *dflows, const float *const *dstates, const float *const *dstates2, int want_dx,
                                float *gtape, unsigned long long *ran_out, long long *dstate_off, int *dx_valid, void *ws,
                                size_t ws_bytes, void *stream)
{
    return pass_backward_impl(p, part, 0, TEF_NET_MAX_LEVELS, 0, x, states_in, tape, dflows, dstates, dstates2, want_dx, gtape, ran_out,
                              dstate_off, dx_valid, ws, ws_bytes, stream);
}

int tef_net_pass_backward_levels(const tef_net_plan *p, int lo, int hi, int above_valid, const float *x, const float *const *states_in,
                                 const float *tape, const float *const *dstates, const float *const *dstates2, int want_dx,
                                 float *gtape, unsigned long long *ran_out, long long *dstate_off, int *dx_valid, void *ws,
                                 size_t ws_bytes, void *stream)
{
    if (lo < 0 || hi <= lo) return tef::fail("tef_net_pass_backward_levels: empty level range"), TEF_ERR_INVALID;
    return pass_backward_impl(p, TEF_NET_ENCODERS, lo, hi, above_valid, x, states_in, tape, nullptr, dstates, dstates2, want_dx, gtape,
                              ran_out, dstate_off, dx_valid, ws, ws_bytes, stream);
}

namespace {
int pass_backward_impl(const tef_net_plan *p, int part, int lo, int hi, int above_valid, const float *x, const float *const *states_in,
                       const float *tape, const float *const *dflows, const float *const *dstates, const float *const *dstates2,
                       int want_dx, float *gtape, unsigned long long *ran_out, long long *dstate_off, int *dx_valid, void *ws,
                       size_t ws_bytes, void *stream)
{
    Geo g;
    if (!make_geo(p, &g)) return TEF_ERR_INVALID;
    const bool enc = part & TEF_NET_ENCODERS, dec = part & TEF_NET_DECODERS;
    if (!enc && !dec) return tef::fail("tef_net_pass_backward: no part selected"), TEF_ERR_INVALID;
    hi = std::min(hi, g.lv);
    if (enc && lo >= hi) return tef::fail("tef_net_pass_backward: level range outside the network"), TEF_ERR_INVALID;
    if (!tape || !gtape || !ws || !dstate_off || (dec && !dflows) || (enc && ((lo == 0 && !x) || !states_in || !dstates || !dx_valid)))
        return tef::fail("tef_net_pass_backward: null pointer"), TEF_ERR_INVALID;
    if (ws_bytes < workspace_need(p, g)) return tef::fail("tef_net_pass_backward: workspace too small"), TEF_ERR_WORKSPACE;
    const Tape t = make_tape(p, g);
    const GTape q = make_gtape(p, g);
    const Descs D = make_descs(p, g, 0);
    uint64_t ran = 0;
    const float *skip_grads[L] = {nullptr};   // d loss / d (features + encoder skip) of decoder k, shared by both addends
    const float *d_prev_pred = nullptr;       // gradient arriving at prediction k from decoder k + 1
    const float *d_feat = nullptr;            // gradient arriving at decoder k's output from decoder k + 1
    for (int k = g.lv - 1; k >= 0 && dec; --k) {
        const int lvl = g.lvl[k], hw = g.hs[k] * g.ws[k];
        const float *srcs[4];
        int ns = 0;
        if (dflows[k]) {
            TEF_LAYER("pred%d.flow bwd", k);
            TEF_TRY(tef_upsample_bilinear_crop_backward(dflows[k], g.B * p->nout, g.hs[k], g.ws[k], g.s[k], g.s[k],
                                                        (float)g.s[k] * p->flow_scale, p->crop_top, p->crop_left, gtape + q.fup[k], stream));
            srcs[ns++] = gtape + q.fup[k];
        }
        if (d_prev_pred) srcs[ns++] = d_prev_pred;
        const float *feat[4];
        int nf = 0;
        // the 1x1 head (onto <= 4 channels) and the decoder's own activation gradient in ONE launch (round 6: four before —
        // the head's pre-activation gradient, its input and weight gradients, the decoder's pre-activation gradient)
        const bool fused = ns && D.pred[k].ksize == 1 && D.pred[k].stride == 1 && D.pred[k].C1 == 0 && D.pred[k].N <= 4;
        if (fused) {
            TEF_LAYER("dec%d.tail bwd", k);
            tef_conv_desc hd = D.pred[k];
            hd.act = p->final_act;
            TEF_TRY(tef_dec_head_backward(&hd, srcs, ns, tape + t.p[k], p->pred[k].w2, tape + t.d[k], TEF_ACT_RELU, d_feat, gtape + q.gp[k],
                                          gtape + q.gd[k], p->pred[k].db, p->pred[k].dw, p->dec[k].db, stream));
            ran |= bit_pred(k);
        } else {
            if (ns) {
                TEF_TRY(tef_grad_act(srcs, ns, tape + t.p[k], p->final_act, g.B, p->nout, hw, gtape + q.gp[k], p->pred[k].db, stream));
                TEF_TRY(conv_bwd(p, D.pred[k], p->pred[k], gtape + q.gp[k], tape + t.d[k], nullptr, gtape + q.dd[k], nullptr, ws, ws_bytes, stream));
                ran |= bit_pred(k);
                feat[nf++] = gtape + q.dd[k];
            }
            if (d_feat) feat[nf++] = d_feat;
            if (!nf) {          // nothing reaches this level (all its flow gradients absent): the chain is dead here
                skip_grads[k] = nullptr;
                d_prev_pred = d_feat = nullptr;
                continue;
            }
            TEF_TRY(tef_grad_act(feat, nf, tape + t.d[k], TEF_ACT_RELU, g.B, g.out[k], hw, gtape + q.gd[k], p->dec[k].db, stream));
        }
        const float *x0 = k ? tape + t.upp[k] : tape + t.upx[k], *x1 = k ? tape + t.upx[k] : nullptr;
        {
            TEF_LAYER("dec%d dgrad", k);
            TEF_TRY(conv_bwd(p, D.dec[k], p->dec[k], gtape + q.gd[k], x0, x1, gtape + q.dx0[k], k ? gtape + q.dx1[k] : nullptr, ws, ws_bytes, stream));
        }
        ran |= bit_dec(k);
        TEF_LAYER("dec%d.up bwd", k);
        const float *dupx = k ? gtape + q.dx1[k] : gtape + q.dx0[k], *dupp = k ? gtape + q.dx0[k] : nullptr;
        skip_grads[k] = d_feat = gtape + q.skip[k];
        d_prev_pred = nullptr;
        if (dupp) {       // both adjoints in one launch
            TEF_TRY(tef_upsample2x_pair_backward(dupx, g.B * g.src[k], gtape + q.skip[k], dupp, g.B * p->nout, gtape + q.dprev[k], g.h[lvl],
                                                 g.w[lvl], stream));
            d_prev_pred = gtape + q.dprev[k];
        } else {
            TEF_TRY(tef_upsample_bilinear_crop_backward(dupx, g.B * g.src[k], g.h[lvl], g.w[lvl], 2, 2, 1.0f, 0, 0, gtape + q.skip[k], stream));
        }
    }
    // residual blocks, last first
    const float *srcs[4];
    int ns = 0;
    if (skip_grads[0]) srcs[ns++] = skip_grads[0];
    const int hwt = g.h[g.top] * g.w[g.top], Ct = g.C[g.top];
    // (round 6) conv -> relu -> conv chains: the input gradient of a convolution is turned into the pre-activation gradient of
    // the layer before it by the convolution's own reduction launch (tef_conv_backward_post) — relu'(mid) * dmid inside a
    // block, relu'(y[j-1]) * (dres[j] + gy[j]) between blocks — instead of a tef_grad_act launch each
    bool gy_formed = false;               // gy[j] was formed by block j + 1's last input gradient
    for (int j = g.nres - 1; j >= 0 && ns && dec; --j) {
        const float *xin = j ? tape + t.y[j - 1] : state_of(p, t, tape, g.top);
        if (!gy_formed)
            TEF_TRY(tef_grad_act(srcs, ns, tape + t.y[j], TEF_ACT_RELU, g.B, Ct, hwt, gtape + q.gy[j], p->res2[j].db, stream));
        {
            TEF_LAYER("res%d.conv2 dgrad", j);
            tef_conv_post post{tape + t.mid[j], TEF_ACT_RELU, nullptr, gtape + q.gmid[j], p->res1[j].db};
            TEF_TRY(conv_bwd_post(p, D.res, p->res2[j], gtape + q.gy[j], tape + t.mid[j], post, ws, ws_bytes, stream));
        }
        TEF_LAYER("res%d.conv1 dgrad", j);
        if (j > 0) {
            tef_conv_post post{tape + t.y[j - 1], TEF_ACT_RELU, gtape + q.gy[j], gtape + q.gy[j - 1], p->res2[j - 1].db};
            TEF_TRY(conv_bwd_post(p, D.res, p->res1[j], gtape + q.gmid[j], xin, post, ws, ws_bytes, stream));
            gy_formed = true;
        } else {
            TEF_TRY(conv_bwd(p, D.res, p->res1[j], gtape + q.gmid[j], xin, nullptr, gtape + q.dres[j], nullptr, ws, ws_bytes, stream));
        }
        ran |= bit_res1(j) | bit_res2(j);
        srcs[0] = gtape + q.dres[j];      // through the two convolutions + the residual connection itself
        srcs[1] = gtape + q.gy[j];
        ns = 2;
    }
    // encoders, deepest first.  (The deepest state is decoder 0's skip addend AND the input of the residual blocks; without
    // residual blocks `srcs` is that same gradient once more — features + skip = 2 x the state — and it is added twice.)
    for (int i = 0; i < g.lv; ++i) dstate_off[i] = -1;
    if (!enc) {
        // Decoder half alone: what it sends to the new state of each level, as ONE tensor per level for the caller's
        // autograd (the encoder half, run later with these added to the next pass's gradients, sums nothing itself).
        // Levels below the top receive their decoder's skip gradient; the deepest state also feeds the residual chain.
        for (int i = 0; i < g.top; ++i)
            if (skip_grads[g.lv - 1 - i]) dstate_off[i] = (long long)q.skip[g.lv - 1 - i];
        if (skip_grads[0]) srcs[ns++] = skip_grads[0];
        if (ns == 1) {
            dstate_off[g.top] = (long long)(srcs[0] - gtape);
        } else if (ns > 1) {
            TEF_TRY(tef_grad_act(srcs, ns, nullptr, TEF_ACT_NONE, g.B, Ct, hwt, gtape + q.dtop, nullptr, stream));
            dstate_off[g.top] = (long long)q.dtop;
        }
        if (ran_out) *ran_out = ran;
        return 0;
    }
    *dx_valid = 0;
    if (!dec && hi < g.lv) {      // a level range: what the call for the levels above left for the state below them
        ns = 0;
        if (above_valid) { srcs[0] = gtape + q.dxin[hi]; ns = 1; }
    }
    for (int i = hi - 1; i >= lo; --i) {
        const float *sources[4];
        int n = 0;
        for (int a = 0; a < ns; ++a) sources[n++] = srcs[a];
        const float *sg = skip_grads[g.lv - 1 - i];
        if (sg) sources[n++] = sg;
        if (dstates[i]) sources[n++] = dstates[i];
        if (dstates2 && dstates2[i]) sources[n++] = dstates2[i];
        ns = 0;
        if (!n) continue;
        if (n > 4) return tef::fail("tef_net_pass_backward: more than four gradient addends at a state"), TEF_ERR_INVALID;
        const bool dur = p->gate_ur[i].defer && tef_conv_wgrad_parts_supported(&D.ur[i]);
        const bool dog = p->gate_o[i].defer && tef_conv_wgrad_parts_supported(&D.og[i]);
        // (the cell's last sweep also forms the head convolution's pre-activation gradient: relu'(e) * d loss / d e + its bias sums)
        {
        TEF_LAYER("enc%d.gru bwd", i);
        TEF_TRY(tef_convgru_cell_bwd_head(&D.gru[i], tape + t.e[i], states_in[i], tape + t.u[i], tape + t.r[i], tape + t.o[i], sources, n,
                                          p->gate_ur[i].w2, p->gate_o[i].w2, gtape + q.g_ur[i], gtape + q.g_o[i], nullptr, gtape + q.dh[i],
                                          dur ? nullptr : p->gate_ur[i].dw, dur ? nullptr : p->gate_ur[i].dw2, dog ? nullptr : p->gate_o[i].dw,
                                          p->gate_ur[i].db, p->gate_ur[i].db2, p->gate_o[i].db, TEF_ACT_RELU, gtape + q.g_e[i], p->head[i].db,
                                          ws, ws_bytes, stream));
        }
        dstate_off[i] = (long long)q.dh[i];
        ran |= bit_ur(i) | bit_og(i);
        TEF_LAYER("enc%d.head dgrad", i);
        const bool want = i > 0 || want_dx;
        const float *xin = i ? tape + t.hn[i - 1] : x;
        TEF_TRY(conv_bwd(p, D.head[i], p->head[i], gtape + q.g_e[i], xin, nullptr, want ? gtape + q.dxin[i] : nullptr, nullptr, ws, ws_bytes, stream));
        ran |= bit_head(i);
        if (want) {
            if (i > 0) { srcs[0] = gtape + q.dxin[i]; ns = 1; }
            else *dx_valid = 1;
        }
    }
    if (lo > 0) *dx_valid = ns == 1;      // (the gradient w.r.t. state lo - 1 waits at gtape + dxin[lo] for the levels below)
    if (ran_out) *ran_out = ran;
    return 0;
}
}  // namespace

// The window's deferred weight gradients: per layer ONE reduction over the pixels of all passes (tef_conv_wgrad_parts, up to
// TEF_CONV_MAX_PARTS passes per launch) instead of one short, atomics-heavy reduction per pass.  x / states_in / tape /
// gtape / ran: the arguments and results of the npass backward calls since the last flush (any order).
size_t tef_net_window_wgrads_workspace(const tef_net_plan *p, int npass)
{
    Geo g;
    if (!make_geo(p, &g) || !p->copy_batch || npass < 1) return 0;
    const Descs D = make_descs(p, g, 0);
    size_t need = 0;
    auto one = [&](const tef_conv_desc &d) {
        if (!copy_batch_ok(p, d)) return;
        const int pad = d.ksize / 2, Ho = (d.H + 2 * pad - d.ksize) / d.stride + 1, Wo = (d.W + 2 * pad - d.ksize) / d.stride + 1;
        const size_t gsz = (size_t)d.B * d.N * Ho * Wo, xsz = (size_t)d.B * d.C0 * d.H * d.W;
        tef_conv_desc db = d;
        db.B = d.B * npass;
        need = std::max(need, ((((size_t)npass * (gsz + xsz) * sizeof(float)) + 255) & ~(size_t)255) + tef_conv_workspace_bytes(&db));
    };
    for (int i = 0; i < g.lv && !p->dec_only; ++i) one(D.head[i]);
    one(D.res);
    for (int k = 0; k < g.lv; ++k) one(D.dec[k]);
    return need;
}

int tef_net_window_wgrads(const tef_net_plan *p, int npass, const float *const *x, const float *const *const *states_in,
                          const float *const *tape, const float *const *gtape, const unsigned long long *ran, void *stream)
{
    return tef_net_window_wgrads_part(p, TEF_NET_ENCODERS | TEF_NET_DECODERS, npass, x, states_in, tape, gtape, ran, stream);
}

int tef_net_window_wgrads_part(const tef_net_plan *p, int part, int npass, const float *const *x,
                               const float *const *const *states_in, const float *const *tape, const float *const *gtape,
                               const unsigned long long *ran, void *stream)
{
    if (part < 1 || part > 3) return tef::fail("tef_net_window_wgrads_part: part must be 1, 2 or 3"), TEF_ERR_INVALID;
    Geo g;
    if (!make_geo(p, &g)) return TEF_ERR_INVALID;
    if (npass < 0 || (npass && (!x || !states_in || !tape || !gtape || !ran))) return tef::fail("tef_net_window_wgrads: null pointer"), TEF_ERR_INVALID;
    const Tape t = make_tape(p, g);
    const GTape q = make_gtape(p, g);
    const Descs D = make_descs(p, g, 0);
    const float *gs[TEF_CONV_MAX_PARTS], *x0s[TEF_CONV_MAX_PARTS], *x1s[TEF_CONV_MAX_PARTS], *gts[TEF_CONV_MAX_PARTS];
    // one layer: collect the passes in which it ran, reduce them in groups
    auto layer = [&](const char *label, const tef_conv_desc &d, const tef_net_conv &c, uint64_t bit, auto part) -> int {
        if (!c.defer) return 0;
        if (!tef_conv_wgrad_parts_supported(&d)) {
            if (!copy_batch_ok(p, d)) return 0;
            // the passes in which the layer ran, side by side in the workspace: one reduction over a batch of n x B samples
            tef::LayerScope layer_scope_(label, (hipStream_t)stream);
            int n = 0;
            for (int s = 0; s < npass; ++s) n += (ran[s] & bit) ? 1 : 0;
            if (!n) return 0;
            const int pad = d.ksize / 2, Ho = (d.H + 2 * pad - d.ksize) / d.stride + 1, Wo = (d.W + 2 * pad - d.ksize) / d.stride + 1;
            const size_t gsz = (size_t)d.B * d.N * Ho * Wo, xsz = (size_t)d.B * d.C0 * d.H * d.W;
            tef_conv_desc db = d;
            db.B = d.B * n;
            const size_t head = (((size_t)n * (gsz + xsz) * sizeof(float)) + 255) & ~(size_t)255, cws = tef_conv_workspace_bytes(&db);
            if (!p->wgrad_ws || p->wgrad_ws_bytes < head + cws)
                return tef::fail("tef_net_window_wgrads: plan.copy_batch needs plan.wgrad_ws (tef_net_window_wgrads_workspace)"), TEF_ERR_WORKSPACE;
            float *G = (float *)p->wgrad_ws, *X = G + (size_t)n * gsz;
            int k = 0;
            for (int s = 0; s < npass; ++s) {
                if (!(ran[s] & bit)) continue;
                const float *gg, *a, *b, *c_;
                part(s, gg, a, b, c_);
                if (hipMemcpyAsync(G + (size_t)k * gsz, gg, gsz * sizeof(float), hipMemcpyDeviceToDevice, (hipStream_t)stream) != hipSuccess ||
                    hipMemcpyAsync(X + (size_t)k * xsz, a, xsz * sizeof(float), hipMemcpyDeviceToDevice, (hipStream_t)stream) != hipSuccess)
                    return tef::fail("tef_net_window_wgrads: hipMemcpyAsync"), TEF_ERR_LAUNCH;
                ++k;
            }
            return tef_conv_backward_keep(&db, X, nullptr, nullptr, nullptr, nullptr, nullptr, G, nullptr, db.N, nullptr, nullptr, c.dw, c.dw2,
                                          nullptr, nullptr, c.dw2 ? db.N / 2 : db.N, nullptr, (char *)p->wgrad_ws + head, p->wgrad_ws_bytes - head, stream);
        }
        tef::LayerScope layer_scope_(label, (hipStream_t)stream);
        int n = 0;
        for (int s = 0; s <= npass; ++s) {
            if (s < npass && (ran[s] & bit)) {
                part(s, gs[n], x0s[n], x1s[n], gts[n]);
                ++n;
            }
            if (n == TEF_CONV_MAX_PARTS || (s == npass && n)) {
                TEF_TRY(tef_conv_wgrad_parts(&d, n, gs, x0s, x1s, gts, c.dw, c.dw2, c.dw2 ? d.N / 2 : d.N, stream));
                n = 0;
            }
        }
        return 0;
    };
    for (int i = 0; i < g.lv && (part & TEF_NET_ENCODERS); ++i) {
        TEF_TRY(layer(lbl("enc%d.head wgrad", i), D.head[i], p->head[i], bit_head(i), [&](int s, const float *&gg, const float *&a, const float *&b, const float *&c_) {
            gg = gtape[s] + q.g_e[i]; a = i ? tape[s] + t.hn[i - 1] : x[s]; b = nullptr; c_ = nullptr; }));
        TEF_TRY(layer(lbl("enc%d.gru.ur wgrad", i), D.ur[i], p->gate_ur[i], bit_ur(i), [&](int s, const float *&gg, const float *&a, const float *&b, const float *&c_) {
            gg = gtape[s] + q.g_ur[i]; a = tape[s] + t.e[i]; b = states_in[s][i]; c_ = nullptr; }));
        TEF_TRY(layer(lbl("enc%d.gru.og wgrad", i), D.og[i], p->gate_o[i], bit_og(i), [&](int s, const float *&gg, const float *&a, const float *&b, const float *&c_) {
            gg = gtape[s] + q.g_o[i]; a = tape[s] + t.e[i]; b = states_in[s][i]; c_ = tape[s] + t.r[i]; }));
    }
    if (!(part & TEF_NET_DECODERS)) return 0;
    for (int j = 0; j < g.nres; ++j) {
        TEF_TRY(layer(lbl("res%d.conv1 wgrad", j), D.res, p->res1[j], bit_res1(j), [&](int s, const float *&gg, const float *&a, const float *&b, const float *&c_) {
            gg = gtape[s] + q.gmid[j]; a = j ? tape[s] + t.y[j - 1] : state_of(p, t, tape[s], g.top); b = nullptr; c_ = nullptr; }));
        TEF_TRY(layer(lbl("res%d.conv2 wgrad", j), D.res, p->res2[j], bit_res2(j), [&](int s, const float *&gg, const float *&a, const float *&b, const float *&c_) {
            gg = gtape[s] + q.gy[j]; a = tape[s] + t.mid[j]; b = nullptr; c_ = nullptr; }));
    }
    for (int k = 0; k < g.lv; ++k) {
        TEF_TRY(layer(lbl("dec%d wgrad", k), D.dec[k], p->dec[k], bit_dec(k), [&](int s, const float *&gg, const float *&a, const float *&b, const float *&c_) {
            gg = gtape[s] + q.gd[k]; a = k ? tape[s] + t.upp[k] : tape[s] + t.upx[k]; b = k ? tape[s] + t.upx[k] : nullptr; c_ = nullptr; }));
        TEF_TRY(layer(lbl("pred%d wgrad", k), D.pred[k], p->pred[k], bit_pred(k), [&](int s, const float *&gg, const float *&a, const float *&b, const float *&c_) {
            gg = gtape[s] + q.gp[k]; a = tape[s] + t.d[k]; b = nullptr; c_ = nullptr; }));
    }
    return 0;
}

}  // extern "C"
