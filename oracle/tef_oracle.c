/*
 * tef_oracle.c — CPU restatement (plain C, fp32) of the reference's event-warping +
 * contrast-maximisation loss path.   *** TEST INFRASTRUCTURE ONLY ***
 *
 * Only tests/, __graft_entry__.smoke() and bench.py's cpu_baseline leg may load this
 * library; the product path (taming_event_flow_amd/) never does and fails loudly when
 * the HIP library is missing.
 *
 * Parity is PINNED: every function here is checked against golden vectors produced by
 * running the reference itself (tests/golden/make_golden.py imports /root/reference and
 * records inputs + outputs; tests/test_oracle_golden.py replays them).
 *
 * Formulation: the reference materialises tensors of temporaries per warping step
 * (loss/flow.py:521-586, 588-746); this restatement follows each event along its own
 * trajectory instead (same arithmetic per event, same summation order for the IWEs).
 *
 * Reference citations are path:line under /root/reference.
 */
#include <math.h>
#include <stdlib.h>
#include <string.h>
#ifdef _OPENMP
#include <omp.h>
#endif

#define TEF_EPS 1e-9f

/* ------------------------------------------------------------------------------------------
 * Bilinear flow lookup = utils/iwe.py:17-40 get_event_flow, i.e. the coordinate normalisation
 * at :30-31 followed by ATen's CPU grid_sampler_2d (bilinear, align_corners=True, zero padding;
 * third-party arithmetic: torch 2.10 aten/src/ATen/native/cpu/GridSamplerKernel.cpp, pinned by
 * tests/golden/primitives.npz).  Returns the 4 taps so that callers can reuse them for the
 * backward pass.
 * ---------------------------------------------------------------------------------------- */
typedef struct {
    int i00, i01, i10, i11;     /* linear indices of nw, ne, sw, se (or -1 when outside) */
    float w00, w01, w10, w11;   /* bilinear weights */
    float wy0, wy1, wx0, wx1;   /* (1-n), n, (1-w), w : 1-D factors for the position Jacobian */
} tef_taps;

static inline float tef_unnormalize(float v, int size)
{
    /* utils/iwe.py:30-31:  n = 2 * v / (size - 1) - 1 ; ATen: (n + 1) * ((size - 1) / 2) */
    float n = (2.0f * v) / (float)(size - 1) - 1.0f;
    return (n + 1.0f) * ((float)(size - 1) / 2.0f);
}

static inline void tef_make_taps(float y, float x, int H, int W, tef_taps *t)
{
    float iy = tef_unnormalize(y, H), ix = tef_unnormalize(x, W);
    float fy = floorf(iy), fx = floorf(ix);
    float n = iy - fy, w = ix - fx;
    float s = 1.0f - n, e = 1.0f - w;
    int y0 = (int)fy, x0 = (int)fx, y1 = y0 + 1, x1 = x0 + 1;
    int vy0 = (y0 >= 0 && y0 < H), vy1 = (y1 >= 0 && y1 < H);
    int vx0 = (x0 >= 0 && x0 < W), vx1 = (x1 >= 0 && x1 < W);
    t->i00 = (vy0 && vx0) ? y0 * W + x0 : -1;
    t->i01 = (vy0 && vx1) ? y0 * W + x1 : -1;
    t->i10 = (vy1 && vx0) ? y1 * W + x0 : -1;
    t->i11 = (vy1 && vx1) ? y1 * W + x1 : -1;
    t->w00 = s * e; t->w01 = s * w; t->w10 = n * e; t->w11 = n * w;
    t->wy0 = s; t->wy1 = n; t->wx0 = e; t->wx1 = w;
}

static inline float tef_tap_value(const float *map, const tef_taps *t)
{
    float v00 = t->i00 >= 0 ? map[t->i00] : 0.0f, v01 = t->i01 >= 0 ? map[t->i01] : 0.0f;
    float v10 = t->i10 >= 0 ? map[t->i10] : 0.0f, v11 = t->i11 >= 0 ? map[t->i11] : 0.0f;
    /* ATen's vectorised CPU kernel is compiled with floating-point contraction: its
     * `nw_val * nw + ne_val * ne + sw_val * sw + se_val * se` is ONE product followed by three fused multiply-adds
     * (established bit for bit against torch 2.10 here: 0 mismatches in 200 000 random lookups, where the unfused sum
     * differs in 35 % of them by an ulp — and an ulp of position is 1 % of a 1e-5 hat weight: round 4) */
    return __builtin_fmaf(v11, t->w11, __builtin_fmaf(v10, t->w10, __builtin_fmaf(v01, t->w01, v00 * t->w00)));
}

/* d value / d(y, x) of the lookup (grid gradient of grid_sampler_2d backward) */
static inline void tef_tap_jacobian(const float *map, const tef_taps *t, float *dy, float *dx)
{
    float v00 = t->i00 >= 0 ? map[t->i00] : 0.0f, v01 = t->i01 >= 0 ? map[t->i01] : 0.0f;
    float v10 = t->i10 >= 0 ? map[t->i10] : 0.0f, v11 = t->i11 >= 0 ? map[t->i11] : 0.0f;
    *dx = (v01 - v00) * t->wy0 + (v11 - v10) * t->wy1;
    *dy = (v10 - v00) * t->wx0 + (v11 - v01) * t->wx1;
}

static inline void tef_tap_scatter(float *dmap, const tef_taps *t, float g)
{
    if (t->i00 >= 0) dmap[t->i00] += g * t->w00;
    if (t->i01 >= 0) dmap[t->i01] += g * t->w01;
    if (t->i10 >= 0) dmap[t->i10] += g * t->w10;
    if (t->i11 >= 0) dmap[t->i11] += g * t->w11;
}

/* utils/iwe.py:43-60 purge_unfeasible: closed interval test */
static inline int tef_inbounds(float y, float x, int H, int W)
{
    return (y >= 0.0f) && (y <= (float)H - 1.0f) && (x >= 0.0f) && (x <= (float)W - 1.0f);
}

/* exported primitive: get_event_flow forward (+ optional backward for upstream gradient gout[B,N,2]=(y,x)) */
void tef_oracle_get_event_flow(const float *fx, const float *fy, int B, int H, int W, const float *loc, int N,
                               float *out, const float *gout, float *dfx, float *dfy, float *dloc)
{
    for (int b = 0; b < B; ++b) {
        const float *mx = fx + (size_t)b * H * W, *my = fy + (size_t)b * H * W;
        for (int e = 0; e < N; ++e) {
            size_t o = ((size_t)b * N + e) * 2;
            tef_taps t;
            tef_make_taps(loc[o], loc[o + 1], H, W, &t);
            out[o] = tef_tap_value(my, &t);
            out[o + 1] = tef_tap_value(mx, &t);
            if (gout) {
                float gy = gout[o], gx = gout[o + 1], jyy, jyx, jxy, jxx;
                tef_tap_scatter(dfy + (size_t)b * H * W, &t, gy);
                tef_tap_scatter(dfx + (size_t)b * H * W, &t, gx);
                tef_tap_jacobian(my, &t, &jyy, &jyx);
                tef_tap_jacobian(mx, &t, &jxy, &jxx);
                dloc[o] = gy * jyy + gx * jxy;
                dloc[o + 1] = gy * jyx + gx * jxx;
            }
        }
    }
}

/* ------------------------------------------------------------------------------------------
 * Splat geometry = utils/iwe.py:63-113 get_interpolation (bilinear branch).
 * Corner order TL, TR, BL, BR (:90-94).  d weight / d pos follows autograd through
 * torch.max(zeros, 1 - |d|) (ties split 0.5, abs'(0) = 0) and torch.prod.
 * ---------------------------------------------------------------------------------------- */
typedef struct {
    int idx[4];      /* linear pixel index, -1 when the corner is outside (:103-104) */
    float w[4];      /* bilinear weight (0 when outside) */
    float dwy[4];    /* d w / d y */
    float dwx[4];    /* d w / d x */
} tef_splat;

static inline float tef_hat(float d, float *slope)
{
    float v = 1.0f - fabsf(d);
    float sg = (d > 0.0f) ? 1.0f : ((d < 0.0f) ? -1.0f : 0.0f);
    if (v > 0.0f) { *slope = -sg; return v; }
    if (v == 0.0f) { *slope = -0.5f * sg; return 0.0f; }
    *slope = 0.0f;
    return 0.0f;
}

static inline void tef_make_splat(float y, float x, int H, int W, tef_splat *s)
{
    float cy[2] = { floorf(y), floorf(y + 1.0f) };
    float cx[2] = { floorf(x), floorf(x + 1.0f) };
    for (int k = 0; k < 4; ++k) {
        float iy = cy[k >> 1], ix = cx[k & 1], sy, sx;
        float wy = tef_hat(y - iy, &sy), wx = tef_hat(x - ix, &sx);
        int ok = (iy >= 0.0f) && (iy < (float)H) && (ix >= 0.0f) && (ix < (float)W);
        s->idx[k] = ok ? (int)iy * W + (int)ix : -1;
        s->w[k] = ok ? wy * wx : 0.0f;
        s->dwy[k] = ok ? sy * wx : 0.0f;
        s->dwx[k] = ok ? wy * sx : 0.0f;
    }
}

/* exported primitive: get_interpolation; idx/w laid out as the reference does: [B, 4N] corner-major blocks */
void tef_oracle_get_interpolation(const float *pos, int B, int N, int H, int W, float *idx, float *w,
                                  const float *r, float *dpos)
{
    for (int b = 0; b < B; ++b)
        for (int e = 0; e < N; ++e) {
            size_t o = ((size_t)b * N + e) * 2;
            tef_splat s;
            tef_make_splat(pos[o], pos[o + 1], H, W, &s);
            float gy = 0.0f, gx = 0.0f;
            for (int k = 0; k < 4; ++k) {
                size_t q = (size_t)b * 4 * N + (size_t)k * N + e;
                idx[q] = s.idx[k] >= 0 ? (float)s.idx[k] : 0.0f;
                w[q] = s.w[k];
                if (r) { gy += r[q] * s.dwy[k]; gx += r[q] * s.dwx[k]; }
            }
            if (dpos) { dpos[o] = gy; dpos[o + 1] = gx; }
        }
}

/* exported primitive: loss/flow.py:81-110 iwe_formatting for one list of N events per sample.
 * pos [B,N,2], mask [B,N,2] (pos,neg), ts [B,N]; out iwe [B,2,H,W], iwe_ts [B,2,H,W]. */
void tef_oracle_iwe_formatting(const float *pos, const float *mask, const float *ts, int B, int N, int H, int W,
                               float tref, float scale, float *iwe, float *iwe_ts)
{
    memset(iwe, 0, sizeof(float) * (size_t)B * 2 * H * W);
    memset(iwe_ts, 0, sizeof(float) * (size_t)B * 2 * H * W);
    for (int b = 0; b < B; ++b)
        for (int k = 0; k < 4; ++k)          /* scatter_add_ visits the 4 corner blocks in turn (:94, :134) */
            for (int e = 0; e < N; ++e) {
                size_t o = (size_t)b * N + e;
                tef_splat s;
                tef_make_splat(pos[o * 2], pos[o * 2 + 1], H, W, &s);
                float tau = 1.0f - fabsf(tref - ts[o]) / scale;     /* :94-95 */
                int p = s.idx[k] >= 0 ? s.idx[k] : 0;               /* masked idx collapses to pixel 0, weight 0 */
                for (int c = 0; c < 2; ++c) {
                    iwe[((size_t)b * 2 + c) * H * W + p] += s.w[k] * mask[o * 2 + c];
                    iwe_ts[((size_t)b * 2 + c) * H * W + p] += (s.w[k] * tau) * mask[o * 2 + c];
                }
            }
}

/* exported primitive: loss/flow.py:112-129 focus_loss on A = iwe_ts / (iwe + 1e-9) (caller divides, :727) */
float tef_oracle_focus_loss(const float *iwe, const float *iwe_ts, int B, int H, int W)
{
    double total = 0.0;
    int HW = H * W;
    for (int b = 0; b < B; ++b) {
        double s = 0.0;
        long nnz = 0;
        for (int p = 0; p < HW; ++p) {
            float c0 = iwe[((size_t)b * 2) * HW + p], c1 = iwe[((size_t)b * 2 + 1) * HW + p];
            float a0 = iwe_ts[((size_t)b * 2) * HW + p] / (c0 + TEF_EPS);
            float a1 = iwe_ts[((size_t)b * 2 + 1) * HW + p] / (c1 + TEF_EPS);
            s += (double)(a0 * a0) + (double)(a1 * a1);
            nnz += ((c0 + c1) != 0.0f);
        }
        total += (double)((float)s / ((float)nnz + TEF_EPS));
    }
    return (float)total;
}

/* ------------------------------------------------------------------------------------------
 * Shared window description.
 *   flows  [P][F][B][2][H][W]   channel 0 = x flow, 1 = y flow (loss/flow.py:62-63)
 *   events SoA per sample, Mt = M + Md slots: grad events of bin 0..P-1 (off[t]..off[t+1]) then
 *   detached events (M + doff[t] ..).  ts already carries the "+ pass index" shift of
 *   loss/flow.py:457-458.
 * ---------------------------------------------------------------------------------------- */
typedef struct {
    int B, H, W, P, F, S;
    int mode_div;                /* Iterative: 1 = "one", 2 = "two" (loss/flow.py:434-441) */
    int M, Md;
    const int *off, *doff;       /* [P+1] each */
    const float *flows;
    const float *ts, *y, *x, *mp, *mn;  /* [B][M+Md] */
    int loss_scaling;            /* loss/flow.py:124-127: divide each image's sum by its number of active pixels */
    int border_compensation;     /* loss/flow.py:671-681 shared mask over a window's reference times (1, the reachable
                                    setting) or each reference time's own cumulative mask (0; Iterative only) */
    float *dmass;                /* optional, laid out like dflows: the gradient's MASS — the same backward pass with every
                                    term replaced by its absolute value (|tau| + |A| for tau - A, |slope| for the hat
                                    derivatives, |J| in the chain, |g| in the scatters), i.e. the sum of the magnitudes
                                    whose signed sum each pixel of d loss / d flow is.  It is the local scale against
                                    which an element-wise comparison of two fp32 gradients is meaningful (a pixel where
                                    large terms cancel cannot agree to 1e-4 of its own value); not a reference output */
} tef_window;

static inline const float *tef_map(const tef_window *wd, int t, int i, int b, int c)
{
    return wd->flows + ((((size_t)t * wd->F + i) * wd->B + b) * 2 + c) * (size_t)wd->H * wd->W;
}
static inline float *tef_dmap(const tef_window *wd, float *dflows, int t, int i, int b, int c)
{
    return dflows + ((((size_t)t * wd->F + i) * wd->B + b) * 2 + c) * (size_t)wd->H * wd->W;
}
/* bilinear scatter of one event's flow-gradient component into map (t, i, b, c) [+ its magnitude into dmass] */
static inline void tef_scatter_grad(const tef_window *wd, float *dflows, int t, int i, int b, int c, const tef_taps *tp,
                                    float g, float m)
{
    tef_tap_scatter(tef_dmap(wd, dflows, t, i, b, c), tp, g);
    if (wd->dmass) tef_tap_scatter(tef_dmap(wd, wd->dmass, t, i, b, c), tp, m);
}

/* One image of warped events for sample b: accumulate grad list and detached list separately
 * (loss/flow.py:698-722), add (:725-726), divide (:727), focus loss (:728, :112-129).
 * py/px: positions per slot; valid: per-slot 0/1 border-compensation flag (:671-681 shared mask).
 * Bins [lo, hi).  Writes A = T/(C+eps) and R = 1/(C+eps) per polarity for the backward. */
typedef struct { float *Cg, *Tg, *Cd, *Td, *A, *R; } tef_imgbuf;

static void tef_accumulate(const tef_window *wd, int b, const float *py, const float *px, const unsigned char *valid,
                           int base, const int *off, int lo, int hi, float tref, float scale, float *C, float *T)
{
    int HW = wd->H * wd->W, Mt = wd->M + wd->Md;
    memset(C, 0, sizeof(float) * 2 * HW);
    memset(T, 0, sizeof(float) * 2 * HW);
    for (int k = 0; k < 4; ++k)
        for (int sl = base + off[lo]; sl < base + off[hi]; ++sl) {
            if (!valid[sl]) continue;     /* mask 0 -> adds exactly 0 in the reference */
            size_t o = (size_t)b * Mt + sl;
            tef_splat s;
            tef_make_splat(py[sl], px[sl], wd->H, wd->W, &s);
            if (s.idx[k] < 0) continue;
            float tau = 1.0f - fabsf(tref - wd->ts[o]) / scale;
            C[s.idx[k]] += s.w[k] * wd->mp[o];
            C[HW + s.idx[k]] += s.w[k] * wd->mn[o];
            T[s.idx[k]] += (s.w[k] * tau) * wd->mp[o];
            T[HW + s.idx[k]] += (s.w[k] * tau) * wd->mn[o];
        }
}

/* returns the per-sample focus loss; *n_out = (#active px + eps) */
static float tef_image_loss(const tef_window *wd, tef_imgbuf *ib, float *n_out)
{
    int HW = wd->H * wd->W;
    double s = 0.0;
    long nnz = 0;
    for (int p = 0; p < HW; ++p) {
        float c0 = ib->Cg[p] + ib->Cd[p], c1 = ib->Cg[HW + p] + ib->Cd[HW + p];
        float t0 = ib->Tg[p] + ib->Td[p], t1 = ib->Tg[HW + p] + ib->Td[HW + p];
        ib->R[p] = 1.0f / (c0 + TEF_EPS);
        ib->R[HW + p] = 1.0f / (c1 + TEF_EPS);
        ib->A[p] = t0 / (c0 + TEF_EPS);
        ib->A[HW + p] = t1 / (c1 + TEF_EPS);
        s += (double)(ib->A[p] * ib->A[p]) + (double)(ib->A[HW + p] * ib->A[HW + p]);
        nnz += ((c0 + c1) != 0.0f);
    }
    *n_out = wd->loss_scaling ? (float)nnz + TEF_EPS : 1.0f;
    return (float)s / *n_out;
}

/* d(coef * image loss) / d position for the grad events of bins [lo, hi); accumulated into gy/gx */
static void tef_image_backward(const tef_window *wd, int b, const tef_imgbuf *ib, float kimg, const float *py,
                               const float *px, const unsigned char *valid, int lo, int hi, float tref, float scale,
                               float *gy, float *gx, float *my, float *mx)
{
    int HW = wd->H * wd->W, Mt = wd->M + wd->Md;
    for (int sl = wd->off[lo]; sl < wd->off[hi]; ++sl) {
        if (!valid[sl]) continue;
        size_t o = (size_t)b * Mt + sl;
        tef_splat s;
        tef_make_splat(py[sl], px[sl], wd->H, wd->W, &s);
        float tau = 1.0f - fabsf(tref - wd->ts[o]) / scale;
        float ay = 0.0f, ax = 0.0f, may = 0.0f, max_ = 0.0f;
        for (int k = 0; k < 4; ++k) {
            if (s.idx[k] < 0) continue;
            int p = s.idx[k];
            /* autograd's own arithmetic, not the closed form 2 A (tau - A) / (C + eps): the division's backward gives
             * dT = gA / (C + eps) and d(C + eps) = -gA * ((T / (C + eps)) / (C + eps)) per pixel (gA = g * 2 A from the
             * square, g = the image's coefficient / n), and an event's weight collects dC + dT * tau from the four
             * scatters (loss/flow.py:100-108, :727).  Mathematically the same; where a pixel holds a single event
             * (tau - A ~ 1e-9 / C) the two forms round differently by 1e-7 of a term that is 1 / C times larger than the
             * result — 1.4e-5 of the gradient's maximum on the BASELINE window, 8e-8 in this form (round 4) */
            float c0 = ib->Cg[p] + ib->Cd[p] + TEF_EPS, c1 = ib->Cg[HW + p] + ib->Cd[HW + p] + TEF_EPS;
            float ga0 = kimg * (2.0f * ib->A[p]), ga1 = kimg * (2.0f * ib->A[HW + p]);
            float dT0 = ga0 / c0, dT1 = ga1 / c1;
            float dC0 = -(ga0 * (ib->A[p] / c0)), dC1 = -(ga1 * (ib->A[HW + p] / c1));
            float dw = (dC0 * wd->mp[o] + dC1 * wd->mn[o]) + (dT0 * wd->mp[o] + dT1 * wd->mn[o]) * tau;
            ay += dw * s.dwy[k];
            ax += dw * s.dwx[k];
            if (my) {       /* mass: the same expression with every term's magnitude */
                float dm = fabsf(wd->mp[o]) * (2.0f * fabsf(ib->A[p]) * (fabsf(tau) + fabsf(ib->A[p])) * ib->R[p])
                         + fabsf(wd->mn[o]) * (2.0f * fabsf(ib->A[HW + p]) * (fabsf(tau) + fabsf(ib->A[HW + p])) * ib->R[HW + p]);
                dm *= fabsf(kimg);
                may += dm * fabsf(s.dwy[k]);
                max_ += dm * fabsf(s.dwx[k]);
            }
        }
        gy[sl] += ay;
        gx[sl] += ax;
        if (my) { my[sl] += may; mx[sl] += max_; }
    }
}

static tef_imgbuf tef_imgbuf_new(int HW)
{
    tef_imgbuf ib;
    ib.Cg = (float *)malloc(sizeof(float) * 2 * HW); ib.Tg = (float *)malloc(sizeof(float) * 2 * HW);
    ib.Cd = (float *)malloc(sizeof(float) * 2 * HW); ib.Td = (float *)malloc(sizeof(float) * 2 * HW);
    ib.A = (float *)malloc(sizeof(float) * 2 * HW);  ib.R = (float *)malloc(sizeof(float) * 2 * HW);
    return ib;
}
static void tef_imgbuf_free(tef_imgbuf *ib)
{
    free(ib->Cg); free(ib->Tg); free(ib->Cd); free(ib->Td); free(ib->A); free(ib->R);
}

static int tef_bin_of(const int *off, int P, int sl)
{
    int t = 0;
    while (t + 1 < P && sl >= off[t + 1]) ++t;
    return t;
}

/* ------------------------------------------------------------------------------------------
 * Iterative contrast-maximisation loss, forward + backward.
 * loss/flow.py:415-746 (Iterative.__init__/update/event_warping/forward).
 * dflows may be NULL (forward only).  Returns the loss (without smoothing terms).
 * ---------------------------------------------------------------------------------------- */
static double tef_iterative_pair(const tef_window *wd, const int *bin, int i, int b, float *dflows, float grad_out)
{
    const int H = wd->H, W = wd->W, P = wd->P, F = wd->F, S = wd->S;
    const int HW = H * W, M = wd->M, Md = wd->Md, Mt = M + Md;
    float *ty = (float *)malloc(sizeof(float) * (size_t)(P + 1) * (Mt + 1));   /* trajectory, plane k = tref */
    float *tx = (float *)malloc(sizeof(float) * (size_t)(P + 1) * (Mt + 1));
    int *kb = (int *)malloc(sizeof(int) * (Mt + 1)), *kf = (int *)malloc(sizeof(int) * (Mt + 1));
    float *gy = (float *)malloc(sizeof(float) * (size_t)(P + 1) * (M + 1));
    float *gx = (float *)malloc(sizeof(float) * (size_t)(P + 1) * (M + 1));
    const int want_mass = dflows && wd->dmass;
    float *gmy = want_mass ? (float *)calloc((size_t)(P + 1) * (M + 1), sizeof(float)) : 0;
    float *gmx = want_mass ? (float *)calloc((size_t)(P + 1) * (M + 1), sizeof(float)) : 0;
    unsigned char *valid = (unsigned char *)malloc(Mt + 1);
    tef_imgbuf ib = tef_imgbuf_new(HW);
    double loss_i = 0.0;
    {
        {
            /* (A) iterative warping of every event to every tref (loss/flow.py:599-654, :521-586) */
            for (int sl = 0; sl < Mt; ++sl) {
                size_t o = (size_t)b * Mt + sl;
                int t = bin[sl];
                float ts = wd->ts[o], y0 = wd->y[o], x0 = wd->x[o];
                kf[sl] = P + 1;
                kb[sl] = -1;
                /* forward: flow map t, t+1, ... , P-1 ; warping_ts = t+1 ... P (:505-510) */
                float y = y0, x = x0, wts = ts;
                for (int k = t; k < P; ++k) {
                    tef_taps tp;
                    tef_make_taps(y, x, H, W, &tp);
                    float fy = tef_tap_value(tef_map(wd, k, i, b, 1), &tp);
                    float fx = tef_tap_value(tef_map(wd, k, i, b, 0), &tp);
                    float dt = (float)(k + 1) - wts;            /* utils/iwe.py:14 */
                    y = y + dt * fy;
                    x = x + dt * fx;
                    wts = (float)(k + 1);
                    ty[(size_t)(k + 1) * Mt + sl] = y;
                    tx[(size_t)(k + 1) * Mt + sl] = x;
                    if (!tef_inbounds(y, x, H, W)) { kf[sl] = k + 1; break; }   /* cumulative purge (:575) */
                }
                /* backward: flow map t, t-1, ..., 0 ; warping_ts = t ... 0 (:512-514) */
                y = y0; x = x0; wts = ts;
                for (int k = t; k >= 0; --k) {
                    tef_taps tp;
                    tef_make_taps(y, x, H, W, &tp);
                    float fy = tef_tap_value(tef_map(wd, k, i, b, 1), &tp);
                    float fx = tef_tap_value(tef_map(wd, k, i, b, 0), &tp);
                    float dt = (float)k - wts;
                    y = y + dt * fy;
                    x = x + dt * fx;
                    wts = (float)k;
                    ty[(size_t)k * Mt + sl] = y;
                    tx[(size_t)k * Mt + sl] = x;
                    if (!tef_inbounds(y, x, H, W)) { kb[sl] = k; break; }
                }
            }
            if (dflows) {
                memset(gy, 0, sizeof(float) * (size_t)(P + 1) * (M + 1));
                memset(gx, 0, sizeof(float) * (size_t)(P + 1) * (M + 1));
            }
            /* (B) per temporal scale / window / reference time (loss/flow.py:657-731) */
            for (int s = 0; s < S; ++s) {
                int scale = P >> s;                       /* :42-44 passes_loss // 2**s */
                int delta = scale / wd->mode_div;         /* :434-441 */
                float coef = 1.0f / ((float)(1 << s) * (float)(2 * delta + 1) * (float)S * (float)F);
                for (int w = 0; w < (1 << s); ++w) {
                    int lo = w * scale, hi = (w + 1) * scale;
                    /* shared border-compensation mask: in bounds at every tref of the window (:671-681) */
                    for (int sl = 0; sl < Mt; ++sl)
                        valid[sl] = (bin[sl] >= lo && bin[sl] < hi && kb[sl] < lo && kf[sl] > hi);
                    for (int tref = lo; tref <= hi; ++tref) {
                        if (!wd->border_compensation)    /* :691-693: the mask the chain carried to this reference time */
                            for (int sl = 0; sl < Mt; ++sl)
                                valid[sl] = (bin[sl] >= lo && bin[sl] < hi && kb[sl] < tref && tref < kf[sl]);
                        int le = tref - delta > lo ? tref - delta : lo;        /* :685 */
                        int he = tref + delta < hi ? tref + delta : hi;        /* :686 */
                        const float *py = ty + (size_t)tref * Mt, *px = tx + (size_t)tref * Mt;
                        tef_accumulate(wd, b, py, px, valid, 0, wd->off, le, he, (float)tref, (float)delta, ib.Cg, ib.Tg);
                        tef_accumulate(wd, b, py, px, valid, M, wd->doff, le, he, (float)tref, (float)delta, ib.Cd, ib.Td);
                        float n;
                        float l = tef_image_loss(wd, &ib, &n);
                        loss_i += (double)l * coef;
                        if (dflows)
                            tef_image_backward(wd, b, &ib, grad_out * coef / n, py, px, valid, le, he, (float)tref,
                                               (float)delta, gy + (size_t)tref * M, gx + (size_t)tref * M,
                                               want_mass ? gmy + (size_t)tref * M : 0, want_mass ? gmx + (size_t)tref * M : 0);
                    }
                }
            }
            if (!dflows) goto done;
            /* (C) reverse sweep along each grad event's trajectory */
            for (int sl = 0; sl < M; ++sl) {
                size_t o = (size_t)b * Mt + sl;
                int t = bin[sl];
                float ts = wd->ts[o];
                float c0y = 0.0f, c0x = 0.0f;   /* gradient w.r.t. the flow sampled at the original location (map t) */
                float m0y = 0.0f, m0x = 0.0f, my = 0.0f, mx = 0.0f;      /* their masses (wd->dmass only) */
                /* forward chain: p_{k} = p_{k-1} + c * f_{k-1}(p_{k-1}),  k = t+1..P */
                float ay = 0.0f, ax = 0.0f;
                for (int k = P; k > t; --k) {
                    if (k >= kf[sl]) continue;
                    ay += gy[(size_t)k * M + sl];
                    ax += gx[(size_t)k * M + sl];
                    if (want_mass) { my += gmy[(size_t)k * M + sl]; mx += gmx[(size_t)k * M + sl]; }
                    if (k - 1 == t) {
                        float c = (float)(t + 1) - ts;
                        c0y += c * ay; c0x += c * ax;
                        m0y += fabsf(c) * my; m0x += fabsf(c) * mx;
                    } else {
                        tef_taps tp;
                        tef_make_taps(ty[(size_t)(k - 1) * Mt + sl], tx[(size_t)(k - 1) * Mt + sl], H, W, &tp);
                        tef_scatter_grad(wd, dflows, k - 1, i, b, 1, &tp, ay, my);
                        tef_scatter_grad(wd, dflows, k - 1, i, b, 0, &tp, ax, mx);
                        float jyy, jyx, jxy, jxx;
                        tef_tap_jacobian(tef_map(wd, k - 1, i, b, 1), &tp, &jyy, &jyx);
                        tef_tap_jacobian(tef_map(wd, k - 1, i, b, 0), &tp, &jxy, &jxx);
                        float ny = ay + (ay * jyy + ax * jxy), nx = ax + (ay * jyx + ax * jxx);
                        ay = ny; ax = nx;
                        float nmy = my + (my * fabsf(jyy) + mx * fabsf(jxy)), nmx = mx + (my * fabsf(jyx) + mx * fabsf(jxx));
                        my = nmy; mx = nmx;
                    }
                }
                /* backward chain: p_k = q + c * f_k(q), q = p_{k+1} (k < t) or the original location (k = t) */
                ay = 0.0f; ax = 0.0f;
                my = 0.0f; mx = 0.0f;
                for (int k = 0; k <= t; ++k) {
                    if (k <= kb[sl]) continue;
                    ay += gy[(size_t)k * M + sl];
                    ax += gx[(size_t)k * M + sl];
                    if (want_mass) { my += gmy[(size_t)k * M + sl]; mx += gmx[(size_t)k * M + sl]; }
                    if (k == t) {
                        float c = (float)t - ts;
                        c0y += c * ay; c0x += c * ax;
                        m0y += fabsf(c) * my; m0x += fabsf(c) * mx;
                    } else {
                        tef_taps tp;
                        tef_make_taps(ty[(size_t)(k + 1) * Mt + sl], tx[(size_t)(k + 1) * Mt + sl], H, W, &tp);
                        tef_scatter_grad(wd, dflows, k, i, b, 1, &tp, -ay, my);
                        tef_scatter_grad(wd, dflows, k, i, b, 0, &tp, -ax, mx);
                        float jyy, jyx, jxy, jxx;
                        tef_tap_jacobian(tef_map(wd, k, i, b, 1), &tp, &jyy, &jyx);
                        tef_tap_jacobian(tef_map(wd, k, i, b, 0), &tp, &jxy, &jxx);
                        float ny = ay - (ay * jyy + ax * jxy), nx = ax - (ay * jyx + ax * jxx);
                        ay = ny; ax = nx;
                        float nmy = my + (my * fabsf(jyy) + mx * fabsf(jxy)), nmx = mx + (my * fabsf(jyx) + mx * fabsf(jxx));
                        my = nmy; mx = nmx;
                    }
                }
                tef_taps tp;
                tef_make_taps(wd->y[o], wd->x[o], H, W, &tp);
                tef_scatter_grad(wd, dflows, t, i, b, 1, &tp, c0y, m0y);
                tef_scatter_grad(wd, dflows, t, i, b, 0, &tp, c0x, m0x);
            }
        }
    }
done:
    free(ty); free(tx); free(kb); free(kf); free(gy); free(gx); free(gmy); free(gmx); free(valid);
    tef_imgbuf_free(&ib);
    return loss_i;
}

float tef_oracle_iterative(const tef_window *wd, float *dflows, float grad_out)
{
    const int B = wd->B, P = wd->P, F = wd->F, M = wd->M, Md = wd->Md, Mt = M + Md;
    if (dflows) memset(dflows, 0, sizeof(float) * (size_t)P * F * B * 2 * wd->H * wd->W);
    int *bin = (int *)malloc(sizeof(int) * (Mt + 1));
    for (int sl = 0; sl < M; ++sl) bin[sl] = tef_bin_of(wd->off, P, sl);
    for (int sl = 0; sl < Md; ++sl) bin[M + sl] = tef_bin_of(wd->doff, P, sl);
    double loss = 0.0;
    /* (head, sample) pairs are independent until the final sum (loss/flow.py:129, :735-736) */
#pragma omp parallel for schedule(dynamic) reduction(+ : loss)
    for (int q = 0; q < F * B; ++q)
        loss += tef_iterative_pair(wd, bin, q / B, q % B, dflows, grad_out);
    free(bin);
    return (float)loss;
}

/* ------------------------------------------------------------------------------------------
 * Linear contrast-maximisation loss (NeurIPS'21), forward + backward.
 * loss/flow.py:216-412.  Per-event flow is sampled once from the map of the event's own pass
 * (:268-283); events are warped linearly to both window ends (:337-338) with a shared
 * border mask (:341-343).
 * ---------------------------------------------------------------------------------------- */
float tef_oracle_linear(const tef_window *wd, float *dflows, float grad_out)
{
    const int B = wd->B, H = wd->H, W = wd->W, P = wd->P, F = wd->F, S = wd->S;
    const int HW = H * W, M = wd->M, Md = wd->Md, Mt = M + Md;
    if (dflows) memset(dflows, 0, sizeof(float) * (size_t)P * F * B * 2 * HW);
    float *efy = (float *)malloc(sizeof(float) * Mt), *efx = (float *)malloc(sizeof(float) * Mt);
    float *py = (float *)malloc(sizeof(float) * 2 * Mt), *px = (float *)malloc(sizeof(float) * 2 * Mt);
    float *gy = (float *)malloc(sizeof(float) * 2 * (M + 1)), *gx = (float *)malloc(sizeof(float) * 2 * (M + 1));
    float *cy = (float *)malloc(sizeof(float) * (M + 1)), *cx = (float *)malloc(sizeof(float) * (M + 1));
    const int want_mass = dflows && wd->dmass;       /* the gradient's mass (see tef_window.dmass) */
    float *gmy = want_mass ? (float *)malloc(sizeof(float) * 2 * (M + 1)) : 0, *gmx = want_mass ? (float *)malloc(sizeof(float) * 2 * (M + 1)) : 0;
    float *cmy = want_mass ? (float *)malloc(sizeof(float) * (M + 1)) : 0, *cmx = want_mass ? (float *)malloc(sizeof(float) * (M + 1)) : 0;
    int *bin = (int *)malloc(sizeof(int) * Mt);
    unsigned char *valid = (unsigned char *)malloc(Mt + 1);
    tef_imgbuf ib = tef_imgbuf_new(HW);
    for (int sl = 0; sl < M; ++sl) bin[sl] = tef_bin_of(wd->off, P, sl);
    for (int sl = 0; sl < Md; ++sl) bin[M + sl] = tef_bin_of(wd->doff, P, sl);

    double loss = 0.0;
    for (int i = 0; i < F; ++i)
        for (int b = 0; b < B; ++b) {
            for (int sl = 0; sl < Mt; ++sl) {
                size_t o = (size_t)b * Mt + sl;
                tef_taps tp;
                tef_make_taps(wd->y[o], wd->x[o], H, W, &tp);
                efy[sl] = tef_tap_value(tef_map(wd, bin[sl], i, b, 1), &tp);
                efx[sl] = tef_tap_value(tef_map(wd, bin[sl], i, b, 0), &tp);
            }
            if (dflows) { memset(cy, 0, sizeof(float) * (M + 1)); memset(cx, 0, sizeof(float) * (M + 1)); }
            if (want_mass) { memset(cmy, 0, sizeof(float) * (M + 1)); memset(cmx, 0, sizeof(float) * (M + 1)); }
            for (int s = 0; s < S; ++s) {
                int scale = P >> s;
                float coef = 1.0f / ((float)(1 << s) * 2.0f * (float)S * (float)F);   /* :396-397, :401-402 */
                for (int w = 0; w < (1 << s); ++w) {
                    int lo = w * scale, hi = (w + 1) * scale;
                    for (int sl = 0; sl < Mt; ++sl) {
                        size_t o = (size_t)b * Mt + sl;
                        valid[sl] = 0;
                        if (bin[sl] < lo || bin[sl] >= hi) continue;
                        float ts = wd->ts[o];
                        for (int e = 0; e < 2; ++e) {        /* e = 0: forward to hi, e = 1: backward to lo */
                            float dt = (float)(e == 0 ? hi : lo) - ts;
                            py[(size_t)e * Mt + sl] = wd->y[o] + dt * efy[sl];
                            px[(size_t)e * Mt + sl] = wd->x[o] + dt * efx[sl];
                        }
                        /* shared purge of both ends (:341-343); without border compensation nothing is purged (:324-328) */
                        valid[sl] = !wd->border_compensation ||
                                    (tef_inbounds(py[sl], px[sl], H, W) && tef_inbounds(py[Mt + sl], px[Mt + sl], H, W));
                    }
                    if (dflows) { memset(gy, 0, sizeof(float) * 2 * (M + 1)); memset(gx, 0, sizeof(float) * 2 * (M + 1)); }
                    if (want_mass) { memset(gmy, 0, sizeof(float) * 2 * (M + 1)); memset(gmx, 0, sizeof(float) * 2 * (M + 1)); }
                    for (int e = 0; e < 2; ++e) {
                        float tref = (float)(e == 0 ? hi : lo);
                        const float *qy = py + (size_t)e * Mt, *qx = px + (size_t)e * Mt;
                        tef_accumulate(wd, b, qy, qx, valid, 0, wd->off, lo, hi, tref, (float)scale, ib.Cg, ib.Tg);
                        tef_accumulate(wd, b, qy, qx, valid, M, wd->doff, lo, hi, tref, (float)scale, ib.Cd, ib.Td);
                        float n;
                        float l = tef_image_loss(wd, &ib, &n);
                        loss += (double)l * coef;
                        if (dflows)
                            tef_image_backward(wd, b, &ib, grad_out * coef / n, qy, qx, valid, lo, hi, tref, (float)scale,
                                               gy + (size_t)e * M, gx + (size_t)e * M,
                                               want_mass ? gmy + (size_t)e * M : 0, want_mass ? gmx + (size_t)e * M : 0);
                    }
                    if (dflows)
                        for (int sl = wd->off[lo]; sl < wd->off[hi]; ++sl) {
                            float ts = wd->ts[(size_t)b * Mt + sl];
                            cy[sl] += ((float)hi - ts) * gy[sl] + ((float)lo - ts) * gy[M + sl];
                            cx[sl] += ((float)hi - ts) * gx[sl] + ((float)lo - ts) * gx[M + sl];
                            if (want_mass) {
                                cmy[sl] += fabsf((float)hi - ts) * gmy[sl] + fabsf((float)lo - ts) * gmy[M + sl];
                                cmx[sl] += fabsf((float)hi - ts) * gmx[sl] + fabsf((float)lo - ts) * gmx[M + sl];
                            }
                        }
                }
            }
            if (dflows)
                for (int sl = 0; sl < M; ++sl) {
                    size_t o = (size_t)b * Mt + sl;
                    tef_taps tp;
                    tef_make_taps(wd->y[o], wd->x[o], H, W, &tp);
                    tef_scatter_grad(wd, dflows, bin[sl], i, b, 1, &tp, cy[sl], want_mass ? cmy[sl] : 0.0f);
                    tef_scatter_grad(wd, dflows, bin[sl], i, b, 0, &tp, cx[sl], want_mass ? cmx[sl] : 0.0f);
                }
        }
    free(efy); free(efx); free(py); free(px); free(gy); free(gx); free(cy); free(cx); free(bin); free(valid);
    free(gmy); free(gmx); free(cmy); free(cmx);
    tef_imgbuf_free(&ib);
    return (float)loss;
}

/* ------------------------------------------------------------------------------------------
 * Smoothness priors on the flow maps of one window (loss/flow.py:170-209 spatial, :131-168 temporal).
 * Both ADD into dflows (caller zeroes / has the CM gradient there already).  Return the weighted term.
 * ---------------------------------------------------------------------------------------- */
static inline float tef_charb(float d, float eps, float *g)
{
    float r = sqrtf(d * d + eps);
    *g = d / r;
    return r;
}

float tef_oracle_spatial_smoothing(const tef_window *wd, float weight, float *dflows, float grad_out)
{
    const int B = wd->B, H = wd->H, W = wd->W, P = wd->P, F = wd->F, HW = H * W;
    double total = 0.0;
    /* four difference families: (dy,dx) offsets between the two pixels, with the number of terms of each */
    const int oy[4] = { 0, 1, 1, -1 }, ox[4] = { 1, 0, 1, 1 };
    for (int i = 0; i < F; ++i)
        for (int b = 0; b < B; ++b)
            for (int fam = 0; fam < 4; ++fam) {
                int ny = H - (oy[fam] != 0), nx = W - (ox[fam] != 0);
                double acc = 0.0;
                float scale = weight / (4.0f * (float)F * (float)P * (float)(ny * nx));
                for (int t = 0; t < P; ++t)
                    for (int c = 0; c < 2; ++c) {
                        const float *m = tef_map(wd, t, i, b, c);
                        float *dm = dflows ? tef_dmap(wd, dflows, t, i, b, c) : 0;
                        for (int yy = 0; yy < ny; ++yy)
                            for (int xx = 0; xx < nx; ++xx) {
                                /* :180-187 — first pixel minus the pixel one step along the family direction */
                                int ya = oy[fam] < 0 ? yy + 1 : yy, yb = oy[fam] < 0 ? yy : yy + oy[fam];
                                int pa = ya * W + xx, pb = yb * W + xx + ox[fam];
                                float g, r = tef_charb(m[pa] - m[pb], 1e-6f, &g);
                                acc += r;
                                if (dm) { dm[pa] += grad_out * scale * g; dm[pb] -= grad_out * scale * g; }
                            }
                    }
                total += acc * scale;
            }
    (void)HW;
    return (float)total;
}

float tef_oracle_temporal_smoothing(const tef_window *wd, float weight, float *dflows, float grad_out)
{
    const int B = wd->B, H = wd->H, W = wd->W, P = wd->P, F = wd->F, HW = H * W;
    if (P < 2) return 0.0f;
    double total = 0.0;
    float *gyv = (float *)malloc(sizeof(float) * HW), *gxv = (float *)malloc(sizeof(float) * HW);
    for (int i = 0; i < F; ++i)
        for (int j = 0; j + 1 < P; ++j)
            for (int b = 0; b < B; ++b) {
                const float *fx0 = tef_map(wd, j, i, b, 0), *fy0 = tef_map(wd, j, i, b, 1);
                const float *fx1 = tef_map(wd, j + 1, i, b, 0), *fy1 = tef_map(wd, j + 1, i, b, 1);
                double acc = 0.0;
                long cnt = 0;
                for (int p = 0; p < HW; ++p) {
                    float wy = (float)(p / W) + fy0[p], wx = (float)(p % W) + fx0[p];   /* :143 */
                    if (!tef_inbounds(wy, wx, H, W)) { gyv[p] = gxv[p] = 0.0f; continue; }   /* :147-152 */
                    tef_taps tp;
                    tef_make_taps(wy, wx, H, W, &tp);
                    float sy = tef_tap_value(fy1, &tp), sx = tef_tap_value(fx1, &tp);       /* :155-157 */
                    float g0, g1;
                    acc += tef_charb(fy0[p] - sy, 1e-9f, &g0) + tef_charb(fx0[p] - sx, 1e-9f, &g1);   /* :161-162 */
                    gyv[p] = g0; gxv[p] = g1;
                    ++cnt;
                }
                float denom = (float)cnt + TEF_EPS;
                float scale = weight / ((float)F * (float)(P - 1) * denom);
                total += acc * scale;
                if (!dflows) continue;
                float *dx0 = tef_dmap(wd, dflows, j, i, b, 0), *dy0 = tef_dmap(wd, dflows, j, i, b, 1);
                float *dx1 = tef_dmap(wd, dflows, j + 1, i, b, 0), *dy1 = tef_dmap(wd, dflows, j + 1, i, b, 1);
                for (int p = 0; p < HW; ++p) {
                    float wy = (float)(p / W) + fy0[p], wx = (float)(p % W) + fx0[p];
                    if (!tef_inbounds(wy, wx, H, W)) continue;
                    float gy_ = grad_out * scale * gyv[p], gx_ = grad_out * scale * gxv[p];
                    tef_taps tp;
                    tef_make_taps(wy, wx, H, W, &tp);
                    /* direct term, sampled-value term (map j+1), sampling-location term (map j) */
                    dy0[p] += gy_; dx0[p] += gx_;
                    tef_tap_scatter(dy1, &tp, -gy_);
                    tef_tap_scatter(dx1, &tp, -gx_);
                    float jyy, jyx, jxy, jxx;
                    tef_tap_jacobian(fy1, &tp, &jyy, &jyx);
                    tef_tap_jacobian(fx1, &tp, &jxy, &jxx);
                    dy0[p] += -(gy_ * jyy + gx_ * jxy);
                    dx0[p] += -(gy_ * jyx + gx_ * jxx);
                }
            }
    free(gyv); free(gxv);
    return (float)total;
}

/* ------------------------------------------------------------------------------------------
 * Input encodings, dataloader/encodings.py.
 * ---------------------------------------------------------------------------------------- */
/* :59-81 events_to_channels -> [2,H,W] (positive counts for both polarities) */
void tef_oracle_events_to_channels(const float *xs, const float *ys, const float *ps, int N, int H, int W, float *out)
{
    memset(out, 0, sizeof(float) * 2 * H * W);
    for (int e = 0; e < N; ++e) {
        int p = (int)ys[e] * W + (int)xs[e];
        float mpos = ps[e] > 0 ? 1.0f : (ps[e] < 0 ? 0.0f : ps[e]);
        float mneg = ps[e] < 0 ? -1.0f : (ps[e] > 0 ? 0.0f : ps[e]);
        out[p] += ps[e] * mpos;
        out[H * W + p] += ps[e] * mneg;
    }
}

/* :32-56 events_to_voxel -> [bins,H,W] (signed, temporal bilinear weights) */
void tef_oracle_events_to_voxel(const float *xs, const float *ys, const float *ts, const float *ps, int N, int bins,
                                int H, int W, float *out)
{
    memset(out, 0, sizeof(float) * (size_t)bins * H * W);
    for (int bi = 0; bi < bins; ++bi)
        for (int e = 0; e < N; ++e) {
            float t = ts[e] * (float)(bins - 1);
            float w = 1.0f - fabsf(t - (float)bi);
            if (w < 0.0f) w = 0.0f;
            out[(size_t)bi * H * W + (int)ys[e] * W + (int)xs[e]] += ps[e] * w;
        }
}

/* :8-29 events_to_image */
void tef_oracle_events_to_image(const float *xs, const float *ys, const float *ps, int N, int H, int W, float *out)
{
    memset(out, 0, sizeof(float) * H * W);
    for (int e = 0; e < N; ++e) out[(int)ys[e] * W + (int)xs[e]] += ps[e];
}

int tef_oracle_version(void) { return 1; }

/* number of threads the OpenMP build uses (1 when built without OpenMP); n > 0 sets it first */
int tef_oracle_threads(int n)
{
#ifdef _OPENMP
    if (n > 0) omp_set_num_threads(n);
    return omp_get_max_threads();
#else
    (void)n;
    return 1;
#endif
}
