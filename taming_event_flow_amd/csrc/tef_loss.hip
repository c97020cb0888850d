// tef_loss.hip — event warping + image-of-warped-events (IWE) scatter + timestamp contrast loss,
// forward and backward, for gfx950 (MI355X).  Replaces the tensor-of-temporaries formulation of the
// reference (loss/flow.py:415-746 Iterative, :216-412 Linear, utils/iwe.py:5-136) by a per-event
// trajectory formulation:
//
//   forward   K1 warp      one thread per (head, sample, event): walks the event through the flow maps,
//                          stores its position at every reference time (trajectory planes) and the
//                          border-compensation flags (loss/flow.py:671-681)
//             K2 splat     one workgroup per (image, head, sample, polarity, quantity[, row band]): one
//                          quantity (event count C or weighted timestamp sum T) of one polarity of one
//                          image lives in LDS as fp64 (128 KiB for 128x128); events stream in coalesced,
//                          bilinear corners go in with ds_add_f64
//             K3 stats     per image: A = T/(C+eps), R = 1/(C+eps) for the backward, sum of squared mean
//                          timestamps, number of active pixels
//             K4 reduce    deterministic sum of the per-image terms -> scalar loss
//   backward  K6 chain     one thread per (head, sample, grad event): gathers d loss / d position at
//                          every reference time from the (A, R) images, reverse sweep along the
//                          trajectory (grid_sample backward w.r.t. grid), emits one flow-gradient
//                          vector per (event, flow map)
//             K7 dflow     one workgroup per (flow map, head, sample, component[, row band]): bilinear
//                          splat of those vectors into an LDS-resident fp64 gradient map, plain coalesced
//                          write-out (no global atomics, no zero-fill pass)
//
// Why fp64 in LDS: on gfx950 ds_add_f32 sustains only ~0.2 T atomics/s chip-wide, ds_add_f64 ~1.3 T/s
// (tools/lds_atomic_bench.hip); fp64 sums also make the result independent of the arrival order to
// far below fp32 resolution.
//
// Arithmetic is fp32 and follows the reference / ATen op order where it matters for parity
// (coordinate normalisation + un-normalisation of grid_sample, floor(y + 1) corners, true division
// for the timestamp normalisation); compiled with -ffp-contract=off.
#include <hip/hip_runtime.h>
#include <hip/hip_ext.h>
#include <stdint.h>
#include <stdio.h>
#include <string.h>

#include <algorithm>
#include <type_traits>

#include "tef.h"
#include "tef_common.h"

namespace {

constexpr float kEps = 1e-9f;
// K7: TWO 512-thread workgroups per CU, each with the two planes of a 32-row band (+ 2 halo rows) at 128 x 128 (round 5; one
// 1024-thread workgroup with 64-row bands before: 640 items over 256 workgroups are 2.5 each — the queue's longest-first
// order still left a sixth of the kernel to the workgroups that drew three; 1280 items over 512 balance to a few per cent,
// and one workgroup's clearing / write-out overlaps the other's sweep: 0.134 -> 0.119 ms)
constexpr int kSplatThreads = 512;
constexpr size_t kSplatLdsBudget = 73 * 1024;    // (beside ~5 KiB of run tables and per-wavefront row lists)
constexpr int kRowPad = 8;                  // 8-byte LDS rows are W + 8 wide: rows 16 banks apart, so the few-row
                                            // neighbourhood a sorted wavefront hits spreads over all 64 banks
constexpr int kMaxImages = 448;             // sum_s 2^s * (P/2^s + 1) <= 6*64 + 63
constexpr int kQueueInts = 24;                // work queues of the persistent scatter kernels: [0, 8) images, [8, 16) flow gradients; [16] = K6's deferred wavefronts
constexpr int kDeferWord = 16;

// meta word written by K1 per (head, sample, slot): .x = flags below | border bits per scale | kb + 1 | kf, .y = the
// event's timestamp (bit pattern), so that the scatter reads everything but the position of an event in ONE 8-byte load
constexpr uint32_t kMetaPos = 1u << 24;      // mask_pos != 0
constexpr uint32_t kMetaNeg = 1u << 25;      // mask_neg != 0
constexpr uint32_t kMetaNonUnit = 1u << 26;  // a mask value is neither 0 nor 1: fetch the float

// ---------------------------------------------------------------------------------------------
// Window descriptor handed to kernels by value (kernarg segment).
// ---------------------------------------------------------------------------------------------
struct Win {
    int kind, B, H, W, P, F, S, mode_div, M, Md, Mt, nplanes, nimg, scaling;
    int comp;                        // border compensation (loss/flow.py:671-681): shared mask over the window's reference times
    int nrow;                        // rows of 16 slots: ceil(Mt / 16)
    int img_base[TEF_MAX_SCALES + 1];
    int off[TEF_MAX_PASSES + 1];
    int doff[TEF_MAX_PASSES + 1];
    uint16_t order[kMaxImages];      // images sorted by decreasing number of bins (longest workgroups first)
    uint8_t korder[TEF_MAX_PASSES];  // flow maps sorted by decreasing number of passes that feed them (K7's items)
};

struct Events {
    const float *ts, *y, *x, *mp, *mn;
    const uint8_t *bin;
    const int *cls;      // [B][TEF_MAX_PASSES][3]: end of the pos-only / neg-only / both-polarity run of each pass
    int cap;
};

struct Img {
    int s, plane, le, he, lo, hi;
    float tref, delta, inv_delta, coef;
};

__host__ __device__ inline int images_of_scale(int kind, int P, int s)
{
    int scale = P >> s;
    return (kind == TEF_KIND_ITERATIVE) ? (1 << s) * (scale + 1) : (1 << s) * 2;
}

// image index -> (scale, window, reference time, bin range, normalisation)   loss/flow.py:657-731 / :309-397
__host__ __device__ inline Img decode_image(const Win &w, int j)
{
    Img im;
    int s = 0;
    while (s + 1 < w.S && j >= w.img_base[s + 1]) ++s;
    int r = j - w.img_base[s];
    int scale = w.P >> s;
    im.s = s;
    if (w.kind == TEF_KIND_ITERATIVE) {
        int per = scale + 1, wi = r / per, q = r - wi * per;
        int delta = scale / w.mode_div;
        im.lo = wi * scale;
        im.hi = im.lo + scale;
        int tref = im.lo + q;
        im.le = (im.lo > tref - delta) ? im.lo : tref - delta;
        im.he = (im.hi < tref + delta) ? im.hi : tref + delta;
        im.plane = tref;
        im.tref = (float)tref;
        im.delta = (float)delta;
        im.inv_delta = 1.0f / (float)delta;
        im.coef = 1.0f / ((float)(1 << s) * (float)(2 * delta + 1) * (float)w.S * (float)w.F);
    } else {
        int wi = r >> 1, e = r & 1;
        im.lo = wi * scale;
        im.hi = im.lo + scale;
        im.le = im.lo;
        im.he = im.hi;
        im.plane = 2 * s + e;
        im.tref = (float)(e ? im.lo : im.hi);
        im.delta = (float)scale;
        im.inv_delta = 1.0f / (float)scale;
        im.coef = 1.0f / ((float)(1 << s) * 2.0f * (float)w.S * (float)w.F);
    }
    return im;
}

// XCD-aware work distribution: workgroups are dealt round-robin over the 8 XCDs (bid % 8), each with a
// private 4 MiB L2.  `item` indexes things that should share an L2 (all chunks of one (head, sample) pair,
// all variants of one image) so that their common inputs are fetched from HBM once per XCD.
// bid -> (item, sub) with item % 8 == XCD; grid = 8 * ceil(nitems / 8) * nsub, guard item < nitems.  [speed only]
__device__ __forceinline__ void xcd_split(int bid, int nsub, int &item, int &sub)
{
    int g = bid & 7, r = bid >> 3;
    item = g + 8 * (r / nsub);
    sub = r % nsub;
}
inline unsigned xcd_grid(int nitems, int nsub) { return 8u * (unsigned)((nitems + 7) / 8) * (unsigned)nsub; }

// wave-uniform base + 32-bit byte offset per lane: the form the scalar-base global loads / stores take
// (the uniform element offset goes through readfirstlane so that loop strength reduction cannot turn base + k * stride +
// lane into a 64-bit per-lane induction pointer)
__device__ __forceinline__ size_t uniform_off(size_t v)
{
    return (size_t)(uint32_t)__builtin_amdgcn_readfirstlane((int)(uint32_t)v) |
           ((size_t)(uint32_t)__builtin_amdgcn_readfirstlane((int)(uint32_t)(v >> 32)) << 32);
}
template <class T>
__device__ __forceinline__ T *at_bytes(T *base, uint32_t off)
{
    typedef typename std::conditional<std::is_const<T>::value, const char, char>::type B;
    return reinterpret_cast<T *>(reinterpret_cast<B *>(base) + off);
}

// ---------------------------------------------------------------------------------------------
// Bilinear flow lookup: utils/iwe.py:17-40 + ATen grid_sampler_2d (bilinear, align_corners=True, zeros).
// Flow maps are interleaved float2 (.x = flow_y, .y = flow_x) per pixel: one 8-byte tap per corner.
// ---------------------------------------------------------------------------------------------
struct Taps {
    int i00, i01, i10, i11;   // -1 when outside
    float s, n, e, w;         // (1-fy), fy, (1-fx), fx
};

// Correctly rounded a / b for an integer constant b (1 <= b < 2^24), bit for bit IEEE division, in three instructions
// instead of the ~12 of the fp32 division sequence (the chain kernels are VALU-bound): RN_f32((double)a * RN_f64(1 / b)).
// The double product is within 2^-51 (relative) of a / b, and a / b cannot be that close to a rounding boundary of fp32:
// a boundary m has a 25-bit significand, a and m * b are multiples of 2^(e_m - 24), so a / b != m implies
// |a / b - m| >= 2^(e_m - 24) / b >= 2^-49 m; and a / b == m would need m * b (odd 25-bit x odd part of b) to fit the 24
// bits of a.  (A fp32 reciprocal instead moves exactly-integer coordinates across a floor() boundary and fails the
// golden parity tests: DESIGN.md section 9.)
__device__ __forceinline__ float div_by_const(float a, double rinv) { return (float)((double)a * rinv); }

// IEEE: the plain division sequence (K1: its chain is paced by fp32 issue and the fp64 multiply is the slower of the two
// there, 0.113 vs 0.116 ms; K6 / K7 gain 3 %).  Both forms give the same bits.
template <bool IEEE = false>
__device__ __forceinline__ float unnormalize(float v, int size)
{
    if (IEEE) return ((2.0f * v) / (float)(size - 1) - 1.0f + 1.0f) * ((float)(size - 1) / 2.0f);
    const double rinv = 1.0 / (double)(size - 1);            // kernel-invariant: hoisted out of the chain loops
    float nn = div_by_const(2.0f * v, rinv) - 1.0f;          // utils/iwe.py:30-31
    return (nn + 1.0f) * ((float)(size - 1) / 2.0f);         // ATen ComputeLocation<align_corners=true>
}

// fractions and the top-left cell of the lookup; y0 / x0 may lie outside the map
template <bool IEEE = false>
__device__ __forceinline__ Taps taps_core(float y, float x, int H, int W, int &y0, int &x0)
{
    Taps t;
    float iy = unnormalize<IEEE>(y, H), ix = unnormalize<IEEE>(x, W);
    float fy = floorf(iy), fx = floorf(ix);
    t.n = iy - fy;
    t.w = ix - fx;
    t.s = 1.0f - t.n;
    t.e = 1.0f - t.w;
    y0 = (int)fy;
    x0 = (int)fx;
    return t;
}

// The lookup's constants of one axis as wave-uniform (scalar-register) values: the exact reciprocal of size - 1 for
// div_by_const and (size - 1) / 2.  (Computed in the kernel they would sit in vector registers of every lane.)
struct AxisConst { double rinv; float half; };
__device__ __forceinline__ double uniform_f64(double v)
{
    const uint64_t u = (uint64_t)__double_as_longlong(v);
    const uint64_t r = (uint64_t)(uint32_t)__builtin_amdgcn_readfirstlane((int)(uint32_t)u) |
                       ((uint64_t)(uint32_t)__builtin_amdgcn_readfirstlane((int)(uint32_t)(u >> 32)) << 32);
    return __longlong_as_double((long long)r);
}
__device__ __forceinline__ float uniform_f32(float v) { return __int_as_float(__builtin_amdgcn_readfirstlane(__float_as_int(v))); }
__device__ __forceinline__ AxisConst axis_const(int size)
{
    AxisConst a;
    a.rinv = uniform_f64(1.0 / (double)(size - 1));
    a.half = uniform_f32((float)(size - 1) / 2.0f);
    return a;
}
// taps_core<false> with the constants handed in (same operations, same bits)
__device__ __forceinline__ Taps taps_core(float y, float x, const AxisConst &ah, const AxisConst &aw, int &y0, int &x0)
{
    Taps t;
    float iy = (div_by_const(2.0f * y, ah.rinv) - 1.0f + 1.0f) * ah.half;
    float ix = (div_by_const(2.0f * x, aw.rinv) - 1.0f + 1.0f) * aw.half;
    float fy = floorf(iy), fx = floorf(ix);
    t.n = iy - fy;
    t.w = ix - fx;
    t.s = 1.0f - t.n;
    t.e = 1.0f - t.w;
    y0 = (int)fy;
    x0 = (int)fx;
    return t;
}

template <bool IEEE = false>
__device__ __forceinline__ Taps make_taps(float y, float x, int H, int W)
{
    int y0, x0;
    Taps t = taps_core<IEEE>(y, x, H, W, y0, x0);
    int y1 = y0 + 1, x1 = x0 + 1;
    bool vy0 = (y0 >= 0) & (y0 < H), vy1 = (y1 >= 0) & (y1 < H);
    bool vx0 = (x0 >= 0) & (x0 < W), vx1 = (x1 >= 0) & (x1 < W);
    t.i00 = (vy0 && vx0) ? y0 * W + x0 : -1;
    t.i01 = (vy0 && vx1) ? y0 * W + x1 : -1;
    t.i10 = (vy1 && vx0) ? y1 * W + x0 : -1;
    t.i11 = (vy1 && vx1) ? y1 * W + x1 : -1;
    return t;
}

struct Quad2 { float2 v00, v01, v10, v11; };

// two horizontally adjacent float2 pixels in one 16-byte load (global_load_dwordx4 only needs 4-byte alignment)
typedef float f32x4_a8 __attribute__((ext_vector_type(4), aligned(8)));

// Branch-free: ONE unconditional 16-byte load whatever the validity of the two taps (an if/else between a vector path
// and scalar paths makes the compiler merge just-loaded registers, i.e. wait for memory after every row — the gathers of
// one chain step then queue up one behind the other instead of flying together).  i0 / i1: pixel indices of the left /
// right tap, -1 when outside; when both are valid they are adjacent (i1 == i0 + 1).  hw = pixels of the map (>= 2).
__device__ __forceinline__ void load_row(const float2 *__restrict__ map, int i0, int i1, int hw, float2 &v0, float2 &v1)
{
    const bool b0 = i0 >= 0, b1 = i1 >= 0;
    int base = b0 ? (b1 ? i0 : i0 - 1) : (b1 ? i1 : 0);      // keep both loaded pixels inside the row / the map
    base = min(max(base, 0), hw - 2);
    f32x4_a8 p = *reinterpret_cast<const f32x4_a8 *>(map + base);
    const float2 lo = make_float2(p.x, p.y), hi = make_float2(p.z, p.w), z = make_float2(0.0f, 0.0f);
    v0 = b0 ? (i0 != base ? hi : lo) : z;
    v1 = b1 ? (i1 != base ? hi : lo) : z;
}

__device__ __forceinline__ Quad2 load_quad(const float2 *__restrict__ map, const Taps &t, int hw)
{
    Quad2 q;
    load_row(map, t.i00, t.i01, hw, q.v00, q.v01);
    load_row(map, t.i10, t.i11, hw, q.v10, q.v11);
    return q;
}

// all four taps inside the map (0 <= y0 < H - 1, 0 <= x0 < W - 1): two plain 16-byte loads, no selects
__device__ __forceinline__ Quad2 load_quad_interior(const float2 *__restrict__ map, int i00, int W)
{
    f32x4_a8 p0 = *reinterpret_cast<const f32x4_a8 *>(map + i00);
    f32x4_a8 p1 = *reinterpret_cast<const f32x4_a8 *>(map + i00 + W);
    Quad2 q;
    q.v00 = make_float2(p0.x, p0.y); q.v01 = make_float2(p0.z, p0.w);
    q.v10 = make_float2(p1.x, p1.y); q.v11 = make_float2(p1.z, p1.w);
    return q;
}

// (flow_y, flow_x) at the tap position
__device__ __forceinline__ float2 quad_value(const Quad2 &q, const Taps &t)
{
    // ATen's vectorised CPU kernel is compiled with contraction: nw_val * nw + ne_val * ne + sw_val * sw + se_val * se is
    // ONE product and three fused multiply-adds, in that order (bit for bit against torch 2.10: 0 mismatches in 200 000
    // lookups; the unfused sum differs by an ulp in 35 % of them, and an ulp of position is 1 % of a 1e-5 hat weight)
    float w00 = t.s * t.e, w01 = t.s * t.w, w10 = t.n * t.e, w11 = t.n * t.w;
    return make_float2(__builtin_fmaf(q.v11.x, w11, __builtin_fmaf(q.v10.x, w10, __builtin_fmaf(q.v01.x, w01, q.v00.x * w00))),
                       __builtin_fmaf(q.v11.y, w11, __builtin_fmaf(q.v10.y, w10, __builtin_fmaf(q.v01.y, w01, q.v00.y * w00))));
}

// Jacobian of the lookup: jyy = d f_y/d y, jyx = d f_y/d x, jxy = d f_x/d y, jxx = d f_x/d x
__device__ __forceinline__ void quad_jacobian(const Quad2 &q, const Taps &t, float &jyy, float &jyx, float &jxy,
                                              float &jxx)
{
    jyx = (q.v01.x - q.v00.x) * t.s + (q.v11.x - q.v10.x) * t.n;
    jyy = (q.v10.x - q.v00.x) * t.e + (q.v11.x - q.v01.x) * t.w;
    jxx = (q.v01.y - q.v00.y) * t.s + (q.v11.y - q.v10.y) * t.n;
    jxy = (q.v10.y - q.v00.y) * t.e + (q.v11.y - q.v01.y) * t.w;
}

__device__ __forceinline__ bool inbounds(float y, float x, int H, int W)   // utils/iwe.py:52-57 (closed interval)
{
    return (y >= 0.0f) & (y <= (float)H - 1.0f) & (x >= 0.0f) & (x <= (float)W - 1.0f);
}

// flows_yx [P][F][B][H*W] float2
__device__ __forceinline__ const float2 *flow_map(const Win &w, const float2 *flows, int t, int i, int b)
{
    return flows + (((size_t)t * w.F + i) * w.B + b) * (size_t)(w.H * w.W);
}

// ---------------------------------------------------------------------------------------------
// Splat geometry: utils/iwe.py:63-113 get_interpolation, corners TL, TR, BL, BR.
// ---------------------------------------------------------------------------------------------
struct Splat {
    int iy[2], ix[2];
    float wy[2], wx[2];     // hat weights per axis (top/bottom, left/right)
    float sy[2], sx[2];     // their derivatives (autograd of max(0, 1 - |d|): ties split 0.5, abs'(0) = 0)
};

__device__ __forceinline__ float hat(float d, float &slope)
{
    float v = 1.0f - fabsf(d);
    float sg = (d > 0.0f) ? 1.0f : ((d < 0.0f) ? -1.0f : 0.0f);
    slope = (v > 0.0f) ? -sg : ((v == 0.0f) ? -0.5f * sg : 0.0f);
    return fmaxf(v, 0.0f);
}

__device__ __forceinline__ Splat make_splat(float y, float x)
{
    Splat s;
    float cy[2] = {floorf(y), floorf(y + 1.0f)};
    float cx[2] = {floorf(x), floorf(x + 1.0f)};
#pragma unroll
    for (int k = 0; k < 2; ++k) {
        s.iy[k] = (int)cy[k];
        s.ix[k] = (int)cx[k];
        s.wy[k] = hat(y - cy[k], s.sy[k]);
        s.wx[k] = hat(x - cx[k], s.sx[k]);
    }
    return s;
}

__device__ __forceinline__ uint32_t pack_meta(uint32_t bits, int kb, int kf, float mp, float mn)
{
    uint32_t m = bits | ((uint32_t)(kb + 1) << 8) | ((uint32_t)kf << 16);
    if (mp != 0.0f) m |= kMetaPos;
    if (mn != 0.0f) m |= kMetaNeg;
    if ((mp != 0.0f && mp != 1.0f) || (mn != 0.0f && mn != 1.0f)) m |= kMetaNonUnit;
    return m;
}

// Does the event enter the image of temporal scale s at reference time tref?  With border compensation (the reference's
// only reachable setting): bit s of the meta word, "inside the frame at every reference time of the scale's window"
// (loss/flow.py:671-681).  Without: inside the frame at THIS reference time, i.e. the chain has reached it alive
// (kb < tref < kf; :691-693 takes the event's own cumulative mask).
__device__ __forceinline__ bool in_image(const Win &w, uint32_t mv, int s, int tref)
{
    if (w.comp || w.kind != TEF_KIND_ITERATIVE) return (mv >> s) & 1u;      // (Linear without: every event of the window)
    const int kb1 = (int)((mv >> 8) & 0xffu), kf = (int)((mv >> 16) & 0xffu);      // kb + 1, kf
    return (mv & (kMetaPos | kMetaNeg)) != 0u && tref >= kb1 && tref < kf;
}

// ---------------------------------------------------------------------------------------------
// Row ranges.  Slots are handled in ROWS of 16 consecutive slots (= one DPP row of a wavefront; pass boundaries are
// multiples of 64 slots, so a wavefront, and with it a row, belongs to one pass).  For every trajectory plane K1 also records, per row, the
// interval [min y, max y] of the row's events that are still inside the frame there.  The scatter kernels split an
// image into row bands: a band workgroup looks at a row's interval (8 bytes per 16 events) and loads the events only
// if it can touch the band — events are sorted by (16x8 tile, pixel row), so a row of slots spans a pixel row or two and
// its interval is tight.
// An event that contributes at a plane is inside the frame there, hence inside its row's interval: nothing that counts
// is ever skipped, and an interval that was never written (all lanes of a wavefront dead) can only cause a useless look.
// ---------------------------------------------------------------------------------------------
// (positions inside the frame are >= 0: their bit patterns order like unsigned integers, so the reductions are integer
// min / max with the DPP operand folded in — no NaN canonicalisation, one instruction per step)
// max over the 16 lanes of a DPP row, delivered to every lane of the row (rotations inside the row)
__device__ __forceinline__ uint32_t row16_max(uint32_t v)
{
#define TEF_ROW_STEP(ctrl) v = max(v, (uint32_t)__builtin_amdgcn_update_dpp(0, (int)v, ctrl, 0xf, 0xf, true));
    TEF_ROW_STEP(0x121) TEF_ROW_STEP(0x122) TEF_ROW_STEP(0x124) TEF_ROW_STEP(0x128)      // row_ror:1, 2, 4, 8
#undef TEF_ROW_STEP
    return v;
}

// Workgroup barrier that orders LDS only.  __syncthreads() is a barrier plus a fence over ALL memory: on gfx9 loads and
// stores share one counter, so it waits (s_waitcnt vmcnt(0)) for every global store and every prefetch in flight — in
// the scatter kernels that was the (A, C + eps) write-out of an item, ~5 us per item, and the next item's first loads.
// Only for barriers that hand over LDS data; global data written before it is NOT visible to the other wavefronts after it.
__device__ __forceinline__ void lds_barrier()
{
    __builtin_amdgcn_fence(__ATOMIC_RELEASE, "workgroup", "local");
    __builtin_amdgcn_s_barrier();
    __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "workgroup", "local");
}

// Wave-wide sums through DPP (row shifts, then the row broadcasts of gfx9): no LDS round trips.  A ds_bpermute (what
// __shfl compiles to) queues behind the LDS atomics of every wavefront on the CU: in the scatter kernels six dependent ones
// took ~2.5 us.
// inclusive prefix sum over the 64 lanes
__device__ __forceinline__ int wave_prefix_sum(int v)
{
#define TEF_DPP_ADD(ctrl, rmask, bound) v += __builtin_amdgcn_update_dpp(0, v, ctrl, rmask, 0xf, bound);
    TEF_DPP_ADD(0x111, 0xf, true) TEF_DPP_ADD(0x112, 0xf, true) TEF_DPP_ADD(0x114, 0xf, true) TEF_DPP_ADD(0x118, 0xf, true)   // row_shr:1, 2, 4, 8
    TEF_DPP_ADD(0x142, 0xa, false)      // row_bcast:15 -> rows 1, 3
    TEF_DPP_ADD(0x143, 0xc, false)      // row_bcast:31 -> rows 2, 3
#undef TEF_DPP_ADD
    return v;
}
// sum over the 64 lanes, delivered to lane 63 (fixed order)
__device__ __forceinline__ double wave_sum_to_last(double v)
{
#define TEF_DPP_ADD(ctrl, rmask, bound)                                                                                  \
    {                                                                                                                    \
        const long long b_ = __double_as_longlong(v);                                                                    \
        const uint32_t lo_ = (uint32_t)__builtin_amdgcn_update_dpp(0, (int)(uint32_t)b_, ctrl, rmask, 0xf, bound);        \
        const uint32_t hi_ = (uint32_t)__builtin_amdgcn_update_dpp(0, (int)(uint32_t)(b_ >> 32), ctrl, rmask, 0xf, bound); \
        v += __longlong_as_double((long long)(((unsigned long long)hi_ << 32) | lo_));                                    \
    }
    TEF_DPP_ADD(0x111, 0xf, true) TEF_DPP_ADD(0x112, 0xf, true) TEF_DPP_ADD(0x114, 0xf, true) TEF_DPP_ADD(0x118, 0xf, true)
    TEF_DPP_ADD(0x142, 0xa, false)
    TEF_DPP_ADD(0x143, 0xc, false)
#undef TEF_DPP_ADD
    return v;
}

// rows of plane `plane` of (head, sample) ib: yr[(ib * (nplanes + 1) + plane) * nrow + row]; plane nplanes = original
// locations.  `in`: the lane's event is inside the frame at this plane (y >= 0).  The workgroup's 16 rows collect their
// intervals in LDS (`rng` [plane][16]) and flush_row_ranges writes every plane's 16 intervals as ONE 128-byte line.
// The store is unconditional and straight-line — all 16 lanes of a row hold the row's interval and store the same value,
// rows that do not keep this plane store to a dump plane: a store predicated on (lane == 15 && row keeps the plane)
// made the kernel 60 % slower (0.12 -> 0.19 ms), an unpredicated one costs 10 %.
// Wave-uniform early out when no row of the wavefront keeps this plane.
constexpr int kRangePlanes = TEF_MAX_PASSES + 3;       // planes 0..P, original locations, dump
constexpr int kRangeDump = kRangePlanes - 1;
__device__ __forceinline__ void store_row_range(float2 (*rng)[16], int plane, bool in, float y, bool row_writes)
{
    if (__builtin_amdgcn_ballot_w64(row_writes) == 0) return;
    const uint32_t bits = __float_as_uint(y);
    // min as the complement of a max of complements; an empty row stores [NaN, 0]: no comparison with it succeeds
    const uint32_t lo = ~row16_max(in ? ~bits : 0u), hi = row16_max(in ? bits : 0u);
    rng[row_writes ? plane : kRangeDump][threadIdx.x >> 4] = make_float2(__uint_as_float(lo), __uint_as_float(hi));
}

__device__ __forceinline__ void init_row_ranges(float2 (*rng)[16], int nplanes1)
{
    for (int k = threadIdx.x; k < nplanes1 * 16; k += blockDim.x) rng[k >> 4][k & 15] = make_float2(__uint_as_float(0xffffffffu), 0.0f);
    __syncthreads();
}

__device__ __forceinline__ void flush_row_ranges(const Win &w, float2 (*rng)[16], float2 *__restrict__ yr, int ib, int chunk)
{
    __syncthreads();
    const int row0 = chunk * ((int)blockDim.x >> 4);
    for (int k = threadIdx.x; k < (w.nplanes + 1) * 16; k += blockDim.x) {
        const int plane = k >> 4, r = k & 15;
        if (row0 + r < w.nrow) yr[((size_t)ib * (w.nplanes + 1) + plane) * w.nrow + row0 + r] = rng[plane][r];
    }
}

// Inputs the integer accumulators cannot represent (to_fixed needs |w * tau| < 32 and finite values): a timestamp or a
// location that is not finite, or a timestamp more than 16 passes away from its own pass (normalised lists hold
// ts in [t, t + 1]; |tau| <= 1 + (window + 16) / delta < 32 then).  The reference propagates such inputs to a NaN — or,
// for unnormalised timestamps, to a finite but meaningless — loss; here K1 records one word per workgroup and K4 turns
// the loss into NaN if any is set, so that bad inputs surface instead of becoming arbitrary integers.
__device__ __forceinline__ bool event_is_sane(float ts, float y, float x, int t)
{
    return fabsf(ts - (float)t) <= 16.0f && fabsf(y) < 3.0e38f && fabsf(x) < 3.0e38f;      // (false for NaN as well)
}
__device__ __forceinline__ void flag_bad_events(int *__restrict__ bad, int slot, bool is_bad)
{
    const int any = __syncthreads_or(is_bad ? 1 : 0);      // (every workgroup writes its word: nothing to clear)
    if (threadIdx.x == 0) bad[slot] = any;
}

// Streaming accesses — trajectory planes out of K1 and into K6, per-map vectors out of K6 — carry the non-temporal hint so
// that they do not push the gathered, re-used data (flow maps, (A, C + eps) images) out of the XCD's L2.  (Round 6: K1's
// plane stores too: K2 -2 us, K6 -2 us, the step 0.626 -> 0.619 ms.)
typedef float f32x2_v __attribute__((ext_vector_type(2)));
__device__ __forceinline__ float2 NT_LD2(const float2 *p)
{
    f32x2_v v = __builtin_nontemporal_load(reinterpret_cast<const f32x2_v *>(p));
    return make_float2(v.x, v.y);
}
__device__ __forceinline__ void NT_ST2(float2 *p, float a, float b)
{
    f32x2_v v = {a, b};
    __builtin_nontemporal_store(v, reinterpret_cast<f32x2_v *>(p));
}
// =============================================================================================
// K1 (Iterative): iterative warping of every event to every reference time.
// loss/flow.py:521-586 event_warping, :492-519 update_warping_indices, :599-654.
// traj plane k (k = 0..P) holds the event position at tref = k; meta packs the per-scale
// border-compensation bits (:671-681), kb (last out-of-bounds tref going backward, -1 if none),
// kf (first out-of-bounds tref going forward, P+1 if none) and the polarity flags.
// Every lane stays in the loops (dead and padding lanes masked): the per-row reductions need the whole row.
// =============================================================================================
__global__ __launch_bounds__(256) void iter_warp_kernel(Win w, const float2 *__restrict__ flows, Events g, Events d,
                                                        float2 *__restrict__ traj, uint2 *__restrict__ meta,
                                                        float2 *__restrict__ yr, int *__restrict__ queue,
                                                        int *__restrict__ bad, int chunks)
{
    if (blockIdx.x == 0 && threadIdx.x < kQueueInts) queue[threadIdx.x] = 0;      // work queues of the later kernels
    int ib, chunk;
    xcd_split(blockIdx.x, chunks, ib, chunk);
    if (ib >= w.F * w.B) return;
    __shared__ float2 rng[kRangePlanes][16];
    init_row_ranges(rng, w.nplanes + 1);
    const int u_raw = chunk * blockDim.x + threadIdx.x;
    const bool in_list = u_raw < w.Mt;
    const int u = in_list ? u_raw : w.Mt - 1;          // lanes past the end shadow the last slot, write nothing
    int i = ib / w.B, b = ib - i * w.B;
    bool isd = u >= w.M;
    int sl = isd ? u - w.M : u;
    const Events &E = isd ? d : g;
    size_t o = (size_t)b * E.cap + sl;
    float mp = E.mp[o], mn = E.mn[o];
    // collate padding (dataloader/base.py:414-421) and the alignment slots between passes contribute nothing
    const bool valid = in_list && (mp != 0.0f || mn != 0.0f);
    const int H = w.H, W = w.W, P = w.P;
    float ts = E.ts[o], y0 = E.y[o], x0 = E.x[o];
    const int t = __builtin_amdgcn_readfirstlane((int)E.bin[sl]);      // passes start at multiples of 64 slots: wave-uniform
    flag_bad_events(bad, ib * chunks + chunk, valid && !event_is_sane(ts, y0, x0, t));
    float2 *tr = traj + (size_t)ib * (w.nplanes + 1) * w.Mt + u;

    // Bilinear flow lookup of map k at (y, x).  The kernel is VALU-bound (~135 vector instructions per chain step): when
    // every lane of the wavefront has its four taps inside the map, two plain 16-byte loads replace the clamped,
    // select-masked ones (same values, same arithmetic).
    auto lookup = [&](float y, float x, int k) -> float2 {
        const float2 *map = flow_map(w, flows, k, i, b);
        int yi, xi;
        Taps c = taps_core<true>(y, x, H, W, yi, xi);
        const bool inside = (yi >= 0) & (yi < H - 1) & (xi >= 0) & (xi < W - 1);
        if (__builtin_amdgcn_ballot_w64(!inside) == 0) return quad_value(load_quad_interior(map, yi * W + xi, W), c);
        Taps q = make_taps<true>(y, x, H, W);
        return quad_value(load_quad(map, q, H * W), q);
    };
    store_row_range(rng, w.nplanes, valid, y0, in_list);
    if (in_list) NT_ST2(&tr[(size_t)w.nplanes * w.Mt], y0, x0);      // plane nplanes: where pass t sampled its own map (K7)
    // flow at the original location, shared by the first forward and the first backward step
    float2 f0 = make_float2(0.0f, 0.0f);
    if (valid) f0 = lookup(y0, x0, t);

    // an event of pass t is only ever looked at (IWEs, gradient sweep, flow-gradient splat) at reference times within
    // delta_passes[0] of t; the chain still runs to both ends of the window because the border mask needs it
    const int reach = P / w.mode_div;
    int kf = P + 1, kb = -1;
    {   // forward: maps t .. P-1, positions at tref = t+1 .. P
        float y = y0, x = x0;
        float2 f = f0;
        float dt = (float)(t + 1) - ts;             // utils/iwe.py:14 (tref - ts)
        bool alive = valid;
        for (int k = t;; ++k) {
            alive = alive && k < P;
            if (__builtin_amdgcn_ballot_w64(alive) == 0) break;
            const bool kept = k < P && k + 1 <= t + reach;          // planes beyond the reach are never read
            if (alive) {
                if (k > t) {
                    f = lookup(y, x, k);
                    dt = 1.0f;
                }
                y = y + dt * f.x;
                x = x + dt * f.y;
                if (kept) NT_ST2(&tr[(size_t)(k + 1) * w.Mt], y, x);
                if (!inbounds(y, x, H, W)) { kf = k + 1; alive = false; }       // cumulative purge, loss/flow.py:575
            }
            store_row_range(rng, min(k + 1, P), alive, y, kept && in_list);
        }
    }
    {   // backward: maps t .. 0, positions at tref = t .. 0
        float y = y0, x = x0;
        float2 f = f0;
        float dt = (float)t - ts;
        bool alive = valid;
        for (int k = t;; --k) {
            alive = alive && k >= 0;
            if (__builtin_amdgcn_ballot_w64(alive) == 0) break;
            const bool kept = k >= 0 && k >= t - reach;
            if (alive) {
                if (k < t) {
                    f = lookup(y, x, k);
                    dt = -1.0f;
                }
                y = y + dt * f.x;
                x = x + dt * f.y;
                if (kept) NT_ST2(&tr[(size_t)k * w.Mt], y, x);
                if (!inbounds(y, x, H, W)) { kb = k; alive = false; }
            }
            store_row_range(rng, max(k, 0), alive, y, kept && in_list);
        }
    }
    flush_row_ranges(w, rng, yr, ib, chunk);
    if (!in_list) return;
    uint32_t bits = 0;
    for (int s = 0; s < w.S; ++s) {
        int scale = P >> s, wi = t / scale;
        if (wi >= (1 << s)) continue;
        int lo = wi * scale, hi = lo + scale;
        if (kb < lo && kf > hi) bits |= 1u << s;
    }
    meta[(size_t)ib * w.Mt + u] = make_uint2(valid ? pack_meta(bits, kb, kf, mp, mn) : 0u, __float_as_uint(ts));
}

// =============================================================================================
// K1 (Linear): one flow sample per event, linear warp to both ends of each window.
// loss/flow.py:268-283 (sample), :337-343 (warp + shared purge).  Plane 2s = forward (tref = hi),
// plane 2s+1 = backward (tref = lo).
// =============================================================================================
__global__ __launch_bounds__(256) void linear_warp_kernel(Win w, const float2 *__restrict__ flows, Events g, Events d,
                                                          float2 *__restrict__ traj, uint2 *__restrict__ meta,
                                                          float2 *__restrict__ yr, int *__restrict__ queue,
                                                          int *__restrict__ bad, int chunks)
{
    if (blockIdx.x == 0 && threadIdx.x < kQueueInts) queue[threadIdx.x] = 0;
    int ib, chunk;
    xcd_split(blockIdx.x, chunks, ib, chunk);
    if (ib >= w.F * w.B) return;
    __shared__ float2 rng[kRangePlanes][16];
    init_row_ranges(rng, w.nplanes + 1);
    const int u_raw = chunk * blockDim.x + threadIdx.x;
    const bool in_list = u_raw < w.Mt;
    const int u = in_list ? u_raw : w.Mt - 1;
    int i = ib / w.B, b = ib - i * w.B;
    bool isd = u >= w.M;
    int sl = isd ? u - w.M : u;
    const Events &E = isd ? d : g;
    size_t o = (size_t)b * E.cap + sl;
    float mp = E.mp[o], mn = E.mn[o];
    const bool valid = in_list && (mp != 0.0f || mn != 0.0f);
    const int H = w.H, W = w.W;
    float ts = E.ts[o], y0 = E.y[o], x0 = E.x[o];
    const int t = E.bin[sl];
    flag_bad_events(bad, ib * chunks + chunk, valid && !event_is_sane(ts, y0, x0, t));
    store_row_range(rng, w.nplanes, valid, y0, in_list);
    float2 f = make_float2(0.0f, 0.0f);
    if (valid) {
        Taps tp = make_taps(y0, x0, H, W);
        f = quad_value(load_quad(flow_map(w, flows, t, i, b), tp, w.H * w.W), tp);
    }
    float2 *tr = traj + (size_t)ib * (w.nplanes + 1) * w.Mt + u;
    if (in_list) tr[(size_t)w.nplanes * w.Mt] = make_float2(y0, x0);      // plane nplanes: the sampling locations (K7)
    uint32_t bits = 0;
    for (int s = 0; s < w.S; ++s) {
        int scale = w.P >> s, wi = t / scale;
        const bool has = wi < (1 << s);                 // uniform over the row
        int lo = wi * scale, hi = lo + scale;
        float dtf = (float)hi - ts, dtb = (float)lo - ts;
        float yf = y0 + dtf * f.x, xf = x0 + dtf * f.y;
        float yb = y0 + dtb * f.x, xb = x0 + dtb * f.y;
        // shared purge of both window ends (loss/flow.py:341-343); without border compensation nothing is purged and
        // the corners outside the frame drop out one by one (:324-328, utils/iwe.py:103-107)
        const bool in = valid && has && (!w.comp || (inbounds(yf, xf, H, W) && inbounds(yb, xb, H, W)));
        if (valid && has) {
            tr[(size_t)(2 * s) * w.Mt] = make_float2(yf, xf);
            tr[(size_t)(2 * s + 1) * w.Mt] = make_float2(yb, xb);
        }
        if (in) bits |= 1u << s;
        // (the interval's integer min / max need non-negative values: without border compensation a position above the
        // frame counts as row 0, which keeps the band test a superset)
        store_row_range(rng, 2 * s, in, fmaxf(yf, 0.0f), has && in_list);
        store_row_range(rng, 2 * s + 1, in, fmaxf(yb, 0.0f), has && in_list);
    }
    flush_row_ranges(w, rng, yr, ib, chunk);
    if (in_list) meta[(size_t)ib * w.Mt + u] = make_uint2(valid ? pack_meta(bits, -1, 0, mp, mn) : 0u, __float_as_uint(ts));
}

// LDS image plane helpers of the two scatter kernels: [rows][W + kRowPad] 8-byte accumulators.
// zero fill with 16-byte stores; write-out [rows][W] floats without a per-element division: when the workgroup covers
// whole rows (blockDim % (W/2) == 0) a thread keeps its column pair and walks down the rows with 16-byte LDS reads.
__device__ __forceinline__ void lds_plane_zero(double *img, int n)
{
    double2 *v = reinterpret_cast<double2 *>(img);
    for (int p = threadIdx.x; p < (n >> 1); p += blockDim.x) v[p] = make_double2(0.0, 0.0);
    if ((n & 1) && threadIdx.x == 0) img[n - 1] = 0.0;
}

// Fixed-point accumulators of the IWE scatter (K2).  Measured on this chip (tools/lds_atomic_patterns.hip, cycles per
// wave-instruction per CU): ds_add_f64 9.6 conflict-free, 17 / 33 at 2- / 4-way bank conflicts, 31 on the footprint of a
// sorted wavefront; ds_add_u64 7.8 / 9.4 / 17 and 18 on that footprint (ds_add_f32: 194 whatever the pattern).  The
// integer atomic is twice as fast where it matters, and integer sums are exact: the images no longer depend on the
// arrival order, so the loss is bitwise reproducible from run to run.
// Format: two's-complement Q17.46.  A contribution v = w [* tau] (fp32, |v| < 32) becomes RN(v * 2^46) in three
// instructions: 96 + v has ulp 2^-46 over (64, 128), so the difference of the bit patterns of (96 + v) and 96 IS that
// integer.  Exact whenever v's lowest mantissa bit is >= 2^-46 (every bilinear weight >= 2^-23), otherwise rounded to
// 2^-47 absolute.  n contributions of at most 1 cannot overflow while n < 2^17: the workgroup knows its event count up
// front and takes the fp64 path beyond that, and when its runs hold general (non-0/1) mask values.
constexpr double kFxMagic = 96.0;
constexpr int kFxMaxEvents = 1 << 17;
__device__ __forceinline__ unsigned long long to_fixed(float v)
{
    return (unsigned long long)(__double_as_longlong((double)v + kFxMagic) - __double_as_longlong(kFxMagic));
}
template <bool FX>
__device__ __forceinline__ float acc_value(double a)
{
    return FX ? (float)((double)__double_as_longlong(a) * 0x1p-46) : (float)a;
}

// =============================================================================================
// K2: images of warped events + their focus-loss statistics.
//   loss/flow.py:81-110 iwe_formatting = utils/iwe.py:63-136 get_interpolation + 4x interpolate (scatter_add_), then
//   :725-727 A = T / (C + 1e-9) and :112-129 focus_loss.
// A workgroup owns a ROW BAND of one image with all four planes — event count C and weighted timestamp sum T of both
// polarities — in LDS (32 rows at 128x128 = 136 KiB).  Per 16-slot row of events it reads K1's [min y, max y] interval
// (8 bytes) and loads the row's events only if the interval can touch the band; an event is split into corner weights
// once for its eight accumulations.  When all events are in, the band's pixels are turned into (A, R = 1/(C + eps)) for
// the backward and into the partial sums of the focus loss: the images themselves never go to memory.
// Accumulators: Q17.46 integers (ds_add_u64) or fp64 (general masks / very long runs).
// Workgroups are persistent (one per CU, the planes fill its LDS): they pull (image, head, sample, band) items from one
// queue per XCD — largest images first, the bands of an image on one XCD so that its trajectory plane is fetched from
// HBM once — because launching a 1024-thread / 136 KiB workgroup costs ~3.6 us and the 1408 items of the BASELINE window
// paid that 5.5 times per CU.
//   ar   [(j*FB + ib)*2 + c][H*W] float2 = (A, C + eps)
//   part [((j*FB + ib)*nbands + band)*2] = sum_px (A_pos^2 + A_neg^2), [+1] = #{C_pos + C_neg != 0} of the band
// =============================================================================================
template <bool FX>
__device__ __forceinline__ void acc_add(double *cell, float v)
{
    if (FX) atomicAdd(reinterpret_cast<unsigned long long *>(cell), to_fixed(v));
    else atomicAdd(cell, (double)v);
}

// p = (y, x) inside the frame (shared border mask), so 0 <= floor(y) <= H - 1 and 0 <= floor(x) <= W - 1; the right
// column may be W (lands in the row padding, weight 0) and the bottom row H (rejected by the band test).  The far
// corners are addressed as +1: floor(v + 1) differs from floor(v) + 1 only where fp32 rounds v + 1 up to an integer,
// and then its weight max(0, 1 - |v - floor(v + 1)|) is 0.
template <bool FX>
__device__ __forceinline__ void splat_one(float2 p, float ts, float m, const Img &im, double rdelta, double *img_c,
                                          double *img_t, int r0, int nrows, int WP)
{
    const float y = p.x, x = p.y;                    // traj stores (y, x) in (.x, .y)
    const float fy0 = floorf(y);
    const int rr = (int)fy0 - r0;
    float wy0 = fmaxf(1.0f - fabsf(y - fy0), 0.0f), wy1 = fmaxf(1.0f - fabsf(y - floorf(y + 1.0f)), 0.0f);   // utils/iwe.py:97-107
    const bool ok0 = (rr >= 0) & (rr < nrows), ok1 = (rr + 1 >= 0) & (rr + 1 < nrows) & (wy1 != 0.0f);
    if (!(ok0 | ok1)) return;
    const float fx0 = floorf(x);
    float wx0 = fmaxf(1.0f - fabsf(x - fx0), 0.0f), wx1 = fmaxf(1.0f - fabsf(x - floorf(x + 1.0f)), 0.0f);
    // tau = 1 - |tref - ts| / delta (:94-95); delta is an integer number of passes: exact division by a constant
    const float tau = 1.0f - div_by_const(fabsf(im.tref - ts), rdelta);
    const int cell = __mul24(rr, WP) + (int)fx0;      // (24-bit multiply: full rate)
    if (ok0) {
        float w00 = wy0 * wx0, w01 = wy0 * wx1;
        acc_add<FX>(img_c + cell, FX ? w00 : w00 * m);
        acc_add<FX>(img_c + cell + 1, FX ? w01 : w01 * m);
        acc_add<FX>(img_t + cell, FX ? w00 * tau : (w00 * tau) * m);
        acc_add<FX>(img_t + cell + 1, FX ? w01 * tau : (w01 * tau) * m);
    }
    if (ok1) {
        float w10 = wy1 * wx0, w11 = wy1 * wx1;
        acc_add<FX>(img_c + cell + WP, FX ? w10 : w10 * m);
        acc_add<FX>(img_c + cell + WP + 1, FX ? w11 : w11 * m);
        acc_add<FX>(img_t + cell + WP, FX ? w10 * tau : (w10 * tau) * m);
        acc_add<FX>(img_t + cell + WP + 1, FX ? w11 * tau : (w11 * tau) * m);
    }
}

// The same for a position ANYWHERE (Linear without border compensation: nothing is purged, every corner is tested on its
// own like utils/iwe.py:103-107 does).  Rows outside the band are outside the frame or another workgroup's.
template <bool FX>
__device__ __noinline__ void splat_any(float2 p, float ts, float m, float tref, double rdelta, double *img_c, double *img_t,
                                       int r0, int nrows, int W, int WP)
{
    const float y = p.x, x = p.y;
    const float fy0 = floorf(y), fx0 = floorf(x);
    // (also rejects NaN / infinite positions before anything is converted to an integer)
    if (!(fy0 + 1.0f >= (float)r0 && fy0 < (float)(r0 + nrows) && fx0 + 1.0f >= 0.0f && fx0 < (float)W)) return;
    const float tau = 1.0f - div_by_const(fabsf(tref - ts), rdelta);
#pragma nounroll
    for (int k = 0; k < 4; ++k) {                       // corners TL, TR, BL, BR (utils/iwe.py:85-94)
        const float cy = (k & 2) ? floorf(y + 1.0f) : fy0, cx = (k & 1) ? floorf(x + 1.0f) : fx0;
        const int rr = (int)cy - r0, ix = (int)cx;
        if (rr < 0 || rr >= nrows || ix < 0 || ix >= W) continue;
        const float wgt = fmaxf(1.0f - fabsf(y - cy), 0.0f) * fmaxf(1.0f - fabsf(x - cx), 0.0f);
        acc_add<FX>(img_c + rr * WP + ix, FX ? wgt : wgt * m);
        acc_add<FX>(img_t + rr * WP + ix, FX ? wgt * tau : (wgt * tau) * m);
    }
}

// general path (masks other than 0 / 1, or more events than the integers hold): one contiguous run of slots of
// polarity c in unified slot space, fp64 accumulators, no interval tests
__device__ __forceinline__ void splat_run_general(const Win &w, const Img &im, double rdelta, const Events &g,
                                                  const Events &d, int b, int c, int u0, int len,
                                                  const float2 *__restrict__ pl, const uint2 *__restrict__ mt,
                                                  double *img_c, double *img_t, int r0, int nrows)
{
    const bool isd = u0 >= w.M;
    const Events &E = isd ? d : g;
    const float *mask = (c ? E.mn : E.mp) + (size_t)b * E.cap - (isd ? w.M : 0);      // indexed by unified slot
    const int WP = w.W + kRowPad;
    for (int v = threadIdx.x; v < len; v += blockDim.x) {
        const int u = u0 + v;
        const uint2 m2 = mt[u];
        const uint32_t mv = m2.x;
        if (!in_image(w, mv, im.s, im.plane)) continue;          // border mask (:671-681)
        float m = 1.0f;
        if (mv & kMetaNonUnit) m = mask[u];
        const float ts = __uint_as_float(m2.y);
        if (w.comp || w.kind == TEF_KIND_ITERATIVE) splat_one<false>(pl[u], ts, m, im, rdelta, img_c, img_t, r0, nrows, WP);
        else splat_any<false>(pl[u], ts, m, im.tref, rdelta, img_c, img_t, r0, nrows, w.W, WP);
    }
}


constexpr int kSplat2Threads = 512;             // K2: two such workgroups per CU
constexpr int kSplat2Waves = kSplat2Threads / 64;
constexpr size_t kSplat2LdsBudget = 74 * 1024;  // planes of one K2 workgroup incl. their two halo rows (beside ~7 KiB of run tables and row lists)
constexpr int kRing = 128;                      // hit-row FIFO entries per wavefront (power of two)

// "C != 0" of a band's pixels travels from K2 to the pixel count as ONE BIT per pixel: 64-bit ballots of the wavefronts that
// walk the band in band_stats (which bit is which pixel only depends on the band's shape, so the words of the two
// polarities line up and their OR counts the pixels of loss/flow.py:125).  Words a band of `nrows` rows writes:
__host__ __device__ inline bool stats_by_column_pairs(int W, int WP, int threads)
{
    const int half = W >> 1;
    return !(W & 1) && !(WP & 1) && half > 0 && threads % half == 0;
}
__host__ __device__ inline int nz_words(int nrows, int W, int WP, int threads)
{
    const int nwaves = threads >> 6;
    if (stats_by_column_pairs(W, WP, threads)) {
        const int rstep = threads / (W >> 1);
        return 2 * nwaves * ((nrows + rstep - 1) / rstep);
    }
    return nwaves * ((nrows * W + threads - 1) / threads);
}

// band pixels of one polarity -> (A, C + eps), the polarity's share of the focus-loss sum, and one bit per pixel "C != 0"
// (the count of active pixels needs both polarities: image_count_kernel).  planes: first band row of C; T is plane_sz further.
// CLEAR: the thread that has read a pixel's accumulators clears them, and the halo rows are cleared as well: the planes are
// all zero again for the workgroup's next item (all-zero bits: 0.0 and integer 0 alike) without a pass of their own.
template <bool FX, bool CLEAR = true>
__device__ __forceinline__ void band_stats(double *planes, size_t plane_sz, int nrows, int W, int WP,
                                           float2 *__restrict__ ar, unsigned long long *__restrict__ nzw, float &acc)
{
    double *cp = planes, *tp = planes + plane_sz;
    auto pixel = [&](float c, float t, float2 &o) {
        float a = t / (c + kEps);                     // :727
        o = make_float2(a, c + kEps);                 // (A, C + eps): what the division's backward needs (pixel_grads, K6)
        acc += a * a;                                 // :122-123
        return c != 0.0f;                             // :125 (masks are non-negative: C_pos + C_neg != 0 <=> either is)
    };
    const int lane = threadIdx.x & 63, wid = threadIdx.x >> 6, nwaves = blockDim.x >> 6;
    if (CLEAR) {
        for (int p = threadIdx.x; p < 2 * WP; p += blockDim.x) {      // halo rows: above the band, below it
            const int i = p < WP ? p - WP : nrows * WP + (p - WP);
            cp[i] = 0.0;
            tp[i] = 0.0;
        }
    }
    if (stats_by_column_pairs(W, WP, (int)blockDim.x)) {
        const int half = W >> 1, rstep = blockDim.x / half;
        const int rt = threadIdx.x / half, c = (threadIdx.x - rt * half) * 2;
        const int iters = (nrows + rstep - 1) / rstep;
        for (int k = 0; k < iters; ++k) {             // (wave-uniform trip count: the ballots see every lane)
            const int r = rt + k * rstep;
            bool z0 = false, z1 = false;
            if (r < nrows) {
                const int i = r * WP + c;
                double2 dc = *reinterpret_cast<const double2 *>(cp + i), dt = *reinterpret_cast<const double2 *>(tp + i);
                if (CLEAR) {
                    *reinterpret_cast<double2 *>(cp + i) = make_double2(0.0, 0.0);
                    *reinterpret_cast<double2 *>(tp + i) = make_double2(0.0, 0.0);
                }
                float2 p0, p1;
                z0 = pixel(acc_value<FX>(dc.x), acc_value<FX>(dt.x), p0);
                z1 = pixel(acc_value<FX>(dc.y), acc_value<FX>(dt.y), p1);
                *reinterpret_cast<float4 *>(ar + (size_t)r * W + c) = make_float4(p0.x, p0.y, p1.x, p1.y);
            }
            const unsigned long long b0 = __builtin_amdgcn_ballot_w64(z0), b1 = __builtin_amdgcn_ballot_w64(z1);
            if (lane == 0) *reinterpret_cast<ulonglong2 *>(nzw + 2 * (k * nwaves + wid)) = make_ulonglong2(b0, b1);
        }
    } else {
        const int n = nrows * W, iters = (n + (int)blockDim.x - 1) / (int)blockDim.x;
        for (int k = 0; k < iters; ++k) {
            const int q = k * blockDim.x + threadIdx.x;
            bool z = false;
            if (q < n) {
                const int r = q / W, i = r * WP + (q - r * W);
                float2 p0;
                z = pixel(acc_value<FX>(cp[i]), acc_value<FX>(tp[i]), p0);
                if (CLEAR) {
                    cp[i] = 0.0;
                    tp[i] = 0.0;
                }
                ar[q] = p0;
            }
            const unsigned long long b0 = __builtin_amdgcn_ballot_w64(z);
            if (lane == 0) nzw[k * nwaves + wid] = b0;
        }
    }
}

// Per image (j, head, sample): the number of pixels that hold events of either polarity (loss/flow.py:125-127) from K2's
// per-polarity bits, and the image's sum of squares from the wavefronts' shares (fixed order) -> (sum, count) for K4
// (+ this block's slice of K1's "unrepresentable input" words, folded into one word per image for K4: a single block
// scanning all of them there took 18 us).   nzw: [(image, head, sample)][polarity][band][cap] words
__global__ __launch_bounds__(256) void image_count_kernel(const unsigned long long *__restrict__ nzw, int cap, int nbands,
                                                          int rows_per_band, int H, int W, const double *__restrict__ part,
                                                          double *__restrict__ cnt, double *__restrict__ sq,
                                                          const int *__restrict__ bad, int nbad, int *__restrict__ bad_img)
{
    __shared__ int red[4];
    __shared__ double dred[4];
    {
        const int per = (nbad + (int)gridDim.x - 1) / (int)gridDim.x, lo = (int)blockIdx.x * per, hi = min(lo + per, nbad);
        int any = 0;
        for (int k = lo + (int)threadIdx.x; k < hi; k += blockDim.x) any |= bad[k];
        any = __syncthreads_or(any);
        if (threadIdx.x == 0) bad_img[blockIdx.x] = any;
    }
    const unsigned long long *p = nzw + (size_t)blockIdx.x * 2 * nbands * cap, *n = p + (size_t)nbands * cap;
    int c = 0;
    for (int band = 0; band < nbands; ++band) {
        const int r0 = band * rows_per_band, nrows = min(H, r0 + rows_per_band) - r0;
        const int used = nz_words(nrows, W, W + kRowPad, kSplat2Threads);
        for (int k = threadIdx.x; k < used; k += blockDim.x) c += __popcll(p[band * cap + k] | n[band * cap + k]);
    }
    // shares: [polarity][band][wavefront] of this image, each thread a strided slice, then a fixed tree
    const int nsh = 2 * nbands * kSplat2Waves;
    double s2 = 0.0;
    for (int k = threadIdx.x; k < nsh; k += blockDim.x) s2 += part[(size_t)blockIdx.x * nsh + k];
    for (int sft = 32; sft > 0; sft >>= 1) {
        c += __shfl_down(c, sft, 64);
        s2 += __shfl_down(s2, sft, 64);
    }
    if ((threadIdx.x & 63) == 0) { red[threadIdx.x >> 6] = c; dred[threadIdx.x >> 6] = s2; }
    __syncthreads();
    if (threadIdx.x == 0) {
        cnt[blockIdx.x] = (double)((red[0] + red[1]) + (red[2] + red[3]));
        sq[blockIdx.x] = (dred[0] + dred[1]) + (dred[2] + dred[3]);
    }
}

typedef float f32x2_e __attribute__((ext_vector_type(2)));      // (y, x) pairs: the compiler picks v_pk_{add,mul}_f32 for these

// One event into the integer planes of a band (the hot path of the whole forward: instruction count is what bounds it).
//   p = (y, x) inside the frame (shared border mask), so 0 <= floor(y) <= H - 1 and 0 <= floor(x) <= W - 1.  utils/iwe.py:85-107:
//   near weights max(0, 1 - |v - floor(v)|) = 1 - (v - floor(v)) exactly (0 <= v - floor(v) < 1: neither the abs nor the max
//   can act); far weights max(0, 1 - |v - floor(v + 1)|) = max(0, 1 + (v - floor(v + 1))) (the difference is negative; the
//   max stays: where fp32 rounds v + 1 up to the next integer the sum is a negative ulp, and the reference clamps it to 0).
//   The far corners are addressed as +1: floor(v + 1) differs from floor(v) + 1 only in that case, with weight 0.
// The planes carry ONE HALO ROW above and below the band, so the eight accumulations are unconditional: an event of the
// row above the band leaves its near row in the halo (never read), one of the band's last row its far row; the right
// column may be W (row padding).  A weight of exactly 0 adds 0.
//   cpl: byte address of the C plane's halo row 0 in LDS; tpl_off / row_off: bytes from C to T / from a row to the next;
//   rr = floor(y) - (r0 - 1) in [0, nrows]
__device__ __forceinline__ void splat_fixed(float y, float x, float tau, int rr, int WP, unsigned long long *cpl,
                                            unsigned long long *tpl)
{
    const f32x2_e p = {y, x};
    const f32x2_e f0 = {floorf(y), floorf(x)};
    const f32x2_e p1 = p + 1.0f;
    const f32x2_e f1 = {floorf(p1.x), floorf(p1.y)};
    const f32x2_e nearw = 1.0f - (p - f0);               // (wy0, wx0)
    f32x2_e farw = 1.0f + (p - f1);                      // (wy1, wx1) before the clamp
    farw.x = fmaxf(farw.x, 0.0f);
    farw.y = fmaxf(farw.y, 0.0f);
    const f32x2_e wx = {nearw.y, farw.y};
    const f32x2_e w0 = nearw.x * wx, w1 = farw.x * wx;   // (w00, w01), (w10, w11)
    const f32x2_e t0 = w0 * tau, t1 = w1 * tau;          // utils/iwe.py:94-95 weights * tau, then the polarity mask (1)
    const int cell = __mul24(rr, WP) + (int)f0.y;        // (24-bit multiply: full rate)
    unsigned long long *c0 = cpl + cell, *q0 = tpl + cell;
    atomicAdd(c0, to_fixed(w0.x));
    atomicAdd(c0 + 1, to_fixed(w0.y));
    atomicAdd(c0 + WP, to_fixed(w1.x));
    atomicAdd(c0 + WP + 1, to_fixed(w1.y));
    atomicAdd(q0, to_fixed(t0.x));
    atomicAdd(q0 + 1, to_fixed(t0.y));
    atomicAdd(q0 + WP, to_fixed(t1.x));
    atomicAdd(q0 + WP + 1, to_fixed(t1.y));
}

// =============================================================================================
// K2: images of warped events + their focus-loss statistics.
//   loss/flow.py:81-110 iwe_formatting = utils/iwe.py:63-136 get_interpolation + 4x interpolate (scatter_add_), then
//   :725-727 A = T / (C + 1e-9) and :112-129 focus_loss.
// A workgroup owns a ROW BAND of one polarity of one image with its two planes — event count C and weighted timestamp
// sum T — in LDS (32 rows + 2 halo rows at 128x128 = 72 KiB; two workgroups per CU).  Per 16-slot row of events it reads
// K1's [min y, max y] interval (8 bytes) and loads the row's events only if the interval can touch the band; an event is
// split into corner weights once for its eight accumulations.  When all events are in, the band's pixels are turned into
// (A, R = 1/(C + eps)) for the backward and into the partial sums of the focus loss: the images themselves never go to
// memory.  Accumulators: Q17.46 integers (ds_add_u64) or fp64 (general masks / very long runs).
// Workgroups are persistent: they pull (image, head, sample, polarity, band) items from one queue per XCD — largest images
// first, the bands of an image on one XCD so that its trajectory plane is fetched from HBM once.
//   ar   [(j*FB + ib)*2 + c][H*W] float2 = (A, C + eps)
//   part [(((j*FB + ib)*2 + c)*nbands + band)*8 + wavefront] = the wavefront's share of sum_px A_c^2 of the band
// Instruction diet of round 3 (the sweep was VALU-bound at ~155 vector instructions per 64 event slots, a third of them
// address arithmetic and lane-range bookkeeping of the event loads):
//   * the item index goes through readfirstlane: everything derived from it (image, planes, base pointers) is scalar;
//   * an event visit is TWO 8-byte loads at one 32-bit offset from scalar bases: the position, and K1's (flags, timestamp)
//     word pair;
//   * whole rows are loaded; which slots of a row belong to this polarity is decided by the event's own flag bits (the
//     integer path runs only when the passes hold no general-mask events, so a slot is pos-only, neg-only or padding):
//     a list entry is just the row;
//   * halo rows instead of per-corner band tests; near weights without abs / max; (y, x) as packed pairs.
// =============================================================================================
// MODE 0: border compensation (the reference's reachable setting): the border bit of the image's temporal scale in the meta
// word.  MODE 1: Iterative without: alive at this reference time (kb < tref < kf).  Two instantiations (the test as a
// run-time branch cost the default path 4 %).  Linear without compensation splats positions
// outside the frame and takes the general path.
// (second argument: at least 4 wavefronts per SIMD, i.e. at most 128 VGPRs — two of these workgroups per CU)
template <int MODE>
__global__ __launch_bounds__(kSplat2Threads, 4) void splat_stats_kernel(Win w, Events g, Events d,
                                                                     const float2 *__restrict__ traj,
                                                                     const uint2 *__restrict__ meta,
                                                                     const float2 *__restrict__ yr,
                                                                     float2 *__restrict__ ar,
                                                                     unsigned long long *__restrict__ nzw, int nz_cap,
                                                                     double *__restrict__ part, int rows_per_band,
                                                                     int nbands, int *__restrict__ queue)
{
    extern __shared__ double lds_img[];
    constexpr int kRuns = 2 * TEF_MAX_PASSES;      // (grad, detached) per pass, one polarity
    __shared__ int run_u0[kRuns], run_len[kRuns], run_cum[kRuns + 1], s_item, s_flags[2];
    __shared__ int hit_ring[(kSplat2Threads / 64) * kRing];      // rows that can touch the band: one FIFO per wavefront
    const int FB = w.F * w.B, H = w.H, W = w.W, WP = W + kRowPad, HW = H * W;
    const int xcd = blockIdx.x & 7;
    const int nitems = w.nimg * FB, nsub = 2 * nbands;
    for (int k = threadIdx.x; k < (kSplat2Threads / 64) * kRing; k += blockDim.x) hit_ring[k] = 0;      // (stale entries are read, as rows)
    if (threadIdx.x == 0) s_item = atomicAdd(&queue[xcd], 1);
    lds_plane_zero(lds_img, 2 * (min(H, rows_per_band) + 2) * WP);
    __syncthreads();
    for (;;) {
        const int q = __builtin_amdgcn_readfirstlane(s_item);      // wave-uniform: everything derived from it is scalar
        const int it = xcd + 8 * (q / nsub), sub = q - (q / nsub) * nsub;
        if (it >= nitems) break;
        const int c = sub & 1, band = sub >> 1;          // polarity, row band
        const int j = w.order[it / FB], ib = it % FB, b = ib % w.B;
        const int r0 = band * rows_per_band, nrows = min(H, r0 + rows_per_band) - r0;
        const Img im = decode_image(w, j);
        const double rdelta = 1.0 / (double)im.delta;
        const float band_lo = (float)(r0 - 1), band_hi = (float)(r0 + nrows);     // rows floor(y), floor(y) + 1 of an event
        // planes: [C | T], each nrows + 2 rows (halo row, the band, halo row) of WP accumulators
        const size_t plane_sz = (size_t)(nrows + 2) * WP;
        double *img_c = lds_img, *img_t = lds_img + plane_sz;
        // (the planes are all zero here — cleared once before the first item, then by every item's statistics read-out: the
        // thread that has read a pixel's accumulators clears them, halo rows included; all-zero bits are 0.0 and integer 0 alike)
        // run list of the integer path (the [pos-only] or [neg-only] slots of every pass / list) + accumulator choice
        const int nb = im.he - im.le, nlists = w.Md > 0 ? 2 : 1, nruns = nb * nlists;
        if (threadIdx.x < 2) s_flags[threadIdx.x] = 0;
        __syncthreads();                                  // (also: everybody has read s_item)
        int next_item = 0;
        if (threadIdx.x == 0) next_item = atomicAdd(&queue[xcd], 1);      // in flight while this item is worked on
        if ((int)threadIdx.x < nruns) {
            const int r = threadIdx.x;
            const bool isd = r >= nb;
            const int t = im.le + (isd ? r - nb : r);
            const int *cl = (isd ? d.cls : g.cls) + ((size_t)b * TEF_MAX_PASSES + t) * 3;
            run_u0[r] = (isd ? w.M + w.doff[t] : w.off[t]) + (c ? cl[0] : 0);
            run_len[r] = c ? cl[1] - cl[0] : cl[0];
            if (cl[2] != cl[1]) atomicOr(&s_flags[0], 1);                  // general masks present
            atomicAdd(&s_flags[1], run_len[r]);
        }
        __syncthreads();
        if (threadIdx.x < 64) {       // rows (16 slots) spanned by every run, as a running total: one wavefront scans
            int carry = 0;
            for (int base_r = 0; base_r < nruns; base_r += 64) {
                const int r = base_r + (int)threadIdx.x;
                int cnt = 0;
                if (r < nruns && run_len[r] > 0) cnt = ((run_u0[r] + run_len[r] - 1) >> 4) - (run_u0[r] >> 4) + 1;
                int incl = cnt;
                for (int sft = 1; sft < 64; sft <<= 1) {
                    int up = __shfl_up(incl, sft, 64);
                    if ((int)threadIdx.x >= sft) incl += up;
                }
                if (r < nruns) run_cum[r] = carry + incl - cnt;
                carry += __shfl(incl, 63, 64);
            }
            if (threadIdx.x == 0) run_cum[nruns] = carry;
        }
        __syncthreads();
        const bool fixed = s_flags[0] == 0 && s_flags[1] < kFxMaxEvents && (w.comp || w.kind == TEF_KIND_ITERATIVE);
        const float2 *pl = traj + uniform_off(((size_t)ib * (w.nplanes + 1) + im.plane) * w.Mt);
        const uint2 *mt = meta + uniform_off((size_t)ib * w.Mt);
        if (fixed) {
            // Wave-centric sweep.  The rows (16 slots) of all runs form one flattened sequence; a wavefront takes 128 rows at
            // a time (eight chunks of 16, see load_range): every lane reads two row intervals, the rows that can touch
            // the band are compacted into the wavefront's list, and the wavefront then works through the list in batches of
            // 16 rows (four quads of 4 x 16 lanes) with the next batch's events already in flight.  Iterations are dense in
            // work whatever the fraction of rows that hit (a workgroup-wide chunk loop spent a memory round trip per mostly
            // skipped chunk: 0.30 ms instead of 0.21).
            const float2 *rows = yr + uniform_off(((size_t)ib * (w.nplanes + 1) + im.plane) * w.nrow);
            const int lane = threadIdx.x & 63, wid = threadIdx.x >> 6, nwaves = blockDim.x >> 6;
            const int total_rows = run_cum[nruns];
            int *ring = hit_ring + wid * kRing;      // this wavefront's FIFO of hit rows (entry i lives at i & (kRing - 1))
            int run_hint = 0;                        // run of the first row of the group being located (wave-uniform, monotone)
            // Rows are dealt to the wavefronts in chunks of 16 (one 128-byte line of intervals), chunk c to wavefront
            // c % nwaves: hits cluster over hundreds of rows (the passes are sorted by tile), so with 128 consecutive rows
            // per wavefront and ~3 such groups per item the wavefronts waited 30 % of an item for the slowest of them.
            // m-th load of a wavefront: its chunks 4m .. 4m + 3, 16 lanes each.
            auto load_range = [&](int m, int &row_out) -> float2 {
                const int first = (m * 4 * nwaves + wid) * 16;                      // (wave-uniform)
                const int fr = first + (lane >> 4) * (nwaves * 16) + (lane & 15);
                float2 rg = make_float2(__uint_as_float(0xffffffffu), 0.0f);
                row_out = 0;
                if (first < total_rows) {
                    while (__builtin_amdgcn_readfirstlane(run_cum[run_hint + 1]) <= first) ++run_hint;
                }
                if (fr < total_rows) {
                    int r = run_hint;                // the lanes' rows follow the group's first: a step or two at most
                    while (run_cum[r + 1] <= fr) ++r;
                    row_out = (run_u0[r] >> 4) + (fr - run_cum[r]);
                    rg = rows[row_out];
                }
                return rg;
            };
            // A batch = 16 ring entries = four quads of (4 rows x 16 slots).  An event visit is two 8-byte loads at the same
            // 32-bit offset from scalar bases.  Entries past the tail hold rows seen earlier (or row 0): their loads are
            // harmless and they are discarded when the batch is processed (`rem`).
            constexpr int kQ = 4;
            struct Quad { uint32_t mv; float y, x, ts; };
            const uint32_t lane_off = (uint32_t)(lane & 15) * 8u;
            const int lane_grp = lane >> 4;
            auto load_batch = [&](Quad (&qd)[kQ], int pos) {
#pragma unroll
                for (int k = 0; k < kQ; ++k) {
                    const uint32_t off = (uint32_t)ring[(pos + 4 * k + lane_grp) & (kRing - 1)] * 128u + lane_off;   // slot (16 row + lane) x 8 bytes
                    const uint2 m2 = *at_bytes(mt, off);
                    const float2 pp = *at_bytes(pl, off);
                    // (nothing here may LOOK at the loaded registers: a select on them makes the compiler wait for the
                    // load right where it was issued — that is how round 3's first version of this loop lost its
                    // prefetch)
                    qd[k].mv = m2.x;
                    qd[k].ts = __uint_as_float(m2.y);
                    qd[k].y = pp.x;                    // traj stores (y, x) in (.x, .y)
                    qd[k].x = pp.y;
                }
            };
            unsigned long long *cpl = reinterpret_cast<unsigned long long *>(img_c), *tpl = reinterpret_cast<unsigned long long *>(img_t);
            const uint32_t polbit = c ? kMetaNeg : kMetaPos;
            const int r0m1 = r0 - 1;
            const uint32_t need = polbit | (1u << im.s);
            // rem: list entries left at the batch's first one (a quad's lane group g holds entry 4 k + g)
            auto process = [&](const Quad (&qd)[kQ], int rem) {
#pragma unroll
                for (int k = 0; k < kQ; ++k) {
                    const uint32_t mv = qd[k].mv;
                    bool take = (mv & need) == need;                       // this polarity + border mask (:671-681)
                    if (MODE == 1) {
                        const int kb1 = (int)((mv >> 8) & 0xffu), kf = (int)((mv >> 16) & 0xffu);
                        take = (mv & polbit) != 0u && im.plane >= kb1 && im.plane < kf;
                    }
                    const int rr = (int)floorf(qd[k].y) - r0m1;          // halo row 0 = image row r0 - 1
                    take = take && (unsigned)rr <= (unsigned)nrows && lane_grp < rem - 4 * k;
                    if (take) {
                        // tau = 1 - |tref - ts| / delta (:94-95); delta is an integer number of passes: exact division by a constant
                        const float tau = 1.0f - div_by_const(fabsf(im.tref - qd[k].ts), rdelta);
                        splat_fixed(qd[k].y, qd[k].x, tau, rr, WP, cpl, tpl);
                    }
                }
            };
            // The hit rows of the whole item form ONE stream per wavefront: candidate groups of 64 rows (one interval per lane,
            // two groups in flight) are tested and their hits appended to the ring; batches are taken from its head.  The
            // event loads of the next batch are always in flight, across candidate groups (round 2 / the first round-3
            // version drained and restarted the load pipeline every 128 candidates and issued a batch of discarded loads
            // each time; the scatter is bound by how many bytes a wavefront keeps in flight).
            int tail = 0, head = 0;                  // entries produced / handed to load_batch (wave-uniform)
            int next_m = 0;                          // oldest candidate group in flight
            float2 rg_a, rg_b;
            int row_a, row_b;
            rg_a = load_range(0, row_a);
            rg_b = load_range(1, row_b);
            auto more = [&]() { return (next_m * 4 * nwaves + wid) * 16 < total_rows; };      // (wave-uniform)
            auto top_up = [&]() {                    // keep two batches' worth of entries ahead of the loads while candidates last
                while (tail - head < 8 * kQ && more()) {
                    const bool hit = rg_a.y >= band_lo && rg_a.x < band_hi;      // (NaN for an empty / absent row)
                    const unsigned long long mask = __builtin_amdgcn_ballot_w64(hit);
                    if (hit) ring[(tail + __builtin_amdgcn_mbcnt_hi((unsigned)(mask >> 32), __builtin_amdgcn_mbcnt_lo((unsigned)mask, 0))) & (kRing - 1)] = row_a;
                    tail += __builtin_popcountll(mask);
                    rg_a = rg_b;
                    row_a = row_b;
                    ++next_m;
                    rg_b = load_range(next_m + 1, row_b);
                }
                __builtin_amdgcn_wave_barrier();
            };
            // Two register sets in turn.  Every load_batch is UNCONDITIONAL on the path to the process() that follows it: the
            // hardware counts outstanding loads in order, and where a path may or may not have issued the younger batch the
            // compiler has to wait for the smaller count — i.e. for the prefetch itself.
            // (ring capacity: at most 8 kQ - 1 + 64 entries are ahead of `head`, 8 kQ behind it are still being worked on)
            Quad qa[kQ], qb[kQ];
            top_up();
            load_batch(qa, head);
            int rem_a = tail - head, rem_b;
            head += 4 * kQ;
            for (;;) {
                top_up();
                load_batch(qb, head);
                rem_b = tail - head;
                head += 4 * kQ;
                process(qa, rem_a);
                if (rem_b <= 0) break;               // (the stream had run dry when qb was issued)
                top_up();
                load_batch(qa, head);
                rem_a = tail - head;
                head += 4 * kQ;
                process(qb, rem_b);
                if (rem_a <= 0) break;
            }
        } else {
            for (int li = 0; li < nlists; ++li)
                for (int t = im.le; t < im.he; ++t) {
                    const int *cl = (li ? d.cls : g.cls) + ((size_t)b * TEF_MAX_PASSES + t) * 3;
                    const int s0 = li ? w.M + w.doff[t] : w.off[t];
                    const int n0 = cl[0], n01 = cl[1], n012 = cl[2];
                    double *gc = img_c + WP, *gt = img_t + WP;          // (the band proper: past the halo row)
                    if (c == 0) {
                        splat_run_general(w, im, rdelta, g, d, b, c, s0, n0, pl, mt, gc, gt, r0, nrows);
                        splat_run_general(w, im, rdelta, g, d, b, c, s0 + n01, n012 - n01, pl, mt, gc, gt, r0, nrows);
                    } else {
                        splat_run_general(w, im, rdelta, g, d, b, c, s0 + n0, n012 - n0, pl, mt, gc, gt, r0, nrows);
                    }
                }
        }
        __syncthreads();
        // ---- band statistics of this polarity (round 1: a separate launch over the stored images) ----
        float acc = 0.0f;
        const size_t qpol = ((size_t)j * FB + ib) * 2 + c, o = qpol * HW + (size_t)r0 * W;
        unsigned long long *nzo = nzw + (qpol * nbands + band) * nz_cap;
        if (fixed) band_stats<true, true>(lds_img + WP, plane_sz, nrows, W, WP, ar + o, nzo, acc);
        else band_stats<false, true>(lds_img + WP, plane_sz, nrows, W, WP, ar + o, nzo, acc);
        const double dacc = wave_sum_to_last((double)acc);
        if ((threadIdx.x & 63) == 63) part[(qpol * nbands + band) * kSplat2Waves + (threadIdx.x >> 6)] = dacc;      // (image_count_kernel adds the wavefronts' shares, fixed order)
        if (threadIdx.x == 0) s_item = next_item;
        __syncthreads();                                  // the planes have been read, the next item may clear them
    }
}

// K4: per-image statistics from the band parts of K2, stats[q] = (sum A^2 / n, n = #active pixels + 1e-9), and
// loss = sum_images coef * (sum over samples of the per-sample term); fixed summation order.
__global__ __launch_bounds__(256) void loss_reduce_kernel(Win w, const double *__restrict__ sq,
                                                          const double *__restrict__ counts, float *__restrict__ stats,
                                                          const int *__restrict__ bad_img, float *__restrict__ loss_out)
{
    __shared__ double ssum[256];
    const int FB = w.F * w.B;
    int any_bad = 0;                                  // K1's "unrepresentable input" words, folded per image by image_count
    double acc = 0.0;
    for (int q = threadIdx.x; q < w.nimg * FB; q += blockDim.x) {
        any_bad |= bad_img[q];
        const double s2 = sq[q], cnt = counts[q];      // (image_count_kernel: the bands' and polarities' shares, fixed order)
        float n = w.scaling ? (float)cnt + kEps : 1.0f;      // loss/flow.py:124-127
        float term = (float)s2 / n;
        stats[(size_t)q * 2] = term;
        stats[(size_t)q * 2 + 1] = n;
        Img im = decode_image(w, q / FB);
        acc += (double)term * (double)im.coef;
    }
    any_bad = __syncthreads_or(any_bad);
    ssum[threadIdx.x] = acc;
    __syncthreads();
    for (int s = blockDim.x >> 1; s > 0; s >>= 1) {
        if ((int)threadIdx.x < s) ssum[threadIdx.x] += ssum[threadIdx.x + s];
        __syncthreads();
    }
    if (threadIdx.x == 0) loss_out[0] = any_bad ? __uint_as_float(0x7fc00000u) : (float)ssum[0];
}

// ---------------------------------------------------------------------------------------------
// The gradient of an image's focus term w.r.t. its two images, PER PIXEL and in autograd's own arithmetic.  The reference
// never forms the closed form 2 A (tau - A) / (C + eps): the backward of iwe_ts / (iwe + 1e-9) (loss/flow.py:727) hands
// dT = gA / (C + eps) to the T image and dC = -gA * ((T / (C + eps)) / (C + eps)) to the C image, with gA = g * (2 A)
// from the square (:122) and g the upstream gradient after the scalar divisions of :740-741, :731-732 and :127 —
// ((((grad_out / F) / S) / D) / 2^s) / n, D = 2 delta + 1 (Iterative) or 2 (Linear), true fp32 divisions in that order —
// and an event's corner weight then collects dC + dT * tau through the four scatters.  Where a pixel holds a single event,
// tau - A is ~1e-9 / C and dC, dT * tau cancel to a 1e-7 share of themselves: the closed form (more accurate) and the
// reference then differ by the reference's own rounding, 1 / C times larger than the result — 1.4e-5 of the largest
// gradient on the BASELINE window, 1e-3 on windows of a few events.  With the same operations on the same bits the
// difference is the images' summation order only.
// The divisions are the compiler's IEEE sequence (correctly rounded, like the CPU's): K6 is bound by its L1 gathers and
// has the vector-ALU slots for them (as a separate pass over the (A, C + eps) planes, read + written once more, the same
// arithmetic cost 31 us and 185 MB per step).
// ---------------------------------------------------------------------------------------------
// upstream gradient of one image's per-sample focus term: the reference's division chain (n = active pixels + 1e-9, or 1)
__device__ __forceinline__ float image_upstream(const Win &w, float gout, int s, float delta, float n)
{
    float g = gout / (float)w.F;                      // loss /= num_flows            (:741 / :402)
    g = g / (float)w.S;                               // loss /= scales_loss          (:740 / :401)
    g = g / (w.kind == TEF_KIND_ITERATIVE ? 2.0f * delta + 1.0f : 2.0f);      // deblurring points per window (:732 / :397)
    g = g / (float)(1 << s);                          // windows of the scale         (:731 / :396)
    return g / n;                                     //                              (:127)
}
// (A, C + eps) of a pixel -> d/dw of one corner: dC + dT * tau.  A corner outside the image was loaded as (0, 0): 0.
__device__ __forceinline__ void pixel_grads(float a, float ce, float g, float &dc, float &dt)
{
    const float ga = g * (2.0f * a);                  // pow backward: grad * (2 * self)
    dt = ga / ce;                                     // div backward, self:  grad / other
    dc = -(ga * (a / ce));                            // div backward, other: -grad * ((self / other) / other), self / other = A
}
__device__ __forceinline__ float2 pixel_grads_guarded(float2 p, float g)
{
    float dc, dt;
    pixel_grads(p.x, p.y != 0.0f ? p.y : 1.0f, g, dc, dt);
    return make_float2(dc, dt);
}

// ---------------------------------------------------------------------------------------------
// d(coef * image loss)/d position of one event at one image:
//   dl/dw_k = sum_c (dC_c m_c) + (sum_c dT_c m_c) * tau     (pixel_grads on K2's (A, C + eps) planes; autograd's order for one-hot masks)
// followed by the derivative of the bilinear hat weights.
// ---------------------------------------------------------------------------------------------
// FAST (the backward of the single-scale Iterative loss): delta is kernel-invariant there, an integer number of passes,
// so tau takes the cheaper exact form of the same division (div_by_const).
// INTERIOR (decided per wavefront by the caller): all four corners inside the image, the right column adjacent to the
// left one and a single polarity — the validity selects, the fp32 corner case and the second-polarity branch drop out.
// pos: the image's positive-polarity (A, C + eps) plane (the negative one follows it); g = image_upstream of the image
template <bool FAST = false, bool INTERIOR = false>
__device__ __forceinline__ float2 image_grad_at(int H, int W, const float2 *__restrict__ pos, float g, float tref,
                                                float delta, const Splat &sp, float ts, float mp, float mn)
{
    const int HW = H * W;
    const float2 *neg = pos + HW;
    struct { int H, W; } w = {H, W};
    float tau = FAST ? 1.0f - div_by_const(fabsf(tref - ts), 1.0 / (double)delta) : 1.0f - fabsf(tref - ts) / delta;
    float gy = 0.0f, gx = 0.0f;
    if (INTERIOR) {
        const float2 *pl = (mp != 0.0f) ? pos : neg;
        const float m1 = (mp != 0.0f) ? mp : mn;
        f32x4_a8 r[2];
#pragma unroll
        for (int k = 0; k < 2; ++k) r[k] = *reinterpret_cast<const f32x4_a8 *>(pl + sp.iy[k] * w.W + sp.ix[0]);
#pragma unroll
        for (int k = 0; k < 2; ++k) {
            float c0, t0, c1, t1;
            pixel_grads(r[k].x, r[k].y, g, c0, t0);
            pixel_grads(r[k].z, r[k].w, g, c1, t1);
            const float dw0 = c0 * m1 + (t0 * m1) * tau;
            const float dw1 = c1 * m1 + (t1 * m1) * tau;
            gy += dw0 * (sp.sy[k] * sp.wx[0]);
            gx += dw0 * (sp.wy[k] * sp.sx[0]);
            gy += dw1 * (sp.sy[k] * sp.wx[1]);
            gx += dw1 * (sp.wy[k] * sp.sx[1]);
        }
        return make_float2(gy, gx);
    }
    const bool vx0 = (sp.ix[0] >= 0) & (sp.ix[0] < w.W), vx1 = (sp.ix[1] >= 0) & (sp.ix[1] < w.W);
    // One polarity plane per event in the common case (masks are (1,0) / (0,1)): its two rows are fetched by two
    // unconditional 16-byte loads; an event carrying both polarities adds the second plane in a (rare) branch.
    const bool hp = mp != 0.0f, both = hp & (mn != 0.0f);
    const float2 *pl = hp ? pos : neg;
    const float m1 = hp ? mp : mn;
    float2 a0[2], a1[2];          // (A, C + eps) at [row][left / right]
    int i1s[2];
#pragma unroll
    for (int r = 0; r < 2; ++r) {
        int iy = sp.iy[r];
        bool vy = (iy >= 0) & (iy < w.H);
        int i0 = (vy && vx0) ? iy * w.W + sp.ix[0] : -1;
        int i1 = (vy && vx1 && sp.ix[1] == sp.ix[0] + 1) ? iy * w.W + sp.ix[1] : -1;   // adjacent unless x+1 rounded up
        i1s[r] = (vy && vx1 && i1 < 0) ? iy * w.W + sp.ix[1] : -1;
        load_row(pl, i0, i1, HW, a0[r], a1[r]);
    }
#pragma unroll
    for (int r = 0; r < 2; ++r) {
        if (i1s[r] >= 0) a1[r] = pl[i1s[r]];            // fp32 corner case: floor(x + 1) == floor(x) + 2
        const float2 p0 = pixel_grads_guarded(a0[r], g), p1 = pixel_grads_guarded(a1[r], g);
        float c0 = p0.x * m1, t0 = p0.y * m1, c1 = p1.x * m1, t1 = p1.y * m1;
        if (both) {
            int iy = sp.iy[r];
            bool vy = (iy >= 0) & (iy < w.H);
            float2 n0 = make_float2(0.f, 0.f), n1 = n0;
            if (vy && vx0) n0 = neg[iy * w.W + sp.ix[0]];
            if (vy && vx1) n1 = neg[iy * w.W + sp.ix[1]];
            n0 = pixel_grads_guarded(n0, g);
            n1 = pixel_grads_guarded(n1, g);
            c0 += n0.x * mn; t0 += n0.y * mn;
            c1 += n1.x * mn; t1 += n1.y * mn;
        }
        // invalid corners read (A, C + eps) = (0, 0): their dw is 0
        const float dw0 = c0 + t0 * tau, dw1 = c1 + t1 * tau;
        gy += dw0 * (sp.sy[r] * sp.wx[0]);
        gx += dw0 * (sp.wy[r] * sp.sx[0]);
        gy += dw1 * (sp.sy[r] * sp.wx[1]);
        gx += dw1 * (sp.wy[r] * sp.sx[1]);
    }
    return make_float2(gy, gx);
}

// s: the image's temporal scale; gout: d (total loss) / d (this loss)
template <bool FAST = false, bool INTERIOR = false>
__device__ __forceinline__ float2 image_grad(const Win &w, const float2 *__restrict__ ar, const float *__restrict__ stats,
                                             float gout, int s, int ib, int j, float tref, float delta, const Splat &sp,
                                             float ts, float mp, float mn)
{
    const size_t q = (size_t)j * (w.F * w.B) + ib;
    const float g = image_upstream(w, gout, s, w.kind == TEF_KIND_ITERATIVE ? delta : 0.0f, stats[q * 2 + 1]);
    return image_grad_at<FAST, INTERIOR>(w.H, w.W, ar + q * 2 * (size_t)(w.H * w.W), g, tref, delta, sp, ts, mp, mn);
}

// The same gradient when the cell is inside the frame, one polarity with mask value 1, one temporal scale (decided per
// wavefront by the caller): the validity selects, the fp32 corner case's extra load and the second-polarity branch drop
// out.  The hat weights and their slopes come from the caller — near corner: 1 - d with slope -1 (0 at d == 0: abs'(0));
// far corner: max(0, v), v = 1 - |p - floor(p + 1)|, with slope +1, +0.5 at the tie v == 0 (torch.max splits it) and 0
// where fp32 rounded p + 1 up (v < 0) — so exact ties stay on this path.  Same operations on the same values in the same
// order as image_grad<true, true>.  r0 / r1: the (A, C + eps) pairs of the cell's upper and lower pixel rows (loaded by the
// caller, one chain step ahead); g = image_upstream of the image.
__device__ __forceinline__ float corner_dw(float a, float ce, float g, float tau)
{
    float dc, dt;
    pixel_grads(a, ce, g, dc, dt);
    return dc + dt * tau;
}
__device__ __forceinline__ float2 cell_grad(const f32x4_a8 r0, const f32x4_a8 r1, float g, float tau, float wy0, float sy0,
                                            float wy1, float sy1, float wx0, float sx0, float wx1, float sx1)
{
    float gy = 0.0f, gx = 0.0f;
    {
        const float dw0 = corner_dw(r0.x, r0.y, g, tau), dw1 = corner_dw(r0.z, r0.w, g, tau);
        gy += dw0 * (sy0 * wx0);
        gx += dw0 * (wy0 * sx0);
        gy += dw1 * (sy0 * wx1);
        gx += dw1 * (wy0 * sx1);
    }
    {
        const float dw0 = corner_dw(r1.x, r1.y, g, tau), dw1 = corner_dw(r1.z, r1.w, g, tau);
        gy += dw0 * (sy1 * wx0);
        gx += dw0 * (wy1 * sx0);
        gy += dw1 * (sy1 * wx1);
        gx += dw1 * (wy1 * sx1);
    }
    return make_float2(gy, gx);
}

template <bool FAST = false>
__device__ __forceinline__ float2 image_grad(const Win &w, const float2 *__restrict__ ar, const float *__restrict__ stats,
                                             float gout, int s, int ib, int j, float tref, float delta, float2 p, float ts,
                                             float mp, float mn)
{
    Splat sp = make_splat(p.x, p.y);
    return image_grad<FAST, false>(w, ar, stats, gout, s, ib, j, tref, delta, sp, ts, mp, mn);
}

// gradient w.r.t. the event position at tref = k, summed over the temporal scales that use it (Iterative)
__device__ __forceinline__ float2 iter_position_grad(const Win &w, const float2 *__restrict__ ar,
                                                     const float *__restrict__ stats, float gout, int ib, uint32_t bits,
                                                     int t, int k, float2 p, float ts, float mp, float mn)
{
    float2 g = make_float2(0.0f, 0.0f);
    for (int s = 0; s < w.S; ++s) {
        if (!((bits >> s) & 1u)) continue;
        int scale = w.P >> s, wi = t / scale;
        int lo = wi * scale, hi = lo + scale;
        if (k < lo || k > hi) continue;
        int delta = scale / w.mode_div;
        int le = max(lo, k - delta), he = min(hi, k + delta);
        if (t < le || t >= he) continue;
        int j = w.img_base[s] + wi * (scale + 1) + (k - lo);
        float2 a = image_grad(w, ar, stats, gout, s, ib, j, (float)k, (float)delta, p, ts, mp, mn);
        g.x += a.x;
        g.y += a.y;
    }
    return g;
}

// =============================================================================================
// K6 (Iterative): reverse sweep along each grad event's trajectory.
// Autograd counterpart of loss/flow.py:555-584: p' = p + dt * f(p) with f = bilinear lookup, so the
// adjoint picks up (I + dt * J^T) per step, and every step leaves dt * adjoint as the gradient of the
// sampled flow vector.  cy/cx[(ib*P + k)*M + sl] = d/d f_y, d/d f_x of this event's sample of map k.
// =============================================================================================
// Streaming accesses of K6 (trajectory planes in, per-map vectors out) carry the non-temporal hint so that they do not
// push the (A, R) images and flow maps — the gathered, re-used data — out of the XCD's L2.
// Largest magnitude among the flow-gradient vectors of one (head, sample), as the bit pattern of |c| (orders like an
// unsigned integer; NaN / infinity sort above every finite value): K7 scales its integer accumulators by it.
__device__ __forceinline__ void track_mag(uint32_t &m, float a, float b)
{
    m = max(m, max(__float_as_uint(a) & 0x7fffffffu, __float_as_uint(b) & 0x7fffffffu));
}
// One plain store per wavefront into wmax[(head, sample)][wavefront]; mag_reduce_kernel folds them.  (4e5 wavefronts on
// 32 words: an atomicMax each doubled K6's time, 0.23 -> 0.50 ms; guarded by a coherent read of the word still 0.28.)
__device__ __forceinline__ uint32_t commit_mag(uint32_t m, uint32_t *__restrict__ wave_slot)
{
    for (int sft = 32; sft > 0; sft >>= 1) m = max(m, (uint32_t)__shfl_xor((int)m, sft, 64));
    if ((threadIdx.x & 63) == 0) *wave_slot = m;
    return m;
}

__global__ __launch_bounds__(256) void mag_reduce_kernel(const uint32_t *__restrict__ wmax, int nwaves, uint32_t *__restrict__ cmax,
                                                         int *__restrict__ queue7)
{
    __shared__ uint32_t red[4];
    if (blockIdx.x == 0 && threadIdx.x < 9) queue7[threadIdx.x] = 0;      // K7's work queues (+ K6's deferred-wavefront count), once per backward call
    uint32_t m = 0u;
    for (int k = threadIdx.x; k < nwaves; k += blockDim.x) m = max(m, wmax[(size_t)blockIdx.x * nwaves + k]);
    for (int sft = 32; sft > 0; sft >>= 1) m = max(m, (uint32_t)__shfl_xor((int)m, sft, 64));
    if ((threadIdx.x & 63) == 0) red[threadIdx.x >> 6] = m;
    __syncthreads();
    if (threadIdx.x == 0) cmax[blockIdx.x] = max(max(red[0], red[1]), max(red[2], red[3]));
}


// ONE: a single temporal scale (scales_loss = 1, the headline configuration).  The kernel is VALU-bound (~450 vector
// instructions per chain step, 2.9e7 steps per BASELINE window): the per-step scale loop with its integer division
// (t / scale) and the normalisation constant are hoisted, and 1 / n is a reciprocal (image_grad<true>).
// PART (single-scale kernel only; 0 = everything in one launch): 1 = the fast form of every step, unconditionally — a
// wavefront in which some active lane needed the general form somewhere (a corner or tap outside the frame, two polarities,
// a mask value other than 1) computes that lane from clamped addresses, appends itself to `defer` and is recomputed by
// PART 2, a few workgroups that walk the list with both forms.  In one launch the general form's code (sixteen guarded
// divisions since round 4) sat in the sweeps' loop bodies and cost every wavefront its schedule: 0.228 ms where the fast form
// alone takes 0.206 (a second loop, an out-of-line call or an early exit in the same kernel: 0.235 / 0.232 / 0.215 — the
// register allocation and the loop shape are the kernel's).
template <bool ONE, int PART>
__device__ __forceinline__ void chain_bwd_wave(const Win &w, const float2 *__restrict__ flows, const Events &g,
                                               const float2 *__restrict__ traj, const uint2 *__restrict__ meta,
                                               const float2 *__restrict__ ar, const float *__restrict__ stats,
                                               const float *__restrict__ grad_out, float2 *__restrict__ cyx,
                                               uint32_t *__restrict__ cmax, int ib, int sl, int *__restrict__ defer_count,
                                               int *__restrict__ defer, uint32_t *__restrict__ cmax_ib)
{
    uint32_t mag = 0u;
    unsigned long long strayed = 0ull;                  // PART 1: lanes that needed the general form at some step
    int i = ib / w.B, b = ib - i * w.B;
    const int H = w.H, W = w.W, P = w.P, M = w.M;
    float2 *co = cyx + (size_t)ib * P * M + sl;
    const uint2 mv_ts = meta[(size_t)ib * w.Mt + sl];
    const uint32_t mv = mv_ts.x;
    // Passes start at multiples of 64 slots, so a wavefront belongs to ONE pass: t, and with it the reference time k of
    // every loop iteration below, is wave-uniform — map / image / plane base addresses and the image statistics are
    // scalar-register arithmetic and scalar loads instead of per-lane 64-bit index math in a VALU-bound kernel.  Lanes
    // whose chain has not started yet or has left the frame are masked per iteration.
    const int t = __builtin_amdgcn_readfirstlane((int)g.bin[sl]);
    uint32_t bits = mv & 0xffu;
    if (!w.comp) {
        // without border compensation: every scale whose windows hold pass t (the trailing passes of a window length that
        // does not divide P belong to no window of that scale, exactly as in K1's bits), as far as the chain is alive
        uint32_t all = 0u;
        for (int s = 0; s < w.S; ++s)
            if (t / (P >> s) < (1 << s)) all |= 1u << s;
        bits = (mv & (kMetaPos | kMetaNeg)) ? all : 0u;
    }
    // map k receives something from pass t only if |k - t| < delta_passes[0] (largest window and reach); K7 reads
    // exactly those (pass, map) pairs, so only they need a value
    const int reach = P / w.mode_div;
    int kb = (int)((mv >> 8) & 0xffu) - 1, kf = (int)((mv >> 16) & 0xffu);
    size_t o = (size_t)b * g.cap + sl;
    float ts = __uint_as_float(mv_ts.y), mp = g.mp[o], mn = g.mn[o];
    const float gout = grad_out[0];
    const float2 *tr = traj + (size_t)ib * (w.nplanes + 1) * w.Mt + sl;
    float c0y = 0.0f, c0x = 0.0f;

    // reference times that can carry a gradient for this event: tref k contributes at scale s iff the event's window is
    // valid, lo_s <= k <= hi_s and k - delta_s <= t < k + delta_s; a lane joins the sweeps at the outermost such tref
    // (beyond it the adjoint is still zero, so neither the IWE nor the flow Jacobian need to be looked up)
    int k_top = t, k_bot = t + 1;
    for (int s = 0; s < w.S; ++s) {
        if (!((bits >> s) & 1u)) continue;
        int scale = P >> s, wi = t / scale, lo = wi * scale, hi = lo + scale, delta = scale / w.mode_div;
        k_top = max(k_top, min(hi, t + delta));
        k_bot = min(k_bot, max(lo, t - delta + 1));
    }
    // first reference time of each sweep for this lane; an event without any valid window never joins
    const int ks_f = bits ? min(min(P, kf - 1), k_top) : t, ks_b = bits ? max(max(0, kb + 1), k_bot) : t + 1;
    // gradient w.r.t. the position at tref = k
    const float one_delta = (float)reach;
    // ONE: the upstream gradient of every image up to its pixel count (the reference's division chain, scale 0)
    const float one_g4 = ((gout / (float)w.F) / (float)w.S) / (2.0f * one_delta + 1.0f);
    auto pos_grad = [&](int k, float2 p) -> float2 {
        if (!ONE) return iter_position_grad(w, ar, stats, gout, ib, bits, t, k, p, ts, mp, mn);
        if (t < k - reach || t >= k + reach) return make_float2(0.0f, 0.0f);      // window [0, P], delta = reach
        return image_grad<true>(w, ar, stats, gout, 0, ib, w.img_base[0] + k, (float)k, one_delta, p, ts, mp, mn);
    };
    // (the fast form of a step also wants the event's mask value to be exactly 1: a product the reference makes, exact then)
    const bool one_pol = !((mp != 0.0f) & (mn != 0.0f)) & (((mp != 0.0f) ? mp : mn) == 1.0f);
    const int FB = w.F * w.B;
    float ay = 0.0f, ax = 0.0f;
    // ---- ONE: both sweeps as straight-line loop bodies ---------------------------------------------------------------
    // The reference time k of an iteration is wave-uniform, so everything that depends on it alone (map / image / plane /
    // output base addresses, the image's pixel count, "is tref k within reach of pass t") lives in scalar registers, the
    // gathers take the scalar-base form with a 32-bit lane offset, and lanes whose chain has not started yet or has left
    // the frame are handled by selects instead of divergent branches.
    // The fast form of a step needs EVERY active lane of the wavefront to have its four flow taps and its four image
    // corners inside the frame and one polarity with mask value 1 (~25 % fewer vector instructions; events on the frame's
    // last row / column send their wavefront through the general code; exact ties are handled here since round 4 — they
    // were all of the 0.26 % of wavefront-steps that left the fast form on the BASELINE window).  Same arithmetic in the
    // same order either way.
    const AxisConst ach = axis_const(H), acw = axis_const(W);
    const double rdelta = uniform_f64(1.0 / (double)one_delta);
    // D = -1: forward chain, newest tref first (k = min(P, t + reach) .. t + 1, map k - 1, p_k = p_{k-1} + dt f_{k-1});
    // D = +1: backward chain, oldest first (k = max(0, t - reach + 1) .. t, map k, p_k = p_{k+1} - f_k(p_{k+1})).
    auto sweep = [&](auto dir) {
        constexpr int D = decltype(dir)::value;
        const int k0 = D < 0 ? min(P, t + reach) : max(0, t - reach + 1);
        const int kend = D < 0 ? t + 1 : t;                       // last reference time of the sweep
        auto plane = [&](int k) {
            const float2 *pb = traj + uniform_off(((size_t)ib * (w.nplanes + 1) + (D < 0 ? max(k, kend) : min(k, kend))) * w.Mt);
            return NT_LD2(at_bytes(pb, (uint32_t)sl * 8u));
        };
        auto active = [&](int k) { return D < 0 ? k <= ks_f : k >= ks_b; };
        auto in_reach = [&](int k) { return t >= k - reach && t < k + reach; };      // window [0, P], delta = reach
        // the adjoint through step k: (ay, ax) += gk, then either the pass's own map (last step) or store + (I +- J^T)
        auto advance = [&](int k, bool act, float2 gk, float jyy, float jyx, float jxy, float jxx) {
            const int km = D < 0 ? k - 1 : k;
            // lanes whose chain has not started yet or has left the frame carry a zero adjoint
            ay += act ? gk.x : 0.0f;
            ax += act ? gk.y : 0.0f;
            if (km == t) {
                const float c = (float)kend - ts;
                c0y += c * ay;
                c0x += c * ax;
            } else {
                const size_t co = uniform_off(((size_t)ib * P + km) * M);
                NT_ST2(at_bytes(cyx + co, (uint32_t)sl * 8u), D < 0 ? ay : -ay, D < 0 ? ax : -ax);
                track_mag(mag, ay, ax);
                const float uy = ay * jyy + ax * jxy, ux = ay * jyx + ax * jxx;
                const float ny = D < 0 ? ay + uy : ay - uy, nx = D < 0 ? ax + ux : ax - ux;
                ay = act ? ny : 0.0f;
                ax = act ? nx : 0.0f;
            }
        };
        auto step = [&](int k, float2 cur, float2 nxt) {
            const bool act0 = active(k);
            const int km = D < 0 ? k - 1 : k;
            float jyy = 0.f, jyx = 0.f, jxy = 0.f, jxx = 0.f;
            float2 gk = make_float2(0.0f, 0.0f);
            int y0, x0;
            Taps tp = taps_core(nxt.x, nxt.y, ach, acw, y0, x0);
            // bilinear cell of `cur` (utils/iwe.py:85-107): near weights 1 - d with d = p - floor(p) in [0, 1), far weights
            // max(0, v), v = 1 - |p - floor(p + 1)|; slopes as hat() gives them, exact ties included (cell_grad)
            const float fy = floorf(cur.x), fx = floorf(cur.y);
            const float dy = cur.x - fy, dx = cur.y - fx;
            const float vy1 = 1.0f - fabsf(cur.x - floorf(cur.x + 1.0f)), vx1 = 1.0f - fabsf(cur.y - floorf(cur.y + 1.0f));
            const float wy1 = fmaxf(vy1, 0.0f), wx1 = fmaxf(vx1, 0.0f);
            const float sy0 = dy > 0.0f ? -1.0f : -0.0f, sx0 = dx > 0.0f ? -1.0f : -0.0f;
            const float sy1 = vy1 > 0.0f ? 1.0f : (vy1 == 0.0f ? 0.5f : 0.0f), sx1 = vx1 > 0.0f ? 1.0f : (vx1 == 0.0f ? 0.5f : 0.0f);
            const int iy0 = (int)fy, ix0 = (int)fx;
            const bool inside = one_pol & (y0 >= 0) & (y0 < H - 1) & (x0 >= 0) & (x0 < W - 1) & (iy0 >= 0) & (iy0 < H - 1) &
                                (ix0 >= 0) & (ix0 < W - 1);
            const unsigned long long stray = __builtin_amdgcn_ballot_w64(act0 & !inside);
            if (PART == 1) strayed |= stray;
            // (PART 1: a straying lane is computed like an inactive one — clamped addresses, zero adjoint; its wavefront is redone)
            const bool act = PART == 1 ? (act0 & inside) : act0;
            if (PART == 1 || stray == 0) {
                const int kmc = min(max(km, 0), P - 1);               // (the jacobian of the last step is not used)
                const float2 *fm = flows + uniform_off((((size_t)kmc * w.F + i) * w.B + b) * (size_t)(H * W));
                const uint32_t fo = act ? (uint32_t)(__mul24(y0, W) + x0) : 0u;
                const f32x4_a8 q0 = *reinterpret_cast<const f32x4_a8 *>(at_bytes(fm, fo * 8u));
                const f32x4_a8 q1 = *reinterpret_cast<const f32x4_a8 *>(at_bytes(fm, (fo + (uint32_t)W) * 8u));
                if (in_reach(k)) {
                    const size_t q = (size_t)(w.img_base[0] + k) * FB + ib;
                    const float gimg = one_g4 / stats[q * 2 + 1];
                    const float2 *pl = ar + uniform_off(q * 2 * (size_t)(H * W));
                    const uint32_t po = ((mp != 0.0f) ? 0u : (uint32_t)(H * W)) + (act ? (uint32_t)(__mul24(iy0, W) + ix0) : 0u);
                    const f32x4_a8 r0 = *reinterpret_cast<const f32x4_a8 *>(at_bytes(pl, po * 8u));
                    const f32x4_a8 r1 = *reinterpret_cast<const f32x4_a8 *>(at_bytes(pl, (po + (uint32_t)W) * 8u));
                    const float tau = 1.0f - div_by_const(fabsf((float)k - ts), rdelta);
                    gk = cell_grad(r0, r1, gimg, tau, 1.0f - dy, sy0, wy1, sy1, 1.0f - dx, sx0, wx1, sx1);
                }
                Quad2 q;
                q.v00 = make_float2(q0.x, q0.y); q.v01 = make_float2(q0.z, q0.w);
                q.v10 = make_float2(q1.x, q1.y); q.v11 = make_float2(q1.z, q1.w);
                quad_jacobian(q, tp, jyy, jyx, jxy, jxx);
            } else if (act) {
                if constexpr (PART != 1) {
                    Taps tg = make_taps(nxt.x, nxt.y, H, W);
                    quad_jacobian(load_quad(flow_map(w, flows, max(km, 0), i, b), tg, H * W), tg, jyy, jyx, jxy, jxx);
                    gk = pos_grad(k, cur);
                }
            }
            advance(k, act, gk, jyy, jyx, jxy, jxx);
        };
        // the positions of a step are loaded one iteration ahead (planes a lane's chain never reached hold stale values:
        // such lanes are inactive there)
        float2 cur = plane(k0), nxt = plane(k0 + D);
        // (wait for the two planes HERE: with loads pending at loop entry the compiler's wait-count merge makes every
        // iteration wait for the previous iteration's stores before it touches `nxt`)
        __builtin_amdgcn_s_waitcnt(0x0F70);                     // vmcnt(0)
        for (int k = k0;; k += D) {
            const float2 nn = plane(k + 2 * D);
            step(k, cur, nxt);
            if (k == kend) break;
            cur = nxt;
            nxt = nn;
        }
    };
    if (ONE) {
        sweep(std::integral_constant<int, -1>());
        ay = 0.0f;
        ax = 0.0f;
        sweep(std::integral_constant<int, 1>());
        if (PART == 1 && strayed != 0ull && (threadIdx.x & 63) == 0)      // (the second launch redoes the whole wavefront)
            defer[atomicAdd(defer_count, 1)] = ib * (M >> 6) + (sl >> 6);
    } else {
        {   // forward chain, newest first: p_k = p_{k-1} + dt * f_{k-1}(p_{k-1}), k = min(P, t + reach) .. t+1
            const int k0 = min(P, t + reach);
            // the positions of a step are loaded for every lane, one iteration ahead (planes a lane's chain never reached
            // hold stale values: such lanes are inactive there)
            float2 cur = NT_LD2(&tr[(size_t)max(k0, t + 1) * w.Mt]), nxt = NT_LD2(&tr[(size_t)max(k0 - 1, t + 1) * w.Mt]);
            for (int k = k0; k > t; --k) {
                float2 nn = NT_LD2(&tr[(size_t)max(k - 2, t + 1) * w.Mt]);
                if (k <= ks_f) {
                    float jyy = 0.f, jyx = 0.f, jxy = 0.f, jxx = 0.f;
                    Taps tp = make_taps(nxt.x, nxt.y, H, W);
                    quad_jacobian(load_quad(flow_map(w, flows, max(k - 1, 0), i, b), tp, H * W), tp, jyy, jyx, jxy, jxx);
                    float2 gk = pos_grad(k, cur);
                    ay += gk.x;
                    ax += gk.y;
                    if (k - 1 == t) {
                        float c = (float)(t + 1) - ts;
                        c0y += c * ay;
                        c0x += c * ax;
                    } else {
                        NT_ST2(&co[(size_t)(k - 1) * M], ay, ax);
                        track_mag(mag, ay, ax);
                        float ny = ay + (ay * jyy + ax * jxy), nx = ax + (ay * jyx + ax * jxx);
                        ay = ny;
                        ax = nx;
                    }
                } else if (k - 1 > t) {          // the chain has not started yet: this map gets nothing from the event
                    NT_ST2(&co[(size_t)(k - 1) * M], 0.0f, 0.0f);
                }
                cur = nxt;
                nxt = nn;
            }
        }
        ay = 0.0f;
        ax = 0.0f;
        {   // backward chain, oldest first: p_k = p_{k+1} - f_k(p_{k+1}), k = max(0, t - reach + 1) .. t
            const int k0 = max(0, t - reach + 1);
            float2 cur = NT_LD2(&tr[(size_t)min(k0, t) * w.Mt]), nxt = NT_LD2(&tr[(size_t)min(k0 + 1, t) * w.Mt]);
            for (int k = k0; k <= t; ++k) {
                float2 nn = NT_LD2(&tr[(size_t)min(k + 2, t) * w.Mt]);
                if (k >= ks_b) {
                    float jyy = 0.f, jyx = 0.f, jxy = 0.f, jxx = 0.f;
                    Taps tp = make_taps(nxt.x, nxt.y, H, W);
                    quad_jacobian(load_quad(flow_map(w, flows, k, i, b), tp, H * W), tp, jyy, jyx, jxy, jxx);
                    float2 gk = pos_grad(k, cur);
                    ay += gk.x;
                    ax += gk.y;
                    if (k == t) {
                        float c = (float)t - ts;
                        c0y += c * ay;
                        c0x += c * ax;
                    } else {
                        NT_ST2(&co[(size_t)k * M], -ay, -ax);
                        track_mag(mag, ay, ax);
                        float ny = ay - (ay * jyy + ax * jxy), nx = ax - (ay * jyx + ax * jxx);
                        ay = ny;
                        ax = nx;
                    }
                } else if (k < t) {
                    NT_ST2(&co[(size_t)k * M], 0.0f, 0.0f);
                }
                cur = nxt;
                nxt = nn;
            }
        }
    }
    NT_ST2(&co[(size_t)t * M], c0y, c0x);
    track_mag(mag, c0y, c0x);
    // (here `cmax`: the per-wavefront slots.  PART 1: a wavefront that strayed leaves 0 — its vectors are recomputed by the
    // finishing launch, which publishes their magnitude with an atomic of its own: chain_bwd_finish_kernel)
    const uint32_t wm = commit_mag((PART == 1 && strayed != 0ull) ? 0u : mag, cmax + (size_t)ib * (M >> 6) + (sl >> 6));
    if (PART == 2 && (threadIdx.x & 63) == 0) atomicMax(cmax_ib + ib, wm);
}

template <bool ONE, int PART = 0>
__global__ __launch_bounds__(256) void iter_chain_bwd_kernel(Win w, const float2 *__restrict__ flows, Events g,
                                                             const float2 *__restrict__ traj,
                                                             const uint2 *__restrict__ meta,
                                                             const float2 *__restrict__ ar,
                                                             const float *__restrict__ stats,
                                                             const float *__restrict__ grad_out,
                                                             float2 *__restrict__ cyx,
                                                             uint32_t *__restrict__ cmax, int chunks,
                                                             int *__restrict__ defer_count, int *__restrict__ defer,
                                                             uint32_t *__restrict__ cmax_ib)
{
    int ib, chunk;
    xcd_split(blockIdx.x, chunks, ib, chunk);
    if (ib >= w.F * w.B) return;
    if (PART == 1 && chunk == 0 && threadIdx.x == 0) cmax_ib[ib] = 0u;      // the finishing launch folds into it with atomicMax
    const int sl = chunk * blockDim.x + threadIdx.x;
    if (sl >= w.M) return;                              // (M is a multiple of 64: whole wavefronts)
    chain_bwd_wave<ONE, PART>(w, flows, g, traj, meta, ar, stats, grad_out, cyx, cmax, ib, sl, defer_count, defer, cmax_ib);
}

// K6's finishing launch (single-scale Iterative): R workgroups per (head, sample).  First every workgroup takes its share of
// the wavefronts the fast launch listed and recomputes them with both forms (normally none: the launch then costs what the
// magnitude reduce alone cost, and the separate 10 us launch round 4 first had is gone); then workgroup (ib, r) folds its
// slice of ib's per-wavefront magnitudes into cmax[ib].  No workgroup waits for another: a recomputed wavefront's slot was
// left at 0 by the fast launch and its magnitude goes into cmax[ib] by the recomputing wavefront's own atomicMax.
// The list's counter is cleared by K7 (and by K1): a workgroup of THIS launch may not have read it yet when another is done.
__global__ __launch_bounds__(256) void chain_bwd_finish_kernel(Win w, const float2 *__restrict__ flows, Events g,
                                                               const float2 *__restrict__ traj,
                                                               const uint2 *__restrict__ meta,
                                                               const float2 *__restrict__ ar,
                                                               const float *__restrict__ stats,
                                                               const float *__restrict__ grad_out,
                                                               float2 *__restrict__ cyx, uint32_t *__restrict__ wmax,
                                                               int *__restrict__ defer_count, int *__restrict__ defer,
                                                               uint32_t *__restrict__ cmax_ib, int R, int *__restrict__ queue7)
{
    __shared__ uint32_t red[4];
    if (blockIdx.x == 0 && threadIdx.x < 8) queue7[threadIdx.x] = 0;      // K7's work queues, once per backward call
    const int n = defer_count[0], per = w.M >> 6;
    for (int e = (int)blockIdx.x * 4 + (int)(threadIdx.x >> 6); e < n; e += (int)gridDim.x * 4) {
        const int id = defer[e], ib = id / per;
        chain_bwd_wave<true, 2>(w, flows, g, traj, meta, ar, stats, grad_out, cyx, wmax, ib,
                                (id - ib * per) * 64 + (int)(threadIdx.x & 63), defer_count, defer, cmax_ib);
    }
    const int ib = (int)blockIdx.x / R, r = (int)blockIdx.x - ib * R;
    const int lo = (int)(((long)per * r) / R), hi = (int)(((long)per * (r + 1)) / R);
    uint32_t m = 0u;
    for (int k = lo + (int)threadIdx.x; k < hi; k += blockDim.x) m = max(m, wmax[(size_t)ib * per + k]);
    for (int sft = 32; sft > 0; sft >>= 1) m = max(m, (uint32_t)__shfl_xor((int)m, sft, 64));
    if ((threadIdx.x & 63) == 0) red[threadIdx.x >> 6] = m;
    __syncthreads();
    if (threadIdx.x == 0) atomicMax(cmax_ib + ib, max(max(red[0], red[1]), max(red[2], red[3])));
}

// K6 (Linear): d/d(sampled flow) = sum over scales and both window ends of (tref - ts) * d/d position.
__global__ __launch_bounds__(256) void linear_bwd_kernel(Win w, Events g, const float2 *__restrict__ traj,
                                                         const uint2 *__restrict__ meta,
                                                         const float2 *__restrict__ ar,
                                                         const float *__restrict__ stats,
                                                         const float *__restrict__ grad_out, float2 *__restrict__ cyx,
                                                         uint32_t *__restrict__ cmax, int chunks)
{
    int ib, chunk;
    xcd_split(blockIdx.x, chunks, ib, chunk);
    if (ib >= w.F * w.B) return;
    int sl = chunk * blockDim.x + threadIdx.x;
    if (sl >= w.M) return;
    int b = ib % w.B;
    uint32_t bits = meta[(size_t)ib * w.Mt + sl].x & 0xffu;
    float gy = 0.0f, gx = 0.0f;
    if (bits) {
        size_t o = (size_t)b * g.cap + sl;
        float ts = g.ts[o], mp = g.mp[o], mn = g.mn[o];
        int t = g.bin[sl];
        const float gout = grad_out[0];
        const float2 *tr = traj + (size_t)ib * (w.nplanes + 1) * w.Mt + sl;
        for (int s = 0; s < w.S; ++s) {
            if (!((bits >> s) & 1u)) continue;
            int scale = w.P >> s, wi = t / scale;
            int lo = wi * scale, hi = lo + scale;
            for (int e = 0; e < 2; ++e) {
                float tref = (float)(e ? lo : hi);
                int j = w.img_base[s] + wi * 2 + e;
                float2 p = tr[(size_t)(2 * s + e) * w.Mt];
                float2 gp = image_grad(w, ar, stats, gout, s, ib, j, tref, (float)scale, p, ts, mp, mn);
                gy += (tref - ts) * gp.x;
                gx += (tref - ts) * gp.y;
            }
        }
    }
    cyx[(size_t)ib * w.M + sl] = make_float2(gy, gx);
    uint32_t mag = 0u;
    track_mag(mag, gy, gx);
    commit_mag(mag, cmax + (size_t)ib * (w.M >> 6) + (sl >> 6));
}

// =============================================================================================
// K7: flow-map gradient = bilinear splat (grid_sample backward w.r.t. input) of the per-event vectors.
// One workgroup per (pass k, head, sample, component[, band]); LDS holds one gradient plane.
// Sample position of event (bin t) on map k: t < k -> trajectory plane k, t > k -> plane k+1,
// t == k -> original location.  Linear: only the events of pass k sample map k.
//   dflows [P][F][B][2][H][W] (channel 0 = x, 1 = y) is fully overwritten.
// This kernel IS bound by its LDS atomics (230 M per launch: at ds_add_f64's 31 cycles per wave-instruction on a sorted
// wavefront's footprint that is the whole 0.18 ms), so it accumulates integers like K2 (ds_add_u64: 18 cycles) in a
// block-floating format: K6 publishes the largest magnitude among the vectors of each (head, sample), 2^e >= max |c|,
// and a contribution c * w (|w| <= 1) is accumulated as RN(c * w * 2^(46 - e)).  Sums are exact and order-independent
// (bitwise reproducible gradients); a contribution is resolved to 2^-46 of the largest vector of its (head, sample).
// fp64 accumulators remain for non-finite vectors and for 2^17 or more events per map.
// =============================================================================================
template <bool FX>
__device__ __forceinline__ float dflow_value(double a, double unscale)
{
    return FX ? (float)((double)__double_as_longlong(a) * unscale) : (float)a;
}

template <bool FX>
__device__ __forceinline__ void dflow_plane_store(const double *img, int rows, int W, int WP, double unscale,
                                                  float *__restrict__ o)
{
    const int half = W >> 1;
    if (!(W & 1) && !(WP & 1) && half > 0 && (int)blockDim.x % half == 0) {
        const int rstep = blockDim.x / half;
        int r = threadIdx.x / half, c = (threadIdx.x - r * half) * 2;
        for (; r < rows; r += rstep) {
            double2 d = *reinterpret_cast<const double2 *>(img + r * WP + c);
            *reinterpret_cast<float2 *>(o + (size_t)r * W + c) = make_float2(dflow_value<FX>(d.x, unscale), dflow_value<FX>(d.y, unscale));
        }
    } else {
        for (int p = threadIdx.x; p < rows * W; p += blockDim.x) {
            int r = p / W;
            o[p] = dflow_value<FX>(img[r * WP + (p - r * W)], unscale);
        }
    }
}

// one event's contribution to both gradient planes of a band (planes: [d/d flow_y | d/d flow_x], each nrows x WP)
template <bool FX>
__device__ __forceinline__ void dflow_one(float2 p, float cvy, float cvx, int H, int W, int WP, int e, double *img_y,
                                          double *img_x, int r0, int nrows)
{
    // same taps as the forward lookup (make_taps), kept as (row, column) to address the padded LDS planes
    float fiy = unnormalize(p.x, H), fix = unnormalize(p.y, W);
    float fy0 = floorf(fiy), fx0 = floorf(fix);
    float tn = fiy - fy0, tw = fix - fx0, tsv = 1.0f - tn, te = 1.0f - tw;
    int y0 = (int)fy0, x0 = (int)fx0;
    const float wt[4] = {tsv * te, tsv * tw, tn * te, tn * tw};
#pragma unroll
    for (int c = 0; c < 4; ++c) {
        const int iy = y0 + (c >> 1) - r0, ix = x0 + (c & 1);
        if (iy < 0 || iy >= nrows || ix < 0 || ix >= W) continue;
        const int cell = __mul24(iy, WP) + ix;
        const float vy = cvy * wt[c], vx = cvx * wt[c];
        if (FX) {
            atomicAdd(reinterpret_cast<unsigned long long *>(img_y + cell), to_fixed(ldexpf(vy, -e)));
            atomicAdd(reinterpret_cast<unsigned long long *>(img_x + cell), to_fixed(ldexpf(vx, -e)));
        } else {
            atomicAdd(img_y + cell, (double)vy);
            atomicAdd(img_x + cell, (double)vx);
        }
    }
}

// (the tap-by-tap integer form as a real call: it runs only for sampling locations outside the frame)
__device__ __noinline__ void dflow_one_fixed_slow(float2 p, float cvy, float cvx, int H, int W, int WP, int e, double *img_y,
                                                  double *img_x, int r0, int nrows)
{
    dflow_one<true>(p, cvy, cvx, H, W, WP, e, img_y, img_x, r0, nrows);
}

// The integer path's event (the hot path of K7; same diet as K2's splat_fixed): planes with one halo row above and below
// the band, so the eight accumulations are unconditional — grid_sample's zero padding drops the taps of row -1 / row H,
// which are halo rows of the first / last band, and a tap of column W lands in the row padding.  The vector arrives
// already scaled by 2^-e (a power of two commutes with the products).  Caller guarantees 0 <= x0 <= W - 1 and
// rr = y0 - (r0 - 1) in [0, nrows].  (fiy, fix) = unnormalize(p): the forward lookup's own coordinates.
__device__ __forceinline__ void dflow_fixed(float fiy, float fix, float fy0, float fx0, float sy, float sx, int rr, int WP,
                                            unsigned long long *py, unsigned long long *px)
{
    const f32x2_e fi = {fiy, fix}, f0 = {fy0, fx0};
    const f32x2_e tf = fi - f0;                          // (tn, tw)
    const f32x2_e tc = 1.0f - tf;                        // (ts, te)
    const f32x2_e wxp = {tc.y, tf.y};                    // (te, tw)
    const f32x2_e w0 = tc.x * wxp, w1 = tf.x * wxp;      // (w00, w01), (w10, w11)
    const f32x2_e a0 = sy * w0, a1 = sy * w1, b0 = sx * w0, b1 = sx * w1;
    const int cell = __mul24(rr, WP) + (int)fx0;
    unsigned long long *c0 = py + cell, *q0 = px + cell;
    atomicAdd(c0, to_fixed(a0.x));
    atomicAdd(c0 + 1, to_fixed(a0.y));
    atomicAdd(c0 + WP, to_fixed(a1.x));
    atomicAdd(c0 + WP + 1, to_fixed(a1.y));
    atomicAdd(q0, to_fixed(b0.x));
    atomicAdd(q0 + 1, to_fixed(b0.y));
    atomicAdd(q0 + WP, to_fixed(b1.x));
    atomicAdd(q0 + WP + 1, to_fixed(b1.y));
}

// Persistent like K2: two workgroups per CU pull (map k, head, sample, row band) items from the per-XCD queues
// [8, 16); the band holds BOTH components of the map's gradient (2 planes x (64 + 2 halo) rows at 128x128), so an
// event's position, taps and weights are computed once for its eight accumulations, and per 16-slot row the workgroup
// reads K1's interval of the plane the events sampled the map at (plane k for earlier passes, k + 1 for later ones, the
// original locations — trajectory plane `nplanes` — for pass k) and loads the row only if it can touch the band.  An event
// visit is two 8-byte loads (vector, position) at 32-bit offsets from scalar bases.
__global__ __launch_bounds__(kSplatThreads, 4) void dflow_splat_kernel(Win w, Events g, const float2 *__restrict__ traj,
                                                                    const float2 *__restrict__ yr,
                                                                    const float2 *__restrict__ cyx,
                                                                    const uint32_t *__restrict__ cmax,
                                                                    float *__restrict__ dflows, int rows_per_band,
                                                                    int nbands, int *__restrict__ queue)
{
    extern __shared__ double lds_img[];
    __shared__ int run_u0[TEF_MAX_PASSES], run_len[TEF_MAX_PASSES], run_src[TEF_MAX_PASSES], run_cum[TEF_MAX_PASSES + 1], s_item;
    __shared__ int hit_ring[(kSplatThreads / 64) * kRing];   // hit rows (row | source plane << 24): one FIFO per wavefront
    const int FB = w.F * w.B, H = w.H, W = w.W, WP = W + kRowPad, M = w.M;
    const int xcd = blockIdx.x & 7;
    const int nitems = w.P * FB;
    const bool iter = (w.kind == TEF_KIND_ITERATIVE);
    const int reach = w.P / max(1, w.mode_div);     // Iterative: pass t feeds map k only if |k - t| < delta_passes[0]
    const int lane = threadIdx.x & 63, wid = threadIdx.x >> 6, nwaves = blockDim.x >> 6;
    for (int k = threadIdx.x; k < (kSplatThreads / 64) * kRing; k += blockDim.x) hit_ring[k] = 0;      // (stale entries are read)
    if (blockIdx.x == 0 && threadIdx.x == 0) queue[kDeferWord - 8] = 0;      // K6's list of deferred wavefronts: consumed
    if (threadIdx.x == 0) s_item = atomicAdd(&queue[xcd], 1);
    __syncthreads();
    for (;;) {
        const int q = __builtin_amdgcn_readfirstlane(s_item);      // wave-uniform: everything derived from it is scalar
        const int it = xcd + 8 * (q / nbands), band = q - (q / nbands) * nbands;
        if (it >= nitems) break;
        const int k = w.korder[it / FB], ib = it % FB, i = ib / w.B, b = ib - i * w.B;
        const int r0 = band * rows_per_band, nrows = min(H, r0 + rows_per_band) - r0;
        // planes: [d/d flow_y | d/d flow_x], each nrows + 2 rows (halo row, the band, halo row) of WP accumulators
        const size_t plane_sz = (size_t)(nrows + 2) * WP;
        double *img_y = lds_img, *img_x = lds_img + plane_sz;
        lds_plane_zero(lds_img, 2 * (nrows + 2) * WP);
        const int t_lo = iter ? max(0, k - reach + 1) : k, t_hi = iter ? min(w.P, k + reach) : k + 1, nruns = t_hi - t_lo;
        __syncthreads();                                  // (everybody has read s_item)
        int next_item = 0;
        if (threadIdx.x == 0) next_item = atomicAdd(&queue[xcd], 1);      // in flight while this item is worked on
        if (threadIdx.x < 64) {       // run = the slots of one pass (multiples of 64: whole rows); rows as a running total
            int carry = 0;
            for (int base_r = 0; base_r < nruns; base_r += 64) {
                const int r = base_r + (int)threadIdx.x, t = t_lo + r;
                int cnt = 0;
                if (r < nruns) {
                    run_u0[r] = w.off[t];
                    run_len[r] = w.off[t + 1] - w.off[t];
                    run_src[r] = t < k ? k : (t > k ? k + 1 : w.nplanes);      // plane the events sampled map k at
                    cnt = run_len[r] >> 4;
                }
                int incl = cnt;
                for (int sft = 1; sft < 64; sft <<= 1) {
                    int up = __shfl_up(incl, sft, 64);
                    if ((int)threadIdx.x >= sft) incl += up;
                }
                if (r < nruns) run_cum[r] = carry + incl - cnt;
                carry += __shfl(incl, 63, 64);
            }
            if (threadIdx.x == 0) run_cum[nruns] = carry;
        }
        __syncthreads();
        const uint32_t mbits = cmax[ib];                // bit pattern of max |c| over this (head, sample), from K6
        const int nev = w.off[t_hi] - w.off[t_lo];
        // (the integer sweep addresses the trajectory planes of a (head, sample) with 32-bit byte offsets)
        const bool fixed = mbits < 0x7f800000u && nev < kFxMaxEvents && (size_t)(w.nplanes + 1) * w.Mt * sizeof(float2) < 0x7fffffffull &&
                           w.nrow < (1 << 24) && w.nplanes < 127;
        const int e = (int)(mbits >> 23) - 127 + 1;     // 2^e > max |c| (a denormal or zero maximum: any small exponent does)
        const float2 *co = cyx + uniform_off(iter ? ((size_t)ib * w.P + k) * M : (size_t)ib * M);
        const float2 *tr = traj + uniform_off((size_t)ib * (w.nplanes + 1) * w.Mt);
        // lookup rows are floor(unnormalize(y)) and the next one; unnormalize(y) is y up to a few ulps: a hundredth of a
        // pixel of slack keeps the test a superset
        const float band_lo = (float)(r0 - 1) - 0.01f, band_hi = (float)(r0 + nrows) + 0.01f;
        if (fixed) {
            const float2 *rows = yr + uniform_off((size_t)ib * (w.nplanes + 1) * w.nrow);
            const int total_rows = run_cum[nruns];
            int *ring = hit_ring + wid * kRing;      // this wavefront's FIFO of hit rows: row | source plane << 24
            int run_hint = 0;
            const uint32_t plane_bytes = (uint32_t)w.Mt * 8u;
            // (rows dealt to the wavefronts in chunks of 16, chunk c to wavefront c % nwaves, as in K2)
            auto load_range = [&](int m, int &entry) -> float2 {
                const int first = (m * 4 * nwaves + wid) * 16;                      // (wave-uniform)
                const int fr = first + (lane >> 4) * (nwaves * 16) + (lane & 15);
                float2 rg = make_float2(__uint_as_float(0xffffffffu), 0.0f);
                entry = 0;
                if (first < total_rows) {
                    while (__builtin_amdgcn_readfirstlane(run_cum[run_hint + 1]) <= first) ++run_hint;
                }
                if (fr < total_rows) {
                    int r = run_hint;
                    while (run_cum[r + 1] <= fr) ++r;
                    const int row = (run_u0[r] >> 4) + (fr - run_cum[r]), src = run_src[r];
                    rg = rows[(size_t)src * w.nrow + row];
                    entry = row | (src << 24);
                }
                return rg;
            };
            constexpr int kQ = 4;
            struct Quad { float cvy, cvx, y, x; };
            const uint32_t lane_off = (uint32_t)(lane & 15) * 8u;
            const int lane_grp = lane >> 4;
            auto load_batch = [&](Quad (&qd)[kQ], int pos) {
#pragma unroll
                for (int kq = 0; kq < kQ; ++kq) {
                    const uint32_t en = (uint32_t)ring[(pos + 4 * kq + lane_grp) & (kRing - 1)];
                    const uint32_t voff = (en & 0xffffffu) * 128u + lane_off;          // slot (16 row + lane) x 8 bytes
                    const float2 cv = *at_bytes(co, voff);
                    const float2 pp = *at_bytes(tr, (en >> 24) * plane_bytes + voff);
                    qd[kq].cvy = cv.x;          // (no select on loaded registers here: see K2's load_batch)
                    qd[kq].cvx = cv.y;
                    qd[kq].y = pp.x;
                    qd[kq].x = pp.y;
                }
            };
            unsigned long long *py = reinterpret_cast<unsigned long long *>(img_y), *px = reinterpret_cast<unsigned long long *>(img_x);
            const AxisConst ach = axis_const(H), acw = axis_const(W);
            const int r0m1 = r0 - 1;
            auto process = [&](const Quad (&qd)[kQ], int rem) {      // rem: ring entries left at the batch's first one
#pragma unroll
                for (int kq = 0; kq < kQ; ++kq) {
                    const float cvy = qd[kq].cvy, cvx = qd[kq].cvx;
                    // same taps as the forward lookup (taps_core), kept as (row, column) to address the padded LDS planes
                    const float fiy = (div_by_const(2.0f * qd[kq].y, ach.rinv) - 1.0f + 1.0f) * ach.half;
                    const float fix = (div_by_const(2.0f * qd[kq].x, acw.rinv) - 1.0f + 1.0f) * acw.half;
                    const float fy0 = floorf(fiy), fx0 = floorf(fix);
                    const int rr = (int)fy0 - r0m1, x0 = (int)fx0;
                    const bool take = (cvy != 0.0f || cvx != 0.0f) && (unsigned)rr <= (unsigned)nrows && lane_grp < rem - 4 * kq;
                    const bool fast = take && (unsigned)x0 < (unsigned)W;
                    if (fast) dflow_fixed(fiy, fix, fy0, fx0, ldexpf(cvy, -e), ldexpf(cvx, -e), rr, WP, py, px);
                    // a sampling location left of / beyond the frame (only an event list with coordinates outside the
                    // frame has them): the tap-by-tap form, as a real call
                    if (__builtin_amdgcn_ballot_w64(take && !fast) != 0ull) {
                        if (take && !fast)
                            dflow_one_fixed_slow(make_float2(qd[kq].y, qd[kq].x), cvy, cvx, H, W, WP, e, img_y + WP, img_x + WP, r0, nrows);
                    }
                }
            };
            // (one stream of hit rows per wavefront and item, the next batch's loads always in flight: see K2)
            int tail = 0, head = 0;
            int next_m = 0;
            float2 rg_a, rg_b;
            int en_a, en_b;
            rg_a = load_range(0, en_a);
            rg_b = load_range(1, en_b);
            auto more = [&]() { return (next_m * 4 * nwaves + wid) * 16 < total_rows; };      // (wave-uniform)
            auto top_up = [&]() {
                while (tail - head < 8 * kQ && more()) {
                    const bool hit = rg_a.y >= band_lo && rg_a.x < band_hi;      // (NaN for an empty / absent row)
                    const unsigned long long mask = __builtin_amdgcn_ballot_w64(hit);
                    if (hit) ring[(tail + __builtin_amdgcn_mbcnt_hi((unsigned)(mask >> 32), __builtin_amdgcn_mbcnt_lo((unsigned)mask, 0))) & (kRing - 1)] = en_a;
                    tail += __builtin_popcountll(mask);
                    rg_a = rg_b;
                    en_a = en_b;
                    ++next_m;
                    rg_b = load_range(next_m + 1, en_b);
                }
                __builtin_amdgcn_wave_barrier();
            };
            Quad qa[kQ], qb[kQ];
            top_up();
            load_batch(qa, head);
            int rem_a = tail - head, rem_b;
            head += 4 * kQ;
            for (;;) {
                top_up();
                load_batch(qb, head);
                rem_b = tail - head;
                head += 4 * kQ;
                process(qa, rem_a);
                if (rem_b <= 0) break;
                top_up();
                load_batch(qa, head);
                rem_a = tail - head;
                head += 4 * kQ;
                process(qb, rem_b);
                if (rem_a <= 0) break;
            }
        } else {
            for (int t = t_lo; t < t_hi; ++t) {
                const int src = t < k ? k : (t > k ? k + 1 : w.nplanes);
                for (int sl = w.off[t] + threadIdx.x; sl < w.off[t + 1]; sl += blockDim.x) {
                    const float cvy = co[sl].x, cvx = co[sl].y;
                    if (cvy == 0.0f && cvx == 0.0f) continue;
                    dflow_one<false>(tr[(size_t)src * w.Mt + sl], cvy, cvx, H, W, WP, 0, img_y + WP, img_x + WP, r0, nrows);
                }
            }
        }
        __syncthreads();
        // dflows [P][F][B][2][H][W]: channel 0 = d/d flow_x, 1 = d/d flow_y
        float *ox = dflows + ((((size_t)k * w.F + i) * w.B + b) * 2) * (size_t)(H * W) + (size_t)r0 * W, *oy = ox + (size_t)H * W;
        const double unscale = __builtin_ldexp(1.0, e - 46);
        if (fixed) {
            dflow_plane_store<true>(img_x + WP, nrows, W, WP, unscale, ox);
            dflow_plane_store<true>(img_y + WP, nrows, W, WP, unscale, oy);
        } else {
            dflow_plane_store<false>(img_x + WP, nrows, W, WP, 1.0, ox);
            dflow_plane_store<false>(img_y + WP, nrows, W, WP, 1.0, oy);
        }
        __syncthreads();                                  // the planes have been read: the next item may clear them
        if (threadIdx.x == 0) s_item = next_item;
        __syncthreads();
    }
}

// K0: AoS -> SoA packing of one pass (Iterative.update / Linear.update bookkeeping, loss/flow.py:457-473),
// with a counting sort of the pass's events by (polarity class, 16x8 pixel tile, pixel row).  The loss is a sum over
// events, so the order inside a pass is free; sorting makes the 64 events of a wavefront spatially
// coherent (their bilinear flow / IWE lookups share cache lines: 2x on the gather-bound kernels) and
// polarity-uniform (a splat workgroup skips the other polarity a wavefront at a time).
// Order inside a (class, tile, row) bucket follows LDS-atomic arrival.
constexpr int kPackThreads = 1024;
constexpr int kPackBatch = 10;          // events per thread in flight
constexpr int kPackSlices = 8;          // workgroups per sample
constexpr int kMaxSortBins = 12288;     // 48 KiB of LDS counters (4 classes x tiles x rows)

// Sort geometry of a pass: tiles of tw x th pixels in row-major order of the frame and, inside a tile, its `sub` pixel
// rows (sub == th, or 1 when the frame has too many tiles for the LDS counters).  A tile is 16 pixels wide — one 128-byte
// line of an interleaved flow map or (A, R) image per pixel row — so events that follow each other in a pass start on
// the same line: the 4 lanes the texture path handles per cycle of a 16-byte gather then ask for one or two lines
// instead of three or four (the chain kernels are bound by the L1's line lookups: DESIGN.md section 9b).
struct SortGeom { int tw, th, ntx, nty, sub, ltw, lth; };      // (tw = 1 << ltw, th = 1 << lth)
__device__ __forceinline__ int sort_key(float y, float x, float mp, float mn, int H, int W, const SortGeom &g)
{
    // pos-only (mask exactly (1, 0)), neg-only ((0, 1)), general (both polarities or other values), collate padding
    int cls = (mp != 0.0f) ? (mn != 0.0f ? 2 : 0) : (mn != 0.0f ? 1 : 3);
    if ((mp != 0.0f && mp != 1.0f) || (mn != 0.0f && mn != 1.0f)) cls = 2;     // general mask values: fp64 splat path
    const int yi = min(max((int)y, 0), H - 1), xi = min(max((int)x, 0), W - 1);
    const int ty = yi >> g.lth, tx = xi >> g.ltw;
    const int row = g.sub > 1 ? yi - ty * g.th : 0;
    return ((cls * g.nty + ty) * g.ntx + tx) * g.sub + row;
}

// kPackSlices workgroups per sample.  Every one of them counts ALL events of the sample into the bins (the list is a
// few hundred KB, read from L2) and, in a second set of counters, the events of the slices before its own; the scan of
// the first gives a bin's first slot, the second the part of the bin that belongs to earlier slices.  So no workgroup
// needs another's result: one launch, no scratch memory, and the scattered stores — about one cache-line access per
// stored word, which is what paces this kernel: 42 us per pass with one workgroup per sample — are spread over eight
// compute units per sample.
// Round 5: the counting sweep reads whole events (ONE 16-byte load each instead of two strided 4-byte ones) with all of a
// thread's loads in flight at once (kPackBatch = 10 covers the reference's 10 000 events per pass in one round trip instead
// of three), and the scan of the bin counters is two DPP wave scans around one barrier instead of a 10-step ladder with 20.
struct PackArgs {
    float *ev;
    const float *pm;
    int N;
    float ts_shift;
    const float *ts_override;
    int pass_idx, slot0, cap, H, W;
    int stage;          // pack_events_fast: slots per staging round
    SortGeom geo;
    float *ts, *y, *x, *mp, *mn;
    uint8_t *bin;
    int *cls;
};

// Lists of up to kPackFastMax events (the reference's 10 000 per pass and sample) — round 5.  What paces a pack workgroup:
// one compute unit streams ~60 GB/s (a 240 KB list: 4-6 us per sweep), and a scattered 4-byte store or a 16-byte gather costs
// the L1 one line look-up per lane.  The slice scheme below has each of eight workgroups count the whole list, then sweep its
// EVENT slice and scatter five words per event (17.7 us per pass); one workgroup doing everything through LDS needs three
// sweeps (21 us); walking the slots and gathering the events 28 us.
// Here the slices are SLOT ranges.  Every workgroup counts the whole list once; the counting atomic returns the event's
// rank inside its bin, so after the scan of the counters each thread knows the slot of its events (bin and rank stay in
// registers).  Workgroup s takes the slots [lo_s, hi_s): the boundaries s N / S moved up to the next bin boundary, the same
// in every workgroup — whole bins, so the ranks (which differ between workgroups: arrival order) never cross a boundary.
// It re-reads just ITS events (exec-masked loads), writes them into an SoA staging area in LDS at slot - lo_s, and copies
// that out with coalesced stores; a range longer than the staging area (skewed lists) takes several rounds.  No workgroup
// needs another's result.  The in-place shift of an event's time stamp is done by the workgroup that stages it (nobody else
// reads that word: the counting sweep looks at the coordinates only).
//   cnt: [nbins] counters, [16] wavefront totals, [2] range boundaries, then five staging arrays of `stage` words
constexpr int kPackFastMax = 16 * kPackThreads;
constexpr size_t kPackFastLds = 144 * 1024;
__device__ __forceinline__ void pack_events_fast(const PackArgs &a, int b, int slice, int nslices, int *cnt)
{
    const SortGeom &geo = a.geo;
    const int N = a.N, H = a.H, W = a.W, stage = a.stage;
    const int ntiles = geo.ntx * geo.nty * geo.sub, nbins = 4 * ntiles;
    int *part = cnt + nbins, *bound = part + 16;
    float *st_ts = reinterpret_cast<float *>(bound + 2), *st_y = st_ts + stage, *st_x = st_y + stage, *st_mp = st_x + stage,
          *st_mn = st_mp + stage;
    const int tid = threadIdx.x;
    float *evw = a.ev + (size_t)b * N * 4;
    const float4 *evb = reinterpret_cast<const float4 *>(evw);
    const float2 *pmb = reinterpret_cast<const float2 *>(a.pm) + (size_t)b * N;
    for (int k = tid; k < nbins; k += kPackThreads) cnt[k] = 0;
    if (tid < 2) bound[tid] = N;
    __syncthreads();
    constexpr int kHalf = 8, kRounds = 2;    // staging sweep: events per thread and load round (the rounds cover kPackFastMax)
    constexpr int kEv = kHalf * kRounds;
    int slot[kEv] = {}, rank[kEv] = {};      // slot: first the bin, after the scan the event's slot
    {   // counting sweep: coordinates and masks only (8 + 8 bytes per event), every load of the thread in flight at once
        float2 yx[kEv], mk[kEv];
#pragma unroll
        for (int j = 0; j < kEv; ++j) {
            if (j * kPackThreads >= N) break;
            const int e = min(tid + j * kPackThreads, N - 1);
            yx[j] = *reinterpret_cast<const float2 *>(reinterpret_cast<const char *>(evb) + (uint32_t)e * 16u + 4u);
            mk[j] = *at_bytes(pmb, (uint32_t)e * 8u);
        }
#pragma unroll
        for (int j = 0; j < kEv; ++j) {
            if (j * kPackThreads >= N) break;
            slot[j] = sort_key(yx[j].x, yx[j].y, mk[j].x, mk[j].y, H, W, geo);
            if (tid + j * kPackThreads < N) rank[j] = atomicAdd(&cnt[slot[j]], 1);
        }
    }
    __syncthreads();
    // exclusive scan of the counters, in place (as in the slice scheme); the slot range of this workgroup
    const int t_lo = (int)((long)slice * N / nslices), t_hi = slice + 1 == nslices ? N : (int)((long)(slice + 1) * N / nslices);
    const int per = (nbins + kPackThreads - 1) / kPackThreads;
    const int lo = min(tid * per, nbins), hi = min(lo + per, nbins);
    int run = 0;
    for (int k = lo; k < hi; ++k) run += cnt[k];
    const int incl = wave_prefix_sum(run);
    if ((tid & 63) == 63) part[tid >> 6] = incl;
    __syncthreads();
    int base = incl - run;
    for (int wv = 0; wv < (tid >> 6); ++wv) base += part[wv];
    // The range boundaries: the first bin start at or past the targets.  Starts never decrease, so exactly one thread has the
    // start of its first bin below a target and the start of the next thread's first bin (or N) at or past it: it writes the
    // boundary with a plain store (an atomicMin by every thread — up to 1024 lanes on ONE LDS word — took 5 us).
    const int first = base;
    int b_lo = N, b_hi = N;
    for (int k = lo; k < hi; ++k) {
        const int c = cnt[k];
        cnt[k] = base;
        if (base >= t_lo) b_lo = min(b_lo, base);
        if (base >= t_hi) b_hi = min(b_hi, base);
        base += c;
    }
    if (lo < hi) {
        const int next = hi == nbins ? N : base;      // start of the next thread's first bin
        if (first < t_lo && next >= t_lo) bound[0] = min(b_lo, next);
        if (first < t_hi && next >= t_hi) bound[1] = min(b_hi, next);
    }
    if (tid == 0) {
        if (t_lo <= 0) bound[0] = 0;
        if (t_hi <= 0) bound[1] = 0;
    }
    __syncthreads();
    if (slice == 0 && tid < 3) a.cls[((size_t)b * TEF_MAX_PASSES + a.pass_idx) * 3 + tid] = cnt[(tid + 1) * ntiles];      // run ends of the three event classes
    const int s_lo = slice == 0 ? 0 : bound[0], s_hi = slice + 1 == nslices ? N : bound[1];
#pragma unroll
    for (int j = 0; j < kEv; ++j)            // (every entry: the staging rounds look at whole load rounds)
        slot[j] = tid + j * kPackThreads < N ? cnt[slot[j]] + rank[j] : -1;
    const bool ts_fixed = a.ts_override != nullptr;
    const float ts_value = ts_fixed ? a.ts_override[0] : 0.0f;
    const size_t o0 = (size_t)b * a.cap + a.slot0;
    for (int s0 = s_lo; s0 < s_hi; s0 += stage) {
        const int n = min(stage, s_hi - s0);
#pragma unroll
        for (int h = 0; h < kRounds; ++h) {
            if (h * kHalf * kPackThreads >= N) break;
            float4 v[kHalf];
            float2 m[kHalf];
            int tv = tid;
            asm volatile("" : "+v"(tv));     // (addresses formed here, per round: hoisted out of the loops they were 96 registers and spills)
#pragma unroll
            for (int j = 0; j < kHalf; ++j) {
                const unsigned q = (unsigned)(slot[h * kHalf + j] - s0);
                if (q < (unsigned)n) {       // this round stages the event: the only read of its time stamp, then the shift
                    const uint32_t e = (uint32_t)(tv + (h * kHalf + j) * kPackThreads);
                    v[j] = *at_bytes(evb, e * 16u);
                    m[j] = *at_bytes(pmb, e * 8u);
                }
            }
#pragma unroll
            for (int j = 0; j < kHalf; ++j) {
                const unsigned q = (unsigned)(slot[h * kHalf + j] - s0);
                if (q < (unsigned)n) {
                    const uint32_t e = (uint32_t)(tv + (h * kHalf + j) * kPackThreads);
                    const float t = v[j].x + a.ts_shift;
                    *at_bytes(evw, e * 16u) = t;                 // in-place shift of the caller's list (:457-458)
                    st_ts[q] = ts_fixed ? ts_value : t;
                    st_y[q] = v[j].y;
                    st_x[q] = v[j].z;
                    st_mp[q] = m[j].x;
                    st_mn[q] = m[j].y;
                }
            }
            asm volatile("" ::: "memory");
        }
        lds_barrier();                       // (LDS only: __syncthreads() would wait for the time-stamp stores to be acknowledged)
        for (int q = tid; q < n; q += kPackThreads) {
            const size_t o = o0 + s0 + q;
            a.ts[o] = st_ts[q];
            a.y[o] = st_y[q];
            a.x[o] = st_x[q];
            a.mp[o] = st_mp[q];
            a.mn[o] = st_mn[q];
            if (b == 0) a.bin[a.slot0 + s0 + q] = (uint8_t)a.pass_idx;
        }
        if (s0 + stage < s_hi) lds_barrier();
    }
    // alignment slots up to the next multiple of 64: empty events
    if (slice + 1 == nslices)
        for (int e = N + tid; e < ((N + 63) & ~63); e += kPackThreads) {
            const size_t o = o0 + e;
            a.ts[o] = a.y[o] = a.x[o] = a.mp[o] = a.mn[o] = 0.0f;
            if (b == 0) a.bin[a.slot0 + e] = (uint8_t)a.pass_idx;
        }
}

__device__ __forceinline__ void pack_events_block(const PackArgs &a, int b, int slice, int nslices, int *cnt)
{
    if (a.stage > 0) {
        pack_events_fast(a, b, slice, nslices, cnt);
        return;
    }
    // cnt: [nbins] all events, [nbins] events of earlier slices, [kPackThreads] scan scratch
    const SortGeom &geo = a.geo;
    const int N = a.N, H = a.H, W = a.W;
    const int ntiles = geo.ntx * geo.nty * geo.sub, nbins = 4 * ntiles;       // (bins per event class)
    int *before = cnt + nbins, *part = before + nbins;
    const int tid = threadIdx.x;
    const int per_slice = (N + nslices - 1) / nslices;
    const int e_lo = min(slice * per_slice, N), e_hi = min(e_lo + per_slice, N);
    const float *evf = a.ev + (size_t)b * N * 4;
    const float4 *evb = reinterpret_cast<const float4 *>(evf);
    const float2 *pmb = reinterpret_cast<const float2 *>(a.pm) + (size_t)b * N;
    for (int k = tid; k < 2 * nbins; k += kPackThreads) cnt[k] = 0;
    __syncthreads();
    // (the time stamps are being shifted in place by the slices' owners while every workgroup reads the list: the counting
    // sweep looks at the coordinates only)
    for (int e0 = tid; e0 < N; e0 += kPackThreads * kPackBatch) {
        float4 v[kPackBatch];
        float2 m[kPackBatch];
#pragma unroll
        for (int j = 0; j < kPackBatch; ++j) {
            const int e = min(e0 + j * kPackThreads, N - 1);
            v[j] = evb[e];
            m[j] = pmb[e];
        }
#pragma unroll
        for (int j = 0; j < kPackBatch; ++j) {
            const int e = e0 + j * kPackThreads;
            if (e >= N) break;
            const int key = sort_key(v[j].y, v[j].z, m[j].x, m[j].y, H, W, geo);
            atomicAdd(&cnt[key], 1);
            if (e < e_lo) atomicAdd(&before[key], 1);
        }
    }
    __syncthreads();
    // exclusive scan of the counters: per-thread run of consecutive bins, the runs' totals scanned per wavefront (DPP),
    // the wavefronts' totals by every thread for itself
    const int per = (nbins + kPackThreads - 1) / kPackThreads;
    int lo = min(tid * per, nbins), hi = min(lo + per, nbins), run = 0;
    for (int k = lo; k < hi; ++k) run += cnt[k];
    const int incl = wave_prefix_sum(run);
    if ((tid & 63) == 63) part[tid >> 6] = incl;
    __syncthreads();
    int base = incl - run;
    for (int wv = 0; wv < (tid >> 6); ++wv) base += part[wv];
    for (int k = lo; k < hi; ++k) {
        int c = cnt[k];
        cnt[k] = base + before[k];          // first slot of this slice's share of the bin
        base += c;
        if (slice == 0 && k + 1 < nbins && (k + 1) % ntiles == 0)
            a.cls[((size_t)b * TEF_MAX_PASSES + a.pass_idx) * 3 + (k + 1) / ntiles - 1] = base;   // run ends of the three
    }                                                                                            // event classes
    __syncthreads();
    const bool ts_fixed = a.ts_override != nullptr;
    const float ts_value = ts_fixed ? a.ts_override[0] : 0.0f;
    for (int e0 = e_lo + tid; e0 < e_hi; e0 += kPackThreads * kPackBatch) {
        float4 v[kPackBatch];
        float2 m[kPackBatch];
#pragma unroll
        for (int j = 0; j < kPackBatch; ++j) {
            const int e = min(e0 + j * kPackThreads, e_hi - 1);
            v[j] = evb[e];
            m[j] = pmb[e];
        }
#pragma unroll
        for (int j = 0; j < kPackBatch; ++j) {
            const int e = e0 + j * kPackThreads;
            if (e >= e_hi) break;
            const float t = v[j].x + a.ts_shift;
            a.ev[((size_t)b * N + e) * 4] = t;                    // in-place shift of the caller's list (:457-458)
            const int pos = atomicAdd(&cnt[sort_key(v[j].y, v[j].z, m[j].x, m[j].y, H, W, geo)], 1);
            const size_t o = (size_t)b * a.cap + a.slot0 + pos;
            a.ts[o] = ts_fixed ? ts_value : t;
            a.y[o] = v[j].y;
            a.x[o] = v[j].z;
            a.mp[o] = m[j].x;
            a.mn[o] = m[j].y;
            if (b == 0) a.bin[a.slot0 + e] = (uint8_t)a.pass_idx;
        }
    }
    // alignment slots up to the next multiple of 64 (a wavefront of the chain kernels belongs to one pass): empty events
    if (slice == 0)
        for (int e = N + tid; e < ((N + 63) & ~63); e += kPackThreads) {
            size_t o = (size_t)b * a.cap + a.slot0 + e;
            a.ts[o] = a.y[o] = a.x[o] = a.mp[o] = a.mn[o] = 0.0f;
            if (b == 0) a.bin[a.slot0 + e] = (uint8_t)a.pass_idx;
        }
}

__global__ __launch_bounds__(kPackThreads) void pack_events_kernel(PackArgs a)
{
    extern __shared__ int cnt[];
    pack_events_block(a, blockIdx.x, blockIdx.y, gridDim.y, cnt);
}

// flow map of one head of one pass: [B,2,H,W] (ch0 = x, ch1 = y; any batch/channel strides, dense rows)
//   -> planar copy [B][2][H*W] (smoothing kernels) and interleaved [B][H*W] float2 (flow_y, flow_x) (lookups)
__global__ __launch_bounds__(256) void pack_flow_kernel(const float *__restrict__ src, long sb, long sc, int B, int HW,
                                                        float *__restrict__ planar, float2 *__restrict__ yx)
{
    int p = blockIdx.x * blockDim.x + threadIdx.x;
    if (p >= HW) return;
    int b = blockIdx.y;
    float fx = src[(size_t)b * sb + p], fy = src[(size_t)b * sb + sc + p];
    planar[((size_t)b * 2) * HW + p] = fx;
    planar[((size_t)b * 2 + 1) * HW + p] = fy;
    yx[(size_t)b * HW + p] = make_float2(fy, fx);
}

// the F heads of one pass in one launch (blockIdx.z = head)
constexpr int kMaxHeads = 16;
struct FlowHeads {
    const float *src[kMaxHeads];
    long sb[kMaxHeads], sc[kMaxHeads];
};
__global__ __launch_bounds__(256) void pack_flows_kernel(FlowHeads hd, int B, int HW, float *__restrict__ planar,
                                                         float2 *__restrict__ yx)
{
    int p = blockIdx.x * blockDim.x + threadIdx.x;
    if (p >= HW) return;
    const int b = blockIdx.y, i = blockIdx.z;
    const float *src = hd.src[i];
    float fx = src[(size_t)b * hd.sb[i] + p], fy = src[(size_t)b * hd.sb[i] + hd.sc[i] + p];
    float *pl = planar + (size_t)i * B * 2 * HW;
    pl[((size_t)b * 2) * HW + p] = fx;
    pl[((size_t)b * 2 + 1) * HW + p] = fy;
    yx[((size_t)i * B + b) * HW + p] = make_float2(fy, fx);
}

// One pass of update() in ONE launch (round 5; three launches before: the flow maps and the two event lists): blockIdx.y
// selects the job — [0, sg) slices of the gradient list, [sg, sg + sd) slices of the detached list, then the flow maps
// (1024 pixels of all F heads per workgroup).
__global__ __launch_bounds__(kPackThreads) void update_pass_kernel(PackArgs ga, int sg, PackArgs da, int sd, FlowHeads hd, int F,
                                                                   int B, int HW, float *__restrict__ planar,
                                                                   float2 *__restrict__ yx)
{
    extern __shared__ int cnt[];
    const int b = blockIdx.x, job = blockIdx.y;
    if (job < sg) {
        pack_events_block(ga, b, job, sg, cnt);
    } else if (job < sg + sd) {
        pack_events_block(da, b, job - sg, sd, cnt);
    } else {
        const int p = (job - sg - sd) * kPackThreads + threadIdx.x;
        if (p < HW)
        for (int i = 0; i < F; ++i) {
            const float *src = hd.src[i];
            const float fx = src[(size_t)b * hd.sb[i] + p], fy = src[(size_t)b * hd.sb[i] + hd.sc[i] + p];
            float *pl = planar + (size_t)i * B * 2 * HW;
            pl[((size_t)b * 2) * HW + p] = fx;
            pl[((size_t)b * 2 + 1) * HW + p] = fy;
            yx[((size_t)i * B + b) * HW + p] = make_float2(fy, fx);
        }
    }
}

// All passes of a window in ONE launch (tef_update_window; blockIdx.z = pass): what a caller that holds the whole window —
// a loss-only caller, a staged benchmark window, a deferred update() — pays ten launches and ten host calls for otherwise.
// The per-pass records travel in the kernel arguments: a fixed header, then `npass` records of a PassRec followed by the F
// (pointer, batch stride, channel stride) triples of the pass's flow maps.
struct PassRec {
    float *ev; const float *pm; const float *tso;
    float *dev; const float *dpm; const float *dtso;
    int N, Nd, slot0, dslot0, pass_idx, sg, sd, stage_g, stage_d, pad_;
};
constexpr int kWindowRecBytes = 3328;
struct WindowArgs {
    int npass, F, B, H, W, rec_bytes;
    SortGeom geo;
    Events g, d;
    float *planar;
    float2 *yx;
    unsigned char recs[kWindowRecBytes] __attribute__((aligned(8)));
};
__global__ __launch_bounds__(kPackThreads) void update_window_kernel(WindowArgs wa)
{
    extern __shared__ int cnt[];
    const int b = blockIdx.x, job = blockIdx.y, z = blockIdx.z;
    const unsigned char *rp = wa.recs + (size_t)z * wa.rec_bytes;
    const PassRec &r = *reinterpret_cast<const PassRec *>(rp);
    const int HW = wa.H * wa.W;
    auto args = [&](bool det) {
        const Events &E = det ? wa.d : wa.g;
        PackArgs a;
        a.ev = det ? r.dev : r.ev; a.pm = det ? r.dpm : r.pm; a.N = det ? r.Nd : r.N; a.ts_shift = (float)r.pass_idx;
        a.ts_override = det ? r.dtso : r.tso; a.pass_idx = r.pass_idx; a.slot0 = det ? r.dslot0 : r.slot0; a.cap = E.cap;
        a.H = wa.H; a.W = wa.W; a.stage = det ? r.stage_d : r.stage_g; a.geo = wa.geo;
        a.ts = const_cast<float *>(E.ts); a.y = const_cast<float *>(E.y); a.x = const_cast<float *>(E.x);
        a.mp = const_cast<float *>(E.mp); a.mn = const_cast<float *>(E.mn); a.bin = const_cast<uint8_t *>(E.bin);
        a.cls = const_cast<int *>(E.cls);
        return a;
    };
    if (job < r.sg) {
        pack_events_block(args(false), b, job, r.sg, cnt);
    } else if (job < r.sg + r.sd) {
        pack_events_block(args(true), b, job - r.sg, r.sd, cnt);
    } else {
        const int p = (job - r.sg - r.sd) * kPackThreads + threadIdx.x;
        if (p >= HW) return;
        const unsigned char *fp = rp + sizeof(PassRec);
        float *planar = wa.planar + (size_t)r.pass_idx * wa.F * wa.B * 2 * HW;
        float2 *yx = wa.yx + (size_t)r.pass_idx * wa.F * wa.B * HW;
        for (int i = 0; i < wa.F; ++i) {
            const float *src = *reinterpret_cast<const float *const *>(fp + (size_t)i * 24);
            const long sb = *reinterpret_cast<const long *>(fp + (size_t)i * 24 + 8), sc = *reinterpret_cast<const long *>(fp + (size_t)i * 24 + 16);
            const float fx = src[(size_t)b * sb + p], fy = src[(size_t)b * sb + sc + p];
            float *pl = planar + (size_t)i * wa.B * 2 * HW;
            pl[((size_t)b * 2) * HW + p] = fx;
            pl[((size_t)b * 2 + 1) * HW + p] = fy;
            yx[((size_t)i * wa.B + b) * HW + p] = make_float2(fy, fx);
        }
    }
}

// ---------------------------------------------------------------------------------------------
// Host side
// ---------------------------------------------------------------------------------------------
struct Layout {
    size_t traj, meta, yr, ar, nz, counts, sq, stats, parts, queue, cmax, wmax, defer, cyx, bad, total;
    int nz_cap;      // 64-bit words of "C != 0" bits per (image, head, sample, polarity, band)
};

inline size_t align_up(size_t v) { return (v + 255) & ~(size_t)255; }

bool make_win(const tef_loss_cfg *c, Win *w)
{
    if (!c) return tef::fail("null config");
    if (c->kind != TEF_KIND_ITERATIVE && c->kind != TEF_KIND_LINEAR) return tef::fail("unknown loss kind");
    if (c->B < 1 || c->H < 2 || c->W < 2 || c->F < 1) return tef::fail("bad B/H/W/F");
    if (c->P < 1 || c->P > TEF_MAX_PASSES) return tef::fail("passes_loss out of range [1, 64]");
    if (c->S < 1 || c->S > TEF_MAX_SCALES) return tef::fail("scales_loss out of range [1, 6]");
    if (2 * 3 * (size_t)(c->W + kRowPad) * sizeof(double) > kSplat2LdsBudget) return tef::fail("image row does not fit the LDS band");
    if (c->kind == TEF_KIND_ITERATIVE) {
        // iterative_mode "four" raises TypeError in the reference itself (loss/flow.py:666-692); only one/two exist here
        if (c->mode_div != 1 && c->mode_div != 2) return tef::fail("iterative_mode must be 'one' or 'two'");
        if (c->P < 2) return tef::fail("Iterative needs passes_loss >= 2 (the reference fails in torch.cat for 1)");
    }
    memset(w, 0, sizeof(*w));
    w->kind = c->kind; w->B = c->B; w->H = c->H; w->W = c->W; w->P = c->P; w->F = c->F; w->S = c->S;
    w->mode_div = c->mode_div; w->M = c->M; w->Md = c->Md; w->Mt = c->M + c->Md;
    w->scaling = c->loss_scaling ? 1 : 0;
    w->comp = c->border_compensation ? 1 : 0;
    w->nrow = (w->Mt + 15) / 16;
    w->nplanes = (c->kind == TEF_KIND_ITERATIVE) ? c->P + 1 : 2 * c->S;
    if (c->M < 0 || c->Md < 0 || c->off[0] != 0 || c->doff[0] != 0 || c->off[c->P] != c->M || c->doff[c->P] != c->Md)
        return tef::fail("inconsistent slot offsets");
    for (int t = 0; t <= c->P; ++t) {
        w->off[t] = c->off[t];
        w->doff[t] = c->doff[t];
        if (t && (c->off[t] < c->off[t - 1] || c->doff[t] < c->doff[t - 1])) return tef::fail("offsets not monotone");
        if ((c->off[t] | c->doff[t]) & 63) return tef::fail("pass offsets must be multiples of 64 slots (tef_pack_events pads)");
    }
    int n = 0;
    for (int s = 0; s < c->S; ++s) {
        int scale = c->P >> s;
        if (scale < 1) return tef::fail("passes_loss // 2**scale is zero");
        if (c->kind == TEF_KIND_ITERATIVE && scale / c->mode_div < 1)
            return tef::fail("delta_passes is zero for a temporal scale (the reference divides by it)");
        w->img_base[s] = n;
        n += images_of_scale(c->kind, c->P, s);
    }
    w->img_base[c->S] = n;
    w->nimg = n;
    if (n > kMaxImages) return tef::fail("too many images");
    // longest-processing-time-first order of the splat workgroups: images with the most events first
    int idx[kMaxImages];
    long work[kMaxImages];
    for (int j = 0; j < n; ++j) {
        Img im = decode_image(*w, j);
        idx[j] = j;
        work[j] = (long)(w->off[im.he] - w->off[im.le]) + (long)(w->doff[im.he] - w->doff[im.le]);
    }
    std::stable_sort(idx, idx + n, [&](int a, int b) { return work[a] > work[b]; });
    for (int j = 0; j < n; ++j) w->order[j] = (uint16_t)idx[j];
    {   // the same for the flow-gradient maps: map k is fed by the passes within the reach
        const int reach = (c->kind == TEF_KIND_ITERATIVE) ? c->P / c->mode_div : 1;
        int kidx[TEF_MAX_PASSES];
        long kwork[TEF_MAX_PASSES];
        for (int k = 0; k < c->P; ++k) {
            const int lo = (c->kind == TEF_KIND_ITERATIVE) ? std::max(0, k - reach + 1) : k;
            const int hi = (c->kind == TEF_KIND_ITERATIVE) ? std::min(c->P, k + reach) : k + 1;
            kidx[k] = k;
            kwork[k] = w->off[hi] - w->off[lo];
        }
        std::stable_sort(kidx, kidx + c->P, [&](int a, int b) { return kwork[a] > kwork[b]; });
        for (int k = 0; k < c->P; ++k) w->korder[k] = (uint8_t)kidx[k];
    }
    return true;
}

// halo: extra rows above + below the band that a plane carries (K2: 2, its unconditional corner accumulations)
inline void band_geometry(const Win &w, int planes, int *rows_per_band, int *nbands, size_t *lds, size_t budget = kSplatLdsBudget,
                          int halo = 0)
{
    int rows = (int)(budget / ((size_t)planes * (w.W + kRowPad) * sizeof(double))) - halo;
    if (rows > w.H) rows = w.H;
    if (rows < 1) rows = 1;                          // (make_win has checked that a band of one row fits)
    *nbands = (w.H + rows - 1) / rows;
    rows = (w.H + *nbands - 1) / *nbands;            // equal bands
    *rows_per_band = rows;
    *lds = (size_t)planes * (rows + halo) * (w.W + kRowPad) * sizeof(double);
}

Layout make_layout(const Win &w)
{
    Layout L;
    const size_t FB = (size_t)w.F * w.B, HW = (size_t)w.H * w.W;
    const size_t img = (size_t)w.nimg * FB * 2 * HW;
    const size_t nc = FB * (size_t)(w.kind == TEF_KIND_ITERATIVE ? w.P : 1) * (size_t)w.M;
    size_t o = 0;
    L.traj = o;   o += align_up(FB * (size_t)(w.nplanes + 1) * (size_t)w.Mt * sizeof(float2));      // + the original locations
    L.meta = o;   o += align_up(FB * (size_t)w.Mt * sizeof(uint2));
    L.yr = o;     o += align_up(FB * (size_t)(w.nplanes + 1) * (size_t)w.nrow * sizeof(float2));
    L.ar = o;     o += align_up(img * sizeof(float2));
    L.stats = o;  o += align_up((size_t)w.nimg * FB * 2 * sizeof(float));
    {
        int rows, nbands;
        size_t lds;
        band_geometry(w, 2, &rows, &nbands, &lds, kSplat2LdsBudget, 2);
        L.nz_cap = nz_words(rows, w.W, w.W + kRowPad, kSplat2Threads);
        L.nz = o;     o += align_up((size_t)w.nimg * FB * 2 * nbands * (size_t)L.nz_cap * sizeof(unsigned long long));
        L.parts = o;  o += align_up((size_t)w.nimg * FB * nbands * 2 * kSplat2Waves * sizeof(double));
    }
    L.counts = o; o += align_up((size_t)w.nimg * FB * sizeof(double));
    L.sq = o;     o += align_up((size_t)w.nimg * FB * sizeof(double));
    L.queue = o;  o += align_up((kQueueInts + FB) * sizeof(int));      // + one magnitude word per (head, sample) for K6 -> K7
    L.cmax = L.queue + kQueueInts * sizeof(int);
    L.wmax = o;   o += align_up(FB * (size_t)(w.M / 64 + 1) * sizeof(uint32_t));
    L.defer = o;  o += align_up(FB * (size_t)(w.M / 64 + 1) * sizeof(int));      // K6: wavefronts left to its second launch
    L.cyx = o;    o += align_up(nc * sizeof(float2));
    L.bad = o;    o += align_up((FB * (size_t)((w.Mt + 255) / 256 + 1) + (size_t)w.nimg * FB) * sizeof(int));   // one word per K1 workgroup, then one per image
    L.total = o;
    return L;
}

inline Events to_events(const tef_events *e)
{
    Events r;
    if (e) { r.ts = e->ts; r.y = e->y; r.x = e->x; r.mp = e->mp; r.mn = e->mn; r.bin = e->bin; r.cls = e->cls; r.cap = e->cap; }
    else { r.ts = r.y = r.x = r.mp = r.mn = nullptr; r.bin = nullptr; r.cls = nullptr; r.cap = 0; }
    return r;
}

// launch with the kernel's own start / stop timestamps when per-kernel timing is on (tef_profile_enable)
#define TEF_LAUNCH_TIMED(slot, kernel, grid, block, lds, st, ...)                                   \
    do {                                                                                            \
        hipEvent_t ev_a_, ev_b_;                                                                    \
        tef::prof_events(slot, &ev_a_, &ev_b_);                                                     \
        hipExtLaunchKernelGGL(kernel, grid, block, lds, st, ev_a_, ev_b_, 0, __VA_ARGS__);          \
    } while (0)

inline int num_cus()
{
    static const int n = [] {
        int dev = 0, v = 0;
        if (hipGetDevice(&dev) != hipSuccess || hipDeviceGetAttribute(&v, hipDeviceAttributeMultiprocessorCount, dev) != hipSuccess || v < 8) v = 256;
        return v;
    }();
    return n;
}

// opt in to > 64 KiB dynamic LDS for the two LDS-resident splat kernels: once per process (thread-safe static init)
bool ensure_attrs()
{
    static const hipError_t e1a = hipFuncSetAttribute((const void *)splat_stats_kernel<0>,
                                                      hipFuncAttributeMaxDynamicSharedMemorySize, (int)kSplat2LdsBudget);
    static const hipError_t e1 = e1a != hipSuccess ? e1a : hipFuncSetAttribute((const void *)splat_stats_kernel<1>,
                                                     hipFuncAttributeMaxDynamicSharedMemorySize, (int)kSplat2LdsBudget);
    static const hipError_t e2 = hipFuncSetAttribute((const void *)dflow_splat_kernel,
                                                     hipFuncAttributeMaxDynamicSharedMemorySize, (int)kSplatLdsBudget);
    static const hipError_t e3a = hipFuncSetAttribute((const void *)pack_events_kernel,
                                                      hipFuncAttributeMaxDynamicSharedMemorySize,
                                                      (int)std::max((2 * kMaxSortBins + kPackThreads) * sizeof(int), kPackFastLds));
    static const hipError_t e3 = e3a != hipSuccess ? e3a : hipFuncSetAttribute((const void *)update_pass_kernel,
                                                     hipFuncAttributeMaxDynamicSharedMemorySize,
                                                     (int)std::max((2 * kMaxSortBins + kPackThreads) * sizeof(int), kPackFastLds));
    static const hipError_t e3b = hipFuncSetAttribute((const void *)update_window_kernel,
                                                      hipFuncAttributeMaxDynamicSharedMemorySize,
                                                      (int)std::max((2 * kMaxSortBins + kPackThreads) * sizeof(int), kPackFastLds));
    if (e3 != hipSuccess || e3b != hipSuccess) return tef::fail_hip("hipFuncSetAttribute", e3 != hipSuccess ? e3 : e3b);
    if (e1 != hipSuccess || e2 != hipSuccess) return tef::fail_hip("hipFuncSetAttribute", e1 != hipSuccess ? e1 : e2);
    return true;
}

}  // namespace

extern "C" {

size_t tef_loss_workspace_bytes(const tef_loss_cfg *cfg)
{
    Win w;
    if (!make_win(cfg, &w)) return 0;
    return make_layout(w).total;
}

}  // extern "C"

namespace {
// argument block, LDS bytes and workgroups per sample of one pack job (N > 0)
bool pack_job(float *ev, const float *pm, int B, int N, float ts_shift, const float *ts_override, int pass_idx, int slot0, int cap,
              int H, int W, float *ts, float *y, float *x, float *mp, float *mn, uint8_t *bin, int *cls, PackArgs *a,
              size_t *lds, int *slices)
{
    if (B < 1 || N < 0 || slot0 < 0 || (slot0 & 63) || slot0 + ((N + 63) & ~63) > cap || pass_idx < 0 ||
        pass_idx >= TEF_MAX_PASSES || H < 1 || W < 1)
        return tef::fail("tef_pack_events: bad sizes (slot0 must be a multiple of 64, cap must hold N rounded up to 64)");
    SortGeom geo;
    geo.tw = 16;
    geo.th = 8;
    for (;;) {
        geo.ntx = (W + geo.tw - 1) / geo.tw;
        geo.nty = (H + geo.th - 1) / geo.th;
        const int tiles4 = 4 * geo.ntx * geo.nty;
        geo.sub = (tiles4 * geo.th <= kMaxSortBins) ? geo.th : 1;
        if (tiles4 * geo.sub <= kMaxSortBins) break;
        if (geo.tw <= geo.th) geo.tw *= 2; else geo.th *= 2;
    }
    geo.ltw = __builtin_ctz(geo.tw);
    geo.lth = __builtin_ctz(geo.th);
    int nbins = 4 * geo.ntx * geo.nty * geo.sub;
    *slices = N > kPackFastMax ? kPackSlices : 1;
    *lds = (size_t)(2 * nbins + kPackThreads) * sizeof(int);
    a->stage = 0;
    if (N <= kPackFastMax) {
        // slot-range slices (pack_events_fast); the staging area holds twice the even share of a workgroup, whatever is over
        // that (skewed lists) takes more rounds
        *slices = N >= 2 * kPackThreads ? kPackSlices : 1;
        const size_t head = (size_t)(nbins + 16 + 2) * sizeof(int);
        const int room = (int)((kPackFastLds - head) / (5 * sizeof(float))) & ~63;      // slots the staging area can hold
        a->stage = std::min(room, (2 * ((N + *slices - 1) / *slices) + 63) & ~63);
        // (at least 84 KiB: ONE workgroup per compute unit.  With less, two of a launch's 64 + 128 workgroups were placed on one
        // compute unit while others stayed empty, and every phase of both took twice as long)
        *lds = std::max(head + (size_t)5 * a->stage * sizeof(float), (size_t)84 * 1024);
    }
    a->ev = ev; a->pm = pm; a->N = N; a->ts_shift = ts_shift; a->ts_override = ts_override; a->pass_idx = pass_idx;
    a->slot0 = slot0; a->cap = cap; a->H = H; a->W = W; a->geo = geo;
    a->ts = ts; a->y = y; a->x = x; a->mp = mp; a->mn = mn; a->bin = bin; a->cls = cls;
    return true;
}
}  // namespace

extern "C" {

int tef_pack_events(float *ev, const float *pm, int B, int N, float ts_shift, const float *ts_override, int pass_idx,
                    int slot0, int cap, int H, int W, float *ts, float *y, float *x, float *mp, float *mn,
                    uint8_t *bin, int *cls, void *stream)
{
    PackArgs a;
    size_t lds;
    int slices;
    if (!pack_job(ev, pm, B, N, ts_shift, ts_override, pass_idx, slot0, cap, H, W, ts, y, x, mp, mn, bin, cls, &a, &lds, &slices))
        return TEF_ERR_INVALID;
    if (N == 0) return 0;
    if (!ensure_attrs()) return TEF_ERR_LAUNCH;
    hipStream_t st = (hipStream_t)stream;
    {
        tef::ProfScope ps(tef::PROF_PACK, st);
        hipLaunchKernelGGL(pack_events_kernel, dim3(B, slices), dim3(kPackThreads), lds, st, a);
    }
    return tef::check_launch("pack_events_kernel");
}

int tef_pack_flow(const float *flow, long stride_b, long stride_c, int B, int H, int W, float *planar, float *yx,
                  void *stream)
{
    if (!flow || !planar || !yx || B < 1 || H < 1 || W < 1) return tef::fail("tef_pack_flow: bad arguments"), TEF_ERR_INVALID;
    hipStream_t st = (hipStream_t)stream;
    dim3 grid((H * W + 255) / 256, B);
    {
        tef::ProfScope ps(tef::PROF_PACK, st);
        hipLaunchKernelGGL(pack_flow_kernel, grid, dim3(256), 0, st, flow, stride_b, stride_c, B, H * W, planar,
                           (float2 *)yx);
    }
    return tef::check_launch("pack_flow_kernel");
}

int tef_pack_flows(const float *const *flows, const long *stride_b, const long *stride_c, int F, int B, int H, int W,
                   float *planar, float *yx, void *stream)
{
    if (!flows || !stride_b || !stride_c || !planar || !yx || F < 1 || F > kMaxHeads || B < 1 || H < 1 || W < 1)
        return tef::fail("tef_pack_flows: bad arguments (1..16 heads)"), TEF_ERR_INVALID;
    FlowHeads hd{};
    for (int i = 0; i < F; ++i) {
        if (!flows[i]) return tef::fail("tef_pack_flows: null flow map"), TEF_ERR_INVALID;
        hd.src[i] = flows[i]; hd.sb[i] = stride_b[i]; hd.sc[i] = stride_c[i];
    }
    hipStream_t st = (hipStream_t)stream;
    dim3 grid((H * W + 255) / 256, B, F);
    {
        tef::ProfScope ps(tef::PROF_PACK, st);
        hipLaunchKernelGGL(pack_flows_kernel, grid, dim3(256), 0, st, hd, B, H * W, planar, (float2 *)yx);
    }
    return tef::check_launch("pack_flows_kernel");
}

int tef_update_pass(const float *const *flows, const long *stride_b, const long *stride_c, int F, int B, int H, int W,
                    float *planar, float *yx, float *ev, const float *pm, int N, const float *ts_override, float *dev,
                    const float *dpm, int Nd, const float *dts_override, int pass_idx, int slot0, int dslot0,
                    const tef_events *grad, const tef_events *det, void *stream)
{
    if (!grad || !det) return tef::fail("tef_update_pass: null event store"), TEF_ERR_INVALID;
    if (flows && stride_b && stride_c && planar && yx && F >= 1 && F <= kMaxHeads && B >= 1 && H >= 1 && W >= 1 && (N > 0 || Nd > 0)) {
        // the whole pass in ONE launch: [gradient list slices | detached list slices | flow-map blocks] per sample
        PackArgs ga{}, da{};
        size_t lg = 0, ld = 0;
        int sg = 0, sd = 0;
        if (N > 0 && !pack_job(ev, pm, B, N, (float)pass_idx, ts_override, pass_idx, slot0, grad->cap, H, W, (float *)grad->ts,
                               (float *)grad->y, (float *)grad->x, (float *)grad->mp, (float *)grad->mn, (uint8_t *)grad->bin,
                               (int *)grad->cls, &ga, &lg, &sg))
            return TEF_ERR_INVALID;
        if (Nd > 0 && !pack_job(dev, dpm, B, Nd, (float)pass_idx, dts_override, pass_idx, dslot0, det->cap, H, W, (float *)det->ts,
                                (float *)det->y, (float *)det->x, (float *)det->mp, (float *)det->mn, (uint8_t *)det->bin,
                                (int *)det->cls, &da, &ld, &sd))
            return TEF_ERR_INVALID;
        FlowHeads hd{};
        for (int i = 0; i < F; ++i) {
            if (!flows[i]) return tef::fail("tef_update_pass: null flow map"), TEF_ERR_INVALID;
            hd.src[i] = flows[i]; hd.sb[i] = stride_b[i]; hd.sc[i] = stride_c[i];
        }
        if (!ensure_attrs()) return TEF_ERR_LAUNCH;
        const int HW = H * W, sf = (HW + kPackThreads - 1) / kPackThreads;
        hipStream_t st = (hipStream_t)stream;
        {
            tef::ProfScope ps(tef::PROF_PACK, st);
            hipLaunchKernelGGL(update_pass_kernel, dim3(B, sg + sd + sf), dim3(kPackThreads), std::max(lg, ld), st, ga, sg, da, sd, hd,
                               F, B, HW, planar, (float2 *)yx);
        }
        return tef::check_launch("update_pass_kernel");
    }
    if (int rc = tef_pack_flows(flows, stride_b, stride_c, F, B, H, W, planar, yx, stream)) return rc;
    if (N > 0) {
        if (int rc = tef_pack_events(ev, pm, B, N, (float)pass_idx, ts_override, pass_idx, slot0, grad->cap, H, W, (float *)grad->ts,
                                     (float *)grad->y, (float *)grad->x, (float *)grad->mp, (float *)grad->mn, (uint8_t *)grad->bin,
                                     (int *)grad->cls, stream))
            return rc;
    }
    if (Nd > 0) {
        if (int rc = tef_pack_events(dev, dpm, B, Nd, (float)pass_idx, dts_override, pass_idx, dslot0, det->cap, H, W, (float *)det->ts,
                                     (float *)det->y, (float *)det->x, (float *)det->mp, (float *)det->mn, (uint8_t *)det->bin,
                                     (int *)det->cls, stream))
            return rc;
    }
    return 0;
}

int tef_update_window(const tef_update_desc *passes, int npass, int F, int B, int H, int W, float *planar, float *yx,
                      const tef_events *grad, const tef_events *det, void *stream)
{
    if (!passes || npass < 1 || !grad || !det || !planar || !yx || F < 1 || F > kMaxHeads || B < 1 || H < 1 || W < 1)
        return tef::fail("tef_update_window: bad arguments (1..16 heads)"), TEF_ERR_INVALID;
    if (!ensure_attrs()) return TEF_ERR_LAUNCH;
    const int rec_bytes = (int)sizeof(PassRec) + 24 * F;
    const int per_launch = kWindowRecBytes / rec_bytes;
    const int HW = H * W, sf = (HW + kPackThreads - 1) / kPackThreads;
    hipStream_t st = (hipStream_t)stream;
    for (int p0 = 0; p0 < npass; p0 += per_launch) {
        WindowArgs wa{};
        wa.npass = std::min(per_launch, npass - p0); wa.F = F; wa.B = B; wa.H = H; wa.W = W; wa.rec_bytes = rec_bytes;
        wa.g = to_events(grad); wa.d = to_events(det);
        wa.planar = planar; wa.yx = (float2 *)yx;
        size_t lds = 0;
        int jobs = 0;
        for (int q = 0; q < wa.npass; ++q) {
            const tef_update_desc &u = passes[p0 + q];
            if (!u.flows || !u.stride_b || !u.stride_c || u.N < 0 || u.Nd < 0 || (u.N > 0 && (!u.ev || !u.pm)) || (u.Nd > 0 && (!u.dev || !u.dpm)))
                return tef::fail("tef_update_window: null pointer in a pass record"), TEF_ERR_INVALID;
            PassRec r{};
            PackArgs ga{}, da{};
            size_t lg = 0, ld = 0;
            int sg = 0, sd = 0;
            if (u.N > 0 && !pack_job(u.ev, u.pm, B, u.N, (float)u.pass_idx, u.ts_override, u.pass_idx, u.slot0, grad->cap, H, W,
                                     (float *)grad->ts, (float *)grad->y, (float *)grad->x, (float *)grad->mp, (float *)grad->mn,
                                     (uint8_t *)grad->bin, (int *)grad->cls, &ga, &lg, &sg))
                return TEF_ERR_INVALID;
            if (u.Nd > 0 && !pack_job(u.dev, u.dpm, B, u.Nd, (float)u.pass_idx, u.dts_override, u.pass_idx, u.dslot0, det->cap, H, W,
                                      (float *)det->ts, (float *)det->y, (float *)det->x, (float *)det->mp, (float *)det->mn,
                                      (uint8_t *)det->bin, (int *)det->cls, &da, &ld, &sd))
                return TEF_ERR_INVALID;
            if (u.pass_idx < 0 || u.pass_idx >= TEF_MAX_PASSES) return tef::fail("tef_update_window: pass index out of range"), TEF_ERR_INVALID;
            if (u.N > 0) wa.geo = ga.geo; else if (u.Nd > 0) wa.geo = da.geo;      // (a function of H and W only)
            r.ev = u.ev; r.pm = u.pm; r.tso = u.ts_override; r.dev = u.dev; r.dpm = u.dpm; r.dtso = u.dts_override;
            r.N = u.N; r.Nd = u.Nd; r.slot0 = u.slot0; r.dslot0 = u.dslot0; r.pass_idx = u.pass_idx; r.sg = sg; r.sd = sd;
            r.stage_g = ga.stage; r.stage_d = da.stage;
            unsigned char *rp = wa.recs + (size_t)q * rec_bytes;
            memcpy(rp, &r, sizeof(r));
            for (int i = 0; i < F; ++i) {
                if (!u.flows[i]) return tef::fail("tef_update_window: null flow map"), TEF_ERR_INVALID;
                memcpy(rp + sizeof(PassRec) + (size_t)i * 24, &u.flows[i], 8);
                memcpy(rp + sizeof(PassRec) + (size_t)i * 24 + 8, &u.stride_b[i], 8);
                memcpy(rp + sizeof(PassRec) + (size_t)i * 24 + 16, &u.stride_c[i], 8);
            }
            lds = std::max(lds, std::max(lg, ld));
            jobs = std::max(jobs, sg + sd + sf);
        }
        {
            tef::ProfScope ps(tef::PROF_PACK, st);
            hipLaunchKernelGGL(update_window_kernel, dim3(B, jobs, wa.npass), dim3(kPackThreads), lds, st, wa);
        }
        if (int rc = tef::check_launch("update_window_kernel")) return rc;
    }
    return 0;
}

int tef_loss_forward(const tef_loss_cfg *cfg, const float *flows_yx, const tef_events *grad, const tef_events *det,
                     void *workspace, size_t workspace_bytes, float *loss_out, void *stream)
{
    Win w;
    if (!make_win(cfg, &w)) return TEF_ERR_INVALID;
    if (!flows_yx || !grad || !workspace || !loss_out) return tef::fail("null pointer"), TEF_ERR_INVALID;
    if (w.Md > 0 && !det) return tef::fail("detached events missing"), TEF_ERR_INVALID;
    if (grad->cap < w.M || (det && w.Md > 0 && det->cap < w.Md)) return tef::fail("event capacity < slots"), TEF_ERR_INVALID;
    Layout L = make_layout(w);
    if (workspace_bytes < L.total) return tef::fail("workspace too small"), TEF_ERR_WORKSPACE;
    if (!ensure_attrs()) return TEF_ERR_LAUNCH;
    hipStream_t st = (hipStream_t)stream;
    char *ws = (char *)workspace;
    float2 *traj = (float2 *)(ws + L.traj);
    uint2 *meta = (uint2 *)(ws + L.meta);
    float2 *yr = (float2 *)(ws + L.yr);
    int *queue = (int *)(ws + L.queue);
    int *bad = (int *)(ws + L.bad);
    float2 *ar = (float2 *)(ws + L.ar);
    float *stats = (float *)(ws + L.stats);
    const float2 *fl = (const float2 *)flows_yx;
    Events g = to_events(grad), d = to_events(w.Md > 0 ? det : nullptr);
    const int FB = w.F * w.B;
    const int nbad = w.Mt > 0 ? FB * ((w.Mt + 255) / 256) : 0;

    if (w.Mt > 0) {
        int chunks = (w.Mt + 255) / 256;
        dim3 grid(xcd_grid(FB, chunks));
        if (w.kind == TEF_KIND_ITERATIVE)
            TEF_LAUNCH_TIMED(tef::PROF_WARP, iter_warp_kernel, grid, dim3(256), 0, st, w, fl, g, d, traj, meta, yr, queue, bad, chunks);
        else
            TEF_LAUNCH_TIMED(tef::PROF_WARP, linear_warp_kernel, grid, dim3(256), 0, st, w, fl, g, d, traj, meta, yr, queue, bad, chunks);
    }
    if (int rc = tef::check_launch("warp_kernel")) return rc;
    int rows, nbands;
    size_t lds;
    band_geometry(w, 2, &rows, &nbands, &lds, kSplat2LdsBudget, 2);
    if (w.Mt == 0 && hipMemsetAsync(queue, 0, (kQueueInts + FB) * sizeof(int), st) != hipSuccess)      // (K1 clears them otherwise)
        return tef::fail("hipMemsetAsync(queue)"), TEF_ERR_LAUNCH;
    unsigned long long *nz = (unsigned long long *)(ws + L.nz);
    double *counts = (double *)(ws + L.counts);
    {
        // persistent workgroups, two per CU (each holds the two planes of one polarity of a 32-row band at 128 x 128), a
        // multiple of 8 so that every XCD queue is served
        const long items = (long)w.nimg * FB * nbands * 2;
        unsigned grid = (unsigned)std::min<long>(items, 2 * num_cus());
        grid = std::max(8u, (grid + 7u) & ~7u);
        if (w.comp || w.kind != TEF_KIND_ITERATIVE)
            TEF_LAUNCH_TIMED(tef::PROF_SPLAT, splat_stats_kernel<0>, dim3(grid), dim3(kSplat2Threads), lds, st, w, g, d, traj, meta,
                             yr, ar, nz, L.nz_cap, (double *)(ws + L.parts), rows, nbands, queue);
        else
            TEF_LAUNCH_TIMED(tef::PROF_SPLAT, splat_stats_kernel<1>, dim3(grid), dim3(kSplat2Threads), lds, st, w, g, d, traj, meta,
                             yr, ar, nz, L.nz_cap, (double *)(ws + L.parts), rows, nbands, queue);
    }
    if (int rc = tef::check_launch("splat_stats_kernel")) return rc;
    int *bad_img = bad + FB * ((w.Mt + 255) / 256 + 1);
    TEF_LAUNCH_TIMED(tef::PROF_COUNT, image_count_kernel, dim3((unsigned)(w.nimg * FB)), dim3(256), 0, st, nz, L.nz_cap, nbands,
                     rows, w.H, w.W, (const double *)(ws + L.parts), counts, (double *)(ws + L.sq), bad, nbad, bad_img);
    if (int rc = tef::check_launch("image_count_kernel")) return rc;
    TEF_LAUNCH_TIMED(tef::PROF_REDUCE, loss_reduce_kernel, dim3(1), dim3(256), 0, st, w, (const double *)(ws + L.sq), counts, stats,
                     bad_img, loss_out);
    return tef::check_launch("loss_reduce_kernel");
}

int tef_loss_backward(const tef_loss_cfg *cfg, const float *flows_yx, const tef_events *grad, const tef_events *det,
                      void *workspace, size_t workspace_bytes, const float *grad_out, float *dflows, void *stream)
{
    (void)det;
    Win w;
    if (!make_win(cfg, &w)) return TEF_ERR_INVALID;
    if (!flows_yx || !grad || !workspace || !grad_out || !dflows) return tef::fail("null pointer"), TEF_ERR_INVALID;
    Layout L = make_layout(w);
    if (workspace_bytes < L.total) return tef::fail("workspace too small"), TEF_ERR_WORKSPACE;
    if (!ensure_attrs()) return TEF_ERR_LAUNCH;
    hipStream_t st = (hipStream_t)stream;
    char *ws = (char *)workspace;
    float2 *traj = (float2 *)(ws + L.traj);
    uint2 *meta = (uint2 *)(ws + L.meta);
    float2 *ar = (float2 *)(ws + L.ar);
    float *stats = (float *)(ws + L.stats);
    float2 *cyx = (float2 *)(ws + L.cyx);
    uint32_t *cmax = (uint32_t *)(ws + L.cmax), *wmax = (uint32_t *)(ws + L.wmax);
    const float2 *fl = (const float2 *)flows_yx;
    Events g = to_events(grad);
    const int FB = w.F * w.B;
    bool finished = false;
    if (w.M > 0) {
        int chunks = (w.M + 255) / 256;
        dim3 grid(xcd_grid(FB, chunks));
        int *defer_count = (int *)(ws + L.queue) + kDeferWord, *defer = (int *)(ws + L.defer);
        if (w.kind == TEF_KIND_ITERATIVE && w.S == 1) {
            // the fast form of every step; then the few wavefronts that needed the general form somewhere (the count is on the
            // device: one workgroup per CU walks the list)
            TEF_LAUNCH_TIMED(tef::PROF_CHAIN_BWD, (iter_chain_bwd_kernel<true, 1>), grid, dim3(256), 0, st, w, fl, g, traj, meta, ar,
                             stats, grad_out, cyx, wmax, chunks, defer_count, defer, cmax);
            if (int rc = tef::check_launch("chain_bwd_kernel")) return rc;
            const int R = std::max(1, std::min(8, num_cus() / std::max(1, FB)));
            TEF_LAUNCH_TIMED(tef::PROF_CHAIN_BWD_REST, chain_bwd_finish_kernel, dim3((unsigned)(FB * R)), dim3(256), 0, st, w, fl, g, traj,
                             meta, ar, stats, grad_out, cyx, wmax, defer_count, defer, cmax, R, (int *)(ws + L.queue) + 8);
            finished = true;
        } else if (w.kind == TEF_KIND_ITERATIVE)
            TEF_LAUNCH_TIMED(tef::PROF_CHAIN_BWD, (iter_chain_bwd_kernel<false, 0>), grid, dim3(256), 0, st, w, fl, g, traj, meta, ar,
                             stats, grad_out, cyx, wmax, chunks, defer_count, defer, cmax);
        else
            TEF_LAUNCH_TIMED(tef::PROF_CHAIN_BWD, linear_bwd_kernel, grid, dim3(256), 0, st, w, g, traj, meta, ar, stats,
                             grad_out, cyx, wmax, chunks);
    }
    if (int rc = tef::check_launch("chain_bwd_kernel")) return rc;
    if (!finished)       // (the single-scale Iterative path folds the magnitudes in its finishing launch)
        TEF_LAUNCH_TIMED(tef::PROF_STATS, mag_reduce_kernel, dim3((unsigned)FB), dim3(256), 0, st, wmax, w.M / 64, cmax,
                         (int *)(ws + L.queue) + 8);      // (M = 0: writes zeros)
    if (int rc = tef::check_launch("mag_reduce_kernel")) return rc;
    int rows, nbands;
    size_t lds;
    band_geometry(w, 2, &rows, &nbands, &lds, kSplatLdsBudget, 2);
    {
        const long items = (long)w.P * FB * nbands;
        unsigned grid = (unsigned)std::min<long>(items, 2 * num_cus());
        grid = std::max(8u, (grid + 7u) & ~7u);
        TEF_LAUNCH_TIMED(tef::PROF_DFLOW, dflow_splat_kernel, dim3(grid), dim3(kSplatThreads), lds, st, w, g, traj,
                         (const float2 *)(ws + L.yr), cyx, cmax, dflows, rows, nbands, (int *)(ws + L.queue) + 8);
    }
    return tef::check_launch("dflow_splat_kernel");
}

}  // extern "C"
