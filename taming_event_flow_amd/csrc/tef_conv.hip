// tef_conv.hip — dense convolution contractions of RecEVFlowNet on the gfx950 matrix cores.
// Reference: models/submodules.py ConvLayer :8-62, ConvGRU :111-152, ResidualBlock :155-227,
// UpsampleConvLayer :230-273 (all nn.Conv2d 3x3 / 1x1, fp32).
//
// Numerics: fp32 in, fp32 accumulate on v_mfma_f32_32x32x2_f32 (bit-wise a k-ordered fp32 fma chain; gfx950 has
// no xf32/TF32), so results match the reference's fp32 convolutions to summation-order noise — the north star's
// 1e-4 bar rules out plain bf16.
//
// Structure: implicit GEMM.  One "NT" kernel  C[r][c] = sum_k A[r][k] * B[c][k]  whose operand loaders either read a
// plain k-contiguous matrix or gather the im2col element on the fly (channel concat of two NCHW sources, optional
// gating product, stride / transposed-stride geometry), so no im2col / col2im / transposed copies exist in memory:
//     forward   out[n][m]   = Wp[n][k]   x gather_x[m][k]       k = (ci, ky, kx)     epilogue: bias, act, NCHW
//     dgrad     dx[ci][m']  = W2[ci][k'] x gather_g[m'][k']     k' = (n, ky, kx)     epilogue: NCHW, split x0 | x1
//     wgrad     dW[n][k]   += g[n][m]    x gather_x^T[k][m]     reduction over pixels, split over blockIdx.z, atomics
// where g = dout * act'(out) (one elementwise pass that also reduces the bias gradient).  Deep levels (M = B*h*w as
// small as 512, K up to 9216) are split over K into slabs and reduced with the epilogue.
// 128x128 (or 64x128 / 32x128) workgroup tiles, 4 waves, 32x32x2 MFMA tiles, LDS staged with register prefetch.
#include <hip/hip_runtime.h>
#include <stdint.h>
#include <stdlib.h>
#include <string.h>

#include <algorithm>

#include "tef.h"
#include "tef_common.h"

namespace {

typedef float f32x16 __attribute__((ext_vector_type(16)));

constexpr int BK = 32;          // k-depth of one LDS stage (long enough to cover the gather latency of the next)
constexpr int PPR = BK / 4;     // 4-float pieces per tile row
constexpr int LOGP = 3;         // log2(PPR)
static_assert((1 << LOGP) == PPR, "pieces per row");
constexpr int LDK = BK + 4;     // padded LDS row (words): 16-byte aligned rows, spreads b128 reads over the banks

enum { EPI_FWD = 0, EPI_ATOMIC = 2, EPI_SLAB = 3 };
enum { A_PLAIN = 0, A_NCHW = 1 };
enum { B_GATHER = 1, B_GATHER_T = 2 };

// im2col element source: value(pixel p = (b, py, px) of the output grid, k = (ci, ky, kx))
struct Gather {
    const float *src0, *src1, *gate1;   // NCHW sources, channel-concatenated; src1 optionally times gate1
    int C0, C1;          // channels of the two sources
    int SH, SW;          // source spatial size
    int OH, OW;          // output-grid size (pixels are decomposed against this)
    int ks, pad;         // kernel size (1 | 3), padding
    int mul, div, sgn;   // source y = (py * mul + sgn * (ky - pad)) / div   (forward: mul = stride, div = 1, sgn = +1;
                         //  input gradient: mul = 1, div = stride, sgn = -1, only exact multiples contribute)
    int K;               // true reduction length (ci, ky, kx); k >= K reads as zero
    int npix;            // B * OH * OW; pixels >= npix read as zero
    long npix_src;       // B * SH * SW: elements per channel-of-all-images of a source (bounds of the 16-byte gather)
};


struct GemmArgs {
    const float *A;       // A_PLAIN: [rows][lda];  A_NCHW: g [B][rows][hwA] read as A[r][kk = (b, p)]
    Gather G;             // B operand
    float *C, *C2;        // EPI_FWD: NCHW outputs, rows < split go to C ([B][split][hw]), the rest to C2
    const float *bias;
    int rows, cols, K;    // K % 16 == 0 (padded reduction length)
    int lda, ldc;
    int act, hw, split;
    int hwA;              // A_NCHW: pixels per image of g
    int ksplit;           // EPI_ATOMIC / EPI_SLAB: reduction range per blockIdx.z
    int valid_cols;
    int s2d_ct;           // stride-2 input gradient on the halo kernel: rows = 4 parity classes x s2d_ct channels
    // ConvGRU state update fused into the out gate's epilogue (EPI_FWD, one output tensor): with v = act(...) the new
    // state bl_out = bl_h * (1 - bl_u) + v * bl_u is stored beside v (reference submodules.py:150); all null otherwise
    const float *bl_h, *bl_u;
    float *bl_out;
    // halo kernel: this launch covers rows [row_off, row_end) of the problem (row_end = 0: all of them) — a row count just
    // past a multiple of the tile height is worked as whole large tiles plus one launch of 32-row tiles (launch_halo_w)
    int row_off, row_end;
};

__device__ __forceinline__ void store_blend(const GemmArgs &g, size_t idx, float v)
{
    if (g.bl_out) {
        const float uu = g.bl_u[idx];
        g.bl_out[idx] = g.bl_h[idx] * (1.0f - uu) + v * uu;
    }
}

__device__ __forceinline__ float apply_act(float v, int act)
{
    if (act == TEF_ACT_RELU) return fmaxf(v, 0.0f);
    if (act == TEF_ACT_TANH) return tanhf(v);
    if (act == TEF_ACT_SIGMOID) return 1.0f / (1.0f + expf(-v));
    return v;
}

struct Pix { int b, py, px; bool ok; };

__device__ __forceinline__ Pix decode_pixel(const Gather &G, int p)
{
    Pix q;
    q.ok = p < G.npix;
    int pp = q.ok ? p : 0;
    int ohw = G.OH * G.OW;
    q.b = pp / ohw;
    int r = pp - q.b * ohw;
    q.py = r / G.OW;
    q.px = r - q.py * G.OW;
    return q;
}

typedef float f32x4_a4 __attribute__((ext_vector_type(4), aligned(4)));   // 16-byte load at 4-byte alignment
typedef int i32x4 __attribute__((ext_vector_type(4)));

// TR x TC workgroup tile, WR x WC wave tile (multiples of 32), NT = 256 or 512 threads (4 or 8 waves).
// QV: quad-vector gathers (stride-1 geometry, rows / images multiples of 4 pixels); GATED: the second source is multiplied
// by gate1.  Both are template parameters so that the staging code is straight-line.
//
// Operand staging (the part that decides the speed of an implicit GEMM):
//   A, plain:        float4 along k, LDS row-major [TR][BK+4], fragments by ds_read_b128.
//   A, NCHW (wgrad): g[n][(b, p)], float4 along the pixels (QV) or 4 scalar loads.
//   B, gather:       a thread owns a QUAD of 4 consecutive output pixels and one k per load: with QV the valid source
//                    elements are contiguous, so the im2col gather is ONE 16-byte load (wave-coalesced along the
//                    pixels) anchored at element 0's position, staged k-major [BK][TC+4] with one ds_write_b128.  The 9
//                    tap offsets of every pixel live in an LDS table; (ci, tap) advance incrementally.  Other geometry
//                    (strides, ragged rows) uses 4 scalar loads per quad.
//   B, gather^T (wgrad): the tile row is a fixed (ci, ky, kx), the reduction walks over pixels; same 16-byte trick,
//                    LDS row-major [TC][BK+4].
//   Every load is unconditional (invalid lanes read element 0); validity, gating and the end-of-tensor shifts are bit
//   flags applied when the registers go to LDS, after the MFMA block of the stage.
template <int NT, int TR, int TC, int WR, int WC, int AM, int BM, int EPI, bool QV, bool GATED>
__global__ __launch_bounds__(NT) void gemm_nt_kernel(GemmArgs g)
{
    static_assert((TR / WR) * (TC / WC) == NT / 64, "one wave per WR x WC sub-tile");
    static_assert(NT == 256 || NT == 512, "4 or 8 waves per workgroup");
    static_assert(TC == 128 && BK == 32, "the quad staging below is written for 128 columns x 32 k");
    constexpr int MR = WR / 32, MC = WC / 32;          // MFMA tiles per wave
    constexpr int LDB = TC + 4;                        // k-major B rows (floats), 16-byte multiple
    constexpr int BFLOATS = (BM == B_GATHER) ? BK * LDB : TC * LDK;
    constexpr int NBUF = 2;          // LDS stages (a single-buffered variant with two barriers per stage was slower)
    __shared__ __attribute__((aligned(16))) float As[NBUF][TR][LDK];
    __shared__ __attribute__((aligned(16))) float Bs[NBUF][BFLOATS];
    __shared__ __attribute__((aligned(16))) int soff[BM == B_GATHER ? 9 : 1][BM == B_GATHER ? TC : 4];
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int wr = wave / (TC / WC), wc = wave % (TC / WC);
    const int row0 = blockIdx.y * TR, col0 = blockIdx.x * TC;
    int k_begin = 0, k_end = g.K;
    if (EPI == EPI_ATOMIC || EPI == EPI_SLAB) {
        k_begin = blockIdx.z * g.ksplit;
        k_end = min(g.K, k_begin + g.ksplit);
        if (k_begin >= k_end) return;
    }
    const int SHW = g.G.SH * g.G.SW;
    const int kk2 = g.G.ks * g.G.ks;
    const int Ct = g.G.C0 + g.G.C1;
    const bool k3 = kk2 == 9;

    // ---- A staging: 4-float pieces (tile row = piece / PPR, k offset = 4 * (piece % PPR)) ----
    constexpr int AP = (TR * PPR + NT - 1) / NT;
    float4 ra[AP];
    unsigned amask = 0;            // A_NCHW: bit 4p+e = element e of piece p is inside the reduction range
    auto load_a = [&](int k0) {
        amask = 0;
#pragma unroll
        for (int p = 0; p < AP; ++p) {
            int piece = tid + p * NT;
            int r = piece >> LOGP, kq = (piece & (PPR - 1)) * 4;
            bool ok = (TR * PPR % NT == 0 || piece < TR * PPR) && (row0 + r) < g.rows;
            if (AM == A_PLAIN) {
                // unconditional: rows beyond g.rows (clamped to the last row) only feed output rows that are never stored
                int rc = min(row0 + r, g.rows - 1);
                ra[p] = *reinterpret_cast<const float4 *>(g.A + (size_t)rc * g.lda + k0 + kq);
            } else {            // A[r][kk], kk = (image, pixel) of an NCHW tensor with `rows` channels
                // unconditional loads (out-of-range pieces read element 0 and are zeroed at store): no if/else merge of
                // loaded values, nothing in the load phase waits on memory
                int kk = k0 + kq;
                int img = kk / g.hwA, px = kk - img * g.hwA;
                if (QV) {       // images are multiples of 4 pixels: the piece is inside one image, all in or all out
                    bool in = ok && kk < g.G.npix;
                    size_t o = in ? ((size_t)img * g.rows + row0 + r) * g.hwA + px : 0;
                    ra[p] = *reinterpret_cast<const float4 *>(g.A + o);
                    if (in) amask |= 0xfu << (4 * p);
                } else {
                    float *v = reinterpret_cast<float *>(&ra[p]);
#pragma unroll
                    for (int j = 0; j < 4; ++j) {
                        bool in = ok && kk + j < g.G.npix;
                        size_t o = in ? ((size_t)img * g.rows + row0 + r) * g.hwA + px : 0;
                        v[j] = g.A[o];
                        if (in) amask |= 1u << (4 * p + j);
                        if (++px == g.hwA) { px = 0; ++img; }
                    }
                }
            }
        }
    };
    auto store_a = [&](int buf) {
#pragma unroll
        for (int p = 0; p < AP; ++p) {
            int piece = tid + p * NT;
            float4 v = ra[p];
            if (AM == A_NCHW) {
                unsigned m = amask >> (4 * p);
                v.x = (m & 1u) ? v.x : 0.0f;
                v.y = (m & 2u) ? v.y : 0.0f;
                v.z = (m & 4u) ? v.z : 0.0f;
                v.w = (m & 8u) ? v.w : 0.0f;
            }
            if (TR * PPR % NT == 0 || piece < TR * PPR)
                *reinterpret_cast<float4 *>(&As[buf][piece >> LOGP][(piece & (PPR - 1)) * 4]) = v;
        }
    };

    // ---- B staging ----
    constexpr int BP = 1024 / NT;  // 16-byte loads per thread per stage in both B modes (128 x 32 floats / NT / 4)
    constexpr int KS = NT / 32;    // k slots of the gather
    float4 rb[BP];
    // B_GATHER: thread = (quad of 4 pixels, k slot); k = k0 + kslot + KS j
    const int quad = tid & 31, kslot = tid >> 5;
    int qb[4];                     // image index of the quad's pixels
    int ci0 = 0, r0 = 0;           // (ci, tap) of k = k0 + kslot
    if (BM == B_GATHER) {
        for (int t = tid; t < TC * 9; t += NT) {
            int r = t / TC, pix = t - r * TC;
            Pix q = decode_pixel(g.G, col0 + pix);
            int off = -1;
            if (q.ok && r < kk2) {
                int ky = r / g.G.ks, kx = r - ky * g.G.ks;
                int ty = q.py * g.G.mul + g.G.sgn * (ky - g.G.pad), tx = q.px * g.G.mul + g.G.sgn * (kx - g.G.pad);
                bool ok = ty >= 0 && tx >= 0;
                if (ok && g.G.div > 1) {
                    ok = !((ty % g.G.div) | (tx % g.G.div));
                    ty /= g.G.div;
                    tx /= g.G.div;
                }
                if (ok && ty < g.G.SH && tx < g.G.SW) off = ty * g.G.SW + tx;
            }
            soff[r][pix] = off;
        }
#pragma unroll
        for (int e = 0; e < 4; ++e) qb[e] = decode_pixel(g.G, col0 + 4 * quad + e).b;
        int k = k_begin + kslot;
        ci0 = k / kk2;
        r0 = k - ci0 * kk2;
        __syncthreads();
    }
    // B_GATHER_T: piece = (tile row = (ci, ky, kx), 4 consecutive pixels of the reduction)
    const float *tsrc[BP], *tgate[BP];      // source tensor (and gate) of the tile row's channel, null beyond K
    int tky[BP], tkx[BP], tcs[BP], tcl[BP];  // tap, channels of that source, channel index within it
    if (BM == B_GATHER_T) {
#pragma unroll
        for (int p = 0; p < BP; ++p) {
            int k = col0 + ((tid + p * NT) >> LOGP);
            int ci = k / kk2, r = k - ci * kk2;
            tky[p] = r / g.G.ks;
            tkx[p] = r - tky[p] * g.G.ks;
            bool second = ci >= g.G.C0;
            tcs[p] = second ? g.G.C1 : g.G.C0;
            tcl[p] = second ? ci - g.G.C0 : ci;
            tsrc[p] = k < g.G.K ? (second ? g.G.src1 : g.G.src0) : nullptr;
            tgate[p] = (k < g.G.K && second && g.G.gate1) ? g.G.gate1 : nullptr;
        }
    }

    // Loads only ISSUE here (values, gate values and validity bits stay in registers); masking and the gating product
    // happen in store_b, after the MFMA block.  Every destination register has exactly one (predicated) load writing
    // it — no if/else merges of loaded values, which would make the compiler wait for memory inside the load phase — so
    // all gathers of a stage are in flight together.
    //   QV (stride-1 geometry, rows a multiple of 4 pixels): the valid elements of a quad are contiguous in the source
    //   row, so ONE 16-byte load anchored at element 0's (possibly out-of-row) position covers them; invalid elements
    //   are masked at store.  At the two ends of a source tensor the anchor would fall 1 element outside the
    //   allocation: the load is moved by one element and the value rotated back at store (shift flag).
    float4 rg[BP];                 // gate values of the second source (ConvGRU reset gate), when present
    unsigned bmask = 0;            // bit 4j+e: element e of load j valid;  bit 16+j: load j gated;  bits 20+2j: shift
    constexpr bool has_gate = GATED;
    // every load is unconditional (invalid lanes read element 0 of the tensor and are masked at store)
    auto quad_load = [&](int j, const float *src, bool gated, long flat, long total, unsigned vm) {
        unsigned sh = 0;
        if (flat < 0) { flat += 1; sh = 1; }
        else if (flat + 3 >= total) { flat -= 1; sh = 2; }
        if (!vm) { flat = 0; sh = 0; }
        f32x4_a4 t = *reinterpret_cast<const f32x4_a4 *>(src + flat);
        rb[j] = make_float4(t.x, t.y, t.z, t.w);
        if (has_gate) {
            f32x4_a4 gt = *reinterpret_cast<const f32x4_a4 *>(g.G.gate1 + (gated ? flat : 0));
            rg[j] = make_float4(gt.x, gt.y, gt.z, gt.w);
            if (gated && vm) bmask |= 1u << (16 + j);
        }
        bmask |= (vm << (4 * j)) | (sh << (20 + 2 * j));
    };
    auto elem_load = [&](int j, int e, const float *src, bool gated, size_t o, bool ok) {
        float *v = reinterpret_cast<float *>(&rb[j]), *gv = reinterpret_cast<float *>(&rg[j]);
        o = ok ? o : 0;
        v[e] = src[o];
        if (has_gate) gv[e] = g.G.gate1[gated ? o : 0];
        if (ok) bmask |= 1u << (4 * j + e);
        if (ok && gated && has_gate) bmask |= 1u << (16 + j);
    };
    auto load_b = [&](int k0) {
        bmask = 0;
        if (BM == B_GATHER) {
            int ci = ci0, rr = r0;
#pragma unroll
            for (int j = 0; j < BP; ++j) {
                bool in_k = ci < Ct;
                int cc = in_k ? ci : 0;
                i32x4 so = *reinterpret_cast<const i32x4 *>(&soff[rr][4 * quad]);
                bool second = cc >= g.G.C0;
                const float *src = second ? g.G.src1 : g.G.src0;
                int cs = second ? g.G.C1 : g.G.C0, cl = second ? cc - g.G.C0 : cc;
                unsigned vm = (so.x >= 0 ? 1u : 0u) | (so.y >= 0 ? 2u : 0u) | (so.z >= 0 ? 4u : 0u) | (so.w >= 0 ? 8u : 0u);
                if (!in_k) vm = 0;
                if (QV) {
                    int o0 = so.x >= 0 ? so.x : (so.y >= 0 ? so.y - 1 : (so.z >= 0 ? so.z - 2 : so.w - 3));
                    quad_load(j, src, second, ((long)qb[0] * cs + cl) * SHW + o0, g.G.npix_src * cs, vm);
                } else {
                    const int sov[4] = {so.x, so.y, so.z, so.w};
#pragma unroll
                    for (int e = 0; e < 4; ++e)
                        elem_load(j, e, src, second, ((size_t)qb[e] * cs + cl) * SHW + sov[e], (vm >> e) & 1u);
                }
                // k += KS (branch-free: nothing but straight-line code between a load and its use)
                ci += k3 ? KS / 9 : KS;
                rr += k3 ? KS % 9 : 0;
                int wrap = rr >= 9 ? 1 : 0;
                rr -= 9 * wrap;
                ci += wrap;
            }
            // k0 += BK (32) for the next stage
            ci0 += k3 ? 3 : BK;
            r0 += k3 ? 5 : 0;
            int wrap0 = r0 >= 9 ? 1 : 0;
            r0 -= 9 * wrap0;
            ci0 += wrap0;
        } else {
#pragma unroll
            for (int p = 0; p < BP; ++p) {
                int piece = tid + p * NT;
                int r = piece >> LOGP, kq = (piece & (PPR - 1)) * 4;
                bool row_ok = (col0 + r) < g.cols && tsrc[p] != nullptr;
                const float *src = tsrc[p] ? tsrc[p] : g.G.src0;
                bool gated = tgate[p] != nullptr;
                int m = k0 + kq;
                Pix q = decode_pixel(g.G, m);
                if (QV) {          // 4 pixels of one row: source (ty, tx .. tx + 3)
                    int ty = q.py + g.G.sgn * (tky[p] - g.G.pad), tx = q.px + g.G.sgn * (tkx[p] - g.G.pad);
                    unsigned vm = 0;
#pragma unroll
                    for (int e = 0; e < 4; ++e)
                        if (m + e < g.G.npix && tx + e >= 0 && tx + e < g.G.SW) vm |= 1u << e;
                    if (!row_ok || ty < 0 || ty >= g.G.SH) vm = 0;
                    quad_load(p, src, gated, ((long)q.b * tcs[p] + tcl[p]) * SHW + ty * g.G.SW + tx, g.G.npix_src * tcs[p], vm);
                } else {
#pragma unroll
                    for (int j = 0; j < 4; ++j) {
                        int yy = q.py * g.G.mul + g.G.sgn * (tky[p] - g.G.pad);
                        int xx = q.px * g.G.mul + g.G.sgn * (tkx[p] - g.G.pad);
                        bool ok = row_ok && (m + j < g.G.npix) && yy >= 0 && xx >= 0 && yy < g.G.SH && xx < g.G.SW;
                        elem_load(p, j, src, gated, ((size_t)q.b * tcs[p] + tcl[p]) * SHW + yy * g.G.SW + xx, ok);
                        if (++q.px == g.G.OW) { q.px = 0; if (++q.py == g.G.OH) { q.py = 0; ++q.b; } }
                    }
                }
            }
        }
    };
    // the masked, gated value of load j as it goes to LDS
    auto staged = [&](int j) {
        float4 v = rb[j];
        if (bmask & (1u << (16 + j))) { v.x *= rg[j].x; v.y *= rg[j].y; v.z *= rg[j].z; v.w *= rg[j].w; }
        unsigned sh = (bmask >> (20 + 2 * j)) & 3u;
        if (sh == 1) v = make_float4(0.f, v.x, v.y, v.z);
        else if (sh == 2) v = make_float4(v.y, v.z, v.w, 0.f);
        unsigned m = bmask >> (4 * j);
        v.x = (m & 1u) ? v.x : 0.0f;
        v.y = (m & 2u) ? v.y : 0.0f;
        v.z = (m & 4u) ? v.z : 0.0f;
        v.w = (m & 8u) ? v.w : 0.0f;
        return v;
    };
    auto store_b = [&](int buf) {
        if (BM == B_GATHER) {
#pragma unroll
            for (int j = 0; j < BP; ++j)
                *reinterpret_cast<float4 *>(&Bs[buf][(kslot + KS * j) * LDB + 4 * quad]) = staged(j);
        } else {
#pragma unroll
            for (int p = 0; p < BP; ++p) {
                int piece = tid + p * NT;
                *reinterpret_cast<float4 *>(&Bs[buf][(piece >> LOGP) * LDK + (piece & (PPR - 1)) * 4]) = staged(p);
            }
        }
    };

    f32x16 acc[MR][MC];
#pragma unroll
    for (int i = 0; i < MR; ++i)
#pragma unroll
        for (int j = 0; j < MC; ++j)
#pragma unroll
            for (int e = 0; e < 16; ++e) acc[i][j][e] = 0.0f;

    // one k-stage of MFMAs from LDS buffer `buf`: lane l feeds row/col (l & 31) and k = 4 * (l >> 5) + j of each
    // 8-wide half stage
    auto multiply = [&](int buf) {
#pragma unroll
        for (int kh = 0; kh < BK; kh += 8) {
            float4 fa[MR], fb[MC];
#pragma unroll
            for (int i = 0; i < MR; ++i)
                fa[i] = *reinterpret_cast<const float4 *>(&As[buf][wr * WR + i * 32 + (lane & 31)][kh + 4 * (lane >> 5)]);
#pragma unroll
            for (int j = 0; j < MC; ++j) {
                if (BM == B_GATHER) {
                    const float *bp = &Bs[buf][(kh + 4 * (lane >> 5)) * LDB + wc * WC + j * 32 + (lane & 31)];
                    fb[j] = make_float4(bp[0], bp[LDB], bp[2 * LDB], bp[3 * LDB]);
                } else {
                    fb[j] = *reinterpret_cast<const float4 *>(&Bs[buf][(wc * WC + j * 32 + (lane & 31)) * LDK + kh + 4 * (lane >> 5)]);
                }
            }
#pragma unroll
            for (int i = 0; i < MR; ++i)
#pragma unroll
                for (int j = 0; j < MC; ++j) {
                    acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x2f32(fa[i].x, fb[j].x, acc[i][j], 0, 0, 0);
                    acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x2f32(fa[i].y, fb[j].y, acc[i][j], 0, 0, 0);
                    acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x2f32(fa[i].z, fb[j].z, acc[i][j], 0, 0, 0);
                    acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x2f32(fa[i].w, fb[j].w, acc[i][j], 0, 0, 0);
                }
        }
    };
    // The values in flight must not be touched before the MFMA block is issued: the compiler otherwise hoists the
    // copies / selects of store_b up to the loads and waits for memory there.  The loop body is straight-line (last
    // stage peeled), a scheduling barrier closes the MFMA block and the registers are re-defined by an empty asm.
    auto pin = [&]() {
        __builtin_amdgcn_sched_barrier(0);
#pragma unroll
        for (int p = 0; p < AP; ++p) asm volatile("" : "+v"(ra[p].x), "+v"(ra[p].y), "+v"(ra[p].z), "+v"(ra[p].w));
        if (AM == A_NCHW) asm volatile("" : "+v"(amask));
#pragma unroll
        for (int p = 0; p < BP; ++p) {
            asm volatile("" : "+v"(rb[p].x), "+v"(rb[p].y), "+v"(rb[p].z), "+v"(rb[p].w));
            if (GATED) asm volatile("" : "+v"(rg[p].x), "+v"(rg[p].y), "+v"(rg[p].z), "+v"(rg[p].w));
        }
        asm volatile("" : "+v"(bmask));
    };

    load_a(k_begin);
    load_b(k_begin);
    store_a(0);
    store_b(0);
    __syncthreads();
    int buf = 0;
    for (int k0 = k_begin + BK; k0 < k_end; k0 += BK) {
        load_a(k0);                 // next stage in flight while this one is multiplied
        load_b(k0);
        __builtin_amdgcn_sched_barrier(0);      // every load is issued before the first MFMA
        multiply(buf);
        pin();
        store_a(buf ^ 1);
        store_b(buf ^ 1);
        __syncthreads();
        buf ^= 1;
    }
    multiply(buf);

    // C/D layout of 32x32 MFMA: col = lane & 31, row = (reg & 3) + 8 * (reg >> 2) + 4 * (lane >> 5)
#pragma unroll
    for (int i = 0; i < MR; ++i)
#pragma unroll
        for (int j = 0; j < MC; ++j) {
            int c = col0 + wc * WC + j * 32 + (lane & 31);
            if (c >= g.cols) continue;
            int img = 0, px = 0;
            if (EPI == EPI_FWD) {
                img = c / g.hw;
                px = c - img * g.hw;
            } else if (c >= g.valid_cols) {
                continue;
            }
#pragma unroll
            for (int e = 0; e < 16; ++e) {
                int r = row0 + wr * WR + i * 32 + (e & 3) + 8 * (e >> 2) + 4 * (lane >> 5);
                if (r >= g.rows) continue;
                float v = acc[i][j][e];
                if (EPI == EPI_FWD) {
                    if (g.bias) v += g.bias[r];
                    v = apply_act(v, g.act);
                    if (r < g.split) {
                        const size_t idx = ((size_t)img * g.split + r) * g.hw + px;
                        g.C[idx] = v;
                        store_blend(g, idx, v);
                    } else {
                        g.C2[((size_t)img * (g.rows - g.split) + (r - g.split)) * g.hw + px] = v;
                    }
                } else if (EPI == EPI_SLAB) {
                    g.C[((size_t)blockIdx.z * g.rows + r) * g.ldc + c] = v;
                } else {        // rows < split accumulate into C, the rest into C2 (row-concatenated parameters)
                    atomicAdd(r < g.split ? g.C + (size_t)r * g.ldc + c : g.C2 + (size_t)(r - g.split) * g.ldc + c, v);
                }
            }
        }
}

// =====================================================================================================================
// 3x3 stride-1 convolution with the input patch staged in LDS ("halo" kernel) — forward and input gradient of the
// layers whose rows are 16 / 32 / 64 / 128 pixels wide.  The implicit-GEMM kernel above gathers every im2col element from
// L2 (each input element 9 times); here a workgroup's 128 output pixels are R = 128 / W whole image rows, the 8 input
// channels of a k-chunk are loaded ONCE as an (R + 2) x (W + 2) patch (zero halo) and the nine taps are read from it.
//   k order inside a chunk: (tap, channel) = 9 x 8 = 72;  A = weights repacked [row][chunk][tap][8] (input gradient:
//   taps flipped at pack time, so the kernel always reads source pixel (y + ky - 1, x + kx - 1)).
//   A is staged in 3 sub-stages of 24 k (3 taps), double-buffered; the patch is double-buffered per chunk.
// Waves: (TR / WR) x 4, every wave a WR x 32 tile (MC = 1).  Epilogues as above (EPI_FWD / EPI_SLAB).
// =====================================================================================================================
constexpr int HC = 8;            // channels per k-chunk
constexpr int HK = 9 * HC;       // k per chunk
constexpr int HS = 3 * HC;       // k per A sub-stage (3 taps)
constexpr int HLDA = HS + 4;     // LDS row pitch of the A sub-tile (floats)

// IPT > 1: images smaller than a tile (8 x 8 pixels at the deepest level) — the tile holds IPT whole images, each with
// its own zero halo rows in the patch.
// GEN: images of any size (width a multiple of 4): the 128 pixels are a W x R rectangle of the image (W = tile width),
// the halo columns hold real neighbours and are loaded with the patch, edge tiles are masked.
// S = 2: the stride-2 encoder heads (forward only): the tile's R x W outputs read a (2R + 1) x (2W + 1) input patch with
// stride-2 windows (padding 1: only the left / top halo exists, and it is zero).
#define HALO_SYNC() __syncthreads()
// S2D: input gradient of a STRIDE-2 layer as a stride-1 convolution over the output-gradient grid.  Input pixel
// (2a + py, 2b + px) only sees g at (a + dy, b + dx), dy, dx in {0, 1}: rows = (parity class (py, px), input channel),
// weights packed per class with zeros at the taps a class does not use (pack_s2d_kernel), those taps' MFMAs skipped
// (1 / 2 / 2 / 4 taps instead of 9), results stored depth-to-space.  4x less matrix work than gathering through the
// transposed stride.
template <int NT, int TR, int WR, int LOGW, int IPT, int EPI, bool GATED, bool GEN, int S = 1, bool S2D = false>
__global__ __launch_bounds__(NT) void conv3x3_halo_kernel(GemmArgs g)
{
    static_assert(!S2D || (S == 1 && !GATED), "the stride-2 input gradient reads one ungated source at stride 1");
    static_assert(!GEN || IPT == 1, "general tiles hold one image");
    static_assert(S == 1 || (!GEN && IPT == 1), "stride-2 tiles are whole output rows of one image");
    constexpr int W = 1 << LOGW, R = 128 >> LOGW, RI = R / IPT, PR = S == 1 ? IPT * (RI + 2) : 2 * R + 1,
                  PP = S * W + 8, PLANE = PR * PP;
    constexpr int MR = WR / 32;
    static_assert((TR / WR) * 4 == NT / 64, "one wave per WR x 32 sub-tile");
    // Three A buffers (one per sub-stage of a chunk) when they fit the 64 KiB of static LDS: a sub-stage's data is then
    // visible one barrier before it is needed, so its first operands are read BEFORE the barrier and no wave waits for
    // LDS right after it (PF3 loop below).  Otherwise two buffers and the plain store -> barrier -> read sequence.
    constexpr bool PF3 = (3 * TR * HLDA + 2 * HC * PLANE) * 4 <= 65536;
    constexpr int NA = PF3 ? 3 : 2;
    __shared__ __attribute__((aligned(16))) float As[NA][TR][HLDA];
    __shared__ __attribute__((aligned(16))) float Ps[2][HC][PLANE];

    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int wr = wave >> 2, wc = wave & 3, h = lane >> 5;
    const int row0 = blockIdx.y * TR + g.row_off, col0 = blockIdx.x * 128;
    const int row_end = g.row_end ? g.row_end : g.rows;
    const int WI = GEN ? g.G.SW : S * W;                  // (input) image width
    const int HWi = g.G.SH * WI;                          // pixels per input image (!GEN: a multiple of 128, or 128 / IPT)
    const int HWo = S == 1 ? HWi : g.G.OH * W;            // pixels per output image
    int img, y0, x0 = 0;                                  // image, first row and first column of the tile
    if (GEN) {
        const int ntx = (WI + W - 1) / W, nty = (g.G.SH + R - 1) / R;
        img = blockIdx.x / (ntx * nty);
        int rem = blockIdx.x - img * ntx * nty, ty = rem / ntx;
        y0 = ty * R;
        x0 = (rem - ty * ntx) * W;
    } else {
        img = col0 / HWo;
        y0 = (col0 - img * HWo) >> LOGW;
    }
    const int Ct = g.G.C0 + g.G.C1;
    const int nch = g.lda / HK;
    unsigned tapmask = 0x1ffu;                            // S2D: taps (ky' * 3 + kx') some class of this row tile uses
    if (S2D) {
        tapmask = 0;
        const int cls0 = row0 / g.s2d_ct, cls1 = (min(row0 + TR, g.rows) - 1) / g.s2d_ct;
        for (int cls = cls0; cls <= cls1; ++cls) {
            const unsigned ys = (cls >> 1) ? 6u : 2u, xs = (cls & 1) ? 6u : 2u;      // bit k: window row / column k used
            for (int ky = 0; ky < 3; ++ky)
                for (int kx = 0; kx < 3; ++kx)
                    if (((ys >> ky) & 1u) && ((xs >> kx) & 1u)) tapmask |= 1u << (ky * 3 + kx);
        }
    }
    int c_begin = 0, c_end = nch;
    if (EPI == EPI_SLAB) {
        c_begin = blockIdx.z * g.ksplit;
        c_end = min(nch, c_begin + g.ksplit);
        if (c_begin >= c_end) return;
    }

    // zero the halo columns of both patch buffers once (data columns are 4 .. 4 + W - 1); GEN rewrites them per chunk
    for (int t = tid; t < (GEN ? 0 : 2 * HC * PR); t += NT) {
        float *row = &Ps[0][0][0] + (size_t)t * PP;
        row[3] = 0.0f;
        row[4 + S * W] = 0.0f;
    }

    // ---- A sub-tile staging: TR rows x 6 float4 ----
    constexpr int AP = (TR * 6 + NT - 1) / NT;
    float4 ra[AP];
    auto load_a_to = [&](float4 (&ra)[AP], int chunk, int sub) {
#pragma unroll
        for (int p = 0; p < AP; ++p) {
            int piece = min(tid + p * NT, TR * 6 - 1);
            int r = piece / 6, q4 = piece - r * 6;
            int rc = min(row0 + r, g.rows - 1);
            ra[p] = *reinterpret_cast<const float4 *>(g.A + (size_t)rc * g.lda + (size_t)chunk * HK + sub * HS + q4 * 4);
        }
    };
    auto store_a_from = [&](const float4 (&ra)[AP], int buf) {
#pragma unroll
        for (int p = 0; p < AP; ++p) {
            int piece = tid + p * NT;
            if (TR * 6 % NT == 0 || piece < TR * 6) {
                int r = piece / 6, q4 = piece - r * 6;
                *reinterpret_cast<float4 *>(&As[buf][r][q4 * 4]) = ra[p];
            }
        }
    };
    auto load_a = [&](int chunk, int sub) { load_a_to(ra, chunk, sub); };
    auto store_a = [&](int buf) { store_a_from(ra, buf); };

    // ---- patch staging: 8 channels x PR rows x S W / 4 float4 ----
    constexpr int Q = S * W / 4, PIECES = HC * PR * Q, PPT = (PIECES + NT - 1) / NT;
    float4 rp[PPT], rq[PPT];
    int pofs[PPT], pcl[PPT];         // offset inside an image plane (-1: row outside the image / no piece), local channel
#pragma unroll
    for (int p = 0; p < PPT; ++p) {
        int piece = tid + p * NT;
        bool has = PIECES % NT == 0 || piece < PIECES;
        int pc = has ? piece : 0;
        int cl = pc / (PR * Q), rem = pc - cl * (PR * Q), prow = rem / Q, q4 = rem - prow * Q;
        int sub = S == 1 ? prow / (RI + 2) : 0;                                  // image of the tile, row inside it
        int y = S == 1 ? y0 - 1 + (prow - sub * (RI + 2)) : 2 * y0 - 1 + prow;
        pcl[p] = cl | (prow << 8) | (q4 << 16);
        if (GEN) pofs[p] = (has && y >= 0 && y < g.G.SH && x0 + q4 * 4 < WI) ? y * WI + x0 + q4 * 4 : -1;
        else pofs[p] = (has && y >= 0 && y < g.G.SH && (long)(img + sub) * HWo < (long)g.cols) ? sub * HWi + y * WI + q4 * 4 : -1;
    }
    unsigned pmask = 0;              // bit p: piece p holds data; bit 8 + p: gated; bits 16 / 17: halo element valid / gated
    float rh = 0.0f, rhq = 1.0f;     // GEN: this thread's halo-column element (x0 - 1 or x0 + W of one patch row)
    constexpr int HPIECES = HC * PR * 2;
    static_assert(!GEN || HPIECES <= NT, "one halo element per thread");
    auto load_p = [&](int chunk) {
        pmask = 0;
        if (GEN) {
            int pc = tid < HPIECES ? tid : 0;
            int cl = pc / (PR * 2), rem = pc - cl * (PR * 2), prow = rem >> 1, side = rem & 1;
            int ci = chunk * HC + cl, y = y0 - 1 + prow, x = side ? x0 + W : x0 - 1;
            bool ok = tid < HPIECES && ci < Ct && y >= 0 && y < g.G.SH && x >= 0 && x < WI;
            bool second = ok && ci >= g.G.C0;
            const float *src = second ? g.G.src1 : g.G.src0;
            int cs = second ? g.G.C1 : g.G.C0, clc = second ? ci - g.G.C0 : ci;
            size_t o = ok ? ((size_t)img * cs + clc) * HWi + y * WI + x : 0;
            rh = src[o];
            if (GATED) {
                rhq = g.G.gate1[second ? o : 0];
                if (second) pmask |= 1u << 17;
            }
            if (ok) pmask |= 1u << 16;
        }
#pragma unroll
        for (int p = 0; p < PPT; ++p) {
            int ci = chunk * HC + (pcl[p] & 0xff);
            bool ok = pofs[p] >= 0 && ci < Ct;
            bool second = ok && ci >= g.G.C0;          // padded channels (ci >= Ct) read element 0 of src0
            const float *src = second ? g.G.src1 : g.G.src0;
            int cs = second ? g.G.C1 : g.G.C0, clc = second ? ci - g.G.C0 : ci;
            // pofs = (image of the tile) * HWi + offset inside the plane: move the image part to the channel stride
            int sub = IPT > 1 ? pofs[p] / HWi : 0;
            size_t o = ok ? ((size_t)(img + sub) * cs + clc) * HWi + (pofs[p] - sub * HWi) : 0;
            rp[p] = *reinterpret_cast<const float4 *>(src + o);
            if (GATED) {
                rq[p] = *reinterpret_cast<const float4 *>(g.G.gate1 + ((ok && second) ? o : 0));
                if (ok && second) pmask |= 1u << (8 + p);
            }
            if (ok) pmask |= 1u << p;
        }
    };
    auto store_p = [&](int buf) {
#pragma unroll
        for (int p = 0; p < PPT; ++p) {
            int piece = tid + p * NT;
            if (PIECES % NT == 0 || piece < PIECES) {
                float4 v = rp[p];
                if (GATED && (pmask & (1u << (8 + p)))) { v.x *= rq[p].x; v.y *= rq[p].y; v.z *= rq[p].z; v.w *= rq[p].w; }
                if (!(pmask & (1u << p))) v = make_float4(0.f, 0.f, 0.f, 0.f);
                int cl = pcl[p] & 0xff, prow = (pcl[p] >> 8) & 0xff, q4 = pcl[p] >> 16;
                *reinterpret_cast<float4 *>(&Ps[buf][cl][prow * PP + 4 + q4 * 4]) = v;
            }
        }
        if (GEN && tid < HPIECES) {
            int cl = tid / (PR * 2), rem = tid - cl * (PR * 2), prow = rem >> 1, side = rem & 1;
            float v = rh;
            if (GATED && (pmask & (1u << 17))) v *= rhq;
            if (!(pmask & (1u << 16))) v = 0.0f;
            Ps[buf][cl][prow * PP + (side ? 4 + W : 3)] = v;
        }
    };
    auto pin = [&]() {
        __builtin_amdgcn_sched_barrier(0);
#pragma unroll
        for (int p = 0; p < AP; ++p) asm volatile("" : "+v"(ra[p].x), "+v"(ra[p].y), "+v"(ra[p].z), "+v"(ra[p].w));
    };
    auto pin_p = [&]() {
#pragma unroll
        for (int p = 0; p < PPT; ++p) {
            asm volatile("" : "+v"(rp[p].x), "+v"(rp[p].y), "+v"(rp[p].z), "+v"(rp[p].w));
            if (GATED) asm volatile("" : "+v"(rq[p].x), "+v"(rq[p].y), "+v"(rq[p].z), "+v"(rq[p].w));
        }
        asm volatile("" : "+v"(pmask));
        if (GEN) asm volatile("" : "+v"(rh), "+v"(rhq));
    };

    f32x16 acc[MR];
#pragma unroll
    for (int i = 0; i < MR; ++i)
#pragma unroll
        for (int e = 0; e < 16; ++e) acc[i][e] = 0.0f;

    // lane's pixel inside the tile and its patch base (tap (ky, kx) adds ky * PP + kx; channel j adds j * PLANE)
    const int pl = wc * 32 + (lane & 31);
    const int prow_l = IPT > 1 ? ((pl >> LOGW) / RI) * (RI + 2) + (pl >> LOGW) % RI : (pl >> LOGW);
    const int pbase = (4 * h) * PLANE + S * prow_l * PP + S * (pl & (W - 1)) + 3;
    // operands of one tap: A rows of the wave's MR blocks, patch values of the lane's pixel (4 channels of this half)
    struct Ops { float4 fa[MR]; float4 fb; };
    auto read_ops = [&](int abuf, int pbuf, int tap, Ops &o) {
        const int ky = tap / 3, kx = tap - ky * 3;
#pragma unroll
        for (int i = 0; i < MR; ++i)
            o.fa[i] = *reinterpret_cast<const float4 *>(&As[abuf][wr * WR + i * 32 + (lane & 31)][kx * HC + 4 * h]);
        const float *bp = &Ps[pbuf][0][0] + pbase + ky * PP + kx;
        o.fb = make_float4(bp[0], bp[PLANE], bp[2 * PLANE], bp[3 * PLANE]);
    };
    auto mfma_ops = [&](const Ops &o) {
        // MFMA / DS-read interleave hint for the scheduler: 31.30 -> 31.07 ms per captured training window (the same hint in the
        // weight-gradient kernel costs the captured window 0.07 ms and is left out; in the implicit GEMM it changes nothing)
        __builtin_amdgcn_iglp_opt(0);
#pragma unroll
        for (int i = 0; i < MR; ++i) {
            acc[i] = __builtin_amdgcn_mfma_f32_32x32x2f32(o.fa[i].x, o.fb.x, acc[i], 0, 0, 0);
            acc[i] = __builtin_amdgcn_mfma_f32_32x32x2f32(o.fa[i].y, o.fb.y, acc[i], 0, 0, 0);
            acc[i] = __builtin_amdgcn_mfma_f32_32x32x2f32(o.fa[i].z, o.fb.z, acc[i], 0, 0, 0);
            acc[i] = __builtin_amdgcn_mfma_f32_32x32x2f32(o.fa[i].w, o.fb.w, acc[i], 0, 0, 0);
        }
    };
    auto multiply = [&](int abuf, int pbuf, int sub) {
#pragma unroll
        for (int t = 0; t < 3; ++t) {
            if (S2D && !((tapmask >> (sub * 3 + t)) & 1u)) continue;
            Ops o;
            read_ops(abuf, pbuf, sub * 3 + t, o);
            mfma_ops(o);
        }
    };

    if constexpr (!PF3) {
        load_p(c_begin);
        load_a(c_begin, 0);
        store_p(0);
        store_a(0);
        HALO_SYNC();
        int abuf = 0, pbuf = 0;
        for (int c = c_begin; c < c_end; ++c) {
            const int cn = min(c + 1, c_end - 1);       // the last chunk re-loads itself (stored, never read)
            // sub-stage 0
            load_a(c, 1);
            __builtin_amdgcn_sched_barrier(0);
            multiply(abuf, pbuf, 0);
            pin();
            store_a(abuf ^ 1);
            HALO_SYNC();
            abuf ^= 1;
            // sub-stage 1: also start the next chunk's patch
            load_a(c, 2);
            load_p(cn);
            __builtin_amdgcn_sched_barrier(0);
            multiply(abuf, pbuf, 1);
            pin();
            store_a(abuf ^ 1);
            HALO_SYNC();
            abuf ^= 1;
            // sub-stage 2
            load_a(cn, 0);
            __builtin_amdgcn_sched_barrier(0);
            multiply(abuf, pbuf, 2);
            pin();
            pin_p();
            store_a(abuf ^ 1);
            store_p(pbuf ^ 1);
            HALO_SYNC();
            abuf ^= 1;
            pbuf ^= 1;
        }
    } else {
        // Sub-stage `sub` of a chunk lives in As[sub].  Iteration (c, sub): store the A data of the sub-stage two ahead
        // (its buffer was last read one iteration ago), start the global loads of the one three ahead, multiply while
        // reading the next tap's operands — the last tap reads the first operands of the NEXT sub-stage, stored an
        // iteration ago and visible since the previous barrier — then barrier.  The patch of chunk c + 1 is loaded in
        // (c, 0) and stored in (c, 1).
        float4 ra1[AP];
        load_p(c_begin);
        load_a(c_begin, 0);
        load_a_to(ra1, c_begin, 1);
        store_p(0);
        store_a(0);
        store_a_from(ra1, 1);
        load_a(c_begin, 2);
        HALO_SYNC();
        Ops cur, nxt;
        read_ops(0, 0, 0, cur);
        int pbuf = 0;
        for (int c = c_begin; c < c_end; ++c) {
            const int cn = min(c + 1, c_end - 1);       // the last chunk re-loads itself (stored, never read)
#pragma unroll
            for (int sub = 0; sub < 3; ++sub) {
                pin();
                store_a((sub + 2) % 3);
                if (sub == 1) { pin_p(); store_p(pbuf ^ 1); }
                load_a(cn, sub);                        // three ahead: sub-stage (c + 1, sub)
                if (sub == 0) load_p(cn);
                __builtin_amdgcn_sched_barrier(0);
#pragma unroll
                for (int t = 0; t < 3; ++t) {
                    if (t < 2) read_ops(sub, pbuf, sub * 3 + t + 1, nxt);
                    else read_ops((sub + 1) % 3, sub == 2 ? pbuf ^ 1 : pbuf, sub == 2 ? 0 : (sub + 1) * 3, nxt);
                    if (!S2D || ((tapmask >> (sub * 3 + t)) & 1u)) mfma_ops(cur);
                    cur = nxt;
                }
                HALO_SYNC();
            }
            pbuf ^= 1;
        }
    }

    // C/D layout of 32x32 MFMA: col = lane & 31, row = (reg & 3) + 8 * (reg >> 2) + 4 * (lane >> 5)
    int cc = col0 + pl;
    int img_l = IPT > 1 ? cc / HWo : img;
    int px = cc - img_l * HWo;
    bool pix_ok = cc < g.cols;
    if (GEN) {        // the lane's pixel inside the image; the column index becomes the true pixel index
        int y = y0 + (pl >> LOGW), x = x0 + (pl & (W - 1));
        pix_ok = y < g.G.SH && x < WI;
        px = y * WI + x;
        cc = img * HWi + px;
    }
#pragma unroll
    for (int i = 0; i < MR; ++i) {
        if (!pix_ok) continue;
#pragma unroll
        for (int e = 0; e < 16; ++e) {
            int r = row0 + wr * WR + i * 32 + (e & 3) + 8 * (e >> 2) + 4 * h;
            if (r >= row_end) continue;
            float v = acc[i][e];
            if (EPI == EPI_FWD && S2D) {        // depth-to-space: class (py, px) of channel ci -> pixel (2a + py, 2b + px)
                const int wg = GEN ? WI : W, cls = r / g.s2d_ct, ci = r - cls * g.s2d_ct;
                const int a = px / wg, b = px - a * wg;
                g.C[(((size_t)img_l * g.s2d_ct + ci) * (2 * g.G.SH) + 2 * a + (cls >> 1)) * (size_t)(2 * wg) + 2 * b + (cls & 1)] = v;
            } else if (EPI == EPI_FWD) {
                if (g.bias) v += g.bias[r];
                v = apply_act(v, g.act);
                if (r < g.split) {
                    const size_t idx = ((size_t)img_l * g.split + r) * g.hw + px;
                    g.C[idx] = v;
                    store_blend(g, idx, v);
                } else {
                    g.C2[((size_t)img_l * (g.rows - g.split) + (r - g.split)) * g.hw + px] = v;
                }
            } else {
                g.C[((size_t)blockIdx.z * g.rows + r) * g.ldc + cc] = v;
            }
        }
    }
}

// =====================================================================================================================
// Weight gradient of the same layers with the input patch in LDS:  dW[n][(ci, tap)] += sum_pixels g[n][m] x[ci][m + tap].
// rows = output channels (A = g, NCHW, k = pixels, read as in the implicit GEMM), columns = (ci, tap) in the weight's own
// order, reduction over pixels in stages of 32.  The 128 columns of a tile touch at most 16 input channels: their
// (rows + 2) x (32 + 2) patch of the stage's pixels is loaded once and every lane reads its own (channel, tap) window
// from it — 9x fewer input loads than the transposed im2col gather, which is what the narrow layers are bound by.
// =====================================================================================================================
constexpr int WGC = 16;          // input channels a 128-column tile can touch

// Deferred weight gradient over several backward calls of one layer (the passes of a BPTT window): part p holds images
// [p * B, (p + 1) * B) of one long pixel reduction; n == 0: the single set of tensors in GemmArgs.
struct WgradParts {
    const float *A[TEF_CONV_MAX_PARTS], *src0[TEF_CONV_MAX_PARTS], *src1[TEF_CONV_MAX_PARTS], *gate1[TEF_CONV_MAX_PARTS];
    int n, B;
};

// S = 2: the stride-2 encoder heads.  W is the OUTPUT width; the stage's RB x CW output pixels read a (2 RB + 1) x
// (2 CW + 1) input patch (padding 1: top row and left column are the halo, nothing is needed on the right), and a
// lane's window steps two input pixels per output pixel.
template <int NT, int TR, int WR, int LOGW, bool GATED, int S = 1>
__global__ __launch_bounds__(NT) void wgrad3x3_halo_kernel(GemmArgs g, WgradParts wp)
{
    static_assert(S == 1 || !GATED, "the stride-2 heads have one ungated source");
    constexpr int W = 1 << LOGW, CW = W < 32 ? W : 32, RB = 32 / CW, PR = S == 1 ? RB + 2 : 2 * RB + 1,
                  PP = S * CW + 8, PLANE = PR * PP + 1;
    constexpr int MR = WR / 32;
    static_assert((TR / WR) * 4 == NT / 64, "one wave per WR x 32 sub-tile");
    __shared__ __attribute__((aligned(16))) float As[2][TR][LDK];
    __shared__ __attribute__((aligned(16))) float Ps[2][WGC * PLANE + 3];

    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int wr = wave >> 2, wc = wave & 3, h = lane >> 5;
    const int row0 = blockIdx.y * TR, col0 = blockIdx.x * 128;
    const int H = g.G.SH, WI = S * W, HWi = H * WI;      // input image; HWo: pixels per output image (= HWi at stride 1)
    const int HWo = S == 1 ? HWi : g.G.OH * W;
    const int Ct = g.G.C0 + g.G.C1;
    const int c_lo = col0 / 9;
    const int nst = g.K / 32;                                  // pixel stages in total (g.K = padded pixel count)
    int s_begin = blockIdx.z * g.ksplit, s_end = min(nst, s_begin + g.ksplit);
    if (s_begin >= s_end) return;

    // ---- A staging: TR rows x 8 float4 of g[n][m0 .. m0 + 31] ----
    constexpr int AP = (TR * 8 + NT - 1) / NT;
    float4 ra[AP];
    unsigned amask = 0;
    auto load_a = [&](int st) {
        amask = 0;
        const int m0 = st * 32;
        int img = m0 / HWo;
        const int po = m0 - img * HWo;
        const float *Ab = g.A;
        if (wp.n) {                                   // the stage's part (uniform): its tensors, image index inside it
            int part = min(img / wp.B, wp.n - 1);
            Ab = wp.A[part];
            img -= part * wp.B;
        }
#pragma unroll
        for (int p = 0; p < AP; ++p) {
            int piece = tid + p * NT;
            int r = piece >> 3, q4 = piece & 7;
            bool ok = (TR * 8 % NT == 0 || piece < TR * 8) && (row0 + r) < g.rows && m0 < g.G.npix;
            size_t o = ok ? ((size_t)img * g.rows + row0 + r) * HWo + po + q4 * 4 : 0;
            ra[p] = *reinterpret_cast<const float4 *>(Ab + o);
            if (ok) amask |= 1u << p;
        }
    };
    auto store_a = [&](int buf) {
#pragma unroll
        for (int p = 0; p < AP; ++p) {
            int piece = tid + p * NT;
            if (TR * 8 % NT == 0 || piece < TR * 8) {
                float4 v = (amask & (1u << p)) ? ra[p] : make_float4(0.f, 0.f, 0.f, 0.f);
                *reinterpret_cast<float4 *>(&As[buf][piece >> 3][(piece & 7) * 4]) = v;
            }
        }
    };

    // ---- patch staging: 16 channels x PR rows x (CW / 4 float4 + 2 halo scalars) ----
    constexpr int Q = S * CW / 4, PIECES = WGC * PR * Q, PPT = (PIECES + NT - 1) / NT;
    constexpr int HSIDES = S == 1 ? 2 : 1;                 // halo columns per patch row
    constexpr int HPIECES = WGC * PR * HSIDES;
    static_assert(HPIECES <= NT, "one halo element per thread");
    float4 rp[PPT], rq[PPT];
    float rh = 0.0f, rhq = 1.0f;
    unsigned pmask = 0;              // bit p: piece p holds data; bit 8 + p: gated; bit 16: halo valid; bit 17: halo gated
    auto load_p = [&](int st) {
        pmask = 0;
        const int m0 = st * 32;
        int img = m0 / HWo;
        const int po = m0 - img * HWo;
        const float *s0 = g.G.src0, *s1 = g.G.src1, *gt = g.G.gate1;
        if (wp.n) {
            int part = min(img / wp.B, wp.n - 1);
            s0 = wp.src0[part]; s1 = wp.src1[part]; gt = wp.gate1[part];
            img -= part * wp.B;
        }
        const int y0 = S * (po >> LOGW), x0 = S * (po & (W - 1));     // first input row / column of the stage's windows (+ halo)
        const bool live = m0 < g.G.npix;
#pragma unroll
        for (int p = 0; p < PPT; ++p) {
            int piece = tid + p * NT;
            bool has = PIECES % NT == 0 || piece < PIECES;
            int pc = has ? piece : 0;
            int cl = pc / (PR * Q), rem = pc - cl * (PR * Q), prow = rem / Q, q4 = rem - prow * Q;
            int ci = c_lo + cl, y = y0 - 1 + prow;
            bool ok = has && live && ci < Ct && y >= 0 && y < H;
            bool second = ok && ci >= g.G.C0;
            const float *src = second ? s1 : s0;
            int cs = second ? g.G.C1 : g.G.C0, clc = second ? ci - g.G.C0 : ci;
            size_t o = ok ? ((size_t)img * cs + clc) * HWi + y * WI + x0 + q4 * 4 : 0;
            rp[p] = *reinterpret_cast<const float4 *>(src + o);
            if (GATED) {
                rq[p] = *reinterpret_cast<const float4 *>(gt + (second ? o : 0));
                if (second) pmask |= 1u << (8 + p);
            }
            if (ok) pmask |= 1u << p;
        }
        {   // halo columns x0 - 1 and x0 + CW (inside the row only when the row is wider than the stage)
            int pc = tid < HPIECES ? tid : 0;
            int cl = pc / (PR * HSIDES), rem = pc - cl * (PR * HSIDES), prow = rem / HSIDES, side = rem - prow * HSIDES;
            int ci = c_lo + cl, y = y0 - 1 + prow, x = side ? x0 + CW : x0 - 1;
            bool ok = tid < HPIECES && live && ci < Ct && y >= 0 && y < H && x >= 0 && x < WI;
            bool second = ok && ci >= g.G.C0;
            const float *src = second ? s1 : s0;
            int cs = second ? g.G.C1 : g.G.C0, clc = second ? ci - g.G.C0 : ci;
            size_t o = ok ? ((size_t)img * cs + clc) * HWi + y * WI + x : 0;
            rh = src[o];
            if (GATED) {
                rhq = gt[second ? o : 0];
                if (second) pmask |= 1u << 17;
            }
            if (ok) pmask |= 1u << 16;
        }
    };
    auto store_p = [&](int buf) {
#pragma unroll
        for (int p = 0; p < PPT; ++p) {
            int piece = tid + p * NT;
            if (PIECES % NT == 0 || piece < PIECES) {
                float4 v = rp[p];
                if (GATED && (pmask & (1u << (8 + p)))) { v.x *= rq[p].x; v.y *= rq[p].y; v.z *= rq[p].z; v.w *= rq[p].w; }
                if (!(pmask & (1u << p))) v = make_float4(0.f, 0.f, 0.f, 0.f);
                int cl = piece / (PR * Q), rem = piece - cl * (PR * Q), prow = rem / Q, q4 = rem - prow * Q;
                float *dst = &Ps[buf][cl * PLANE + prow * PP + 4 + q4 * 4];      // PLANE is odd: scalar stores
                dst[0] = v.x; dst[1] = v.y; dst[2] = v.z; dst[3] = v.w;
            }
        }
        if (tid < HPIECES) {
            int cl = tid / (PR * HSIDES), rem = tid - cl * (PR * HSIDES), prow = rem / HSIDES, side = rem - prow * HSIDES;
            float v = rh;
            if (GATED && (pmask & (1u << 17))) v *= rhq;
            if (!(pmask & (1u << 16))) v = 0.0f;
            Ps[buf][cl * PLANE + prow * PP + (side ? 4 + CW : 3)] = v;
        }
    };
    auto pin = [&]() {
        __builtin_amdgcn_sched_barrier(0);
#pragma unroll
        for (int p = 0; p < AP; ++p) asm volatile("" : "+v"(ra[p].x), "+v"(ra[p].y), "+v"(ra[p].z), "+v"(ra[p].w));
#pragma unroll
        for (int p = 0; p < PPT; ++p) {
            asm volatile("" : "+v"(rp[p].x), "+v"(rp[p].y), "+v"(rp[p].z), "+v"(rp[p].w));
            if (GATED) asm volatile("" : "+v"(rq[p].x), "+v"(rq[p].y), "+v"(rq[p].z), "+v"(rq[p].w));
        }
        asm volatile("" : "+v"(rh), "+v"(rhq), "+v"(pmask), "+v"(amask));
    };

    f32x16 acc[MR];
#pragma unroll
    for (int i = 0; i < MR; ++i)
#pragma unroll
        for (int e = 0; e < 16; ++e) acc[i][e] = 0.0f;

    // lane's column = (input channel, tap): its window base in the patch (pixel (ry, rx) of the stage adds ry * PP + rx)
    const int col = col0 + wc * 32 + (lane & 31);
    const int cil = min(col / 9 - c_lo, WGC - 1), tap = col % 9, ky = tap / 3, kx = tap - ky * 3;
    const int pbase = cil * PLANE + ky * PP + kx + 3 + S * 4 * h;
    // operands of one 8-pixel k-step: g rows of the wave's MR blocks, the lane's (channel, tap) window of the patch
    struct Ops { float4 fa[MR]; float4 fb; };
    auto read_ops = [&](int buf, int kh, Ops &o) {
#pragma unroll
        for (int i = 0; i < MR; ++i)
            o.fa[i] = *reinterpret_cast<const float4 *>(&As[buf][wr * WR + i * 32 + (lane & 31)][kh * 8 + 4 * h]);
        const float *bp = &Ps[buf][pbase + S * (((kh * 8) / CW) * PP + (kh * 8) % CW)];
        o.fb = make_float4(bp[0], bp[S], bp[2 * S], bp[3 * S]);
    };
    auto mfma_ops = [&](const Ops &o) {
#pragma unroll
        for (int i = 0; i < MR; ++i) {
            acc[i] = __builtin_amdgcn_mfma_f32_32x32x2f32(o.fa[i].x, o.fb.x, acc[i], 0, 0, 0);
            acc[i] = __builtin_amdgcn_mfma_f32_32x32x2f32(o.fa[i].y, o.fb.y, acc[i], 0, 0, 0);
            acc[i] = __builtin_amdgcn_mfma_f32_32x32x2f32(o.fa[i].z, o.fb.z, acc[i], 0, 0, 0);
            acc[i] = __builtin_amdgcn_mfma_f32_32x32x2f32(o.fa[i].w, o.fb.w, acc[i], 0, 0, 0);
        }
    };

    // The barrier of a stage sits before its LAST k-step, whose operands are already in registers: the buffer is free
    // and the next stage (stored at the top of the iteration) visible when the barrier opens, and the first operands of
    // the next stage are read behind the last k-step's MFMAs instead of right after a barrier.  Stages st + 1 and st + 2
    // beyond the slice are loaded and stored but never multiplied.
    load_a(s_begin);
    load_p(s_begin);
    store_a(0);
    store_p(0);
    load_a(s_begin + 1);
    load_p(s_begin + 1);
    __syncthreads();
    Ops cur, nxt;
    read_ops(0, 0, cur);
    int buf = 0;
    for (int st = s_begin; st < s_end; ++st) {
        pin();
        store_a(buf ^ 1);
        store_p(buf ^ 1);
        load_a(st + 2);
        load_p(st + 2);
        __builtin_amdgcn_sched_barrier(0);
#pragma unroll
        for (int kh = 0; kh < 3; ++kh) {
            read_ops(buf, kh + 1, nxt);
            mfma_ops(cur);
            cur = nxt;
        }
        __syncthreads();
        read_ops(buf ^ 1, 0, nxt);
        mfma_ops(cur);
        cur = nxt;
        buf ^= 1;
    }

    if (col >= g.valid_cols) return;
#pragma unroll
    for (int i = 0; i < MR; ++i)
#pragma unroll
        for (int e = 0; e < 16; ++e) {
            int r = row0 + wr * WR + i * 32 + (e & 3) + 8 * (e >> 2) + 4 * h;
            if (r >= g.rows) continue;
            atomicAdd(r < g.split ? g.C + (size_t)r * g.ldc + col : g.C2 + (size_t)(r - g.split) * g.ldc + col, acc[i][e]);
        }
}

// split-K epilogue: out = act(bias[r] + sum_z slab[z][r][c]) scattered to the NCHW output(s); 4 columns (pixels of one
// image) per thread when the geometry allows 16-byte accesses
__global__ __launch_bounds__(256) void splitk_reduce_kernel(const float *__restrict__ slab, int z, int rows, int cols,
                                                            const float *__restrict__ bias, int act, int hw, int split,
                                                            float *__restrict__ out, float *__restrict__ out2,
                                                            const float *__restrict__ bl_h, const float *__restrict__ bl_u,
                                                            float *__restrict__ bl_out)
{
    const size_t plane = (size_t)rows * cols;
    if (((cols | hw) & 3) == 0) {
        const int c4 = cols >> 2;
        size_t idx = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
        if (idx >= (size_t)rows * c4) return;
        int r = (int)(idx / c4), c = (int)(idx - (size_t)r * c4) << 2;
        float b0 = bias ? bias[r] : 0.0f;
        float4 v = make_float4(b0, b0, b0, b0);
        const float *sp = slab + (size_t)r * cols + c;
        for (int k = 0; k < z; ++k) {
            float4 t = *reinterpret_cast<const float4 *>(sp + (size_t)k * plane);
            v.x += t.x; v.y += t.y; v.z += t.z; v.w += t.w;
        }
        v = make_float4(apply_act(v.x, act), apply_act(v.y, act), apply_act(v.z, act), apply_act(v.w, act));
        int img = c / hw, px = c - img * hw;
        float *o = r < split ? out + ((size_t)img * split + r) * hw + px
                             : out2 + ((size_t)img * (rows - split) + (r - split)) * hw + px;
        *reinterpret_cast<float4 *>(o) = v;
        if (bl_out && r < split) {          // ConvGRU state update beside the out gate's activation (submodules.py:150)
            const size_t idx = ((size_t)img * split + r) * hw + px;
            const float4 hh = *reinterpret_cast<const float4 *>(bl_h + idx), uu = *reinterpret_cast<const float4 *>(bl_u + idx);
            *reinterpret_cast<float4 *>(bl_out + idx) = make_float4(hh.x * (1.0f - uu.x) + v.x * uu.x, hh.y * (1.0f - uu.y) + v.y * uu.y,
                                                                    hh.z * (1.0f - uu.z) + v.z * uu.z, hh.w * (1.0f - uu.w) + v.w * uu.w);
        }
        return;
    }
    size_t idx = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (idx >= plane) return;
    int r = (int)(idx / cols), c = (int)(idx - (size_t)r * cols);
    float v = bias ? bias[r] : 0.0f;
    for (int k = 0; k < z; ++k) v += slab[(size_t)k * plane + (size_t)r * cols + c];
    v = apply_act(v, act);
    int img = c / hw, px = c - img * hw;
    if (r < split) {
        const size_t idx = ((size_t)img * split + r) * hw + px;
        out[idx] = v;
        if (bl_out) {
            const float uu = bl_u[idx];
            bl_out[idx] = bl_h[idx] * (1.0f - uu) + v * uu;
        }
    } else {
        out2[((size_t)img * (rows - split) + (r - split)) * hw + px] = v;
    }
}

// split-K epilogue of an INPUT GRADIENT whose only consumer is the pre-activation gradient of the layer that produced the
// input (round 6: the residual blocks — dgrad -> relu' -> next dgrad):  g = act'(mask) * (sum_z slab[z][r][c] + addend),
// dbias[r] += sum g, scattered to NCHW like splitk_reduce_kernel.  grad_act_kernel's operations on the same values; the
// reduce launch and the activation-gradient launch become one.  z == 0: no slabs, g_out holds the input gradient already
// (the convolution ran unsplit) and is rewritten in place.
__global__ __launch_bounds__(256) void splitk_reduce_post_kernel(const float *__restrict__ slab, int z, int rows, int cols, int hw,
                                                                 const float *__restrict__ mask, int act,
                                                                 const float *__restrict__ addend, float *__restrict__ g_out,
                                                                 float *__restrict__ dbias)
{
    const size_t plane = (size_t)rows * cols;
    const bool vec = ((cols | hw) & 3) == 0;
    const int per = vec ? 4 : 1, cq = cols / per;
    const size_t idx = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
    const bool in = idx < (size_t)rows * cq;
    const int r = in ? (int)(idx / cq) : 0, c = in ? (int)(idx - (size_t)r * cq) * per : 0;
    const int img = c / hw, px = c - img * hw;
    const size_t o = ((size_t)img * rows + r) * hw + px;
    float local = 0.0f;
    if (in) {
        float v[4] = {0.0f, 0.0f, 0.0f, 0.0f};
        if (z == 0) {
            for (int e = 0; e < per; ++e) v[e] = g_out[o + e];
        } else {
            const float *sp = slab + (size_t)r * cols + c;
            for (int k = 0; k < z; ++k)
                for (int e = 0; e < per; ++e) v[e] += sp[(size_t)k * plane + e];
        }
        for (int e = 0; e < per; ++e) {
            float t = v[e];
            if (addend) t = t + addend[o + e];
            if (act != TEF_ACT_NONE) {
                const float y = mask[o + e];
                t = act == TEF_ACT_RELU ? (y > 0.0f ? t : 0.0f) : act == TEF_ACT_TANH ? t * (1.0f - y * y) : t * (y * (1.0f - y));
            }
            g_out[o + e] = t;
            local += t;
        }
    }
    if (!dbias) return;
    if ((cq & 63) == 0) {          // a wavefront lies inside one row: one atomic per wavefront
        for (int s_ = 32; s_ > 0; s_ >>= 1) local += __shfl_down(local, s_, 64);
        if ((threadIdx.x & 63) == 0 && in) atomicAdd(dbias + r, local);
    } else if (in) {
        atomicAdd(dbias + r, local);
    }
}

// split-K epilogue of the stride-2 input gradient (conv3x3_halo_kernel, S2D): slab rows = (parity class, channel), slab
// columns = pixels (img, a, b) of the hg x wg output-gradient grid; dx[img][ci][2a + py][2b + px] = sum of the slabs.
// One thread per pixel PAIR of the input row: it adds the px = 0 and px = 1 classes and stores 8 bytes.
__global__ __launch_bounds__(256) void splitk_reduce_s2d_kernel(const float *__restrict__ slab, int z, int ct, int cols,
                                                                int hg, int wg, float *__restrict__ dx)
{
    const size_t plane = (size_t)4 * ct * cols;
    size_t idx = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (idx >= (size_t)2 * ct * cols) return;
    int rr = (int)(idx / cols), c = (int)(idx - (size_t)rr * cols);        // rr = py * ct + ci
    int py = rr / ct, ci = rr - py * ct;
    const float *s0 = slab + ((size_t)(2 * py) * ct + ci) * cols + c, *s1 = s0 + (size_t)ct * cols;
    float v0 = 0.0f, v1 = 0.0f;
    for (int k = 0; k < z; ++k) { v0 += s0[(size_t)k * plane]; v1 += s1[(size_t)k * plane]; }
    int img = c / (hg * wg), p = c - img * (hg * wg), a = p / wg, b = p - a * wg;
    float *o = dx + (((size_t)img * ct + ci) * (2 * hg) + 2 * a + py) * (size_t)(2 * wg) + 2 * b;
    *reinterpret_cast<float2 *>(o) = make_float2(v0, v1);
}

// g = dout * act'(out) (out = post-activation: relu' = out > 0, tanh' = 1 - out^2, sigmoid' = out (1 - out)),
// db[n] += sum over batch and pixels of g.   One block row per channel.
// dy / out may come as two tensors (channels [0, iosplit) and [iosplit, N): the split outputs of a fused conv).
__device__ __forceinline__ float act_grad_of(float gval, float y, int act)
{
    if (act == TEF_ACT_RELU) return y > 0.0f ? gval : 0.0f;
    if (act == TEF_ACT_TANH) return gval * (1.0f - y * y);
    if (act == TEF_ACT_SIGMOID) return gval * (y * (1.0f - y));
    return gval;
}

__global__ __launch_bounds__(256) void act_bwd_kernel(const float *__restrict__ dy, const float *__restrict__ dy2,
                                                      const float *__restrict__ out, const float *__restrict__ out2,
                                                      int iosplit, int act, int B, int N, int HW,
                                                      float *__restrict__ gbuf, float *__restrict__ dbias,
                                                      float *__restrict__ dbias2, int split)
{
    __shared__ float red[4];
    const int n = blockIdx.y;
    const bool second = n >= iosplit;
    const float *src = second ? dy2 : dy, *ysrc = second ? out2 : out;
    const int cs = second ? N - iosplit : iosplit, cl = second ? n - iosplit : n;      // channels / index in its tensor
    const bool store = act != TEF_ACT_NONE || iosplit < N;
    float local = 0.0f;
    if ((HW & 3) == 0) {       // 4 pixels of one image per 16-byte access
        const int Q = (B * HW) >> 2;
        for (int q = blockIdx.x * blockDim.x + threadIdx.x; q < Q; q += gridDim.x * blockDim.x) {
            int m = q << 2, b = m / HW, p = m - b * HW;
            size_t oi = ((size_t)b * cs + cl) * HW + p, o = ((size_t)b * N + n) * HW + p;
            float4 gv = *reinterpret_cast<const float4 *>(src + oi);
            if (act != TEF_ACT_NONE) {
                float4 y = *reinterpret_cast<const float4 *>(ysrc + oi);
                gv = make_float4(act_grad_of(gv.x, y.x, act), act_grad_of(gv.y, y.y, act), act_grad_of(gv.z, y.z, act),
                                 act_grad_of(gv.w, y.w, act));
            }
            if (store) *reinterpret_cast<float4 *>(gbuf + o) = gv;
            local += (gv.x + gv.y) + (gv.z + gv.w);
        }
    } else {
        const int M = B * HW;
        for (int m = blockIdx.x * blockDim.x + threadIdx.x; m < M; m += gridDim.x * blockDim.x) {
            int b = m / HW, p = m - b * HW;
            size_t oi = ((size_t)b * cs + cl) * HW + p, o = ((size_t)b * N + n) * HW + p;
            float gval = src[oi];
            if (act != TEF_ACT_NONE) gval = act_grad_of(gval, ysrc[oi], act);
            if (store) gbuf[o] = gval;
            local += gval;
        }
    }
    if (!dbias) return;
    for (int s = 32; s > 0; s >>= 1) local += __shfl_down(local, s, 64);
    if ((threadIdx.x & 63) == 0) red[threadIdx.x >> 6] = local;
    __syncthreads();
    if (threadIdx.x == 0) {
        float t = (red[0] + red[1]) + (red[2] + red[3]);
        atomicAdd(n < split ? dbias + n : dbias2 + (n - split), t);
    }
}

// ---- 1x1 convolutions onto a handful of channels (the flow prediction heads, Cin -> 2, tanh) --------------------
// Streaming kernels instead of GEMM tiles: a 2-row output would leave 126 of the 128 tile rows empty.
constexpr int kPwMaxN = 4;

// out[b][n][p] = act(bias[n] + sum_ci W[n][ci] x[b][ci][p]);  wp [N][Kp] (k = ci).
// Workgroup = 32 pixels x 8 channel groups (the groups' partial sums meet in LDS), so that small feature maps with
// many channels still fill the chip.
__global__ __launch_bounds__(256) void pw_fwd_kernel(const float *__restrict__ x, const float *__restrict__ wp,
                                                     const float *__restrict__ bias, int B, int C, int N, int HW, int Kp,
                                                     int act, float *__restrict__ out)
{
    __shared__ float part[8][kPwMaxN][32];
    const int px = threadIdx.x & 31, grp = threadIdx.x >> 5;
    const int m = blockIdx.x * 32 + px;
    const bool ok = m < B * HW;
    const int b = ok ? m / HW : 0, p = ok ? m - b * HW : 0;
    float acc[kPwMaxN];
#pragma unroll
    for (int n = 0; n < kPwMaxN; ++n) acc[n] = 0.0f;
    const float *xp = x + (size_t)b * C * HW + p;
    for (int ci = grp; ci < C; ci += 8) {
        float v = ok ? xp[(size_t)ci * HW] : 0.0f;
#pragma unroll
        for (int n = 0; n < kPwMaxN; ++n)
            if (n < N) acc[n] += wp[n * Kp + ci] * v;
    }
#pragma unroll
    for (int n = 0; n < kPwMaxN; ++n) part[grp][n][px] = acc[n];
    __syncthreads();
    if (grp < N && ok) {            // wave `grp` finishes output channel n = grp
        const int n = grp;
        float v = bias ? bias[n] : 0.0f;
#pragma unroll
        for (int k = 0; k < 8; ++k) v += part[k][n][px];
        out[((size_t)b * N + n) * HW + p] = apply_act(v, act);
    }
}

__device__ __forceinline__ float pw_g(const float *__restrict__ dy, const float *__restrict__ out, size_t o, int act)
{
    float g = dy[o];
    if (act == TEF_ACT_NONE) return g;
    float y = out[o];
    if (act == TEF_ACT_RELU) return y > 0.0f ? g : 0.0f;
    if (act == TEF_ACT_TANH) return g * (1.0f - y * y);
    return g * (y * (1.0f - y));
}

// dx[b][ci][p] = sum_n W[n][ci] g[b][n][p],  g = dY * act'(out);  w2 [C][K2p] (k' = n).  grid (pixels / 256, channel
// groups of 16)
__global__ __launch_bounds__(256) void pw_dx_kernel(const float *__restrict__ dy, const float *__restrict__ out,
                                                    const float *__restrict__ w2, int B, int C, int N, int HW, int K2p,
                                                    int act, float *__restrict__ dx)
{
    int m = blockIdx.x * blockDim.x + threadIdx.x;
    if (m >= B * HW) return;
    int b = m / HW, p = m - b * HW;
    float g[kPwMaxN];
#pragma unroll
    for (int n = 0; n < kPwMaxN; ++n) g[n] = n < N ? pw_g(dy, out, ((size_t)b * N + n) * HW + p, act) : 0.0f;
    float *dp = dx + (size_t)b * C * HW + p;
    const int c0 = blockIdx.y * 16, c1 = min(C, c0 + 16);
    for (int ci = c0; ci < c1; ++ci) {
        float v = 0.0f;
#pragma unroll
        for (int n = 0; n < kPwMaxN; ++n)
            if (n < N) v += w2[ci * K2p + n] * g[n];
        dp[(size_t)ci * HW] = v;
    }
}

// dW[n][ci] += sum_{b,p} g[b][n][p] x[b][ci][p];  grid (input channel, pixel part); channel 0 also adds db[n] += sum g
__global__ __launch_bounds__(256) void pw_dw_kernel(const float *__restrict__ dy, const float *__restrict__ out,
                                                    const float *__restrict__ x, int B, int C, int N, int HW, int act,
                                                    float *__restrict__ dw, float *__restrict__ db)
{
    __shared__ float red[4][2 * kPwMaxN];
    const int ci = blockIdx.x;
    float sw[kPwMaxN], sb[kPwMaxN];
#pragma unroll
    for (int n = 0; n < kPwMaxN; ++n) sw[n] = sb[n] = 0.0f;
    for (int m = blockIdx.y * blockDim.x + threadIdx.x; m < B * HW; m += gridDim.y * blockDim.x) {
        int b = m / HW, p = m - b * HW;
        float v = x[((size_t)b * C + ci) * HW + p];
#pragma unroll
        for (int n = 0; n < kPwMaxN; ++n)
            if (n < N) {
                float g = pw_g(dy, out, ((size_t)b * N + n) * HW + p, act);
                sw[n] += g * v;
                sb[n] += g;
            }
    }
#pragma unroll
    for (int n = 0; n < kPwMaxN; ++n) {
        for (int s = 32; s > 0; s >>= 1) {
            sw[n] += __shfl_down(sw[n], s, 64);
            sb[n] += __shfl_down(sb[n], s, 64);
        }
        if ((threadIdx.x & 63) == 0) {
            red[threadIdx.x >> 6][n] = sw[n];
            red[threadIdx.x >> 6][kPwMaxN + n] = sb[n];
        }
    }
    __syncthreads();
    if ((int)threadIdx.x < N) {
        int n = threadIdx.x;
        if (dw) atomicAdd(dw + (size_t)n * C + ci, (red[0][n] + red[1][n]) + (red[2][n] + red[3][n]));
        if (db && ci == 0)
            atomicAdd(db + n, (red[0][kPwMaxN + n] + red[1][kPwMaxN + n]) + (red[2][kPwMaxN + n] + red[3][kPwMaxN + n]));
    }
}

// ---- the tail of a decoder level, backward, in ONE launch (round 6) -----------------------------------------------------
// models/arch.py:238-240 backward: prediction head (1x1, N <= 4 channels, tanh) and the decoder convolution's activation.
//   gp[b][n][p]  = (sum_s dpred[s][b][n][p]) * act'(pred[b][n][p])
//   db_pred[n]  += sum gp          dw_pred[n][c] += sum_{b, p} gp[b][n][p] * dec[b][c][p]
//   gd[b][c][p]  = dec_act'(dec[b][c][p]) * (sum_n w2[c][n] gp[b][n][p] + dfeat[b][c][p])       (dfeat may be null)
//   db_dec[c]   += sum gd
// = grad_act_kernel (prediction) + pw_dx_kernel + pw_dw_kernel + grad_act_kernel (decoder) with the same operations in the
// same order per element: four launches and a round trip of the [B, C, HW] input gradient (16.8 MB at the finest level)
// become one sweep.  Thread = kDhPix pixels (256 apart) x a group of kDhCh channels; the per-channel sums are reduced per
// wavefront through DPP, per workgroup through LDS, and leave as one atomic each.
constexpr int kDhCh = 8;
__device__ __forceinline__ float dh_wave_sum(float v)       // sum over the 64 lanes, valid in lane 63
{
#define TEF_DH_ADD(ctrl, rmask, bound) v += __int_as_float(__builtin_amdgcn_update_dpp(0, __float_as_int(v), ctrl, rmask, 0xf, bound));
    TEF_DH_ADD(0x111, 0xf, true) TEF_DH_ADD(0x112, 0xf, true) TEF_DH_ADD(0x114, 0xf, true) TEF_DH_ADD(0x118, 0xf, true)   // row_shr:1, 2, 4, 8
    TEF_DH_ADD(0x142, 0xa, false)      // row_bcast:15 -> rows 1, 3
    TEF_DH_ADD(0x143, 0xc, false)      // row_bcast:31 -> rows 2, 3
#undef TEF_DH_ADD
    return v;
}
__device__ __forceinline__ float dh_act_grad(float v, float y, int act)
{
    if (act == TEF_ACT_RELU) return y > 0.0f ? v : 0.0f;
    if (act == TEF_ACT_TANH) return v * (1.0f - y * y);
    if (act == TEF_ACT_SIGMOID) return v * (y * (1.0f - y));
    return v;
}
struct DhSrc { const float *p[4]; int n; };
// V = 4: a thread owns four consecutive pixels (16-byte loads and stores; HW % 4 == 0), V = 1: one pixel.  Every load of a
// thread's channel group is issued before the first use.
template <int V>
__global__ __launch_bounds__(256) void dec_head_bwd_kernel(DhSrc dpred, const float *__restrict__ pred, int pred_act,
                                                           const float *__restrict__ w2, int K2p, const float *__restrict__ dec,
                                                           int dec_act, const float *__restrict__ dfeat, int B, int C, int N,
                                                           int HW, float *__restrict__ gp, float *__restrict__ gd,
                                                           float *__restrict__ db_pred, float *__restrict__ dw_pred,
                                                           float *__restrict__ db_dec)
{
    __shared__ float red[4][kDhCh * (kPwMaxN + 1) + kPwMaxN];
    typedef float vec_t __attribute__((ext_vector_type(V)));
    const int M = B * HW, c0 = blockIdx.y * kDhCh, nc = min(kDhCh, C - c0);
    float sgd[kDhCh], sw[kPwMaxN][kDhCh], sb[kPwMaxN];
#pragma unroll
    for (int j = 0; j < kDhCh; ++j) {
        sgd[j] = 0.0f;
#pragma unroll
        for (int n = 0; n < kPwMaxN; ++n) sw[n][j] = 0.0f;
    }
#pragma unroll
    for (int n = 0; n < kPwMaxN; ++n) sb[n] = 0.0f;
    const int m = (blockIdx.x * 256 + threadIdx.x) * V;
    if (m < M) {
        const int b = m / HW, p = m - b * HW;
        const size_t base = ((size_t)b * C + c0) * HW + p;
        vec_t yv[kDhCh], fv[kDhCh];
#pragma unroll
        for (int j = 0; j < kDhCh; ++j) {
            const size_t o = base + (size_t)(j < nc ? j : 0) * HW;
            yv[j] = *reinterpret_cast<const vec_t *>(dec + o);
            if (dfeat) fv[j] = *reinterpret_cast<const vec_t *>(dfeat + o);
        }
        vec_t g[kPwMaxN];
#pragma unroll
        for (int n = 0; n < kPwMaxN; ++n) {
            if (n < N) {
                const size_t o = ((size_t)b * N + n) * HW + p;
                vec_t v = *reinterpret_cast<const vec_t *>(dpred.p[0] + o);
                for (int s_ = 1; s_ < dpred.n; ++s_) v += *reinterpret_cast<const vec_t *>(dpred.p[s_] + o);
                if (pred_act != TEF_ACT_NONE) {
                    const vec_t y = *reinterpret_cast<const vec_t *>(pred + o);
#pragma unroll
                    for (int e = 0; e < V; ++e) v[e] = dh_act_grad(v[e], y[e], pred_act);
                }
                g[n] = v;
                if (blockIdx.y == 0) {
                    *reinterpret_cast<vec_t *>(gp + o) = v;
#pragma unroll
                    for (int e = 0; e < V; ++e) sb[n] += v[e];
                }
            }
        }
#pragma unroll
        for (int j = 0; j < kDhCh; ++j) {
            if (j < nc) {
                float wj[kPwMaxN];
#pragma unroll
                for (int n = 0; n < kPwMaxN; ++n) wj[n] = n < N ? w2[(size_t)(c0 + j) * K2p + n] : 0.0f;      // (wave-uniform)
                vec_t out;
#pragma unroll
                for (int e = 0; e < V; ++e) {
                    float v = 0.0f;
#pragma unroll
                    for (int n = 0; n < kPwMaxN; ++n)
                        if (n < N) v += wj[n] * g[n][e];
                    if (dfeat) v = v + fv[j][e];
                    v = dh_act_grad(v, yv[j][e], dec_act);
                    out[e] = v;
                    sgd[j] += v;
#pragma unroll
                    for (int n = 0; n < kPwMaxN; ++n)
                        if (n < N) sw[n][j] += g[n][e] * yv[j][e];
                }
                *reinterpret_cast<vec_t *>(gd + base + (size_t)j * HW) = out;
            }
        }
    }
    // per-channel sums: wavefront (DPP) -> workgroup (LDS) -> one atomic each
    const int wave = threadIdx.x >> 6, lane = threadIdx.x & 63;
#pragma unroll
    for (int j = 0; j < kDhCh; ++j) {
        const float a = dh_wave_sum(sgd[j]);
        if (lane == 63) red[wave][j] = a;
#pragma unroll
        for (int n = 0; n < kPwMaxN; ++n) {
            if (n < N) {
                const float w_ = dh_wave_sum(sw[n][j]);
                if (lane == 63) red[wave][kDhCh * (1 + n) + j] = w_;
            }
        }
    }
#pragma unroll
    for (int n = 0; n < kPwMaxN; ++n) {
        const float a = dh_wave_sum(sb[n]);
        if (lane == 63) red[wave][kDhCh * (kPwMaxN + 1) + n] = a;
    }
    __syncthreads();
    const int t = threadIdx.x;
    if (t < kDhCh * (N + 1)) {
        const int q = t / kDhCh, j = t - q * kDhCh;
        if (j < nc) {
            const float v = (red[0][t] + red[1][t]) + (red[2][t] + red[3][t]);
            if (q == 0) { if (db_dec) atomicAdd(db_dec + c0 + j, v); }
            else if (dw_pred) atomicAdd(dw_pred + (size_t)(q - 1) * C + c0 + j, v);
        }
    } else if (t >= 128 && t < 128 + N && blockIdx.y == 0 && db_pred) {
        const int k = kDhCh * (kPwMaxN + 1) + (t - 128);
        atomicAdd(db_pred + (t - 128), (red[0][k] + red[1][k]) + (red[2][k] + red[3][k]));
    }
}

// weight part [rows][Ct][ks][ks] -> rows [row0, row0 + rows) of
//   wp [N][Kp]      k  = (ci, ky, kx)   forward A operand
//   w2 [Ct][K2p]    k' = (n, ky, kx)    input-gradient A operand
__device__ __forceinline__ void pack_weight_item(size_t idx, const float *__restrict__ w, int rows, int row0, int N, int Ct, int kk,
                                                 int Kp, int K2p, float *__restrict__ wp, float *__restrict__ w2)
{
    const int K = Ct * kk;
    if (idx < (size_t)rows * Kp) {
        int n = (int)(idx / Kp), k = (int)(idx - (size_t)n * Kp);
        wp[(size_t)(row0 + n) * Kp + k] = k < K ? w[(size_t)n * K + k] : 0.0f;
    }
    // w2 columns owned by this part: n in [row0, row0 + rows) (+ the zero padding when it is the last part)
    int c_lo = row0 * kk, c_hi = (row0 + rows == N) ? K2p : (row0 + rows) * kk;
    int wcols = c_hi - c_lo;
    if (idx < (size_t)Ct * wcols) {
        int ci = (int)(idx / wcols), c = c_lo + (int)(idx - (size_t)ci * wcols);
        float v = 0.0f;
        if (c < N * kk) {
            int n = c / kk, r = c - n * kk;
            v = w[((size_t)(n - row0) * Ct + ci) * kk + r];
        }
        w2[(size_t)ci * K2p + c] = v;
    }
}
__global__ __launch_bounds__(256) void pack_weight_kernel(const float *__restrict__ w, int rows, int row0, int N,
                                                          int Ct, int kk, int Kp, int K2p, float *__restrict__ wp,
                                                          float *__restrict__ w2)
{
    pack_weight_item((size_t)blockIdx.x * blockDim.x + threadIdx.x, w, rows, row0, N, Ct, kk, Kp, K2p, wp, w2);
}

// Halo-kernel operands of a 3x3 weight part [rows][Ct][3][3] (k-chunks of 8 channels, (tap, channel) inside a chunk):
//   wh  [N][nch][9][8]     forward:        wh[n][chunk][tap][c]   = W[n][8 chunk + c][tap]
//   w2h [Ct][nch2][9][8]   input gradient: w2h[ci][chunk][tap][c] = W[8 chunk + c][ci][8 - tap]   (taps flipped)
// zero beyond Ct / N.  As pack_weight_kernel, a row part fills its own rows of wh and its own columns of w2h.
__device__ __forceinline__ void pack_halo_item(size_t idx, const float *__restrict__ w, int rows, int row0, int N, int Ct, int nch,
                                               int nch2, float *__restrict__ wh, float *__restrict__ w2h)
{
    const int lda = nch * HK, lda2 = nch2 * HK;
    if (idx < (size_t)rows * lda) {
        int n = (int)(idx / lda), k = (int)(idx - (size_t)n * lda);
        int chunk = k / HK, r = k - chunk * HK, tap = r / HC, c = r - tap * HC, ci = chunk * HC + c;
        wh[(size_t)(row0 + n) * lda + k] = ci < Ct ? w[((size_t)n * Ct + ci) * 9 + tap] : 0.0f;
    }
    int n_lo = row0, n_hi = (row0 + rows == N) ? nch2 * HC : row0 + rows;      // the last part also writes the padding
    int span = (n_hi - n_lo) * 9;
    if (idx < (size_t)Ct * span) {
        int ci = (int)(idx / span), r = (int)(idx - (size_t)ci * span);
        int n = n_lo + r / 9, tap = r - (r / 9) * 9;
        float v = n < N ? w[((size_t)(n - row0) * Ct + ci) * 9 + (8 - tap)] : 0.0f;
        w2h[(size_t)ci * lda2 + (size_t)(n / HC) * HK + tap * HC + (n % HC)] = v;
    }
}
__global__ __launch_bounds__(256) void pack_halo_kernel(const float *__restrict__ w, int rows, int row0, int N, int Ct,
                                                        int nch, int nch2, float *__restrict__ wh, float *__restrict__ w2h)
{
    pack_halo_item((size_t)blockIdx.x * blockDim.x + threadIdx.x, w, rows, row0, N, Ct, nch, nch2, wh, w2h);
}

// Stride-2 input gradient on the halo kernel (S2D): w2s [4 Ct][nch2][9][8], row = class (py, px) * Ct + ci, window tap
// (ky', kx') reads g at (a + ky' - 1, b + kx' - 1).  Row 2a + py of the input meets output row a + dy through kernel row
// ky = 1 (py = 0, dy = 0), ky = 2 (py = 1, dy = 0) or ky = 0 (py = 1, dy = 1); columns likewise; every other tap is zero.
__device__ __forceinline__ void pack_s2d_item(size_t idx, const float *__restrict__ w, int rows, int row0, int N, int Ct, int nch2,
                                              float *__restrict__ w2s)
{
    const int lda2 = nch2 * HK;
    int n_lo = row0, n_hi = (row0 + rows == N) ? nch2 * HC : row0 + rows;      // the last part also writes the padding
    int span = (n_hi - n_lo) * 9;
    if (idx >= (size_t)4 * Ct * span) return;
    int row = (int)(idx / span), r = (int)(idx - (size_t)row * span);
    int cls = row / Ct, ci = row - cls * Ct, py = cls >> 1, px = cls & 1;
    int n = n_lo + r / 9, tap = r - (r / 9) * 9, kyw = tap / 3, kxw = tap - kyw * 3;
    int ky = py ? (kyw == 2 ? 0 : (kyw == 1 ? 2 : -1)) : (kyw == 1 ? 1 : -1);
    int kx = px ? (kxw == 2 ? 0 : (kxw == 1 ? 2 : -1)) : (kxw == 1 ? 1 : -1);
    float v = (n < N && ky >= 0 && kx >= 0) ? w[((size_t)(n - row0) * Ct + ci) * 9 + ky * 3 + kx] : 0.0f;
    w2s[(size_t)row * lda2 + (size_t)(n / HC) * HK + tap * HC + (n % HC)] = v;
}
__global__ __launch_bounds__(256) void pack_s2d_kernel(const float *__restrict__ w, int rows, int row0, int N, int Ct,
                                                       int nch2, float *__restrict__ w2s)
{
    pack_s2d_item((size_t)blockIdx.x * blockDim.x + threadIdx.x, w, rows, row0, N, Ct, nch2, w2s);
}

// Every weight part of a set of layers in ONE launch (round 6: tef_conv_pack_weights).  After an optimiser step the ~50 pack
// launches of a window ran one after the other in front of the first convolutions — ~1 ms in which nothing but small pack
// kernels was on the chip; as one launch the same 250 MB per half of the network move at the chip's rate.
// A workgroup looks its job up in the block-offset table and works one chunk of kPackIter x 256 items of it.
constexpr int kPackJobs = 40, kPackIter = 4;
struct PackJob {
    const float *w;
    float *wp, *w2;
    int rows, row0, N, Ct, kk, Kp, K2p, nch, nch2, halo, s2d;      // halo: 3x3 layouts too; s2d: the stride-2 input-gradient rows
};
struct PackJobs {
    int n;
    unsigned blk0[kPackJobs + 1];
    PackJob job[kPackJobs];
};
__global__ __launch_bounds__(256) void pack_jobs_kernel(PackJobs J)
{
    int j = 0;
    while (j + 1 < J.n && blockIdx.x >= J.blk0[j + 1]) ++j;
    const PackJob &q = J.job[j];
    const size_t first = (size_t)(blockIdx.x - J.blk0[j]) * (256 * kPackIter) + threadIdx.x;
    float *wh = q.wp + (size_t)q.N * q.Kp, *w2h = q.w2 + (size_t)q.Ct * q.K2p;
#pragma unroll
    for (int it = 0; it < kPackIter; ++it) {
        const size_t idx = first + (size_t)it * 256;
        pack_weight_item(idx, q.w, q.rows, q.row0, q.N, q.Ct, q.kk, q.Kp, q.K2p, q.wp, q.w2);
        if (q.halo) {
            pack_halo_item(idx, q.w, q.rows, q.row0, q.N, q.Ct, q.nch, q.nch2, wh, w2h);
            if (q.s2d) pack_s2d_item(idx, q.w, q.rows, q.row0, q.N, q.Ct, q.nch2, w2h);
        }
    }
}

// ConvGRU state update (models/submodules.py:150) and its backward.
__global__ __launch_bounds__(256) void gru_blend_kernel(const float *__restrict__ h, const float *__restrict__ u,
                                                        const float *__restrict__ o, size_t n, float *__restrict__ out)
{
    size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n) return;
    float uu = u[i];
    out[i] = h[i] * (1.0f - uu) + o[i] * uu;
}

__global__ __launch_bounds__(256) void gru_blend_bwd_kernel(const float *__restrict__ dhn, const float *__restrict__ h,
                                                            const float *__restrict__ u, const float *__restrict__ o,
                                                            size_t n, float *__restrict__ dh, float *__restrict__ du,
                                                            float *__restrict__ dout)
{
    size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n) return;
    float g = dhn[i], uu = u[i];
    dh[i] = g * (1.0f - uu);
    du[i] = g * (o[i] - h[i]);
    dout[i] = g * uu;
}

inline int round_up(int v, int m) { return (v + m - 1) / m * m; }

template <int AM, int BM, int EPI, bool QV, bool GATED>
int launch_gemm_qv(const GemmArgs &g, int zsplits, hipStream_t st)
{
    // 8 waves per workgroup (2 x 4 wave grid): with the 75 KB of LDS of a double-buffered 128 x 128 x 32 stage two
    // workgroups fit a CU, i.e. 4 waves per SIMD — the MFMA pipe stays fed while other waves gather and stage
    if (g.rows > 64) {
        dim3 grid((g.cols + 127) / 128, (g.rows + 127) / 128, zsplits);
        hipLaunchKernelGGL((gemm_nt_kernel<512, 128, 128, 64, 32, AM, BM, EPI, QV, GATED>), grid, dim3(512), 0, st, g);
    } else if (g.rows > 32) {
        dim3 grid((g.cols + 127) / 128, (g.rows + 63) / 64, zsplits);
        hipLaunchKernelGGL((gemm_nt_kernel<512, 64, 128, 32, 32, AM, BM, EPI, QV, GATED>), grid, dim3(512), 0, st, g);
    } else {
        dim3 grid((g.cols + 127) / 128, (g.rows + 31) / 32, zsplits);
        hipLaunchKernelGGL((gemm_nt_kernel<256, 32, 128, 32, 32, AM, BM, EPI, QV, GATED>), grid, dim3(256), 0, st, g);
    }
    return tef::check_launch("gemm_nt_kernel");
}

template <int AM, int BM, int EPI>
int launch_gemm(const GemmArgs &g, int zsplits, hipStream_t st)
{
    // quad-vector gathers need stride-1 geometry and rows / images that are multiples of 4 pixels
    const Gather &G = g.G;
    bool qv = G.mul == 1 && G.div == 1 && (G.OW & 3) == 0 && ((G.OH * G.OW) & 3) == 0 && G.SH == G.OH && G.SW == G.OW;
    if (G.gate1)
        return qv ? launch_gemm_qv<AM, BM, EPI, true, true>(g, zsplits, st) : launch_gemm_qv<AM, BM, EPI, false, true>(g, zsplits, st);
    return qv ? launch_gemm_qv<AM, BM, EPI, true, false>(g, zsplits, st) : launch_gemm_qv<AM, BM, EPI, false, false>(g, zsplits, st);
}

// GEMMs with few output tiles and a long reduction (deep levels: M = B*h*w = 512 at 8x8, K up to 9216) are split over
// the reduction so that the launch covers the chip; the slabs are reduced together with the epilogue.
inline int k_splits(int rows, int cols, int K)
{
    int tr = rows > 64 ? 128 : (rows > 32 ? 64 : 32);
    int tiles = ((cols + 127) / 128) * ((rows + tr - 1) / tr);
constexpr int kConvWgTarget = 512;          // two 512-thread workgroups per CU
    if (tiles >= kConvWgTarget / 2 || K < 512) return 1;
    int z = kConvWgTarget / tiles;      // at or just below two full rounds of the chip, never just past them
    int zmax = K / 128;
    if (z > zmax) z = zmax;
    if (z > 16) z = 16;
    return z < 1 ? 1 : z;
}

struct Geo {
    int Ct, Ho, Wo, kk, K, Kp, K2, K2p, M, Mp, Min;   // M = output pixels, Min = input pixels
};

bool make_geo(const tef_conv_desc *d, Geo *q)
{
    if (!d || d->B < 1 || d->C0 < 1 || d->C1 < 0 || d->N < 1 || d->H < 1 || d->W < 1) return tef::fail("tef_conv: bad shape");
    if (d->ksize != 1 && d->ksize != 3) return tef::fail("tef_conv: kernel size must be 1 or 3");
    if (d->stride != 1 && d->stride != 2) return tef::fail("tef_conv: stride must be 1 or 2");
    int pad = d->ksize / 2;
    q->Ct = d->C0 + d->C1;
    q->Ho = (d->H + 2 * pad - d->ksize) / d->stride + 1;
    q->Wo = (d->W + 2 * pad - d->ksize) / d->stride + 1;
    q->kk = d->ksize * d->ksize;
    q->K = q->Ct * q->kk;
    q->Kp = round_up(q->K, BK);
    q->K2 = d->N * q->kk;
    q->K2p = round_up(q->K2, BK);
    q->M = d->B * q->Ho * q->Wo;
    q->Mp = round_up(q->M, BK);
    q->Min = d->B * d->H * d->W;
    return true;
}

// flow prediction heads: 1x1, stride 1, one source, at most kPwMaxN output channels
inline bool pointwise_small(const tef_conv_desc *d)
{
    return d->ksize == 1 && d->stride == 1 && d->C1 == 0 && d->N <= kPwMaxN;
}

// The halo kernel covers 3x3 stride-1 layers whose rows are 16 / 32 / 64 / 128 pixels and whose images are multiples
// of 128 pixels (a workgroup's 128 pixels are whole rows of one image).
// (also 8 x 8 images: two whole images per tile; returned as 3)
inline int halo_logw(const tef_conv_desc *d)
{
    if (d->ksize != 3 || d->stride != 1) return 0;
    if (d->W == 8 && d->H == 8 && (d->B & 1) == 0) return 3;
    if ((d->H * d->W) % 128) return 0;
    return d->W == 16 ? 4 : d->W == 32 ? 5 : d->W == 64 ? 6 : d->W == 128 ? 7 : 0;
}

// Any other 3x3 stride-1 layer whose rows are a multiple of 4 pixels runs on W x (128 / W) rectangles: returns
// log2(tile width), the widest of 128 / 64 / 32 / 16 / 8 that wastes the least of the image's columns and rows
// (30 x 40 at the deepest level of a 480 x 640 input: 8 x 16 tiles cover 94 %, 16 x 8 tiles 78 %).
inline int halo_gen_logw(const tef_conv_desc *d)
{
    if (d->ksize != 3 || d->stride != 1 || (d->W & 3) || d->W < 8 || d->H * d->W < 128) return 0;
    int best = 0;
    double best_eff = 0.0;
    for (int lw = 7; lw >= 3; --lw) {
        int tw = 1 << lw, th = 128 >> lw;
        double eff = (double)d->W / (((d->W + tw - 1) / tw) * tw) * (double)d->H / (((d->H + th - 1) / th) * th);
        if (eff > best_eff + 1e-9) { best_eff = eff; best = lw; }
    }
    return best_eff >= 0.6 ? best : 0;
}

// Row tile of the halo forward / input-gradient kernel: 128 rows, 64 when there are few of them — or few pixels (the
// 8 x 8 level: 64-row tiles double the workgroups per slice, i.e. half the slabs to write and reduce).
constexpr int kHaloTr64Cols = 0;
inline int halo_row_tile(int rows, int cols)
{
    if (rows > 64 && cols > kHaloTr64Cols) return 128;
    return rows > 32 ? 64 : 32;
}

template <int LOGW, int EPI, bool GATED, bool GEN, bool S2D = false>
int launch_halo_w(const GemmArgs &g, int z, hipStream_t st)
{
    constexpr int IPT = (LOGW == 3 && !GEN) ? 2 : 1;
    unsigned tiles = (g.cols + 127) / 128;
    if (GEN) tiles = (unsigned)((g.G.SW + (1 << LOGW) - 1) >> LOGW) * ((g.G.SH + (128 >> LOGW) - 1) / (128 >> LOGW)) *
                     (g.G.npix / (g.G.SH * g.G.SW));
    dim3 grid(tiles, 1, z);
    // A row count just past whole tiles — the decoders' input gradients: 2 prediction channels beside 64 / 128 / 256 feature
    // channels — would pay a whole extra 128-row tile (or half of its only one) for 2 rows: 33-49 % of the launch's MFMA work.
    // The whole tiles run as they are, the remainder as a second launch of 32-row tiles.
    if (!S2D && g.row_end == 0 && g.rows > 64 && g.rows % 64 > 0 && g.rows % 64 <= 32 &&
        ((g.rows - g.rows % 64) % 128 == 0 || g.rows - g.rows % 64 == 64)) {
        const int main_rows = g.rows - g.rows % 64;
        GemmArgs a = g, b = g;
        a.row_end = main_rows;
        b.row_off = main_rows; b.row_end = g.rows;
        if (main_rows == 64) {
            hipLaunchKernelGGL((conv3x3_halo_kernel<512, 64, 32, LOGW, IPT, EPI, GATED, GEN, 1, S2D>), grid, dim3(512), 0, st, a);
        } else {
            grid.y = main_rows / 128;
            hipLaunchKernelGGL((conv3x3_halo_kernel<512, 128, 64, LOGW, IPT, EPI, GATED, GEN, 1, S2D>), grid, dim3(512), 0, st, a);
        }
        if (int rc = tef::check_launch("conv3x3_halo_kernel")) return rc;
        grid.y = 1;
        if constexpr (GEN && LOGW == 3)      // (8 x 16 rectangles: the 288 halo elements need more than 256 threads)
            hipLaunchKernelGGL((conv3x3_halo_kernel<512, 64, 32, LOGW, IPT, EPI, GATED, GEN>), grid, dim3(512), 0, st, b);
        else
            hipLaunchKernelGGL((conv3x3_halo_kernel<256, 32, 32, LOGW, IPT, EPI, GATED, GEN>), grid, dim3(256), 0, st, b);
        return tef::check_launch("conv3x3_halo_kernel (row remainder)");
    }
    const int tr = halo_row_tile(g.rows, g.cols);
    if (tr == 128) {
        grid.y = (g.rows + 127) / 128;
        hipLaunchKernelGGL((conv3x3_halo_kernel<512, 128, 64, LOGW, IPT, EPI, GATED, GEN, 1, S2D>), grid, dim3(512), 0, st, g);
    } else if (tr == 64 || S2D) {
        grid.y = (g.rows + 63) / 64;
        hipLaunchKernelGGL((conv3x3_halo_kernel<512, 64, 32, LOGW, IPT, EPI, GATED, GEN, 1, S2D>), grid, dim3(512), 0, st, g);
    } else if constexpr (GEN && LOGW == 3) {     // 8 x 16 rectangles: the 288 halo elements need more than 256 threads
        grid.y = (g.rows + 63) / 64;
        hipLaunchKernelGGL((conv3x3_halo_kernel<512, 64, 32, LOGW, IPT, EPI, GATED, GEN>), grid, dim3(512), 0, st, g);
    } else {
        grid.y = (g.rows + 31) / 32;
        hipLaunchKernelGGL((conv3x3_halo_kernel<256, 32, 32, LOGW, IPT, EPI, GATED, GEN>), grid, dim3(256), 0, st, g);
    }
    return tef::check_launch("conv3x3_halo_kernel");
}

// logw: 3..7 whole-row tiles (halo_logw), 100 + 3..7 general rectangles (halo_gen_logw)
template <int EPI>
int launch_halo(const GemmArgs &g, int logw, int z, hipStream_t st)
{
    const bool gated = g.G.gate1 != nullptr;
    switch (logw) {
    case 3: return gated ? launch_halo_w<3, EPI, true, false>(g, z, st) : launch_halo_w<3, EPI, false, false>(g, z, st);
    case 4: return gated ? launch_halo_w<4, EPI, true, false>(g, z, st) : launch_halo_w<4, EPI, false, false>(g, z, st);
    case 5: return gated ? launch_halo_w<5, EPI, true, false>(g, z, st) : launch_halo_w<5, EPI, false, false>(g, z, st);
    case 6: return gated ? launch_halo_w<6, EPI, true, false>(g, z, st) : launch_halo_w<6, EPI, false, false>(g, z, st);
    case 7: return gated ? launch_halo_w<7, EPI, true, false>(g, z, st) : launch_halo_w<7, EPI, false, false>(g, z, st);
    case 103: return gated ? launch_halo_w<3, EPI, true, true>(g, z, st) : launch_halo_w<3, EPI, false, true>(g, z, st);
    case 104: return gated ? launch_halo_w<4, EPI, true, true>(g, z, st) : launch_halo_w<4, EPI, false, true>(g, z, st);
    case 105: return gated ? launch_halo_w<5, EPI, true, true>(g, z, st) : launch_halo_w<5, EPI, false, true>(g, z, st);
    case 106: return gated ? launch_halo_w<6, EPI, true, true>(g, z, st) : launch_halo_w<6, EPI, false, true>(g, z, st);
    default: return gated ? launch_halo_w<7, EPI, true, true>(g, z, st) : launch_halo_w<7, EPI, false, true>(g, z, st);
    }
}

// the stride-2 input gradient (S2D) on the same geometries, never gated
template <int EPI>
int launch_halo_s2d(const GemmArgs &g, int logw, int z, hipStream_t st)
{
    switch (logw) {
    case 3: return launch_halo_w<3, EPI, false, false, true>(g, z, st);
    case 4: return launch_halo_w<4, EPI, false, false, true>(g, z, st);
    case 5: return launch_halo_w<5, EPI, false, false, true>(g, z, st);
    case 6: return launch_halo_w<6, EPI, false, false, true>(g, z, st);
    case 7: return launch_halo_w<7, EPI, false, false, true>(g, z, st);
    case 103: return launch_halo_w<3, EPI, false, true, true>(g, z, st);
    case 104: return launch_halo_w<4, EPI, false, true, true>(g, z, st);
    case 105: return launch_halo_w<5, EPI, false, true, true>(g, z, st);
    case 106: return launch_halo_w<6, EPI, false, true, true>(g, z, st);
    default: return launch_halo_w<7, EPI, false, true, true>(g, z, st);
    }
}

// stride-2 heads (forward): even input sizes, output rows of 16 / 32 / 64 / 128 pixels in whole 128-pixel tiles
inline int halo_s2_logw(const tef_conv_desc *d)
{
    if (d->ksize != 3 || d->stride != 2 || (d->H & 1) || (d->W & 1) || d->C1 != 0) return 0;
    int Wo = d->W / 2, Ho = d->H / 2;
    if ((Ho * Wo) % 128) return 0;
    return Wo == 16 ? 4 : Wo == 32 ? 5 : Wo == 64 ? 6 : Wo == 128 ? 7 : 0;
}

template <int LOGW, int EPI>
int launch_halo_s2_w(const GemmArgs &g, int z, hipStream_t st)
{
    dim3 grid((g.cols + 127) / 128, 1, z);
    if (g.rows > 64) {
        grid.y = (g.rows + 127) / 128;
        hipLaunchKernelGGL((conv3x3_halo_kernel<512, 128, 64, LOGW, 1, EPI, false, false, 2>), grid, dim3(512), 0, st, g);
    } else if (g.rows > 32) {
        grid.y = (g.rows + 63) / 64;
        hipLaunchKernelGGL((conv3x3_halo_kernel<512, 64, 32, LOGW, 1, EPI, false, false, 2>), grid, dim3(512), 0, st, g);
    } else {
        grid.y = (g.rows + 31) / 32;
        hipLaunchKernelGGL((conv3x3_halo_kernel<256, 32, 32, LOGW, 1, EPI, false, false, 2>), grid, dim3(256), 0, st, g);
    }
    return tef::check_launch("conv3x3_halo_kernel (stride 2)");
}

template <int EPI>
int launch_halo_s2(const GemmArgs &g, int logw, int z, hipStream_t st)
{
    switch (logw) {
    case 4: return launch_halo_s2_w<4, EPI>(g, z, st);
    case 5: return launch_halo_s2_w<5, EPI>(g, z, st);
    case 6: return launch_halo_s2_w<6, EPI>(g, z, st);
    default: return launch_halo_s2_w<7, EPI>(g, z, st);
    }
}

// the halo forward / input-gradient kernel to use for this layer: 0 = none (implicit GEMM)
inline int halo_mode(const tef_conv_desc *d)
{
    if (int lw = halo_logw(d)) return lw;
    if (int lw = halo_gen_logw(d)) return 100 + lw;
    return 0;
}

// Split factor of the halo kernel over its `nch` k-chunks.  A CU finishes its workgroups at a nearly fixed rate (one
// resident workgroup already keeps its MFMA pipe busy), so a launch runs as long as its fullest CU: `wg` workgroups use
// wg / (256 * ceil(wg / 256)) of the chip (300 tiles: 59 %; measured 73 vs 97-100 TFLOP/s with 3-4 slices, and one full
// round of 256 is as fast as two, tools/zsweep.sh).  The factor maximises that fill, with a small preference for fewer
// slabs (less reduce traffic).
inline int halo_splits(const tef_conv_desc *d, int rows, int cols, int nch)
{
    int tr = halo_row_tile(rows, cols);
    int ctiles = (cols + 127) / 128;
    if (int m = halo_mode(d); m > 100) {      // general rectangles: edge tiles are partly empty
        int lw = m - 100;
        ctiles = ((d->W + (1 << lw) - 1) >> lw) * ((d->H + (128 >> lw) - 1) / (128 >> lw)) * d->B;
    }
    int tiles = ctiles * ((rows + tr - 1) / tr);
    if (nch < 8 || tiles >= 1024) return 1;
    const int zmax = std::min(nch / 2, 16);
    int best = 1;
    double best_score = -1.0;
    for (int z = 1; z <= zmax; ++z) {
        int wg = tiles * z;
        double score = (double)wg / (256.0 * ((wg + 255) / 256)) - 0.02 * z;     // fill of the rounds, fewer slabs preferred
        if (score > best_score + 1e-9) { best_score = score; best = z; }
    }
    return best;
}

// Stride-2 3x3 layers whose input gradient runs on the halo kernel (S2D): one source, even input size, at least 16
// input channels (64 rows), and the output-gradient grid is a halo geometry.  *gd: that grid as a stride-1 layer with
// N inputs and 4 Ct outputs; returns its halo mode (0: not eligible).
inline int s2d_mode(const tef_conv_desc *d, tef_conv_desc *gd)
{
    if (d->ksize != 3 || d->stride != 2 || d->C1 != 0 || (d->H & 1) || (d->W & 1) || d->C0 < 16) return 0;
    tef_conv_desc t = *d;
    t.C0 = d->N; t.C1 = 0; t.N = 4 * d->C0; t.H = d->H / 2; t.W = d->W / 2; t.stride = 1;
    if (gd) *gd = t;
    return halo_mode(&t);
}

template <int LOGW, bool GATED>
int launch_wgrad_halo_w(const GemmArgs &g, const WgradParts &wp, int z, hipStream_t st)
{
    dim3 grid((g.cols + 127) / 128, 1, z);
    if (g.rows > 64) {
        grid.y = (g.rows + 127) / 128;
        hipLaunchKernelGGL((wgrad3x3_halo_kernel<512, 128, 64, LOGW, GATED>), grid, dim3(512), 0, st, g, wp);
    } else if (g.rows > 32) {
        grid.y = (g.rows + 63) / 64;
        hipLaunchKernelGGL((wgrad3x3_halo_kernel<512, 64, 32, LOGW, GATED>), grid, dim3(512), 0, st, g, wp);
    } else {
        grid.y = (g.rows + 31) / 32;
        hipLaunchKernelGGL((wgrad3x3_halo_kernel<256, 32, 32, LOGW, GATED>), grid, dim3(256), 0, st, g, wp);
    }
    return tef::check_launch("wgrad3x3_halo_kernel");
}

// stride-2 heads: logw = log2(output width) from halo_s2_logw
inline int wgrad_s2_logw(const tef_conv_desc *d)
{
    return halo_s2_logw(d);
}

template <int LOGW>
int launch_wgrad_halo_s2_w(const GemmArgs &g, const WgradParts &wp, int z, hipStream_t st)
{
    dim3 grid((g.cols + 127) / 128, 1, z);
    if (g.rows > 64) {
        grid.y = (g.rows + 127) / 128;
        hipLaunchKernelGGL((wgrad3x3_halo_kernel<512, 128, 64, LOGW, false, 2>), grid, dim3(512), 0, st, g, wp);
    } else if (g.rows > 32) {
        grid.y = (g.rows + 63) / 64;
        hipLaunchKernelGGL((wgrad3x3_halo_kernel<512, 64, 32, LOGW, false, 2>), grid, dim3(512), 0, st, g, wp);
    } else {
        grid.y = (g.rows + 31) / 32;
        hipLaunchKernelGGL((wgrad3x3_halo_kernel<256, 32, 32, LOGW, false, 2>), grid, dim3(256), 0, st, g, wp);
    }
    return tef::check_launch("wgrad3x3_halo_kernel (stride 2)");
}

inline int launch_wgrad_halo_s2(const GemmArgs &g, const WgradParts &wp, int logw, int z, hipStream_t st)
{
    switch (logw) {
    case 4: return launch_wgrad_halo_s2_w<4>(g, wp, z, st);
    case 5: return launch_wgrad_halo_s2_w<5>(g, wp, z, st);
    case 6: return launch_wgrad_halo_s2_w<6>(g, wp, z, st);
    default: return launch_wgrad_halo_s2_w<7>(g, wp, z, st);
    }
}

inline int launch_wgrad_halo(const GemmArgs &g, const WgradParts &wp, int logw, int z, hipStream_t st)
{
    const bool gated = g.G.gate1 != nullptr;
    switch (logw) {
    case 3: return gated ? launch_wgrad_halo_w<3, true>(g, wp, z, st) : launch_wgrad_halo_w<3, false>(g, wp, z, st);
    case 4: return gated ? launch_wgrad_halo_w<4, true>(g, wp, z, st) : launch_wgrad_halo_w<4, false>(g, wp, z, st);
    case 5: return gated ? launch_wgrad_halo_w<5, true>(g, wp, z, st) : launch_wgrad_halo_w<5, false>(g, wp, z, st);
    case 6: return gated ? launch_wgrad_halo_w<6, true>(g, wp, z, st) : launch_wgrad_halo_w<6, false>(g, wp, z, st);
    default: return gated ? launch_wgrad_halo_w<7, true>(g, wp, z, st) : launch_wgrad_halo_w<7, false>(g, wp, z, st);
    }
}

struct ConvLayout {
    size_t gbuf, slab, total;
};

ConvLayout conv_layout(const tef_conv_desc *d, const Geo &q)
{
    ConvLayout L;
    size_t o = 0;
    auto take = [&](size_t n) { size_t r = o; o += (n * sizeof(float) + 255) & ~(size_t)255; return r; };
    L.gbuf = take((size_t)d->N * q.M);
    // split factors of both kernels (implicit GEMM / halo: their padded reduction lengths differ slightly)
    int kh1 = ((q.Ct + HC - 1) / HC) * HK, kh2 = ((d->N + HC - 1) / HC) * HK;
    size_t s_fwd = (size_t)std::max(k_splits(d->N, q.M, q.Kp), halo_splits(d, d->N, q.M, kh1 / HK)) * d->N * q.M;
    size_t s_bwd = (size_t)std::max(k_splits(q.Ct, q.Min, q.K2p), halo_splits(d, q.Ct, q.Min, kh2 / HK)) * q.Ct * q.Min;
    tef_conv_desc gd;
    if (s2d_mode(d, &gd)) s_bwd = std::max(s_bwd, (size_t)halo_splits(&gd, gd.N, q.M, kh2 / HK) * gd.N * q.M);
    L.slab = take(std::max(s_fwd, s_bwd));
    L.total = o;
    return L;
}

Gather forward_gather(const tef_conv_desc *d, const Geo &q, const float *x0, const float *x1, const float *gate1)
{
    Gather G{};
    G.src0 = x0; G.src1 = x1; G.gate1 = gate1;
    G.C0 = d->C0; G.C1 = d->C1; G.SH = d->H; G.SW = d->W; G.OH = q.Ho; G.OW = q.Wo;
    G.ks = d->ksize; G.pad = d->ksize / 2; G.mul = d->stride; G.div = 1; G.sgn = 1;
    G.K = q.K; G.npix = q.M; G.npix_src = (long)d->B * d->H * d->W;
    return G;
}

// dW[n][k] += sum_m g[n][m] * x_gather[m][k].  wp.n > 0: one reduction over the images of all parts (the halo kernels only).
static int conv_wgrad(const tef_conv_desc *d, const Geo &q, const float *gsrc, const float *x0, const float *x1,
                      const float *gate1, const WgradParts &wp, float *dweight, float *dweight2, int split_rows,
                      hipStream_t st)
{
    const int N = d->N, HW = q.Ho * q.Wo;
    const int nimg = wp.n ? wp.n : 1;
    const int Mtot = q.M * nimg, Mp = round_up(Mtot, BK);
    GemmArgs g{};
    g.A = gsrc; g.rows = N; g.hwA = HW;
    g.G = forward_gather(d, q, x0, x1, gate1);
    g.G.npix = Mtot;
    g.cols = q.Kp; g.K = Mp;
    g.C = dweight; g.C2 = dweight2; g.split = split_rows; g.ldc = q.K; g.valid_cols = q.K;
    int tiles = ((q.Kp + 127) / 128) * ((N + 127) / 128);
    // Slices of the pixel reduction: a launch runs ceil(workgroups / 256) rounds (one workgroup keeps a CU busy), a
    // round costs its slice's 32-pixel stages plus about three stages of prologue and atomics epilogue.
    // (32-row layers run 256-thread workgroups, one wave per SIMD each: four of them share a CU)
    const int stages = Mp / BK;
constexpr int kWgradNarrowSlots = 1024;
    // (a layer with a single output tile — the first encoder head, 64 x 18 — fills the chip only through its slices: at most
    // 64 of them left 3/4 of the CUs idle, 0.61 ms for 0.8 GFLOP per window)
    const int slots = N <= 32 ? kWgradNarrowSlots : 256, zcap = N <= 32 ? 256 : std::max(64, 256 / tiles);
    int want = 1;
    double best = 1e30;
    for (int zc = 1; zc <= zcap && zc * 4 <= std::max(4, stages); ++zc) {
        int per = (stages + zc - 1) / zc, zz = (stages + per - 1) / per;
        double cost = (double)((tiles * zz + slots - 1) / slots) * (per + 3.0);
        if (cost < best - 1e-9) { best = cost; want = zz; }
    }
    int ks = round_up((Mp + want - 1) / want, BK);
    g.ksplit = ks;
    int z = (Mp + ks - 1) / ks;
    tef::ProfScope ps(tef::PROF_CONV_WGRAD, st);
    if (int logw = halo_logw(d)) {      // reduction in 32-pixel stages; ksplit counts stages
        g.cols = q.K; g.valid_cols = q.K;
        g.ksplit = ks / 32;
        return launch_wgrad_halo(g, wp, logw, z, st);
    }
    if (int logw = wgrad_s2_logw(d)) {
        g.cols = q.K; g.valid_cols = q.K;
        g.ksplit = ks / 32;
        return launch_wgrad_halo_s2(g, wp, logw, z, st);
    }
    if (wp.n) return tef::fail("tef_conv_wgrad_parts: layer is not on the halo weight-gradient kernels"), TEF_ERR_INVALID;
    return launch_gemm<A_NCHW, B_GATHER_T, EPI_ATOMIC>(g, z, st);
}

}  // namespace

extern "C" {


size_t tef_conv_workspace_bytes(const tef_conv_desc *d)
{
    Geo q;
    if (!make_geo(d, &q)) return 0;
    return conv_layout(d, q).total;
}

size_t tef_conv_packed_weight_floats(const tef_conv_desc *d, size_t *wp_floats, size_t *w2_floats)
{
    Geo q;
    if (!make_geo(d, &q)) return 0;
    size_t np_ = (size_t)d->N * q.Kp, n2 = (size_t)q.Ct * q.K2p;
    if (d->ksize == 3) {           // + the halo-kernel layouts
        np_ += (size_t)d->N * ((q.Ct + HC - 1) / HC) * HK;
        n2 += (size_t)(s2d_mode(d, nullptr) ? 4 : 1) * q.Ct * ((d->N + HC - 1) / HC) * HK;      // S2D: four parity classes
    }
    if (wp_floats) *wp_floats = np_;
    if (w2_floats) *w2_floats = n2;
    return np_ + n2;
}

int tef_conv_pack_weight(const tef_conv_desc *d, const float *weight, int rows, int row0, float *wp, float *w2,
                         void *stream)
{
    Geo q;
    if (!make_geo(d, &q)) return TEF_ERR_INVALID;
    if (!weight || !wp || !w2 || rows < 1 || row0 < 0 || row0 + rows > d->N)
        return tef::fail("tef_conv_pack_weight: bad arguments"), TEF_ERR_INVALID;
    int c_lo = row0 * q.kk, c_hi = (row0 + rows == d->N) ? q.K2p : (row0 + rows) * q.kk;
    size_t n = std::max((size_t)rows * q.Kp, (size_t)q.Ct * (c_hi - c_lo));
    hipLaunchKernelGGL(pack_weight_kernel, dim3((unsigned)((n + 255) / 256)), dim3(256), 0, (hipStream_t)stream, weight,
                       rows, row0, d->N, q.Ct, q.kk, q.Kp, q.K2p, wp, w2);
    if (int rc = tef::check_launch("pack_weight_kernel")) return rc;
    if (d->ksize == 3) {
        int nch = (q.Ct + HC - 1) / HC, nch2 = (d->N + HC - 1) / HC;
        int n_hi = (row0 + rows == d->N) ? nch2 * HC : row0 + rows;
        size_t nh = std::max((size_t)rows * nch * HK, (size_t)q.Ct * (n_hi - row0) * 9);
        hipLaunchKernelGGL(pack_halo_kernel, dim3((unsigned)((nh + 255) / 256)), dim3(256), 0, (hipStream_t)stream, weight, rows,
                           row0, d->N, q.Ct, nch, nch2, wp + (size_t)d->N * q.Kp, w2 + (size_t)q.Ct * q.K2p);
        if (int rc = tef::check_launch("pack_halo_kernel")) return rc;
        if (s2d_mode(d, nullptr)) {      // the stride-1 input-gradient layout is unused at stride 2: the region holds the S2D rows
            size_t ns = (size_t)4 * q.Ct * (n_hi - row0) * 9;
            hipLaunchKernelGGL(pack_s2d_kernel, dim3((unsigned)((ns + 255) / 256)), dim3(256), 0, (hipStream_t)stream, weight, rows,
                               row0, d->N, q.Ct, nch2, w2 + (size_t)q.Ct * q.K2p);
            if (int rc = tef::check_launch("pack_s2d_kernel")) return rc;
        }
    }
    return 0;
}

int tef_conv_pack_weights(const tef_pack_job *jobs, int njobs, void *stream)
{
    if (njobs < 0 || (njobs && !jobs)) return tef::fail("tef_conv_pack_weights: bad arguments"), TEF_ERR_INVALID;
    for (int j0 = 0; j0 < njobs; j0 += kPackJobs) {
        PackJobs J{};
        J.n = std::min(kPackJobs, njobs - j0);
        unsigned blocks = 0;
        for (int j = 0; j < J.n; ++j) {
            const tef_pack_job &u = jobs[j0 + j];
            const tef_conv_desc *d = &u.desc;
            Geo q;
            if (!make_geo(d, &q)) return TEF_ERR_INVALID;
            if (!u.weight || !u.wp || !u.w2 || u.rows < 1 || u.row0 < 0 || u.row0 + u.rows > d->N)
                return tef::fail("tef_conv_pack_weights: bad job"), TEF_ERR_INVALID;
            PackJob &o = J.job[j];
            o.w = u.weight; o.wp = u.wp; o.w2 = u.w2; o.rows = u.rows; o.row0 = u.row0; o.N = d->N; o.Ct = q.Ct; o.kk = q.kk;
            o.Kp = q.Kp; o.K2p = q.K2p; o.nch = (q.Ct + HC - 1) / HC; o.nch2 = (d->N + HC - 1) / HC;
            o.halo = d->ksize == 3; o.s2d = o.halo && s2d_mode(d, nullptr) ? 1 : 0;
            const int c_lo = u.row0 * q.kk, c_hi = (u.row0 + u.rows == d->N) ? q.K2p : (u.row0 + u.rows) * q.kk;
            size_t n = std::max((size_t)u.rows * q.Kp, (size_t)q.Ct * (c_hi - c_lo));
            if (o.halo) {
                const int n_hi = (u.row0 + u.rows == d->N) ? o.nch2 * HC : u.row0 + u.rows;
                n = std::max(n, std::max((size_t)u.rows * o.nch * HK, (size_t)q.Ct * (n_hi - u.row0) * 9));
                if (o.s2d) n = std::max(n, (size_t)4 * q.Ct * (n_hi - u.row0) * 9);
            }
            J.blk0[j] = blocks;
            blocks += (unsigned)((n + 256 * kPackIter - 1) / (256 * kPackIter));
        }
        J.blk0[J.n] = blocks;
        hipLaunchKernelGGL(pack_jobs_kernel, dim3(blocks), dim3(256), 0, (hipStream_t)stream, J);
        if (int rc = tef::check_launch("pack_jobs_kernel")) return rc;
    }
    return 0;
}

int tef_conv_forward(const tef_conv_desc *d, const float *x0, const float *x1, const float *gate1, const float *wp,
                     const float *bias, float *out, void *workspace, size_t workspace_bytes, void *stream)
{
    return tef_conv_forward_split(d, x0, x1, gate1, wp, bias, out, nullptr, d ? d->N : 0, workspace, workspace_bytes, stream);
}

int tef_conv_forward_split(const tef_conv_desc *d, const float *x0, const float *x1, const float *gate1,
                           const float *wp, const float *bias, float *out, float *out2, int out_split, void *workspace,
                           size_t workspace_bytes, void *stream)
{
    return tef_conv_forward_blend(d, x0, x1, gate1, wp, bias, out, out2, out_split, nullptr, nullptr, nullptr, workspace,
                                  workspace_bytes, stream);
}

int tef_conv_forward_blend(const tef_conv_desc *d, const float *x0, const float *x1, const float *gate1, const float *wp,
                           const float *bias, float *out, float *out2, int out_split, const float *bl_h, const float *bl_u,
                           float *bl_out, void *workspace, size_t workspace_bytes, void *stream)
{
    Geo q;
    if (!make_geo(d, &q)) return TEF_ERR_INVALID;
    if (!x0 || (d->C1 > 0 && !x1) || !wp || !out || !workspace) return tef::fail("tef_conv_forward: null pointer"), TEF_ERR_INVALID;
    if (out_split < 1 || out_split > d->N || (out_split < d->N && !out2))
        return tef::fail("tef_conv_forward: channels beyond out_split need out2"), TEF_ERR_INVALID;
    ConvLayout L = conv_layout(d, q);
    if (workspace_bytes < L.total) return tef::fail("tef_conv_forward: workspace too small"), TEF_ERR_WORKSPACE;
    hipStream_t st = (hipStream_t)stream;
    char *ws = (char *)workspace;
    if ((bl_out != nullptr) != (bl_h != nullptr && bl_u != nullptr) || (bl_out && out_split != d->N))
        return tef::fail("tef_conv_forward_blend: bl_h, bl_u and bl_out go together, with a single output tensor"), TEF_ERR_INVALID;
    if (pointwise_small(d) && out_split == d->N && !bl_out) {
        tef::ProfScope ps(tef::PROF_CONV_FWD, st);
        hipLaunchKernelGGL(pw_fwd_kernel, dim3((unsigned)((q.M + 31) / 32)), dim3(256), 0, st, x0, wp, bias, d->B, d->C0, d->N,
                           q.Ho * q.Wo, q.Kp, d->act, out);
        return tef::check_launch("pw_fwd_kernel");
    }
    GemmArgs g{};
    g.A = wp; g.lda = q.Kp; g.rows = d->N;
    g.G = forward_gather(d, q, x0, x1, gate1);
    g.cols = q.M; g.K = q.Kp;
    g.C = out; g.C2 = out2; g.split = out_split; g.bias = bias; g.act = d->act; g.hw = q.Ho * q.Wo;
    g.bl_h = bl_h; g.bl_u = bl_u; g.bl_out = bl_out;
    tef::ProfScope ps(tef::PROF_CONV_FWD, st);
    if (int logw = halo_s2_logw(d)) {
        int nch = (q.Ct + HC - 1) / HC;
        g.A = wp + (size_t)d->N * q.Kp; g.lda = nch * HK;
        int z = halo_splits(d, d->N, q.M, nch);
        if (z <= 1) return launch_halo_s2<EPI_FWD>(g, logw, 1, st);
        float *slab = (float *)(ws + L.slab);
        g.C = slab; g.ldc = q.M; g.valid_cols = q.M;
        g.ksplit = (nch + z - 1) / z;
        z = (nch + g.ksplit - 1) / g.ksplit;
        if (int rc = launch_halo_s2<EPI_SLAB>(g, logw, z, st)) return rc;
        size_t n = (size_t)d->N * q.M;
        hipLaunchKernelGGL(splitk_reduce_kernel, dim3((unsigned)((n + 255) / 256)), dim3(256), 0, st, slab, z, d->N, q.M, bias,
                           d->act, q.Ho * q.Wo, out_split, out, out2, bl_h, bl_u, bl_out);
        return tef::check_launch("splitk_reduce_kernel");
    }
    if (int logw = halo_mode(d)) {
        int nch = (q.Ct + HC - 1) / HC;
        g.A = wp + (size_t)d->N * q.Kp; g.lda = nch * HK;
        int z = halo_splits(d, d->N, q.M, nch);
        if (z <= 1) return launch_halo<EPI_FWD>(g, logw, 1, st);
        float *slab = (float *)(ws + L.slab);
        g.C = slab; g.ldc = q.M; g.valid_cols = q.M;
        g.ksplit = (nch + z - 1) / z;
        z = (nch + g.ksplit - 1) / g.ksplit;
        if (int rc = launch_halo<EPI_SLAB>(g, logw, z, st)) return rc;
        size_t n = (size_t)d->N * q.M;
        hipLaunchKernelGGL(splitk_reduce_kernel, dim3((unsigned)((n + 255) / 256)), dim3(256), 0, st, slab, z, d->N, q.M, bias,
                           d->act, q.Ho * q.Wo, out_split, out, out2, bl_h, bl_u, bl_out);
        return tef::check_launch("splitk_reduce_kernel");
    }
    int z = k_splits(d->N, q.M, q.Kp);
    if (z == 1) return launch_gemm<A_PLAIN, B_GATHER, EPI_FWD>(g, 1, st);
    float *slab = (float *)(ws + L.slab);
    g.C = slab; g.ldc = q.M; g.valid_cols = q.M;
    g.ksplit = round_up((q.Kp + z - 1) / z, BK);
    z = (q.Kp + g.ksplit - 1) / g.ksplit;
    if (int rc = launch_gemm<A_PLAIN, B_GATHER, EPI_SLAB>(g, z, st)) return rc;
    size_t n = (size_t)d->N * q.M;
    hipLaunchKernelGGL(splitk_reduce_kernel, dim3((unsigned)((n + 255) / 256)), dim3(256), 0, st, slab, z, d->N, q.M, bias,
                       d->act, q.Ho * q.Wo, out_split, out, out2, bl_h, bl_u, bl_out);
    return tef::check_launch("splitk_reduce_kernel");
}

int tef_conv_backward(const tef_conv_desc *d, const float *x0, const float *x1, const float *gate1, const float *w2,
                      const float *out, const float *dout, float *dx0, float *dx1, float *dweight, float *dbias,
                      void *workspace, size_t workspace_bytes, void *stream)
{
    return tef_conv_backward_split(d, x0, x1, gate1, w2, out, nullptr, dout, nullptr, d ? d->N : 0, dx0, dx1, dweight, nullptr,
                                   dbias, nullptr, d ? d->N : 0, workspace, workspace_bytes, stream);
}

int tef_conv_backward_split(const tef_conv_desc *d, const float *x0, const float *x1, const float *gate1,
                            const float *w2, const float *out, const float *out2, const float *dout, const float *dout2,
                            int io_split, float *dx0, float *dx1, float *dweight, float *dweight2, float *dbias,
                            float *dbias2, int split_rows, void *workspace, size_t workspace_bytes, void *stream)
{
    return tef_conv_backward_keep(d, x0, x1, gate1, w2, out, out2, dout, dout2, io_split, dx0, dx1, dweight, dweight2, dbias,
                                  dbias2, split_rows, nullptr, workspace, workspace_bytes, stream);
}

int tef_conv_wgrad_parts_supported(const tef_conv_desc *d)
{
    Geo q;
    if (!make_geo(d, &q)) return 0;
    return (halo_logw(d) || wgrad_s2_logw(d)) ? 1 : 0;
}

int tef_conv_wgrad_parts(const tef_conv_desc *d, int nparts, const float *const *g, const float *const *x0,
                         const float *const *x1, const float *const *gate1, float *dweight, float *dweight2,
                         int split_rows, void *stream)
{
    Geo q;
    if (!make_geo(d, &q)) return TEF_ERR_INVALID;
    if (!g || !x0 || !dweight || nparts < 1 || nparts > TEF_CONV_MAX_PARTS || (d->C1 > 0 && !x1))
        return tef::fail("tef_conv_wgrad_parts: bad arguments"), TEF_ERR_INVALID;
    if (split_rows < 0 || split_rows > d->N || (split_rows < d->N && !dweight2))
        return tef::fail("tef_conv_wgrad_parts: rows beyond split_rows need dweight2"), TEF_ERR_INVALID;
    if ((long)nparts * q.M > 0x7fffffffL - 64) return tef::fail("tef_conv_wgrad_parts: too many pixels"), TEF_ERR_INVALID;
    WgradParts wp{};
    wp.n = nparts; wp.B = d->B;
    const bool gated = gate1 && gate1[0];
    for (int p = 0; p < nparts; ++p) {
        if (!g[p] || !x0[p] || (d->C1 > 0 && !x1[p]) || (gated && !gate1[p]))
            return tef::fail("tef_conv_wgrad_parts: null part"), TEF_ERR_INVALID;
        wp.A[p] = g[p]; wp.src0[p] = x0[p]; wp.src1[p] = d->C1 > 0 ? x1[p] : nullptr; wp.gate1[p] = gated ? gate1[p] : nullptr;
    }
    return conv_wgrad(d, q, wp.A[0], wp.src0[0], wp.src1[0], wp.gate1[0], wp, dweight, dweight2, split_rows, (hipStream_t)stream);
}

int tef_dec_head_backward(const tef_conv_desc *head, const float *const *dpred, int ndpred, const float *pred, const float *w2,
                          const float *dec, int dec_act, const float *dfeat, float *gp, float *gd, float *db_pred, float *dw_pred,
                          float *db_dec, void *stream)
{
    Geo q;
    if (!make_geo(head, &q)) return TEF_ERR_INVALID;
    if (!pointwise_small(head)) return tef::fail("tef_dec_head_backward: the head must be a 1x1 stride-1 convolution onto <= 4 channels"), TEF_ERR_INVALID;
    if (!dpred || ndpred < 1 || ndpred > 4 || !w2 || !dec || !gp || !gd || (head->act != TEF_ACT_NONE && !pred))
        return tef::fail("tef_dec_head_backward: null pointer / 1..4 gradient addends"), TEF_ERR_INVALID;
    DhSrc src{};
    src.n = ndpred;
    for (int k = 0; k < ndpred; ++k) {
        if (!dpred[k]) return tef::fail("tef_dec_head_backward: null gradient addend"), TEF_ERR_INVALID;
        src.p[k] = dpred[k];
    }
    hipStream_t st = (hipStream_t)stream;
    const int HW = q.Ho * q.Wo;
    uintptr_t al = (uintptr_t)pred | (uintptr_t)dec | (uintptr_t)dfeat | (uintptr_t)gp | (uintptr_t)gd;
    for (int k = 0; k < ndpred; ++k) al |= (uintptr_t)dpred[k];
    const bool v4 = (HW & 3) == 0 && (al & 15) == 0;
    dim3 grid((unsigned)((q.M + 256 * (v4 ? 4 : 1) - 1) / (256 * (v4 ? 4 : 1))), (unsigned)((head->C0 + kDhCh - 1) / kDhCh));
    {
        tef::ProfScope ps(tef::PROF_CONV_DGRAD, st);
        if (v4)
            hipLaunchKernelGGL(dec_head_bwd_kernel<4>, grid, dim3(256), 0, st, src, pred, head->act, w2, q.K2p, dec, dec_act, dfeat, head->B,
                               head->C0, head->N, HW, gp, gd, db_pred, dw_pred, db_dec);
        else
            hipLaunchKernelGGL(dec_head_bwd_kernel<1>, grid, dim3(256), 0, st, src, pred, head->act, w2, q.K2p, dec, dec_act, dfeat, head->B,
                               head->C0, head->N, HW, gp, gd, db_pred, dw_pred, db_dec);
    }
    return tef::check_launch("dec_head_bwd_kernel");
}

}  // extern "C"

namespace {
int conv_backward_impl(const tef_conv_desc *d, const float *x0, const float *x1, const float *gate1,
                       const float *w2, const float *out, const float *out2, const float *dout, const float *dout2,
                       int io_split, float *dx0, float *dx1, float *dweight, float *dweight2, float *dbias,
                       float *dbias2, int split_rows, float *g_keep, void *workspace, size_t workspace_bytes,
                       void *stream, const tef_conv_post *post);
}

extern "C" {

int tef_conv_backward_keep(const tef_conv_desc *d, const float *x0, const float *x1, const float *gate1,
                           const float *w2, const float *out, const float *out2, const float *dout, const float *dout2,
                           int io_split, float *dx0, float *dx1, float *dweight, float *dweight2, float *dbias,
                           float *dbias2, int split_rows, float *g_keep, void *workspace, size_t workspace_bytes,
                           void *stream)
{
    return conv_backward_impl(d, x0, x1, gate1, w2, out, out2, dout, dout2, io_split, dx0, dx1, dweight, dweight2, dbias, dbias2,
                              split_rows, g_keep, workspace, workspace_bytes, stream, nullptr);
}

int tef_conv_backward_post(const tef_conv_desc *d, const float *x0, const float *w2, const float *g, float *dweight,
                           const tef_conv_post *post, void *workspace, size_t workspace_bytes, void *stream)
{
    if (!d || !post || !post->g_out || (post->act != TEF_ACT_NONE && !post->mask))
        return tef::fail("tef_conv_backward_post: null pointer"), TEF_ERR_INVALID;
    if (d->C1 != 0 || d->act != TEF_ACT_NONE || pointwise_small(d))
        return tef::fail("tef_conv_backward_post: one source, g already formed (TEF_ACT_NONE), not a 1x1 head"), TEF_ERR_INVALID;
    return conv_backward_impl(d, x0, nullptr, nullptr, w2, nullptr, nullptr, g, nullptr, d->N, post->g_out, nullptr, dweight, nullptr,
                              nullptr, nullptr, d->N, nullptr, workspace, workspace_bytes, stream, post);
}

}  // extern "C"

namespace {
// `post` (tef_conv_backward_post): the input gradient goes to post->g_out and is turned there into the pre-activation
// gradient of the layer that produced the input — inside the split-K reduction when the convolution is split, by one
// in-place sweep otherwise.
int conv_backward_impl(const tef_conv_desc *d, const float *x0, const float *x1, const float *gate1,
                       const float *w2, const float *out, const float *out2, const float *dout, const float *dout2,
                       int io_split, float *dx0, float *dx1, float *dweight, float *dweight2, float *dbias,
                       float *dbias2, int split_rows, float *g_keep, void *workspace, size_t workspace_bytes,
                       void *stream, const tef_conv_post *post)
{
    Geo q;
    if (!make_geo(d, &q)) return TEF_ERR_INVALID;
    if (!x0 || (d->C1 > 0 && !x1) || !dout || !workspace) return tef::fail("tef_conv_backward: null pointer"), TEF_ERR_INVALID;
    if (d->act != TEF_ACT_NONE && !out) return tef::fail("tef_conv_backward: activation needs the forward output"), TEF_ERR_INVALID;
    if (split_rows < 0 || split_rows > d->N || (split_rows < d->N && ((dweight && !dweight2) || (dbias && !dbias2))))
        return tef::fail("tef_conv_backward: rows beyond split_rows need dweight2 / dbias2"), TEF_ERR_INVALID;
    if (io_split < 1 || io_split > d->N || (io_split < d->N && (!dout2 || (d->act != TEF_ACT_NONE && !out2))))
        return tef::fail("tef_conv_backward: channels beyond io_split need dout2 / out2"), TEF_ERR_INVALID;
    const bool need_dx = dx0 || dx1;
    if (need_dx && !w2) return tef::fail("tef_conv_backward: input gradient needs the packed weight w2"), TEF_ERR_INVALID;
    if (need_dx && ((d->C1 > 0) != (dx1 != nullptr) || !dx0))
        return tef::fail("tef_conv_backward: dx0 and dx1 must be requested together"), TEF_ERR_INVALID;
    ConvLayout L = conv_layout(d, q);
    if (workspace_bytes < L.total) return tef::fail("tef_conv_backward: workspace too small"), TEF_ERR_WORKSPACE;
    hipStream_t st = (hipStream_t)stream;
    char *ws = (char *)workspace;
    float *gbuf = g_keep ? g_keep : (float *)(ws + L.gbuf), *slab = (float *)(ws + L.slab);
    const int N = d->N, HW = q.Ho * q.Wo;
    const float *gsrc = dout;
    if (pointwise_small(d) && io_split == N && split_rows == N) {
        if (dweight || dbias) {
            tef::ProfScope ps(tef::PROF_CONV_WGRAD, st);
            int parts = std::max(1, std::min((q.M + 1023) / 1024, (1024 + d->C0 - 1) / d->C0));
            hipLaunchKernelGGL(pw_dw_kernel, dim3((unsigned)d->C0, (unsigned)parts), dim3(256), 0, st, dout, out, x0, d->B, d->C0,
                               N, HW, d->act, dweight, dbias);
            if (int rc = tef::check_launch("pw_dw_kernel")) return rc;
        }
        if (need_dx) {
            tef::ProfScope ps(tef::PROF_CONV_DGRAD, st);
            hipLaunchKernelGGL(pw_dx_kernel, dim3((unsigned)((q.M + 255) / 256), (unsigned)((d->C0 + 15) / 16)), dim3(256), 0, st,
                               dout, out, w2, d->B, d->C0, N, HW, q.K2p, d->act, dx0);
            if (int rc = tef::check_launch("pw_dx_kernel")) return rc;
        }
        return 0;
    }

    if (d->act != TEF_ACT_NONE || dbias || io_split < N) {   // g = dY * act'(out), gathered into one tensor (+ bias gradient)
        dim3 grid((unsigned)std::min<size_t>(32, ((size_t)q.M / 4 + 255) / 256 + 1), N);
        hipLaunchKernelGGL(act_bwd_kernel, grid, dim3(256), 0, st, dout, dout2, out, out2, io_split, d->act, d->B, N, HW, gbuf,
                           dbias, dbias2, split_rows);
        if (int rc = tef::check_launch("act_bwd_kernel")) return rc;
        if (d->act != TEF_ACT_NONE || io_split < N) gsrc = gbuf;
    }
    if (dweight) {   // dW[n][k] += sum_m g[n][m] * x_gather[m][k]
        WgradParts none{};
        if (int rc = conv_wgrad(d, q, gsrc, x0, x1, gate1, none, dweight, dweight2, split_rows, st)) return rc;
    }
    // post (tef_conv_backward_post): the reduction of z slabs — or, z == 0, one in-place sweep over the finished input
    // gradient — forms act'(mask) * (dx + addend) and its per-channel sums
    auto post_reduce = [&](int z) -> int {
        const int hw_in = d->H * d->W;
        const bool vec = ((q.Min | hw_in) & 3) == 0;
        const size_t n = (size_t)q.Ct * (size_t)(vec ? q.Min / 4 : q.Min);
        hipLaunchKernelGGL(splitk_reduce_post_kernel, dim3((unsigned)((n + 255) / 256)), dim3(256), 0, st, slab, z, q.Ct, q.Min, hw_in,
                           post->mask, post->act, post->addend, post->g_out, post->dbias);
        return tef::check_launch("splitk_reduce_post_kernel");
    };
    auto post_inplace = [&]() -> int { return (post && need_dx) ? post_reduce(0) : 0; };
    if (need_dx) {   // dx[ci][m'] = sum_{n,ky,kx} W[n][ci][ky][kx] * g[n][(m' + pad - k) / stride]
        GemmArgs g{};
        g.A = w2; g.lda = q.K2p; g.rows = q.Ct;
        Gather G{};
        G.src0 = gsrc; G.src1 = nullptr; G.gate1 = nullptr;
        G.C0 = N; G.C1 = 0; G.SH = q.Ho; G.SW = q.Wo; G.OH = d->H; G.OW = d->W;
        G.ks = d->ksize; G.pad = d->ksize / 2; G.mul = 1; G.div = d->stride; G.sgn = -1;
        G.K = q.K2; G.npix = q.Min; G.npix_src = (long)d->B * q.Ho * q.Wo;
        g.G = G;
        g.cols = q.Min; g.K = q.K2p;
        g.C = dx0; g.C2 = dx1; g.split = d->C0; g.bias = nullptr; g.act = TEF_ACT_NONE; g.hw = d->H * d->W;
        tef::ProfScope ps(tef::PROF_CONV_DGRAD, st);
        tef_conv_desc gd;
        if (int logw = s2d_mode(d, &gd)) {   // stride 2: four parity classes over the output-gradient grid (S2D)
            Geo gq;
            if (!make_geo(&gd, &gq)) return TEF_ERR_INVALID;
            const int nch2 = (N + HC - 1) / HC;
            GemmArgs s{};
            s.A = w2 + (size_t)q.Ct * q.K2p; s.lda = nch2 * HK; s.rows = 4 * q.Ct;
            s.G = forward_gather(&gd, gq, gsrc, nullptr, nullptr);
            s.cols = q.M; s.K = gq.Kp;
            s.C = dx0; s.C2 = nullptr; s.split = 4 * q.Ct; s.bias = nullptr; s.act = TEF_ACT_NONE; s.hw = q.Ho * q.Wo;
            s.s2d_ct = q.Ct;
            int z = halo_splits(&gd, 4 * q.Ct, q.M, nch2);
            if (z <= 1) {
                if (int rc = launch_halo_s2d<EPI_FWD>(s, logw, 1, st)) return rc;
                return post_inplace();
            }
            s.C = slab; s.ldc = q.M; s.valid_cols = q.M;
            s.ksplit = (nch2 + z - 1) / z;
            z = (nch2 + s.ksplit - 1) / s.ksplit;
            if (int rc = launch_halo_s2d<EPI_SLAB>(s, logw, z, st)) return rc;
            size_t n = (size_t)2 * q.Ct * q.M;
            hipLaunchKernelGGL(splitk_reduce_s2d_kernel, dim3((unsigned)((n + 255) / 256)), dim3(256), 0, st, slab, z, q.Ct, q.M,
                               q.Ho, q.Wo, dx0);
            if (int rc = tef::check_launch("splitk_reduce_s2d_kernel")) return rc;
            return post_inplace();
        }
        if (int logw = halo_mode(d)) {      // stride 1: the gradient grid is the input grid; taps were flipped at pack time
            int nch2 = (N + HC - 1) / HC;
            g.A = w2 + (size_t)q.Ct * q.K2p; g.lda = nch2 * HK;
            int z = halo_splits(d, q.Ct, q.Min, nch2);
            if (z <= 1) {
                if (int rc = launch_halo<EPI_FWD>(g, logw, 1, st)) return rc;
                return post_inplace();
            }
            g.C = slab; g.ldc = q.Min; g.valid_cols = q.Min;
            g.ksplit = (nch2 + z - 1) / z;
            z = (nch2 + g.ksplit - 1) / g.ksplit;
            if (int rc = launch_halo<EPI_SLAB>(g, logw, z, st)) return rc;
            size_t n = (size_t)q.Ct * q.Min;
            if (post) return post_reduce(z);
            hipLaunchKernelGGL(splitk_reduce_kernel, dim3((unsigned)((n + 255) / 256)), dim3(256), 0, st, slab, z, q.Ct, q.Min,
                               (const float *)nullptr, TEF_ACT_NONE, d->H * d->W, d->C0, dx0, dx1, (const float *)nullptr, (const float *)nullptr,
                               (float *)nullptr);
            return tef::check_launch("splitk_reduce_kernel");
        }
        int z = k_splits(q.Ct, q.Min, q.K2p);
        if (z == 1) {
            if (int rc = launch_gemm<A_PLAIN, B_GATHER, EPI_FWD>(g, 1, st)) return rc;
        } else {
            g.C = slab; g.ldc = q.Min; g.valid_cols = q.Min;
            g.ksplit = round_up((q.K2p + z - 1) / z, BK);
            z = (q.K2p + g.ksplit - 1) / g.ksplit;
            if (int rc = launch_gemm<A_PLAIN, B_GATHER, EPI_SLAB>(g, z, st)) return rc;
            size_t n = (size_t)q.Ct * q.Min;
            if (post) return post_reduce(z);
            hipLaunchKernelGGL(splitk_reduce_kernel, dim3((unsigned)((n + 255) / 256)), dim3(256), 0, st, slab, z, q.Ct, q.Min,
                               (const float *)nullptr, TEF_ACT_NONE, d->H * d->W, d->C0, dx0, dx1, (const float *)nullptr, (const float *)nullptr,
                               (float *)nullptr);
            if (int rc = tef::check_launch("splitk_reduce_kernel")) return rc;
        }
    }
    return post_inplace();
}
}  // namespace

extern "C" {

int tef_gru_blend(const float *h, const float *u, const float *o, size_t n, float *out, void *stream)
{
    if (!h || !u || !o || !out) return tef::fail("tef_gru_blend: null pointer"), TEF_ERR_INVALID;
    hipLaunchKernelGGL(gru_blend_kernel, dim3((unsigned)((n + 255) / 256)), dim3(256), 0, (hipStream_t)stream, h, u, o, n, out);
    return tef::check_launch("gru_blend_kernel");
}

int tef_gru_blend_backward(const float *dhn, const float *h, const float *u, const float *o, size_t n, float *dh,
                           float *du, float *dout, void *stream)
{
    if (!dhn || !h || !u || !o || !dh || !du || !dout) return tef::fail("tef_gru_blend_backward: null pointer"), TEF_ERR_INVALID;
    hipLaunchKernelGGL(gru_blend_bwd_kernel, dim3((unsigned)((n + 255) / 256)), dim3(256), 0, (hipStream_t)stream, dhn, h, u,
                       o, n, dh, du, dout);
    return tef::check_launch("gru_blend_bwd_kernel");
}

}  // extern "C"
