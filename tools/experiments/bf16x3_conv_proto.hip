// Prototype (not product): a 3x3 stride-1 convolution forward on error-compensated bf16 splits, to see what the matrix cores
// give once the operands come from a halo patch instead of a plain GEMM tile (tools/bf16x3_gemm_bench.hip: 3.0x the fp32 MFMA
// kernel of the same structure, 4e-6 of max |C|).  One layer shape: B x Cin x H x 64 -> B x Cout x H x 64, padding 1, bias +
// relu, fp32 NCHW in and out — with B = 8, Cin = Cout = 128, H = 64 the update|reset gate GEMM of the first ConvGRU
// (9.66 GFLOP; csrc/tef_conv.hip: 0.095 ms = 102 TFLOP/s).
//   x = hi + lo, hi = bf16(x), lo = bf16(x - hi);  out += w_lo x_hi + w_hi x_lo + w_hi x_hi on v_mfma_f32_32x32x16_bf16.
//   Weights are split once (pack kernel: [row][chunk of 16 channels][tap][16] as hi and lo); activations when they are staged.
//   Workgroup: 128 output channels x 128 pixels (two image rows), four wavefronts of 64 x 64; k-step = one tap x 16 channels.
//   LDS: patch [hi|lo][channel half][4 x 66 pixels][8 channels] (a lane's 8 channels = one 16-byte read, consecutive lanes
//   consecutive pixels), weights of three taps [hi|lo][tap][channel half][128 rows][8], double buffered.
// Build + run:  hipcc --offload-arch=gfx950 -O3 tools/bf16x3_conv_proto.hip -o /tmp/bf16x3_conv_proto && /tmp/bf16x3_conv_proto
#include <hip/hip_runtime.h>
#include <stdio.h>
#include <stdlib.h>
#include <math.h>
#include <vector>

typedef __bf16 bf16x8 __attribute__((ext_vector_type(8)));
typedef __bf16 bf16x4 __attribute__((ext_vector_type(4)));
typedef float f32x16 __attribute__((ext_vector_type(16)));

constexpr int W = 64, PW = W + 2, PROWS = 4, NPIX = PROWS * PW;      // patch: rows y0 - 1 .. y0 + 2, columns -1 .. 64
constexpr int CH = 16;                                              // channels per chunk
constexpr int TR = 128;                                             // output channels per workgroup

// packed weights: [row][chunk][tap][16] bf16, hi plane then lo plane
__global__ void pack_weights(const float *__restrict__ w, int Cout, int Cin, __bf16 *__restrict__ hi, __bf16 *__restrict__ lo)
{
    const int idx = blockIdx.x * blockDim.x + threadIdx.x;
    if (idx >= Cout * Cin * 9) return;
    const int c16 = idx % CH, tap = (idx / CH) % 9, chunk = (idx / (CH * 9)) % (Cin / CH), row = idx / (CH * 9 * (Cin / CH));
    const float v = w[((size_t)row * Cin + chunk * CH + c16) * 9 + tap];
    const __bf16 h = (__bf16)v;
    hi[idx] = h;
    lo[idx] = (__bf16)(v - (float)h);
}

__global__ __launch_bounds__(256, 2) void conv3x3_bf16x3(const float *__restrict__ x, const __bf16 *__restrict__ whi,
                                                          const __bf16 *__restrict__ wlo, const float *__restrict__ bias,
                                                          float *__restrict__ y, int B, int Cin, int Cout, int H)
{
    // patch planes: [hi|lo][half][NPIX][8]; weight sub-stage (3 taps): [buf][hi|lo][tap][half][TR][8]
    extern __shared__ __attribute__((aligned(16))) unsigned char lds_raw[];      // 66 KiB: above the static limit
    typedef __bf16 (*PatchT)[2][NPIX][8];
    typedef __bf16 (*WeightT)[2][3][2][TR][8];
    PatchT P = reinterpret_cast<PatchT>(lds_raw);
    WeightT Ws = reinterpret_cast<WeightT>(lds_raw + sizeof(__bf16) * 2 * 2 * NPIX * 8);
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int wr = wave >> 1, wc = wave & 1, h = lane >> 5;
    const int row0 = blockIdx.y * TR;
    const int tile = blockIdx.x, rows_per_img = H / 2, img = tile / rows_per_img, y0 = (tile - img * rows_per_img) * 2;
    const int nch = Cin / CH;
    const size_t HW = (size_t)H * W;

    // ---- patch staging: item = (pixel, group of 4 channels); 4 scalar loads -> 4 hi + 4 lo -> two 8-byte LDS writes ----
    constexpr int PITEMS = NPIX * 4, PPT = (PITEMS + 255) / 256;
    float pv[PPT][4];
    auto load_patch = [&](int chunk) {
#pragma unroll
        for (int p = 0; p < PPT; ++p) {
            const int item = min(tid + p * 256, PITEMS - 1), pix = item % NPIX, cg = item / NPIX;
            const int pr = pix / PW, px = pix - pr * PW, yy = y0 - 1 + pr, xx = px - 1;
            const bool in = yy >= 0 && yy < H && xx >= 0 && xx < W;
            const float *src = x + ((size_t)img * Cin + chunk * CH + cg * 4) * HW + (in ? (size_t)yy * W + xx : 0);
#pragma unroll
            for (int j = 0; j < 4; ++j) {
                const float v = src[(size_t)j * HW];
                pv[p][j] = in ? v : 0.0f;
            }
        }
    };
    auto store_patch = [&]() {
#pragma unroll
        for (int p = 0; p < PPT; ++p) {
            const int item = tid + p * 256;
            if (item < PITEMS) {
                const int pix = item % NPIX, cg = item / NPIX;
                bf16x4 hh = {(__bf16)pv[p][0], (__bf16)pv[p][1], (__bf16)pv[p][2], (__bf16)pv[p][3]};
                bf16x4 ll = {(__bf16)(pv[p][0] - (float)hh[0]), (__bf16)(pv[p][1] - (float)hh[1]), (__bf16)(pv[p][2] - (float)hh[2]),
                             (__bf16)(pv[p][3] - (float)hh[3])};
                *reinterpret_cast<bf16x4 *>(&P[0][cg >> 1][pix][(cg & 1) * 4]) = hh;
                *reinterpret_cast<bf16x4 *>(&P[1][cg >> 1][pix][(cg & 1) * 4]) = ll;
            }
        }
    };
    // ---- weight staging: sub-stage = 3 taps x 16 channels of 128 rows, hi and lo: 128 x 3 x 2 x 2 = 1536 16-byte pieces ----
    constexpr int WPT = 1536 / 256;
    bf16x8 wv[WPT];
    auto load_w = [&](int chunk, int sub) {
#pragma unroll
        for (int p = 0; p < WPT; ++p) {
            const int piece = tid + p * 256;                 // (plane, row, tap, half)
            const int half = piece & 1, tap = (piece >> 1) % 3, row = (piece / 6) % TR, plane = piece / (6 * TR);
            const __bf16 *src = (plane ? wlo : whi) + (((size_t)(row0 + row) * nch + chunk) * 9 + sub * 3 + tap) * CH + half * 8;
            wv[p] = *reinterpret_cast<const bf16x8 *>(src);
        }
    };
    auto store_w = [&](int buf) {
#pragma unroll
        for (int p = 0; p < WPT; ++p) {
            const int piece = tid + p * 256;
            const int half = piece & 1, tap = (piece >> 1) % 3, row = (piece / 6) % TR, plane = piece / (6 * TR);
            *reinterpret_cast<bf16x8 *>(&Ws[buf][plane][tap][half][row][0]) = wv[p];
        }
    };

    f32x16 acc[2][2];
#pragma unroll
    for (int i = 0; i < 2; ++i)
#pragma unroll
        for (int j = 0; j < 2; ++j)
#pragma unroll
            for (int e = 0; e < 16; ++e) acc[i][j][e] = 0.0f;
    // the lane's two pixels (column tiles j = 0, 1 of the wavefront's 64): patch index of the pixel at tap (0, 0)
    int pbase[2];
#pragma unroll
    for (int j = 0; j < 2; ++j) {
        const int pl = wc * 64 + j * 32 + (lane & 31);      // pixel of the tile: row pl / 64, column pl % 64
        pbase[j] = (pl >> 6) * PW + (pl & 63);              // patch (row + ky, column + kx) = pbase + ky * PW + kx
    }
    auto taps3 = [&](int buf, int sub) {
#pragma unroll
        for (int t = 0; t < 3; ++t) {
            const int tap = sub * 3 + t, ky = tap / 3, kx = tap - ky * 3;
            bf16x8 ah[2], al[2], bh[2], bl[2];
#pragma unroll
            for (int i = 0; i < 2; ++i) {
                const int r = wr * 64 + i * 32 + (lane & 31);
                ah[i] = *reinterpret_cast<const bf16x8 *>(&Ws[buf][0][t][h][r][0]);
                al[i] = *reinterpret_cast<const bf16x8 *>(&Ws[buf][1][t][h][r][0]);
            }
#pragma unroll
            for (int j = 0; j < 2; ++j) {
                const int pi = pbase[j] + ky * PW + kx;
                bh[j] = *reinterpret_cast<const bf16x8 *>(&P[0][h][pi][0]);
                bl[j] = *reinterpret_cast<const bf16x8 *>(&P[1][h][pi][0]);
            }
#pragma unroll
            for (int i = 0; i < 2; ++i)
#pragma unroll
                for (int j = 0; j < 2; ++j) {
                    acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(al[i], bh[j], acc[i][j], 0, 0, 0);
                    acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(ah[i], bl[j], acc[i][j], 0, 0, 0);
                    acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(ah[i], bh[j], acc[i][j], 0, 0, 0);
                }
        }
    };

    load_patch(0);
    load_w(0, 0);
    store_patch();
    store_w(0);
    __syncthreads();
    int wb = 0;
    for (int c = 0; c < nch; ++c) {
#pragma unroll
        for (int sub = 0; sub < 3; ++sub) {
            const bool last = c == nch - 1 && sub == 2;
            const int nc = sub == 2 ? c + 1 : c, ns = sub == 2 ? 0 : sub + 1;
            if (!last) load_w(nc, ns);                       // next sub-stage's weights in flight under the matrix work
            if (sub == 2 && !last) load_patch(c + 1);
            taps3(wb, sub);
            if (sub == 2) __syncthreads();                   // everybody has read the patch: it may be replaced
            if (!last) store_w(wb ^ 1);
            if (sub == 2 && !last) store_patch();
            __syncthreads();
            wb ^= 1;
        }
    }
    // epilogue: bias + relu, NCHW.  C/D layout: col = lane & 31, row = (e & 3) + 8 * (e >> 2) + 4 * (lane >> 5)
#pragma unroll
    for (int i = 0; i < 2; ++i)
#pragma unroll
        for (int j = 0; j < 2; ++j) {
            const int pl = wc * 64 + j * 32 + (lane & 31);
            const size_t po = (size_t)(y0 + (pl >> 6)) * W + (pl & 63);
#pragma unroll
            for (int e = 0; e < 16; ++e) {
                const int r = row0 + wr * 64 + i * 32 + (e & 3) + 8 * (e >> 2) + 4 * h;
                const float v = acc[i][j][e] + bias[r];
                y[((size_t)img * Cout + r) * HW + po] = v > 0.0f ? v : 0.0f;
            }
        }
}

int main()
{
    const int B = 8, Cin = 128, Cout = 128, H = 64;
    const size_t nx = (size_t)B * Cin * H * W, ny = (size_t)B * Cout * H * W, nw = (size_t)Cout * Cin * 9;
    std::vector<float> hx(nx), hw(nw), hb(Cout), hy(ny);
    srand(2);
    for (auto &v : hx) v = (float)rand() / RAND_MAX - 0.3f;
    for (auto &v : hw) v = ((float)rand() / RAND_MAX - 0.5f) * 0.05f;
    for (auto &v : hb) v = (float)rand() / RAND_MAX - 0.5f;
    float *dx, *dw, *db, *dy;
    __bf16 *dhi, *dlo;
    hipMalloc(&dx, nx * 4); hipMalloc(&dw, nw * 4); hipMalloc(&db, Cout * 4); hipMalloc(&dy, ny * 4);
    hipMalloc(&dhi, nw * 2); hipMalloc(&dlo, nw * 2);
    hipMemcpy(dx, hx.data(), nx * 4, hipMemcpyHostToDevice);
    hipMemcpy(dw, hw.data(), nw * 4, hipMemcpyHostToDevice);
    hipMemcpy(db, hb.data(), Cout * 4, hipMemcpyHostToDevice);
    hipLaunchKernelGGL(pack_weights, dim3((unsigned)((nw + 255) / 256)), dim3(256), 0, 0, dw, Cout, Cin, dhi, dlo);
    dim3 grid(B * H / 2, Cout / TR);
    const size_t lds = sizeof(__bf16) * (2 * 2 * NPIX * 8 + 2 * 2 * 3 * 2 * TR * 8);
    hipFuncSetAttribute((const void *)conv3x3_bf16x3, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
    hipEvent_t a, b; hipEventCreate(&a); hipEventCreate(&b);
    for (int rep = 0; rep < 3; ++rep) hipLaunchKernelGGL(conv3x3_bf16x3, grid, dim3(256), lds, 0, dx, dhi, dlo, db, dy, B, Cin, Cout, H);
    hipEventRecord(a);
    const int reps = 50;
    for (int rep = 0; rep < reps; ++rep) hipLaunchKernelGGL(conv3x3_bf16x3, grid, dim3(256), lds, 0, dx, dhi, dlo, db, dy, B, Cin, Cout, H);
    hipEventRecord(b); hipEventSynchronize(b);
    float ms; hipEventElapsedTime(&ms, a, b); ms /= reps;
    hipMemcpy(hy.data(), dy, ny * 4, hipMemcpyDeviceToHost);
    double worst = 0.0, scale = 0.0;
    for (int s = 0; s < 3000; ++s) {
        const int img = rand() % B, n = rand() % Cout, yy = s < 200 ? (s & 1 ? 0 : H - 1) : rand() % H, xx = s < 200 ? (s & 2 ? 0 : W - 1) : rand() % W;
        double ref = hb[n];
        for (int c = 0; c < Cin; ++c)
            for (int ky = 0; ky < 3; ++ky)
                for (int kx = 0; kx < 3; ++kx) {
                    const int sy = yy + ky - 1, sx = xx + kx - 1;
                    if (sy < 0 || sy >= H || sx < 0 || sx >= W) continue;
                    ref += (double)hw[((size_t)n * Cin + c) * 9 + ky * 3 + kx] * (double)hx[((size_t)img * Cin + c) * H * W + (size_t)sy * W + sx];
                }
        ref = ref > 0.0 ? ref : 0.0;
        worst = fmax(worst, fabs((double)hy[((size_t)img * Cout + n) * H * W + (size_t)yy * W + xx] - ref));
        scale = fmax(scale, fabs(ref));
    }
    const double gflop = 2.0 * B * H * W * (double)Cout * Cin * 9 / 1e9;
    printf("conv 3x3 %dx%dx%dx%d -> %d channels (%.2f GFLOP): %.4f ms = %.1f TFLOP/s (fp32-equivalent); max |err| / max |out| over 3000 outputs "
           "(200 of them on the border): %.2e\n", B, Cin, H, W, Cout, gflop, ms, gflop / ms, worst / scale);
    return 0;
}
