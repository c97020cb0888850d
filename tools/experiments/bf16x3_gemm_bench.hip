// Micro-benchmark: what would error-compensated bf16 splits buy the convolution contractions?  ("what
// is left".)  C[M][N] = sum_k A[M][K] * B[N][K] in fp32 in / fp32 out, computed
//   (0) on v_mfma_f32_32x32x2_f32 (what csrc/tef_conv.hip uses: exact fp32 products), and
//   (1) as three bf16 products per k-step on v_mfma_f32_32x32x16_bf16: x = hi + lo with hi = bf16(x), lo = bf16(x - hi);
//       C += a_lo b_hi + a_hi b_lo + a_hi b_hi  (the lo * lo term, 2^-16 of a product, is dropped),
// both with the same structure: 128 x 128 workgroup tiles, four wavefronts with 64 x 64 tiles, k-chunks of 32 staged in LDS
// (double buffered, the next chunk's global loads in flight under the matrix work; the split happens when the registers go to
// LDS), operands read with ds_read_b128.  Prints the time, fp32-equivalent TFLOP/s and the error against a float64 sample.
// Build + run:  hipcc --offload-arch=gfx950 -O3 tools/bf16x3_gemm_bench.hip -o /tmp/bf16x3_gemm_bench && /tmp/bf16x3_gemm_bench
#include <hip/hip_runtime.h>
#include <stdio.h>
#include <stdlib.h>
#include <math.h>
#include <vector>

typedef __bf16 bf16x8 __attribute__((ext_vector_type(8)));
typedef __bf16 bf16x4 __attribute__((ext_vector_type(4)));
typedef float f32x16 __attribute__((ext_vector_type(16)));

constexpr int TM = 128, TN = 128, BK = 32;

// ---- (1) bf16 x 3 ---------------------------------------------------------------------------------------------------------
constexpr int LDH = BK + 8;      // bf16 elements per LDS row (80 bytes: 16-byte aligned, rows spread over the banks)
__global__ __launch_bounds__(256) void gemm_bf16x3(const float *__restrict__ A, const float *__restrict__ B, float *__restrict__ C,
                                                   int M, int N, int K)
{
    __shared__ __attribute__((aligned(16))) __bf16 Ah[2][TM][LDH], Al[2][TM][LDH], Bh[2][TN][LDH], Bl[2][TN][LDH];
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int wr = wave >> 1, wc = wave & 1;                 // 2 x 2 wavefronts of 64 x 64
    const int row0 = blockIdx.y * TM, col0 = blockIdx.x * TN;
    // staging: 128 rows x 8 float4 per operand = 1024 pieces, 4 per thread
    float4 ra[4], rb[4];
    auto load = [&](int k0) {
#pragma unroll
        for (int p = 0; p < 4; ++p) {
            const int piece = tid + p * 256, r = piece >> 3, q = (piece & 7) * 4;
            ra[p] = *reinterpret_cast<const float4 *>(A + (size_t)(row0 + r) * K + k0 + q);
            rb[p] = *reinterpret_cast<const float4 *>(B + (size_t)(col0 + r) * K + k0 + q);
        }
    };
    auto split_store = [&](float4 v, __bf16 *hi, __bf16 *lo) {
        bf16x4 h = {(__bf16)v.x, (__bf16)v.y, (__bf16)v.z, (__bf16)v.w};
        bf16x4 l = {(__bf16)(v.x - (float)h[0]), (__bf16)(v.y - (float)h[1]), (__bf16)(v.z - (float)h[2]), (__bf16)(v.w - (float)h[3])};
        *reinterpret_cast<bf16x4 *>(hi) = h;
        *reinterpret_cast<bf16x4 *>(lo) = l;
    };
    auto store = [&](int buf) {
#pragma unroll
        for (int p = 0; p < 4; ++p) {
            const int piece = tid + p * 256, r = piece >> 3, q = (piece & 7) * 4;
            split_store(ra[p], &Ah[buf][r][q], &Al[buf][r][q]);
            split_store(rb[p], &Bh[buf][r][q], &Bl[buf][r][q]);
        }
    };
    f32x16 acc[2][2];
#pragma unroll
    for (int i = 0; i < 2; ++i)
#pragma unroll
        for (int j = 0; j < 2; ++j)
#pragma unroll
            for (int e = 0; e < 16; ++e) acc[i][j][e] = 0.0f;
    auto multiply = [&](int buf) {
#pragma unroll
        for (int ks = 0; ks < BK; ks += 16) {
            // 32x32x16: lane l holds row (l & 31), k = 8 * (l >> 5) .. + 7 of the 16
            const int ko = ks + 8 * (lane >> 5);
            bf16x8 ah[2], al[2], bh[2], bl[2];
#pragma unroll
            for (int i = 0; i < 2; ++i) {
                const int r = wr * 64 + i * 32 + (lane & 31);
                ah[i] = *reinterpret_cast<const bf16x8 *>(&Ah[buf][r][ko]);
                al[i] = *reinterpret_cast<const bf16x8 *>(&Al[buf][r][ko]);
            }
#pragma unroll
            for (int j = 0; j < 2; ++j) {
                const int c = wc * 64 + j * 32 + (lane & 31);
                bh[j] = *reinterpret_cast<const bf16x8 *>(&Bh[buf][c][ko]);
                bl[j] = *reinterpret_cast<const bf16x8 *>(&Bl[buf][c][ko]);
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
    load(0);
    store(0);
    __syncthreads();
    int buf = 0;
    for (int k0 = BK; k0 < K; k0 += BK) {
        load(k0);
        multiply(buf);
        store(buf ^ 1);
        __syncthreads();
        buf ^= 1;
    }
    multiply(buf);
    // C/D layout of the 32x32 tiles: col = lane & 31, row = (e & 3) + 8 * (e >> 2) + 4 * (lane >> 5)
#pragma unroll
    for (int i = 0; i < 2; ++i)
#pragma unroll
        for (int j = 0; j < 2; ++j)
#pragma unroll
            for (int e = 0; e < 16; ++e) {
                const int r = row0 + wr * 64 + i * 32 + (e & 3) + 8 * (e >> 2) + 4 * (lane >> 5);
                const int c = col0 + wc * 64 + j * 32 + (lane & 31);
                C[(size_t)r * N + c] = acc[i][j][e];
            }
}

// ---- (2) three-way split (hi + mid + lo = 24 bits), six products: exact-fp32-level products ------------------------------------
__global__ __launch_bounds__(256) void gemm_bf16x6(const float *__restrict__ A, const float *__restrict__ B, float *__restrict__ C,
                                                   int M, int N, int K)
{
    __shared__ __attribute__((aligned(16))) __bf16 Ah[2][TM][LDH], Am[2][TM][LDH], Al[2][TM][LDH], Bh[2][TN][LDH], Bm[2][TN][LDH], Bl[2][TN][LDH];
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int wr = wave >> 1, wc = wave & 1;                 // 2 x 2 wavefronts of 64 x 64
    const int row0 = blockIdx.y * TM, col0 = blockIdx.x * TN;
    // staging: 128 rows x 8 float4 per operand = 1024 pieces, 4 per thread
    float4 ra[4], rb[4];
    auto load = [&](int k0) {
#pragma unroll
        for (int p = 0; p < 4; ++p) {
            const int piece = tid + p * 256, r = piece >> 3, q = (piece & 7) * 4;
            ra[p] = *reinterpret_cast<const float4 *>(A + (size_t)(row0 + r) * K + k0 + q);
            rb[p] = *reinterpret_cast<const float4 *>(B + (size_t)(col0 + r) * K + k0 + q);
        }
    };
    auto split_store = [&](float4 v, __bf16 *hi, __bf16 *mid, __bf16 *lo) {
        float x[4] = {v.x, v.y, v.z, v.w};
        bf16x4 h, m, l;
#pragma unroll
        for (int e = 0; e < 4; ++e) {
            h[e] = (__bf16)x[e];
            const float r1 = x[e] - (float)h[e];
            m[e] = (__bf16)r1;
            l[e] = (__bf16)(r1 - (float)m[e]);
        }
        *reinterpret_cast<bf16x4 *>(hi) = h;
        *reinterpret_cast<bf16x4 *>(mid) = m;
        *reinterpret_cast<bf16x4 *>(lo) = l;
    };
    auto store = [&](int buf) {
#pragma unroll
        for (int p = 0; p < 4; ++p) {
            const int piece = tid + p * 256, r = piece >> 3, q = (piece & 7) * 4;
            split_store(ra[p], &Ah[buf][r][q], &Am[buf][r][q], &Al[buf][r][q]);
            split_store(rb[p], &Bh[buf][r][q], &Bm[buf][r][q], &Bl[buf][r][q]);
        }
    };
    f32x16 acc[2][2];
#pragma unroll
    for (int i = 0; i < 2; ++i)
#pragma unroll
        for (int j = 0; j < 2; ++j)
#pragma unroll
            for (int e = 0; e < 16; ++e) acc[i][j][e] = 0.0f;
    auto multiply = [&](int buf) {
#pragma unroll
        for (int ks = 0; ks < BK; ks += 16) {
            // 32x32x16: lane l holds row (l & 31), k = 8 * (l >> 5) .. + 7 of the 16
            const int ko = ks + 8 * (lane >> 5);
            bf16x8 ah[2], am[2], al[2], bh[2], bm[2], bl[2];
#pragma unroll
            for (int i = 0; i < 2; ++i) {
                const int r = wr * 64 + i * 32 + (lane & 31);
                ah[i] = *reinterpret_cast<const bf16x8 *>(&Ah[buf][r][ko]);
                am[i] = *reinterpret_cast<const bf16x8 *>(&Am[buf][r][ko]);
                al[i] = *reinterpret_cast<const bf16x8 *>(&Al[buf][r][ko]);
            }
#pragma unroll
            for (int j = 0; j < 2; ++j) {
                const int c = wc * 64 + j * 32 + (lane & 31);
                bh[j] = *reinterpret_cast<const bf16x8 *>(&Bh[buf][c][ko]);
                bm[j] = *reinterpret_cast<const bf16x8 *>(&Bm[buf][c][ko]);
                bl[j] = *reinterpret_cast<const bf16x8 *>(&Bl[buf][c][ko]);
            }
#pragma unroll
            for (int i = 0; i < 2; ++i)
#pragma unroll
                for (int j = 0; j < 2; ++j) {
                    acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(al[i], bh[j], acc[i][j], 0, 0, 0);
                    acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(ah[i], bl[j], acc[i][j], 0, 0, 0);
                    acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(am[i], bm[j], acc[i][j], 0, 0, 0);
                    acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(am[i], bh[j], acc[i][j], 0, 0, 0);
                    acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(ah[i], bm[j], acc[i][j], 0, 0, 0);
                    acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(ah[i], bh[j], acc[i][j], 0, 0, 0);
                }
        }
    };
    load(0);
    store(0);
    __syncthreads();
    int buf = 0;
    for (int k0 = BK; k0 < K; k0 += BK) {
        load(k0);
        multiply(buf);
        store(buf ^ 1);
        __syncthreads();
        buf ^= 1;
    }
    multiply(buf);
    // C/D layout of the 32x32 tiles: col = lane & 31, row = (e & 3) + 8 * (e >> 2) + 4 * (lane >> 5)
#pragma unroll
    for (int i = 0; i < 2; ++i)
#pragma unroll
        for (int j = 0; j < 2; ++j)
#pragma unroll
            for (int e = 0; e < 16; ++e) {
                const int r = row0 + wr * 64 + i * 32 + (e & 3) + 8 * (e >> 2) + 4 * (lane >> 5);
                const int c = col0 + wc * 64 + j * 32 + (lane & 31);
                C[(size_t)r * N + c] = acc[i][j][e];
            }
}

// ---- (0) fp32 MFMA, same structure ------------------------------------------------------------------------------------------
constexpr int LDF = BK + 4;
__global__ __launch_bounds__(256) void gemm_f32(const float *__restrict__ A, const float *__restrict__ B, float *__restrict__ C,
                                                int M, int N, int K)
{
    __shared__ __attribute__((aligned(16))) float As[2][TM][LDF], Bs[2][TN][LDF];
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int wr = wave >> 1, wc = wave & 1;
    const int row0 = blockIdx.y * TM, col0 = blockIdx.x * TN;
    float4 ra[4], rb[4];
    auto load = [&](int k0) {
#pragma unroll
        for (int p = 0; p < 4; ++p) {
            const int piece = tid + p * 256, r = piece >> 3, q = (piece & 7) * 4;
            ra[p] = *reinterpret_cast<const float4 *>(A + (size_t)(row0 + r) * K + k0 + q);
            rb[p] = *reinterpret_cast<const float4 *>(B + (size_t)(col0 + r) * K + k0 + q);
        }
    };
    auto store = [&](int buf) {
#pragma unroll
        for (int p = 0; p < 4; ++p) {
            const int piece = tid + p * 256, r = piece >> 3, q = (piece & 7) * 4;
            *reinterpret_cast<float4 *>(&As[buf][r][q]) = ra[p];
            *reinterpret_cast<float4 *>(&Bs[buf][r][q]) = rb[p];
        }
    };
    f32x16 acc[2][2];
#pragma unroll
    for (int i = 0; i < 2; ++i)
#pragma unroll
        for (int j = 0; j < 2; ++j)
#pragma unroll
            for (int e = 0; e < 16; ++e) acc[i][j][e] = 0.0f;
    auto multiply = [&](int buf) {
        // 32x32x2: lane l holds row (l & 31), k = (l >> 5); a 16-byte read gives the lane its k, k + 2, k + 4, k + 6
#pragma unroll
        for (int ks = 0; ks < BK; ks += 8) {
            float4 a[2], b[2];
#pragma unroll
            for (int i = 0; i < 2; ++i) a[i] = *reinterpret_cast<const float4 *>(&As[buf][wr * 64 + i * 32 + (lane & 31)][ks + 4 * (lane >> 5)]);
#pragma unroll
            for (int j = 0; j < 2; ++j) b[j] = *reinterpret_cast<const float4 *>(&Bs[buf][wc * 64 + j * 32 + (lane & 31)][ks + 4 * (lane >> 5)]);
            // (lane half h holds k = ks + 4h .. + 3: four MFMAs over the pairs (k, k + 4) of the two halves)
#pragma unroll
            for (int i = 0; i < 2; ++i)
#pragma unroll
                for (int j = 0; j < 2; ++j) {
                    acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x2f32(a[i].x, b[j].x, acc[i][j], 0, 0, 0);
                    acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x2f32(a[i].y, b[j].y, acc[i][j], 0, 0, 0);
                    acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x2f32(a[i].z, b[j].z, acc[i][j], 0, 0, 0);
                    acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x2f32(a[i].w, b[j].w, acc[i][j], 0, 0, 0);
                }
        }
    };
    load(0);
    store(0);
    __syncthreads();
    int buf = 0;
    for (int k0 = BK; k0 < K; k0 += BK) {
        load(k0);
        multiply(buf);
        store(buf ^ 1);
        __syncthreads();
        buf ^= 1;
    }
    multiply(buf);
#pragma unroll
    for (int i = 0; i < 2; ++i)
#pragma unroll
        for (int j = 0; j < 2; ++j)
#pragma unroll
            for (int e = 0; e < 16; ++e) {
                const int r = row0 + wr * 64 + i * 32 + (e & 3) + 8 * (e >> 2) + 4 * (lane >> 5);
                const int c = col0 + wc * 64 + j * 32 + (lane & 31);
                C[(size_t)r * N + c] = acc[i][j][e];
            }
}

int main(int argc, char **argv)
{
    int M = 1024, N = 16384, K = 2304;      // a ConvGRU gate GEMM of the 32 x 32 level, B = 8 ... x 2
    if (argc == 4) { M = atoi(argv[1]); N = atoi(argv[2]); K = atoi(argv[3]); }
    std::vector<float> hA((size_t)M * K), hB((size_t)N * K), hC((size_t)M * N);
    srand(1);
    for (auto &v : hA) v = (float)rand() / RAND_MAX - 0.5f;
    for (auto &v : hB) v = (float)rand() / RAND_MAX - 0.5f;
    float *dA, *dB, *dC;
    hipMalloc(&dA, hA.size() * 4); hipMalloc(&dB, hB.size() * 4); hipMalloc(&dC, hC.size() * 4);
    hipMemcpy(dA, hA.data(), hA.size() * 4, hipMemcpyHostToDevice);
    hipMemcpy(dB, hB.data(), hB.size() * 4, hipMemcpyHostToDevice);
    dim3 grid(N / TN, M / TM);
    auto run = [&](int which, const char *name) {
        hipEvent_t a, b; hipEventCreate(&a); hipEventCreate(&b);
        for (int rep = 0; rep < 3; ++rep) {
            if (which == 2) hipLaunchKernelGGL(gemm_bf16x6, grid, dim3(256), 0, 0, dA, dB, dC, M, N, K);
            else if (which) hipLaunchKernelGGL(gemm_bf16x3, grid, dim3(256), 0, 0, dA, dB, dC, M, N, K);
            else hipLaunchKernelGGL(gemm_f32, grid, dim3(256), 0, 0, dA, dB, dC, M, N, K);
        }
        hipEventRecord(a);
        const int reps = 20;
        for (int rep = 0; rep < reps; ++rep) {
            if (which == 2) hipLaunchKernelGGL(gemm_bf16x6, grid, dim3(256), 0, 0, dA, dB, dC, M, N, K);
            else if (which) hipLaunchKernelGGL(gemm_bf16x3, grid, dim3(256), 0, 0, dA, dB, dC, M, N, K);
            else hipLaunchKernelGGL(gemm_f32, grid, dim3(256), 0, 0, dA, dB, dC, M, N, K);
        }
        hipEventRecord(b); hipEventSynchronize(b);
        float ms; hipEventElapsedTime(&ms, a, b); ms /= reps;
        hipMemcpy(hC.data(), dC, hC.size() * 4, hipMemcpyDeviceToHost);
        double worst = 0.0, scale = 0.0;
        for (int s = 0; s < 2000; ++s) {
            const int r = rand() % M, c = rand() % N;
            double ref = 0.0;
            for (int k = 0; k < K; ++k) ref += (double)hA[(size_t)r * K + k] * (double)hB[(size_t)c * K + k];
            worst = fmax(worst, fabs((double)hC[(size_t)r * N + c] - ref));
            scale = fmax(scale, fabs(ref));
        }
        printf("%-28s %8.3f ms  %7.1f TFLOP/s (fp32-equivalent)   max |err| / max |C| over 2000 entries: %.2e\n", name, ms,
               2.0 * M * N * K / ms / 1e9, worst / scale);
    };
    printf("M = %d, N = %d, K = %d (%.1f GFLOP)\n", M, N, K, 2.0 * M * N * K / 1e9);
    run(0, "fp32 MFMA 32x32x2");
    run(1, "bf16 x 3 on MFMA 32x32x16");
    run(2, "bf16 3-way, 6 products");
    return 0;
}
