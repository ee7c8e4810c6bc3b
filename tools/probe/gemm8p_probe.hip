// gemm8p_probe.hip -- the guide's "256^2 8-phase" K loop (cdna_hip_programming.md, "The 256^2 8-phase template") written
// from its description, as a standalone yardstick:  C[M,N] (bf16) = A[M,K] . B[N,K]^T, both operands bf16 row-major.
//
//   hipcc -O3 -std=c++17 --offload-arch=gfx950 gemm8p_probe.hip -o gemm8p_probe
//   ./gemm8p_probe [M N K] [iters]
//
// 256 x 256 tile, 8 waves as 2 (M) x 4 (N), 128 x 64 per wave, v_mfma_f32_16x16x32_bf16; K in 64-deep tiles, two LDS
// buffers of four 16 KB half-tiles each (A0 A1 B0 B1 = 128 KB); one half-tile staged per phase by LDS-DMA (2 per thread),
// three half-tiles in flight across the raw barriers, counted vmcnt(6) once per K-tile; waves 4-7 run one barrier behind
// waves 0-3, so on every SIMD one wave multiplies while its partner reads fragments and issues DMAs.
// Checks against a plain fp32 kernel on uniform random [-1, 1) operands, then times back-to-back launches.
#include <hip/hip_runtime.h>
#include <stdint.h>
#include <stdio.h>
#include <stdlib.h>
#include <vector>
#include <algorithm>

typedef __bf16 bf16;
typedef __attribute__((ext_vector_type(8))) __bf16 bf16x8;
typedef __attribute__((ext_vector_type(4))) float f32x4;
typedef __attribute__((ext_vector_type(2))) uint32_t u32x2;
typedef __attribute__((address_space(3))) void lds_void;

#define HIPCHK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { printf("HIP error %s at %s:%d\n", hipGetErrorString(e_), __FILE__, __LINE__); exit(1); } } while (0)

#include "../../quadruplet-sentence-transformer_amd/csrc/gemm8p.h"

__global__ __launch_bounds__(512, 1) void gemm8p_kernel(const bf16* A, const bf16* B, bf16* C, int M, int N, int K, int lda, int ldb, int ldc) {
    extern __shared__ __attribute__((aligned(16))) char smem[];
    const int ntn = (N + 255) / 256, ntm = (M + 255) / 256;
    const int wg = g8p::xcd_remap(blockIdx.x, ntm * ntn);
    const int m0 = (wg / ntn) * 256, n0 = (wg % ntn) * 256;
    f32x4 acc[8][4];
    g8p::kloop_nt(A, lda, M - m0 < 256 ? M - m0 : 256, B, ldb, N - n0 < 256 ? N - n0 : 256, K, m0, n0, smem, acc);
    // plain epilogue: D rows = n (4 consecutive per lane), D column = m = lane & 15
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6, wr = wave >> 2, wc = wave & 3;
#pragma unroll
    for (int i = 0; i < 8; ++i)
#pragma unroll
        for (int j = 0; j < 4; ++j) {
            const int m = m0 + wr * 128 + i * 16 + (lane & 15);
            const int n = n0 + wc * 64 + j * 16 + 4 * (lane >> 4);
            if (m < M && n < N) {
                u32x2 pk;
                pk[0] = g8p::pack2(acc[i][j][0], acc[i][j][1]);
                pk[1] = g8p::pack2(acc[i][j][2], acc[i][j][3]);
                *(u32x2*)(C + (size_t)m * ldc + n) = pk;
            }
        }
}

__global__ void ref_kernel(const bf16* A, const bf16* B, float* C, int M, int N, int K) {
    const int n = blockIdx.x * 16 + (threadIdx.x & 15), m = blockIdx.y * 16 + (threadIdx.x >> 4);
    if (m >= M || n >= N) return;
    float s = 0.f;
    for (int k = 0; k < K; ++k) s += (float)A[(size_t)m * K + k] * (float)B[(size_t)n * K + k];
    C[(size_t)m * N + n] = s;
}
__global__ void fill_kernel(bf16* p, size_t n, uint32_t seed, int mode) {
    size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n) return;
    uint32_t x = (uint32_t)i * 2654435761u + seed;
    x ^= x >> 16; x *= 0x7feb352du; x ^= x >> 15; x *= 0x846ca68bu; x ^= x >> 16;
    const float u = (float)(x >> 8) * (1.0f / 16777216.0f);
    p[i] = (bf16)(mode == 0 ? 0.f : 2.f * u - 1.f);
}
__global__ void cmp_kernel(const bf16* C, const float* R, size_t n, float* out) {
    size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n) return;
    const float d = fabsf((float)C[i] - R[i]);
    const float tol = 0.02f + 0.01f * fabsf(R[i]);
    if (d > tol) atomicAdd(&out[0], 1.0f);
    atomicMax((int*)&out[1], __float_as_int(d));
}

int main(int argc, char** argv) {
    int M = 4096, N = 4096, K = 4096, iters = 30;
    if (argc >= 4) { M = atoi(argv[1]); N = atoi(argv[2]); K = atoi(argv[3]); }
    if (argc >= 5) iters = atoi(argv[4]);
    if (K % 128 != 0) { printf("K must be a multiple of 128\n"); return 1; }
    bf16 *A, *B, *C; float *R, *stat;
    HIPCHK(hipMalloc(&A, (size_t)M * K * 2)); HIPCHK(hipMalloc(&B, (size_t)N * K * 2)); HIPCHK(hipMalloc(&C, (size_t)M * N * 2));
    HIPCHK(hipMalloc(&R, (size_t)M * N * 4)); HIPCHK(hipMalloc(&stat, 8));
    HIPCHK(hipFuncSetAttribute((const void*)gemm8p_kernel, hipFuncAttributeMaxDynamicSharedMemorySize, g8p::LDS_BYTES));
    const int grid = ((M + 255) / 256) * ((N + 255) / 256);
    for (int mode = 1; mode >= 0; --mode) {
        fill_kernel<<<(unsigned)(((size_t)M * K + 255) / 256), 256>>>(A, (size_t)M * K, 1u, mode);
        fill_kernel<<<(unsigned)(((size_t)N * K + 255) / 256), 256>>>(B, (size_t)N * K, 7u, mode);
        HIPCHK(hipMemset(C, 0xFF, (size_t)M * N * 2));
        gemm8p_kernel<<<grid, 512, g8p::LDS_BYTES>>>(A, B, C, M, N, K, K, K, N);
        HIPCHK(hipDeviceSynchronize());
        if (mode == 1) {
            ref_kernel<<<dim3((N + 15) / 16, (M + 15) / 16), 256>>>(A, B, R, M, N, K);
            HIPCHK(hipMemset(stat, 0, 8));
            cmp_kernel<<<(unsigned)(((size_t)M * N + 255) / 256), 256>>>(C, R, (size_t)M * N, stat);
            float h[2]; HIPCHK(hipMemcpy(h, stat, 8, hipMemcpyDeviceToHost));
            printf("check M=%d N=%d K=%d: %g elements out of tolerance, max |d| = %g\n", M, N, K, h[0], h[1]);
            // repeat-run race screen: the same launch 20 times must reproduce C bit for bit
            std::vector<uint16_t> c0((size_t)M * N), c1((size_t)M * N);
            HIPCHK(hipMemcpy(c0.data(), C, (size_t)M * N * 2, hipMemcpyDeviceToHost));
            int bad_runs = 0;
            for (int r = 0; r < 20; ++r) {
                gemm8p_kernel<<<grid, 512, g8p::LDS_BYTES>>>(A, B, C, M, N, K, K, K, N);
                HIPCHK(hipMemcpy(c1.data(), C, (size_t)M * N * 2, hipMemcpyDeviceToHost));
                if (c0 != c1) ++bad_runs;
            }
            printf("race screen: %d of 20 repeat launches differ\n", bad_runs);
        }
        hipEvent_t e0, e1; HIPCHK(hipEventCreate(&e0)); HIPCHK(hipEventCreate(&e1));
        for (int i = 0; i < 5; ++i) gemm8p_kernel<<<grid, 512, g8p::LDS_BYTES>>>(A, B, C, M, N, K, K, K, N);
        std::vector<float> ts;
        for (int rep = 0; rep < 5; ++rep) {
            HIPCHK(hipEventRecord(e0));
            for (int i = 0; i < iters; ++i) gemm8p_kernel<<<grid, 512, g8p::LDS_BYTES>>>(A, B, C, M, N, K, K, K, N);
            HIPCHK(hipEventRecord(e1)); HIPCHK(hipEventSynchronize(e1));
            float ms; HIPCHK(hipEventElapsedTime(&ms, e0, e1)); ts.push_back(ms / iters);
        }
        std::sort(ts.begin(), ts.end());
        const double fl = 2.0 * M * N * K;
        printf("%s operands: median %.1f us = %.0f TF/s (best %.1f us = %.0f TF/s)\n", mode ? "random" : "zero  ",
               ts[2] * 1e3, fl / (ts[2] * 1e-3) * 1e-12, ts[0] * 1e3, fl / (ts[0] * 1e-3) * 1e-12);
    }
    return 0;
}
