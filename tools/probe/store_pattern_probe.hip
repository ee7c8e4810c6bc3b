// store_pattern_probe.hip -- HBM write rate of the epilogue store shapes of a tiled GEMM (gfx950).
// A [M x N] bf16 matrix (N = 1536: rows of 3,072 B) is written once by 256 persistent workgroups of 512 threads, tile by
// tile (128 x 192 tiles, a wave owns 64 x 96 or 32 x 192 of it), with 16-byte stores shaped as:
//   0: 16 rows x  64 B per wave-instruction, the two halves of a 128-B line ~1 us apart        (32 x 32 blocks, one block per step)
//   1: 16 rows x  64 B, the halves of a line in back-to-back instructions
//   2:  8 rows x 128 B  (whole aligned lines)
//   3: 32 rows x  32 B, the four quarters of a line in back-to-back instructions                (register-only epilogue)
//   4: 5.33 rows x 192 B (1.5 lines, the round-2 kernel's shape)
// build: hipcc -O3 --offload-arch=gfx950 store_pattern_probe.hip -o store_pattern_probe
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdint>
#include <vector>
typedef __attribute__((ext_vector_type(4))) uint32_t u32x4;

template <int MODE>
__global__ __launch_bounds__(512) void probe(uint16_t* C, int M, int N) {
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const int ntn = N / 192, T = (M / 256) * ntn;                 // 256 x 192 tiles, 8 waves as 4 x 2 of 64 x 96
    const __amdgpu_buffer_rsrc_t rs = __builtin_amdgcn_make_buffer_rsrc(C, 0, (int)((size_t)M * N * 2), 0x00020000);
    const u32x4 v = {(uint32_t)lane, 2u, 3u, 4u};
    for (int t = blockIdx.x; t < T; t += gridDim.x) {
        const int m0 = (t / ntn) * 256 + (wave >> 1) * 64, n0 = (t % ntn) * 192 + (wave & 1) * 96;
        if (MODE == 0 || MODE == 1) {
            for (int i = 0; i < 2; ++i)
                for (int j = 0; j < 3; ++j) {
                    for (int tt = 0; tt < 2; ++tt) {
                        const int row = 16 * tt + (lane >> 2), c8 = lane & 3;
                        const uint32_t off = ((uint32_t)(m0 + i * 32 + row) * N + n0 + j * 32 + c8 * 8) * 2u;
                        __builtin_amdgcn_raw_buffer_store_b128(v, rs, (int)off, 0, 0);
                    }
                    if (MODE == 0) __builtin_amdgcn_s_sleep(100);  // ~6400 cycles between the halves of a line
                }
        } else if (MODE == 2) {
            // the wave's 64 rows x 96 cols do not hold whole lines; write 64-column groups of the TILE instead: wave w takes
            // rows [32 w, 32 w + 32) x 192 cols = 3 lines per row -> 12 instructions of 8 rows x 128 B
            const int mt = (t / ntn) * 256 + wave * 32, nt = (t % ntn) * 192;
            for (int q = 0; q < 12; ++q) {
                const int idx = q * 64 + lane, row = idx / 24, c = idx % 24;
                const uint32_t off = ((uint32_t)(mt + row) * N + nt + c * 8) * 2u;
                __builtin_amdgcn_raw_buffer_store_b128(v, rs, (int)off, 0, 0);
            }
        } else if (MODE == 3) {
            for (int i = 0; i < 2; ++i)
                for (int j = 0; j < 3; ++j)
                    for (int p = 0; p < 2; ++p) {
                        const int row = lane & 31, fh = lane >> 5;
                        const uint32_t off = ((uint32_t)(m0 + i * 32 + row) * N + n0 + j * 32 + p * 16 + fh * 8) * 2u;
                        __builtin_amdgcn_raw_buffer_store_b128(v, rs, (int)off, 0, 0);
                    }
        } else {
            for (int i = 0; i < 2; ++i)
                for (int q = 0; q < 6; ++q) {
                    const int idx = q * 64 + lane, row = idx / 12, c = idx % 12;
                    const uint32_t off = ((uint32_t)(m0 + i * 32 + row) * N + n0 + c * 8) * 2u;
                    __builtin_amdgcn_raw_buffer_store_b128(v, rs, (int)off, 0, 0);
                }
        }
    }
}

template <int MODE> float run(uint16_t* C, int M, int N) {
    hipEvent_t e0, e1;
    hipEventCreate(&e0); hipEventCreate(&e1);
    for (int w = 0; w < 2; ++w) probe<MODE><<<256, 512>>>(C, M, N);
    hipEventRecord(e0);
    for (int r = 0; r < 10; ++r) probe<MODE><<<256, 512>>>(C, M, N);
    hipEventRecord(e1);
    hipEventSynchronize(e1);
    float ms = 0;
    hipEventElapsedTime(&ms, e0, e1);
    return ms / 10;
}

int main() {
    const int M = 32768, N = 1536;
    uint16_t* C;
    hipMalloc(&C, (size_t)M * N * 2);
    const double gb = (double)M * N * 2 / 1e9;
    const char* names[5] = {"16 rows x 64 B, halves of a line far apart", "16 rows x 64 B, halves back to back", "8 rows x 128 B (whole lines)",
                            "32 rows x 32 B, quarters back to back", "5.33 rows x 192 B (1.5 lines)"};
    float ms[5] = {run<0>(C, M, N), run<1>(C, M, N), run<2>(C, M, N), run<3>(C, M, N), run<4>(C, M, N)};
    for (int i = 0; i < 5; ++i) printf("mode %d  %-45s %8.1f us  %6.2f TB/s\n", i, names[i], ms[i] * 1e3, gb / ms[i]);
    return 0;
}
