// mfma_rate_probe.hip -- cycles per MFMA, operands in registers, for the bf16 and the block-scaled fp8 instruction
// (6 independent accumulators, 1 or 2 waves per SIMD, every CU busy). Prints cycles per instruction and TFLOP/s.
#include <hip/hip_runtime.h>
#include <cstdio>
#include <vector>
typedef __attribute__((ext_vector_type(8))) int i32x8;
typedef __attribute__((ext_vector_type(8))) __bf16 bf16x8;
typedef __attribute__((ext_vector_type(16))) float f32x16;

template <int KIND>
__global__ __launch_bounds__(512) void rate(const int* src, float* out, long long* cyc, int iters) {
    const int l = threadIdx.x;
    i32x8 a, b;
    for (int j = 0; j < 8; ++j) { a[j] = src[(l * 8 + j) & 1023]; b[j] = src[(l * 8 + j + 77) & 1023]; }
    const bf16x8 ab = __builtin_bit_cast(bf16x8, *(const __attribute__((ext_vector_type(4))) int*)&a);
    const bf16x8 bb = __builtin_bit_cast(bf16x8, *(const __attribute__((ext_vector_type(4))) int*)&b);
    f32x16 acc[6] = {};
    const int sc = 127 + (l & 1);
    const long long t0 = __builtin_readcyclecounter();
    for (int i = 0; i < iters; ++i) {
#pragma unroll
        for (int j = 0; j < 6; ++j) {
            if (KIND == 0) acc[j] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(ab, bb, acc[j], 0, 0, 0);
            else acc[j] = __builtin_amdgcn_mfma_scale_f32_32x32x64_f8f6f4(a, b, acc[j], 0, 0, 0, sc, 0, sc);
        }
    }
    const long long t1 = __builtin_readcyclecounter();
    float s = 0;
    for (int j = 0; j < 6; ++j) for (int r = 0; r < 16; ++r) s += acc[j][r];
    out[blockIdx.x * blockDim.x + l] = s;
    if (l == 0) cyc[blockIdx.x] = t1 - t0;
}

int main() {
    std::vector<int> h(1024);
    for (int i = 0; i < 1024; ++i) h[i] = 0x3c383c38 ^ (i * 2654435761u & 0x07070707);     // e4m3 bytes near 1.0 / plausible bf16 pairs
    int* src; float* out; long long* cyc;
    hipMalloc(&src, 4096); hipMalloc(&out, 512 * 512 * 4); hipMalloc(&cyc, 512 * 8);
    hipMemcpy(src, h.data(), 4096, hipMemcpyHostToDevice);
    const int iters = 2000;
    for (int kind = 0; kind < 2; ++kind)
        for (int threads : {256, 512}) {
            hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
            for (int rep = 0; rep < 2; ++rep) {
                hipEventRecord(e0);
                if (kind == 0) rate<0><<<256, threads>>>(src, out, cyc, iters); else rate<1><<<256, threads>>>(src, out, cyc, iters);
                hipEventRecord(e1); hipEventSynchronize(e1);
            }
            float ms; hipEventElapsedTime(&ms, e0, e1);
            std::vector<long long> c(256); hipMemcpy(c.data(), cyc, 256 * 8, hipMemcpyDeviceToHost);
            double avg = 0; for (auto x : c) avg += x; avg /= 256;
            const double nm = (double)iters * 6;
            const double flop = 256.0 * (threads / 64) * nm * 32 * 32 * (kind ? 64 : 16) * 2;
            printf("%s, %d waves/SIMD: %.1f counter ticks per MFMA per wave (s_memtime units), %.1f us, %.0f TFLOP/s\n",
                   kind ? "mfma_scale_f32_32x32x64_f8f6f4 (e4m3)" : "mfma_f32_32x32x16_bf16", threads / 256, avg / nm, ms * 1e3,
                   flop / (ms * 1e-3) / 1e12);
        }
    return 0;
}
