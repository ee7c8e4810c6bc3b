// mfma_f8_probe.hip -- operand layout of v_mfma_scale_f32_32x32x64_f8f6f4 (fp8 e4m3 x fp8 e4m3, E8M0 block scales),
// established with exact small-integer data before any kernel relies on it (cdna guide section 3: "check the map with
// exact integer data"). Build: hipcc --offload-arch=gfx950 -O2 tools/probe/mfma_f8_probe.hip -o tools/probe/mfma_f8_probe
//
// Layout under test (found with mfma_f8_scalemap.hip after the "lane holds 32 consecutive k" guess failed on the scales):
// the 64-deep step is two 32-deep halves; lane l = 32h + r holds, in operand registers 0-3, A[row r][k = 16h .. 16h+15]
// (first half) and in registers 4-7 A[row r][k = 32 + 16h .. 32 + 16h + 15] (second half), bytes little-endian; B likewise
// with B[k][col r]. The E8M0 scale of MX block b (k = 32b .. 32b+31) of row / column r is byte `opsel` of the scale
// register of lane 32b + r. C/D as the bf16 32x32 map: col = l & 31, row = (reg&3) + 8(reg>>2) + 4h.
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdint>
#include <cstring>
#include <cmath>
#include <vector>

typedef __attribute__((ext_vector_type(8))) int i32x8;
typedef __attribute__((ext_vector_type(16))) float f32x16;

__global__ void mfma_once(const i32x8* a, const i32x8* b, const int* sa, const int* sb, f32x16* c) {
    const int l = threadIdx.x;
    f32x16 acc = {};
    acc = __builtin_amdgcn_mfma_scale_f32_32x32x64_f8f6f4(a[l], b[l], acc, 0, 0, 0, sa[l], 0, sb[l]);
    c[l] = acc;
}

static uint8_t f8(int v) {            // small non-negative integers and simple fractions as OCP e4m3 (bias 7, 3 mantissa bits)
    if (v == 0) return 0;
    int e = 0, m = v;
    while (m >= 16) { m >>= 1; ++e; }
    while (m < 8) { m <<= 1; --e; }   // m in [8, 16): 1.mmm
    return (uint8_t)(((e + 3 + 7) << 3) | (m & 7));
}

int run(int vary_a, int vary_b);
int main() {
    int bad = 0;
    bad += run(0, 0);       // unit scales: K-slot pairing and C layout alone
    bad += run(1, 0);       // A block scales
    bad += run(0, 1);       // B block scales
    bad += run(1, 1);
    return bad != 0;
}
int run(int vary_a, int vary_b) {
    const int M = 32, N = 32, K = 64;
    std::vector<float> A(M * K), B(K * N), SA(M * 2), SB(N * 2);
    std::vector<uint8_t> ab(64 * 32), bb(64 * 32);
    std::vector<int> sa(64), sb(64);
    srand(7);
    for (auto& x : A) x = (float)(rand() % 8);
    for (auto& x : B) x = (float)(rand() % 8);
    int ea[64], eb[64];
    for (int l = 0; l < 64; ++l) { ea[l] = vary_a ? rand() % 5 - 2 : 0; eb[l] = vary_b ? rand() % 5 - 2 : 0; }     // block scales 2^-2 .. 2^2
    for (int l = 0; l < 64; ++l) {
        const int r = l & 31, h = l >> 5;
        for (int j = 0; j < 32; ++j) {
            const int k = (j < 16) ? 16 * h + j : 32 + 16 * h + (j - 16);
            ab[l * 32 + j] = f8((int)A[r * K + k]);
            bb[l * 32 + j] = f8((int)B[k * N + r]);
        }
        sa[l] = 127 + ea[l];            // E8M0 in byte 0 (opsel 0); other bytes poisoned to catch a wrong byte select
        sb[l] = 127 + eb[l];
        sa[l] |= 0x90909000; sb[l] |= 0x70707000;
    }
    std::vector<double> ref(M * N, 0.0);
    for (int m = 0; m < M; ++m)
        for (int n = 0; n < N; ++n)
            for (int k = 0; k < K; ++k) {
                const int h = k >> 5;
                ref[m * N + n] += (double)A[m * K + k] * B[k * N + n] * std::ldexp(1.0, ea[32 * h + m] + eb[32 * h + n]);
            }
    void *da, *db, *dsa, *dsb, *dc;
    hipMalloc(&da, 64 * 32); hipMalloc(&db, 64 * 32); hipMalloc(&dsa, 256); hipMalloc(&dsb, 256); hipMalloc(&dc, 64 * 64);
    hipMemcpy(da, ab.data(), 64 * 32, hipMemcpyHostToDevice); hipMemcpy(db, bb.data(), 64 * 32, hipMemcpyHostToDevice);
    hipMemcpy(dsa, sa.data(), 256, hipMemcpyHostToDevice); hipMemcpy(dsb, sb.data(), 256, hipMemcpyHostToDevice);
    mfma_once<<<1, 64>>>((const i32x8*)da, (const i32x8*)db, (const int*)dsa, (const int*)dsb, (f32x16*)dc);
    std::vector<float> c(64 * 16);
    hipMemcpy(c.data(), dc, 64 * 64, hipMemcpyDeviceToHost);
    int bad = 0;
    double worst = 0;
    for (int l = 0; l < 64; ++l)
        for (int reg = 0; reg < 16; ++reg) {
            const int col = l & 31, row = (reg & 3) + 8 * (reg >> 2) + 4 * (l >> 5);
            const double d = std::fabs((double)c[l * 16 + reg] - ref[row * N + col]);
            worst = d > worst ? d : worst;
            if (d > 1e-3) { if (bad < 5) printf("mismatch lane %d reg %d: got %g want %g\n", l, reg, c[l * 16 + reg], ref[row * N + col]); ++bad; }
        }
    printf("[scales A %s, B %s] mfma_scale_f32_32x32x64 fp8 x fp8: %s (%d of 1024 wrong, max |d| %g)\n", vary_a ? "varied" : "unit",
           vary_b ? "varied" : "unit", bad ? "HYPOTHESIS WRONG" : "layout confirmed", bad, worst);
    return bad != 0;
}
