// mfma_f8_16x16_probe.hip -- operand and scale layout of v_mfma_scale_f32_16x16x128_f8f6f4 (fp8 e4m3 x e4m3, E8M0 block scales),
// found by trying the candidate maps against exact small-integer data (the 32x32x64 form needed the same: mfma_f8_probe.hip).
//   hipcc --offload-arch=gfx950 -O2 mfma_f8_16x16_probe.hip -o mfma_f8_16x16_probe
// Candidates: K map  C1: lane (r = l & 15, g = l >> 4) byte j <-> k = 32 g + j
//                    C2: byte j < 16 <-> k = 16 g + j, j >= 16 <-> k = 64 + 16 g + (j - 16)
//             scale  S1: block b = k / 32 of row r from the lane's OWN scale register, i.e. lane (r, g) supplies block g
//                    S2: block b of row r from lane 16 b + r  (same thing as S1 under C1; differs under C2)
//             byte of the scale register = opsel (0 here; the others are poisoned)
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdint>
#include <cmath>
#include <vector>
typedef __attribute__((ext_vector_type(8))) int i32x8;
typedef __attribute__((ext_vector_type(4))) float f32x4;
__global__ void mfma_once(const i32x8* a, const i32x8* b, const int* sa, const int* sb, f32x4* c) {
    const int l = threadIdx.x;
    f32x4 acc = {0.f, 0.f, 0.f, 0.f};
    acc = __builtin_amdgcn_mfma_scale_f32_16x16x128_f8f6f4(a[l], b[l], acc, 0, 0, 0, sa[l], 0, sb[l]);
    c[l] = acc;
}
static uint8_t f8(int v) {
    if (v == 0) return 0;
    int e = 0, m = v;
    while (m >= 16) { m >>= 1; ++e; }
    while (m < 8) { m <<= 1; --e; }
    return (uint8_t)(((e + 3 + 7) << 3) | (m & 7));
}
static int kmap(int cand, int g, int j) { return cand == 1 ? 32 * g + j : (j < 16 ? 16 * g + j : 64 + 16 * g + (j - 16)); }
int main() {
    const int M = 16, N = 16, K = 128;
    int found = 0;
    for (int kc = 1; kc <= 2; ++kc)
        for (int vary = 0; vary < 2; ++vary) {
            std::vector<float> A(M * K), B(K * N);
            srand(11 + kc);
            for (auto& x : A) x = (float)(rand() % 8);
            for (auto& x : B) x = (float)(rand() % 8);
            int ea[16][4], eb[16][4];
            for (int r = 0; r < 16; ++r) for (int b = 0; b < 4; ++b) { ea[r][b] = vary ? rand() % 5 - 2 : 0; eb[r][b] = vary ? rand() % 5 - 2 : 0; }
            std::vector<double> ref(M * N, 0.0);
            for (int m = 0; m < M; ++m) for (int n = 0; n < N; ++n) for (int k = 0; k < K; ++k)
                ref[m * N + n] += (double)A[m * K + k] * B[k * N + n] * std::ldexp(1.0, ea[m][k >> 5] + eb[n][k >> 5]);
            for (int sc = 1; sc <= 2; ++sc) {
                std::vector<uint8_t> ab(64 * 32), bb(64 * 32);
                std::vector<int> sa(64), sb(64);
                for (int l = 0; l < 64; ++l) {
                    const int r = l & 15, g = l >> 4;
                    for (int j = 0; j < 32; ++j) {
                        const int k = kmap(kc, g, j);
                        ab[l * 32 + j] = f8((int)A[r * K + k]);
                        bb[l * 32 + j] = f8((int)B[k * N + r]);
                    }
                    // S1 and S2 both: lane 16 b + r carries block b of row r (under C1 that IS the lane's own block)
                    const int blk = g;
                    sa[l] = (127 + ea[r][blk]) | 0x90909000;
                    sb[l] = (127 + eb[r][blk]) | 0x70707000;
                    (void)sc;
                }
                void *da, *db, *dsa, *dsb, *dc;
                (void)hipMalloc(&da, 64 * 32); (void)hipMalloc(&db, 64 * 32); (void)hipMalloc(&dsa, 256); (void)hipMalloc(&dsb, 256); (void)hipMalloc(&dc, 64 * 16);
                (void)hipMemcpy(da, ab.data(), 64 * 32, hipMemcpyHostToDevice); (void)hipMemcpy(db, bb.data(), 64 * 32, hipMemcpyHostToDevice);
                (void)hipMemcpy(dsa, sa.data(), 256, hipMemcpyHostToDevice); (void)hipMemcpy(dsb, sb.data(), 256, hipMemcpyHostToDevice);
                mfma_once<<<1, 64>>>((const i32x8*)da, (const i32x8*)db, (const int*)dsa, (const int*)dsb, (f32x4*)dc);
                std::vector<float> c(64 * 4);
                (void)hipMemcpy(c.data(), dc, 64 * 16, hipMemcpyDeviceToHost);
                // C/D: col = l & 15, row = 4 (l >> 4) + reg  (A operand rows, B operand columns)
                int bad = 0;
                for (int l = 0; l < 64; ++l) for (int q = 0; q < 4; ++q) {
                    const int row = 4 * (l >> 4) + q, col = l & 15;
                    if (std::fabs(c[l * 4 + q] - ref[row * N + col]) > 1e-3 * (1 + std::fabs(ref[row * N + col]))) ++bad;
                }
                printf("K map C%d, scales %s (lane 16 b + r carries block b of row r): %s (%d of 256 wrong)\n", kc, vary ? "varied" : "unit",
                       bad ? "MISMATCH" : "ok", bad);
                if (!bad && vary) found = kc;
                break;
            }
        }
    printf("=> %s\n", found == 1 ? "lane (r, g) holds k = 32 g .. 32 g + 31 (one MX block) and supplies that block's scale" :
                      found == 2 ? "split map (as 32x32x64)" : "no candidate matched with varied scales");
    return 0;
}
