// Which lane's E8M0 scale byte applies to which (row, 32-element K block) of v_mfma_scale_f32_32x32x64_f8f6f4?
// A = ones; B = ones inside K block hb only; scale of ONE lane L doubled -> rows whose result doubles belong to lane L.
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdint>
#include <vector>
typedef __attribute__((ext_vector_type(8))) int i32x8;
typedef __attribute__((ext_vector_type(16))) float f32x16;
__global__ void mfma_once(const i32x8* a, const i32x8* b, const int* sa, const int* sb, f32x16* c) {
    const int l = threadIdx.x;
    f32x16 acc = {};
    acc = __builtin_amdgcn_mfma_scale_f32_32x32x64_f8f6f4(a[l], b[l], acc, 0, 0, 0, sa[l], 0, sb[l]);
    c[l] = acc;
}
int main() {
    const uint8_t one = 0x38;       // 1.0 in e4m3
    void *da, *db, *dsa, *dsb, *dc;
    hipMalloc(&da, 2048); hipMalloc(&db, 2048); hipMalloc(&dsa, 256); hipMalloc(&dsb, 256); hipMalloc(&dc, 4096);
    std::vector<uint8_t> ab(2048, one), bb(2048);
    std::vector<int> sa(64), sb(64, 127);
    std::vector<float> c(1024);
    for (int which = 0; which < 2; ++which) {           // 0: vary A's scale, 1: vary B's scale (roles swapped)
        printf("%s scale: lane -> (row%s, k-block) it scales\n", which ? "B" : "A", which ? " = column" : "");
        for (int L = 0; L < 64; ++L) {
            printf("  lane %2d:", L);
            for (int hb = 0; hb < 2; ++hb) {
                for (int l = 0; l < 64; ++l)
                    for (int j = 0; j < 32; ++j) bb[l * 32 + j] = ((l >> 5) == hb) ? one : 0;
                for (int l = 0; l < 64; ++l) sa[l] = (l == L) ? 128 : 127;
                // operand with the one-block pattern is the one NOT being scale-tested
                hipMemcpy(which ? db : da, ab.data(), 2048, hipMemcpyHostToDevice);
                hipMemcpy(which ? da : db, bb.data(), 2048, hipMemcpyHostToDevice);
                hipMemcpy(which ? dsb : dsa, sa.data(), 256, hipMemcpyHostToDevice);
                hipMemcpy(which ? dsa : dsb, sb.data(), 256, hipMemcpyHostToDevice);
                mfma_once<<<1, 64>>>((const i32x8*)da, (const i32x8*)db, (const int*)dsa, (const int*)dsb, (f32x16*)dc);
                hipMemcpy(c.data(), dc, 4096, hipMemcpyDeviceToHost);
                for (int idx = 0; idx < 32; ++idx) {     // idx = row (A test) or column (B test)
                    float v;
                    if (!which) { const int row = idx, l = (((row >> 2) & 1) << 5), reg = (row & 3) + 4 * (row >> 3); v = c[l * 16 + reg]; }
                    else v = c[idx * 16];                // column idx, row 0
                    if (v != 32.f) printf(" (%d, blk %d: x%.2f)", idx, hb, v / 32.f);
                }
            }
            printf("\n");
        }
    }
    return 0;
}
