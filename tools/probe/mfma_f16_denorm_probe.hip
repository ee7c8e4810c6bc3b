// Does v_mfma_f32_32x32x16_f16 / 16x16x32_f16 keep SUBNORMAL f16 inputs (the low halves of split-f16 weights are mostly
// subnormal: |W| * 2^-12 < 6.1e-5) or flush them to zero? And does FP16_OVFL (hwreg MODE bit 23) clamp v_cvt_pk_f16_f32?
// Build: hipcc -O2 --offload-arch=gfx950 mfma_f16_denorm_probe.hip -o mfma_f16_denorm_probe
#include <hip/hip_runtime.h>
#include <stdio.h>
#include <stdint.h>
typedef __attribute__((ext_vector_type(8))) _Float16 h8;
typedef __attribute__((ext_vector_type(2))) _Float16 h2;
typedef __attribute__((ext_vector_type(16))) float f16v;
typedef __attribute__((ext_vector_type(4))) float f4v;
__global__ void k(float* out, float big) {
    h8 a, b;
    for (int e = 0; e < 8; ++e) { a[e] = (_Float16)1.0f; b[e] = (_Float16)9.5367431640625e-07f; }   // 2^-20: subnormal in f16
    f16v c = {};
    c = __builtin_amdgcn_mfma_f32_32x32x16_f16(a, b, c, 0, 0, 0);
    f4v d = {};
    d = __builtin_amdgcn_mfma_f32_16x16x32_f16(a, b, d, 0, 0, 0);
    h2 v; v[0] = (_Float16)big; v[1] = (_Float16)(-big);
    __builtin_amdgcn_s_setreg(1 | (23 << 6), 1);
    h2 w; w[0] = (_Float16)(big * 1.0001f); w[1] = (_Float16)(-big * 1.0001f);
    if (threadIdx.x == 0) { out[0] = c[0]; out[1] = d[0]; out[2] = (float)v[0]; out[3] = (float)v[1]; out[4] = (float)w[0]; out[5] = (float)w[1]; }
}
int main() {
    float* d; hipMalloc(&d, 64);
    k<<<1, 64>>>(d, 1.0e6f);
    float h[6]; hipMemcpy(h, d, 24, hipMemcpyDeviceToHost);
    printf("32x32x16: sum of 16 products 1 * 2^-20 = %g (kept: %g)\n16x16x32: %g (kept: %g)\n", h[0], 16 * 9.5367431640625e-07, h[1], 32 * 9.5367431640625e-07);
    printf("cvt f16(1e6), f16(-1e6) default mode: %g %g ; with FP16_OVFL: %g %g\n", h[2], h[3], h[4], h[5]);
    return 0;
}
