// isa_probe.hip -- three facts the 8-phase GEMM kernels rest on, checked on the device:
//  1. v_permlane16_swap_b32 (builtin, two different operands): which 16-lane rows trade places
//  2. v_dot2c_f32_bf16 with a (1, 1) multiplier sums a bf16 pair into an fp32 accumulator
//  3. raw buffer loads: is the SGPR offset part of the range check against num_records?
#include <hip/hip_runtime.h>
#include <stdio.h>
#include <stdint.h>
typedef __attribute__((ext_vector_type(2))) __bf16 v2bf;
typedef __attribute__((ext_vector_type(4))) uint32_t u32x4;
__global__ void k(unsigned* out, float* fo, const uint32_t* buf) {
    unsigned lane = threadIdx.x;
    unsigned a = 1000 + lane, b = 2000 + lane;
    auto r = __builtin_amdgcn_permlane16_swap(a, b, false, false);
    out[lane] = r[0]; out[64 + lane] = r[1];
    v2bf x; x[0] = (__bf16)1.5f; x[1] = (__bf16)2.25f;
    v2bf one; one[0] = (__bf16)1.f; one[1] = (__bf16)1.f;
    fo[lane] = __builtin_amdgcn_fdot2_f32_bf16(x, one, 10.f, false);
    // buffer of 256 bytes; element i holds 100 + i. voffset = 16 * lane (lanes 0..15 in range), soffset = 0 / 192 / 4096
    __amdgpu_buffer_rsrc_t rs = __builtin_amdgcn_make_buffer_rsrc(const_cast<uint32_t*>(buf), 0, 256, 0x00020000);
    u32x4 v0 = __builtin_amdgcn_raw_buffer_load_b128(rs, (int)(16 * (lane & 15)), 0, 0);
    u32x4 v1 = __builtin_amdgcn_raw_buffer_load_b128(rs, (int)(16 * (lane & 15)), 192, 0);
    u32x4 v2 = __builtin_amdgcn_raw_buffer_load_b128(rs, (int)(16 * (lane & 15)), 4096, 0);
    out[128 + lane] = v0[0]; out[192 + lane] = v1[0]; out[256 + lane] = v2[0];
}
int main() {
    unsigned* d; float* f; uint32_t* b;
    (void)hipMalloc(&d, 320 * 4); (void)hipMalloc(&f, 64 * 4); (void)hipMalloc(&b, 65536);
    uint32_t hb[16384]; for (int i = 0; i < 16384; ++i) hb[i] = 100 + i;
    (void)hipMemcpy(b, hb, sizeof(hb), hipMemcpyHostToDevice);
    k<<<1, 64>>>(d, f, b);
    unsigned h[320]; float hf[64];
    (void)hipMemcpy(h, d, sizeof(h), hipMemcpyDeviceToHost); (void)hipMemcpy(hf, f, sizeof(hf), hipMemcpyDeviceToHost);
    printf("permlane16_swap(a = 1000 + lane, b = 2000 + lane), every 8th lane:\n  r[0]:"); for (int l = 0; l < 64; l += 8) printf(" %u", h[l]);
    printf("\n  r[1]:"); for (int l = 0; l < 64; l += 8) printf(" %u", h[64 + l]);
    printf("\ndot2(1.5, 2.25; 1, 1) + 10 = %f\n", hf[0]);
    printf("raw buffer load, num_records 256 B, voffset 16*lane (lanes 0..15):\n  soffset 0   :"); for (int l = 0; l < 16; l += 3) printf(" %u", h[128 + l]);
    printf("\n  soffset 192 :"); for (int l = 0; l < 16; l += 3) printf(" %u", h[192 + l]);
    printf("\n  soffset 4096:"); for (int l = 0; l < 16; l += 3) printf(" %u", h[256 + l]);
    printf("\n");
    return 0;
}
