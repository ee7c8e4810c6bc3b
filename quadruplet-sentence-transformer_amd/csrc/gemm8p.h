// gemm8p.h -- the 8-wave, 8-phase K loop for long-reduction bf16 GEMMs on gfx950 (device code, header-only).
//
// Structure (cdna_hip_programming.md, "The 256^2 8-phase template"): a 256 x 256 output tile per 512-thread workgroup,
// 8 waves as 2 (M) x 4 (N), 128 x 64 per wave on v_mfma_f32_16x16x32_bf16 (128 accumulator registers). K is walked in
// 64-deep tiles; a K-tile's operands sit in LDS as four 16 KB half-tiles
//     A0 / A1 : the rows   wr * 128 + h * 64 + [0, 64)  of both wave rows     (h = 0 / 1)
//     B0 / B1 : the columns wc * 64 + h * 32 + [0, 32)  of all four wave columns
// so that quadrant (qm, qn) of every wave's 128 x 64 block needs exactly A<qm> and B<qn>. A K-tile is four phases, one
// quadrant each, in the order (0,0) (0,1) (1,1) (1,0); the fragments a phase needs that are not yet in registers are read at
// its start (12 / 4 / 8 / 0 ds_read_b128), and every phase stages ONE half-tile (two LDS-DMAs per thread) of a later
// K-tile. Two LDS buffers (K-tile parity) x 64 KB = 128 KB.
//
// Each phase is   { fragment reads ; stage ; s_barrier ; 16 MFMA ; s_barrier }   and waves 4-7 run one barrier behind
// waves 0-3: on every SIMD one wave multiplies while its partner reads and issues DMAs.
//
// Hazards, by the count (phase numbers p = 4t + 1 .. 4t + 4 for K-tile t; K-tile t lives in buffer t & 1):
//   issue    : A1(t+1) at 4t+1, B0(t+2) at 4t+2, A0(t+2) at 4t+3, B1(t+2) at 4t+4   (three half-tiles in flight)
//   RAW      : `s_waitcnt vmcnt(6)` in phase 4t+4, before its first barrier, retires everything up to A1(t+1), i.e. all of
//              K-tile t+1, whose first reads are in phase 4t+5 -- one phase after the wait (the staggered half reaches the
//              wait half a phase later; its partner's phase-4t+5 reads come after that barrier)
//   WAR      : last reads  B0(t), A0(t): 4t+1;  B1(t): 4t+2;  A1(t): 4t+3.  Restaged at 4t+2 (B0: one phase later -- legal
//              because phase 4t+1 issues its four B reads FIRST and retires them with lgkmcnt(8) before its first barrier),
//              4t+3 (A0), 4t+4 (B1), 4t+5 (A1): two phases later.
// K-tiles past the end are staged from a zero-length buffer (hardware zero fill, no memory traffic), so the DMA count per
// phase -- and with it every vmcnt -- is the same in the tail as in steady state.
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>

namespace g8p {

typedef __bf16 bf16_t;
typedef __attribute__((ext_vector_type(8))) __bf16 bf16x8_t;
typedef __attribute__((ext_vector_type(4))) float f32x4_t;
typedef __attribute__((address_space(3))) void lds_void_t;

constexpr int BUF_BYTES = 65536;                 // one K-tile: A [256][64] + B [256][64] bf16
constexpr int LDS_BYTES = 2 * BUF_BYTES;

__device__ __forceinline__ int xcd_remap(int b, int nwg) {
    const int q = nwg >> 3, r = nwg & 7, x = b & 7;
    return (x < r ? x * (q + 1) : r * (q + 1) + (x - r) * q) + (b >> 3);
}
__device__ __forceinline__ uint32_t pack2(float lo, float hi) {
    typedef __attribute__((ext_vector_type(2))) __bf16 v2;
    v2 v; v[0] = (__bf16)lo; v[1] = (__bf16)hi;
    return __builtin_bit_cast(uint32_t, v);
}
__device__ __forceinline__ __amdgpu_buffer_rsrc_t rsrc(const void* base, uint32_t bytes) {
    return __builtin_amdgcn_make_buffer_rsrc(const_cast<void*>(base), 0, (int)bytes, 0x00020000);
}
__device__ __forceinline__ void dma16(__amdgpu_buffer_rsrc_t r, char* lds_wave_base, uint32_t voff, uint32_t soff) {
    __builtin_amdgcn_raw_ptr_buffer_load_lds(r, (lds_void_t*)lds_wave_base, 16, (int)voff, (int)soff, 0, 0);
}

// C tile = A[m0 .. m0+256, :] . B[n0 .. n0+256, :]^T over K (K % 128 == 0), both operands bf16 with K contiguous.
// rows_a / rows_b: valid rows of the tile (rows beyond read as zero). Leaves the wave's 128 x 64 block in acc:
//   acc[i][j][r] = C[m0 + wr*128 + i*16 + (lane & 15)][n0 + wc*64 + j*16 + 4*(lane >> 4) + r]
// and returns after a workgroup barrier with no DMA outstanding (the caller may reuse the LDS).
__device__ __forceinline__ void kloop_nt(const bf16_t* A, int lda, int rows_a, const bf16_t* B, int ldb, int rows_b, int K,
                                         int m0, int n0, char* smem, f32x4_t (&acc)[8][4]) {
    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int wr = wave >> 2, wc = wave & 3;
    const int nk = K >> 6;

    const __amdgpu_buffer_rsrc_t ra = rsrc(A + (size_t)m0 * lda, (uint32_t)rows_a * (uint32_t)lda * 2u);
    const __amdgpu_buffer_rsrc_t rb = rsrc(B + (size_t)n0 * ldb, (uint32_t)rows_b * (uint32_t)ldb * 2u);
    const __amdgpu_buffer_rsrc_t rz = rsrc(A, 0u);

    // DMA geometry: one wave-instruction = 8 rows x 128 B; a half-tile = 16 instructions = 2 per wave
    const int arow = wr * 128 + (wave & 3) * 16 + (lane >> 3);            // h = 0, t = 0 (t adds 8 rows, h adds 64)
    const int brow = (wave >> 1) * 64 + (wave & 1) * 16 + (lane >> 3);    // h = 0, t = 0 (t adds 8 rows, h adds 32)
    uint32_t va[2], vb[2];
#pragma unroll
    for (int t = 0; t < 2; ++t) {
        const int r1 = arow + 8 * t, r2 = brow + 8 * t;
        va[t] = (uint32_t)r1 * lda * 2u + (uint32_t)(((lane & 7) ^ ((r1 >> 1) & 7)) << 4);
        vb[t] = (uint32_t)r2 * ldb * 2u + (uint32_t)(((lane & 7) ^ ((r2 >> 1) & 7)) << 4);
    }
    char* const dst_a = smem + (wr * 128 + (wave & 3) * 16) * 128;
    char* const dst_b = smem + 32768 + ((wave >> 1) * 64 + (wave & 1) * 16) * 128;
    const uint32_t ha = 64u * lda * 2u, hb = 32u * ldb * 2u;

    auto stage_a = [&](int kt, int h) {
        const __amdgpu_buffer_rsrc_t r = kt < nk ? ra : rz;
        const uint32_t so = (uint32_t)kt * 128u + (h ? ha : 0u);
        char* d = dst_a + (kt & 1) * BUF_BYTES + h * 8192;
        dma16(r, d, va[0], so);
        dma16(r, d + 1024, va[1], so);
    };
    auto stage_b = [&](int kt, int h) {
        const __amdgpu_buffer_rsrc_t r = kt < nk ? rb : rz;
        const uint32_t so = (uint32_t)kt * 128u + (h ? hb : 0u);
        char* d = dst_b + (kt & 1) * BUF_BYTES + h * 4096;
        dma16(r, d, vb[0], so);
        dma16(r, d + 1024, vb[1], so);
    };

    // fragment read offsets: row = 16-aligned base + (lane & 15), chunk = 4 * kstep + (lane >> 4)
    const uint32_t lo0 = (uint32_t)((lane & 15) * 128 + (((lane >> 4) ^ ((lane & 15) >> 1)) << 4));
    const uint32_t ao0 = lo0 + wr * 128 * 128, ao1 = ao0 ^ 64u;
    const uint32_t bo0 = lo0 + 32768 + wc * 64 * 128, bo1 = bo0 ^ 64u;

#pragma unroll
    for (int i = 0; i < 8; ++i)
#pragma unroll
        for (int j = 0; j < 4; ++j) acc[i][j] = f32x4_t{0.f, 0.f, 0.f, 0.f};

    bf16x8_t fa[4][2], fb0[2][2], fb1[2][2];

#define G8P_RD_A(buf_, qm_)                                                                                      \
    _Pragma("unroll") for (int i = 0; i < 4; ++i) {                                                              \
        fa[i][0] = *(const bf16x8_t*)(smem + (buf_) * BUF_BYTES + ((qm_) * 64 + i * 16) * 128 + ao0);            \
        fa[i][1] = *(const bf16x8_t*)(smem + (buf_) * BUF_BYTES + ((qm_) * 64 + i * 16) * 128 + ao1);            \
    }
#define G8P_RD_B(buf_, qn_, dst_)                                                                                \
    _Pragma("unroll") for (int j = 0; j < 2; ++j) {                                                              \
        dst_[j][0] = *(const bf16x8_t*)(smem + (buf_) * BUF_BYTES + ((qn_) * 32 + j * 16) * 128 + bo0);          \
        dst_[j][1] = *(const bf16x8_t*)(smem + (buf_) * BUF_BYTES + ((qn_) * 32 + j * 16) * 128 + bo1);          \
    }
#define G8P_MM(qm_, qn_, fb_)                                                                                    \
    do {                                                                                                         \
        __builtin_amdgcn_s_setprio(1);                                                                           \
        _Pragma("unroll") for (int s = 0; s < 2; ++s)                                                            \
            _Pragma("unroll") for (int i = 0; i < 4; ++i)                                                        \
                _Pragma("unroll") for (int j = 0; j < 2; ++j)                                                    \
                    acc[(qm_) * 4 + i][(qn_) * 2 + j] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(                 \
                        fb_[j][s], fa[i][s], acc[(qm_) * 4 + i][(qn_) * 2 + j], 0, 0, 0);                        \
        __builtin_amdgcn_s_setprio(0);                                                                           \
    } while (0)
#define G8P_BAR()                                                                                                \
    do {                                                                                                         \
        __builtin_amdgcn_sched_barrier(0);                                                                       \
        __builtin_amdgcn_s_barrier();                                                                            \
        __builtin_amdgcn_sched_barrier(0);                                                                       \
    } while (0)

    // prologue: K-tile 0 complete + three half-tiles of K-tile 1 in flight
    stage_b(0, 0); stage_a(0, 0); stage_b(0, 1); stage_a(0, 1);
    stage_b(1, 0); stage_a(1, 0); stage_b(1, 1);
    asm volatile("s_waitcnt vmcnt(6)" ::: "memory");
    G8P_BAR();
    if (wr == 1) G8P_BAR();                              // waves 4-7 run one barrier behind

#pragma unroll 1
    for (int t = 0; t < nk; t += 2) {
        // ---- K-tile t (buffer 0)
        G8P_RD_B(0, 0, fb0);
        __builtin_amdgcn_sched_barrier(0);
        G8P_RD_A(0, 0);
        stage_a(t + 1, 1);
        asm volatile("s_waitcnt lgkmcnt(8)" ::: "memory");      // the four B0 reads have returned: B0 may be restaged next phase
        G8P_BAR(); G8P_MM(0, 0, fb0); G8P_BAR();

        G8P_RD_B(0, 1, fb1);
        stage_b(t + 2, 0);
        G8P_BAR(); G8P_MM(0, 1, fb1); G8P_BAR();

        G8P_RD_A(0, 1);
        stage_a(t + 2, 0);
        G8P_BAR(); G8P_MM(1, 1, fb1); G8P_BAR();

        stage_b(t + 2, 1);
        asm volatile("s_waitcnt vmcnt(6)" ::: "memory");        // K-tile t+1 has landed (this wave's part)
        G8P_BAR(); G8P_MM(1, 0, fb0); G8P_BAR();

        // ---- K-tile t + 1 (buffer 1)
        G8P_RD_B(1, 0, fb0);
        __builtin_amdgcn_sched_barrier(0);
        G8P_RD_A(1, 0);
        stage_a(t + 2, 1);
        asm volatile("s_waitcnt lgkmcnt(8)" ::: "memory");
        G8P_BAR(); G8P_MM(0, 0, fb0); G8P_BAR();

        G8P_RD_B(1, 1, fb1);
        stage_b(t + 3, 0);
        G8P_BAR(); G8P_MM(0, 1, fb1); G8P_BAR();

        G8P_RD_A(1, 1);
        stage_a(t + 3, 0);
        G8P_BAR(); G8P_MM(1, 1, fb1); G8P_BAR();

        stage_b(t + 3, 1);
        asm volatile("s_waitcnt vmcnt(6)" ::: "memory");
        G8P_BAR(); G8P_MM(1, 0, fb0); G8P_BAR();
    }
    if (wr == 0) G8P_BAR();                              // waves 0-3 wait for the staggered half's last phase
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");     // the tail's zero-fill DMAs
    G8P_BAR();
#undef G8P_RD_A
#undef G8P_RD_B
#undef G8P_MM
#undef G8P_BAR
}

}  // namespace g8p
