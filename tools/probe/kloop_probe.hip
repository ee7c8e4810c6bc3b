// kloop_probe.hip -- what bounds the NT GEMM K loop on MI355X? Standalone probe (not part of libqst).
// A 256-thread workgroup streams 128 x 64 (A) + 192 x 64 (B) bf16 stages by LDS-DMA into a 2-slot ring, exactly as
// gemm_nt_kernel does, with parts of the loop body switched off:
//   mode 0: DMA + barrier only          mode 1: + fragment reads (ds_read_b128)
//   mode 2: DMA + MFMA (no LDS reads)   mode 3: everything (the real loop)     mode 4: MFMA only (no DMA)
// src = 0: every workgroup streams its own rows of a buffer much larger than the caches (HBM/MALL);
// src = 1: every workgroup re-reads the same 40 KB (L2-resident).
// Prints ns per stage per workgroup, ingest bytes/clk/CU (at 2.4 GHz nominal) and MFMA utilisation.
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#include <cstdint>
#include <vector>

typedef __attribute__((ext_vector_type(8))) __bf16 bf16x8;
typedef __attribute__((ext_vector_type(16))) float f32x16;
typedef __attribute__((address_space(3))) void lds_void;

#define CHECK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { printf("HIP error %d at %s:%d\n", (int)e_, __FILE__, __LINE__); exit(1); } } while (0)

__device__ __forceinline__ __amdgpu_buffer_rsrc_t make_rsrc(const void* base, uint32_t bytes) {
    return __builtin_amdgcn_make_buffer_rsrc((void*)base, 0, bytes, 0x00020000);
}
__device__ __forceinline__ void dma16(__amdgpu_buffer_rsrc_t r, char* lds, uint32_t voff, uint32_t soff) {
    __builtin_amdgcn_raw_ptr_buffer_load_lds(r, (lds_void*)lds, 16, (int)voff, (int)soff, 0, 0);
}
__device__ __forceinline__ uint32_t nt_off(int row, int chunk) { return (uint32_t)(row * 128 + ((chunk ^ ((row >> 1) & 7)) << 4)); }

template <int MODE>
__global__ __launch_bounds__(256, 2) void probe(const char* A, const char* B, int lda_bytes, int ldb_bytes, int nk, int src,
                                                float* sink) {
    extern __shared__ __attribute__((aligned(16))) char smem[];
    constexpr int A_BYTES = 128 * 128, B_BYTES = 192 * 128, STAGE = A_BYTES + B_BYTES;
    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int wm = wave >> 1, wn = wave & 1, fr = lane & 31, fh = lane >> 5;
    const size_t blk = src ? 0 : blockIdx.x;
    const __amdgpu_buffer_rsrc_t ra = make_rsrc(A + blk * 128 * (size_t)lda_bytes, 128u * lda_bytes);
    const __amdgpu_buffer_rsrc_t rb = make_rsrc(B + (blk % 8) * 192 * (size_t)ldb_bytes, 192u * ldb_bytes);
    uint32_t va[4], vb[6];
    for (int t = 0; t < 4; ++t) { const int row = (wave * 4 + t) * 8 + (lane >> 3); va[t] = row * lda_bytes + (((lane & 7) ^ ((row >> 1) & 7)) * 16); }
    for (int t = 0; t < 6; ++t) { const int row = (wave * 6 + t) * 8 + (lane >> 3); vb[t] = row * ldb_bytes + (((lane & 7) ^ ((row >> 1) & 7)) * 16); }
    auto issue = [&](int kt) {
        char* st = smem + (kt & 1) * STAGE;
        const uint32_t ko = src ? 0u : (uint32_t)kt * 128u;
        for (int t = 0; t < 4; ++t) dma16(ra, st + (wave * 4 + t) * 1024, va[t], ko);
        for (int t = 0; t < 6; ++t) dma16(rb, st + A_BYTES + (wave * 6 + t) * 1024, vb[t], ko);
    };
    f32x16 acc[2][3];
    for (int i = 0; i < 2; ++i) for (int j = 0; j < 3; ++j) for (int r = 0; r < 16; ++r) acc[i][j][r] = 0.f;
    bf16x8 ca, cb;
    for (int e = 0; e < 8; ++e) { ca[e] = (__bf16)(0.001f * (lane + e)); cb[e] = (__bf16)(0.002f * (lane - e)); }
    if (MODE != 4) issue(0);
    for (int kt = 0; kt < nk; ++kt) {
        if (MODE != 4) {
            asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
            __builtin_amdgcn_s_barrier();
            if (kt + 1 < nk) issue(kt + 1);
        }
        const char* pa = smem + (kt & 1) * STAGE;
        const char* pb = pa + A_BYTES;
#pragma unroll
        for (int ks = 0; ks < 4; ++ks) {
            bf16x8 fa[2], fb[3];
            if (MODE == 1 || MODE == 3) {
                for (int i = 0; i < 2; ++i) fa[i] = *(const bf16x8*)(pa + nt_off(wm * 64 + i * 32 + fr, ks * 2 + fh));
                for (int j = 0; j < 3; ++j) fb[j] = *(const bf16x8*)(pb + nt_off(wn * 96 + j * 32 + fr, ks * 2 + fh));
            } else { fa[0] = fa[1] = ca; fb[0] = fb[1] = fb[2] = cb; }
            if (MODE >= 2) {
                for (int i = 0; i < 2; ++i) for (int j = 0; j < 3; ++j)
                    acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(fb[j], fa[i], acc[i][j], 0, 0, 0);
            } else if (MODE == 1) {
                for (int i = 0; i < 2; ++i) acc[i][0][0] += (float)fa[i][0];
                for (int j = 0; j < 3; ++j) acc[0][j][1] += (float)fb[j][0];
            }
        }
    }
    float s = 0.f;
    for (int i = 0; i < 2; ++i) for (int j = 0; j < 3; ++j) for (int r = 0; r < 16; ++r) s += acc[i][j][r];
    if (s == 12345.678f) sink[0] = s;          // keeps the work alive
}

// DMA-only streaming with an NSLOT-deep ring of 40 KB stages (NSLOT - 1 stages in flight): is the loop bound by the
// latency of one stage's fetch (then depth helps) or by a rate (then it does not)?
template <int NSLOT>
__global__ __launch_bounds__(256, 1) void probe_depth(const char* A, const char* B, int lda_bytes, int ldb_bytes, int nk, int src,
                                                      float* sink) {
    extern __shared__ __attribute__((aligned(16))) char smem[];
    constexpr int A_BYTES = 128 * 128, B_BYTES = 192 * 128, STAGE = A_BYTES + B_BYTES;
    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const size_t blk = src ? 0 : blockIdx.x;
    const __amdgpu_buffer_rsrc_t ra = make_rsrc(A + blk * 128 * (size_t)lda_bytes, 128u * lda_bytes);
    const __amdgpu_buffer_rsrc_t rb = make_rsrc(B + (blk % 8) * 192 * (size_t)ldb_bytes, 192u * ldb_bytes);
    uint32_t va[4], vb[6];
    for (int t = 0; t < 4; ++t) { const int row = (wave * 4 + t) * 8 + (lane >> 3); va[t] = row * lda_bytes + (((lane & 7) ^ ((row >> 1) & 7)) * 16); }
    for (int t = 0; t < 6; ++t) { const int row = (wave * 6 + t) * 8 + (lane >> 3); vb[t] = row * ldb_bytes + (((lane & 7) ^ ((row >> 1) & 7)) * 16); }
    auto issue = [&](int kt) {
        char* st = smem + (kt % NSLOT) * STAGE;
        const uint32_t ko = src ? 0u : (uint32_t)kt * 128u;
        for (int t = 0; t < 4; ++t) dma16(ra, st + (wave * 4 + t) * 1024, va[t], ko);
        for (int t = 0; t < 6; ++t) dma16(rb, st + A_BYTES + (wave * 6 + t) * 1024, vb[t], ko);
    };
    for (int kt = 0; kt < NSLOT - 1 && kt < nk; ++kt) issue(kt);
    for (int kt = 0; kt < nk; ++kt) {
        const int ahead = min(NSLOT - 2, nk - 1 - kt);          // 10 DMA instructions per stage per wave
        if (ahead >= 2) asm volatile("s_waitcnt vmcnt(20)" ::: "memory");
        else if (ahead == 1) asm volatile("s_waitcnt vmcnt(10)" ::: "memory");
        else asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        __builtin_amdgcn_s_barrier();
        if (kt + NSLOT - 1 < nk) issue(kt + NSLOT - 1);
    }
    if (smem[tid] == 123 && sink) sink[1] = 1.f;
}

template <int NSLOT>
static void run_depth(const char* A, const char* B, int ld, int nk, int src, float* sink) {
    const int lds = NSLOT * (128 * 128 + 192 * 128);
    CHECK(hipFuncSetAttribute((const void*)probe_depth<NSLOT>, hipFuncAttributeMaxDynamicSharedMemorySize, lds));
    hipEvent_t e0, e1;
    CHECK(hipEventCreate(&e0)); CHECK(hipEventCreate(&e1));
    const int blocks = 256;
    for (int w = 0; w < 2; ++w) probe_depth<NSLOT><<<blocks, 256, lds>>>(A, B, ld, ld, nk, src, sink);
    CHECK(hipEventRecord(e0));
    const int reps = 5;
    for (int r = 0; r < reps; ++r) probe_depth<NSLOT><<<blocks, 256, lds>>>(A, B, ld, ld, nk, src, sink);
    CHECK(hipEventRecord(e1));
    CHECK(hipEventSynchronize(e1));
    float ms = 0.f;
    CHECK(hipEventElapsedTime(&ms, e0, e1));
    ms /= reps;
    const double ns_stage = ms * 1e6 / nk;
    printf("DMA only, one workgroup per CU, %d-slot ring (%d stages in flight): %7.0f ns/stage  ingest %5.1f B/clk/CU\n", NSLOT,
           NSLOT - 1, ns_stage, 40960.0 / (ns_stage * 2.4));
}

// The full loop with 32-deep stages (20 KB) in an NSLOT-slot ring, two workgroups per CU: same LDS as the 2 x 40 KB ring
// when NSLOT = 4, NSLOT - 1 stages in flight, twice the barriers per K.
__device__ __forceinline__ uint32_t nt_off32(int row, int chunk) { return (uint32_t)(row * 64 + ((chunk ^ ((row >> 2) & 3)) << 4)); }
template <int NSLOT>
__global__ __launch_bounds__(256, 2) void probe32(const char* A, const char* B, int lda_bytes, int ldb_bytes, int nk64, int src,
                                                  float* sink) {
    extern __shared__ __attribute__((aligned(16))) char smem[];
    constexpr int A_BYTES = 128 * 64, B_BYTES = 192 * 64, STAGE = A_BYTES + B_BYTES;      // 20 KB
    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int wm = wave >> 1, wn = wave & 1, fr = lane & 31, fh = lane >> 5;
    const size_t blk = src ? 0 : blockIdx.x;
    const __amdgpu_buffer_rsrc_t ra = make_rsrc(A + blk * 128 * (size_t)lda_bytes, 128u * lda_bytes);
    const __amdgpu_buffer_rsrc_t rb = make_rsrc(B + (blk % 8) * 192 * (size_t)ldb_bytes, 192u * ldb_bytes);
    uint32_t va[2], vb[3];
    for (int t = 0; t < 2; ++t) { const int row = (wave * 2 + t) * 16 + (lane >> 2); va[t] = row * lda_bytes + (((lane & 3) ^ ((row >> 2) & 3)) * 16); }
    for (int t = 0; t < 3; ++t) { const int row = (wave * 3 + t) * 16 + (lane >> 2); vb[t] = row * ldb_bytes + (((lane & 3) ^ ((row >> 2) & 3)) * 16); }
    const int nk = nk64 * 2;
    auto issue = [&](int kt) {
        char* st = smem + (kt % NSLOT) * STAGE;
        const uint32_t ko = src ? 0u : (uint32_t)kt * 64u;
        for (int t = 0; t < 2; ++t) dma16(ra, st + (wave * 2 + t) * 1024, va[t], ko);
        for (int t = 0; t < 3; ++t) dma16(rb, st + A_BYTES + (wave * 3 + t) * 1024, vb[t], ko);
    };
    f32x16 acc[2][3];
    for (int i = 0; i < 2; ++i) for (int j = 0; j < 3; ++j) for (int r = 0; r < 16; ++r) acc[i][j][r] = 0.f;
    for (int kt = 0; kt < NSLOT - 1 && kt < nk; ++kt) issue(kt);
    for (int kt = 0; kt < nk; ++kt) {
        const int ahead = min(NSLOT - 2, nk - 1 - kt);          // 5 DMA instructions per stage per wave
        if (ahead >= 2) asm volatile("s_waitcnt vmcnt(10)" ::: "memory");
        else if (ahead == 1) asm volatile("s_waitcnt vmcnt(5)" ::: "memory");
        else asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        __builtin_amdgcn_s_barrier();
        if (kt + NSLOT - 1 < nk) issue(kt + NSLOT - 1);
        const char* pa = smem + (kt % NSLOT) * STAGE;
        const char* pb = pa + A_BYTES;
#pragma unroll
        for (int ks = 0; ks < 2; ++ks) {
            bf16x8 fa[2], fb[3];
            for (int i = 0; i < 2; ++i) fa[i] = *(const bf16x8*)(pa + nt_off32(wm * 64 + i * 32 + fr, ks * 2 + fh));
            for (int j = 0; j < 3; ++j) fb[j] = *(const bf16x8*)(pb + nt_off32(wn * 96 + j * 32 + fr, ks * 2 + fh));
            for (int i = 0; i < 2; ++i) for (int j = 0; j < 3; ++j)
                acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(fb[j], fa[i], acc[i][j], 0, 0, 0);
        }
    }
    float s = 0.f;
    for (int i = 0; i < 2; ++i) for (int j = 0; j < 3; ++j) for (int r = 0; r < 16; ++r) s += acc[i][j][r];
    if (s == 12345.678f) sink[0] = s;
}
template <int NSLOT>
static void run32(const char* A, const char* B, int ld, int nk64, int src, int blocks, float* sink) {
    const int lds = NSLOT * (128 * 64 + 192 * 64);
    CHECK(hipFuncSetAttribute((const void*)probe32<NSLOT>, hipFuncAttributeMaxDynamicSharedMemorySize, lds));
    hipEvent_t e0, e1;
    CHECK(hipEventCreate(&e0)); CHECK(hipEventCreate(&e1));
    for (int w = 0; w < 2; ++w) probe32<NSLOT><<<blocks, 256, lds>>>(A, B, ld, ld, nk64, src, sink);
    CHECK(hipEventRecord(e0));
    const int reps = 5;
    for (int r = 0; r < reps; ++r) probe32<NSLOT><<<blocks, 256, lds>>>(A, B, ld, ld, nk64, src, sink);
    CHECK(hipEventRecord(e1));
    CHECK(hipEventSynchronize(e1));
    float ms = 0.f;
    CHECK(hipEventElapsedTime(&ms, e0, e1));
    ms /= reps;
    const double rounds = (double)blocks / 512.0;
    const double ns_stage = ms * 1e6 / (nk64 * (rounds < 1 ? 1 : rounds));
    printf("everything, 32-deep stages, %d-slot ring, blocks %4d: %7.0f ns per 64 of K  MFMA %4.0f%%\n", NSLOT, blocks, ns_stage,
           100.0 * 2.0 * 768.0 / (ns_stage * 2.4));
}

// DMA rate against the contiguous run per row: a 40 KB stage made of rows of ROWB bytes (40960 / ROWB rows, each at its own
// leading-dimension stride), two workgroups per CU, 2-slot ring, no compute.
template <int ROWB>
__global__ __launch_bounds__(256, 2) void probe_row(const char* A, int ld_bytes, int nk, int src, float* sink) {
    extern __shared__ __attribute__((aligned(16))) char smem[];
    constexpr int STAGE = 40960, ROWS = STAGE / ROWB, LPR = ROWB / 16, RPI = 64 / LPR;      // lanes per row, rows per instruction
    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    // the A buffer holds 131072 rows of ld_bytes: keep every workgroup's ROWS rows inside it
    const size_t blk = src ? 0 : (blockIdx.x % (131072 / ROWS - 1));
    const __amdgpu_buffer_rsrc_t ra = make_rsrc(A + blk * ROWS * (size_t)ld_bytes, (uint32_t)ROWS * ld_bytes);
    uint32_t vo[10];
    for (int t = 0; t < 10; ++t) { const int row = (wave * 10 + t) * RPI + lane / LPR; vo[t] = row * ld_bytes + (lane % LPR) * 16; }
    auto issue = [&](int kt) {
        char* st = smem + (kt & 1) * STAGE;
        const uint32_t ko = src ? 0u : (uint32_t)kt * ROWB;
        for (int t = 0; t < 10; ++t) dma16(ra, st + (wave * 10 + t) * 1024, vo[t], ko);
    };
    issue(0);
    for (int kt = 0; kt < nk; ++kt) {
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        __builtin_amdgcn_s_barrier();
        if (kt + 1 < nk) issue(kt + 1);
    }
    if (smem[tid] == 123 && sink) sink[1] = 1.f;
}
template <int ROWB>
static void run_row(const char* A, int ld, int nk, int src, float* sink) {
    const int lds = 2 * 40960;
    CHECK(hipFuncSetAttribute((const void*)probe_row<ROWB>, hipFuncAttributeMaxDynamicSharedMemorySize, lds));
    hipEvent_t e0, e1;
    CHECK(hipEventCreate(&e0)); CHECK(hipEventCreate(&e1));
    for (int w = 0; w < 2; ++w) probe_row<ROWB><<<512, 256, lds>>>(A, ld, nk, src, sink);
    CHECK(hipEventRecord(e0));
    const int reps = 5;
    for (int r = 0; r < reps; ++r) probe_row<ROWB><<<512, 256, lds>>>(A, ld, nk, src, sink);
    CHECK(hipEventRecord(e1));
    CHECK(hipEventSynchronize(e1));
    float ms = 0.f;
    CHECK(hipEventElapsedTime(&ms, e0, e1));
    ms /= reps;
    const double ns_stage = ms * 1e6 / nk;
    printf("DMA only, 40 KB stages of %3d-byte rows, two workgroups per CU: %7.0f ns/stage  ingest %5.1f B/clk/CU\n", ROWB, ns_stage,
           2.0 * 40960.0 / (ns_stage * 2.4));
}

// The full loop with v_mfma_f32_16x16x32_bf16 (4 x 6 tiles of 16 x 16 per wave) instead of 32x32x16 (2 x 3 tiles of 32 x 32):
// same MFMA cycles per FLOP and the same LDS bytes, but the guide reports a higher sustained clock for this shape.
typedef __attribute__((ext_vector_type(4))) float f32x4;
__global__ __launch_bounds__(256, 2) void probe16(const char* A, const char* B, int lda_bytes, int ldb_bytes, int nk, int src,
                                                  float* sink) {
    extern __shared__ __attribute__((aligned(16))) char smem[];
    constexpr int A_BYTES = 128 * 128, B_BYTES = 192 * 128, STAGE = A_BYTES + B_BYTES;
    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int wm = wave >> 1, wn = wave & 1, fr = lane & 15, fq = lane >> 4;
    const size_t blk = src ? 0 : blockIdx.x;
    const __amdgpu_buffer_rsrc_t ra = make_rsrc(A + blk * 128 * (size_t)lda_bytes, 128u * lda_bytes);
    const __amdgpu_buffer_rsrc_t rb = make_rsrc(B + (blk % 8) * 192 * (size_t)ldb_bytes, 192u * ldb_bytes);
    uint32_t va[4], vb[6];
    for (int t = 0; t < 4; ++t) { const int row = (wave * 4 + t) * 8 + (lane >> 3); va[t] = row * lda_bytes + (((lane & 7) ^ ((row >> 1) & 7)) * 16); }
    for (int t = 0; t < 6; ++t) { const int row = (wave * 6 + t) * 8 + (lane >> 3); vb[t] = row * ldb_bytes + (((lane & 7) ^ ((row >> 1) & 7)) * 16); }
    auto issue = [&](int kt) {
        char* st = smem + (kt & 1) * STAGE;
        const uint32_t ko = src ? 0u : (uint32_t)kt * 128u;
        for (int t = 0; t < 4; ++t) dma16(ra, st + (wave * 4 + t) * 1024, va[t], ko);
        for (int t = 0; t < 6; ++t) dma16(rb, st + A_BYTES + (wave * 6 + t) * 1024, vb[t], ko);
    };
    f32x4 acc[4][6];
    for (int i = 0; i < 4; ++i) for (int j = 0; j < 6; ++j) for (int r = 0; r < 4; ++r) acc[i][j][r] = 0.f;
    issue(0);
    for (int kt = 0; kt < nk; ++kt) {
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        __builtin_amdgcn_s_barrier();
        if (kt + 1 < nk) issue(kt + 1);
        const char* pa = smem + (kt & 1) * STAGE;
        const char* pb = pa + A_BYTES;
#pragma unroll
        for (int ks = 0; ks < 2; ++ks) {               // two k32 steps per 64-deep stage
            bf16x8 fa[4], fb[6];
            for (int i = 0; i < 4; ++i) fa[i] = *(const bf16x8*)(pa + nt_off(wm * 64 + i * 16 + fr, ks * 4 + fq));
            for (int j = 0; j < 6; ++j) fb[j] = *(const bf16x8*)(pb + nt_off(wn * 96 + j * 16 + fr, ks * 4 + fq));
            for (int i = 0; i < 4; ++i) for (int j = 0; j < 6; ++j)
                acc[i][j] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(fb[j], fa[i], acc[i][j], 0, 0, 0);
        }
    }
    float s = 0.f;
    for (int i = 0; i < 4; ++i) for (int j = 0; j < 6; ++j) for (int r = 0; r < 4; ++r) s += acc[i][j][r];
    if (s == 12345.678f) sink[0] = s;
}
static void run16(const char* A, const char* B, int ld, int nk, int src, int blocks, float* sink) {
    const int lds = 2 * (128 * 128 + 192 * 128);
    CHECK(hipFuncSetAttribute((const void*)probe16, hipFuncAttributeMaxDynamicSharedMemorySize, lds));
    hipEvent_t e0, e1;
    CHECK(hipEventCreate(&e0)); CHECK(hipEventCreate(&e1));
    for (int w = 0; w < 2; ++w) probe16<<<blocks, 256, lds>>>(A, B, ld, ld, nk, src, sink);
    CHECK(hipEventRecord(e0));
    const int reps = 5;
    for (int r = 0; r < reps; ++r) probe16<<<blocks, 256, lds>>>(A, B, ld, ld, nk, src, sink);
    CHECK(hipEventRecord(e1));
    CHECK(hipEventSynchronize(e1));
    float ms = 0.f;
    CHECK(hipEventElapsedTime(&ms, e0, e1));
    ms /= reps;
    const double rounds = (double)blocks / 512.0;
    const double ns_stage = ms * 1e6 / (nk * (rounds < 1 ? 1 : rounds));
    printf("everything with v_mfma_f32_16x16x32_bf16, blocks %4d:          %7.0f ns/stage  MFMA %4.0f%%\n", blocks, ns_stage,
           100.0 * 2.0 * 768.0 / (ns_stage * 2.4));
}

template <int MODE>
static void run(const char* A, const char* B, int ld, int nk, int src, int blocks, float* sink, const char* what) {
    const int lds = 2 * (128 * 128 + 192 * 128);
    CHECK(hipFuncSetAttribute((const void*)probe<MODE>, hipFuncAttributeMaxDynamicSharedMemorySize, lds));
    hipEvent_t e0, e1;
    CHECK(hipEventCreate(&e0)); CHECK(hipEventCreate(&e1));
    for (int w = 0; w < 2; ++w) probe<MODE><<<blocks, 256, lds>>>(A, B, ld, ld, nk, src, sink);
    CHECK(hipEventRecord(e0));
    const int reps = 5;
    for (int r = 0; r < reps; ++r) probe<MODE><<<blocks, 256, lds>>>(A, B, ld, ld, nk, src, sink);
    CHECK(hipEventRecord(e1));
    CHECK(hipEventSynchronize(e1));
    float ms = 0.f;
    CHECK(hipEventElapsedTime(&ms, e0, e1));
    ms /= reps;
    const double rounds = (double)blocks / 512.0;                     // 2 workgroups per CU, 256 CUs
    const double ns_stage = ms * 1e6 / (nk * (rounds < 1 ? 1 : rounds));
    const double per_cu_blocks = blocks >= 512 ? 2.0 : (double)blocks / 256.0;
    const double bytes_clk = per_cu_blocks * 40960.0 / (ns_stage * 2.4);
    const double mfma = per_cu_blocks * 768.0 / (ns_stage * 2.4);       // each workgroup-stage = 768 MFMA cycles on each of the 4 SIMDs
    printf("%-44s blocks %5d  %8.1f us  %7.0f ns/stage  ingest %5.1f B/clk/CU  MFMA %4.0f%%\n", what, blocks, ms * 1e3, ns_stage,
           MODE == 4 ? 0.0 : bytes_clk, MODE >= 2 ? 100.0 * mfma : 0.0);
}

int main() {
    const int ld = 1536 * 2, nk = 24;                                 // K = 1536
    const size_t abytes = (size_t)32768 * 4 * ld, bbytes = (size_t)8 * 192 * ld;
    char *A, *B; float* sink;
    CHECK(hipMalloc(&A, abytes)); CHECK(hipMalloc(&B, bbytes)); CHECK(hipMalloc(&sink, 16));
    CHECK(hipMemset(A, 1, abytes)); CHECK(hipMemset(B, 1, bbytes));
    for (int src = 0; src < 2; ++src) {
        printf("---- %s\n", src ? "every workgroup re-reads the same 40 KB (L2-resident)" : "every workgroup streams its own rows (A from HBM/MALL)");
        for (int blocks : {256, 512, 1024}) {
            run<0>(A, B, ld, nk, src, blocks, sink, "DMA + barrier");
            run<1>(A, B, ld, nk, src, blocks, sink, "DMA + barrier + fragment reads");
            run<2>(A, B, ld, nk, src, blocks, sink, "DMA + barrier + MFMA (operands in registers)");
            run<3>(A, B, ld, nk, src, blocks, sink, "everything");
        }
        run16(A, B, ld, nk, src, 512, sink);
        run16(A, B, ld, nk, src, 1024, sink);
        run_row<64>(A, ld, 12, src, sink);
        run_row<128>(A, ld, 12, src, sink);
        run_row<256>(A, ld, 12, src, sink);
        run_row<512>(A, ld, 6, src, sink);
        run32<2>(A, B, ld, nk, src, 512, sink);
        run32<3>(A, B, ld, nk, src, 512, sink);
        run32<4>(A, B, ld, nk, src, 512, sink);
        run32<4>(A, B, ld, nk, src, 1024, sink);
        run_depth<2>(A, B, ld, nk, src, sink);
        run_depth<3>(A, B, ld, nk, src, sink);
        run_depth<4>(A, B, ld, nk, src, sink);
    }
    run<4>(A, B, ld, nk, 1, 512, sink, "MFMA only");
    run<4>(A, B, ld, nk, 1, 1024, sink, "MFMA only");
    return 0;
}
