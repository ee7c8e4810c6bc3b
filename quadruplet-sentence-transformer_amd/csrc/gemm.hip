// gemm.hip -- bf16 MFMA GEMMs for the encoder (gfx950, wave64, v_mfma_f32_32x32x16_bf16).
//
//   gemm_nt : C[M,N] = A[M,K] . B[N,K]^T  (+ fused epilogue)   forward Linear and dgrad (B = W^T shadow)
//   gemm_tn : C[N,K] += A[M,N]^T . B[M,K]  (fp32 atomics, split over M)  wgrad; optional column sums (bias grad)
//
// These replace nn.Linear forward/backward inside BertLayer (transformers modeling_bert.py:154-156,
// 282-293, 325-351).
//
// Kernels in this file (each is described in detail at its definition):
//   gemm_nt_kernel<EPI, WAVES_M, WAVES_N, TI>  128x192 tile, 4 waves of 64x96, two workgroups per CU (default); 256x192 with 8
//                       waves; 256x192 with four waves of 128x96 ("tall", 32-deep stages) for the K >= 768 bf16-output GEMMs.
//                       K in 64-deep stages by LDS-DMA into a 2-slot XOR-swizzled ring, one raw s_barrier per stage.
//   gemm_nt_f8_kernel   both operands MXFP8 on v_mfma_scale_f32_32x32x64_f8f6f4, 128-deep stages (QST_PREC_FP8, inference)
//   gemm_nt_ln_kernel<MODE, DROPW>  128x384 full-row tile with the LayerNorm (forward) / LayerNorm backward in the epilogue
//   gemm_tn_group_kernel            all weight gradients of a layer in one launch, 192x192 tiles, one M-range per XCD,
//                       4 MFMA + 4 loader waves, 3-slot ring of 64-row stages, float-atomic flush
//   quant_mx_kernel     MXFP8 quantisation of an activation matrix
// LDS images are XOR-swizzled (applied on the DMA SOURCE address, the destination is lane-linear) so that
// ds_read_b128 (nt) and ds_read_b64_tr_b16 (tn) fragment reads are bank-conflict-free; LDS-DMA = buffer_load_dwordx4 ... lds
// (no staging registers, hardware range check zero-fills ragged M / N).
#include "qst_common.h"
#include "qst_kernels.h"

namespace {

constexpr uint32_t kOOB = 0x7FFFFFF0u;              // voffset that always fails the buffer range check -> zero fill

typedef __attribute__((address_space(3))) void lds_void;
typedef __attribute__((ext_vector_type(2))) float f32x2;

// ONCE: no other workgroup reads these lines -> non-temporal (aux bit 1), see st_stream. Only for the LayerNorm-fused
// kernel's activation operand: on operands that other workgroups re-read it costs (same-box A/B of the whole step: the
// plain NT kernels' A tiles 4.60 -> 4.68 ms, the wgrad kernel's operand stages 4.60 -> 4.65 ms).
template <bool ONCE = false>
__device__ __forceinline__ void dma16(__amdgpu_buffer_rsrc_t r, char* lds_wave_base, uint32_t voff, uint32_t soff) {
    // one wave-instruction writes 64 x 16 B = 1 KB at lds_wave_base + lane*16 (base must be wave-uniform)
    __builtin_amdgcn_raw_ptr_buffer_load_lds(r, (lds_void*)lds_wave_base, 16, (int)voff, (int)soff, 0,
                                             (ONCE && QST_STREAM_LOADS) ? 2 : 0);
}
template <int N> __device__ __forceinline__ void wait_vmcnt() {
    if (N == 0) asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    else if (N == 4) asm volatile("s_waitcnt vmcnt(4)" ::: "memory");
    else asm volatile("s_waitcnt vmcnt(8)" ::: "memory");
}
template <int N> __device__ __forceinline__ void wait_vmcnt_n() {
    if (N == 0) asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    else if (N == 6) asm volatile("s_waitcnt vmcnt(6)" ::: "memory");
    else if (N == 12) asm volatile("s_waitcnt vmcnt(12)" ::: "memory");
    else if (N == 18) asm volatile("s_waitcnt vmcnt(18)" ::: "memory");
    else asm volatile("s_waitcnt vmcnt(24)" ::: "memory");
}
// wait until all but the `stages_behind` most recently issued stages (4 DMA per thread each) have landed
__device__ __forceinline__ void wait_stage(int stages_behind) {
    if (stages_behind >= 2) wait_vmcnt<8>();
    else if (stages_behind == 1) wait_vmcnt<4>();
    else wait_vmcnt<0>();
}

__device__ __forceinline__ int xcd_remap(int b, int nwg) {
    // bijective XCD-contiguous remap (blocks b and b+8 share an XCD): neighbours in the remapped id
    // share an A row-panel in the same L2
    const int q = nwg >> 3, r = nwg & 7, x = b & 7;
    return (x < r ? x * (q + 1) : r * (q + 1) + (x - r) * q) + (b >> 3);
}

// ---------------------------------------------------------------- NT
// 256 x 192 output tile, 512 threads = 8 waves as 4 (M) x 2 (N), 64 x 96 per wave (2 x 3 MFMA tiles).
// Why this shape: in-kernel stamps on the 128x128 version showed every stage waiting ~850 cycles on its own DMA
// and ~1100 cycles computing -- the CU's L2->LDS ingest (<= ~56 B/clk), not HBM and not the MFMA pipe, was the
// limiter (the kernel ran at the same speed with zero-filled out-of-range operands). Ingest per MFMA-cycle scales
// with (BM+BN)/(BM*BN): 1/64 at 128x128, 1/110 here. 192 divides every N of the three model families
// (384/1152/1536, 768/2304/3072), so no tile is padded along N, and M = 32768 gives 1-4 whole tiles per CU.
// LDS image per operand tile: [rows][64 bf16] = 128-byte rows (whole cache lines per DMA row); 16-byte chunk c of
// row r sits at chunk position c ^ ((r >> 1) & 7)  (conflict-free for the MFMA fragment ds_read_b128).
constexpr int NBK = 64;
// WAVES_M = 4: 256 x 192 tile, 8 waves, 112 KB ring (one workgroup per CU)
// WAVES_M = 2: 128 x 192 tile, 4 waves,  80 KB ring (two workgroups per CU: their phases interleave) -- used when the
//              256-row tiling would give fewer than two workgroups per CU (N = 384 GEMMs at M = 32768)
constexpr int NT_STG_LD = 100;                       // staging row stride in floats (400 B: ds_write_b128 conflict-free)
__device__ __forceinline__ uint32_t nt_off(int row, int chunk) {
    return (uint32_t)(row * 128 + ((chunk ^ ((row >> 1) & 7)) << 4));
}

// Prologue + K loop shared by every NT kernel: leaves the wave's 64 x 96 sub-tile in acc (D rows = n, D column = m)
// and returns after a workgroup barrier, so the caller may reuse the ring for its epilogue.
struct NoHook { __device__ __forceinline__ void operator()() const {} };
// after_first_issue: run once, right after the first stage's DMAs are on their way (work whose own memory latency should
// overlap that first round trip instead of preceding it)
// (Tried here twice and removed: an L2 touch-ahead of the activation operand -- one 4-byte LDS-DMA per lane on the line a
// row needs three stages later, as csrc/ffn.hip does. Round 2: step 4.802 vs 4.825 ms, +3..9% on back-to-back launches.
// Round 3, on the LayerNorm-fused kernel with operands flushed from the caches as they are inside the step
// (tools/mall_probe.py): K = 1536 cold 111 vs 97 us, warm 62 vs 58 -- the cold penalty is not a per-stage latency. Nor is it
// the access pattern: a tile-blocked A, every stage one contiguous 16 KB read, measured 94 vs 98 us cold.)
// A_SLOTS = 3 (the LayerNorm-fused kernel: one workgroup per CU, 128 x 384 tile): the activation stages get a ring of their own,
// three slots deep, so TWO stages of A rows -- the lines that come from HBM; the weight stages hit in L2 -- are in flight under
// every stage of MFMAs instead of one: ring = 3 x 16 KB of A + 2 x 48 KB of B = 144 KB.
// SPLITW: the split-weight forms (QstGemmArgs.B2) are compiled in -- forward instantiations only: the LayerNorm-backward kernel
// sits at the register limit and spilled with the extra loop in it.
template <int WAVES_M, int WAVES_N, typename HOOK = NoHook, bool A_ONCE = false, int A_SLOTS = 2, bool SPLITW = true>
__device__ __forceinline__ void nt_mainloop(const QstGemmArgs& g, char* smem, int m0, int n0, f32x16 (&acc)[2][3],
                                            HOOK after_first_issue = HOOK()) {
    constexpr int NBM = 64 * WAVES_M, NBN = 96 * WAVES_N, NW = WAVES_M * WAVES_N;
    constexpr int NT_A_BYTES = NBM * NBK * 2, NT_B_BYTES = NBN * NBK * 2;
    constexpr int NT_STAGE = NT_A_BYTES + NT_B_BYTES;
    constexpr int A_PER_WAVE = (NBM / 8) / NW;           // one DMA instruction = 8 tile rows
    constexpr int B_PER_WAVE = (NBN / 8) / NW;
    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int wm = wave / WAVES_N, wn = wave % WAVES_N;

    const int rows_a = min(NBM, g.M - m0), rows_b = min(NBN, g.N - n0);
    const op16* Ab = (const op16*)g.A + (size_t)m0 * g.lda;
    const op16* Bb = (const op16*)g.B + (size_t)n0 * g.ldb;
    const __amdgpu_buffer_rsrc_t ra = make_rsrc(Ab, (uint32_t)rows_a * g.lda * 2u);
    const __amdgpu_buffer_rsrc_t rb = make_rsrc(Bb, (uint32_t)rows_b * g.ldb * 2u);
    // Split weights (QstGemmArgs.B2, QST_PREC_F16W): C = A . (B + B2)^T as a second pass over K -- K-tile kt >= K / 64 takes
    // the A rows of K-tile kt - K / 64 again and the B rows from B2 (same shape and leading dimension as B): the loop below
    // simply runs twice as many stages; fragments, ring and epilogue do not know.
    const bool split = SPLITW && g.B2 != nullptr && n0 + NBN > g.b2_n0;      // (b2_n0: tiles left of it keep the single pass)
    const __amdgpu_buffer_rsrc_t rb2 = split ? make_rsrc((const op16*)g.B2 + (size_t)n0 * g.ldb, (uint32_t)rows_b * g.ldb * 2u) : rb;
    const int nk1 = g.K / NBK;

    // DMA map: one wave-instruction = 1 KB = 8 rows x 128 B. LDS position p (16-B units) = q*64 + lane -> row p/8,
    // chunk position p%8 -> logical chunk = pos ^ swz(row).
    uint32_t va[A_PER_WAVE], vb[B_PER_WAVE];
#pragma unroll
    for (int t = 0; t < A_PER_WAVE; ++t) {
        const int row = (wave * A_PER_WAVE + t) * 8 + (lane >> 3);
        va[t] = (uint32_t)row * g.lda * 2u + (uint32_t)(((lane & 7) ^ ((row >> 1) & 7)) * 16);
    }
#pragma unroll
    for (int t = 0; t < B_PER_WAVE; ++t) {
        const int row = (wave * B_PER_WAVE + t) * 8 + (lane >> 3);
        vb[t] = (uint32_t)row * g.ldb * 2u + (uint32_t)(((lane & 7) ^ ((row >> 1) & 7)) * 16);
    }
    constexpr bool DEEP = A_SLOTS == 3;
    auto slot_a = [&](int kt) { return smem + (DEEP ? (kt % 3) * NT_A_BYTES : (kt & 1) * NT_STAGE); };
    auto slot_b = [&](int kt) { return smem + (DEEP ? 3 * NT_A_BYTES + (kt & 1) * NT_B_BYTES : (kt & 1) * NT_STAGE + NT_A_BYTES); };
    auto issue_a = [&](int kt) {
        char* st = slot_a(kt);
        const uint32_t ko = (uint32_t)(kt >= nk1 ? kt - nk1 : kt) * (NBK * 2);
#pragma unroll
        for (int t = 0; t < A_PER_WAVE; ++t) dma16<A_ONCE>(ra, st + (wave * A_PER_WAVE + t) * 1024, va[t], ko);
    };
    auto issue_b = [&](int kt) {
        char* st = slot_b(kt);
        const bool lo = kt >= nk1;
        const uint32_t ko = (uint32_t)(lo ? kt - nk1 : kt) * (NBK * 2);
        const __amdgpu_buffer_rsrc_t r = lo ? rb2 : rb;
#pragma unroll
        for (int t = 0; t < B_PER_WAVE; ++t) dma16(r, st + (wave * B_PER_WAVE + t) * 1024, vb[t], ko);
    };
    auto issue = [&](int kt) { issue_a(kt); issue_b(kt); };

#pragma unroll
    for (int i = 0; i < 2; ++i)
#pragma unroll
        for (int j = 0; j < 3; ++j)
#pragma unroll
            for (int r = 0; r < 16; ++r) acc[i][j][r] = 0.f;

    const int nk = split ? 2 * nk1 : nk1;
    const int fr = lane & 31, fh = lane >> 5;
    if (DEEP && split) {
        // Split weights on the kernel whose activation stages have a ring of their own: the two weight stages of a K-tile
        // (high halves, low halves) follow one another against the SAME activation stage -- the activation rows are read once
        // (the second pass over K above would fetch the [M, K] panel a second time: 100 MB of HBM reads at K = 1536, the rows
        // carry the read-once hint). Step s = 2 kt + (0: high | 1: low); weight slot s & 1, activation slot kt % 3.
        auto issue_bs = [&](int s_) {
            char* st = slot_b(s_);
            const uint32_t ko = (uint32_t)(s_ >> 1) * (NBK * 2);
            const __amdgpu_buffer_rsrc_t r = (s_ & 1) ? rb2 : rb;
#pragma unroll
            for (int t = 0; t < B_PER_WAVE; ++t) dma16(r, st + (wave * B_PER_WAVE + t) * 1024, vb[t], ko);
        };
        issue_a(0);
        issue_bs(0);
        after_first_issue();
        if (nk1 > 1) issue_a(1);
        const int ns = 2 * nk1;
        for (int s_ = 0; s_ < ns; ++s_) {
            const int kt = s_ >> 1;
            // queue, oldest first -- even step: [A(kt + 1)], B(s): everything must have landed (A(kt + 1) has had two steps);
            // odd step: B(s), [A(kt + 2)]: the activation stage requested one step ago may stay in flight. Step 0 drains the hook's loads.
            if ((s_ & 1) && kt + 2 < nk1) asm volatile("s_waitcnt vmcnt(%0)" :: "n"(A_PER_WAVE) : "memory");
            else wait_vmcnt<0>();
            __builtin_amdgcn_s_barrier();
            if (s_ + 1 < ns) issue_bs(s_ + 1);
            if (!(s_ & 1) && kt + 2 < nk1) issue_a(kt + 2);
            const char* pa = slot_a(kt);
            const char* pb = slot_b(s_);
#pragma unroll
            for (int ks = 0; ks < 4; ++ks) {
                op16x8 fa[2], fb[3];
#pragma unroll
                for (int i = 0; i < 2; ++i) fa[i] = *(const op16x8*)(pa + nt_off(wm * 64 + i * 32 + fr, ks * 2 + fh));
#pragma unroll
                for (int j = 0; j < 3; ++j) fb[j] = *(const op16x8*)(pb + nt_off(wn * 96 + j * 32 + fr, ks * 2 + fh));
#pragma unroll
                for (int i = 0; i < 2; ++i)
#pragma unroll
                    for (int j = 0; j < 3; ++j) acc[i][j] = mfma32_op(fb[j], fa[i], acc[i][j]);
            }
        }
        __builtin_amdgcn_s_barrier();
        return;
    }
    issue(0);
    after_first_issue();
    for (int kt = 0; kt < nk; ++kt) {
        // stage kt has landed for this wave's DMAs. DEEP: the queue holds, oldest first, A(kt), B(kt), A(kt + 1) -- the last
        // one may stay in flight (A_PER_WAVE instructions; stage 0 drains everything, the hook's loads included)
        if (DEEP && kt > 0 && kt + 1 < nk) asm volatile("s_waitcnt vmcnt(%0)" :: "n"(A_PER_WAVE) : "memory");
        else wait_vmcnt<0>();
        __builtin_amdgcn_s_barrier();                 // ... for everyone's; and everyone is done reading the slots of stage kt - 1
        if (DEEP) {
            if (kt + 1 < nk) { issue_b(kt + 1); if (kt == 0) issue_a(1); }
            if (kt + 2 < nk) issue_a(kt + 2);
        } else if (kt + 1 < nk) issue(kt + 1);
        const char* pa = slot_a(kt);
        const char* pb = slot_b(kt);
#pragma unroll
        for (int ks = 0; ks < 4; ++ks) {
            op16x8 fa[2], fb[3];
#pragma unroll
            for (int i = 0; i < 2; ++i) fa[i] = *(const op16x8*)(pa + nt_off(wm * 64 + i * 32 + fr, ks * 2 + fh));
#pragma unroll
            for (int j = 0; j < 3; ++j) fb[j] = *(const op16x8*)(pb + nt_off(wn * 96 + j * 32 + fr, ks * 2 + fh));
#pragma unroll
            for (int i = 0; i < 2; ++i)
#pragma unroll
                for (int j = 0; j < 3; ++j)
                    acc[i][j] = mfma32_op(fb[j], fa[i], acc[i][j]);   // D rows = n, col = m
        }
    }
    __builtin_amdgcn_s_barrier();                     // all waves done with the ring before the epilogue reuses it
}

// Tall variant of the K loop: each wave owns 128 x 96 (4 x 3 MFMA tiles, 192 accumulator registers), the workgroup
// 256 x 192 with the same four waves. Per MFMA it reads 30% fewer fragment bytes from LDS and receives 30% fewer DMA
// bytes than the 64 x 96 wave tile (whose K loop keeps the LDS array ~83% busy at MFMA peak). Stages are 32 deep
// (64-byte LDS rows, chunk c of row r at position c ^ ((r >> 2) & 3)) so that two workgroups still fit on a CU.
__device__ __forceinline__ uint32_t nt_off32(int row, int chunk) {
    return (uint32_t)(row * 64 + ((chunk ^ ((row >> 2) & 3)) << 4));
}
template <int WAVES_M, int WAVES_N>
__device__ __forceinline__ void nt_mainloop_tall(const QstGemmArgs& g, char* smem, int m0, int n0, f32x16 (&acc)[4][3]) {
    constexpr int BK = 32;
    constexpr int NBM = 128 * WAVES_M, NBN = 96 * WAVES_N, NW = WAVES_M * WAVES_N;
    constexpr int A_BYTES = NBM * BK * 2, B_BYTES = NBN * BK * 2, STAGE = A_BYTES + B_BYTES;
    constexpr int A_PER_WAVE = (NBM / 16) / NW;          // one DMA instruction = 16 tile rows of 64 bytes
    constexpr int B_PER_WAVE = (NBN / 16) / NW;
    static_assert(A_PER_WAVE * NW * 16 == NBM && B_PER_WAVE * NW * 16 == NBN, "tile rows must split evenly over waves");
    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int wm = wave / WAVES_N, wn = wave % WAVES_N;
    const int rows_a = min(NBM, g.M - m0), rows_b = min(NBN, g.N - n0);
    const __amdgpu_buffer_rsrc_t ra = make_rsrc((const op16*)g.A + (size_t)m0 * g.lda, (uint32_t)rows_a * g.lda * 2u);
    const __amdgpu_buffer_rsrc_t rb = make_rsrc((const op16*)g.B + (size_t)n0 * g.ldb, (uint32_t)rows_b * g.ldb * 2u);
    const bool split = g.B2 != nullptr && n0 + NBN > g.b2_n0;                  // (see nt_mainloop)
    const __amdgpu_buffer_rsrc_t rb2 = split ? make_rsrc((const op16*)g.B2 + (size_t)n0 * g.ldb, (uint32_t)rows_b * g.ldb * 2u) : rb;
    const int nk1 = g.K / BK;
    uint32_t va[A_PER_WAVE], vb[B_PER_WAVE];
#pragma unroll
    for (int t = 0; t < A_PER_WAVE; ++t) {
        const int row = (wave * A_PER_WAVE + t) * 16 + (lane >> 2);
        va[t] = (uint32_t)row * g.lda * 2u + (uint32_t)(((lane & 3) ^ ((row >> 2) & 3)) * 16);
    }
#pragma unroll
    for (int t = 0; t < B_PER_WAVE; ++t) {
        const int row = (wave * B_PER_WAVE + t) * 16 + (lane >> 2);
        vb[t] = (uint32_t)row * g.ldb * 2u + (uint32_t)(((lane & 3) ^ ((row >> 2) & 3)) * 16);
    }
    auto issue = [&](int kt) {
        char* st = smem + (kt & 1) * STAGE;
        const bool lo = kt >= nk1;
        const uint32_t ko = (uint32_t)(lo ? kt - nk1 : kt) * (BK * 2);
        const __amdgpu_buffer_rsrc_t r = lo ? rb2 : rb;
#pragma unroll
        for (int t = 0; t < A_PER_WAVE; ++t) dma16(ra, st + (wave * A_PER_WAVE + t) * 1024, va[t], ko);
#pragma unroll
        for (int t = 0; t < B_PER_WAVE; ++t) dma16(r, st + A_BYTES + (wave * B_PER_WAVE + t) * 1024, vb[t], ko);
    };
#pragma unroll
    for (int i = 0; i < 4; ++i)
#pragma unroll
        for (int j = 0; j < 3; ++j)
#pragma unroll
            for (int r = 0; r < 16; ++r) acc[i][j][r] = 0.f;
    const int nk = split ? 2 * nk1 : nk1;
    const int fr = lane & 31, fh = lane >> 5;
    issue(0);
    for (int kt = 0; kt < nk; ++kt) {
        wait_vmcnt<0>();
        __builtin_amdgcn_s_barrier();
        if (kt + 1 < nk) issue(kt + 1);
        const char* pa = smem + (kt & 1) * STAGE;
        const char* pb = pa + A_BYTES;
#pragma unroll
        for (int ks = 0; ks < 2; ++ks) {
            op16x8 fa[4], fb[3];
#pragma unroll
            for (int i = 0; i < 4; ++i) fa[i] = *(const op16x8*)(pa + nt_off32(wm * 128 + i * 32 + fr, ks * 2 + fh));
#pragma unroll
            for (int j = 0; j < 3; ++j) fb[j] = *(const op16x8*)(pb + nt_off32(wn * 96 + j * 32 + fr, ks * 2 + fh));
#pragma unroll
            for (int i = 0; i < 4; ++i)
#pragma unroll
                for (int j = 0; j < 3; ++j)
                    acc[i][j] = mfma32_op(fb[j], fa[i], acc[i][j]);
        }
    }
    __builtin_amdgcn_s_barrier();
}

// Epilogue of one wave's (32*TI) x 96 sub-tile whose first element is C[m_base][n_base]. The MFMA operands were
// swapped (D rows = n in registers, D column = m on the lane), so each lane holds 4 consecutive n per register group:
// 32 rows x 96 columns at a time go through the wave-private LDS region `stg` as [m][n] (conflict-free ds_write_b128)
// and are read back row-wise, so bias / residual / GELU and the global accesses run on 16-byte row-contiguous vectors.
// All global LOADS of a pass (residual / saved gelu') are issued before the pass touches LDS and before any store:
// with loads and stores interleaved per row the compiler must keep them in order (C may alias resid), and in-kernel
// stamps showed the epilogue then costing 2-4x the whole K loop in exposed load latency.
// bias_s: this wave's 96 bias values in LDS (written by the caller, same wave).
template <int EPI, int TI>
__device__ __forceinline__ void nt_epilogue(const QstGemmArgs& g, f32x16 (&acc)[TI][3], float* stg, const float* bias_s,
                                            int m_base, int n_base, int lane) {
    const int fr = lane & 31, fh = lane >> 5;
    constexpr bool kF32Out = (EPI == QST_EPI_F32_RESID || EPI == QST_EPI_F32_RESID_BF16);
    DropCtx dc = DropCtx{0u, 0u, 1.f};               // dropout of the projection output, before the residual (QstGemmArgs)
    if (kF32Out && g.drop_where == 1) dc = drop_ctx(g.drop);
#pragma unroll
    for (int i = 0; i < TI; ++i) {
        if (kF32Out) {
            // fp32 outputs: 4 columns (16 B) per lane, 24 lanes per row, 12 items per lane
            f32x4 rv[12];
#pragma unroll
            for (int t = 0; t < 12; ++t) {
                const int idx = t * 64 + lane;
                const int row = idx / 24, c4 = idx % 24;
                const int m = m_base + i * 32 + row;
                const int n = n_base + c4 * 4;
                const f32x4 z = {0.f, 0.f, 0.f, 0.f};
                rv[t] = (m < g.M && n < g.N && g.resid) ? ld_stream((const f32x4*)(g.resid + (size_t)m * g.ldr + n)) : z;
            }
#pragma unroll
            for (int j = 0; j < 3; ++j)
#pragma unroll
                for (int g4 = 0; g4 < 4; ++g4) {
                    f32x4 v;
#pragma unroll
                    for (int e = 0; e < 4; ++e) v[e] = acc[i][j][4 * g4 + e];
                    *(f32x4*)(stg + fr * NT_STG_LD + j * 32 + 8 * g4 + 4 * fh) = v;
                }
            // the same wave reads back what it wrote (wave-private region): no workgroup barrier needed
#pragma unroll
            for (int t = 0; t < 12; ++t) {
                const int idx = t * 64 + lane;
                const int row = idx / 24, c4 = idx % 24;
                const int m = m_base + i * 32 + row;
                const int n = n_base + c4 * 4;
                if (m >= g.M || n >= g.N) continue;
                f32x4 v = *(const f32x4*)(stg + row * NT_STG_LD + c4 * 4);
                if (g.bias) v += *(const f32x4*)(bias_s + c4 * 4);
                if (dc.thr) {
                    const uint32_t e = (uint32_t)m * (uint32_t)g.N + (uint32_t)n;
                    float k0, k1, k2, k3;
                    drop_pair(dc, e, k0, k1);
                    drop_pair(dc, e + 2, k2, k3);
                    v[0] *= k0; v[1] *= k1; v[2] *= k2; v[3] *= k3;
                }
                v += rv[t];
                const size_t o = (size_t)m * g.ldc + n;
                st_stream((f32x4*)((float*)g.C + o), v);
                if (EPI == QST_EPI_F32_RESID_BF16) {
                    u32x2 pk; pk[0] = pack_op2(v[0], v[1]); pk[1] = pack_op2(v[2], v[3]);
                    st_stream((u32x2*)((op16*)g.C2 + o), pk);
                }
            }
        } else {
            // bf16 outputs: 8 columns (16 B) per lane, 12 lanes per row, 6 items per lane -- 16-byte stores issue
            // at twice the byte rate of 8-byte ones, and the epilogue is VMEM-issue/HBM bound
            u32x4 av[6];
#pragma unroll
            for (int t = 0; t < 6; ++t) {
                const int idx = t * 64 + lane;
                const int row = idx / 12, c8 = idx % 12;
                const int m = m_base + i * 32 + row;
                const int n = n_base + c8 * 8;
                const u32x4 z = {0u, 0u, 0u, 0u};
                if (EPI == QST_EPI_GELU_BWD)
                    av[t] = (m < g.M && n < g.N) ? ld_stream((const u32x4*)((const op16*)g.aux + (size_t)m * g.ldc + n)) : z;
            }
#pragma unroll
            for (int j = 0; j < 3; ++j)
#pragma unroll
                for (int g4 = 0; g4 < 4; ++g4) {
                    f32x4 v;
#pragma unroll
                    for (int e = 0; e < 4; ++e) v[e] = acc[i][j][4 * g4 + e];
                    *(f32x4*)(stg + fr * NT_STG_LD + j * 32 + 8 * g4 + 4 * fh) = v;
                }
#pragma unroll
            for (int t = 0; t < 6; ++t) {
                const int idx = t * 64 + lane;
                const int row = idx / 12, c8 = idx % 12;
                const int m = m_base + i * 32 + row;
                const int n = n_base + c8 * 8;
                if (m >= g.M || n >= g.N) continue;
                float v[8];
                {
                    const f32x4 lo = *(const f32x4*)(stg + row * NT_STG_LD + c8 * 8);
                    const f32x4 hi = *(const f32x4*)(stg + row * NT_STG_LD + c8 * 8 + 4);
#pragma unroll
                    for (int e = 0; e < 4; ++e) { v[e] = lo[e]; v[4 + e] = hi[e]; }
                }
                if (g.bias) {
                    const f32x4 lo = *(const f32x4*)(bias_s + c8 * 8), hi = *(const f32x4*)(bias_s + c8 * 8 + 4);
#pragma unroll
                    for (int e = 0; e < 4; ++e) { v[e] += lo[e]; v[4 + e] += hi[e]; }
                }
                const size_t o = (size_t)m * g.ldc + n;
                // N % 8 != 0 tails fall back to two 8-byte halves (N % 4 == 0 is required)
                const bool full = n + 8 <= g.N;
                u32x4 pk;
                if (EPI == QST_EPI_BF16) {
#pragma unroll
                    for (int e = 0; e < 4; ++e) pk[e] = pack_op2(v[2 * e], v[2 * e + 1]);
                } else if (EPI == QST_EPI_GELU) {
                    // h = gelu(u) feeds FFN2; gelu'(u) (not u) is what backward needs: both share one exp and one
                    // rcp, so the dgrad epilogue is a single multiply instead of a second erf evaluation
                    u32x4 pg;
#pragma unroll
                    for (int e = 0; e < 4; ++e) {
                        qst_f32x2 x2, cdf, pdf;
                        x2[0] = v[2 * e]; x2[1] = v[2 * e + 1];
                        gelu_parts2(x2, cdf, pdf);
                        const qst_f32x2 gg = x2 * pdf + cdf, hh = x2 * cdf;
                        pg[e] = pack_op2(gg[0], gg[1]);
                        pk[e] = pack_op2(hh[0], hh[1]);
                    }
                    if (full) st_stream((u32x4*)((op16*)g.C + o), pg);                     // gelu'(u), saved for backward
                    else { u32x2 h2; h2[0] = pg[0]; h2[1] = pg[1]; st_stream((u32x2*)((op16*)g.C + o), h2); }
                } else {   // QST_EPI_GELU_BWD: acc * gelu'(u)
#pragma unroll
                    for (int e = 0; e < 4; ++e)
                        pk[e] = pack_op2(v[2 * e] * op_lo(av[t][e]), v[2 * e + 1] * op_hi(av[t][e]));
                }
                op16* dst = (EPI == QST_EPI_GELU) ? (op16*)g.C2 : (op16*)g.C;
                if (full) st_stream((u32x4*)(dst + o), pk);
                else { u32x2 h2; h2[0] = pk[0]; h2[1] = pk[1]; st_stream((u32x2*)(dst + o), h2); }
            }
        }
    }
}

template <int EPI, int WAVES_M, int WAVES_N = 2, int TI = 2>
__global__ __launch_bounds__(64 * WAVES_M * WAVES_N, WAVES_N == 2 ? 2 : 1) void gemm_nt_kernel(QstGemmArgs g) {
    op_saturate(g.sat16 != 0);                        // f16 build, forward launches: 16-bit outputs clamp at +-65,504
    constexpr int NBM = 32 * TI * WAVES_M, NBN = 96 * WAVES_N, NW = WAVES_M * WAVES_N;
    extern __shared__ __attribute__((aligned(16))) char smem[];
    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int wm = wave / WAVES_N, wn = wave % WAVES_N;
    const int ntn = (g.N + NBN - 1) / NBN, ntm = (g.M + NBM - 1) / NBM;
    const int wg = xcd_remap(blockIdx.x, ntm * ntn);
    const int m0 = (wg / ntn) * NBM, n0 = (wg % ntn) * NBN;
    const int fr = lane & 31, fh = lane >> 5;
    // diagnostic (QstGemmArgs.splits bit 16; g.colsum, unused by NT, = uint64 [workgroup][4]): wall-clock stamps (s_memrealtime,
    // 100 MHz) of wave 0 at kernel entry, after the K loop, after the epilogue -- one scalar branch each when off
    const bool stamp_on = (g.splits & 16) && g.colsum != nullptr;
#define NT_STAMP(k_) do { if (stamp_on && tid == 0) ((unsigned long long*)g.colsum)[blockIdx.x * 4 + (k_)] = __builtin_amdgcn_s_memrealtime(); } while (0)
    NT_STAMP(0);
    f32x16 acc[TI][3];
    if constexpr (TI == 4) nt_mainloop_tall<WAVES_M, WAVES_N>(g, smem, m0, n0, acc);
    else nt_mainloop<WAVES_M, WAVES_N, NoHook, false, 2, EPI != QST_EPI_GELU_BWD>(g, smem, m0, n0, acc);
    NT_STAMP(1);

    float* stg = (float*)smem + wave * (32 * NT_STG_LD);
    float* bias_s = (float*)smem + NW * (32 * NT_STG_LD) + wave * 96;     // this wave's 96 bias values
    if (g.bias) {
        for (int c = lane; c < 96; c += 64) {
            const int n = n0 + wn * 96 + c;
            bias_s[c] = n < g.N ? g.bias[n] : 0.f;
        }
    }
    nt_epilogue<EPI, TI>(g, acc, stg, bias_s, m0 + wm * (32 * TI), n0 + wn * 96, lane);
    NT_STAMP(2);
#undef NT_STAMP
}

#if !QST_OP_F16      // (the fp8 kernels exist once, in the bf16 build of this file)
// ---------------------------------------------------------------- NT on the fp8 matrix cores (MXFP8, inference)
// C = A . B^T with BOTH operands in OCP MXFP8: e4m3 elements + one E8M0 scale (a power of two) per 32 consecutive K
// elements of a row, multiplied by v_mfma_scale_f32_32x32x64_f8f6f4 -- 64 K per instruction at twice the FLOP rate of the
// bf16 MFMA, the de-quantisation done by the instruction. BASELINE configs[4] ("CDNA4 fp8 MFMA GEMMs"); SURVEY.md 7 step 9.
//
// Operand layout of the instruction (not the obvious one; established with exact integer data by
// tools/probe/mfma_f8_probe.hip and mfma_f8_scalemap.hip): the 64-deep step is two 32-deep MX blocks; lane l = 32h + r
// holds, in registers 0-3, bytes k = 16h .. 16h+15 of block 0 of row r and, in registers 4-7, bytes k = 32+16h .. of
// block 1; the scale of block b of row r is the low byte of the scale register of lane 32b + r. So lane (h, r) reads two
// 16-byte chunks of its row (positions 16h and 32 + 16h of the 64-byte step) and supplies the scale of block h.
//
// Same tiling as gemm_nt_kernel<EPI, 2, 2>: 128 x 192 tile, 4 waves of 64 x 96, two workgroups per CU, 2-slot ring of
// 40 KB stages -- a stage is now 128 K deep (128-byte fp8 rows: the whole-line DMA shape), i.e. half as many stages,
// barriers and DMA bytes per FLOP as the bf16 loop. The scales do not go through LDS (two workgroups per CU leave no
// room): each lane loads the five dwords it needs for the NEXT stage (4 scale bytes per row and stage) straight into
// registers right after that stage's DMAs are issued.
typedef __attribute__((ext_vector_type(8))) int i32x8;

__device__ __forceinline__ i32x8 f8_frag(const char* img, int row, int ks, int h) {
    const u32x4 lo = *(const u32x4*)(img + nt_off(row, 4 * ks + h));
    const u32x4 hi = *(const u32x4*)(img + nt_off(row, 4 * ks + 2 + h));
    i32x8 f;
    f[0] = (int)lo[0]; f[1] = (int)lo[1]; f[2] = (int)lo[2]; f[3] = (int)lo[3];
    f[4] = (int)hi[0]; f[5] = (int)hi[1]; f[6] = (int)hi[2]; f[7] = (int)hi[3];
    return f;
}

constexpr int QST_EPI_GELU_MX_TRAIN_ = 6;    // = QST_EPI_GELU_MX_TRAIN: C / C2 as QST_EPI_GELU (gelu'(u), h as bf16) AND h as MXFP8 in C3 / C4
constexpr int QST_EPI_GELU_MX_ = 5;          // = QST_EPI_GELU_MX: C = e4m3 of gelu(acc + bias) [M, ldc bytes], C2 = E8M0 [M, ldc / 32]

template <int EPI>
__global__ __launch_bounds__(256, 2) void gemm_nt_f8_kernel(QstGemmArgs g) {
    constexpr int NBM = 128, NBN = 192, BKB = 128;                  // stage depth in K elements = bytes
    constexpr int A_BYTES = NBM * BKB, B_BYTES = NBN * BKB, STAGE = A_BYTES + B_BYTES;      // 16 + 24 KB
    extern __shared__ __attribute__((aligned(16))) char smem[];
    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int wm = wave >> 1, wn = wave & 1;
    const int ntn = (g.N + NBN - 1) / NBN, ntm = (g.M + NBM - 1) / NBM;
    const int wg = xcd_remap(blockIdx.x, ntm * ntn);
    const int m0 = (wg / ntn) * NBM, n0 = (wg % ntn) * NBN;
    const int fr = lane & 31, fh = lane >> 5;
    const int rows_a = min(NBM, g.M - m0), rows_b = min(NBN, g.N - n0);
    const __amdgpu_buffer_rsrc_t ra = make_rsrc((const uint8_t*)g.A + (size_t)m0 * g.lda, (uint32_t)rows_a * g.lda);
    const __amdgpu_buffer_rsrc_t rb = make_rsrc((const uint8_t*)g.B + (size_t)n0 * g.ldb, (uint32_t)rows_b * g.ldb);
    // scales are stored stage-major, [K/128][rows][4]: the dwords of 32 consecutive rows for one stage are one 128-byte
    // line (row-major [rows][K/32] made every scale load touch 32 lines: four times the line requests of the stage's DMAs)
    const uint32_t* sA = (const uint32_t*)g.aux;
    const uint32_t* sB = (const uint32_t*)g.bscale;
    uint32_t va[4], vb[6];
#pragma unroll
    for (int t = 0; t < 4; ++t) {                                   // 8 rows of 128 B per DMA instruction
        const int row = (wave * 4 + t) * 8 + (lane >> 3);
        va[t] = (uint32_t)row * g.lda + (uint32_t)(((lane & 7) ^ ((row >> 1) & 7)) * 16);
    }
#pragma unroll
    for (int t = 0; t < 6; ++t) {
        const int row = (wave * 6 + t) * 8 + (lane >> 3);
        vb[t] = (uint32_t)row * g.ldb + (uint32_t)(((lane & 7) ^ ((row >> 1) & 7)) * 16);
    }
    auto issue = [&](int kt) {
        char* st = smem + (kt & 1) * STAGE;
#pragma unroll
        for (int t = 0; t < 4; ++t) dma16(ra, st + (wave * 4 + t) * 1024, va[t], (uint32_t)kt * BKB);
#pragma unroll
        for (int t = 0; t < 6; ++t) dma16(rb, st + A_BYTES + (wave * 6 + t) * 1024, vb[t], (uint32_t)kt * BKB);
    };
    // the four scale bytes of this lane's rows for stage kt (rows past the matrix read as 0 = 2^-127 next to zero data)
    uint32_t sa[2], sb[3], sa_n[2], sb_n[3];
    auto load_scales = [&](int kt, uint32_t (&xa)[2], uint32_t (&xb)[3]) {
        if (g.splits & 1) { xa[0] = xa[1] = xb[0] = xb[1] = xb[2] = 0x7f7f7f7fu; return; }      // timing experiment: unit scales
#pragma unroll
        for (int i = 0; i < 2; ++i) {
            const int m = m0 + wm * 64 + i * 32 + fr;
            xa[i] = m < g.M ? sA[(size_t)kt * g.M + m] : 0u;
        }
#pragma unroll
        for (int j = 0; j < 3; ++j) {
            const int n = n0 + wn * 96 + j * 32 + fr;
            xb[j] = n < g.N ? sB[(size_t)kt * g.N + n] : 0u;
        }
    };
    f32x16 acc[2][3];
#pragma unroll
    for (int i = 0; i < 2; ++i)
#pragma unroll
        for (int j = 0; j < 3; ++j)
#pragma unroll
            for (int r = 0; r < 16; ++r) acc[i][j][r] = 0.f;
    const int nk = g.K / BKB;
    issue(0);
    load_scales(0, sa_n, sb_n);
    for (int kt = 0; kt < nk; ++kt) {
        wait_vmcnt<0>();
        __builtin_amdgcn_s_barrier();
#pragma unroll
        for (int i = 0; i < 2; ++i) sa[i] = sa_n[i];
#pragma unroll
        for (int j = 0; j < 3; ++j) sb[j] = sb_n[j];
        if (kt + 1 < nk) { issue(kt + 1); load_scales(kt + 1, sa_n, sb_n); }
        const char* pa = smem + (kt & 1) * STAGE;
        const char* pb = pa + A_BYTES;
#pragma unroll
        for (int ks = 0; ks < 2; ++ks) {
            i32x8 fa[2], fb[3];
#pragma unroll
            for (int i = 0; i < 2; ++i) fa[i] = f8_frag(pa, wm * 64 + i * 32 + fr, ks, fh);
#pragma unroll
            for (int j = 0; j < 3; ++j) fb[j] = f8_frag(pb, wn * 96 + j * 32 + fr, ks, fh);
            const int sh = 16 * ks + 8 * fh;                        // this lane's MX block of the step: 2 ks + h
#pragma unroll
            for (int i = 0; i < 2; ++i)
#pragma unroll
                for (int j = 0; j < 3; ++j)
                    acc[i][j] = __builtin_amdgcn_mfma_scale_f32_32x32x64_f8f6f4(fb[j], fa[i], acc[i][j], 0, 0, 0,
                                                                                 (int)(sb[j] >> sh), 0, (int)(sa[i] >> sh));
        }
    }
    __builtin_amdgcn_s_barrier();

    float* stg = (float*)smem + wave * (32 * NT_STG_LD);
    float* bias_s = (float*)smem + 4 * (32 * NT_STG_LD) + wave * 96;
    for (int c = lane; c < 96; c += 64) {
        const int n = n0 + wn * 96 + c;
        bias_s[c] = (g.bias && n < g.N) ? g.bias[n] : 0.f;
    }
    if constexpr (EPI != QST_EPI_GELU_MX_ && EPI != QST_EPI_GELU_MX_TRAIN_) {
        nt_epilogue<EPI, 2>(g, acc, stg, bias_s, m0 + wm * 64, n0 + wn * 96, lane);
    } else {
        constexpr bool TRAIN = EPI == QST_EPI_GELU_MX_TRAIN_;
        // h = gelu(acc + bias) leaves as MXFP8 for the second feed-forward product: 8 columns per lane, 4 lanes = one
        // 32-column MX block (aligned quads: 12 lanes per row), block amax by two DPP steps inside the quad
        // (training: the same launch also leaves gelu'(u) and h as bf16 for the backward, and the MX copy is taken from the
        //  bf16-rounded h -- what quantising that copy in a pass of its own would give, without the 1.8 GB pass per layer)
        uint8_t* Cq = (uint8_t*)(TRAIN ? g.C3 : g.C);
        uint8_t* Cs = (uint8_t*)(TRAIN ? g.C4 : g.C2);
        const int m_base = m0 + wm * 64, n_base = n0 + wn * 96;
#pragma unroll
        for (int i = 0; i < 2; ++i) {
#pragma unroll
            for (int j = 0; j < 3; ++j)
#pragma unroll
                for (int g4 = 0; g4 < 4; ++g4) {
                    f32x4 v;
#pragma unroll
                    for (int e = 0; e < 4; ++e) v[e] = acc[i][j][4 * g4 + e];
                    *(f32x4*)(stg + fr * NT_STG_LD + j * 32 + 8 * g4 + 4 * fh) = v;
                }
#pragma unroll
            for (int t = 0; t < 6; ++t) {
                const int idx = t * 64 + lane;
                const int row = idx / 12, c8 = idx % 12;
                const int m = m_base + i * 32 + row, n = n_base + c8 * 8;
                const f32x4 lo = *(const f32x4*)(stg + row * NT_STG_LD + c8 * 8), hi = *(const f32x4*)(stg + row * NT_STG_LD + c8 * 8 + 4);
                const f32x4 blo = *(const f32x4*)(bias_s + c8 * 8), bhi = *(const f32x4*)(bias_s + c8 * 8 + 4);
                float hv[8];
                float amax = 0.f;
                if constexpr (TRAIN) {
                    float gp[8];
#pragma unroll
                    for (int e = 0; e < 4; ++e) {
                        float cdf, pdf;
                        const float x0 = lo[e] + blo[e], x1 = hi[e] + bhi[e];
                        gelu_parts(x0, cdf, pdf); hv[e] = x0 * cdf; gp[e] = cdf + x0 * pdf;
                        gelu_parts(x1, cdf, pdf); hv[4 + e] = x1 * cdf; gp[4 + e] = cdf + x1 * pdf;
                    }
                    u32x4 pg, ph;
#pragma unroll
                    for (int e = 0; e < 4; ++e) {
                        pg[e] = pack_op2(gp[2 * e], gp[2 * e + 1]);
                        ph[e] = pack_op2(hv[2 * e], hv[2 * e + 1]);
                        hv[2 * e] = op_lo(ph[e]); hv[2 * e + 1] = op_hi(ph[e]);          // quantise the bf16-rounded h
                    }
                    if (m < g.M && n < g.N) {
                        st_stream((u32x4*)((op16*)g.C + (size_t)m * g.ldc + n), pg);
                        st_stream((u32x4*)((op16*)g.C2 + (size_t)m * g.ldc + n), ph);
                    }
#pragma unroll
                    for (int e = 0; e < 8; ++e) amax = fmaxf(amax, fabsf(hv[e]));
                } else {
#pragma unroll
                    for (int e = 0; e < 4; ++e) {
                        hv[e] = gelu_erf(lo[e] + blo[e]);
                        hv[4 + e] = gelu_erf(hi[e] + bhi[e]);
                        amax = fmaxf(amax, fmaxf(fabsf(hv[e]), fabsf(hv[4 + e])));
                    }
                }
                amax = fmaxf(amax, dpp_mov<0xB1>(amax));             // quad_perm [1,0,3,2]
                amax = fmaxf(amax, dpp_mov<0x4E>(amax));             // quad_perm [2,3,0,1]
                const int ex = mx_exponent(amax);
                const float inv = pow2f(-ex);
                uint32_t p0 = 0, p1 = 0;
                p0 = __builtin_amdgcn_cvt_pk_fp8_f32(hv[0] * inv, hv[1] * inv, p0, false);
                p0 = __builtin_amdgcn_cvt_pk_fp8_f32(hv[2] * inv, hv[3] * inv, p0, true);
                p1 = __builtin_amdgcn_cvt_pk_fp8_f32(hv[4] * inv, hv[5] * inv, p1, false);
                p1 = __builtin_amdgcn_cvt_pk_fp8_f32(hv[6] * inv, hv[7] * inv, p1, true);
                if (m < g.M && n < g.N) {
                    u32x2 pk; pk[0] = p0; pk[1] = p1;
                    st_stream((u32x2*)(Cq + (size_t)m * g.ldc + n), pk);
                    if ((c8 & 3) == 0) Cs[((size_t)(n >> 7) * g.M + m) * 4 + ((n >> 5) & 3)] = (uint8_t)(ex + 127);   // stage-major
                }
            }
        }
    }
}

// MXFP8 quantisation of a [rows, K] matrix (fp32: weights at shadow refresh; bf16: activations between kernels):
// 8 elements per thread, 4 threads = one 32-element block.
template <typename SRC>
__global__ __launch_bounds__(256) void quant_mx_kernel(const SRC* src, int64_t n8, int64_t rows, int K, uint8_t* q, uint8_t* sc) {
    const int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x;     // 8-element group index; n8 % 4 == 0 (K % 32 == 0)
    float v[8];
    if (i < n8) {
        if constexpr (sizeof(SRC) == 4) {
            const f32x4 a = *(const f32x4*)((const float*)src + i * 8), b = *(const f32x4*)((const float*)src + i * 8 + 4);
#pragma unroll
            for (int e = 0; e < 4; ++e) { v[e] = a[e]; v[4 + e] = b[e]; }
        } else {
            const u32x4 a = *(const u32x4*)((const op16*)src + i * 8);
#pragma unroll
            for (int e = 0; e < 4; ++e) { v[2 * e] = op_lo(a[e]); v[2 * e + 1] = op_hi(a[e]); }
        }
    } else {
#pragma unroll
        for (int e = 0; e < 8; ++e) v[e] = 0.f;
    }
    float amax = 0.f;
#pragma unroll
    for (int e = 0; e < 8; ++e) amax = fmaxf(amax, fabsf(v[e]));
    amax = fmaxf(amax, dpp_mov<0xB1>(amax));
    amax = fmaxf(amax, dpp_mov<0x4E>(amax));
    const int ex = mx_exponent(amax);
    const float inv = pow2f(-ex);
    uint32_t p0 = 0, p1 = 0;
    p0 = __builtin_amdgcn_cvt_pk_fp8_f32(v[0] * inv, v[1] * inv, p0, false);
    p0 = __builtin_amdgcn_cvt_pk_fp8_f32(v[2] * inv, v[3] * inv, p0, true);
    p1 = __builtin_amdgcn_cvt_pk_fp8_f32(v[4] * inv, v[5] * inv, p1, false);
    p1 = __builtin_amdgcn_cvt_pk_fp8_f32(v[6] * inv, v[7] * inv, p1, true);
    if (i < n8) {
        u32x2 pk; pk[0] = p0; pk[1] = p1;
        *(u32x2*)(q + i * 8) = pk;
        if ((i & 3) == 0) {                                   // scale of block kb of row r -> stage-major [K/128][rows][4]
            const int64_t blk = i >> 2, r = blk / (K >> 5);
            const int kb = (int)(blk - r * (K >> 5));
            sc[((int64_t)(kb >> 2) * rows + r) * 4 + (kb & 3)] = (uint8_t)(ex + 127);
        }
    }
}

#endif  // !QST_OP_F16

// ---------------------------------------------------------------- NT with a LayerNorm fused into the epilogue
// For N = 384 (MiniLM's hidden size) one 128 x 384 tile spans whole rows, so the LayerNorm that always follows the
// attention-output / FFN-2 projection (forward) and the LayerNorm backward that always follows the FFN-1 / QKV dgrad
// (backward) run on the tile while it is still on the CU: the fp32 pre-norm tensor never goes to HBM (it used to be
// written by the GEMM and read back by a row kernel, 100 MB per LayerNorm at M = 32768, with both kernels HBM-bound).
// 8 waves as 2 (M) x 4 (N), the same 64 x 96 wave tile and K loop as gemm_nt_kernel. Epilogue per 32-row pass: the
// four waves of a row panel stage their sub-tiles into one [32][384] fp32 slab, then each takes 8 complete rows
// (6 columns per lane, as the row kernels in rowops.hip do).
constexpr int LN_N = 384, LN_LD = 388;               // slab row stride 1552 B: ds_write_b128 conflict-free
#ifndef QST_LN_ASLOTS
#define QST_LN_ASLOTS 3
#endif
constexpr int LN_RING = (QST_LN_ASLOTS * 128 + 2 * LN_N) * NBK * 2;  // the K-loop ring (3 A slots + 2 B slots = 144 KB); the epilogue slabs reuse 99 KB of it
constexpr int LN_LDS = LN_RING + 3 * LN_N * 4;       // + bias / gamma / beta, loaded before the K loop

template <int MODE, int DROPW = 0>        // DROPW: QstGemmArgs.drop_where as a compile-time constant (0 = no dropout)
__global__ __launch_bounds__(512, 1) void gemm_nt_ln_kernel(QstGemmArgs g, QstLnEpi e) {
    op_saturate(MODE == 0);                           // f16 build: the forward epilogue saturates, the backward one keeps inf
    constexpr bool DROP = DROPW != 0;
    extern __shared__ __attribute__((aligned(16))) char smem[];
    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int wm = wave >> 2, wn = wave & 3;
    const int ntm = (g.M + 127) / 128;
    const int m0 = xcd_remap(blockIdx.x, ntm) * 128;
    const int fr = lane & 31, fh = lane >> 5;
    // diagnostic (QstGemmArgs.splits bit 3, mode 0 only): s_memtime at the phase boundaries of wave 0 -> e.partials as
    // uint64 [workgroup][8]; costs one scalar branch per stamp when off
    const bool stamp_on = MODE == 0 && (g.splits & 8) && e.partials != nullptr;
#define LN_STAMP(k_) do { if (stamp_on && tid == 0) ((unsigned long long*)e.partials)[blockIdx.x * 8 + (k_)] = __builtin_readcyclecounter(); } while (0)
    LN_STAMP(0);
    // Everything the epilogue reads from global memory is requested BEFORE the K loop (the three vectors into LDS beyond
    // the ring, pass 0's residual / xhat rows into registers) or, for pass 1, while pass 0 is being normalised: issued
    // where they were first used, each of these loads exposed its full miss latency (~2 us apiece, three of them in a row)
    // with every wave of the CU waiting -- a quarter of the K = 384 launches.
    float* slab = (float*)smem + wm * (32 * LN_LD);
    float* vec_s = (float*)(smem + LN_RING);                 // [3][384]: bias, gamma, beta (beyond the ring)
    f32x2 rv[2][8][3];
    uint32_t xv[2][8][3];
    float rs[2][8];
#define LN_PREFETCH(i_) do {                                                                                             \
        _Pragma("unroll") for (int k = 0; k < 8; ++k) {                                                                  \
            const int m = m0 + wm * 64 + (i_) * 32 + wn * 8 + k;                                                         \
            const bool ok = m < g.M;                                                                                     \
            _Pragma("unroll") for (int t = 0; t < 3; ++t) {                                                              \
                const int c = 2 * (lane + 64 * t);                                                                       \
                rv[i_][k][t][0] = rv[i_][k][t][1] = 0.f;                                                                 \
                xv[i_][k][t] = 0u;                                                                                       \
                if (ok && g.resid) rv[i_][k][t] = ld_stream((const f32x2*)(g.resid + (size_t)m * g.ldr + c)); \
                if (MODE == 1 && ok) xv[i_][k][t] = ld_stream((const uint32_t*)((const op16*)e.xhat + (size_t)m * LN_N + c)); \
            }                                                                                                            \
            rs[i_][k] = (MODE == 1 && ok) ? e.rstd[m] : 0.f;                                                             \
        }                                                                                                                \
    } while (0)
    f32x16 acc[2][3];
    auto early_loads = [&]() __attribute__((always_inline)) {
        // (stamps showed 7,300 cycles between kernel entry and the first DMA when these loads came first)
        float v3[3];
#pragma unroll
        for (int q = 0; q < 3; ++q) {
            const int c = tid + 512 * q;                     // 3 * 384 = 1152 <= 3 * 512
            const int which = c / LN_N, n = c - which * LN_N;
            v3[q] = 0.f;
            if (c < 3 * LN_N) {
                if (which == 0) v3[q] = g.bias ? g.bias[n] : 0.f;
                else if (which == 1) v3[q] = e.gamma[n];
                else if (MODE == 0) v3[q] = e.beta[n];
            }
        }
        LN_PREFETCH(0);
#pragma unroll
        for (int q = 0; q < 3; ++q)
            if (tid + 512 * q < 3 * LN_N) vec_s[tid + 512 * q] = v3[q];
    };
    nt_mainloop<2, 4, decltype(early_loads), true, QST_LN_ASLOTS, MODE == 0>(g, smem, m0, 0, acc, early_loads);   // the tile spans whole rows: A is read once
    LN_STAMP(2);

    f32x2 ag[3], ab[3];
#pragma unroll
    for (int t = 0; t < 3; ++t) { ag[t][0] = ag[t][1] = ab[t][0] = ab[t][1] = 0.f; }
    const float inv_n = 1.f / (float)LN_N;
    // dropout (QstGemmArgs.drop_where): mode 0: the projection output before the residual (1); mode 1: the bf16 copy of the
    // result (2) or the incoming gradient (3). Masks are recomputed from (state, site, m * 384 + n), never stored.
    // (a separate instantiation: the backward epilogue has no registers to spare -- 252 of 256 without it)
    const DropCtx dc = DROP ? drop_ctx(g.drop) : DropCtx{0u, 0u, 1.f};
    constexpr int dwhere = DROPW;

#pragma unroll
    for (int i = 0; i < 2; ++i) {
        if (i > 0) __builtin_amdgcn_s_barrier();             // everyone has finished reading pass 0's slab
#pragma unroll
        for (int j = 0; j < 3; ++j)
#pragma unroll
            for (int g4 = 0; g4 < 4; ++g4) {
                f32x4 v;
#pragma unroll
                for (int q = 0; q < 4; ++q) v[q] = acc[i][j][4 * g4 + q];
                *(f32x4*)(slab + fr * LN_LD + wn * 96 + j * 32 + 8 * g4 + 4 * fh) = v;
            }
        __syncthreads();                                     // slab complete
        LN_STAMP(3 + 2 * i);
        if (i == 0) LN_PREFETCH(1);                          // pass 1's rows travel while pass 0 is normalised and stored
#pragma unroll
        for (int k = 0; k < 8; ++k) {
            const int row = wn * 8 + k;
            const int m = m0 + wm * 64 + i * 32 + row;
            f32x2 v[3];
#pragma unroll
            for (int t = 0; t < 3; ++t) v[t] = *(const f32x2*)(slab + row * LN_LD + 2 * (lane + 64 * t));
            if (MODE == 0) {
                float s = 0.f;
#pragma unroll
                for (int t = 0; t < 3; ++t) {
                    const f32x2 b = *(const f32x2*)(vec_s + 2 * (lane + 64 * t));
                    v[t][0] += b[0]; v[t][1] += b[1];
                    if (DROP && dwhere == 1) {
                        float k0, k1;
                        drop_pair(dc, (uint32_t)m * LN_N + 2 * (lane + 64 * t), k0, k1);
                        v[t][0] *= k0; v[t][1] *= k1;
                    }
                    v[t][0] += rv[i][k][t][0];                          // same association as the unfused epilogue
                    v[t][1] += rv[i][k][t][1];
                    s += v[t][0] + v[t][1];
                }
                const float mean = wave_sum(s) * inv_n;
                float q = 0.f;
#pragma unroll
                for (int t = 0; t < 3; ++t) {
                    const float a0 = v[t][0] - mean, a1 = v[t][1] - mean;
                    q += a0 * a0 + a1 * a1;
                }
                const float rstd = rsqrtf(wave_sum(q) * inv_n + e.eps);      // biased variance, as nn.LayerNorm
                if (m < g.M) {
                    if (lane == 0 && e.rstd) e.rstd[m] = rstd;
#pragma unroll
                    for (int t = 0; t < 3; ++t) {
                        const int c = 2 * (lane + 64 * t);
                        const f32x2 ga = *(const f32x2*)(vec_s + LN_N + c), be = *(const f32x2*)(vec_s + 2 * LN_N + c);
                        const float h0 = (v[t][0] - mean) * rstd, h1 = (v[t][1] - mean) * rstd;
                        f32x2 o;
                        o[0] = h0 * ga[0] + be[0];
                        o[1] = h1 * ga[1] + be[1];
                        if (!(g.splits & 4)) st_stream((f32x2*)((float*)g.C + (size_t)m * g.ldc + c), o);
                        if (g.C2 && !(g.splits & 1)) st_stream((uint32_t*)((op16*)g.C2 + (size_t)m * g.ldc + c), pack_op2(o[0], o[1]));
                        if (e.xhat && !(g.splits & 2)) st_stream((uint32_t*)((op16*)e.xhat + (size_t)m * LN_N + c), pack_op2(h0, h1));
                    }
                }
            } else {
                float s1 = 0.f, s2 = 0.f;
                f32x2 x[3];
#pragma unroll
                for (int t = 0; t < 3; ++t) {
                    const f32x2 ga = *(const f32x2*)(vec_s + LN_N + 2 * (lane + 64 * t));
                    x[t][0] = op_lo(xv[i][k][t]); x[t][1] = op_hi(xv[i][k][t]);
                    v[t][0] += rv[i][k][t][0];                        // dy = dgrad + residual-path gradient
                    v[t][1] += rv[i][k][t][1];
                    if (DROP && dwhere == 3) {
                        float k0, k1;
                        drop_pair(dc, (uint32_t)m * LN_N + 2 * (lane + 64 * t), k0, k1);
                        v[t][0] *= k0; v[t][1] *= k1;
                    }
                    ag[t][0] += v[t][0] * x[t][0]; ag[t][1] += v[t][1] * x[t][1];
                    ab[t][0] += v[t][0];           ab[t][1] += v[t][1];
                    v[t][0] *= ga[0]; v[t][1] *= ga[1];            // dxhat
                    s1 += v[t][0] + v[t][1];
                    s2 += v[t][0] * x[t][0] + v[t][1] * x[t][1];
                }
                const float m1 = wave_sum(s1) * inv_n, m2 = wave_sum(s2) * inv_n;
                if (m < g.M) {
#pragma unroll
                    for (int t = 0; t < 3; ++t) {
                        const int c = 2 * (lane + 64 * t);
                        f32x2 o;
                        o[0] = rs[i][k] * (v[t][0] - m1 - x[t][0] * m2);
                        o[1] = rs[i][k] * (v[t][1] - m1 - x[t][1] * m2);
                        st_stream((f32x2*)((float*)g.C + (size_t)m * g.ldc + c), o);
                        if (g.C2) {
                            if (DROP && dwhere == 2) {
                                float k0, k1;
                                drop_pair(dc, (uint32_t)m * LN_N + c, k0, k1);
                                o[0] *= k0; o[1] *= k1;
                            }
                            st_stream((uint32_t*)((op16*)g.C2 + (size_t)m * g.ldc + c), pack_op2(o[0], o[1]));
                        }
                    }
                }
            }
        }
        LN_STAMP(4 + 2 * i);
    }
    LN_STAMP(7);
    if (MODE == 1 && e.partials) {
        // dgamma / dbeta of this tile's 128 rows: 8 waves x 6 columns per lane -> LDS -> one [2][384] row per tile,
        // reduced over tiles by ln_bwd_reduce_batch_kernel in a fixed order
        __syncthreads();
        float* red = (float*)smem;                               // [8 waves][2][384]
#pragma unroll
        for (int t = 0; t < 3; ++t) {
            *(f32x2*)(red + (wave * 2 + 0) * LN_N + 2 * (lane + 64 * t)) = ag[t];
            *(f32x2*)(red + (wave * 2 + 1) * LN_N + 2 * (lane + 64 * t)) = ab[t];
        }
        __syncthreads();
        for (int c = tid; c < 2 * LN_N; c += 512) {
            const int which = c / LN_N, n = c - which * LN_N;
            float a = 0.f;
#pragma unroll
            for (int w = 0; w < 8; ++w) a += red[(w * 2 + which) * LN_N + n];
            e.partials[(size_t)blockIdx.x * 2 * LN_N + c] = a;
        }
    }
}

// ---------------------------------------------------------------- TN (grouped wgrad)
// One launch computes ALL weight gradients of a layer (dW_qkv, dW_o, dW_1, dW_2 and their bias gradients):
//   C_p[N_p, K_p] += A_p[M, N_p]^T . B_p[M, K_p]        (A = dY, B = layer input X; both row-major, bf16)
// 192 x 192 output tile per 256-thread workgroup (4 waves as 2 x 2, 96 x 96 per wave = 3 x 3 MFMA tiles: 12
// transposed fragment reads feed 9 MFMAs per k-step, and the tile's L2->LDS ingest per MFMA-cycle is 1/96 against
// 1/64 for 128 x 128). 192 divides every weight dimension of the three model families, so no tile is padded.
// The reduction over M is split into 8 ranges, ONE PER XCD: every tile of every problem for range s runs on XCD s,
// so each XCD streams its M/8 rows of dY and X from HBM once and serves all re-reads (each dY row by K/192 tiles,
// each X row by N/192 tiles) from its own L2. Partial tiles are combined with fp32 atomics in 128-byte segments
// (8 adders per element). Operands are consumed straight from row-major [M, *] images with ds_read_b64_tr_b16.
// LDS image per operand stage: [32 m-rows][192 bf16] = 384-byte rows (24 chunks of 16 B); chunk c of row r sits at
// chunk position c ^ (((r >> 1) & 1) << 2): the 4 rows x 64 B that one half-wave reads land in 16 distinct 16-B slots.
constexpr int TT = 192, TBK = 64, TSTAGES = 3;            // 3-slot ring of 64-row stages: one being read, two in flight
constexpr int TT_TILE = TBK * TT * 2;                // 24 KB per operand per stage
constexpr int TT_STAGE = 2 * TT_TILE;                // 48 KB
constexpr int TT_LDS = TSTAGES * TT_STAGE;           // 144 KB: one workgroup (4 MFMA + 4 loader waves) per CU
__device__ __forceinline__ int tn_swz(int row) { return ((row >> 1) & 1) << 2; }
__device__ __forceinline__ uint32_t tn_off(int row, int chunk) {
    return (uint32_t)(row * 384 + ((chunk ^ tn_swz(row)) << 4));
}
__device__ __forceinline__ op16x4 lds_tr16(const char* p) {
    return lds_tr16_op(p);
}

// Wave roles: waves 0-3 run the MFMAs (one per SIMD, the whole register file to themselves), waves 4-7 are loaders
// that issue every LDS-DMA (waves 4,5: the two halves of the dY stage, waves 6,7: of the X stage). In-kernel stamps on the earlier
// all-waves-load version showed 2800 cycles per 32-row stage for 672 cycles of MFMA: an in-order wave pays the DMA
// issue cost (60-185 cycles per instruction) and the LDS read latency in series with its MFMAs; with loaders it is 990.
//
// Work decomposition: XCD x owns M-range x and runs W = 32 workgroups on it (one per CU: 144 KB of LDS). With T tiles
// = a*W + b, every workgroup reduces `a` whole tiles and then one stage-piece of a leftover tile (task list in the
// kernel), so all CUs finish together instead of 1.5 tiles per CU being rounded up to 2; split tiles simply receive
// several partial sums through the fp32 atomics.
__global__ __launch_bounds__(512, 1) void gemm_tn_group_kernel(QstTnGroup grp) {
    extern __shared__ __attribute__((aligned(16))) char smem[];
    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int wm = (wave >> 1) & 1, wn = wave & 1;
    const int xcd = blockIdx.x & 7, jloc = blockIdx.x >> 3;        // blocks b and b+8 share an XCD (speed only)
    const int wg_per_range = (int)(gridDim.x >> 3) / grp.ranges_per_xcd;
    const int range = xcd + 8 * (jloc / wg_per_range);             // which M-range
    const int jr = jloc % wg_per_range;                            // workgroup index inside the range
    const int M = grp.prob[0].M;
    const int per = (((M + grp.splits - 1) / grp.splits) + TBK - 1) / TBK * TBK;
    const int mbeg = range * per, mend = min(M, mbeg + per);
    if (mbeg >= mend) return;
    const int S = (mend - mbeg + TBK - 1) / TBK;                   // stages per tile in this range
    // Task list of this workgroup (W = workgroups per range, T = tiles = a*W + b):
    //   a full tiles (tile t*W + jr, all S stages), processed in lockstep with the other workgroups of the range, so
    //   the rows being streamed are shared through the XCD's L2 (a tile-major "stream-K" split lost that: every
    //   workgroup sat at a different stage and HBM traffic tripled);
    //   then one piece of a leftover tile: the b leftover tiles are cut into floor(W/b) stage pieces each.
    const int W = wg_per_range, T = grp.total_tiles;
    const int a_full = T / W, b_left = T % W;
    const int pieces = b_left > 0 ? max(1, W / b_left) : 1;
    const int ntasks = a_full + ((b_left > 0 && jr / pieces < b_left) ? 1 : 0);

    // transposed-read lane geometry (cdna guide T10): lane i = 4q+p of a 16-lane group supplies row q, cols 4p..4p+3
    const int li = lane & 15, q = li >> 2, p = li & 3, gsel = (lane >> 4) & 1, fh = lane >> 5;
    const int fr = lane & 31;

#pragma unroll 1
    for (int task = 0; task < ntasks; ++task) {
        int tile, s0, s1;
        if (task < a_full) { tile = task * W + jr; s0 = 0; s1 = S; }
        else {
            tile = a_full * W + jr / pieces;
            const int pc = jr % pieces;
            const int per_piece = (S + pieces - 1) / pieces;
            s0 = pc * per_piece; s1 = min(S, s0 + per_piece);
        }
        const int nm = s1 - s0;
        if (nm <= 0) continue;                                     // uniform over the workgroup
        int pi = 0;
#pragma unroll 1
        while (pi + 1 < grp.nprob && tile >= grp.tiles[pi]) { tile -= grp.tiles[pi]; ++pi; }
        const QstGemmArgs& g = grp.prob[pi];
        const int ntk = (g.K + TT - 1) / TT;
        const int n0 = (tile / ntk) * TT, k0 = (tile % ntk) * TT;
        const int row0 = mbeg + s0 * TBK;                          // first reduction row of this piece

        if (wave >= 4) {
            // -------------------------------------------------------- loader wave: half an operand, 6 DMA per stage
            // (four loaders instead of two: -1.6% on the MiniLM and mpnet layer shapes, same-process A/B)
            const bool isA = wave < 6;
            const int half = wave & 1;
            const int ld = isA ? g.lda : g.ldb, c0 = isA ? n0 : k0, width = isA ? g.N : g.K;
            const op16* base = (const op16*)(isA ? g.A : g.B) + (size_t)row0 * ld + c0;
            // range = rows [row0, mend); the last row's tail past the allocation reads as zero
            const uint32_t bytes = (uint32_t)min((size_t)(mend - row0) * ld * 2u - (size_t)c0 * 2u, (size_t)0x7FFFFF00u);
            const __amdgpu_buffer_rsrc_t rs = make_rsrc(base, bytes);
            // an operand stage is 768 chunks = 12 wave-instructions of 1 KB. LDS position p = q*64 + lane -> row p/24,
            // chunk position p%24 -> logical chunk = pos ^ swz(row). Columns beyond the matrix width must not alias
            // the next row: those lanes get an out-of-range offset (-> zero fill). (Cutting the stage into three
            // [32][64] sub-images so that every instruction moves 8 whole 128-byte lines changed nothing: 169.2 vs
            // 168.3 us, same-process A/B.)
            constexpr int NDMA = TBK * 24 / 64 / 2;       // DMA instructions per loader wave and stage (1 KB each): 12
            uint32_t vo[NDMA];
#pragma unroll
            for (int t = 0; t < NDMA; ++t) {
                const int pp = (half * NDMA + t) * 64 + lane;
                const int row = pp / 24, chunk = (pp % 24) ^ tn_swz(row);
                vo[t] = (c0 + chunk * 8 < width) ? (uint32_t)row * ld * 2u + chunk * 16u : kOOB;
            }
            auto issue = [&](int mt) {
                char* st = smem + (mt % TSTAGES) * TT_STAGE + (isA ? 0 : TT_TILE);
                const uint32_t so = (uint32_t)mt * TBK * ld * 2u;
#pragma unroll
                for (int t = 0; t < NDMA; ++t) dma16(rs, st + (half * NDMA + t) * 1024, vo[t], so);  // kOOB + so < 2^32: no wrap
            };
            // All TSTAGES-1 slots that are not being read are kept in flight (120 KB per CU). Timing experiments on this
            // kernel (same-process A/B): MFMAs compiled out -17%, atomic flush compiled out -16%, constant LDS slot
            // (no per-stage address arithmetic) -8%, two -> four loaders and 72 -> 120 KB in flight -1.7% together:
            // no single resource bounds it; the skeleton of barriers + L2->LDS streaming is 70% of the time.
            constexpr int AHEAD = TSTAGES - 1;
#pragma unroll 1
            for (int st = 0; st < AHEAD && st < nm; ++st) issue(st);
#pragma unroll 1
            for (int mt = 0; mt < nm; ++mt) {
                // stage mt must have landed before this wave arrives at the barrier that releases it to the MFMA waves
                const int younger = min(AHEAD - 1, nm - 1 - mt);       // 0 or 1 stages (NDMA = 12 instructions each) stay in flight
                if (younger >= 1) wait_vmcnt_n<12>(); else wait_vmcnt_n<0>();
                __builtin_amdgcn_s_barrier();              // MFMA waves are done with stage mt-1 (and older)
                if (mt + AHEAD < nm) issue(mt + AHEAD);    // into the slot of stage mt-1
            }
            __builtin_amdgcn_s_barrier();                  // end of piece: every MFMA wave has left the ring
            continue;
        }

        // ------------------------------------------------------------ MFMA waves
        f32x16 acc[3][3];
        float bsum[3] = {0.f, 0.f, 0.f};       // bias gradient: per-lane partial column sums of the dY fragments
#pragma unroll
        for (int i = 0; i < 3; ++i)
#pragma unroll
            for (int j = 0; j < 3; ++j)
#pragma unroll
                for (int r = 0; r < 16; ++r) acc[i][j][r] = 0.f;
        const bool do_bias = (g.colsum != nullptr) && (k0 == 0) && (wn == 0);

        for (int mt = 0; mt < nm; ++mt) {
            __builtin_amdgcn_s_barrier();                  // stage mt landed (the loaders waited for it before arriving)
            const char* pa = smem + (mt % TSTAGES) * TT_STAGE;
            const char* pb = pa + TT_TILE;
            // One MFMA wave per SIMD: nothing but its own instruction stream hides the LDS latency of its fragment reads. The
            // fragments of k-step ks + 1 are therefore read into a second register set while the MFMAs of ks issue (hipcc's own
            // schedule reads a k-step's 12 fragments, waits, multiplies: ~250 exposed cycles per 288 of MFMA, in-kernel stamps
            // of round 1: 990 cycles per 32-row stage for 576 of MFMA); the order is pinned with sched_group_barrier.
            op16x8 fa[2][3], fb[2][3];
#define TN_LOAD(ks_, set_)                                                                                  \
    do {                                                                                                    \
        _Pragma("unroll") for (int i = 0; i < 3; ++i) {                                                     \
            _Pragma("unroll") for (int jj = 0; jj < 2; ++jj) {                                              \
                const int row = (ks_) * 16 + 8 * fh + 4 * jj + q;                                           \
                const int ca = wm * 12 + i * 4 + gsel * 2 + (p >> 1);                                       \
                const int cb = wn * 12 + i * 4 + gsel * 2 + (p >> 1);                                       \
                const op16x4 ta = lds_tr16(pa + tn_off(row, ca) + 8 * (p & 1));                             \
                const op16x4 tb = lds_tr16(pb + tn_off(row, cb) + 8 * (p & 1));                             \
                _Pragma("unroll") for (int e = 0; e < 4; ++e) { fa[set_][i][jj * 4 + e] = ta[e]; fb[set_][i][jj * 4 + e] = tb[e]; } \
            }                                                                                               \
        }                                                                                                   \
    } while (0)
#define TN_MFMA(set_)                                                                                       \
    do {                                                                                                    \
        _Pragma("unroll") for (int i = 0; i < 3; ++i)                                                       \
            _Pragma("unroll") for (int j = 0; j < 3; ++j)                                                   \
                acc[i][j] = mfma32_op(fa[set_][i], fb[set_][j], acc[i][j]); \
        if (do_bias) {   /* lane (n' = lane&31, half h) holds dY[m = 8h .. 8h+7][n']: 8 of the 16 rows of this k-step */ \
            _Pragma("unroll") for (int i = 0; i < 3; ++i)                                                   \
                _Pragma("unroll") for (int e = 0; e < 8; ++e) bsum[i] += (float)fa[set_][i][e];             \
        }                                                                                                   \
    } while (0)
            TN_LOAD(0, 0);
            TN_LOAD(1, 1);
            TN_MFMA(0);
            TN_LOAD(2, 0);
            TN_MFMA(1);
            TN_LOAD(3, 1);
            TN_MFMA(0);
            TN_MFMA(1);
#undef TN_LOAD
#undef TN_MFMA
            if (!do_bias) {
                // issue order: the 12 reads of k-step 0; then the 12 reads of each later k-step spread behind the 9 MFMAs of the
                // k-step before it (two behind each of the first three, one behind each of the other six); then the last 9 MFMAs
                __builtin_amdgcn_sched_group_barrier(0x100, 12, 0);
#pragma unroll
                for (int rep = 0; rep < 3; ++rep) {
#pragma unroll
                    for (int u = 0; u < 3; ++u) {
                        __builtin_amdgcn_sched_group_barrier(0x008, 1, 0);
                        __builtin_amdgcn_sched_group_barrier(0x100, 2, 0);
                    }
#pragma unroll
                    for (int u = 0; u < 6; ++u) {
                        __builtin_amdgcn_sched_group_barrier(0x008, 1, 0);
                        __builtin_amdgcn_sched_group_barrier(0x100, 1, 0);
                    }
                }
                __builtin_amdgcn_sched_group_barrier(0x008, 9, 0);
            }
        }
        __builtin_amdgcn_s_barrier();                      // end of piece: the loaders may refill the ring while we flush

        float* C = (float*)g.C;
        // Partial tiles are combined with float atomics (executed at the memory side, ~1.3 TB/s chip-wide: the 75 MB of
        // partial sums of a MiniLM layer cost 43 us of the launch). A plain-store flush into per-range slots plus a
        // fixed-order reduce kernel was built in round 2 and removed in round 3: same time, 75 MB more scratch.
#pragma unroll
        for (int j = 0; j < 3; ++j) {
            const int k = k0 + wn * 96 + j * 32 + fr;
            if (k >= g.K) continue;
#pragma unroll
            for (int i = 0; i < 3; ++i) {
#pragma unroll
                for (int r = 0; r < 16; ++r) {
                    const int n = n0 + wm * 96 + i * 32 + (r & 3) + 8 * (r >> 2) + 4 * fh;
                    if (n < g.N) atomicAdd(&C[(size_t)n * g.ldc + k], acc[i][j][r]);
                }
            }
        }
        if (do_bias) {
#pragma unroll
            for (int i = 0; i < 3; ++i) {
                const float t = bsum[i] + swap32(bsum[i]);             // both row halves of the k-steps
                const int n = n0 + wm * 96 + i * 32 + fr;
                if (fh == 0 && n < g.N) atomicAdd(&g.colsum[n], t);
            }
        }
    }
}

}  // namespace

int qst_gemm8_mode_get();                               // gemm8.hip
// Shapes on which the 8-phase kernel (its 256 x 256 tile) is taken without being asked. It needs several tiles per CU (one
// workgroup per CU, nothing overlaps its epilogue: at 384 tiles -- N = 768, M = 32768 -- it loses to the tiled kernel on every
// epilogue) and a long reduction or a cheap epilogue. Back to back at M = 196,608 (tools/g8_bench.py 768 196608, us, tiled ->
// 8-phase): QKV 735 -> 691, FFN-1 + GELU 1,348 -> 1,276, out-proj dgrad (bf16, N = K = 768) 273 -> 232, FFN-2 + residual
// (K = 3072) 1,042 -> 904 / 987 -> 856, QKV dgrad + residual (K = 2304) 761 -> 691; GELU' dgrad 1,157 -> 1,202 and out-proj +
// residual (K = 768) 357 -> 406 stay tiled. In the step, same process (tools/ab_gemm8.py): bert-base B = 128 L = 384 bf16
// 156.2 -> 153.6 ms with the first two alone (round 4, finding 26), with the K >= 2304 residual GEMMs as well: see DESIGN.md.
static bool nt8_auto(const QstGemmArgs* a, int epi) {
    const int64_t tiles = (int64_t)((a->M + 255) / 256) * ((a->N + 255) / 256);
    if (tiles < 1024 || a->K < 768) return false;
    if (epi == QST_EPI_BF16) return true;
    if (epi == QST_EPI_GELU) return a->N >= 2304;
    if (epi == QST_EPI_GELU_BWD) return a->N >= 2304 && tiles >= 4096;     // (1,183 -> 1,149 us at M = 196,608; 198 -> 202 at 32,768)
    if (epi == QST_EPI_F32_RESID || epi == QST_EPI_F32_RESID_BF16) return a->K >= 2304;
    return false;
}

template <int EPI, int WAVES_M, int WAVES_N = 2, int TI = 2>
static int launch_nt(const QstGemmArgs* a, hipStream_t st) {
    constexpr int NBM = 32 * TI * WAVES_M, NBN = 96 * WAVES_N;
    // ring (64-deep stages; 32-deep for the tall wave tile); the epilogue staging (WAVES x 12.8 KB) fits inside
    constexpr int lds = 2 * (NBM + NBN) * (TI == 4 ? 32 : NBK) * 2;
    static QstLdsAttr attr;
    if (int rc = qst_ensure_lds(attr, (const void*)gemm_nt_kernel<EPI, WAVES_M, WAVES_N, TI>, lds)) return rc;
    const int ntm = (a->M + NBM - 1) / NBM, ntn = (a->N + NBN - 1) / NBN;
    gemm_nt_kernel<EPI, WAVES_M, WAVES_N, TI><<<dim3(ntm * ntn), dim3(64 * WAVES_M * WAVES_N), lds, st>>>(*a);
    QST_LAUNCH_CHECK();
    return QST_OK;
}

extern "C" int QST_K(qst_gemm_nt)(const QstGemmArgs* a, int epi, void* stream) {
    if (!a || !a->A || !a->B || !a->C || a->M <= 0 || a->N <= 0 || a->K <= 0) return QST_ERR_BAD_ARG;
    if (a->K % NBK != 0 || a->lda % 8 != 0 || a->ldb % 8 != 0 || a->N % 4 != 0 || a->ldc % 4 != 0) return QST_ERR_UNSUPPORTED;
    if ((int64_t)256 * a->lda * 2 >= 0x7FFFFF00LL || (int64_t)384 * a->ldb * 2 >= 0x7FFFFF00LL) return QST_ERR_UNSUPPORTED;
    if (a->drop.thr16 && a->drop.state) {
        if (a->drop_where != 1 || (epi != QST_EPI_F32_RESID && epi != QST_EPI_F32_RESID_BF16)) return QST_ERR_BAD_ARG;
        if (a->drop.thr16 > 65535u || (int64_t)a->M * a->N >= ((int64_t)1 << 32)) return QST_ERR_UNSUPPORTED;
    }
    hipStream_t st = (hipStream_t)stream;
    // The 8-wave, 8-phase K loop (gemm8.hip): a->splits bit 5 forces its 128 x 384 tile, bit 6 its 256 x 256 tile, bit 7
    // forbids it; otherwise qst_gemm8_mode / the shape decide (nt8_auto).
    const bool no8 = a->B2 != nullptr;                 // split weights: the tiled kernels only
    if (!no8 && (a->splits & 0x60) && QST_K(qst_gemm_nt8_supported)(a, epi)) return QST_K(qst_gemm_nt8)(a, epi, (a->splits & 0x40) ? 1 : 0, stream);
    if (!no8 && !(a->splits & 0xFE7) && QST_K(qst_gemm_nt8_supported)(a, epi)) {
        const int mode = qst_gemm8_mode_get();
        if (mode >= 0 ? (mode & 1) != 0 : nt8_auto(a, epi)) return QST_K(qst_gemm_nt8)(a, epi, mode >= 0 ? 0 : 1, stream);
    }
    // Two 128-row workgroups per CU beat one 256-row workgroup on every shape of the step (their MFMA and
    // store phases interleave); a->splits (unused by nt otherwise) can force the tile height: 1 = 128, 2 = 256 rows.
    const bool small = (a->splits & 3) != 2;
    if ((a->splits & 3) == 3 && epi == QST_EPI_F32_RESID) return launch_nt<QST_EPI_F32_RESID, 2, 4>(a, st);   // experiment
    // Tall wave tiles (256 x 192 workgroup tile, four waves of 128 x 96): measured 4-7% faster than 128 x 192 on the
    // K >= 768 GEMMs with bf16 outputs (H = 768 FFN1 forward 228 vs 245 us, GELU' dgrad 206 vs 215), no gain at
    // K = 384 where the epilogue dominates; a->splits == 4 forces it, == 1 forbids it.
    const bool tall_ok = epi == QST_EPI_BF16 || epi == QST_EPI_GELU || epi == QST_EPI_GELU_BWD;
    const int64_t tall_tiles = (int64_t)((a->M + 255) / 256) * ((a->N + 191) / 192);
    const bool tall_auto = (a->splits & 7) == 0 && a->K >= 768 && a->N >= 1536 && tall_tiles >= 1024;
    if (tall_ok && ((a->splits & 7) == 4 || tall_auto)) {
        switch (epi) {
            case QST_EPI_BF16: return launch_nt<QST_EPI_BF16, 2, 2, 4>(a, st);
            case QST_EPI_GELU: return launch_nt<QST_EPI_GELU, 2, 2, 4>(a, st);
            default: return launch_nt<QST_EPI_GELU_BWD, 2, 2, 4>(a, st);
        }
    }
#define QST_NT_CASE(E) case E: return small ? launch_nt<E, 2>(a, st) : launch_nt<E, 4>(a, st);
    switch (epi) {
        QST_NT_CASE(QST_EPI_BF16)
        QST_NT_CASE(QST_EPI_F32_RESID)
        QST_NT_CASE(QST_EPI_GELU)
        QST_NT_CASE(QST_EPI_GELU_BWD)
        QST_NT_CASE(QST_EPI_F32_RESID_BF16)
        default: return QST_ERR_BAD_ARG;
    }
#undef QST_NT_CASE
}

#if !QST_OP_F16
template <int EPI>
static int launch_nt_f8(const QstGemmArgs* a, hipStream_t st) {
    constexpr int lds = 2 * (128 + 192) * 128;                  // 80 KB ring; the epilogue staging (52.7 KB) fits inside
    static QstLdsAttr attr;
    if (int rc = qst_ensure_lds(attr, (const void*)gemm_nt_f8_kernel<EPI>, lds)) return rc;
    const int ntm = (a->M + 127) / 128, ntn = (a->N + 191) / 192;
    gemm_nt_f8_kernel<EPI><<<dim3(ntm * ntn), dim3(256), lds, st>>>(*a);
    QST_LAUNCH_CHECK();
    return QST_OK;
}

// Shapes on which the MXFP8 GEMM takes the 8-phase kernel by itself. Back to back at M = 196,608 (tools/f8_8phase_bench.py, us,
// tiled -> 8-phase 256 x 256): QKV 440 -> 409, FFN-2 + residual (K = 3072) 647 -> 582, FFN-1 + GELU 877 -> 824, out-proj +
// residual (K = 768, N = 768: epilogue-bound) 248 -> 335.
static bool f8_auto(const QstGemmArgs* a, int epi) {
    if (a->M < 16384) return false;
    if (epi == QST_EPI_BF16) return a->N >= 2304 && a->K >= 768;
    if (epi == QST_EPI_F32_RESID) return a->K >= 2304;
    // (QST_EPI_GELU_MX_TRAIN, FFN-1 of the fp8 training forward, has an 8-phase form too -- a->splits bit 6 -- and stays tiled:
    //  three outputs, 5 bytes per element: 1,104 vs 1,030 us back to back at M = 196,608, 1,355 vs 1,256 us inside the step)
    return false;
}

extern "C" int qst_gemm_nt_f8(const QstGemmArgs* a, int epi, void* stream) {
    if (!a || !a->A || !a->B || !a->C || !a->aux || !a->bscale || a->M <= 0 || a->N <= 0 || a->K <= 0) return QST_ERR_BAD_ARG;
    if (a->K % 128 != 0 || a->lda % 16 != 0 || a->ldb % 16 != 0 || a->N % 8 != 0) return QST_ERR_UNSUPPORTED;
    if (epi == QST_EPI_GELU_MX && a->ldc != a->N) return QST_ERR_UNSUPPORTED;          // stage-major scales: one matrix, no sub-views
    if (a->drop.thr16 && a->drop.state) {             // dropout of a projection output (fp8 training forward), as qst_gemm_nt
        if (a->drop_where != 1 || epi != QST_EPI_F32_RESID) return QST_ERR_BAD_ARG;
        if (a->drop.thr16 > 65535u || (int64_t)a->M * a->N >= ((int64_t)1 << 32)) return QST_ERR_UNSUPPORTED;
    }
    if ((int64_t)128 * a->lda >= 0x7FFFFF00LL || (int64_t)192 * a->ldb >= 0x7FFFFF00LL) return QST_ERR_UNSUPPORTED;
    hipStream_t st = (hipStream_t)stream;
    // the 8-phase form (gemm8.hip: 16x16x128 MFMA, 256 x 256 tile): bit-identical; a->splits bit 6 forces it, bit 7 forbids it,
    // otherwise qst_gemm8_mode bit 0 / the shape decide (f8_auto)
    const bool mxt_ok = epi == QST_EPI_GELU_MX_TRAIN && a->C2 && a->C3 && a->C4 && a->ldc == a->N && a->N % 32 == 0;
    if (!(a->splits & 0x80) && (epi == QST_EPI_BF16 || epi == QST_EPI_F32_RESID || epi == QST_EPI_GELU || mxt_ok) && a->N % 8 == 0 && a->ldc % 8 == 0) {
        const int mode = qst_gemm8_mode_get();
        // the 8-phase entry point has stricter limits than the tiled kernel (32-bit scale offsets, 256 / 384-row descriptors): a
        // shape it would refuse falls through to the tiled launch unless the caller FORCED the form
        const bool fits8 = (int64_t)256 * a->lda < 0x7FFFFF00LL && (int64_t)384 * a->ldb < 0x7FFFFF00LL &&
                           (int64_t)(a->K / 128) * (a->M > a->N ? a->M : a->N) * 4 < 0x7FFFFF00LL &&
                           !(epi == QST_EPI_F32_RESID && a->resid && a->ldr % 4 != 0);
        if (a->splits & 0x40) return qst_gemm_nt8_f8(a, epi, 1, stream);
        if (fits8 && (mode >= 0 ? (mode & 1) != 0 : f8_auto(a, epi))) return qst_gemm_nt8_f8(a, epi, 1, stream);
    }
    switch (epi) {
        case QST_EPI_BF16: return a->ldc % 4 ? QST_ERR_UNSUPPORTED : launch_nt_f8<QST_EPI_BF16>(a, st);
        case QST_EPI_F32_RESID: return a->ldc % 4 ? QST_ERR_UNSUPPORTED : launch_nt_f8<QST_EPI_F32_RESID>(a, st);
        case QST_EPI_GELU: return (!a->C2 || a->ldc % 4) ? QST_ERR_UNSUPPORTED : launch_nt_f8<QST_EPI_GELU>(a, st);
        case QST_EPI_GELU_MX:
            if (!a->C2 || a->ldc % 32 != 0 || a->N % 32 != 0) return QST_ERR_UNSUPPORTED;
            return launch_nt_f8<QST_EPI_GELU_MX_>(a, st);
        case QST_EPI_GELU_MX_TRAIN:
            if (!a->C2 || !a->C3 || !a->C4 || a->ldc != a->N || a->N % 32 != 0) return QST_ERR_UNSUPPORTED;
            return launch_nt_f8<QST_EPI_GELU_MX_TRAIN_>(a, st);
        default: return QST_ERR_BAD_ARG;
    }
}

extern "C" int qst_quant_mx(const void* src, int src_is_bf16, int64_t rows, int K, void* q, void* scales, void* stream) {
    if (!src || !q || !scales || rows <= 0 || K <= 0) return QST_ERR_BAD_ARG;
    if (K % 32 != 0) return QST_ERR_UNSUPPORTED;
    const int64_t n8 = rows * K / 8;
    const unsigned grid = (unsigned)((n8 + 255) / 256);
    if (src_is_bf16) quant_mx_kernel<op16><<<grid, 256, 0, (hipStream_t)stream>>>((const op16*)src, n8, rows, K, (uint8_t*)q, (uint8_t*)scales);
    else quant_mx_kernel<float><<<grid, 256, 0, (hipStream_t)stream>>>((const float*)src, n8, rows, K, (uint8_t*)q, (uint8_t*)scales);
    QST_LAUNCH_CHECK();
    return QST_OK;
}

extern "C" int qst_gemm_nt_ln_supported(int N) { return (N == LN_N || qst_gemm_nt8_ln_supported(N)) ? 1 : 0; }
// rows per partial-sum block of mode 1 (the stride of `partials`): one full-row tile of 128 rows at N = 384, a 256-row panel
// of 256-column tiles above (gemm8.hip)
extern "C" int qst_gemm_nt_ln_block_rows(int N) { return N == LN_N ? 128 : 256; }
// ... exactly: above N = 384 the tile -- and with it the block height -- depends on M too (gemm8.hip: 256 x 256 or 128 x 384)
extern "C" int qst_gemm_nt_ln_block_rows_m(int N, int M) { return N == LN_N ? 128 : qst_gemm_nt8_ln_block_rows(M, N); }
#endif  // !QST_OP_F16

extern "C" int QST_K(qst_gemm_nt_ln)(const QstGemmArgs* a, const QstLnEpi* ln, int mode, void* stream) {
    if (!a || !ln || !a->A || !a->B || !a->C || !ln->gamma || a->M <= 0 || a->K <= 0) return QST_ERR_BAD_ARG;
    if (mode != 0 && mode != 1) return QST_ERR_BAD_ARG;
    if (mode == 0 && !ln->beta) return QST_ERR_BAD_ARG;
    if (mode == 1 && (!ln->xhat || !ln->rstd)) return QST_ERR_BAD_ARG;
    if (a->N != LN_N && qst_gemm_nt8_ln_supported(a->N)) return QST_K(qst_gemm_nt8_ln)(a, ln, mode, stream);   // several tiles per row
    if (a->N != LN_N || a->K % NBK != 0 || a->lda % 8 != 0 || a->ldb % 8 != 0 || a->ldc % 2 != 0 || (a->resid && a->ldr % 2 != 0))
        return QST_ERR_UNSUPPORTED;
    if ((int64_t)128 * a->lda * 2 >= 0x7FFFFF00LL || (int64_t)LN_N * a->ldb * 2 >= 0x7FFFFF00LL) return QST_ERR_UNSUPPORTED;
    if (a->drop.thr16 && a->drop.state) {
        if (a->drop.thr16 > 65535u || (int64_t)a->M * a->N >= ((int64_t)1 << 32)) return QST_ERR_UNSUPPORTED;
        if (mode == 0 ? a->drop_where != 1 : (a->drop_where != 2 && a->drop_where != 3)) return QST_ERR_BAD_ARG;
    }
    static QstLdsAttr attr0, attr1, attr01, attr12, attr13;
    const int ntm = (a->M + 127) / 128;
    if (a->drop.thr16 && a->drop.state) {
        if (int rc = qst_ensure_lds(attr01, (const void*)gemm_nt_ln_kernel<0, 1>, LN_LDS)) return rc;
        if (int rc = qst_ensure_lds(attr12, (const void*)gemm_nt_ln_kernel<1, 2>, LN_LDS)) return rc;
        if (int rc = qst_ensure_lds(attr13, (const void*)gemm_nt_ln_kernel<1, 3>, LN_LDS)) return rc;
        if (mode == 0) gemm_nt_ln_kernel<0, 1><<<dim3(ntm), dim3(512), LN_LDS, (hipStream_t)stream>>>(*a, *ln);
        else if (a->drop_where == 2) gemm_nt_ln_kernel<1, 2><<<dim3(ntm), dim3(512), LN_LDS, (hipStream_t)stream>>>(*a, *ln);
        else gemm_nt_ln_kernel<1, 3><<<dim3(ntm), dim3(512), LN_LDS, (hipStream_t)stream>>>(*a, *ln);
        QST_LAUNCH_CHECK();
        return QST_OK;
    }
    if (int rc = qst_ensure_lds(attr0, (const void*)gemm_nt_ln_kernel<0>, LN_LDS)) return rc;
    if (int rc = qst_ensure_lds(attr1, (const void*)gemm_nt_ln_kernel<1>, LN_LDS)) return rc;
    if (mode == 0) gemm_nt_ln_kernel<0><<<dim3(ntm), dim3(512), LN_LDS, (hipStream_t)stream>>>(*a, *ln);
    else gemm_nt_ln_kernel<1><<<dim3(ntm), dim3(512), LN_LDS, (hipStream_t)stream>>>(*a, *ln);
    QST_LAUNCH_CHECK();
    return QST_OK;
}

static bool tn8_auto(const QstTnGroup* g) { (void)g; return false; }

extern "C" int QST_K(qst_gemm_tn_group)(const QstTnGroup* grp_in, void* stream) {
    if (!grp_in || grp_in->nprob <= 0 || grp_in->nprob > QST_TN_MAX_PROB) return QST_ERR_BAD_ARG;
    {
        const int mode = qst_gemm8_mode_get();
        if (mode >= 0 ? (mode & 2) != 0 : tn8_auto(grp_in)) return QST_K(qst_gemm_tn8_group)(grp_in, stream);
    }
    QstTnGroup g = *grp_in;
    g.total_tiles = 0;
    for (int i = 0; i < g.nprob; ++i) {
        const QstGemmArgs& a = g.prob[i];
        if (!a.A || !a.B || !a.C || a.M <= 0 || a.N <= 0 || a.K <= 0 || a.M != g.prob[0].M) return QST_ERR_BAD_ARG;
        if (a.lda % 8 != 0 || a.ldb % 8 != 0 || a.N % 8 != 0 || a.K % 8 != 0) return QST_ERR_UNSUPPORTED;
        if ((int64_t)a.M * a.lda * 2 >= 0x7FFFFF00LL || (int64_t)a.M * a.ldb * 2 >= 0x7FFFFF00LL) return QST_ERR_UNSUPPORTED;
        g.tiles[i] = ((a.N + TT - 1) / TT) * ((a.K + TT - 1) / TT);
        g.total_tiles += g.tiles[i];
    }
    const int M = g.prob[0].M;
    if (g.splits <= 0) {
        // one M-range per XCD (every operand row leaves HBM once); a second range per XCD only when the ranges are
        // long enough that 32 workgroups per range still get >= 8 stages of work each
        g.splits = 8;
    }
    g.splits = (g.splits + 7) / 8 * 8;
    g.ranges_per_xcd = g.splits / 8;
    // 32 workgroups per XCD (one per CU) share each XCD's ranges; never more workgroups than (tile, stage) pairs
    const int64_t stages = ((int64_t)(M + g.splits - 1) / g.splits + TBK - 1) / TBK;
    int64_t wg_per_range = 32 / g.ranges_per_xcd;
    if (wg_per_range < 1) wg_per_range = 1;
    const int64_t work = (int64_t)g.total_tiles * stages;
    if (wg_per_range > work) wg_per_range = work < 1 ? 1 : work;
    const int grid = (int)(8 * g.ranges_per_xcd * wg_per_range);
    static QstLdsAttr attr;
    if (int rc = qst_ensure_lds(attr, (const void*)gemm_tn_group_kernel, TT_LDS)) return rc;
    gemm_tn_group_kernel<<<dim3(grid), dim3(512), TT_LDS, (hipStream_t)stream>>>(g);
    QST_LAUNCH_CHECK();
    return QST_OK;
}

extern "C" int QST_K(qst_gemm_tn)(const QstGemmArgs* a, void* stream) {
    if (!a) return QST_ERR_BAD_ARG;
    QstTnGroup g{};
    g.nprob = 1;
    g.splits = a->splits;
    g.prob[0] = *a;
    return QST_K(qst_gemm_tn_group)(&g, stream);
}
