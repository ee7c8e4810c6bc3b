// gemm_pp.hip -- persistent "ping-pong" NT GEMM for the large-M launches of the step (gfx950, wave64).
//
//   C[M,N] = A[M,K] . B[N,K]^T (+ fused epilogue), the same contract as gemm_nt_kernel in gemm.hip (nn.Linear forward and
//   dgrad inside BertLayer: transformers modeling_bert.py:154-156, 282-293, 325-351).
//
// Why a second form. Round-2 measurements of the tiled kernel (DESIGN.md findings 3, 4, 13): a workgroup spends 6.9 us in
// its six-stage K loop -- 3.6x the time of its MFMAs, because every MFMA wave also issues the stage's LDS-DMAs (10 per
// wave and stage at 60-185 cycles each, in order with its MFMAs) -- and then 3.5-9.6 us in an epilogue that moves bytes
// and issues no MFMA; all workgroups of the chip sit in the same phase, so HBM idles during K loops and the matrix cores
// idle during epilogues. This kernel separates the three jobs inside ONE persistent workgroup per CU (12 waves):
//   * waves 8-11 are loaders: they issue every LDS-DMA (10 each per 64-deep stage) into a 3-slot ring and never compute;
//   * waves 0-3 and 4-7 are two MFMA groups that ALTERNATE tiles: while one group runs the K loop of tile t (128 x 192,
//     64 x 96 per wave as in gemm.hip), the other writes out tile t-1 -- its epilogue cut into six 32 x 32 slices, one per
//     K stage -- so stores / GELU arithmetic of one tile run under the MFMAs of the next, and the loaders keep streaming
//     across tile boundaries (the first two stages of tile t+1 are in flight before tile t's last MFMA).
//   One s_barrier per stage (a "super-step") synchronises all 12 waves: the loaders wait (counted vmcnt) for the stage that is
//   about to be read, the MFMA group has finished the stage whose slot is refilled next. An MFMA wave never waits on a DMA:
//   its vmcnt counts only its own epilogue loads / stores.
// Each SIMD hosts one loader, one wave of group 0 and one of group 1 (waves w, w+4, w+8 share a SIMD), i.e. an MFMA stream
// beside a VALU/VMEM stream -- the pairing the hardware overlaps (MI355X_MICROARCH.md, wave scheduling). Each role is ONE
// wave per SIMD, so nothing hides its latencies for it; the code is written accordingly:
//   * K loop: the fragments of k-step ks+1 are read into a second register set while the MFMAs of ks issue (pinned with
//     sched_group_barrier; hipcc's own schedule keeps half a k-step in flight and exposes an LDS round trip every 3 MFMAs);
//   * epilogue: two slabs per wave -- slice s is read back from the slab it was written to one super-step earlier while
//     block s+1 goes into the other one, so no write -> read round trip sits in a slice; the GELU arithmetic runs on eight
//     elements in lockstep (hipcc emits one dependent chain per element: 164 cycles per element measured, in-kernel stamps).
// LDS: ring 3 x 40 KB, 2 x (32 x 36 fp32) slabs per wave of a group (the two groups never write out at the same time and
// share them; the 32 x 32 MFMA block is written as it stands and read back row-wise, so global accesses are 16-byte
// row-contiguous), 96 bias values per MFMA wave: 159 KB.
#include "qst_common.h"
#include "qst_kernels.h"

namespace {

typedef __attribute__((address_space(3))) void lds_void;

__device__ __forceinline__ void pp_dma16(__amdgpu_buffer_rsrc_t r, char* lds_wave_base, uint32_t voff, uint32_t soff) {
    __builtin_amdgcn_raw_ptr_buffer_load_lds(r, (lds_void*)lds_wave_base, 16, (int)voff, (int)soff, 0, 0);
}

constexpr int PBM = 128, PBN = 192, PBK = 64;
constexpr int PA_BYTES = PBM * PBK * 2;              // 16 KB
constexpr int PB_BYTES = PBN * PBK * 2;              // 24 KB
constexpr int PSTAGE = PA_BYTES + PB_BYTES;          // 40 KB
constexpr int PSLOTS = 3;
constexpr int PRING = PSLOTS * PSTAGE;               // 120 KB
constexpr int PSTG_LD = 36;                          // slab row stride in floats (144 B: conflict-free ds_write_b128)
constexpr int PSLAB = 32 * PSTG_LD;                  // floats per slab (4608 B)
constexpr int PBIAS_OFF = PRING + 4 * 2 * PSLAB * 4; // after the 4 x 2 slabs
constexpr int PLDS = PBIAS_OFF + 8 * 96 * 4;         // 162,816 B

// 16-byte chunk c of row r of a [rows][64 bf16] operand image sits at chunk position c ^ ((r >> 1) & 7) (as gemm.hip)
__device__ __forceinline__ uint32_t pp_off(int row, int chunk) {
    return (uint32_t)(row * 128 + ((chunk ^ ((row >> 1) & 7)) << 4));
}

template <int EPI> constexpr bool pp_f32_out() { return EPI == QST_EPI_F32_RESID || EPI == QST_EPI_F32_RESID_BF16; }

constexpr uint32_t kPpOOB = 0x7FFFFFF0u;             // voffset that always fails the buffer range check (loads 0, stores nothing)

// Buffer descriptors of the epilogue operands (wave-uniform: SGPRs). Every epilogue access is a raw buffer load / store:
// per-lane 32-bit offsets inside the slice + the slice's origin as the scalar offset, rows / columns outside the matrix
// get kPpOOB instead of a branch.
struct PpRsrc { __amdgpu_buffer_rsrc_t c, c2, in; };
struct PpPf { u32x4 a[4]; };                         // epilogue inputs of the NEXT slice (aux rows or residual rows)

// erf GELU and its derivative for four elements in lockstep, on SCALAR VALU instructions: an epilogue wave runs alone
// beside an MFMA wave on its SIMD, so (a) nothing but its own independent instructions covers the ~8-cycle latency of a
// dependent one -- every step below is applied to all four elements before the next (sched_barrier keeps hipcc from
// re-serialising it) -- and (b) packed f32 instructions, which cost 13-30 issue cycles there against 4 for a scalar one
// (MI355X_MICROARCH.md, per-instruction constants), are avoided: this file is compiled with -fno-slp-vectorize.
// Abramowitz-Stegun 7.1.26 on v_rcp_f32 / v_exp_f32, operation for operation the arithmetic of gelu_parts2 (qst_common.h):
//   z = |x| / sqrt2, t = 1 / (1 + p z), erf(z) = 1 - (a1 t + ... + a5 t^5) exp(-z^2);  h = x Phi(x),  g = Phi(x) + x phi(x).
#define PP_PHASE(expr_) do { _Pragma("unroll") for (int k = 0; k < 4; ++k) { expr_; } __builtin_amdgcn_sched_barrier(0); } while (0)
__device__ __forceinline__ void pp_gelu4(const float* x, float* g, float* h) {
    float z[4], t[4], e[4], p[4];
    __builtin_amdgcn_sched_barrier(0);
    PP_PHASE(z[k] = fabsf(x[k]) * 0.70710678118654752f);
    PP_PHASE(t[k] = __builtin_fmaf(z[k], 0.3275911f, 1.0f));
    PP_PHASE(e[k] = z[k] * z[k]);
    PP_PHASE(t[k] = __builtin_amdgcn_rcpf(t[k]));
    PP_PHASE(e[k] = e[k] * -1.4426950408889634f);
    PP_PHASE(e[k] = __builtin_amdgcn_exp2f(e[k]));                       // exp(-x^2 / 2)
    PP_PHASE(p[k] = __builtin_fmaf(t[k], 1.061405429f, -1.453152027f));
    PP_PHASE(p[k] = __builtin_fmaf(p[k], t[k], 1.421413741f));
    PP_PHASE(p[k] = __builtin_fmaf(p[k], t[k], -0.284496736f));
    PP_PHASE(p[k] = __builtin_fmaf(p[k], t[k], 0.254829592f));
    PP_PHASE(p[k] = p[k] * t[k]);
    PP_PHASE(p[k] = p[k] * e[k]);
    PP_PHASE(p[k] = __builtin_fmaf(p[k], -0.5f, 0.5f));                   // 0.5 erf(|x| / sqrt2)
    PP_PHASE(p[k] = __builtin_copysignf(p[k], x[k]));
    PP_PHASE(p[k] = p[k] + 0.5f);                                         // Phi(x)
    PP_PHASE(e[k] = e[k] * 0.39894228040143268f);                         // phi(x)
    PP_PHASE(g[k] = __builtin_fmaf(x[k], e[k], p[k]));
    PP_PHASE(h[k] = x[k] * p[k]);
}
#undef PP_PHASE

// Slice (i, j) = 32 x 32 block of the wave's 64 x 96 sub-tile whose first element is C[mw][nw].
// bf16 outputs: lane -> (row 16t + lane / 4, 8 columns from 8 (lane % 4)), t = 0, 1;  fp32: (row 8t + lane / 8, 4 columns
// from 4 (lane % 8)), t = 0..3.
template <int EPI>
__device__ __forceinline__ void pp_prefetch(const QstGemmArgs& g, const PpRsrc& rs, int mw, int nw, int i, int j, int lane, PpPf& pf) {
    const int m0 = mw + i * 32, n0 = nw + j * 32;
    if constexpr (EPI == QST_EPI_GELU_BWD) {
        const uint32_t so = ((uint32_t)m0 * (uint32_t)g.ldc + (uint32_t)n0) * 2u;
        const bool cok = n0 + (lane & 3) * 8 < g.N;
#pragma unroll
        for (int t = 0; t < 2; ++t) {
            const int row = 16 * t + (lane >> 2);
            const uint32_t vo = (cok && m0 + row < g.M) ? (uint32_t)row * g.ldc * 2u + (lane & 3) * 16u : kPpOOB;
            pf.a[t] = __builtin_amdgcn_raw_buffer_load_b128(rs.in, (int)vo, (int)so, 0);
        }
    } else if constexpr (pp_f32_out<EPI>()) {
        const uint32_t so = ((uint32_t)m0 * (uint32_t)g.ldr + (uint32_t)n0) * 4u;
        const bool cok = g.resid != nullptr && n0 + (lane & 7) * 4 < g.N;
#pragma unroll
        for (int t = 0; t < 4; ++t) {
            const int row = 8 * t + (lane >> 3);
            const uint32_t vo = (cok && m0 + row < g.M) ? (uint32_t)row * g.ldr * 4u + (lane & 7) * 16u : kPpOOB;
            pf.a[t] = __builtin_amdgcn_raw_buffer_load_b128(rs.in, (int)vo, (int)so, 0);
        }
    }
}

// one 32 x 32 accumulator block (D rows = n, D column = m on the lane: 4 consecutive n per register group) into a slab as [m][n]
__device__ __forceinline__ void pp_put(const f32x16& blk, float* slab, int lane) {
    const int fr = lane & 31, fh = lane >> 5;
#pragma unroll
    for (int g4 = 0; g4 < 4; ++g4) {
        f32x4 v;
#pragma unroll
        for (int e = 0; e < 4; ++e) v[e] = blk[4 * g4 + e];
        *(f32x4*)(slab + fr * PSTG_LD + 8 * g4 + 4 * fh) = v;
    }
}

// the block of a slice, read back row-wise from its slab: 16 floats per lane
struct PpBlk { f32x4 q[4]; };
template <int EPI>
__device__ __forceinline__ void pp_get(const float* slab, int lane, PpBlk& b) {
    if constexpr (pp_f32_out<EPI>()) {
#pragma unroll
        for (int t = 0; t < 4; ++t) b.q[t] = *(const f32x4*)(slab + (8 * t + (lane >> 3)) * PSTG_LD + (lane & 7) * 4);
    } else {
#pragma unroll
        for (int t = 0; t < 2; ++t) {
            b.q[2 * t] = *(const f32x4*)(slab + (16 * t + (lane >> 2)) * PSTG_LD + (lane & 3) * 8);
            b.q[2 * t + 1] = *(const f32x4*)(slab + (16 * t + (lane >> 2)) * PSTG_LD + (lane & 3) * 8 + 4);
        }
    }
}

// epilogue arithmetic + stores of one slice whose block has been read back into b
template <int EPI>
__device__ __forceinline__ void pp_finish(const QstGemmArgs& g, const PpRsrc& rs, const PpBlk& b, const float* bias_s,
                                          int mw, int nw, int i, int j, int lane, const PpPf& pf, const DropCtx& dc) {
    const int m0 = mw + i * 32, n0 = nw + j * 32;
    if constexpr (pp_f32_out<EPI>()) {
        const int c4 = lane & 7;
        const uint32_t so = ((uint32_t)m0 * (uint32_t)g.ldc + (uint32_t)n0) * 4u;
        const bool cok = n0 + c4 * 4 < g.N;
        f32x4 bb = {0.f, 0.f, 0.f, 0.f};
        if (g.bias) bb = *(const f32x4*)(bias_s + j * 32 + c4 * 4);
#pragma unroll
        for (int t = 0; t < 4; ++t) {
            const int row = 8 * t + (lane >> 3);
            const bool ok = cok && m0 + row < g.M;
            f32x4 v = b.q[t] + bb;
            if (dc.thr) {
                const uint32_t e = (uint32_t)(m0 + row) * (uint32_t)g.N + (uint32_t)(n0 + c4 * 4);
                float k0, k1, k2, k3;
                drop_pair(dc, e, k0, k1);
                drop_pair(dc, e + 2, k2, k3);
                v[0] *= k0; v[1] *= k1; v[2] *= k2; v[3] *= k3;
            }
            v += __builtin_bit_cast(f32x4, pf.a[t]);
            const uint32_t vo = ok ? (uint32_t)row * g.ldc * 4u + c4 * 16u : kPpOOB;
            __builtin_amdgcn_raw_buffer_store_b128(__builtin_bit_cast(u32x4, v), rs.c, (int)vo, (int)so, 0);
            if constexpr (EPI == QST_EPI_F32_RESID_BF16) {
                u32x2 pk; pk[0] = pack_bf16x2(v[0], v[1]); pk[1] = pack_bf16x2(v[2], v[3]);
                const uint32_t vo2 = ok ? (uint32_t)row * g.ldc * 2u + c4 * 8u : kPpOOB;
                __builtin_amdgcn_raw_buffer_store_b64(pk, rs.c2, (int)vo2, (int)(so >> 1), 0);
            }
        }
    } else {
        const int c8 = lane & 3;
        const uint32_t so = ((uint32_t)m0 * (uint32_t)g.ldc + (uint32_t)n0) * 2u;
        const bool cok = n0 + c8 * 8 < g.N;                     // N % 8 == 0 (qst_gemm_nt_pp_ok)
        f32x4 blo = {0.f, 0.f, 0.f, 0.f}, bhi = blo;
        if (g.bias) { blo = *(const f32x4*)(bias_s + j * 32 + c8 * 8); bhi = *(const f32x4*)(bias_s + j * 32 + c8 * 8 + 4); }
#pragma unroll
        for (int t = 0; t < 2; ++t) {
            const int row = 16 * t + (lane >> 2);
            const uint32_t vo = (cok && m0 + row < g.M) ? (uint32_t)row * g.ldc * 2u + c8 * 16u : kPpOOB;
            float v[8];
            {
                const f32x4 lo = b.q[2 * t] + blo, hi = b.q[2 * t + 1] + bhi;
#pragma unroll
                for (int e = 0; e < 4; ++e) { v[e] = lo[e]; v[4 + e] = hi[e]; }
            }
            u32x4 pk;
            if constexpr (EPI == QST_EPI_BF16) {
#pragma unroll
                for (int e = 0; e < 4; ++e) pk[e] = pack_bf16x2(v[2 * e], v[2 * e + 1]);
            } else if constexpr (EPI == QST_EPI_GELU) {
                float gg[8], hh[8];                             // C = gelu'(u) (saved for backward), C2 = gelu(u)
                pp_gelu4(v, gg, hh);                            // (four at a time: eight in lockstep cost 14 spilled registers)
                pp_gelu4(v + 4, gg + 4, hh + 4);
                u32x4 pg;
#pragma unroll
                for (int e = 0; e < 4; ++e) { pg[e] = pack_bf16x2(gg[2 * e], gg[2 * e + 1]); pk[e] = pack_bf16x2(hh[2 * e], hh[2 * e + 1]); }
                __builtin_amdgcn_raw_buffer_store_b128(pg, rs.c, (int)vo, (int)so, 0);
            } else {                                            // QST_EPI_GELU_BWD: acc * gelu'(u)
#pragma unroll
                for (int e = 0; e < 4; ++e)
                    pk[e] = pack_bf16x2(v[2 * e] * bf16lo(pf.a[t][e]), v[2 * e + 1] * bf16hi(pf.a[t][e]));
            }
            if constexpr (EPI == QST_EPI_GELU) __builtin_amdgcn_raw_buffer_store_b128(pk, rs.c2, (int)vo, (int)so, 0);
            else __builtin_amdgcn_raw_buffer_store_b128(pk, rs.c, (int)vo, (int)so, 0);
        }
    }
}

// Diagnostic build (-DQST_PP_STAMP, tools/pp_stamp.py): every role sums, in shader cycles (s_memtime), where its time goes
// -- loader: DMA issue / vmcnt wait / barrier wait; MFMA group: K-loop stages / epilogue slices / barrier wait -- and one
// wave of each role writes its sums to g.colsum (unused by NT) as uint64 [workgroup][3 roles][4] when the kernel ends. The
// product build contains none of it. -DQST_PP_NOEPI drops the epilogue (K loop + DMA alone).
#ifdef QST_PP_STAMP
#define PP_T(var_) const unsigned long long var_ = __builtin_amdgcn_s_memtime()
#define PP_ACC(sum_, a_, b_) sum_ += (b_) - (a_)
#else
#define PP_T(var_) do {} while (0)
#define PP_ACC(sum_, a_, b_) do {} while (0)
#endif

template <int EPI>
__global__ __launch_bounds__(768, 3) void gemm_nt_pp_kernel(QstGemmArgs g) {
    extern __shared__ __attribute__((aligned(16))) char smem[];
#ifdef QST_PP_STAMP
    unsigned long long tk0 = 0, tk1 = 0, tk2 = 0, fs0 = 0, fs1 = 0, fs2 = 0;
    const unsigned long long tk_begin = __builtin_amdgcn_s_memtime();
#endif
    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    // Tile list of this workgroup: the tile ids of the launch are cut into 8 contiguous chunks, one per XCD (blocks b and
    // b + 8 share an XCD: speed only), and the W workgroups of an XCD walk their chunk with stride W -- the tiles in
    // flight on an XCD at any time are ~W consecutive ids, i.e. a few A row panels x all their n-tiles, shared through L2.
    const int ntn = (g.N + PBN - 1) / PBN, T = ((g.M + PBM - 1) / PBM) * ntn;
    const int W = (int)gridDim.x >> 3, x = blockIdx.x & 7, jw = blockIdx.x >> 3;
    const int q = T >> 3, r = T & 7;
    const int cnt = q + (x < r ? 1 : 0);
    const int start = x < r ? x * (q + 1) : r * (q + 1) + (x - r) * q;
    const int ntiles = jw < cnt ? (cnt - jw + W - 1) / W : 0;
    if (ntiles == 0) return;                                       // uniform over the workgroup
    const int nk = g.K / PBK;
    constexpr int E = 6;                                           // super-steps an epilogue is spread over: one slice each (K >= 384)
    const int total = ntiles * nk;

    if (wave >= 8) {
        // ---------------------------------------------------------------- loader wave: 4 A + 6 B DMA instructions per stage
        const int lw = wave - 8;
        uint32_t va[4], vb[6];
#pragma unroll
        for (int t = 0; t < 4; ++t) {                              // one instruction = 8 tile rows x 128 B
            const int row = (lw * 4 + t) * 8 + (lane >> 3);
            va[t] = (uint32_t)row * g.lda * 2u + (uint32_t)(((lane & 7) ^ ((row >> 1) & 7)) * 16);
        }
#pragma unroll
        for (int t = 0; t < 6; ++t) {
            const int row = (lw * 6 + t) * 8 + (lane >> 3);
            vb[t] = (uint32_t)row * g.ldb * 2u + (uint32_t)(((lane & 7) ^ ((row >> 1) & 7)) * 16);
        }
        int it = 0, ik = 0, slot = 0;
        const bf16* abase = nullptr; const bf16* bbase = nullptr;
        uint32_t abytes = 0, bbytes = 0;
        auto set_tile = [&](int t) {
            const int id = start + jw + t * W;
            const int m0 = (id / ntn) * PBM, n0 = (id % ntn) * PBN;
            abase = (const bf16*)g.A + (size_t)m0 * g.lda;
            bbase = (const bf16*)g.B + (size_t)n0 * g.ldb;
            abytes = (uint32_t)min(PBM, g.M - m0) * g.lda * 2u;    // rows past the matrix fail the range check: zero fill
            bbytes = (uint32_t)min(PBN, g.N - n0) * g.ldb * 2u;
        };
        set_tile(0);
        auto issue = [&]() {
            const __amdgpu_buffer_rsrc_t ra = make_rsrc(abase, abytes), rb = make_rsrc(bbase, bbytes);
            char* st = smem + slot * PSTAGE;
            const uint32_t ko = (uint32_t)ik * (PBK * 2);
#pragma unroll
            for (int t = 0; t < 4; ++t) pp_dma16(ra, st + (lw * 4 + t) * 1024, va[t], ko);
#pragma unroll
            for (int t = 0; t < 6; ++t) pp_dma16(rb, st + PA_BYTES + (lw * 6 + t) * 1024, vb[t], ko);
            slot = slot == PSLOTS - 1 ? 0 : slot + 1;
            if (++ik == nk) { ik = 0; if (++it < ntiles) set_tile(it); }
        };
        issue();
        if (total > 1) issue();
#pragma unroll 1
        for (int s = 0; s < total; ++s) {
            // stage s must have landed before this wave arrives at the barrier that hands it to the MFMA group
            PP_T(ta);
            if (s + 1 < total) asm volatile("s_waitcnt vmcnt(10)" ::: "memory");
            else asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
            PP_T(tb);
            __builtin_amdgcn_s_barrier();                          // ... and the group has finished stage s - 1
            PP_T(tc);
            if (s + 2 < total) issue();                            // into the slot of stage s - 1
            PP_T(td);
            PP_ACC(tk1, ta, tb); PP_ACC(tk2, tb, tc); PP_ACC(tk0, tc, td);
        }
#pragma unroll 1
        for (int s = 0; s < E; ++s) __builtin_amdgcn_s_barrier();  // drain round (the last tile's epilogue)
#ifdef QST_PP_STAMP
        if (tid == 512 && g.colsum) {
            unsigned long long* o = (unsigned long long*)g.colsum + (size_t)blockIdx.x * 12;
            o[0] = tk0; o[1] = tk1; o[2] = tk2; o[3] = __builtin_amdgcn_s_memtime() - tk_begin;
        }
#endif
        return;
    }

    // -------------------------------------------------------------------- MFMA waves
    const int grp = wave >> 2, wv = wave & 3, wm = wv >> 1, wn = wv & 1;
    const int fr = lane & 31, fh = lane >> 5;
    float* slabs = (float*)(smem + PRING) + wv * 2 * PSLAB;        // two slabs, shared with wave wv of the other group
    float* bias_s = (float*)(smem + PBIAS_OFF) + wave * 96;
    constexpr bool kF32Out = pp_f32_out<EPI>();
    DropCtx dc = DropCtx{0u, 0u, 1.f};                             // dropout of the projection output, before the residual
    if (kF32Out && g.drop_where == 1) dc = drop_ctx(g.drop);
    f32x16 acc[2][3];
    PpPf pf;
#pragma unroll
    for (int t = 0; t < 4; ++t) pf.a[t] = u32x4{0u, 0u, 0u, 0u};
    float bv0 = 0.f, bv1 = 0.f;
    int slot = 0, mw = 0, nw = 0;
    // element sizes: C fp32 (F32 epilogues) or bf16; C2 bf16; the prefetched input is aux (bf16, ld = ldc) or resid (fp32)
    PpRsrc rs;
    {
        const uint32_t cbytes = (uint32_t)g.M * (uint32_t)g.ldc * (kF32Out ? 4u : 2u);
        rs.c = make_rsrc(g.C, cbytes);
        rs.c2 = make_rsrc(g.C2 ? g.C2 : g.C, (uint32_t)g.M * (uint32_t)g.ldc * 2u);
        if (EPI == QST_EPI_GELU_BWD) rs.in = make_rsrc(g.aux, (uint32_t)g.M * (uint32_t)g.ldc * 2u);
        else rs.in = make_rsrc(g.resid ? (const void*)g.resid : g.C, (uint32_t)g.M * (uint32_t)g.ldr * 4u);
    }
    const int rowa = wm * 64 + fr, rowb = wn * 96 + fr;            // this lane's fragment rows in the A / B stage images

    // the accumulator block of slice s (0..5 = (i, j) row-major) into a slab
#define PP_PUT(s_, slab_)                                               \
    switch (s_) {                                                       \
        case 0: pp_put(acc[0][0], slab_, lane); break;                  \
        case 1: pp_put(acc[0][1], slab_, lane); break;                  \
        case 2: pp_put(acc[0][2], slab_, lane); break;                  \
        case 3: pp_put(acc[1][0], slab_, lane); break;                  \
        case 4: pp_put(acc[1][1], slab_, lane); break;                  \
        default: pp_put(acc[1][2], slab_, lane); break;                 \
    }

#pragma unroll 1
    for (int t = 0; t <= ntiles; ++t) {
        const bool comp = t < ntiles && (t & 1) == grp;
        const bool epi = t >= 1 && ((t - 1) & 1) == grp;           // this group computed tile t - 1: (mw, nw) still name it
        const int steps = t < ntiles ? nk : E;
        if (comp) {
            const int id = start + jw + t * W;
            mw = (id / ntn) * PBM + wm * 64;
            nw = (id % ntn) * PBN + wn * 96;
#pragma unroll
            for (int i = 0; i < 2; ++i)
#pragma unroll
                for (int j = 0; j < 3; ++j)
#pragma unroll
                    for (int rr = 0; rr < 16; ++rr) acc[i][j][rr] = 0.f;
        }
#pragma unroll 1
        for (int kt = 0; kt < steps; ++kt) {
            PP_T(ta);
            __builtin_amdgcn_s_barrier();
            PP_T(tb);
            PP_ACC(tk2, ta, tb);
            if (comp) {
#ifndef QST_PP_NOCOMP
#if !defined(QST_PP_PRIO) || QST_PP_PRIO == 2
                __builtin_amdgcn_s_setprio(2);                     // this wave feeds the SIMD's matrix pipe: its LDS reads and MFMAs first
#endif
                const char* pa = smem + slot * PSTAGE;
                const char* pb = pa + PA_BYTES;
                bf16x8 fa[2][2], fb[2][3];                         // two fragment sets: k-step ks + 1 is read while ks multiplies
#define PP_LOAD(ks_, set_)                                                                                          \
    do {                                                                                                            \
        _Pragma("unroll") for (int i = 0; i < 2; ++i) fa[set_][i] = *(const bf16x8*)(pa + pp_off(rowa + i * 32, (ks_) * 2 + fh)); \
        _Pragma("unroll") for (int j = 0; j < 3; ++j) fb[set_][j] = *(const bf16x8*)(pb + pp_off(rowb + j * 32, (ks_) * 2 + fh)); \
    } while (0)
#define PP_MFMA(set_)                                                                                               \
    do {                                                                                                            \
        _Pragma("unroll") for (int i = 0; i < 2; ++i)                                                               \
            _Pragma("unroll") for (int j = 0; j < 3; ++j)                                                           \
                acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(fb[set_][j], fa[set_][i], acc[i][j], 0, 0, 0);   /* D rows = n */ \
    } while (0)
                PP_LOAD(0, 0);
                PP_LOAD(1, 1);
                PP_MFMA(0);
                PP_LOAD(2, 0);
                PP_MFMA(1);
                PP_LOAD(3, 1);
                PP_MFMA(0);
                PP_MFMA(1);
                // issue order: the 5 reads of k-step 0, then the reads of k-steps 1..3 one behind each of the first five
                // MFMAs of the k-step before, then the last 6 MFMAs
                __builtin_amdgcn_sched_group_barrier(0x100, 5, 0);
#pragma unroll
                for (int rep = 0; rep < 3; ++rep) {
#pragma unroll
                    for (int u = 0; u < 5; ++u) {
                        __builtin_amdgcn_sched_group_barrier(0x100, 1, 0);
                        __builtin_amdgcn_sched_group_barrier(0x008, 1, 0);
                    }
                    __builtin_amdgcn_sched_group_barrier(0x008, 1, 0);
                }
                __builtin_amdgcn_sched_group_barrier(0x008, 6, 0);
#undef PP_LOAD
#undef PP_MFMA
#if !defined(QST_PP_PRIO) || QST_PP_PRIO == 2
                __builtin_amdgcn_s_setprio(0);
#endif
#endif
                if (kt == 0 && g.bias) {                           // this wave's 96 bias values travel during the K loop
                    const int n = nw + lane;
                    bv0 = n < g.N ? g.bias[n] : 0.f;
                    bv1 = (lane < 32 && n + 64 < g.N) ? g.bias[n + 64] : 0.f;
                }
                if (kt == nk - 1) {
                    if (g.bias) { bias_s[lane] = bv0; if (lane < 32) bias_s[64 + lane] = bv1; }
                    pp_prefetch<EPI>(g, rs, mw, nw, 0, 0, lane, pf);   // slice 0's inputs, one super-step early
                    // block 0 goes into slab 0 now (the other group, in its last epilogue step, reads slab 1)
                    pp_put(acc[0][0], slabs, lane);
                }
#ifdef QST_PP_STAMP
                asm volatile("s_nop 0" :: "v"(acc[0][0][0]), "v"(acc[1][2][15]));      // the stage's MFMAs have been issued
                PP_T(tc); PP_ACC(tk0, tb, tc);
#endif
            } else if (epi && kt < 6) {
#ifndef QST_PP_NOEPI
                // slice kt of tile t - 1: read block kt back from the slab it went into one super-step ago, send block kt + 1
                // into the other slab, finish and store block kt. One copy of the code per block (a runtime-indexed
                // accumulator would live in scratch).
#define PP_STEP(S_)                                                                                                 \
    case S_: {                                                                                                      \
        PpBlk b;                                                                                                    \
        pp_get<EPI>(slabs + ((S_) & 1) * PSLAB, lane, b);                                                           \
        if ((S_) + 1 < 6) pp_put(acc[((S_) + 1) / 3][((S_) + 1) % 3], slabs + (((S_) + 1) & 1) * PSLAB, lane);      \
        pp_finish<EPI>(g, rs, b, bias_s, mw, nw, (S_) / 3, (S_) % 3, lane, pf, dc);                                 \
        if ((S_) + 1 < 6) pp_prefetch<EPI>(g, rs, mw, nw, ((S_) + 1) / 3, ((S_) + 1) % 3, lane, pf);                \
    } break;
                switch (kt) { PP_STEP(0) PP_STEP(1) PP_STEP(2) PP_STEP(3) PP_STEP(4) PP_STEP(5) }
#undef PP_STEP
#endif
                PP_T(tc); PP_ACC(tk1, tb, tc);
            }
            if (t < ntiles) slot = slot == PSLOTS - 1 ? 0 : slot + 1;
        }
    }
#undef PP_PUT
#ifdef QST_PP_STAMP
    if ((tid == 0 || tid == 256) && g.colsum) {
        unsigned long long* o = (unsigned long long*)g.colsum + (size_t)blockIdx.x * 12 + (tid == 0 ? 4 : 8);
        o[0] = tk0; o[1] = tk1; o[2] = tk2; o[3] = __builtin_amdgcn_s_memtime() - tk_begin;
#if QST_PP_STAMP == 2
        o[0] = fs0; o[2] = fs1; o[3] = fs2;      // get + put issue | LDS wait | arithmetic + stores (o[1] = whole slices)
#endif
    }
#endif
}

int g_cus = 0;

template <int EPI>
int launch_pp(const QstGemmArgs* a, hipStream_t st) {
    static QstLdsAttr attr;
    if (int rc = qst_ensure_lds(attr, (const void*)gemm_nt_pp_kernel<EPI>, PLDS)) return rc;
    if (g_cus == 0) {
        int dev = 0, n = 0;
        QST_HIP_CHECK(hipGetDevice(&dev));
        QST_HIP_CHECK(hipDeviceGetAttribute(&n, hipDeviceAttributeMultiprocessorCount, dev));
        g_cus = n > 0 ? n / 8 * 8 : 256;
        if (g_cus < 8) g_cus = 8;
    }
    const int T = ((a->M + PBM - 1) / PBM) * ((a->N + PBN - 1) / PBN);
    const int grid = min(g_cus, (T + 7) / 8 * 8);                 // one persistent workgroup per CU, a multiple of 8
    gemm_nt_pp_kernel<EPI><<<dim3(grid), dim3(768), PLDS, st>>>(*a);
    QST_LAUNCH_CHECK();
    return QST_OK;
}

}  // namespace

// shapes / options this form takes (the caller falls back to the tiled kernels otherwise)
extern "C" int qst_gemm_nt_pp_ok(const QstGemmArgs* a, int epi) {
    if (!a) return 0;
    if (a->a_head_L || a->c_head_L) return 0;
    if (a->K < 6 * PBK) return 0;                                  // an epilogue takes six K stages
    if (a->K % PBK != 0 || a->lda % 8 != 0 || a->ldb % 8 != 0 || a->N % 8 != 0 || a->ldc % 8 != 0) return 0;
    if (a->resid && a->ldr % 4 != 0) return 0;
    // epilogue accesses are raw buffer operations with 32-bit offsets
    if ((int64_t)a->M * a->ldc * 4 >= 0x7FFFFF00LL || (int64_t)a->M * (a->ldr > 0 ? a->ldr : 1) * 4 >= 0x7FFFFF00LL) return 0;
    if (a->drop.thr16 && a->drop.state && (int64_t)a->M * a->N >= ((int64_t)1 << 32)) return 0;
    if ((int64_t)PBM * a->lda * 2 >= 0x7FFFFF00LL || (int64_t)PBN * a->ldb * 2 >= 0x7FFFFF00LL) return 0;
    switch (epi) {
        case QST_EPI_BF16: case QST_EPI_GELU: case QST_EPI_GELU_BWD: return 1;
        // (the fp32-output epilogues are written -- pp_prefetch / pp_finish -- but need 16 more registers for the residual
        //  rows than three waves per SIMD leave: they spill, and stay with the tiled kernels)
        default: return 0;
    }
}

extern "C" int qst_gemm_nt_pp(const QstGemmArgs* a, int epi, void* stream) {
    if (!qst_gemm_nt_pp_ok(a, epi)) return QST_ERR_UNSUPPORTED;
    hipStream_t st = (hipStream_t)stream;
    switch (epi) {
        case QST_EPI_BF16: return launch_pp<QST_EPI_BF16>(a, st);
        case QST_EPI_GELU: return launch_pp<QST_EPI_GELU>(a, st);
        case QST_EPI_GELU_BWD: return launch_pp<QST_EPI_GELU_BWD>(a, st);
        case QST_EPI_F32_RESID: return launch_pp<QST_EPI_F32_RESID>(a, st);
        case QST_EPI_F32_RESID_BF16: return launch_pp<QST_EPI_F32_RESID_BF16>(a, st);
        default: return QST_ERR_BAD_ARG;
    }
}
