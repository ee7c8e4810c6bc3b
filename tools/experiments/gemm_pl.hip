// gemm_pl.hip -- persistent, loader-fed NT GEMM for the large launches of the step (gfx950, wave64).
//
//   C[M,N] = A[M,K] . B[N,K]^T (+ fused epilogue), the same contract as gemm_nt_kernel in gemm.hip (nn.Linear forward and
//   dgrad inside BertLayer: transformers modeling_bert.py:154-156, 282-293, 325-351).
//
// What it changes against the tiled kernels of gemm.hip (round-3 measurements):
//   * Four LOADER waves issue every LDS-DMA. In the tiled kernels each MFMA wave issues its share of a stage's DMAs (10
//     instructions at ~63 cycles each, in order with its MFMAs): a workgroup's six-stage K loop took 3.6x the time of its
//     MFMAs. Here the eight MFMA waves (4 x 2, 64 x 96 each: a 256 x 192 tile) only read fragments and multiply.
//   * The fragments of k-step ks+1 are read into a second register set while the MFMAs of ks issue (sched_group_barrier).
//   * One persistent workgroup per CU walks its tiles; a tile's first stage is already in the ring when the MFMA waves come
//     back from writing out the previous tile (the loaders run one stage ahead across tile boundaries).
//   * The epilogue goes from registers to memory: v_permlane32_swap pairs the two half-waves so that every lane holds eight
//     consecutive columns of one row (16-byte stores of 32 rows x 32 B; measured 5.4 TB/s for that shape against 6.0 for whole
//     lines, tools/probe/store_pattern_probe.hip) -- no LDS staging, no LDS round trip, all 160 KB free for the ring. (An LDS
//     slab per wave, as gemm.hip stages its tiles, returned stale data on this kernel when the two waves of a SIMD staged at
//     the same time -- tools/experiments/gemm_pp_c2_variant.patch; not understood, avoided.)
// One s_barrier per stage synchronises the 12 waves: a loader waits (vmcnt) for the stage about to be read, then issues the
// next one into the slot the MFMA waves have just left. Ring: 2 slots x 56 KB (a third does not fit beside them).
#include "qst_common.h"
#include "qst_kernels.h"

namespace {

typedef __attribute__((address_space(3))) void lds_void;

__device__ __forceinline__ void pl_dma16(__amdgpu_buffer_rsrc_t r, char* lds_wave_base, uint32_t voff, uint32_t soff) {
    __builtin_amdgcn_raw_ptr_buffer_load_lds(r, (lds_void*)lds_wave_base, 16, (int)voff, (int)soff, 0, 0);
}

constexpr int LBM = 256, LBN = 192, LBK = 64;
constexpr int LA_BYTES = LBM * LBK * 2;              // 32 KB
constexpr int LB_BYTES = LBN * LBK * 2;              // 24 KB
constexpr int LSTAGE = LA_BYTES + LB_BYTES;          // 56 KB
constexpr int LRING = 2 * LSTAGE;                    // 112 KB
constexpr int LLDS = LRING + 8 * 96 * 4;             // + 96 bias values per MFMA wave

// 16-byte chunk c of row r of a [rows][64 bf16] operand image sits at chunk position c ^ ((r >> 1) & 7) (as gemm.hip)
__device__ __forceinline__ uint32_t pl_off(int row, int chunk) {
    return (uint32_t)(row * 128 + ((chunk ^ ((row >> 1) & 7)) << 4));
}

template <int EPI> constexpr bool pl_f32_out() { return EPI == QST_EPI_F32_RESID || EPI == QST_EPI_F32_RESID_BF16; }

constexpr uint32_t kPlOOB = 0x7FFFFFF0u;             // voffset that always fails the buffer range check (loads 0, stores nothing)

// Buffer descriptors of the epilogue operands (wave-uniform: SGPRs): per-lane 32-bit offsets + the block's origin as the
// scalar offset; rows / columns outside the matrix get kPlOOB instead of a branch.
struct PlRsrc { __amdgpu_buffer_rsrc_t c, c2, in; };

// erf GELU and its derivative for four elements in lockstep (independent instructions cover each other's latency; hipcc
// otherwise emits one dependent chain per element). Abramowitz-Stegun 7.1.26 on v_rcp_f32 / v_exp_f32, operation for
// operation the arithmetic of gelu_parts2 (qst_common.h):
//   z = |x| / sqrt2, t = 1 / (1 + p z), erf(z) = 1 - (a1 t + ... + a5 t^5) exp(-z^2);  h = x Phi(x),  g = Phi(x) + x phi(x).
#define PL_PHASE(expr_) do { _Pragma("unroll") for (int k = 0; k < 4; ++k) { expr_; } __builtin_amdgcn_sched_barrier(0); } while (0)
__device__ __forceinline__ void pl_gelu4(const float* x, float* g, float* h) {
    float z[4], t[4], e[4], p[4];
    __builtin_amdgcn_sched_barrier(0);
    PL_PHASE(z[k] = fabsf(x[k]) * 0.70710678118654752f);
    PL_PHASE(t[k] = __builtin_fmaf(z[k], 0.3275911f, 1.0f));
    PL_PHASE(e[k] = z[k] * z[k]);
    PL_PHASE(t[k] = __builtin_amdgcn_rcpf(t[k]));
    PL_PHASE(e[k] = e[k] * -1.4426950408889634f);
    PL_PHASE(e[k] = __builtin_amdgcn_exp2f(e[k]));                       // exp(-x^2 / 2)
    PL_PHASE(p[k] = __builtin_fmaf(t[k], 1.061405429f, -1.453152027f));
    PL_PHASE(p[k] = __builtin_fmaf(p[k], t[k], 1.421413741f));
    PL_PHASE(p[k] = __builtin_fmaf(p[k], t[k], -0.284496736f));
    PL_PHASE(p[k] = __builtin_fmaf(p[k], t[k], 0.254829592f));
    PL_PHASE(p[k] = p[k] * t[k]);
    PL_PHASE(p[k] = p[k] * e[k]);
    PL_PHASE(p[k] = __builtin_fmaf(p[k], -0.5f, 0.5f));                   // 0.5 erf(|x| / sqrt2)
    PL_PHASE(p[k] = __builtin_copysignf(p[k], x[k]));
    PL_PHASE(p[k] = p[k] + 0.5f);                                         // Phi(x)
    PL_PHASE(e[k] = e[k] * 0.39894228040143268f);                         // phi(x)
    PL_PHASE(g[k] = __builtin_fmaf(x[k], e[k], p[k]));
    PL_PHASE(h[k] = x[k] * p[k]);
}
#undef PL_PHASE

// Half p (columns 16 p .. 16 p + 15) of a 32 x 32 accumulator block as rows: the block has D rows = n, D column = m on the
// lane -- lane (m, h = lane / 32) holds, per register group g4, the four columns 8 g4 + 4 h + 0..3 of row m. Swapping the
// upper half-wave of group 2p with the lower half-wave of group 2p + 1 (v_permlane32_swap) leaves lane (m, h)
// with the EIGHT consecutive columns 16 p + 8 h + 0..7 of row m.
__device__ __forceinline__ void pl_rows8(const f32x16& blk, int p, float (&v)[8]) {
    // (written with swap32 -- v_permlane32_swap of a register with itself -- and a select: ROCm 7.2's hipcc folds the second
    //  result of __builtin_amdgcn_permlane32_swap(a, b) with a != b into the first, seen in the emitted code)
    const bool up = (threadIdx.x & 32) != 0;
#pragma unroll
    for (int e = 0; e < 4; ++e) {
        const float x = p ? blk[8 + e] : blk[e], y = p ? blk[12 + e] : blk[4 + e];
        const float sx = swap32(x), sy = swap32(y);              // the other half-wave's value of the same row
        v[e] = up ? sy : x;                                      // h = 0: columns 16p + e (own);      h = 1: 16p + 8 + e  (from h = 0)
        v[4 + e] = up ? y : sx;                                  // h = 0: columns 16p + 4 + e (h = 1); h = 1: 16p + 12 + e (own)
    }
}

// bf16-output epilogues of block (i, j) of the wave's 64 x 96 sub-tile at C[mw][nw]: two 16-byte stores per lane.
// aux[p]: the lane's eight gelu'(u) values of half p (QST_EPI_GELU_BWD), loaded by pl_aux one block ahead.
template <int EPI>
__device__ __forceinline__ void pl_aux(const QstGemmArgs& g, const PlRsrc& rs, int mw, int nw, int i, int j, int lane, u32x4 (&aux)[2]) {
    const int m0 = mw + i * 32, n0 = nw + j * 32, row = lane & 31, fh = lane >> 5;
    const uint32_t so = ((uint32_t)m0 * (uint32_t)g.ldc + (uint32_t)n0) * 2u;
#pragma unroll
    for (int p = 0; p < 2; ++p) {
        const bool ok = m0 + row < g.M && n0 + 16 * p + 8 * fh < g.N;
        const uint32_t vo = ok ? (uint32_t)row * g.ldc * 2u + (uint32_t)(16 * p + 8 * fh) * 2u : kPlOOB;
        aux[p] = __builtin_amdgcn_raw_buffer_load_b128(rs.in, (int)vo, (int)so, 0);
    }
}

template <int EPI>
__device__ __forceinline__ void pl_block_bf16(const QstGemmArgs& g, const PlRsrc& rs, const f32x16& blk, const float* bias_s,
                                              int mw, int nw, int i, int j, int lane, const u32x4 (&aux)[2]) {
    const int m0 = mw + i * 32, n0 = nw + j * 32, row = lane & 31, fh = lane >> 5;
    const uint32_t so = ((uint32_t)m0 * (uint32_t)g.ldc + (uint32_t)n0) * 2u;
#pragma unroll
    for (int p = 0; p < 2; ++p) {
        float v[8];
        pl_rows8(blk, p, v);
        const int c = 16 * p + 8 * fh;                           // first of this lane's 8 columns inside the block
        const bool ok = m0 + row < g.M && n0 + c < g.N;          // N % 8 == 0 (qst_gemm_nt_pl_ok)
        const uint32_t vo = ok ? (uint32_t)row * g.ldc * 2u + (uint32_t)c * 2u : kPlOOB;
        if (g.bias) {
            const f32x4 lo = *(const f32x4*)(bias_s + j * 32 + c), hi = *(const f32x4*)(bias_s + j * 32 + c + 4);
#pragma unroll
            for (int e = 0; e < 4; ++e) { v[e] += lo[e]; v[4 + e] += hi[e]; }
        }
        u32x4 pk;
        if constexpr (EPI == QST_EPI_BF16) {
#pragma unroll
            for (int e = 0; e < 4; ++e) pk[e] = pack_bf16x2(v[2 * e], v[2 * e + 1]);
        } else if constexpr (EPI == QST_EPI_GELU) {
            float gg[8], hh[8];                                  // C = gelu'(u) (saved for backward), C2 = gelu(u)
            pl_gelu4(v, gg, hh);
            pl_gelu4(v + 4, gg + 4, hh + 4);
            u32x4 pg;
#pragma unroll
            for (int e = 0; e < 4; ++e) { pg[e] = pack_bf16x2(gg[2 * e], gg[2 * e + 1]); pk[e] = pack_bf16x2(hh[2 * e], hh[2 * e + 1]); }
            __builtin_amdgcn_raw_buffer_store_b128(pg, rs.c, (int)vo, (int)so, 0);
        } else {                                                 // QST_EPI_GELU_BWD: acc * gelu'(u)
#pragma unroll
            for (int e = 0; e < 4; ++e)
                pk[e] = pack_bf16x2(v[2 * e] * bf16lo(aux[p][e]), v[2 * e + 1] * bf16hi(aux[p][e]));
        }
        if constexpr (EPI == QST_EPI_GELU) __builtin_amdgcn_raw_buffer_store_b128(pk, rs.c2, (int)vo, (int)so, 0);
        else __builtin_amdgcn_raw_buffer_store_b128(pk, rs.c, (int)vo, (int)so, 0);
    }
}

// fp32-output epilogues: the accumulator layout already gives every lane four consecutive columns per register group
// (16-byte fp32 accesses, the two half-waves side by side: 32 rows x 32 B per instruction, as above).
// res[g4]: the lane's four residual values of group g4, loaded by pl_resid one block ahead.
__device__ __forceinline__ void pl_resid(const QstGemmArgs& g, const PlRsrc& rs, int mw, int nw, int i, int j, int lane, u32x4 (&res)[4]) {
    const int m0 = mw + i * 32, n0 = nw + j * 32, row = lane & 31, fh = lane >> 5;
    const uint32_t so = ((uint32_t)m0 * (uint32_t)g.ldr + (uint32_t)n0) * 4u;
#pragma unroll
    for (int g4 = 0; g4 < 4; ++g4) {
        const bool ok = g.resid != nullptr && m0 + row < g.M && n0 + 8 * g4 + 4 * fh < g.N;
        const uint32_t vo = ok ? (uint32_t)row * g.ldr * 4u + (uint32_t)(8 * g4 + 4 * fh) * 4u : kPlOOB;
        res[g4] = __builtin_amdgcn_raw_buffer_load_b128(rs.in, (int)vo, (int)so, 0);
    }
}

template <int EPI>
__device__ __forceinline__ void pl_block_f32(const QstGemmArgs& g, const PlRsrc& rs, const f32x16& blk, const float* bias_s,
                                             int mw, int nw, int i, int j, int lane, const u32x4 (&res)[4], const DropCtx& dc) {
    const int m0 = mw + i * 32, n0 = nw + j * 32, row = lane & 31, fh = lane >> 5;
    const uint32_t so = ((uint32_t)m0 * (uint32_t)g.ldc + (uint32_t)n0) * 4u;
    f32x16 out;
#pragma unroll
    for (int g4 = 0; g4 < 4; ++g4) {
        const int c = 8 * g4 + 4 * fh;
        const bool ok = m0 + row < g.M && n0 + c < g.N;
        f32x4 v;
#pragma unroll
        for (int e = 0; e < 4; ++e) v[e] = blk[4 * g4 + e];
        if (g.bias) v += *(const f32x4*)(bias_s + j * 32 + c);
        if (dc.thr) {
            const uint32_t e0 = (uint32_t)(m0 + row) * (uint32_t)g.N + (uint32_t)(n0 + c);
            float k0, k1, k2, k3;
            drop_pair(dc, e0, k0, k1);
            drop_pair(dc, e0 + 2, k2, k3);
            v[0] *= k0; v[1] *= k1; v[2] *= k2; v[3] *= k3;
        }
        v += __builtin_bit_cast(f32x4, res[g4]);
        const uint32_t vo = ok ? (uint32_t)row * g.ldc * 4u + (uint32_t)c * 4u : kPlOOB;
        __builtin_amdgcn_raw_buffer_store_b128(__builtin_bit_cast(u32x4, v), rs.c, (int)vo, (int)so, 0);
#pragma unroll
        for (int e = 0; e < 4; ++e) out[4 * g4 + e] = v[e];
    }
    if constexpr (EPI == QST_EPI_F32_RESID_BF16) {
        // the bf16 copy as 16-byte stores of eight consecutive columns (the half-wave swap of pl_rows8 on the finished values)
#pragma unroll
        for (int p = 0; p < 2; ++p) {
            float v[8];
            pl_rows8(out, p, v);
            const int c = 16 * p + 8 * fh;
            const bool ok = m0 + row < g.M && n0 + c < g.N;
            const uint32_t vo = ok ? (uint32_t)row * g.ldc * 2u + (uint32_t)c * 2u : kPlOOB;
            u32x4 pk;
#pragma unroll
            for (int e = 0; e < 4; ++e) pk[e] = pack_bf16x2(v[2 * e], v[2 * e + 1]);
            __builtin_amdgcn_raw_buffer_store_b128(pk, rs.c2, (int)vo, (int)(so >> 1), 0);
        }
    }
}

// Diagnostic builds (tools/experiments): -DQST_PL_NOEPI keeps the K loop alone (the accumulators stay live), -DQST_PL_NOCOMP
// drops the MFMAs.
template <int EPI>
__global__ __launch_bounds__(768, 3) void gemm_nt_pl_kernel(QstGemmArgs g) {
    extern __shared__ __attribute__((aligned(16))) char smem[];
    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    // Tile list of this workgroup: the tile ids of the launch are cut into 8 contiguous chunks, one per XCD (blocks b and
    // b + 8 share an XCD: speed only), and the W workgroups of an XCD walk their chunk with stride W -- the tiles in
    // flight on an XCD at any time are ~W consecutive ids, i.e. a few A row panels x all their n-tiles, shared through L2.
    const int ntn = (g.N + LBN - 1) / LBN, T = ((g.M + LBM - 1) / LBM) * ntn;
    const int W = (int)gridDim.x >> 3, x = blockIdx.x & 7, jw = blockIdx.x >> 3;
    const int q = T >> 3, r = T & 7;
    const int cnt = q + (x < r ? 1 : 0);
    const int start = x < r ? x * (q + 1) : r * (q + 1) + (x - r) * q;
    const int ntiles = jw < cnt ? (cnt - jw + W - 1) / W : 0;
    if (ntiles == 0) return;                                       // uniform over the workgroup
    const int nk = g.K / LBK;
    const int total = ntiles * nk;

    if (wave >= 8) {
        // ---------------------------------------------------------------- loader wave: 8 A + 6 B DMA instructions per stage
        const int lw = wave - 8;
        uint32_t va[8], vb[6];
#pragma unroll
        for (int t = 0; t < 8; ++t) {                              // one instruction = 8 tile rows x 128 B
            const int row = (lw * 8 + t) * 8 + (lane >> 3);
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
            const int m0 = (id / ntn) * LBM, n0 = (id % ntn) * LBN;
            abase = (const bf16*)g.A + (size_t)m0 * g.lda;
            bbase = (const bf16*)g.B + (size_t)n0 * g.ldb;
            abytes = (uint32_t)min(LBM, g.M - m0) * g.lda * 2u;    // rows past the matrix fail the range check: zero fill
            bbytes = (uint32_t)min(LBN, g.N - n0) * g.ldb * 2u;
        };
        set_tile(0);
        auto issue = [&]() {
            const __amdgpu_buffer_rsrc_t ra = make_rsrc(abase, abytes), rb = make_rsrc(bbase, bbytes);
            char* st = smem + slot * LSTAGE;
            const uint32_t ko = (uint32_t)ik * (LBK * 2);
#pragma unroll
            for (int t = 0; t < 8; ++t) pl_dma16(ra, st + (lw * 8 + t) * 1024, va[t], ko);
#pragma unroll
            for (int t = 0; t < 6; ++t) pl_dma16(rb, st + LA_BYTES + (lw * 6 + t) * 1024, vb[t], ko);
            slot ^= 1;
            if (++ik == nk) { ik = 0; if (++it < ntiles) set_tile(it); }
        };
        issue();
#pragma unroll 1
        for (int s = 0; s < total; ++s) {
            asm volatile("s_waitcnt vmcnt(0)" ::: "memory");       // stage s has landed ...
            __builtin_amdgcn_s_barrier();                          // ... and the MFMA waves have finished stage s - 1
            if (s + 1 < total) issue();                            // into the slot of stage s - 1 (a tile's first stage travels
        }                                                          // while the MFMA waves write out the tile before it)
        return;
    }

    // -------------------------------------------------------------------- MFMA waves: 4 (M) x 2 (N), 64 x 96 each
    const int wm = wave >> 1, wn = wave & 1;
    const int fr = lane & 31, fh = lane >> 5;
    float* bias_s = (float*)(smem + LRING) + wave * 96;
    constexpr bool kF32Out = pl_f32_out<EPI>();
    DropCtx dc = DropCtx{0u, 0u, 1.f};                             // dropout of the projection output, before the residual
    if (kF32Out && g.drop_where == 1) dc = drop_ctx(g.drop);
    f32x16 acc[2][3];
    // element sizes: C fp32 (F32 epilogues) or bf16; C2 bf16; the prefetched input is aux (bf16, ld = ldc) or resid (fp32)
    PlRsrc rs;
    {
        const uint32_t cbytes = (uint32_t)g.M * (uint32_t)g.ldc * (kF32Out ? 4u : 2u);
        rs.c = make_rsrc(g.C, cbytes);
        rs.c2 = make_rsrc(g.C2 ? g.C2 : g.C, (uint32_t)g.M * (uint32_t)g.ldc * 2u);
        if (EPI == QST_EPI_GELU_BWD) rs.in = make_rsrc(g.aux, (uint32_t)g.M * (uint32_t)g.ldc * 2u);
        else rs.in = make_rsrc(g.resid ? (const void*)g.resid : g.C, (uint32_t)g.M * (uint32_t)g.ldr * 4u);
    }
    const int rowa = wm * 64 + fr, rowb = wn * 96 + fr;            // this lane's fragment rows in the A / B stage images
    int slot = 0;

#pragma unroll 1
    for (int t = 0; t < ntiles; ++t) {
        const int id = start + jw + t * W;
        const int mw = (id / ntn) * LBM + wm * 64, nw = (id % ntn) * LBN + wn * 96;
#pragma unroll
        for (int i = 0; i < 2; ++i)
#pragma unroll
            for (int j = 0; j < 3; ++j)
#pragma unroll
                for (int rr = 0; rr < 16; ++rr) acc[i][j][rr] = 0.f;
        float bv0 = 0.f, bv1 = 0.f;
        if (g.bias) {                                              // this wave's 96 bias values travel during the K loop
            const int n = nw + lane;
            bv0 = n < g.N ? g.bias[n] : 0.f;
            bv1 = (lane < 32 && n + 64 < g.N) ? g.bias[n + 64] : 0.f;
        }
#pragma unroll 1
        for (int kt = 0; kt < nk; ++kt) {
            __builtin_amdgcn_s_barrier();
#ifndef QST_PL_NOCOMP
            const char* pa = smem + slot * LSTAGE;
            const char* pb = pa + LA_BYTES;
            bf16x8 fa[2][2], fb[2][3];                             // two fragment sets: k-step ks + 1 is read while ks multiplies
#define PL_LOAD(ks_, set_)                                                                                          \
    do {                                                                                                            \
        _Pragma("unroll") for (int i = 0; i < 2; ++i) fa[set_][i] = *(const bf16x8*)(pa + pl_off(rowa + i * 32, (ks_) * 2 + fh)); \
        _Pragma("unroll") for (int j = 0; j < 3; ++j) fb[set_][j] = *(const bf16x8*)(pb + pl_off(rowb + j * 32, (ks_) * 2 + fh)); \
    } while (0)
#define PL_MFMA(set_)                                                                                               \
    do {                                                                                                            \
        _Pragma("unroll") for (int i = 0; i < 2; ++i)                                                               \
            _Pragma("unroll") for (int j = 0; j < 3; ++j)                                                           \
                acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(fb[set_][j], fa[set_][i], acc[i][j], 0, 0, 0);   /* D rows = n */ \
    } while (0)
            PL_LOAD(0, 0);
            PL_LOAD(1, 1);
            PL_MFMA(0);
            PL_LOAD(2, 0);
            PL_MFMA(1);
            PL_LOAD(3, 1);
            PL_MFMA(0);
            PL_MFMA(1);
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
#undef PL_LOAD
#undef PL_MFMA
#endif
            slot ^= 1;
        }
#ifdef QST_PL_NOEPI
#pragma unroll
        for (int i = 0; i < 2; ++i)
#pragma unroll
            for (int j = 0; j < 3; ++j) asm volatile("" :: "v"(acc[i][j]));
#else
        // ---- epilogue, from registers; no workgroup barrier (the loaders' next DMA goes into the slot this tile's last stage
        // did not use, and they wait at the next tile's first barrier)
        if (g.bias) { bias_s[lane] = bv0; if (lane < 32) bias_s[64 + lane] = bv1; }
        if constexpr (kF32Out) {
            u32x4 res[2][4];
            pl_resid(g, rs, mw, nw, 0, 0, lane, res[0]);
#define PL_BLOCK(S_)                                                                                                \
    do {                                                                                                            \
        if ((S_) + 1 < 6) pl_resid(g, rs, mw, nw, ((S_) + 1) / 3, ((S_) + 1) % 3, lane, res[((S_) + 1) & 1]);      \
        pl_block_f32<EPI>(g, rs, acc[(S_) / 3][(S_) % 3], bias_s, mw, nw, (S_) / 3, (S_) % 3, lane, res[(S_) & 1], dc); \
    } while (0)
            PL_BLOCK(0); PL_BLOCK(1); PL_BLOCK(2); PL_BLOCK(3); PL_BLOCK(4); PL_BLOCK(5);
#undef PL_BLOCK
        } else {
            u32x4 aux[2][2];
#pragma unroll
            for (int a = 0; a < 2; ++a)
#pragma unroll
                for (int p = 0; p < 2; ++p) aux[a][p] = u32x4{0u, 0u, 0u, 0u};
            if constexpr (EPI == QST_EPI_GELU_BWD) pl_aux<EPI>(g, rs, mw, nw, 0, 0, lane, aux[0]);
#define PL_BLOCK(S_)                                                                                                \
    do {                                                                                                            \
        if (EPI == QST_EPI_GELU_BWD && (S_) + 1 < 6)                                                                \
            pl_aux<EPI>(g, rs, mw, nw, ((S_) + 1) / 3, ((S_) + 1) % 3, lane, aux[((S_) + 1) & 1]);                  \
        pl_block_bf16<EPI>(g, rs, acc[(S_) / 3][(S_) % 3], bias_s, mw, nw, (S_) / 3, (S_) % 3, lane, aux[(S_) & 1]); \
    } while (0)
            PL_BLOCK(0); PL_BLOCK(1); PL_BLOCK(2); PL_BLOCK(3); PL_BLOCK(4); PL_BLOCK(5);
#undef PL_BLOCK
        }
#endif
    }
}

int g_cus = 0;

template <int EPI>
int launch_pl(const QstGemmArgs* a, hipStream_t st) {
    static QstLdsAttr attr;
    if (int rc = qst_ensure_lds(attr, (const void*)gemm_nt_pl_kernel<EPI>, LLDS)) return rc;
    if (g_cus == 0) {
        int dev = 0, n = 0;
        QST_HIP_CHECK(hipGetDevice(&dev));
        QST_HIP_CHECK(hipDeviceGetAttribute(&n, hipDeviceAttributeMultiprocessorCount, dev));
        g_cus = n > 0 ? n / 8 * 8 : 256;
        if (g_cus < 8) g_cus = 8;
    }
    const int T = ((a->M + LBM - 1) / LBM) * ((a->N + LBN - 1) / LBN);
    const int grid = min(g_cus, (T + 7) / 8 * 8);                 // one persistent workgroup per CU, a multiple of 8
    gemm_nt_pl_kernel<EPI><<<dim3(grid), dim3(768), LLDS, st>>>(*a);
    QST_LAUNCH_CHECK();
    return QST_OK;
}

}  // namespace

// shapes / options this form takes (the caller falls back to the tiled kernels otherwise)
extern "C" int qst_gemm_nt_pl_ok(const QstGemmArgs* a, int epi) {
    if (!a) return 0;
    if (a->a_head_L || a->c_head_L) return 0;
    if (a->K % LBK != 0 || a->lda % 8 != 0 || a->ldb % 8 != 0 || a->N % 8 != 0 || a->ldc % 8 != 0) return 0;
    if (a->resid && a->ldr % 4 != 0) return 0;
    // epilogue accesses are raw buffer operations with 32-bit offsets
    if ((int64_t)a->M * a->ldc * 4 >= 0x7FFFFF00LL || (int64_t)a->M * (a->ldr > 0 ? a->ldr : 1) * 4 >= 0x7FFFFF00LL) return 0;
    if (a->drop.thr16 && a->drop.state && (int64_t)a->M * a->N >= ((int64_t)1 << 32)) return 0;
    if ((int64_t)LBM * a->lda * 2 >= 0x7FFFFF00LL || (int64_t)LBN * a->ldb * 2 >= 0x7FFFFF00LL) return 0;
    switch (epi) {
        case QST_EPI_BF16: case QST_EPI_GELU: case QST_EPI_GELU_BWD: case QST_EPI_F32_RESID: case QST_EPI_F32_RESID_BF16: return 1;
        default: return 0;
    }
}

extern "C" int qst_gemm_nt_pl(const QstGemmArgs* a, int epi, void* stream) {
    if (!qst_gemm_nt_pl_ok(a, epi)) return QST_ERR_UNSUPPORTED;
    hipStream_t st = (hipStream_t)stream;
    switch (epi) {
        case QST_EPI_BF16: return launch_pl<QST_EPI_BF16>(a, st);
        case QST_EPI_GELU: return launch_pl<QST_EPI_GELU>(a, st);
        case QST_EPI_GELU_BWD: return launch_pl<QST_EPI_GELU_BWD>(a, st);
        case QST_EPI_F32_RESID: return launch_pl<QST_EPI_F32_RESID>(a, st);
        case QST_EPI_F32_RESID_BF16: return launch_pl<QST_EPI_F32_RESID_BF16>(a, st);
        default: return QST_ERR_BAD_ARG;
    }
}
