// gemm8.hip -- the long-reduction bf16 GEMMs of the encoder on the 8-wave, 8-phase K loop of gemm8p.h (gfx950).
//
//   gemm_nt8_kernel<EPI, TM, TN>     C[M,N] = A[M,K] . B[N,K]^T + the epilogues of qst_gemm_nt (QST_EPI_*); forward Linear and
//                                    dgrad of the K >= 768 models (nn.Linear inside BertLayer / MPNetLayer, transformers
//                                    modeling_bert.py:154-156, 282-293, 325-351)
//   gemm_tn8_group_kernel<TM, TN>    all weight gradients of a layer in one launch: C_p[N_p, K_p] += A_p[M, N_p]^T . B_p[M, K_p]
//                                    with fp32 atomics over per-XCD ranges of M, bias gradients as column sums of A
// Both are one 512-thread workgroup per CU (128 KB of LDS, <= 256 registers). The epilogues work on registers only: a
// v_permlane16_swap per accumulator register turns the 16x16 MFMA layout (4 consecutive columns per lane) into 8
// consecutive columns per lane, so every global access is a 16-byte, row-contiguous piece (64 to 128 bytes per row and
// instruction) -- no LDS staging, no workgroup barrier after the K loop.
#include "qst_common.h"
#include "qst_kernels.h"
#include "gemm8p.h"

namespace {

using g8p::f32x4_t;

// after the swap, lane (g = lane >> 4, c = lane & 15) holds row c of the 16-row tile and the 8 consecutive columns
// starting at pair_col(g) of the 32 columns of two adjacent 16-column tiles:  g = 0: 0, g = 1: 16, g = 2: 8, g = 3: 24
__device__ __forceinline__ int pair_col(int g) { return (g & 1) ? 16 + 4 * (g - 1) : 4 * g; }
__device__ __forceinline__ void pair8(const f32x4_t& x, const f32x4_t& y, float (&v)[8]) {
    // x: this lane's 4 columns of tile j, y: of tile j + 1. v_permlane16_swap trades the odd 16-lane rows of its first
    // operand with the even rows of its second (tools/probe/isa_probe.hip): even g then holds [x own | x of g + 1], odd g
    // [y of g - 1 | y own]. Inline asm: hipcc (ROCm 7.2) folds several __builtin_amdgcn_permlane16_swap calls with different
    // operands into the first one (seen in the ISA of this epilogue: one swap per tile pair, its first result used for
    // all eight values); `s_nop 1` = the two wait states a VALU write of an operand needs before the swap reads it.
#pragma unroll
    for (int r = 0; r < 4; ++r) {
        float a = x[r], b = y[r];
        asm volatile("s_nop 1\n\tv_permlane16_swap_b32 %0, %1" : "+v"(a), "+v"(b));
        v[r] = a;
        v[4 + r] = b;
    }
}

// Epilogue of an 8-phase NT tile (bf16 or MXFP8 operands: the accumulator layout is the same), registers only.
template <int EPI, class OPS>
__device__ __forceinline__ void nt8_epilogue(const QstGemmArgs& g, OPS& o, int m0, int n0) {
    constexpr int BM = OPS::BM, BN = OPS::BN, TM = BM / 32, TN = BN / 64, NP = TN / 2;
    const int lane = threadIdx.x & 63, gq = lane >> 4;
    const int mw = m0 + o.wr * (BM / 2) + (lane & 15);                 // + 16 i
    const int nw = n0 + o.wc * (BN / 4) + pair_col(gq);                // + 32 jp
    constexpr bool kF32 = (EPI == QST_EPI_F32_RESID || EPI == QST_EPI_F32_RESID_BF16);
    const f32x4 z4 = {0.f, 0.f, 0.f, 0.f};
    f32x4 bv[NP][2];
#pragma unroll
    for (int jp = 0; jp < NP; ++jp) {
        const int n = nw + 32 * jp;
        const bool ok = g.bias != nullptr && n < g.N;
        bv[jp][0] = ok ? *(const f32x4*)(g.bias + n) : z4;
        bv[jp][1] = ok ? *(const f32x4*)(g.bias + n + 4) : z4;
    }
    DropCtx dc = DropCtx{0u, 0u, 1.f};
    if (kF32 && g.drop_where == 1) dc = drop_ctx(g.drop);

    if constexpr (kF32) {
        // The residual rows of HALF the wave's row-tiles are requested in one burst before that half's first store (TM NP
        // 16-byte loads per lane: 64 registers at TM = 8): vmcnt retires in order, so a load issued behind stores waits for
        // them, and a burst per row-tile (two loads, then two stores, eight times over) exposed most of a memory round trip per
        // row-tile -- one workgroup per CU has nothing else to run meanwhile.
        constexpr int HT = TM / 2;
        f32x4 rv[HT][NP][2];
        auto load_resid = [&](int i, f32x4 (&dst)[NP][2]) {
            const int m = mw + 16 * i;
#pragma unroll
            for (int jp = 0; jp < NP; ++jp) {
                const int n = nw + 32 * jp;
                const bool ok = g.resid != nullptr && m < g.M && n < g.N;
                const float* p = g.resid + (size_t)m * g.ldr + n;
                dst[jp][0] = ok ? ld_stream((const f32x4*)p) : z4;
                dst[jp][1] = ok ? ld_stream((const f32x4*)(p + 4)) : z4;
            }
        };
#pragma unroll
        for (int i = 0; i < TM; ++i) {
            if (i % HT == 0) {
#pragma unroll
                for (int k = 0; k < HT; ++k) load_resid(i + k, rv[k]);
            }
            const int m = mw + 16 * i;
#pragma unroll
            for (int jp = 0; jp < NP; ++jp) {
                const int n = nw + 32 * jp;
                float v[8];
                pair8(o.acc[i][2 * jp], o.acc[i][2 * jp + 1], v);
                if (m >= g.M || n >= g.N) continue;
#pragma unroll
                for (int e = 0; e < 4; ++e) { v[e] += bv[jp][0][e]; v[4 + e] += bv[jp][1][e]; }
                if (dc.thr) {
                    const uint32_t e0 = (uint32_t)m * (uint32_t)g.N + (uint32_t)n;
#pragma unroll
                    for (int e = 0; e < 8; e += 2) {
                        float k0, k1;
                        drop_pair(dc, e0 + e, k0, k1);
                        v[e] *= k0; v[e + 1] *= k1;
                    }
                }
                f32x4 lo, hi;
#pragma unroll
                for (int e = 0; e < 4; ++e) { lo[e] = v[e] + rv[i % HT][jp][0][e]; hi[e] = v[4 + e] + rv[i % HT][jp][1][e]; }
                float* c = (float*)g.C + (size_t)m * g.ldc + n;
                st_stream((f32x4*)c, lo);
                st_stream((f32x4*)(c + 4), hi);
                if (EPI == QST_EPI_F32_RESID_BF16) {
                    u32x4 pk;
                    pk[0] = pack_op2(lo[0], lo[1]); pk[1] = pack_op2(lo[2], lo[3]);
                    pk[2] = pack_op2(hi[0], hi[1]); pk[3] = pack_op2(hi[2], hi[3]);
                    st_stream((u32x4*)((op16*)g.C2 + (size_t)m * g.ldc + n), pk);
                }
            }
        }
    } else {
        u32x4 av[TM][NP];                                  // every saved gelu'(u) row of the wave's block, requested up front
        const u32x4 zu = {0u, 0u, 0u, 0u};
        auto load_aux = [&](int i, u32x4 (&dst)[NP]) {
            const int m = mw + 16 * i;
#pragma unroll
            for (int jp = 0; jp < NP; ++jp) {
                const int n = nw + 32 * jp;
                dst[jp] = (m < g.M && n < g.N) ? ld_stream((const u32x4*)((const op16*)g.aux + (size_t)m * g.ldc + n)) : zu;
            }
        };
        if (EPI == QST_EPI_GELU_BWD) {
#pragma unroll
            for (int i = 0; i < TM; ++i) load_aux(i, av[i]);
        }
#pragma unroll
        for (int i = 0; i < TM; ++i) {
            const int m = mw + 16 * i;
#pragma unroll
            for (int jp = 0; jp < NP; ++jp) {
                const int n = nw + 32 * jp;
                float v[8];
                pair8(o.acc[i][2 * jp], o.acc[i][2 * jp + 1], v);
                if (m >= g.M || n >= g.N) continue;
#pragma unroll
                for (int e = 0; e < 4; ++e) { v[e] += bv[jp][0][e]; v[4 + e] += bv[jp][1][e]; }
                const size_t off = (size_t)m * g.ldc + n;
                u32x4 pk;
                if (EPI == QST_EPI_BF16) {
#pragma unroll
                    for (int e = 0; e < 4; ++e) pk[e] = pack_op2(v[2 * e], v[2 * e + 1]);
                    st_stream((u32x4*)((op16*)g.C + off), pk);
                } else if (EPI == QST_EPI_GELU || EPI == QST_EPI_GELU_MX_TRAIN) {
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
                    st_stream((u32x4*)((op16*)g.C + off), pg);             // gelu'(u), saved for backward
                    st_stream((u32x4*)((op16*)g.C2 + off), pk);            // h = gelu(u)
                    if (EPI == QST_EPI_GELU_MX_TRAIN) {
                        // ... and the bf16-rounded h as MXFP8 (C3 = e4m3 [M, N], C4 = E8M0, stage-major): the four lanes of a
                        // row (one per 16-lane row of the wave) hold the 32 columns of one MX block, so its amax is two
                        // row-swaps away. (All four take this branch together: same m, same 32-column block, N % 32 == 0.)
                        float hv[8], amax = 0.f;
#pragma unroll
                        for (int e = 0; e < 4; ++e) { hv[2 * e] = op_lo(pk[e]); hv[2 * e + 1] = op_hi(pk[e]); }
#pragma unroll
                        for (int e = 0; e < 8; ++e) amax = fmaxf(amax, fabsf(hv[e]));
                        amax = fmaxf(amax, swap32(amax));
                        {
                            float a2 = amax, b2 = amax;
                            asm volatile("s_nop 1\n\tv_permlane16_swap_b32 %0, %1" : "+v"(a2), "+v"(b2));
                            amax = fmaxf(a2, b2);
                        }
                        const int ex = mx_exponent(amax);
                        const float inv = pow2f(-ex);
                        uint32_t p0 = 0, p1 = 0;
                        p0 = __builtin_amdgcn_cvt_pk_fp8_f32(hv[0] * inv, hv[1] * inv, p0, false);
                        p0 = __builtin_amdgcn_cvt_pk_fp8_f32(hv[2] * inv, hv[3] * inv, p0, true);
                        p1 = __builtin_amdgcn_cvt_pk_fp8_f32(hv[4] * inv, hv[5] * inv, p1, false);
                        p1 = __builtin_amdgcn_cvt_pk_fp8_f32(hv[6] * inv, hv[7] * inv, p1, true);
                        u32x2 q2; q2[0] = p0; q2[1] = p1;
                        st_stream((u32x2*)((uint8_t*)g.C3 + off), q2);
                        if (gq == 0) ((uint8_t*)g.C4)[((size_t)(n >> 7) * g.M + m) * 4 + ((n >> 5) & 3)] = (uint8_t)(ex + 127);
                    }
                } else {                                                    // QST_EPI_GELU_BWD: acc * gelu'(u)
#pragma unroll
                    for (int e = 0; e < 4; ++e)
                        pk[e] = pack_op2(v[2 * e] * op_lo(av[i][jp][e]), v[2 * e + 1] * op_hi(av[i][jp][e]));
                    st_stream((u32x4*)((op16*)g.C + off), pk);
                }
            }
        }
    }
}


template <int EPI, int TM, int TN>
__global__ __launch_bounds__(512, 1) void gemm_nt8_kernel(QstGemmArgs g) {
    op_saturate(g.sat16 != 0);
    using OPS = g8p::NtOps<TM, TN>;
    constexpr int BM = OPS::BM, BN = OPS::BN;
    extern __shared__ __attribute__((aligned(16))) char smem[];
    const int ntn = (g.N + BN - 1) / BN, ntm = (g.M + BM - 1) / BM;
    const int wg = g8p::xcd_remap(blockIdx.x, ntm * ntn);
    // (column-major tile order inside an XCD's share -- one weight panel per L2, the activation rows streamed past it -- was
    // measured on the H = 768 shapes: within 1% of this order on every one)
    const int m0 = (wg / ntn) * BM, n0 = (wg % ntn) * BN;
    OPS o;
    o.init((const op16*)g.A + (size_t)m0 * g.lda, g.lda, min(BM, g.M - m0), (const op16*)g.B + (size_t)n0 * g.ldb, g.ldb,
           min(BN, g.N - n0), g.K, smem);
    g8p::kloop8(o, o.nk);
    nt8_epilogue<EPI>(g, o, m0, n0);
}

// The same tile on the fp8 matrix cores: A, B = e4m3 [rows, K] (lda / ldb in bytes), aux / bscale = their E8M0 block scales
// (qst_quant_mx's stage-major layout), K % 128 == 0 -- the operands of qst_gemm_nt_f8 (gemm.hip).
template <int EPI, int TM, int TN>
__global__ __launch_bounds__(512, 1) void gemm_nt8_f8_kernel(QstGemmArgs g) {
    using OPS = g8p::NtOpsF8<TM, TN>;
    constexpr int BM = OPS::BM, BN = OPS::BN;
    extern __shared__ __attribute__((aligned(16))) char smem[];
    const int ntn = (g.N + BN - 1) / BN, ntm = (g.M + BM - 1) / BM;
    const int wg = g8p::xcd_remap(blockIdx.x, ntm * ntn);
    const int m0 = (wg / ntn) * BM, n0 = (wg % ntn) * BN;
    OPS o;
    o.init((const uint8_t*)g.A + (size_t)m0 * g.lda, g.lda, min(BM, g.M - m0), (const uint8_t*)g.aux, g.M, m0,
           (const uint8_t*)g.B + (size_t)n0 * g.ldb, g.ldb, min(BN, g.N - n0), (const uint8_t*)g.bscale, g.N, n0, g.K, smem);
    g8p::kloop8(o, o.nk);
    nt8_epilogue<EPI>(g, o, m0, n0);
}

// ---------------------------------------------------------------- grouped weight gradients
// Bias gradient = column sums of A (= dY) over the reduction rows: the four waves of a wave row hold the same A fragments,
// so each sums a quarter of the row's tiles (v_dot2c_f32_bf16 against (1, 1): one VALU instruction per register).
template <int TM>
struct BiasHook {
    static constexpr int PER = TM / 4;                     // tiles per wave
    bool on;
    float bsum[PER];
    template <int QM, class O> __device__ __forceinline__ void after_a(O& o) {
        typedef __attribute__((ext_vector_type(2))) op16 v2bf;
        if (!on || (o.wc >> 1) != QM) return;              // wave-uniform
        v2bf one; one[0] = (op16)1.f; one[1] = (op16)1.f;
#pragma unroll
        for (int e = 0; e < PER; ++e) {
#pragma unroll
            for (int s = 0; s < 2; ++s) {
                const g8p::op16x8 f = (o.wc & 1) ? o.fa[PER + e][s] : o.fa[e][s];
#pragma unroll
                for (int r = 0; r < 4; ++r) {
                    v2bf pr; pr[0] = f[2 * r]; pr[1] = f[2 * r + 1];
#if QST_OP_F16
                    bsum[e] = __builtin_amdgcn_fdot2(pr, one, bsum[e], false);              // v_dot2_f32_f16
#else
                    bsum[e] = __builtin_amdgcn_fdot2_f32_bf16(pr, one, bsum[e], false);
#endif
                }
            }
        }
    }
};

// Work decomposition as gemm_tn_group_kernel (gemm.hip): the reduction over M is cut into ranges, one (or more) per XCD;
// the W workgroups of a range reduce T tiles = a W + b as `a` whole tiles each, in lockstep (so the rows being streamed
// are shared through the XCD's L2), and the b leftover tiles in floor(W / b) row pieces each; partial tiles meet in
// fp32 atomics.
template <int TM, int TN>
__global__ __launch_bounds__(512, 1) void gemm_tn8_group_kernel(QstTnGroup grp) {
    using OPS = g8p::TnOps<TM, TN, BiasHook<TM>>;
    constexpr int BMn = OPS::BMn, BNk = OPS::BNk;
    extern __shared__ __attribute__((aligned(16))) char smem[];
    const int lane = threadIdx.x & 63;
    const int xcd = blockIdx.x & 7, jloc = blockIdx.x >> 3;        // blocks b and b + 8 share an XCD (speed only)
    const int wg_per_range = (int)(gridDim.x >> 3) / grp.ranges_per_xcd;
    const int range = xcd + 8 * (jloc / wg_per_range);
    const int jr = jloc % wg_per_range;
    const int M = grp.prob[0].M;
    const int nsplit = grp.splits;
    const int per = (((M + nsplit - 1) / nsplit) + 63) / 64 * 64;
    const int mbeg = range * per, mend = min(M, mbeg + per);
    if (mbeg >= mend) return;
    const int S = (mend - mbeg + 63) / 64;
    const int W = wg_per_range, T = grp.total_tiles;
    const int a_full = T / W, b_left = T % W;
    const int pieces = b_left > 0 ? max(1, W / b_left) : 1;
    const int ntasks = a_full + ((b_left > 0 && jr / pieces < b_left) ? 1 : 0);

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
        if (s1 <= s0) continue;                                    // uniform over the workgroup
        int pi = 0;
#pragma unroll 1
        while (pi + 1 < grp.nprob && tile >= grp.tiles[pi]) { tile -= grp.tiles[pi]; ++pi; }
        const QstGemmArgs& g = grp.prob[pi];
        const int ntk = (g.K + BNk - 1) / BNk;
        const int n0 = (tile / ntk) * BMn, k0 = (tile % ntk) * BNk;
        const int row0 = mbeg + s0 * 64, row1 = min(mend, mbeg + s1 * 64);

        OPS o;
        o.hook.on = (g.colsum != nullptr) && (k0 == 0);
#pragma unroll
        for (int e = 0; e < BiasHook<TM>::PER; ++e) o.hook.bsum[e] = 0.f;
        o.init((const op16*)g.A + (size_t)row0 * g.lda, g.lda, g.N, n0, (const op16*)g.B + (size_t)row0 * g.ldb, g.ldb, g.K, k0,
               row1 - row0, smem);
        g8p::kloop8(o, o.nk);

        float* C = (float*)g.C;
        const int gq = lane >> 4, cl = lane & 15;
        {
            // Float atomics run at the memory side at full rate when a wave-instruction covers 128-byte segments (guide:
            // "two 128-B segments in two rows"); a 16x16 accumulator register is four 64-byte segments in four rows. One
            // v_permlane16_swap per register pair of two adjacent column tiles turns it into two rows x 128 bytes: after
            // the swap lanes g = 0, 1 hold row (4 g' + r) of tiles j, j + 1 side by side, lanes g = 2, 3 the row 8 below.
            // (Measured: 64- and 128-byte segments flush at the same rate -- 229.9 vs 230.3 us for the MiniLM layer -- the form
            // stays because it is the documented full-rate one.)
#pragma unroll
            for (int i = 0; i < TM; ++i) {
                const int nrow = n0 + (i / (TM / 2)) * (BMn / 2) + o.wr * (BMn / 4) + (i % (TM / 2)) * 16 + 4 * (gq & 2);
#pragma unroll
                for (int jp = 0; jp < TN / 2; ++jp) {
                    // pairs of adjacent tiles inside a quadrant column; with three tiles per quadrant column (TN = 6) the third
                    // tiles of the two quadrant columns pair up (not adjacent: 64-byte segments for that third of the flush)
                    constexpr int H2 = TN / 2;
                    const int ja = (H2 % 2 == 0) ? 2 * jp : (jp == H2 - 1 ? H2 - 1 : (jp < H2 / 2 ? 2 * jp : H2 + 2 * (jp - H2 / 2)));
                    const int jb = (H2 % 2 == 0) ? 2 * jp + 1 : (jp == H2 - 1 ? 2 * H2 - 1 : ja + 1);
                    const int jmine = (gq & 1) ? jb : ja;
                    const int k = k0 + (jmine / H2) * (BNk / 2) + o.wc * (BNk / 8) + (jmine % H2) * 16 + cl;
#pragma unroll
                    for (int r = 0; r < 4; ++r) {
                        float a = o.acc[i][ja][r], b = o.acc[i][jb][r];
                        asm volatile("s_nop 1\n\tv_permlane16_swap_b32 %0, %1" : "+v"(a), "+v"(b));
                        if (k < g.K) {
                            if (nrow + r < g.N) atomicAdd(&C[(size_t)(nrow + r) * g.ldc + k], a);
                            if (nrow + 4 + r < g.N) atomicAdd(&C[(size_t)(nrow + 4 + r) * g.ldc + k], b);
                        }
                    }
                }
            }
        }
        if (o.hook.on) {
            const int qm = o.wc >> 1;
#pragma unroll
            for (int e = 0; e < BiasHook<TM>::PER; ++e) {
                const int n = n0 + qm * (BMn / 2) + o.wr * (BMn / 4) + ((o.wc & 1) * BiasHook<TM>::PER + e) * 16 + cl;
                if (n < g.N) atomicAdd(&g.colsum[n], o.hook.bsum[e]);         // the four lane groups hold four row subsets
            }
        }
        // kloop8 left a barrier behind it; the next task's first DMAs overwrite LDS that every wave has finished reading
    }
}

#if !QST_OP_F16
std::atomic<int> g_mode{-1};       // qst_gemm8_mode
#endif

}  // namespace

// -1 (default): the library chooses per call; otherwise bit 0 = NT GEMMs on this path where supported, bit 1 = weight
// gradients on this path. Process-wide; for tools and A/B runs.
#if !QST_OP_F16      // (one switch for both builds of this file; the f16 build reads it through qst_gemm8_mode_get)
extern "C" int qst_gemm8_mode(int mode) {
    const int old = g_mode.load();
    if (mode >= -1) g_mode.store(mode);
    return old;
}
int qst_gemm8_mode_get() { return g_mode.load(); }
#else
int qst_gemm8_mode_get();
#endif

template <int EPI, int TM, int TN>
static int launch_nt8(const QstGemmArgs* a, hipStream_t st) {
    using OPS = g8p::NtOps<TM, TN>;
    static QstLdsAttr attr;
    if (int rc = qst_ensure_lds(attr, (const void*)gemm_nt8_kernel<EPI, TM, TN>, g8p::LDS_BYTES)) return rc;
    const int ntm = (a->M + OPS::BM - 1) / OPS::BM, ntn = (a->N + OPS::BN - 1) / OPS::BN;
    gemm_nt8_kernel<EPI, TM, TN><<<dim3(ntm * ntn), dim3(512), g8p::LDS_BYTES, st>>>(*a);
    QST_LAUNCH_CHECK();
    return QST_OK;
}

// tile: 0 = 128 x 384 (8 waves of 64 x 96), 1 = 256 x 256 (8 waves of 128 x 64)
extern "C" int QST_K(qst_gemm_nt8_supported)(const QstGemmArgs* a, int epi) {
    if (!a || a->K % 64 != 0 || a->lda % 8 != 0 || a->ldb % 8 != 0 || a->N % 8 != 0 || a->ldc % 8 != 0) return 0;
    if ((epi == QST_EPI_F32_RESID || epi == QST_EPI_F32_RESID_BF16) && a->resid && a->ldr % 4 != 0) return 0;
    if ((int64_t)256 * a->lda * 2 >= 0x7FFFFF00LL || (int64_t)384 * a->ldb * 2 >= 0x7FFFFF00LL) return 0;
    return epi >= QST_EPI_BF16 && epi <= QST_EPI_F32_RESID_BF16;
}

extern "C" int QST_K(qst_gemm_nt8)(const QstGemmArgs* a, int epi, int tile, void* stream) {
    if (!a || !a->A || !a->B || !a->C || a->M <= 0 || a->N <= 0 || a->K <= 0) return QST_ERR_BAD_ARG;
    if (!QST_K(qst_gemm_nt8_supported)(a, epi)) return QST_ERR_UNSUPPORTED;
    if ((epi == QST_EPI_GELU || epi == QST_EPI_F32_RESID_BF16) && !a->C2) return QST_ERR_BAD_ARG;
    if (epi == QST_EPI_GELU_BWD && !a->aux) return QST_ERR_BAD_ARG;
    if (a->drop.thr16 && a->drop.state) {
        if (a->drop_where != 1 || (epi != QST_EPI_F32_RESID && epi != QST_EPI_F32_RESID_BF16)) return QST_ERR_BAD_ARG;
        if (a->drop.thr16 > 65535u || (int64_t)a->M * a->N >= ((int64_t)1 << 32)) return QST_ERR_UNSUPPORTED;
    }
    hipStream_t st = (hipStream_t)stream;
#define QST_NT8_CASE(E) case E: return tile == 1 ? launch_nt8<E, 8, 4>(a, st) : launch_nt8<E, 4, 6>(a, st);
    switch (epi) {
        QST_NT8_CASE(QST_EPI_BF16)
        QST_NT8_CASE(QST_EPI_F32_RESID)
        QST_NT8_CASE(QST_EPI_GELU)
        QST_NT8_CASE(QST_EPI_GELU_BWD)
        QST_NT8_CASE(QST_EPI_F32_RESID_BF16)
        default: return QST_ERR_BAD_ARG;
    }
#undef QST_NT8_CASE
}

#if !QST_OP_F16
template <int EPI, int TM, int TN>
static int launch_nt8_f8(const QstGemmArgs* a, hipStream_t st) {
    using OPS = g8p::NtOpsF8<TM, TN>;
    static QstLdsAttr attr;
    if (int rc = qst_ensure_lds(attr, (const void*)gemm_nt8_f8_kernel<EPI, TM, TN>, OPS::LDS_TOTAL)) return rc;
    const int ntm = (a->M + OPS::BM - 1) / OPS::BM, ntn = (a->N + OPS::BN - 1) / OPS::BN;
    gemm_nt8_f8_kernel<EPI, TM, TN><<<dim3(ntm * ntn), dim3(512), OPS::LDS_TOTAL, st>>>(*a);
    QST_LAUNCH_CHECK();
    return QST_OK;
}

// MXFP8 x MXFP8 on the 8-phase loop: operands as qst_gemm_nt_f8; epi QST_EPI_BF16, QST_EPI_F32_RESID (+ dropout of the
// projection output), QST_EPI_GELU (gelu'(u) and h as bf16). tile: 0 = 128 x 384, 1 = 256 x 256.
extern "C" int qst_gemm_nt8_f8(const QstGemmArgs* a, int epi, int tile, void* stream) {
    if (!a || !a->A || !a->B || !a->C || !a->aux || !a->bscale || a->M <= 0 || a->N <= 0 || a->K <= 0) return QST_ERR_BAD_ARG;
    if (a->K % 128 != 0 || a->lda % 16 != 0 || a->ldb % 16 != 0 || a->N % 8 != 0 || a->ldc % 8 != 0) return QST_ERR_UNSUPPORTED;
    if (epi == QST_EPI_F32_RESID && a->resid && a->ldr % 4 != 0) return QST_ERR_UNSUPPORTED;
    if ((int64_t)256 * a->lda >= 0x7FFFFF00LL || (int64_t)384 * a->ldb >= 0x7FFFFF00LL) return QST_ERR_UNSUPPORTED;
    if ((int64_t)(a->K / 128) * (a->M > a->N ? a->M : a->N) * 4 >= 0x7FFFFF00LL) return QST_ERR_UNSUPPORTED;
    if (epi == QST_EPI_GELU && !a->C2) return QST_ERR_BAD_ARG;
    if (epi == QST_EPI_GELU_MX_TRAIN && (!a->C2 || !a->C3 || !a->C4)) return QST_ERR_BAD_ARG;
    if (epi == QST_EPI_GELU_MX_TRAIN && (a->ldc != a->N || a->N % 32 != 0)) return QST_ERR_UNSUPPORTED;
    if (a->drop.thr16 && a->drop.state) {
        if (a->drop_where != 1 || epi != QST_EPI_F32_RESID) return QST_ERR_BAD_ARG;
        if (a->drop.thr16 > 65535u || (int64_t)a->M * a->N >= ((int64_t)1 << 32)) return QST_ERR_UNSUPPORTED;
    }
    hipStream_t st = (hipStream_t)stream;
#define QST_NT8F_CASE(E) case E: return tile == 1 ? launch_nt8_f8<E, 8, 4>(a, st) : launch_nt8_f8<E, 4, 6>(a, st);
    switch (epi) {
        QST_NT8F_CASE(QST_EPI_BF16)
        QST_NT8F_CASE(QST_EPI_F32_RESID)
        QST_NT8F_CASE(QST_EPI_GELU)
        QST_NT8F_CASE(QST_EPI_GELU_MX_TRAIN)
        default: return QST_ERR_UNSUPPORTED;
    }
#undef QST_NT8F_CASE
}

#endif  // !QST_OP_F16

template <int TM, int TN>
static int launch_tn8(const QstTnGroup* grp_in, hipStream_t st) {
    constexpr int BMn = 32 * TM, BNk = 64 * TN;
    QstTnGroup g = *grp_in;
    g.total_tiles = 0;
    for (int i = 0; i < g.nprob; ++i) {
        const QstGemmArgs& a = g.prob[i];
        g.tiles[i] = ((a.N + BMn - 1) / BMn) * ((a.K + BNk - 1) / BNk);
        g.total_tiles += g.tiles[i];
    }
    const int M = g.prob[0].M;
    if (g.splits <= 0) g.splits = 8;                  // one M-range per XCD: every operand row leaves HBM once
    g.splits = (g.splits + 7) / 8 * 8;
    g.ranges_per_xcd = g.splits / 8;
    const int64_t stages = ((int64_t)(M + g.splits - 1) / g.splits + 63) / 64;
    int64_t wg_per_range = 32 / g.ranges_per_xcd;
    if (wg_per_range < 1) wg_per_range = 1;
    const int64_t work = (int64_t)g.total_tiles * stages;
    if (wg_per_range > work) wg_per_range = work < 1 ? 1 : work;
    const int grid = (int)(8 * g.ranges_per_xcd * wg_per_range);
    static QstLdsAttr attr;
    if (int rc = qst_ensure_lds(attr, (const void*)gemm_tn8_group_kernel<TM, TN>, g8p::LDS_BYTES)) return rc;
    gemm_tn8_group_kernel<TM, TN><<<dim3(grid), dim3(512), g8p::LDS_BYTES, st>>>(g);
    QST_LAUNCH_CHECK();
    return QST_OK;
}

extern "C" int QST_K(qst_gemm_tn8_group)(const QstTnGroup* grp_in, void* stream) {
    if (!grp_in || grp_in->nprob <= 0 || grp_in->nprob > QST_TN_MAX_PROB) return QST_ERR_BAD_ARG;
    bool all256 = true;
    for (int i = 0; i < grp_in->nprob; ++i) {
        const QstGemmArgs& a = grp_in->prob[i];
        if (!a.A || !a.B || !a.C || a.M <= 0 || a.N <= 0 || a.K <= 0 || a.M != grp_in->prob[0].M) return QST_ERR_BAD_ARG;
        if (a.lda % 8 != 0 || a.ldb % 8 != 0 || a.N % 8 != 0 || a.K % 8 != 0) return QST_ERR_UNSUPPORTED;
        if ((int64_t)a.M * a.lda * 2 >= 0x7FFFFF00LL || (int64_t)a.M * a.ldb * 2 >= 0x7FFFFF00LL) return QST_ERR_UNSUPPORTED;
        all256 = all256 && a.N % 256 == 0 && a.K % 256 == 0;
    }
    // 256 x 256 tiles (a third more multiplications per operand byte staged) where they divide every gradient of the group
    // (the H = 768 models) would stage a third fewer operand bytes per multiplication, but measured slower there (mpnet layer,
    // flush excluded: 624 vs 551 us), so the 128 x 384 form, which divides every dimension of the three model families, is
    // the one used; the instantiation stays for tools/g8_bench.py.
    if (all256 && grp_in->splits == -256) return launch_tn8<8, 4>(grp_in, (hipStream_t)stream);
    return launch_tn8<4, 6>(grp_in, (hipStream_t)stream);
}
