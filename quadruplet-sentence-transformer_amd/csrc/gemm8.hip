// gemm8.hip -- the long-reduction bf16 GEMMs of the encoder on the 8-wave, 8-phase K loop of gemm8p.h (gfx950).
//
//   gemm_nt8_kernel<EPI, TM, TN>     C[M,N] = A[M,K] . B[N,K]^T + the epilogues of qst_gemm_nt (QST_EPI_*); forward Linear and
//                                    dgrad of the K >= 768 models (nn.Linear inside BertLayer / MPNetLayer, transformers
//                                    modeling_bert.py:154-156, 282-293, 325-351)
//   gemm_nt8_ln_kernel<MODE, DROPW, F8>   the same GEMM with the LayerNorm that follows it (mode 0: BertSelfOutput / BertOutput,
//                                    modeling_bert.py:296-306, 340-351) or the LayerNorm backward that follows the dgrad (mode 1)
//                                    in the epilogue, for rows of 2-4 tiles (H = 512 / 768 / 1024): the workgroups of a row
//                                    panel exchange row statistics inside the launch; F8: on the fp8 matrix cores, + MX emission
//   gemm_tn8_group_kernel<TM, TN>    all weight gradients of a layer in one launch: C_p[N_p, K_p] += A_p[M, N_p]^T . B_p[M, K_p]
//                                    with fp32 atomics over per-XCD ranges of M, bias gradients as column sums of A
// All are one 512-thread workgroup per CU (128 KB of LDS, <= 256 registers). The epilogues work on registers only: a
// v_permlane16_swap per accumulator register turns the 16x16 MFMA layout (4 consecutive columns per lane) into 8
// consecutive columns per lane, so every global access is a 16-byte, row-contiguous piece (64 to 128 bytes per row and
// instruction) -- no LDS staging, no workgroup barrier after the K loop.
#include <cstdlib>
#include "qst_common.h"
#include "qst_kernels.h"
#include "gemm8p.h"

namespace {

using g8p::f32x4_t;
typedef qst_f32x2 f32x2;

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


// Staggered start (qst_gemm8_stagger): one workgroup per CU runs { K loop ; epilogue } with nothing else resident, every CU
// of the chip starts together and does the same work, so all of them store at the same time and HBM's write side idles during
// the K loops. The workgroups of the FIRST round sleep for a fraction `key / 256` of `cycles` before they start; later rounds
// inherit the phase of the CU they land on.
__device__ __forceinline__ void stagger_start(int cycles, unsigned key) {
    if (cycles <= 0) return;
    const unsigned r = __brev(key & 255u) >> 24;                  // 0 .. 255, neighbours far apart
    const int n = (int)(((long long)cycles * r) >> 8) >> 10;      // s_sleep 16 = 1,024 cycles
#pragma unroll 1
    for (int i = 0; i < n; ++i) __builtin_amdgcn_s_sleep(16);
}

template <int EPI, int TM, int TN>
__global__ __launch_bounds__(512, 1) void gemm_nt8_kernel(QstGemmArgs g, int stagger) {
    op_saturate(g.sat16 != 0);
    if (blockIdx.x < 256) stagger_start(stagger, blockIdx.x);
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

// ---------------------------------------------------------------- GEMM + LayerNorm over rows wider than one tile
// C = LayerNorm(A.B^T + bias + resid) (mode 0) / the LayerNorm backward of dy = A.B^T + resid (mode 1), as gemm_nt_ln_kernel
// (gemm.hip) computes them for N = 384 in one full-row tile -- here for N = ntn x 256 on the 256 x 256 tile (or ntn x 384 on the 128 x 384 tile) of the 8-phase
// loop: the ntn workgroups of a 256-row panel each hold a third (N = 768) of every row in their accumulators and hand the
// row statistics to each other through global memory INSIDE the launch:
//   pass 1   v = acc + bias (+ dropout) + resid, back into the accumulator registers; per row and wave the mean and the
//            centred sum of squares of its 64 columns (mode 1: sum(g dy), sum(g dy xhat)) -> LDS
//   exchange waves (wr, 0) merge the four wave-columns (Chan's update: no E[v^2] - mean^2), publish the tile's pair for
//            each of their 128 rows as two 8-byte granules {tag, value} (sc1 stores: the data is the flag,
//            cdna_hip_programming.md Guideline 16 form R2), sweep the other tiles' granules of the same rows until every
//            tag is set (sc1 loads, bounded: a timeout sets *tmo and the launch finishes with wrong rows instead of
//            hanging), merge, and leave (mean, rstd) / (m1, m2) per row in LDS
//   pass 2   normalise from the registers and store y (f32), y (bf16), xhat (bf16), rstd -- the GEMM's output never
//            exists un-normalised in memory (the unfused pair writes it as f32 and reads it back: 1.2 GB per launch at
//            M = 196,608).
// Liveness needs in-order dispatch only: block b's partners are b +- 8, b +- 16 (same XCD under round-robin placement, but
// nothing depends on placement), so the set of dispatched workgroups is a prefix of the grid and at most ntn - 1 resident
// workgroups wait for a partner that has not started; every other one finishes and frees its CU. A granule's tag is the
// epoch of its buffer + 1; the last workgroup of a launch to finish advances the epoch (device-resident: a replayed graph
// counts on like an eager launch, nothing is zeroed between launches, no per-launch salt in the arguments).
// Mode 1 keeps dy in the accumulators and the tile's xhat fragment in LDS (lane-private slots in the 128 KB the K loop
// has left) across the exchange; the gamma / beta gradient partials (column sums over the panel's rows) go to
// partials[panel][2][N].
struct LnXchg {
    unsigned long long* gran;      // [ntm][ntn][BM rows][2] granules
    unsigned* ctl;                 // {epoch, workgroups done} of the buffer `gran` belongs to
    unsigned* tmo;                 // sticky timeout word
    int ntm, ntn, ppx;             // panels, tiles per panel, panels per XCD queue
    int frag;                      // fp32 rows stored as each lane's own 16-byte halves (qst_gemm8_ln_store)
};
typedef __attribute__((address_space(1))) unsigned long long gu64;
constexpr int LNX_RED = 256 * 4 * 2 * 4, LNX_STATS = 256 * 2 * 4, LNX_PR = 2 * 2 * 256 * 4;   // 256 x 256 tile; 128 x 384 needs 3 KB less in all
constexpr int LNX_LDS0 = g8p::LDS_BYTES;                                   // mode 0: the scratch aliases the K-loop buffers
constexpr int LNX_LDS1 = g8p::LDS_BYTES + LNX_RED + LNX_STATS + LNX_PR;   // mode 1: behind the xhat stash

// fp32 rows leave in 16-byte pieces, one per lane and instruction. A lane holds 8 consecutive columns = 32 bytes, its row's
// four lanes the columns 0-7 / 16-23 / 8-15 / 24-31 of a 32-column group: storing each lane's two halves as they are makes
// every instruction write four 16-byte pieces with 16-byte holes between them -- half sectors. After trading halves between
// the lanes 16 apart (v_permlane16_swap: even g keeps its low half and receives g + 1's low half, odd g receives g - 1's high
// half and keeps its own) lane g's first piece is columns 4 r .. 4 r + 3 and its second 16 + 4 r .. with r = 0, 1, 2, 3 for
// g = 0, 1, 2, 3: each instruction covers 64 contiguous bytes of a row -- two whole sectors.
__device__ __forceinline__ void store_f32_row8(float* row_base /* the 32-column group */, int gq, f32x4 lo, f32x4 hi, bool frag = false) {
    if (frag) {                                 // each lane's own two halves, as the plain epilogue stores them (qst_gemm8_ln_store)
        st_stream((f32x4*)(row_base + pair_col(gq)), lo);
        st_stream((f32x4*)(row_base + pair_col(gq) + 4), hi);
        return;
    }
#pragma unroll
    for (int r = 0; r < 4; ++r) {
        float a = lo[r], b = hi[r];
        asm volatile("s_nop 1\n\tv_permlane16_swap_b32 %0, %1" : "+v"(a), "+v"(b));
        lo[r] = a; hi[r] = b;
    }
    st_stream((f32x4*)(row_base + 4 * gq), lo);
    st_stream((f32x4*)(row_base + 16 + 4 * gq), hi);
}

// sum over the four lanes (lane & 15 equal) that share an accumulator row; every one of them receives the total
__device__ __forceinline__ float xg_sum(float v) {
    v += swap32(v);
    float a = v, b = v;
    asm volatile("s_nop 1\n\tv_permlane16_swap_b32 %0, %1" : "+v"(a), "+v"(b));
    return a + b;
}
typedef __attribute__((address_space(1))) unsigned gu32;
__device__ __forceinline__ void put_granule(unsigned long long* p, unsigned tag, float v) {
    __hip_atomic_store((gu64*)p, ((unsigned long long)tag << 32) | (unsigned long long)__builtin_bit_cast(uint32_t, v),
                       __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
}
// Every workgroup of a launch passes here once (thread 0): the last one advances the buffer's epoch for the next launch.
// The epoch cannot move while a workgroup of this launch has not arrived, so every workgroup reads the same value.
__device__ __forceinline__ void lnx_finish(unsigned* ctl) {
    if (threadIdx.x != 0) return;
    const unsigned ep = __hip_atomic_load((gu32*)ctl, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    const unsigned old = __hip_atomic_fetch_add((gu32*)(ctl + 1), 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    if (old == gridDim.x - 1) {
        __hip_atomic_store((gu32*)(ctl + 1), 0u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        __hip_atomic_store((gu32*)ctl, ep + 1u == 0xFFFFFFFFu ? 0u : ep + 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);   // tag = epoch + 1 is never 0
    }
}

// F8: both operands MXFP8 (operands as gemm_nt8_f8_kernel: g.aux / g.bscale = their E8M0 scales), mode 0 only, and the
// normalised output leaves a fourth time, as MXFP8 (g.C3 = e4m3 [M, N], g.C4 = scales, stage-major), quantised from the
// 16-bit-rounded values exactly as qst_ln_fwd_mx_train does it -- the A operand of the next fp8 GEMM (QST_PREC_FP8 forward).
// TM, TN: 8, 4 = the 256 x 256 tile (two tiles per CU from M = 43,691 at H = 768); 4, 6 = 128 x 384, which gives M = 32,768
// token rows at H = 768 (configs[2]) 512 whole tiles -- two per CU -- where 256 x 256 leaves 384, a round and a half.
template <bool F8, int TM, int TN> struct LnOpsOf { using type = g8p::NtOps<TM, TN>; };
template <int TM, int TN> struct LnOpsOf<true, TM, TN> { using type = g8p::NtOpsF8<TM, TN>; };

template <int MODE, int DROPW, bool F8 = false, int TM = 8, int TN = 4>
__global__ __launch_bounds__(512, 1) void gemm_nt8_ln_kernel(QstGemmArgs g, QstLnEpi e, LnXchg x, int stagger) {
    static_assert(!F8 || MODE == 0, "the fp8 form is a forward kernel");
    static_assert(TM % 4 == 0 && TN % 2 == 0, "tile geometry");
    constexpr int BM = 32 * TM, BN = 64 * TN, NP = TN / 2, WC = BN / 4, KR = TM / 4;   // WC: columns per wave; KR: rows per polling lane
    constexpr int RED_B = BM * 4 * 2 * 4, STATS_B = BM * 2 * 4;
    op_saturate(MODE == 0);
    if (blockIdx.x < 256) stagger_start(stagger, (blockIdx.x & 7) * 32 + (blockIdx.x >> 3) / x.ntn);   // one phase per row panel
    using OPS = typename LnOpsOf<F8, TM, TN>::type;
    extern __shared__ __attribute__((aligned(16))) char smem[];
    const int bx = blockIdx.x & 7, bj = blockIdx.x >> 3;
    const int panel = bx * x.ppx + bj / x.ntn, tile_n = bj % x.ntn;
    if (bj / x.ntn >= x.ppx || panel >= x.ntm) { lnx_finish(x.ctl); return; }
    const int m0 = panel * BM, n0 = tile_n * BN;
    OPS o;
    if constexpr (F8)
        o.init((const uint8_t*)g.A + (size_t)m0 * g.lda, g.lda, min(BM, g.M - m0), (const uint8_t*)g.aux, g.M, m0,
               (const uint8_t*)g.B + (size_t)n0 * g.ldb, g.ldb, BN, (const uint8_t*)g.bscale, g.N, n0, g.K, smem);
    else
        o.init((const op16*)g.A + (size_t)m0 * g.lda, g.lda, min(BM, g.M - m0), (const op16*)g.B + (size_t)n0 * g.ldb, g.ldb,
               BN, g.K, smem);
    g8p::kloop8(o, o.nk);                       // returns behind a workgroup barrier, no DMA outstanding: the LDS is free

    const int tid = threadIdx.x, lane = tid & 63, gq = lane >> 4, c16 = lane & 15;
    // the tag of this launch's granules = the buffer's epoch + 1 (0 = never written); requested now, used after pass 1
    unsigned tag = 0;
    if (o.wc == 0) tag = __hip_atomic_load((gu32*)x.ctl, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) + 1u;
    const int rw = o.wr * (BM / 2) + c16;       // + 16 i: this lane's rows inside the panel
    const int mw = m0 + rw;
    const int nw = n0 + o.wc * WC + pair_col(gq);      // + 32 jp: 8 consecutive columns
    char* scratch = smem + (MODE == 1 ? g8p::LDS_BYTES : 0);
    float* red = (float*)scratch;                        // [BM rows][4 wave columns][2]
    float* stats = (float*)(scratch + RED_B);            // [BM rows][2]
    const f32x4 z4 = {0.f, 0.f, 0.f, 0.f};
    DropCtx dc = DropCtx{0u, 0u, 1.f};
    if (DROPW != 0) dc = drop_ctx(g.drop);

    // ------------------------------------------------------------ pass 1
    // per row-tile: mode 0: mean and centred squares of the wave's 64 columns; mode 1: sum(g dy), sum(g dy xhat) -> LDS
    {
        f32x4 ga[NP][2];                        // mode 1: gamma of this lane's columns
#pragma unroll
        for (int jp = 0; jp < NP; ++jp) {
            ga[jp][0] = MODE == 1 ? *(const f32x4*)(e.gamma + nw + 32 * jp) : z4;
            ga[jp][1] = MODE == 1 ? *(const f32x4*)(e.gamma + nw + 32 * jp + 4) : z4;
        }
        f32x4 bv[NP][2];
#pragma unroll
        for (int jp = 0; jp < NP; ++jp) {
            const bool ok = MODE == 0 && g.bias != nullptr;
            bv[jp][0] = ok ? *(const f32x4*)(g.bias + nw + 32 * jp) : z4;
            bv[jp][1] = ok ? *(const f32x4*)(g.bias + nw + 32 * jp + 4) : z4;
        }
        constexpr int HT = NP == 2 ? 4 : 2;     // row-tiles whose residual rows are requested in one burst (HT NP 8 registers)
        f32x4 rv[HT][NP][2];
        if (MODE == 1) {
            // the tile's xhat fragment goes straight into its lane-private LDS slots (the 128 KB the K loop has left), all 16
            // LDS-DMAs of the wave at once, ahead of the residual bursts: no registers, one round trip for the whole tile
            // (rows past M lie beyond the descriptor: zero fill)
            const int rows_here = min(BM, g.M - m0);
            const __amdgpu_buffer_rsrc_t rx = g8p::rsrc((const op16*)e.xhat + (size_t)m0 * g.N, (uint32_t)rows_here * (uint32_t)g.N * 2u);
            const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
#pragma unroll
            for (int i = 0; i < TM; ++i)
#pragma unroll
                for (int jp = 0; jp < NP; ++jp)
                    g8p::dma16(rx, smem + ((size_t)(i * NP + jp) * 512 + wave * 64) * 16,
                               (uint32_t)(((rw + 16 * i) * g.N + (nw + 32 * jp)) * 2), 0u);
        }
#pragma unroll
        for (int i = 0; i < TM; ++i) {
            if (i % HT == 0) {
#pragma unroll
                for (int k = 0; k < HT; ++k) {
                    const int m = mw + 16 * (i + k);
#pragma unroll
                    for (int jp = 0; jp < NP; ++jp) {
                        const int n = nw + 32 * jp;
                        const bool ok = g.resid != nullptr && m < g.M;
                        const float* p = g.resid + (size_t)m * g.ldr + n;
                        // (contiguous form: the lane fetches columns 4 r .. and 16 + 4 r .. of the 32-column group -- 64 contiguous
                        //  bytes per row and instruction -- and trades halves back where the rows are used; see store_f32_row8)
                        const float* p0 = x.frag ? p : p - pair_col(gq) + 4 * gq;
                        const float* p1 = x.frag ? p + 4 : p - pair_col(gq) + 16 + 4 * gq;
                        rv[k][jp][0] = ok ? ld_stream((const f32x4*)p0) : z4;
                        rv[k][jp][1] = ok ? ld_stream((const f32x4*)p1) : z4;
                    }
                }
                if (MODE == 1 && i == 0) asm volatile("s_waitcnt vmcnt(0)" ::: "memory");   // the xhat DMAs (and this first burst) have landed
            }
            const int m = mw + 16 * i;
            float s1 = 0.f, s2 = 0.f;
#pragma unroll
            for (int jp = 0; jp < NP; ++jp) {
                const int n = nw + 32 * jp;
                float v[8];
                pair8(o.acc[i][2 * jp], o.acc[i][2 * jp + 1], v);
#pragma unroll
                for (int q = 0; q < 4; ++q) { v[q] += bv[jp][0][q]; v[4 + q] += bv[jp][1][q]; }
                if (MODE == 0 && DROPW == 1 && dc.thr) {
                    const uint32_t e0 = (uint32_t)m * (uint32_t)g.N + (uint32_t)n;
#pragma unroll
                    for (int q = 0; q < 8; q += 2) {
                        float k0, k1;
                        drop_pair(dc, e0 + q, k0, k1);
                        v[q] *= k0; v[q + 1] *= k1;
                    }
                }
#pragma unroll
                for (int q = 0; q < 4; ++q) {
                    float r0 = rv[i % HT][jp][0][q], r1 = rv[i % HT][jp][1][q];
                    if (!x.frag) asm volatile("s_nop 1\n\tv_permlane16_swap_b32 %0, %1" : "+v"(r0), "+v"(r1));     // uniform over the launch
                    v[q] += r0; v[4 + q] += r1;
                }
                if (MODE == 1 && DROPW == 3 && dc.thr) {
                    const uint32_t e0 = (uint32_t)m * (uint32_t)g.N + (uint32_t)n;
#pragma unroll
                    for (int q = 0; q < 8; q += 2) {
                        float k0, k1;
                        drop_pair(dc, e0 + q, k0, k1);
                        v[q] *= k0; v[q + 1] *= k1;
                    }
                }
                if (MODE == 0) {
#pragma unroll
                    for (int q = 0; q < 8; ++q) s1 += v[q];
                } else {
                    const u32x4 xq = *(const u32x4*)(smem + ((size_t)(i * NP + jp) * 512 + tid) * 16);   // this lane's slot (stays for pass 2)
#pragma unroll
                    for (int q = 0; q < 4; ++q) {
                        const float g0 = v[2 * q] * ga[jp][q >> 1][(2 * q) & 3], g1 = v[2 * q + 1] * ga[jp][q >> 1][(2 * q + 1) & 3];
                        s1 += g0 + g1;
                        s2 += g0 * op_lo(xq[q]) + g1 * op_hi(xq[q]);
                    }
                }
#pragma unroll
                for (int q = 0; q < 4; ++q) { o.acc[i][2 * jp][q] = v[q]; o.acc[i][2 * jp + 1][q] = v[4 + q]; }
            }
            s1 = xg_sum(s1);
            if (MODE == 0) {
                const float mu = s1 * (1.f / (float)WC);
                float q2 = 0.f;
#pragma unroll
                for (int j = 0; j < TN; ++j)
#pragma unroll
                    for (int q = 0; q < 4; ++q) { const float d = o.acc[i][j][q] - mu; q2 += d * d; }
                s1 = mu;
                s2 = xg_sum(q2);
            } else {
                s2 = xg_sum(s2);
            }
            if (gq == 0) {
                f32x2 pr; pr[0] = s1; pr[1] = s2;
                *(f32x2*)(red + ((rw + 16 * i) * 4 + o.wc) * 2) = pr;
            }
        }
    }
    __syncthreads();

    // ------------------------------------------------------------ exchange (waves (wr, 0): rows wr BM/2 + [0, BM/2), KR per lane)
    if (o.wc == 0) {
        float ta[KR], tb[KR];                   // this tile's pair for the lane's rows
        int row[KR];
#pragma unroll
        for (int k = 0; k < KR; ++k) {
            row[k] = o.wr * (BM / 2) + 16 * (KR * gq + k) + c16;
            const f32x4 p0 = *(const f32x4*)(red + row[k] * 8), p1 = *(const f32x4*)(red + row[k] * 8 + 4);
            if (MODE == 0) {
                const float mu = (p0[0] + p0[2] + p1[0] + p1[2]) * 0.25f;
                const float d0 = p0[0] - mu, d1 = p0[2] - mu, d2 = p1[0] - mu, d3 = p1[2] - mu;
                ta[k] = mu;
                tb[k] = (p0[1] + p0[3] + p1[1] + p1[3]) + (float)WC * (d0 * d0 + d1 * d1 + d2 * d2 + d3 * d3);
            } else {
                ta[k] = p0[0] + p0[2] + p1[0] + p1[2];
                tb[k] = p0[1] + p0[3] + p1[1] + p1[3];
            }
            unsigned long long* gp = x.gran + (((size_t)panel * x.ntn + tile_n) * BM + row[k]) * 2;
            put_granule(gp, tag, ta[k]);
            put_granule(gp + 1, tag, tb[k]);
        }
        float oa[KR][4], ob[KR][4];             // every tile's pair (ntn <= 4)
        bool done = false;
#pragma unroll 1
        for (unsigned spins = 0; !done; ++spins) {
            bool ok = true;
#pragma unroll
            for (int k = 0; k < KR; ++k)
#pragma unroll
                for (int t = 0; t < 4; ++t) {
                    if (t >= x.ntn) continue;
                    if (t == tile_n) { oa[k][t] = ta[k]; ob[k][t] = tb[k]; continue; }
                    const gu64* gp = (const gu64*)(x.gran + (((size_t)panel * x.ntn + t) * BM + row[k]) * 2);
                    const unsigned long long u0 = __hip_atomic_load(gp, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
                    const unsigned long long u1 = __hip_atomic_load(gp + 1, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
                    ok = ok && (unsigned)(u0 >> 32) == tag && (unsigned)(u1 >> 32) == tag;
                    oa[k][t] = __builtin_bit_cast(float, (uint32_t)u0);
                    ob[k][t] = __builtin_bit_cast(float, (uint32_t)u1);
                }
            done = __all(ok) != 0;
            if (!done) {
                if (spins > (1u << 18)) {       // >= 100 ms of polling: report and finish (the rows of this tile are wrong)
                    if (lane == 0) atomicOr(x.tmo, 1u);
                    done = true;
                }
                __builtin_amdgcn_s_sleep(8);
            }
        }
#pragma unroll
        for (int k = 0; k < KR; ++k) {
            f32x2 st;
            if (MODE == 0) {
                float mu = 0.f;
#pragma unroll
                for (int t = 0; t < 4; ++t) if (t < x.ntn) mu += oa[k][t];
                mu /= (float)x.ntn;
                float m2 = 0.f;
#pragma unroll
                for (int t = 0; t < 4; ++t) if (t < x.ntn) { const float d = oa[k][t] - mu; m2 += ob[k][t] + (float)BN * d * d; }
                st[0] = mu;
                st[1] = rsqrtf(m2 / (float)g.N + e.eps);              // biased variance, as nn.LayerNorm
            } else {
                float a1 = 0.f, a2 = 0.f;
#pragma unroll
                for (int t = 0; t < 4; ++t) if (t < x.ntn) { a1 += oa[k][t]; a2 += ob[k][t]; }
                st[0] = a1 / (float)g.N;
                st[1] = a2 / (float)g.N;
            }
            *(f32x2*)(stats + row[k] * 2) = st;
        }
    }
    __syncthreads();

    // ------------------------------------------------------------ pass 2 (column pair jp outermost: half the per-column constants live)
    if (MODE == 0) {
#pragma unroll
        for (int jp = 0; jp < NP; ++jp) {
            const int n = nw + 32 * jp;
            const f32x4 ga0 = *(const f32x4*)(e.gamma + n), ga1 = *(const f32x4*)(e.gamma + n + 4);
            const f32x4 be0 = *(const f32x4*)(e.beta + n), be1 = *(const f32x4*)(e.beta + n + 4);
#pragma unroll
            for (int i = 0; i < TM; ++i) {
                const int m = mw + 16 * i;
                const f32x2 st = *(const f32x2*)(stats + (rw + 16 * i) * 2);
                if (m >= g.M) continue;
                if (jp == 0 && tile_n == 0 && o.wc == 0 && gq == 0 && e.rstd) e.rstd[m] = st[1];
                f32x4 lo, hi, hl, hh;
#pragma unroll
                for (int q = 0; q < 4; ++q) {
                    hl[q] = (o.acc[i][2 * jp][q] - st[0]) * st[1];
                    hh[q] = (o.acc[i][2 * jp + 1][q] - st[0]) * st[1];
                    lo[q] = hl[q] * ga0[q] + be0[q];
                    hi[q] = hh[q] * ga1[q] + be1[q];
                }
                const size_t off = (size_t)m * g.ldc + n;
                store_f32_row8((float*)g.C + (off - pair_col(gq)), gq, lo, hi, x.frag != 0);
                if (g.C2) {
                    u32x4 pk;
                    pk[0] = pack_op2(lo[0], lo[1]); pk[1] = pack_op2(lo[2], lo[3]);
                    pk[2] = pack_op2(hi[0], hi[1]); pk[3] = pack_op2(hi[2], hi[3]);
                    st_stream((u32x4*)((op16*)g.C2 + off), pk);
                }
                if (e.xhat) {
                    u32x4 pk;
                    pk[0] = pack_op2(hl[0], hl[1]); pk[1] = pack_op2(hl[2], hl[3]);
                    pk[2] = pack_op2(hh[0], hh[1]); pk[3] = pack_op2(hh[2], hh[3]);
                    st_stream((u32x4*)((op16*)e.xhat + (size_t)m * g.N + n), pk);
                }
                if (F8 && g.C3) {
                    // the 16-bit-rounded y as MXFP8: the four lanes of a row (one per 16-lane row of the wave) hold the 32
                    // columns of one MX block (all four take this branch together: same m, same block)
                    float hv[8], amax = 0.f;
#pragma unroll
                    for (int q = 0; q < 2; ++q) {
                        const uint32_t p0 = pack_op2(lo[2 * q], lo[2 * q + 1]), p1 = pack_op2(hi[2 * q], hi[2 * q + 1]);
                        hv[2 * q] = op_lo(p0); hv[2 * q + 1] = op_hi(p0);
                        hv[4 + 2 * q] = op_lo(p1); hv[4 + 2 * q + 1] = op_hi(p1);
                    }
#pragma unroll
                    for (int q = 0; q < 8; ++q) amax = fmaxf(amax, fabsf(hv[q]));
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
                    st_stream((u32x2*)((uint8_t*)g.C3 + (size_t)m * g.N + n), q2);
                    if (gq == 0) ((uint8_t*)g.C4)[((size_t)(n >> 7) * g.M + m) * 4 + ((n >> 5) & 3)] = (uint8_t)(ex + 127);
                }
            }
        }
    } else {
        float* pr = (float*)(scratch + RED_B + STATS_B);               // [2 wave rows][2][BN columns]
#pragma unroll
        for (int jp = 0; jp < NP; ++jp) {
            const int n = nw + 32 * jp;
            const f32x4 ga0 = *(const f32x4*)(e.gamma + n), ga1 = *(const f32x4*)(e.gamma + n + 4);
            float ag[8], ab[8];
#pragma unroll
            for (int q = 0; q < 8; ++q) { ag[q] = 0.f; ab[q] = 0.f; }
#pragma unroll
            for (int i = 0; i < TM; ++i) {
                const int m = mw + 16 * i;
                const f32x2 st = *(const f32x2*)(stats + (rw + 16 * i) * 2);
                const float rs = m < g.M ? e.rstd[m] : 0.f;
                const u32x4 xq = *(const u32x4*)(smem + ((size_t)(i * NP + jp) * 512 + tid) * 16);
                float ov[8];
#pragma unroll
                for (int q = 0; q < 8; ++q) {
                    const float dy = (q < 4) ? o.acc[i][2 * jp][q & 3] : o.acc[i][2 * jp + 1][q & 3];
                    const float xh = (q & 1) ? op_hi(xq[q >> 1]) : op_lo(xq[q >> 1]);
                    const float gm = (q < 4) ? ga0[q & 3] : ga1[q & 3];
                    ag[q] += dy * xh;                                // rows past M carry dy = 0
                    ab[q] += dy;
                    ov[q] = rs * (dy * gm - st[0] - xh * st[1]);
                }
                if (m >= g.M) continue;
                const size_t off = (size_t)m * g.ldc + n;
                f32x4 lo, hi;
#pragma unroll
                for (int q = 0; q < 4; ++q) { lo[q] = ov[q]; hi[q] = ov[4 + q]; }
                store_f32_row8((float*)g.C + (off - pair_col(gq)), gq, lo, hi, x.frag != 0);
                if (g.C2) {
                    if (DROPW == 2 && dc.thr) {
                        const uint32_t e0 = (uint32_t)m * (uint32_t)g.N + (uint32_t)n;
#pragma unroll
                        for (int q = 0; q < 8; q += 2) {
                            float k0, k1;
                            drop_pair(dc, e0 + q, k0, k1);
                            ov[q] *= k0; ov[q + 1] *= k1;
                        }
                    }
                    u32x4 pk;
#pragma unroll
                    for (int q = 0; q < 4; ++q) pk[q] = pack_op2(ov[2 * q], ov[2 * q + 1]);
                    st_stream((u32x4*)((op16*)g.C2 + off), pk);
                }
            }
            if (e.partials) {
                // column sums over the panel's 256 rows: over the 16 rows of a lane row by DPP, over the two wave rows through LDS
#pragma unroll
                for (int q = 0; q < 8; ++q) {
                    const float a = row16_sum(ag[q]), b = row16_sum(ab[q]);
                    if (c16 == 0) {
                        const int col = o.wc * WC + pair_col(gq) + 32 * jp + q;
                        pr[(o.wr * 2 + 0) * BN + col] = a;
                        pr[(o.wr * 2 + 1) * BN + col] = b;
                    }
                }
            }
        }
        if (e.partials) {
            __syncthreads();
            for (int idx = tid; idx < 2 * BN; idx += 512) {
                const int which = idx / BN, col = idx - which * BN;
                e.partials[((size_t)panel * 2 + which) * g.N + n0 + col] = pr[which * BN + col] + pr[(2 + which) * BN + col];
            }
        }
    }
    lnx_finish(x.ctl);
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
std::atomic<int> g_stagger{-1};    // qst_gemm8_stagger (-1: the library's defaults)
std::atomic<int> g_ln_store{-1};   // qst_gemm8_ln_store (-1: by tile)
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
// cycles over which the first round of an 8-phase NT launch spreads its start (0 = all together; -1 = the library's defaults:
// 0 for the plain GEMMs, whose workgroups share operand rows through L2 while they run in step, kLnStagger for the GEMM +
// LayerNorm launches); returns the previous value, an argument below -1 only reads. Process-wide.
// Measured at M = 196,608 (tools/stagger_bench.py, us, spread 0 / 20k / 40k / 80k / 160k cycles): QKV 709 / 722 / 737 / 742 /
// 769, FFN-1 + GELU 1,246 / 1,315 / 1,345 / 1,365 / 1,381, GELU' dgrad 1,151 / 1,155 / 1,153 / 1,173 / 1,186; FFN-2 +
// LayerNorm 1,097 / 1,068 / 1,071 / 1,099 / 1,116, out-projection + LayerNorm 615 / 601 / 583 / 580 / 597, FFN-1 dgrad +
// LayerNorm' 1,104 / 1,077 / 1,084 / 1,087 / 1,109.
extern "C" int qst_gemm8_stagger(int cycles) {
    const int old = g_stagger.load();
    if (cycles >= -1) g_stagger.store(cycles);
    return old;
}
int qst_gemm8_stagger_get() { return g_stagger.load(); }
// How the GEMM + LayerNorm launches store their fp32 rows: 0 = 64 contiguous bytes per row and instruction (halves traded
// between lanes first), 1 = each lane's own two 16-byte halves (fragments with 16-byte holes, filled by the second
// instruction), -1 (default) = by tile. Returns the previous value; an argument below -1 only reads. Process-wide.
extern "C" int qst_gemm8_ln_store(int mode) {
    const int old = g_ln_store.load();
    if (mode >= -1 && mode <= 1) g_ln_store.store(mode);
    return old;
}
int qst_gemm8_ln_store_get() { return g_ln_store.load(); }
#else
int qst_gemm8_mode_get();
int qst_gemm8_stagger_get();
int qst_gemm8_ln_store_get();
#endif

template <int EPI, int TM, int TN>
static int launch_nt8(const QstGemmArgs* a, hipStream_t st) {
    using OPS = g8p::NtOps<TM, TN>;
    static QstLdsAttr attr;
    if (int rc = qst_ensure_lds(attr, (const void*)gemm_nt8_kernel<EPI, TM, TN>, g8p::LDS_BYTES)) return rc;
    const int ntm = (a->M + OPS::BM - 1) / OPS::BM, ntn = (a->N + OPS::BN - 1) / OPS::BN;
    gemm_nt8_kernel<EPI, TM, TN><<<dim3(ntm * ntn), dim3(512), g8p::LDS_BYTES, st>>>(*a, ntm * ntn > 512 && qst_gemm8_stagger_get() > 0 ? qst_gemm8_stagger_get() : 0);
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

// The granules of the row-statistics exchange: one buffer (with its epoch) per stream that has launched the kernel -- two
// launches in flight on different streams must not share one -- grown on demand.
namespace {
struct LnxBuf { int dev; hipStream_t st; unsigned long long* gran; size_t bytes; bool used; };     // gran[0 .. 31] = control block
LnxBuf g_lnx[16];
unsigned* g_lnx_tmo[16] = {};        // per device
constexpr size_t kLnxMinBytes = (size_t)16 << 20;      // covers 262,144 token rows at H = 1024: regrowth is the exception
}  // namespace

static int lnx_get(hipStream_t st, size_t bytes, LnXchg& x) {
    int dev = 0;
    QST_HIP_CHECK(hipGetDevice(&dev));
    if (dev < 0 || dev >= 16) return QST_ERR_UNSUPPORTED;
    hipStreamCaptureStatus cap = hipStreamCaptureStatusNone;
    QST_HIP_CHECK(hipStreamIsCapturing(st, &cap));
    const bool capturing = cap != hipStreamCaptureStatusNone;
    if (!g_lnx_tmo[dev]) {
        if (capturing) return QST_ERR_UNSUPPORTED;       // no eager launch on this device before the capture
        QST_HIP_CHECK(hipMalloc((void**)&g_lnx_tmo[dev], 256));
        QST_HIP_CHECK(hipMemset(g_lnx_tmo[dev], 0, 256));
    }
    LnxBuf* b = nullptr;
    for (auto& q : g_lnx) if (q.used && q.dev == dev && q.st == st) { b = &q; break; }
    if (capturing && (!b || b->bytes < bytes)) {
        // a capturing stream cannot allocate: the launch being recorded takes the buffer an eager launch of this size has
        // left (the warm-up step every capture follows); a graph and eager launches must then not run side by side
        b = nullptr;
        for (auto& q : g_lnx) if (q.used && q.dev == dev && q.bytes >= bytes && (!b || q.bytes < b->bytes)) b = &q;
        if (!b) return QST_ERR_UNSUPPORTED;              // no eager launch of this size before the capture
    } else {
        if (!b) for (auto& q : g_lnx) if (!q.used) { b = &q; *b = LnxBuf{dev, st, nullptr, 0, true}; break; }
        if (!b) return QST_ERR_UNSUPPORTED;              // more than sixteen (device, stream) pairs in one process
        if (b->bytes < bytes) {
            // (a buffer that is outgrown is NOT freed: a captured graph may hold its address; it is 16 MB or a few times that)
            const size_t want = bytes > kLnxMinBytes ? bytes : kLnxMinBytes;
            unsigned long long* p = nullptr;
            QST_HIP_CHECK(hipMalloc((void**)&p, want + 256));
            QST_HIP_CHECK(hipMemset(p, 0, want + 256));  // epoch 0, no tag set; never zeroed again (tags carry the epoch)
            b->gran = p; b->bytes = want;
        }
    }
    x.gran = b->gran + 32; x.ctl = (unsigned*)b->gran; x.tmo = g_lnx_tmo[dev];
    return QST_OK;
}

// 1 when qst_gemm_nt_ln can take N on this kernel (whole 256-column tiles, at most four per row panel)
extern "C" int QST_K(qst_gemm_nt8_ln_supported)(int N) { return (N % 256 == 0 && N >= 512 && N <= 1024) ? 1 : 0; }
// the sticky timeout word of the exchange on the current device (0 = no launch of this process has ever given up waiting);
// reads synchronously
extern "C" int QST_K(qst_gemm_nt8_ln_timeouts)(void) {
    unsigned v = 0;
    int dev = 0;
    if (hipGetDevice(&dev) != hipSuccess || dev < 0 || dev >= 16) return -1;
    if (g_lnx_tmo[dev] && hipMemcpy(&v, g_lnx_tmo[dev], 4, hipMemcpyDeviceToHost) != hipSuccess) return -1;
    return (int)v;
}

constexpr int kLnStagger = 40000;     // cycles; see qst_gemm8_stagger
// Which tile a (M, N) problem takes: 0 = 256 x 256, 1 = 128 x 384. The first from two tiles per CU on (and wherever 384 does not
// divide N); the second where it -- and not the first -- gives two tiles per CU (N = 768: 32,768 <= M < 43,691).
static int ln8_tile(int M, int N) {
    static const int forced = [] { const char* e = getenv("QST_LN8_TILE"); return e ? atoi(e) : -1; }();   // tools only: 0 / 1 where both divide N
    if (forced >= 0 && N % 256 == 0 && N % 384 == 0) return forced ? 1 : 0;
    const int64_t t256 = (int64_t)((M + 255) / 256) * (N / 256);
    if (N % 256 == 0 && (t256 >= 512 || N % 384 != 0)) return 0;
    const int64_t t384 = (int64_t)((M + 127) / 128) * (N / 384);
    return (N % 384 == 0 && (t384 >= 512 || N % 256 != 0)) ? 1 : 0;
}
extern "C" int QST_K(qst_gemm_nt8_ln_block_rows)(int M, int N) { return ln8_tile(M, N) ? 128 : 256; }

template <int MODE, int DROPW, bool F8 = false, int TM = 8, int TN = 4>
static int launch_nt8_ln(const QstGemmArgs* a, const QstLnEpi* ln, hipStream_t st) {
    constexpr int BM = 32 * TM, BN = 64 * TN;
    constexpr int lds = F8 ? (int)g8p::NtOpsF8<8, 4>::LDS_TOTAL : (MODE == 0 ? LNX_LDS0 : LNX_LDS1);
    static QstLdsAttr attr;
    if (int rc = qst_ensure_lds(attr, (const void*)gemm_nt8_ln_kernel<MODE, DROPW, F8, TM, TN>, lds)) return rc;
    LnXchg x{};
    x.ntm = (a->M + BM - 1) / BM; x.ntn = a->N / BN; x.ppx = (x.ntm + 7) / 8;
    const size_t bytes = (size_t)x.ntm * x.ntn * BM * 2 * sizeof(unsigned long long);
    if (int rc = lnx_get(st, bytes, x)) return rc;
    x.frag = qst_gemm8_ln_store_get() < 0 ? (TM == 8 ? 1 : 0) : qst_gemm8_ln_store_get();      // by tile: see qst_gemm8_ln_store
    gemm_nt8_ln_kernel<MODE, DROPW, F8, TM, TN><<<dim3(8 * x.ntn * x.ppx), dim3(512), lds, st>>>(*a, *ln, x, x.ntm * x.ntn > 512 ? (qst_gemm8_stagger_get() < 0 ? kLnStagger : qst_gemm8_stagger_get()) : 0);
    QST_LAUNCH_CHECK();
    return QST_OK;
}

// Arguments, modes and dropout sites as qst_gemm_nt_ln (which forwards here for N = 512 / 768 / 1024); partials are
// f32 [ceil(M/256)][2][N].
extern "C" int QST_K(qst_gemm_nt8_ln)(const QstGemmArgs* a, const QstLnEpi* ln, int mode, void* stream) {
    if (!a || !ln || !a->A || !a->B || !a->C || !ln->gamma || a->M <= 0 || a->K <= 0) return QST_ERR_BAD_ARG;
    if (mode != 0 && mode != 1) return QST_ERR_BAD_ARG;
    if (mode == 0 && !ln->beta) return QST_ERR_BAD_ARG;
    if (mode == 1 && (!ln->xhat || !ln->rstd)) return QST_ERR_BAD_ARG;
    // (N = 384 -- one 128 x 384 tile per row panel, no partner -- is accepted here for tools/ln8_n384_bench.py; the library takes
    //  gemm_nt_ln_kernel there)
    if ((!QST_K(qst_gemm_nt8_ln_supported)(a->N) && a->N != 384) || a->B2) return QST_ERR_UNSUPPORTED;
    if (a->K % 64 != 0 || a->lda % 8 != 0 || a->ldb % 8 != 0 || a->ldc % 8 != 0 || (a->resid && a->ldr % 4 != 0)) return QST_ERR_UNSUPPORTED;
    if ((int64_t)256 * a->lda * 2 >= 0x7FFFFF00LL || (int64_t)384 * a->ldb * 2 >= 0x7FFFFF00LL) return QST_ERR_UNSUPPORTED;
    const bool drop = a->drop.thr16 && a->drop.state;
    if (drop) {
        if (a->drop.thr16 > 65535u || (int64_t)a->M * a->N >= ((int64_t)1 << 32)) return QST_ERR_UNSUPPORTED;
        if (mode == 0 ? a->drop_where != 1 : (a->drop_where != 2 && a->drop_where != 3)) return QST_ERR_BAD_ARG;
    }
    hipStream_t st = (hipStream_t)stream;
    if (ln8_tile(a->M, a->N)) {
        if (mode == 0) return drop ? launch_nt8_ln<0, 1, false, 4, 6>(a, ln, st) : launch_nt8_ln<0, 0, false, 4, 6>(a, ln, st);
        if (!drop) return launch_nt8_ln<1, 0, false, 4, 6>(a, ln, st);
        return a->drop_where == 2 ? launch_nt8_ln<1, 2, false, 4, 6>(a, ln, st) : launch_nt8_ln<1, 3, false, 4, 6>(a, ln, st);
    }
    if (mode == 0) return drop ? launch_nt8_ln<0, 1>(a, ln, st) : launch_nt8_ln<0, 0>(a, ln, st);
    if (!drop) return launch_nt8_ln<1, 0>(a, ln, st);
    return a->drop_where == 2 ? launch_nt8_ln<1, 2>(a, ln, st) : launch_nt8_ln<1, 3>(a, ln, st);
}

#if !QST_OP_F16
// The forward of the same fusion on the fp8 matrix cores: operands as qst_gemm_nt_f8 (A, B = e4m3, a->aux / a->bscale their
// E8M0 scales, K % 128 == 0); outputs as mode 0 of qst_gemm_nt8_ln plus, when a->C3 / a->C4 are given, the normalised rows
// as MXFP8 (e4m3 [M, N] + scales in qst_quant_mx's layout, quantised from the 16-bit-rounded values as qst_ln_fwd_mx_train
// does) -- the A operand of the next fp8 GEMM. Dropout of the projection output as mode 0 (drop_where 1).
extern "C" int qst_gemm_nt8_f8_ln(const QstGemmArgs* a, const QstLnEpi* ln, void* stream) {
    if (!a || !ln || !a->A || !a->B || !a->C || !a->aux || !a->bscale || !ln->gamma || !ln->beta || a->M <= 0 || a->K <= 0) return QST_ERR_BAD_ARG;
    if ((a->C3 == nullptr) != (a->C4 == nullptr)) return QST_ERR_BAD_ARG;
    if (!qst_gemm_nt8_ln_supported(a->N) || a->B2) return QST_ERR_UNSUPPORTED;
    if (a->K % 128 != 0 || a->lda % 16 != 0 || a->ldb % 16 != 0 || a->ldc != a->N || (a->resid && a->ldr % 4 != 0)) return QST_ERR_UNSUPPORTED;
    if ((int64_t)256 * a->lda >= 0x7FFFFF00LL || (int64_t)256 * a->ldb >= 0x7FFFFF00LL) return QST_ERR_UNSUPPORTED;
    if ((int64_t)(a->K / 128) * (a->M > a->N ? a->M : a->N) * 4 >= 0x7FFFFF00LL) return QST_ERR_UNSUPPORTED;
    const bool drop = a->drop.thr16 && a->drop.state;
    if (drop) {
        if (a->drop.thr16 > 65535u || (int64_t)a->M * a->N >= ((int64_t)1 << 32)) return QST_ERR_UNSUPPORTED;
        if (a->drop_where != 1) return QST_ERR_BAD_ARG;
    }
    hipStream_t st = (hipStream_t)stream;
    return drop ? launch_nt8_ln<0, 1, true>(a, ln, st) : launch_nt8_ln<0, 0, true>(a, ln, st);
}

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
