// x3_bwd.hip -- the pieces the parity-precision BACKWARD (QST_PREC_BF16X3 training) needs beside gemm_nt_x3.
//
// The reference trains in fp32 (training/main.py:142, use_amp=False); the bf16 path's gradients agree with fp32 autograd
// to 1-4e-2 relative L2 (operand rounding), this path to fp32-class. It is the parity path, not the fast one: every
// contraction goes through the split-bf16 x3 GEMM (dgrad: against a transposed copy of the weight; wgrad: dW = dY^T X as an
// NT product of transposed copies of dY and X, reduction over the token rows), every row kernel works on fp32 and
// recomputes the LayerNorm statistics from the saved pre-norm tensor, and the attention backward is a plain fp32 kernel
// (one workgroup per (sequence, head), L <= 512) that recomputes the probabilities row by row.
//
//   qst_transpose_f32    : dst[C, R] = src[R, C]^T
//   qst_gelu_f32         : h = gelu(u) (exact erf), forward; qst_gelu_bwd_f32: du = dh * gelu'(u)
//   qst_colsum_f32       : out[n] += sum_m x[m, n]                     (bias gradients)
//   qst_embed_sum_f32    : s[m, :] = word[ids[m]] + type[...] + pos[...]  (the embedding LayerNorm's pre-norm input)
//   qst_ln_bwd_f32       : LayerNorm backward from the PRE-NORM tensor (mean / rstd recomputed): ds, dgamma +=, dbeta +=
//   qst_attention_bwd_f32: dqkv (and the [A, L, L] position-bias gradient) from fp32 qkv, ctx and dctx
#include "qst_common.h"
#include "qst_kernels.h"

namespace {

__global__ __launch_bounds__(256) void transpose_f32_kernel(const float* src, int R, int C, int lds_, float* dst, int ldd) {
    __shared__ float tile[32][33];
    const int c0 = blockIdx.x * 32, r0 = blockIdx.y * 32;
    const int tx = threadIdx.x & 31, ty = threadIdx.x >> 5;          // 32 x 8
#pragma unroll
    for (int k = 0; k < 4; ++k) {
        const int r = r0 + ty + 8 * k, c = c0 + tx;
        tile[ty + 8 * k][tx] = (r < R && c < C) ? src[(size_t)r * lds_ + c] : 0.f;
    }
    __syncthreads();
#pragma unroll
    for (int k = 0; k < 4; ++k) {
        const int c = c0 + ty + 8 * k, r = r0 + tx;
        if (c < C && r < R) dst[(size_t)c * ldd + r] = tile[tx][ty + 8 * k];
    }
}

__device__ __forceinline__ float gelu_exact(float x) { return 0.5f * x * (1.0f + erff(x * 0.70710678118654752f)); }
__device__ __forceinline__ float gelu_exact_grad(float x) {
    return 0.5f * (1.0f + erff(x * 0.70710678118654752f)) + x * 0.3989422804014327f * expf(-0.5f * x * x);
}
__global__ __launch_bounds__(256) void gelu_f32_kernel(const float* u, int64_t n, float* h) {
    const int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x;
    if (i < n) h[i] = gelu_exact(u[i]);
}
__global__ __launch_bounds__(256) void gelu_bwd_f32_kernel(const float* dh, const float* u, int64_t n, float* du) {
    const int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x;
    if (i < n) du[i] = dh[i] * gelu_exact_grad(u[i]);
}

// bias gradients: a workgroup sums CS_ROWS rows of every column (thread t: columns 4t .. 4t+3 of each 1024-column panel, rows
// in order, whole 16-byte loads) and adds its partial sums to out[] -- M / CS_ROWS workgroups instead of N / 64
constexpr int CS_ROWS = 64;
__global__ __launch_bounds__(256) void colsum_f32_kernel(const float* x, int M, int N, int ld, float* out) {
    const int r0 = blockIdx.x * CS_ROWS, r1 = min(M, r0 + CS_ROWS);
    for (int c = threadIdx.x * 4; c < N; c += 1024) {
        if (c + 3 < N && (ld & 3) == 0 && ((uintptr_t)x & 15) == 0) {
            f32x4 s = {0.f, 0.f, 0.f, 0.f};
            for (int m = r0; m < r1; ++m) s += *(const f32x4*)(x + (size_t)m * ld + c);
#pragma unroll
            for (int e = 0; e < 4; ++e) atomicAdd(out + c + e, s[e]);
        } else {
            for (int e = 0; e < 4 && c + e < N; ++e) {
                float s = 0.f;
                for (int m = r0; m < r1; ++m) s += x[(size_t)m * ld + c + e];
                atomicAdd(out + c + e, s);
            }
        }
    }
}

__global__ __launch_bounds__(256) void embed_sum_f32_kernel(const int64_t* ids, const int64_t* type_ids, const int32_t* pos_ids,
                                                           const float* word, const float* pos, const float* type, int M, int H,
                                                           float* s) {
    const int lane = threadIdx.x & 63;
    const int row = blockIdx.x * 4 + (threadIdx.x >> 6);
    if (row >= M) return;
    const size_t wrow = (size_t)ids[row] * H, prow = (size_t)pos_ids[row] * H;
    const size_t trow = (size_t)(type_ids ? type_ids[row] : 0) * H;
    for (int c = lane; c < H; c += 64) {
        float w = word[wrow + c];
        if (type) w += type[trow + c];                    // HF order: (word + type) + position
        s[(size_t)row * H + c] = w + pos[prow + c];
    }
}

// LayerNorm backward from the pre-norm rows: wave per row, 4 rows per workgroup and LN_ROWS rows per wave; column sums of
// dy * xhat and dy are kept per lane and leave as one atomic per column per workgroup
constexpr int LNF_ROWS = 8, LNF_KMAX = 16;                  // H <= 1024
__global__ __launch_bounds__(256) void ln_bwd_f32_kernel(const float* dy, const float* s, const float* gamma, float eps, int M,
                                                        int H, float* ds, float* dgamma, float* dbeta) {
    extern __shared__ __attribute__((aligned(16))) char smem[];   // [4 waves][2][H]
    float* sh = (float*)smem;
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    float ag[LNF_KMAX], ab[LNF_KMAX];
#pragma unroll
    for (int k = 0; k < LNF_KMAX; ++k) { ag[k] = 0.f; ab[k] = 0.f; }
    const int row0 = (blockIdx.x * 4 + wave) * LNF_ROWS;
    const float inv_n = 1.0f / (float)H;
    for (int rr = 0; rr < LNF_ROWS; ++rr) {
        const int row = row0 + rr;
        if (row >= M) break;
        float x[LNF_KMAX], g[LNF_KMAX];
        float sum = 0.f;
#pragma unroll
        for (int k = 0; k < LNF_KMAX; ++k) {
            const int c = lane + 64 * k;
            x[k] = c < H ? s[(size_t)row * H + c] : 0.f;
            g[k] = c < H ? dy[(size_t)row * H + c] : 0.f;
            sum += x[k];
        }
        const float mean = wave_sum(sum) * inv_n;
        float q = 0.f;
#pragma unroll
        for (int k = 0; k < LNF_KMAX; ++k) {
            const int c = lane + 64 * k;
            x[k] = c < H ? x[k] - mean : 0.f;
            q += x[k] * x[k];
        }
        const float rstd = rsqrtf(wave_sum(q) * inv_n + eps);
        float s1 = 0.f, s2 = 0.f;
#pragma unroll
        for (int k = 0; k < LNF_KMAX; ++k) {
            const int c = lane + 64 * k;
            x[k] *= rstd;                                           // xhat
            ag[k] += g[k] * x[k];
            ab[k] += g[k];
            g[k] *= c < H ? gamma[c] : 0.f;                         // dxhat
            s1 += g[k];
            s2 += g[k] * x[k];
        }
        s1 = wave_sum(s1) * inv_n;
        s2 = wave_sum(s2) * inv_n;
#pragma unroll
        for (int k = 0; k < LNF_KMAX; ++k) {
            const int c = lane + 64 * k;
            if (c < H) ds[(size_t)row * H + c] = rstd * (g[k] - s1 - x[k] * s2);
        }
    }
#pragma unroll
    for (int k = 0; k < LNF_KMAX; ++k) {
        const int c = lane + 64 * k;
        if (c < H) { sh[(wave * 2 + 0) * H + c] = ag[k]; sh[(wave * 2 + 1) * H + c] = ab[k]; }
    }
    __syncthreads();
    for (int c = threadIdx.x; c < 2 * H; c += 256) {
        const int which = c / H, cc = c - which * H;
        const float v = (sh[(0 * 2 + which) * H + cc] + sh[(1 * 2 + which) * H + cc]) + (sh[(2 * 2 + which) * H + cc] + sh[(3 * 2 + which) * H + cc]);
        atomicAdd((which ? dbeta : dgamma) + cc, v);
    }
}

// ---------------------------------------------------------------- attention backward, fp32
// One workgroup of 256 threads per (sequence, head), L <= 512. The keys of the item sit in LDS as fp32 (padded rows); q, dO,
// ctx and v rows come from global memory (a row read by every thread is one broadcast line). Nothing was saved by the
// forward: pass 1 recomputes, row by row, lse_i = logsumexp_j s_ij and delta_i = dO_i . ctx_i; pass 2 walks blocks of 256
// keys -- thread j owns key j of the block, its v row and its dK / dV rows in registers -- and for every query row i
//   s_ij = scale * q_i.k_j + rel[i][j] + mask_j ; p = exp(s - lse_i) ; dp = dO_i.v_j ; ds = p (dp - delta_i)
//   dk_j += scale * ds * q_i ; dv_j += p * dO_i ; drel[i][j] += ds ; dq_i += scale * sum_{j in block} ds_ij k_j
// (dq accumulates over the key blocks in global memory: one workgroup owns the item, so a plain read-modify-write).
constexpr float kMaskMinF = -3.4028234663852886e38f;
struct AttnBwdF32Args {
    const float* qkv; const float* ctx; const float* dctx; const int64_t* mask; const float* rel; float* dqkv; float* drel;
    int nseq, L, A, H; float scale;
    QstDrop drop;          // dropout of the probabilities, as the forward applied it (8-bit generator)
};
__device__ __forceinline__ float block_max256(float v, float* red, int tid) {
    v = wave_max(v);
    if ((tid & 63) == 0) red[tid >> 6] = v;
    __syncthreads();
    const float r = fmaxf(fmaxf(red[0], red[1]), fmaxf(red[2], red[3]));
    __syncthreads();
    return r;
}
__device__ __forceinline__ float block_sum256(float v, float* red, int tid) {
    v = wave_sum(v);
    if ((tid & 63) == 0) red[tid >> 6] = v;
    __syncthreads();
    const float r = (red[0] + red[1]) + (red[2] + red[3]);
    __syncthreads();
    return r;
}
template <int D>
__global__ __launch_bounds__(256) void attn_bwd_f32_kernel(AttnBwdF32Args a) {
    extern __shared__ __attribute__((aligned(16))) char smem[];
    constexpr int LD = D + 1;
    const int L = a.L;
    float* ks = (float*)smem;                  // [L][LD]
    float* lse = ks + (size_t)L * LD;          // [L]: the row maximum m_i ...
    float* lz = lse + L;                       // [L]: ... and log sum_j exp(s_ij - m_i), kept apart: a row with every key masked
                                               //      has m_i = -3.4e38, which would absorb the log of the sum
    float* delta = lz + L;                     // [L]
    float* dsrow = delta + L;                  // [256]
    float* red = dsrow + 256;                  // [4]
    const int tid = threadIdx.x;
    const int head = blockIdx.x % a.A, seq = blockIdx.x / a.A;
    const int ld = 3 * a.H;
    const float* qbase = a.qkv + (size_t)seq * L * ld + head * D;
    const float* kbase = qbase + a.H;
    const float* vbase = qbase + 2 * a.H;
    const float* obase = a.ctx + (size_t)seq * L * a.H + head * D;
    const float* dbase = a.dctx + (size_t)seq * L * a.H + head * D;
    for (int idx = tid; idx < L * D; idx += 256) {
        const int r = idx / D, c = idx - r * D;
        ks[r * LD + c] = kbase[(size_t)r * ld + c];
    }
    for (int i = tid; i < L; i += 256) {
        float acc = 0.f;
        for (int c = 0; c < D; ++c) acc += dbase[(size_t)i * a.H + c] * obase[(size_t)i * a.H + c];
        delta[i] = acc;
    }
    __syncthreads();
    // ---- pass 1: lse_i
    for (int i = 0; i < L; ++i) {
        float q[D];
#pragma unroll
        for (int c = 0; c < D; ++c) q[c] = qbase[(size_t)i * ld + c];
        float s[2] = {-INFINITY, -INFINITY};
#pragma unroll
        for (int u = 0; u < 2; ++u) {
            const int j = tid + 256 * u;
            if (j < L) {
                float acc = 0.f;
#pragma unroll
                for (int c = 0; c < D; ++c) acc += q[c] * ks[j * LD + c];
                acc *= a.scale;
                if (a.rel) acc += a.rel[((size_t)head * L + i) * L + j];
                s[u] = acc + (a.mask[(size_t)seq * L + j] ? 0.f : kMaskMinF);
            }
        }
        const float m = block_max256(fmaxf(s[0], s[1]), red, tid);
        const float e = (tid < L ? expf(s[0] - m) : 0.f) + (tid + 256 < L ? expf(s[1] - m) : 0.f);
        const float z = block_sum256(e, red, tid);
        if (tid == 0) { lse[i] = m; lz[i] = logf(z); }
    }
    __syncthreads();
    // ---- pass 2: key blocks
    for (int kb = 0; kb * 256 < L; ++kb) {
        const int j = kb * 256 + tid;
        const bool live = j < L;
        float v[D], dk[D], dv[D];
#pragma unroll
        for (int c = 0; c < D; ++c) { v[c] = live ? vbase[(size_t)j * ld + c] : 0.f; dk[c] = 0.f; dv[c] = 0.f; }
        const float madd = (live && a.mask[(size_t)seq * L + j]) ? 0.f : kMaskMinF;
        const DropCtx dcx = drop_ctx8(a.drop);
        for (int i = 0; i < L; ++i) {
            float dsj = 0.f;
            if (live) {
                float acc = 0.f, dp = 0.f;
#pragma unroll
                for (int c = 0; c < D; ++c) {
                    const float qc = qbase[(size_t)i * ld + c], dc = dbase[(size_t)i * a.H + c];
                    acc += qc * ks[j * LD + c];
                    dp += dc * v[c];
                }
                float s = acc * a.scale;
                if (a.rel) s += a.rel[((size_t)head * L + i) * L + j];
                s += madd;
                const float p = expf((s - lse[i]) - lz[i]);
                // dropout: O = sum_j (m_ij P_ij) V_j, so dP_ij = m_ij (dO_i . V_j), dV takes the dropped probabilities, and
                // delta_i = dO_i . O_i (from the dropped O the forward saved) still equals sum_k P_ik dP_ik
                float mk = 1.f;
                if (dcx.thr) {
                    const uint32_t idx = ((uint32_t)(seq * a.A + head) * L + (uint32_t)i) * L + (uint32_t)j;
                    mk = (((drop_word4(dcx, idx) >> (8u * (idx & 3u))) & 0xFFu) >= dcx.thr) ? dcx.scale : 0.f;
                }
                dsj = p * (dp * mk - delta[i]);
                if (a.drel) atomicAdd(a.drel + ((size_t)head * L + i) * L + j, dsj);
                const float dss = dsj * a.scale, pm = p * mk;
#pragma unroll
                for (int c = 0; c < D; ++c) {
                    dk[c] += dss * qbase[(size_t)i * ld + c];
                    dv[c] += pm * dbase[(size_t)i * a.H + c];
                }
            }
            dsrow[tid] = dsj;
            __syncthreads();
            if (tid < D) {
                float acc = 0.f;
                const int nj = min(256, L - kb * 256);
                for (int jj = 0; jj < nj; ++jj) acc += dsrow[jj] * ks[(kb * 256 + jj) * LD + tid];
                float* dq = a.dqkv + ((size_t)seq * L + i) * ld + head * D + tid;
                *dq = (kb == 0 ? 0.f : *dq) + acc * a.scale;
            }
            __syncthreads();
        }
        if (live) {
            float* ok = a.dqkv + ((size_t)seq * L + j) * ld + a.H + head * D;
            float* ov = ok + a.H;
#pragma unroll
            for (int c = 0; c < D; ++c) { ok[c] = dk[c]; ov[c] = dv[c]; }
        }
    }
}

}  // namespace

extern "C" int qst_transpose_f32(const float* src, int R, int C, int ld_src, float* dst, int ld_dst, void* stream) {
    if (!src || !dst || R <= 0 || C <= 0 || ld_src < C || ld_dst < R) return QST_ERR_BAD_ARG;
    transpose_f32_kernel<<<dim3((C + 31) / 32, (R + 31) / 32), 256, 0, (hipStream_t)stream>>>(src, R, C, ld_src, dst, ld_dst);
    QST_LAUNCH_CHECK();
    return QST_OK;
}
extern "C" int qst_gelu_f32(const float* u, int64_t n, float* h, void* stream) {
    if (!u || !h || n <= 0) return QST_ERR_BAD_ARG;
    gelu_f32_kernel<<<(unsigned)((n + 255) / 256), 256, 0, (hipStream_t)stream>>>(u, n, h);
    QST_LAUNCH_CHECK();
    return QST_OK;
}
extern "C" int qst_gelu_bwd_f32(const float* dh, const float* u, int64_t n, float* du, void* stream) {
    if (!dh || !u || !du || n <= 0) return QST_ERR_BAD_ARG;
    gelu_bwd_f32_kernel<<<(unsigned)((n + 255) / 256), 256, 0, (hipStream_t)stream>>>(dh, u, n, du);
    QST_LAUNCH_CHECK();
    return QST_OK;
}
extern "C" int qst_colsum_f32(const float* x, int M, int N, int ld, float* out, void* stream) {
    if (!x || !out || M <= 0 || N <= 0 || ld < N) return QST_ERR_BAD_ARG;
    colsum_f32_kernel<<<(M + CS_ROWS - 1) / CS_ROWS, 256, 0, (hipStream_t)stream>>>(x, M, N, ld, out);
    QST_LAUNCH_CHECK();
    return QST_OK;
}
extern "C" int qst_embed_sum_f32(const int64_t* ids, const int64_t* type_ids, const int32_t* pos_ids, const float* word_emb,
                                 const float* pos_emb, const float* type_emb, int M, int H, float* s, void* stream) {
    if (!ids || !pos_ids || !word_emb || !pos_emb || !s || M <= 0 || H <= 0) return QST_ERR_BAD_ARG;
    embed_sum_f32_kernel<<<(M + 3) / 4, 256, 0, (hipStream_t)stream>>>(ids, type_ids, pos_ids, word_emb, pos_emb, type_emb, M, H, s);
    QST_LAUNCH_CHECK();
    return QST_OK;
}
extern "C" int qst_ln_bwd_f32(const float* dy, const float* prenorm, const float* gamma, float eps, int M, int H, float* ds,
                              float* dgamma, float* dbeta, void* stream) {
    if (!dy || !prenorm || !gamma || !ds || !dgamma || !dbeta || M <= 0 || H <= 0) return QST_ERR_BAD_ARG;
    if (H > 64 * LNF_KMAX) return QST_ERR_UNSUPPORTED;
    const int rows_per_block = 4 * LNF_ROWS;
    ln_bwd_f32_kernel<<<(M + rows_per_block - 1) / rows_per_block, 256, (size_t)8 * H * sizeof(float), (hipStream_t)stream>>>(
        dy, prenorm, gamma, eps, M, H, ds, dgamma, dbeta);
    QST_LAUNCH_CHECK();
    return QST_OK;
}
extern "C" int qst_attention_bwd_f32(const float* qkv, const float* ctx, const float* dctx, const int64_t* mask,
                                     const float* rel_bias, int nseq, int L, int A, int d, float* dqkv, float* drel_bias,
                                     void* stream) {
    return qst_attention_bwd_f32_drop(qkv, ctx, dctx, mask, rel_bias, nseq, L, A, d, dqkv, drel_bias, nullptr, stream);
}
extern "C" int qst_attention_bwd_f32_drop(const float* qkv, const float* ctx, const float* dctx, const int64_t* mask,
                                          const float* rel_bias, int nseq, int L, int A, int d, float* dqkv, float* drel_bias,
                                          const QstDrop* drop, void* stream) {
    if (!qkv || !ctx || !dctx || !mask || !dqkv || nseq <= 0 || L <= 0 || A <= 0) return QST_ERR_BAD_ARG;
    if (drel_bias && !rel_bias) return QST_ERR_BAD_ARG;
    if ((d != 32 && d != 64) || L > 512) return QST_ERR_UNSUPPORTED;
    const bool dropping = drop && drop->thr16 && drop->state;
    if (dropping && (drop->thr16 > 65535u || (int64_t)nseq * A * L * L >= ((int64_t)1 << 32))) return QST_ERR_UNSUPPORTED;
    AttnBwdF32Args a{};
    if (dropping) a.drop = *drop;
    a.qkv = qkv; a.ctx = ctx; a.dctx = dctx; a.mask = mask; a.rel = rel_bias; a.dqkv = dqkv; a.drel = drel_bias;
    a.nseq = nseq; a.L = L; a.A = A; a.H = A * d; a.scale = 1.0f / sqrtf((float)d);
    const size_t lds = ((size_t)L * (d + 1) + 3 * (size_t)L + 256 + 8) * sizeof(float);
    hipStream_t st = (hipStream_t)stream;
    if (d == 32) {
        QST_HIP_CHECK(hipFuncSetAttribute((const void*)attn_bwd_f32_kernel<32>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds));
        attn_bwd_f32_kernel<32><<<nseq * A, 256, lds, st>>>(a);
    } else {
        QST_HIP_CHECK(hipFuncSetAttribute((const void*)attn_bwd_f32_kernel<64>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds));
        attn_bwd_f32_kernel<64><<<nseq * A, 256, lds, st>>>(a);
    }
    QST_LAUNCH_CHECK();
    return QST_OK;
}
