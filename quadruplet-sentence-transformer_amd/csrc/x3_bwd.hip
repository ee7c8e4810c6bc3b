// x3_bwd.hip -- the pieces the parity-precision BACKWARD (QST_PREC_BF16X3 training) needs beside gemm_nt_x3.
//
// The reference trains in fp32 (training/main.py:142, use_amp=False); the bf16 path's gradients agree with fp32 autograd
// to 1-4e-2 relative L2 (operand rounding), this path to fp32-class. It is the parity path, not the fast one: every
// contraction goes through the split-bf16 x3 GEMM (dgrad: against a transposed copy of the weight; wgrad: dW = dY^T X as an
// NT product of transposed copies of dY and X, reduction over the token rows), every row kernel works on fp32 and
// recomputes the LayerNorm statistics from the saved pre-norm tensor, and the attention backward is a plain fp32 kernel
// (one workgroup per (sequence, head), L <= 128) that recomputes the probabilities row by row.
//
//   qst_transpose_f32    : dst[C, R] = src[R, C]^T
//   qst_gelu_f32         : h = gelu(u) (exact erf), forward; qst_gelu_bwd_f32: du = dh * gelu'(u)
//   qst_colsum_f32       : out[n] += sum_m x[m, n]                     (bias gradients)
//   qst_embed_sum_f32    : s[m, :] = word[ids[m]] + type[...] + pos[...]  (the embedding LayerNorm's pre-norm input)
//   qst_ln_bwd_f32       : LayerNorm backward from the PRE-NORM tensor (mean / rstd recomputed): ds, dgamma +=, dbeta +=
//   qst_attention_bwd_f32: dqkv (and the [A, L, L] position-bias gradient) from fp32 qkv and dctx
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

// one workgroup per 64 columns; its four waves take every fourth row; fixed summation order
__global__ __launch_bounds__(256) void colsum_f32_kernel(const float* x, int M, int N, int ld, float* out) {
    __shared__ float red[4][64];
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const int c = blockIdx.x * 64 + lane;
    float s = 0.f;
    if (c < N)
        for (int m = wave; m < M; m += 4) s += x[(size_t)m * ld + c];
    red[wave][lane] = s;
    __syncthreads();
    if (wave == 0 && c < N) out[c] += (red[0][lane] + red[1][lane]) + (red[2][lane] + red[3][lane]);
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
// One workgroup of 128 threads per (sequence, head); thread j owns key j (its dK / dV rows live in its registers) and the
// probabilities of the query row at hand are recomputed from q, k (no saved statistics): for row i
//   s_j = scale * q_i.k_j + rel[i][j] + mask_j ; p = softmax(s) ; dp_j = dO_i.v_j ; delta = sum_j p_j dp_j ;
//   ds_j = p_j (dp_j - delta) ; dq_i = scale * sum_j ds_j k_j ; dk_j += scale * ds_j q_i ; dv_j += p_j dO_i ; drel[i][j] += ds_j
// q, k, v, dO of the item sit in LDS as fp32 with padded rows. L <= 128.
constexpr float kMaskMinF = -3.4028234663852886e38f;
struct AttnBwdF32Args {
    const float* qkv; const float* dctx; const int64_t* mask; const float* rel; float* dqkv; float* drel;
    int nseq, L, A, H; float scale;
};
__device__ __forceinline__ float block_max128(float v, float* red, int tid) {
    v = wave_max(v);
    if ((tid & 63) == 0) red[tid >> 6] = v;
    __syncthreads();
    const float r = fmaxf(red[0], red[1]);
    __syncthreads();
    return r;
}
__device__ __forceinline__ float block_sum128(float v, float* red, int tid) {
    v = wave_sum(v);
    if ((tid & 63) == 0) red[tid >> 6] = v;
    __syncthreads();
    const float r = red[0] + red[1];
    __syncthreads();
    return r;
}
template <int D>
__global__ __launch_bounds__(128) void attn_bwd_f32_kernel(AttnBwdF32Args a) {
    extern __shared__ __attribute__((aligned(16))) char smem[];
    constexpr int LD = D + 1;
    float* qs = (float*)smem;                  // [128][LD]
    float* ks = qs + 128 * LD;
    float* vs = ks + 128 * LD;
    float* ds_ = vs + 128 * LD;                // dO rows
    float* dsrow = ds_ + 128 * LD;             // [128]
    float* red = dsrow + 128;                  // [2]
    const int tid = threadIdx.x, j = tid;
    const int head = blockIdx.x % a.A, seq = blockIdx.x / a.A;
    const int L = a.L, ld = 3 * a.H;
    const float* base = a.qkv + (size_t)seq * L * ld + head * D;
    const float* dbase = a.dctx + (size_t)seq * L * a.H + head * D;
    for (int idx = tid; idx < L * D; idx += 128) {
        const int r = idx / D, c = idx - r * D;
        qs[r * LD + c] = base[(size_t)r * ld + c];
        ks[r * LD + c] = base[(size_t)r * ld + a.H + c];
        vs[r * LD + c] = base[(size_t)r * ld + 2 * a.H + c];
        ds_[r * LD + c] = dbase[(size_t)r * a.H + c];
    }
    const bool live = j < L;
    const float madd = (live && a.mask[(size_t)seq * L + j]) ? 0.f : kMaskMinF;
    float dk[D], dv[D];
#pragma unroll
    for (int c = 0; c < D; ++c) { dk[c] = 0.f; dv[c] = 0.f; }
    __syncthreads();
    for (int i = 0; i < L; ++i) {
        float s = -INFINITY, dp = 0.f;
        if (live) {
            float acc = 0.f;
#pragma unroll
            for (int c = 0; c < D; ++c) { acc += qs[i * LD + c] * ks[j * LD + c]; dp += ds_[i * LD + c] * vs[j * LD + c]; }
            s = acc * a.scale;
            if (a.rel) s += a.rel[((size_t)head * L + i) * L + j];
            s += madd;
        }
        const float m = block_max128(s, red, tid);
        const float e = live ? expf(s - m) : 0.f;
        const float z = block_sum128(e, red, tid);
        const float p = e / z;
        const float delta = block_sum128(p * dp, red, tid);
        const float dsj = p * (dp - delta);
        if (live) {
            if (a.drel) atomicAdd(a.drel + ((size_t)head * L + i) * L + j, dsj);
            const float dss = dsj * a.scale;
#pragma unroll
            for (int c = 0; c < D; ++c) { dk[c] += dss * qs[i * LD + c]; dv[c] += p * ds_[i * LD + c]; }
        }
        dsrow[j] = live ? dsj : 0.f;
        __syncthreads();
        if (tid < D) {
            float acc = 0.f;
            for (int jj = 0; jj < L; ++jj) acc += dsrow[jj] * ks[jj * LD + tid];
            a.dqkv[((size_t)seq * L + i) * ld + head * D + tid] = acc * a.scale;
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
    colsum_f32_kernel<<<(N + 63) / 64, 256, 0, (hipStream_t)stream>>>(x, M, N, ld, out);
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
extern "C" int qst_attention_bwd_f32(const float* qkv, const float* dctx, const int64_t* mask, const float* rel_bias, int nseq,
                                     int L, int A, int d, float* dqkv, float* drel_bias, void* stream) {
    if (!qkv || !dctx || !mask || !dqkv || nseq <= 0 || L <= 0 || A <= 0) return QST_ERR_BAD_ARG;
    if (drel_bias && !rel_bias) return QST_ERR_BAD_ARG;
    if ((d != 32 && d != 64) || L > 128) return QST_ERR_UNSUPPORTED;
    AttnBwdF32Args a{};
    a.qkv = qkv; a.dctx = dctx; a.mask = mask; a.rel = rel_bias; a.dqkv = dqkv; a.drel = drel_bias;
    a.nseq = nseq; a.L = L; a.A = A; a.H = A * d; a.scale = 1.0f / sqrtf((float)d);
    const size_t lds = ((size_t)4 * 128 * (d + 1) + 128 + 8) * sizeof(float);
    hipStream_t st = (hipStream_t)stream;
    static QstLdsAttr a32, a64;
    if (d == 32) {
        if (int rc = qst_ensure_lds(a32, (const void*)attn_bwd_f32_kernel<32>, lds)) return rc;
        attn_bwd_f32_kernel<32><<<nseq * A, 128, lds, st>>>(a);
    } else {
        if (int rc = qst_ensure_lds(a64, (const void*)attn_bwd_f32_kernel<64>, lds)) return rc;
        attn_bwd_f32_kernel<64><<<nseq * A, 128, lds, st>>>(a);
    }
    QST_LAUNCH_CHECK();
    return QST_OK;
}
