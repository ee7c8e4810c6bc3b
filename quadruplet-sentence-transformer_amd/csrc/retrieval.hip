// retrieval.hip -- similarity scoring + top-k for the encode()-driven evaluators.
//
// Replaces the device part of sentence-transformers' InformationRetrievalEvaluator.compute_metrices (2.2.2,
// third-party; reference call sites /root/reference/models/evaluators.py:572-588 and
// /root/reference/ir_evauation_script.py:107-131): for every corpus chunk, score_function(query_emb, chunk_emb)
// (util.cos_sim = normalise both sides, then a matmul; util.dot_score = matmul) followed by torch.topk(k, dim=1,
// largest=True, sorted=False) and a host-side merge of the per-chunk results.
//
// Here: rows are normalised (cosine) or copied into a 4-row-padded workspace, the [nq, nc] score matrix comes from the
// split-bf16 x3 GEMM (fp32-class products: near-ties keep the order an fp32 matmul would give them), and one
// workgroup per query selects the k best by a 4-pass radix select on order-preserving integer keys, then sorts them.
#include "qst_common.h"
#include "qst_kernels.h"

namespace {

// L2-normalise rows (torch.nn.functional.normalize(p=2, dim=1, eps=1e-12)) or copy them; rows >= n are zero-filled.
__global__ __launch_bounds__(256) void prep_rows_kernel(const float* x, int n, int n_pad, int dim, int normalize, float* y) {
    const int lane = threadIdx.x & 63;
    const int row = blockIdx.x * 4 + (threadIdx.x >> 6);
    if (row >= n_pad) return;
    float* dst = y + (size_t)row * dim;
    if (row >= n) {
        for (int c = lane; c < dim; c += 64) dst[c] = 0.f;
        return;
    }
    const float* src = x + (size_t)row * dim;
    float inv = 1.f;
    if (normalize) {
        float s = 0.f;
        for (int c = lane; c < dim; c += 64) { const float v = src[c]; s += v * v; }
        inv = 1.f / fmaxf(sqrtf(wave_sum(s)), 1e-12f);
    }
    for (int c = lane; c < dim; c += 64) dst[c] = src[c] * inv;
}

// order-preserving map float -> uint32 (larger float <=> larger key); NaNs sort above +inf, as torch.topk ranks them
__device__ __forceinline__ uint32_t fkey(float v) {
    const uint32_t u = __builtin_bit_cast(uint32_t, v);
    return (u & 0x80000000u) ? ~u : (u | 0x80000000u);
}

constexpr int TOPK_MAX = 1024;

// One workgroup per row. Pass p fixes 8 more bits of the k-th largest key: histogram the digit of every element that
// still matches the prefix, walk the 256 bins from the top. After 4 passes `prefix` IS the k-th largest key, `need`
// the number of elements equal to it that belong to the result. Then collect (> prefix: all, == prefix: the first
// `need` in index order of arrival), sort by (score desc, index asc) with a bitonic network in LDS, write.
// cap: entries above it do not take part (they rank below everything and come out as -inf / index -1): the
// "not too similar" filter of negative mining. +inf = plain top-k.
__global__ __launch_bounds__(256) void topk_rows_kernel(const float* scores, int64_t ld, const int64_t* index_map, int n,
                                                        int k, float cap, float* out_scores, int64_t* out_index) {
    __shared__ uint32_t hist[256];
    __shared__ uint32_t sh_prefix, sh_need, sh_cnt, sh_cnt_eq;
    __shared__ float sv[TOPK_MAX];
    __shared__ int64_t si[TOPK_MAX];
    const int tid = threadIdx.x;
    const float* row = scores + (size_t)blockIdx.x * ld;
    const int64_t* imap = index_map ? index_map + (size_t)blockIdx.x * ld : nullptr;

    uint32_t prefix = 0, mask = 0, need = (uint32_t)k;
    for (int pass = 0; pass < 4; ++pass) {
        const int shift = 24 - 8 * pass;
        hist[tid] = 0;
        __syncthreads();
        for (int i = tid; i < n; i += 256) {
            const float v = row[i];
            const uint32_t key = v > cap ? 0u : fkey(v);
            if ((key & mask) == prefix) atomicAdd(&hist[(key >> shift) & 255u], 1u);
        }
        __syncthreads();
        if (tid == 0) {
            uint32_t acc = 0;
            int d = 255;
            for (; d > 0; --d) {
                if (acc + hist[d] >= need) break;
                acc += hist[d];
            }
            sh_prefix = prefix | ((uint32_t)d << shift);
            sh_need = need - acc;                      // still to take among the elements with this digit
        }
        __syncthreads();
        prefix = sh_prefix;
        need = sh_need;
        mask |= 255u << shift;
        __syncthreads();
    }
    // collect
    if (tid == 0) { sh_cnt = 0; sh_cnt_eq = 0; }
    __syncthreads();
    const int kp = k <= 1 ? 1 : 1 << (32 - __builtin_clz((unsigned)(k - 1)));     // next power of two
    for (int i = tid; i < kp; i += 256) { sv[i] = -INFINITY; si[i] = INT64_MAX; }
    __syncthreads();
    for (int i = tid; i < n; i += 256) {
        const float v0 = row[i];
        const bool over = v0 > cap;
        const float v = over ? -INFINITY : v0;
        const uint32_t key = over ? 0u : fkey(v0);
        bool take = key > prefix;
        if (key == prefix) take = atomicAdd(&sh_cnt_eq, 1u) < need;
        if (take) {
            const uint32_t pos = atomicAdd(&sh_cnt, 1u);
            if (pos < (uint32_t)k) { sv[pos] = v; si[pos] = over ? INT64_MAX : (imap ? imap[i] : (int64_t)i); }
        }
    }
    __syncthreads();
    // bitonic sort of kp entries: descending score, ascending index among equal scores
    for (int size = 2; size <= kp; size <<= 1) {
        for (int stride = size >> 1; stride > 0; stride >>= 1) {
            for (int t = tid; t < (kp >> 1); t += 256) {
                const int lo = 2 * t - (t & (stride - 1));
                const int hi = lo + stride;
                const bool desc = ((lo & size) == 0);
                const float a = sv[lo], b = sv[hi];
                const int64_t ia = si[lo], ib = si[hi];
                const bool a_first = (a > b) || (a == b && ia < ib);       // a ranks before b
                if (desc ? !a_first : a_first) { sv[lo] = b; sv[hi] = a; si[lo] = ib; si[hi] = ia; }
            }
            __syncthreads();
        }
    }
    for (int i = tid; i < k; i += 256) {
        out_scores[(size_t)blockIdx.x * k + i] = sv[i];
        out_index[(size_t)blockIdx.x * k + i] = si[i] == INT64_MAX ? -1 : si[i];
    }
}

// Euclidean score 1 / (1 + ||q - c||_2): the reference's own third score function (models/evaluators.py:392-405,
// `1 / (1 + torch.cdist(a, b, p=2))`, passed as score_functions['euclid_score'] by training/main.py:57 and
// ir_evauation_script.py:71). Distances are summed from the differences themselves (an fp32 FMA chain per pair): the
// |q|^2 + |c|^2 - 2 q.c form loses the distance of near-duplicates to cancellation, and those are the pairs that decide
// a ranking. 64 x 64 pairs per workgroup, 4 x 4 per thread, 32-dimension slabs staged transposed ([dim][row]) so that a
// thread's four rows are one 16-byte LDS read. VALU-bound: 2 instructions per pair and dimension.
__global__ __launch_bounds__(256) void euclid_scores_kernel(const float* q, const float* c, int nq, int nc, int dim,
                                                            float* out, int ldo) {
    __shared__ __attribute__((aligned(16))) float qs[32][68];
    __shared__ __attribute__((aligned(16))) float cs[32][68];
    const int tid = threadIdx.x, tx = tid & 15, ty = tid >> 4;
    const int q0 = blockIdx.y * 64, c0 = blockIdx.x * 64;
    const int lrow = tid >> 2, lk = (tid & 3) * 8;          // staging: row lrow, dims lk .. lk+7 of the slab
    float acc[4][4];
#pragma unroll
    for (int i = 0; i < 4; ++i)
#pragma unroll
        for (int j = 0; j < 4; ++j) acc[i][j] = 0.f;
    for (int k0 = 0; k0 < dim; k0 += 32) {
        f32x4 a0 = {0.f, 0.f, 0.f, 0.f}, a1 = a0, b0 = a0, b1 = a0;
        if (q0 + lrow < nq) {
            a0 = *(const f32x4*)(q + (size_t)(q0 + lrow) * dim + k0 + lk);
            a1 = *(const f32x4*)(q + (size_t)(q0 + lrow) * dim + k0 + lk + 4);
        }
        if (c0 + lrow < nc) {
            b0 = *(const f32x4*)(c + (size_t)(c0 + lrow) * dim + k0 + lk);
            b1 = *(const f32x4*)(c + (size_t)(c0 + lrow) * dim + k0 + lk + 4);
        }
        __syncthreads();                                    // the previous slab's readers are done
#pragma unroll
        for (int e = 0; e < 4; ++e) {
            qs[lk + e][lrow] = a0[e]; qs[lk + 4 + e][lrow] = a1[e];
            cs[lk + e][lrow] = b0[e]; cs[lk + 4 + e][lrow] = b1[e];
        }
        __syncthreads();
#pragma unroll 8
        for (int d = 0; d < 32; ++d) {
            const f32x4 a = *(const f32x4*)(&qs[d][ty * 4]);
            const f32x4 b = *(const f32x4*)(&cs[d][tx * 4]);
#pragma unroll
            for (int i = 0; i < 4; ++i)
#pragma unroll
                for (int j = 0; j < 4; ++j) {
                    const float df = a[i] - b[j];
                    acc[i][j] = fmaf(df, df, acc[i][j]);
                }
        }
    }
#pragma unroll
    for (int i = 0; i < 4; ++i) {
        const int qi = q0 + ty * 4 + i;
        if (qi >= nq) continue;
#pragma unroll
        for (int j = 0; j < 4; ++j) {
            const int cj = c0 + tx * 4 + j;
            if (cj < ldo) out[(size_t)qi * ldo + cj] = cj < nc ? 1.f / (1.f + sqrtf(acc[i][j])) : 0.f;
        }
    }
}

constexpr int kQueryBlock = 2048;        // score matrix rows per GEMM launch (keeps byte offsets inside 32 bits)

inline size_t pad4(size_t n) { return (n + 3) / 4 * 4; }

}  // namespace

extern "C" int qst_topk_rows(const float* scores, int64_t ld, const int64_t* index_map, int nrows, int n, int k,
                             float* out_scores, int64_t* out_index, void* stream) {
    if (!scores || !out_scores || !out_index || nrows <= 0 || n <= 0 || k <= 0 || ld < n) return QST_ERR_BAD_ARG;
    if (k > n || k > TOPK_MAX) return QST_ERR_UNSUPPORTED;
    topk_rows_kernel<<<nrows, 256, 0, (hipStream_t)stream>>>(scores, ld, index_map, n, k, INFINITY, out_scores, out_index);
    QST_LAUNCH_CHECK();
    return QST_OK;
}

extern "C" size_t qst_topk_workspace_bytes(int nq, int nc, int dim) {
    if (nq <= 0 || nc <= 0 || dim <= 0) return 0;
    const size_t qrows = pad4((size_t)(nq < kQueryBlock ? nq : kQueryBlock));
    return (pad4((size_t)nq) + pad4((size_t)nc)) * dim * sizeof(float) + qrows * pad4((size_t)nc) * sizeof(float) + 1024;
}

static int topk_scores_impl(const float* queries, const float* corpus, int nq, int nc, int dim, int k, int mode, float cap,
                            float* out_scores, int64_t* out_index, void* workspace, size_t workspace_bytes, void* stream);

extern "C" int qst_topk_scores(const float* queries, const float* corpus, int nq, int nc, int dim, int k, int mode,
                               float* out_scores, int64_t* out_index, void* workspace, size_t workspace_bytes,
                               void* stream) {
    return topk_scores_impl(queries, corpus, nq, nc, dim, k, mode, INFINITY, out_scores, out_index, workspace,
                            workspace_bytes, stream);
}

extern "C" int qst_topk_scores_capped(const float* queries, const float* corpus, int nq, int nc, int dim, int k, int mode,
                                      float max_score, float* out_scores, int64_t* out_index, void* workspace,
                                      size_t workspace_bytes, void* stream) {
    if (max_score != max_score) return QST_ERR_BAD_ARG;
    return topk_scores_impl(queries, corpus, nq, nc, dim, k, mode, max_score, out_scores, out_index, workspace,
                            workspace_bytes, stream);
}

// scores of query rows [q0, q0 + rows) against the whole corpus into sc [rows, ncp]; qn / cn = prepared operands
static int score_block(const float* qn, const float* cn, int rows, int nc, int ncp, int dim, int mode, float* sc,
                       hipStream_t st) {
    if (mode == QST_SCORE_EUCLID) {
        euclid_scores_kernel<<<dim3((ncp + 63) / 64, (rows + 63) / 64), 256, 0, st>>>(qn, cn, rows, nc, dim, sc, ncp);
        QST_LAUNCH_CHECK();
        return QST_OK;
    }
    QstGemmArgs g{};
    g.A = qn; g.B = cn; g.C = sc;
    g.M = rows; g.N = ncp; g.K = dim; g.lda = dim; g.ldb = dim; g.ldc = ncp;
    return qst_gemm_nt_x3(&g, 0, st);
}

static int prep_operands(const float* queries, const float* corpus, int nq, int nc, int dim, int mode, float* qn, float* cn,
                         hipStream_t st) {
    const int nqp = (int)pad4(nq), ncp = (int)pad4(nc);
    prep_rows_kernel<<<(nqp + 3) / 4, 256, 0, st>>>(queries, nq, nqp, dim, mode == QST_SCORE_COS, qn);
    QST_LAUNCH_CHECK();
    prep_rows_kernel<<<(ncp + 3) / 4, 256, 0, st>>>(corpus, nc, ncp, dim, mode == QST_SCORE_COS, cn);
    QST_LAUNCH_CHECK();
    return QST_OK;
}

static int topk_scores_impl(const float* queries, const float* corpus, int nq, int nc, int dim, int k, int mode, float cap,
                            float* out_scores, int64_t* out_index, void* workspace, size_t workspace_bytes, void* stream) {
    if (!queries || !corpus || !out_scores || !out_index || !workspace || nq <= 0 || nc <= 0 || dim <= 0 || k <= 0)
        return QST_ERR_BAD_ARG;
    if (mode != QST_SCORE_DOT && mode != QST_SCORE_COS && mode != QST_SCORE_EUCLID) return QST_ERR_BAD_ARG;
    if (k > nc || k > TOPK_MAX || dim % 32 != 0) return QST_ERR_UNSUPPORTED;
    if (workspace_bytes < qst_topk_workspace_bytes(nq, nc, dim)) return QST_ERR_WORKSPACE;
    if ((int64_t)pad4(nc) * dim * 4 >= 0x7FFFFF00LL) return QST_ERR_UNSUPPORTED;
    hipStream_t st = (hipStream_t)stream;
    const int nqp = (int)pad4(nq), ncp = (int)pad4(nc);
    float* qn = (float*)workspace;
    float* cn = qn + (size_t)nqp * dim;
    float* sc = cn + (size_t)ncp * dim;
    int rc = prep_operands(queries, corpus, nq, nc, dim, mode, qn, cn, st);
    if (rc != QST_OK) return rc;
    for (int q0 = 0; q0 < nq; q0 += kQueryBlock) {
        const int rows = nq - q0 < kQueryBlock ? nq - q0 : kQueryBlock;
        rc = score_block(qn + (size_t)q0 * dim, cn, rows, nc, ncp, dim, mode, sc, st);
        if (rc != QST_OK) return rc;
        topk_rows_kernel<<<rows, 256, 0, st>>>(sc, ncp, nullptr, nc, k, cap, out_scores + (size_t)q0 * k,
                                               out_index + (size_t)q0 * k);
        QST_LAUNCH_CHECK();
    }
    return QST_OK;
}

// The whole score matrix (sentence_transformers.util.cos_sim / dot_score and the reference's euclidean_score as
// functions): out f32 [nq, ld_out], ld_out = nc rounded up to a multiple of 4 (the pad columns are written too).
extern "C" size_t qst_score_workspace_bytes(int nq, int nc, int dim) {
    if (nq <= 0 || nc <= 0 || dim <= 0) return 0;
    return (pad4((size_t)nq) + pad4((size_t)nc)) * dim * sizeof(float) + 1024;
}
extern "C" int qst_score_matrix(const float* queries, const float* corpus, int nq, int nc, int dim, int mode, float* out,
                                int64_t ld_out, void* workspace, size_t workspace_bytes, void* stream) {
    if (!queries || !corpus || !out || !workspace || nq <= 0 || nc <= 0 || dim <= 0) return QST_ERR_BAD_ARG;
    if (mode != QST_SCORE_DOT && mode != QST_SCORE_COS && mode != QST_SCORE_EUCLID) return QST_ERR_BAD_ARG;
    if (dim % 32 != 0 || ld_out != (int64_t)pad4(nc)) return QST_ERR_UNSUPPORTED;
    if (workspace_bytes < qst_score_workspace_bytes(nq, nc, dim)) return QST_ERR_WORKSPACE;
    if ((int64_t)pad4(nc) * dim * 4 >= 0x7FFFFF00LL) return QST_ERR_UNSUPPORTED;
    hipStream_t st = (hipStream_t)stream;
    const int nqp = (int)pad4(nq), ncp = (int)pad4(nc);
    float* qn = (float*)workspace;
    float* cn = qn + (size_t)nqp * dim;
    int rc = prep_operands(queries, corpus, nq, nc, dim, mode, qn, cn, st);
    if (rc != QST_OK) return rc;
    for (int q0 = 0; q0 < nq; q0 += kQueryBlock) {
        const int rows = nq - q0 < kQueryBlock ? nq - q0 : kQueryBlock;
        rc = score_block(qn + (size_t)q0 * dim, cn, rows, nc, ncp, dim, mode, out + (size_t)q0 * ncp, st);
        if (rc != QST_OK) return rc;
    }
    return QST_OK;
}

extern "C" int qst_normalize_rows(const float* x, int n, int dim, float* out, void* stream) {
    if (!x || !out || n <= 0 || dim <= 0) return QST_ERR_BAD_ARG;
    prep_rows_kernel<<<(n + 3) / 4, 256, 0, (hipStream_t)stream>>>(x, n, n, dim, 1, out);
    QST_LAUNCH_CHECK();
    return QST_OK;
}
