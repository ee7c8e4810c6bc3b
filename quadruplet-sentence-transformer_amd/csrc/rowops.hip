// rowops.hip -- HBM-bound row kernels: embedding gather + LayerNorm, LayerNorm fwd/bwd,
// embedding backward, mean-pool + L2-normalise head fwd/bwd, MPNet position ids / relative bias,
// bf16 shadow refresh.
//
// Replaces BertEmbeddings.forward (modeling_bert.py:53-108), nn.LayerNorm in BertSelfOutput/BertOutput
// (:282-293, :340-351), ST Pooling(mean)+Normalize and their autograd.
// One wave64 per token row; a lane owns float2 columns lane, lane+64, ... so row statistics are one
// wave reduction and all global accesses are coalesced 8-byte-per-lane.
#include "qst_common.h"
#include "qst_kernels.h"

namespace {

typedef __attribute__((ext_vector_type(2))) float f32x2;

// MX scale exponent and power of two as csrc/gemm.hip defines them (mx_exponent / pow2f)
__device__ __forceinline__ int mx_exponent_r(float amax) {
    const uint32_t u = __builtin_bit_cast(uint32_t, amax);
    if ((u & 0x7F800000u) == 0u) return -127;
    int e = (int)((u >> 23) & 0xFF) - 127 - 8 + ((u & 0x7FFFFFu) > 0x600000u ? 1 : 0);
    return e < -127 ? -127 : (e > 126 ? 126 : e);
}

template <int VPL>
__device__ __forceinline__ void ln_row_finish(const f32x2 (&v)[VPL], int lane, int H, int row,
                                              const float* gamma, const float* beta, float eps,
                                              float* y, op16* yb, op16* xh, float* rstd_out,
                                              uint8_t* yq = nullptr, uint8_t* ys = nullptr, int M = 0,
                                              DropCtx dc = DropCtx{0u, 0u, 1.f}) {
    const int nv = H >> 1;
    float s = 0.f;
#pragma unroll
    for (int i = 0; i < VPL; ++i)
        if (lane + 64 * i < nv) s += v[i][0] + v[i][1];
    const float mean = wave_sum(s) / (float)H;
    float q = 0.f;
#pragma unroll
    for (int i = 0; i < VPL; ++i)
        if (lane + 64 * i < nv) {
            const float a = v[i][0] - mean, b = v[i][1] - mean;
            q += a * a + b * b;
        }
    const float var = wave_sum(q) / (float)H;       // biased variance, as nn.LayerNorm
    const float rstd = rsqrtf(var + eps);
    if (lane == 0 && rstd_out) rstd_out[row] = rstd;
    const size_t base = (size_t)row * H;
#pragma unroll
    for (int i = 0; i < VPL; ++i) {
        const int c = lane + 64 * i;
        if (c < nv) {
            const float h0 = (v[i][0] - mean) * rstd, h1 = (v[i][1] - mean) * rstd;
            const f32x2 g = *(const f32x2*)(gamma + 2 * c), b = *(const f32x2*)(beta + 2 * c);
            f32x2 o;
            o[0] = h0 * g[0] + b[0];
            o[1] = h1 * g[1] + b[1];
            if (dc.thr) {                                   // dropout of the LayerNorm output (embeddings); xhat stays undropped
                float m0, m1;
                drop_pair(dc, (uint32_t)(base + 2 * c), m0, m1);
                o[0] *= m0; o[1] *= m1;
            }
            *(f32x2*)(y + base + 2 * c) = o;
            if (yb) *(uint32_t*)(yb + base + 2 * c) = pack_op2(o[0], o[1]);
            if (xh) *(uint32_t*)(xh + base + 2 * c) = pack_op2(h0, h1);
        }
    }
    if (yq) {
        // the same output as MXFP8 for an fp8 GEMM (QST_PREC_FP8): quantised from the bf16-rounded values, as a separate
        // qst_quant_mx pass over yb would; 16 consecutive lanes (2 columns each) form one 32-column block. H % 64 == 0 here,
        // so a block never straddles the valid / invalid column boundary and all 64 lanes take part in the DPP reduction.
#pragma unroll
        for (int i = 0; i < VPL; ++i) {
            const int c = lane + 64 * i;
            float o0 = 0.f, o1 = 0.f;
            if (c < nv) {
                const float h0 = (v[i][0] - mean) * rstd, h1 = (v[i][1] - mean) * rstd;
                const f32x2 g = *(const f32x2*)(gamma + 2 * c), b = *(const f32x2*)(beta + 2 * c);
                float f0 = h0 * g[0] + b[0], f1 = h1 * g[1] + b[1];
                if (dc.thr) {                                   // (the dropped output, as the bf16 copy above holds it)
                    float m0, m1;
                    drop_pair(dc, (uint32_t)(base + 2 * c), m0, m1);
                    f0 *= m0; f1 *= m1;
                }
                const uint32_t pk = pack_op2(f0, f1);
                o0 = op_lo(pk); o1 = op_hi(pk);
            }
            const float amax = row16_max(fmaxf(fabsf(o0), fabsf(o1)));
            const int ex = mx_exponent_r(amax);
            const float inv = __builtin_bit_cast(float, (uint32_t)(127 - ex) << 23);
            const uint32_t p = (uint32_t)__builtin_amdgcn_cvt_pk_fp8_f32(o0 * inv, o1 * inv, 0, false);
            if (c < nv) {
                *(uint16_t*)(yq + base + 2 * c) = (uint16_t)(p & 0xFFFFu);
                if ((lane & 15) == 0) {
                    const int kb = (2 * c) >> 5;
                    ys[((size_t)(kb >> 2) * M + row) * 4 + (kb & 3)] = (uint8_t)(ex + 127);
                }
            }
        }
    }
}

// EMB_ROWS rows per wave, all their gathers in flight before the first row is normalised. One row per wave (8192
// workgroups at M = 32768) runs at 3 TB/s, 50 us; four rows per wave measured SLOWER (54.5 us): the chip then holds the
// whole grid at once and nothing overlaps the waves' reduction / store tails.
constexpr int EMB_ROWS = 1;
template <int VPL>
__global__ __launch_bounds__(256) void embed_ln_fwd_kernel(const int64_t* ids, const int64_t* type_ids,
                                                           const int32_t* pos_ids, const float* word,
                                                           const float* pos, const float* type,
                                                           const float* gamma, const float* beta, float eps,
                                                           int M, int H, float* y, op16* yb, op16* xh, float* rstd,
                                                           uint8_t* yq, uint8_t* ys, QstDrop drop) {
    op_saturate(true);
    const int lane = threadIdx.x & 63;
    const int row0 = (blockIdx.x * 4 + (threadIdx.x >> 6)) * EMB_ROWS;
    if (row0 >= M) return;
    const int nv = H >> 1;
    f32x2 v[EMB_ROWS][VPL];
#pragma unroll
    for (int r = 0; r < EMB_ROWS; ++r) {
        const int row = min(row0 + r, M - 1);             // (a clamped duplicate is loaded and dropped)
        const size_t wrow = (size_t)ids[row] * H, prow = (size_t)pos_ids[row] * H;
        const size_t trow = (size_t)(type_ids ? type_ids[row] : 0) * H;
#pragma unroll
        for (int i = 0; i < VPL; ++i) {
            const int c = lane + 64 * i;
            if (c < nv) {
                f32x2 w = *(const f32x2*)(word + wrow + 2 * c);
                if (type) {                                   // HF order: (word + type) + position
                    const f32x2 t = *(const f32x2*)(type + trow + 2 * c);
                    w[0] += t[0]; w[1] += t[1];
                }
                const f32x2 p = *(const f32x2*)(pos + prow + 2 * c);
                v[r][i][0] = w[0] + p[0];
                v[r][i][1] = w[1] + p[1];
            } else { v[r][i][0] = 0.f; v[r][i][1] = 0.f; }
        }
    }
    const DropCtx dc = drop_ctx(drop);
#pragma unroll
    for (int r = 0; r < EMB_ROWS; ++r)
        if (row0 + r < M) ln_row_finish<VPL>(v[r], lane, H, row0 + r, gamma, beta, eps, y, yb, xh, rstd, yq, ys, M, dc);
}

template <int VPL>
__global__ __launch_bounds__(256) void ln_fwd_kernel(const float* s, const float* gamma, const float* beta, float eps,
                                                     int M, int H, float* y, op16* yb, op16* xh, float* rstd,
                                                     uint8_t* yq, uint8_t* ys) {
    op_saturate(true);
    const int lane = threadIdx.x & 63;
    const int row = blockIdx.x * 4 + (threadIdx.x >> 6);
    if (row >= M) return;
    const int nv = H >> 1;
    f32x2 v[VPL];
#pragma unroll
    for (int i = 0; i < VPL; ++i) {
        const int c = lane + 64 * i;
        if (c < nv) v[i] = *(const f32x2*)(s + (size_t)row * H + 2 * c);
        else { v[i][0] = 0.f; v[i][1] = 0.f; }
    }
    ln_row_finish<VPL>(v, lane, H, row, gamma, beta, eps, y, yb, xh, rstd, yq, ys, M);
}

// LayerNorm backward. Each wave walks ROWS_PER_WAVE rows, keeps dgamma/dbeta partials for its columns in
// registers, and the block adds them to global once (one atomic per column per block).
constexpr int LN_BWD_ROWS_PER_WAVE = 8;

template <int VPL>
__global__ __launch_bounds__(256) void ln_bwd_kernel(const float* dy, const op16* xh, const float* rstd,
                                                     const float* gamma, int M, int H, float* ds, op16* dsb,
                                                     float* dgamma, float* dbeta, float* partials,
                                                     QstDrop drop_in, QstDrop drop_out) {
    extern __shared__ __attribute__((aligned(16))) char smem[];   // [4 waves][2][H] floats
    float* sh = (float*)smem;
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const int nv = H >> 1;
    f32x2 g[VPL], ag[VPL], ab[VPL];
#pragma unroll
    for (int i = 0; i < VPL; ++i) {
        const int c = lane + 64 * i;
        g[i][0] = g[i][1] = 0.f;
        if (c < nv) g[i] = *(const f32x2*)(gamma + 2 * c);
        ag[i][0] = ag[i][1] = ab[i][0] = ab[i][1] = 0.f;
    }
    const int row0 = (blockIdx.x * 4 + wave) * LN_BWD_ROWS_PER_WAVE;
    const DropCtx dci = drop_ctx(drop_in), dco = drop_ctx(drop_out);
    // software pipeline: the loads of row r+1 are in flight while row r is reduced and stored
    f32x2 dn[VPL];
    uint32_t xn[VPL];
    float rsn = 0.f;
    auto fetch = [&](int row) {
        const size_t base = (size_t)row * H;
#pragma unroll
        for (int i = 0; i < VPL; ++i) {
            const int c = lane + 64 * i;
            dn[i][0] = dn[i][1] = 0.f; xn[i] = 0u;
            if (c < nv) {
                dn[i] = *(const f32x2*)(dy + base + 2 * c);
                xn[i] = *(const uint32_t*)(xh + base + 2 * c);
            }
        }
        rsn = rstd[row];
    };
    if (row0 < M) fetch(row0);
    for (int rr = 0; rr < LN_BWD_ROWS_PER_WAVE; ++rr) {
        const int row = row0 + rr;
        if (row >= M) break;
        const size_t base = (size_t)row * H;
        f32x2 d[VPL], x[VPL];
        const float rs = rsn;
#pragma unroll
        for (int i = 0; i < VPL; ++i) { d[i] = dn[i]; x[i][0] = op_lo(xn[i]); x[i][1] = op_hi(xn[i]); }
        if (dci.thr) {                   // the dropout that followed this LayerNorm: its mask on the incoming gradient
#pragma unroll
            for (int i = 0; i < VPL; ++i) {
                float m0, m1;
                drop_pair(dci, (uint32_t)(base + 2 * (lane + 64 * i)), m0, m1);
                d[i][0] *= m0; d[i][1] *= m1;
            }
        }
        if (rr + 1 < LN_BWD_ROWS_PER_WAVE && row + 1 < M) fetch(row + 1);
        float s1 = 0.f, s2 = 0.f;
#pragma unroll
        for (int i = 0; i < VPL; ++i) {
            const int c = lane + 64 * i;
            if (c < nv) {
                ag[i][0] += d[i][0] * x[i][0]; ag[i][1] += d[i][1] * x[i][1];
                ab[i][0] += d[i][0];           ab[i][1] += d[i][1];
                d[i][0] *= g[i][0]; d[i][1] *= g[i][1];          // dxhat
                s1 += d[i][0] + d[i][1];
                s2 += d[i][0] * x[i][0] + d[i][1] * x[i][1];
            }
        }
        const float m1 = wave_sum(s1) / (float)H, m2 = wave_sum(s2) / (float)H;
#pragma unroll
        for (int i = 0; i < VPL; ++i) {
            const int c = lane + 64 * i;
            if (c < nv) {
                f32x2 o;
                o[0] = rs * (d[i][0] - m1 - x[i][0] * m2);
                o[1] = rs * (d[i][1] - m1 - x[i][1] * m2);
                *(f32x2*)(ds + base + 2 * c) = o;
                if (dsb) {
                    if (dco.thr) {       // ds_bf16 = gradient of the dropped projection output; ds = of the residual
                        float m0, m1;
                        drop_pair(dco, (uint32_t)(base + 2 * c), m0, m1);
                        o[0] *= m0; o[1] *= m1;
                    }
                    *(uint32_t*)(dsb + base + 2 * c) = pack_op2(o[0], o[1]);
                }
            }
        }
    }
#pragma unroll
    for (int i = 0; i < VPL; ++i) {
        const int c = lane + 64 * i;
        if (c < nv) {
            *(f32x2*)(sh + (wave * 2 + 0) * H + 2 * c) = ag[i];
            *(f32x2*)(sh + (wave * 2 + 1) * H + 2 * c) = ab[i];
        }
    }
    __syncthreads();
    for (int c = threadIdx.x; c < H; c += 256) {
        float a = 0.f, b = 0.f;
#pragma unroll
        for (int w = 0; w < 4; ++w) { a += sh[(w * 2 + 0) * H + c]; b += sh[(w * 2 + 1) * H + c]; }
        if (partials) {           // [gridDim.x][2][H]: reduced by ln_bwd_reduce_kernel in a fixed order
            partials[((size_t)blockIdx.x * 2 + 0) * H + c] = a;
            partials[((size_t)blockIdx.x * 2 + 1) * H + c] = b;
        } else {
            atomicAdd(dgamma + c, a);
            atomicAdd(dbeta + c, b);
        }
    }
}

// dgamma[c] += sum_b partials[b][0][c], dbeta[c] += sum_b partials[b][1][c]: thread per column (coalesced rows),
// 32 row-slices; each slice adds its partial sum with one atomic (32 adds per address: no contention to speak of)
__global__ __launch_bounds__(256) void ln_bwd_reduce_kernel(const float* partials, int nblocks, int H, float* dgamma,
                                                            float* dbeta) {
    const int col = blockIdx.x * 256 + threadIdx.x;
    if (col >= 2 * H) return;
    float acc = 0.f;
    for (int b = blockIdx.y; b < nblocks; b += gridDim.y) acc += partials[(size_t)b * 2 * H + col];
    atomicAdd(col < H ? dgamma + col : dbeta + (col - H), acc);
}

// Embedding backward in one pass over ds [M, H] (it used to be read twice: word + type rows, then position rows).
//  word: row scatter with float atomics. They execute at the memory side at ~1.5 TB/s chip-wide and are the cost of this
//        kernel (80 us at M = 32768 for 50 MB of adds). Tried and removed: counting the occurrences of every id first and
//        updating rows whose id occurs once with a plain load-add-store -- only 34% of the rows of a batch of 32768 random
//        ids are unique, and the count pass cost more than that saved (85.5 vs 80.0 us).
//  type: every token hits one of <= 2 rows -> per-lane register partials, one atomic per column per block
//  position: a wave walks ONE position t over a slice of 16 sequences; rows whose position id equals the first one's are
//        summed in registers (BERT: all of them), atomics only for the irregular rest (MPNet's pad-dependent ids)
constexpr int EMB_SEQS_PER_WAVE = 16;
__global__ __launch_bounds__(256) void embed_bwd_kernel(const float* ds, const int64_t* ids, const int64_t* type_ids,
                                                        const int32_t* pos_ids, int nseq, int L, int H,
                                                        int num_types, float* dword, float* dpos, float* dtype_) {
    extern __shared__ __attribute__((aligned(16))) char smem[];   // [4 waves][2 types][H] floats
    float* sh = (float*)smem;
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    constexpr int KMAX = 16;                                      // H <= 1024
    float acc0[KMAX], acc1[KMAX], accp[KMAX];
#pragma unroll
    for (int k = 0; k < KMAX; ++k) { acc0[k] = 0.f; acc1[k] = 0.f; accp[k] = 0.f; }
    // block = 4 consecutive positions of one slice of sequences
    const int tblocks = (L + 3) / 4;
    const int t = (blockIdx.x % tblocks) * 4 + wave, slice = blockIdx.x / tblocks;
    const int s0 = slice * EMB_SEQS_PER_WAVE;
    const bool live = t < L;
    // lane rr holds what row rr of this wave needs (one round trip for all 16 rows instead of three dependent ones per row)
    int my_id = 0, my_type = 0, my_pos = 0;
    if (live && lane < EMB_SEQS_PER_WAVE && s0 + lane < nseq) {
        const size_t row = (size_t)(s0 + lane) * L + t;
        my_id = (int)ids[row];
        my_pos = pos_ids[row];
        if (num_types > 0 && type_ids) my_type = (int)type_ids[row] & 1;
    }
    const int p0 = __builtin_amdgcn_readlane(my_pos, 0);
    if (live) {
        for (int rr = 0; rr < EMB_SEQS_PER_WAVE && s0 + rr < nseq; ++rr) {
            const size_t base = ((size_t)(s0 + rr) * L + t) * H;
            const size_t wrow = (size_t)__builtin_amdgcn_readlane(my_id, rr) * H;
            const int ty = __builtin_amdgcn_readlane(my_type, rr);
            const int p = __builtin_amdgcn_readlane(my_pos, rr);
#pragma unroll
            for (int k = 0; k < KMAX; ++k) {
                const int c = lane + 64 * k;
                if (c < H) {
                    const float v = ds[base + c];
                    atomicAdd(dword + wrow + c, v);
                    if (ty == 0) acc0[k] += v; else acc1[k] += v;       // uniform branches
                    if (p == p0) accp[k] += v;
                    else if (v != 0.f) atomicAdd(dpos + (size_t)p * H + c, v);
                }
            }
        }
#pragma unroll
        for (int k = 0; k < KMAX; ++k) {
            const int c = lane + 64 * k;
            if (c < H && accp[k] != 0.f) atomicAdd(dpos + (size_t)p0 * H + c, accp[k]);
        }
    }
    if (num_types <= 0) return;                                  // uniform
#pragma unroll
    for (int k = 0; k < KMAX; ++k) {
        const int c = lane + 64 * k;
        if (c < H) { sh[(wave * 2 + 0) * H + c] = acc0[k]; sh[(wave * 2 + 1) * H + c] = acc1[k]; }
    }
    __syncthreads();
    for (int c = threadIdx.x; c < min(num_types, 2) * H; c += 256) {
        const int ty = c / H, cc = c - ty * H;
        const float v = (sh[(0 * 2 + ty) * H + cc] + sh[(1 * 2 + ty) * H + cc]) + (sh[(2 * 2 + ty) * H + cc] + sh[(3 * 2 + ty) * H + cc]);
        if (v != 0.f) atomicAdd(dtype_ + c, v);
    }
}

// drop_state / drop_snapshot (both or neither): the first thread also advances the dropout step counter and copies the
// four state words next to the activations of the forward at hand -- a training forward's whole prologue in one launch
// (they were three: 5 us each, all launch latency)
__global__ void position_ids_kernel(const int64_t* ids, int nseq, int L, int arch, int pad_id, int32_t* pos,
                                    uint32_t* drop_state, uint32_t* drop_snapshot) {
    const int s = blockIdx.x * blockDim.x + threadIdx.x;
    if (s == 0 && drop_state) {
        drop_state[2] += 1u;
#pragma unroll
        for (int i = 0; i < 4; ++i) drop_snapshot[i] = drop_state[i];
    }
    if (s >= nseq) return;
    if (arch == QST_ARCH_BERT) {
        for (int t = 0; t < L; ++t) pos[s * L + t] = t;
    } else {
        int cum = 0;                                   // modeling_mpnet.py:873-881
        for (int t = 0; t < L; ++t) {
            const int m = ids[(size_t)s * L + t] != pad_id;
            cum += m;
            pos[s * L + t] = cum * m + pad_id;
        }
    }
}

// mean pool + normalise (sentence-transformers Pooling(mean) + Normalize). One workgroup of 8 waves per sequence. A wave takes every 8th token row, a lane two columns per 64 (8-byte accesses,
// a row is read / written by whole 512-byte wave instructions); rows of padding are not read. (One thread per column
// walking the L rows one after the other, 256 threads per sequence, ran at 2.2 TB/s.)
constexpr int POOL_WAVES = 8, POOL_VPL = 8;                // H <= 1024
__global__ __launch_bounds__(64 * POOL_WAVES) void pool_norm_fwd_kernel(const float* tok, const int64_t* mask, int L, int H,
                                                                        int normalize, float* emb, float* pooled) {
    __shared__ float red[POOL_WAVES];
    __shared__ float msk[512];
    extern __shared__ __attribute__((aligned(16))) char smem[];   // [POOL_WAVES][H] floats
    float* part = (float*)smem;
    const int s = blockIdx.x, tid = threadIdx.x, lane = tid & 63, wave = tid >> 6, nv = H >> 1;
    for (int t = tid; t < L; t += 64 * POOL_WAVES) msk[t] = (float)mask[(size_t)s * L + t];
    __syncthreads();
    float cnt = 0.f;
    for (int t = 0; t < L; ++t) cnt += msk[t];
    const float inv = 1.0f / fmaxf(cnt, 1e-9f);
    f32x2 acc[POOL_VPL];
#pragma unroll
    for (int i = 0; i < POOL_VPL; ++i) { acc[i][0] = 0.f; acc[i][1] = 0.f; }
    // four rows of this wave in flight at a time
    for (int t0 = wave; t0 < L; t0 += 4 * POOL_WAVES) {
        f32x2 v[4][POOL_VPL];
        float m[4];
#pragma unroll
        for (int u = 0; u < 4; ++u) {
            const int t = t0 + u * POOL_WAVES;
            m[u] = t < L ? msk[t] : 0.f;
            const float* row = tok + ((size_t)s * L + (t < L ? t : 0)) * H;
#pragma unroll
            for (int i = 0; i < POOL_VPL; ++i) {
                const int c = lane + 64 * i;
                v[u][i][0] = v[u][i][1] = 0.f;
                if (c < nv && m[u] != 0.f) v[u][i] = *(const f32x2*)(row + 2 * c);       // (m: uniform)
            }
        }
#pragma unroll
        for (int u = 0; u < 4; ++u)
#pragma unroll
            for (int i = 0; i < POOL_VPL; ++i) { acc[i][0] += v[u][i][0] * m[u]; acc[i][1] += v[u][i][1] * m[u]; }
    }
#pragma unroll
    for (int i = 0; i < POOL_VPL; ++i) {
        const int c = lane + 64 * i;
        if (c < nv) *(f32x2*)(part + wave * H + 2 * c) = acc[i];
    }
    __syncthreads();
    float sq = 0.f;
    float e[2];
#pragma unroll
    for (int k = 0; k < 2; ++k) {
        const int c = tid + 64 * POOL_WAVES * k;
        e[k] = 0.f;
        if (c < H) {
            float a = 0.f;
#pragma unroll
            for (int w = 0; w < POOL_WAVES; ++w) a += part[w * H + c];
            e[k] = a * inv;
            sq += e[k] * e[k];
            if (pooled) pooled[(size_t)s * H + c] = e[k];
        }
    }
    sq = wave_sum(sq);
    if (lane == 0) red[wave] = sq;
    __syncthreads();
    float tot = 0.f;
#pragma unroll
    for (int w = 0; w < POOL_WAVES; ++w) tot += red[w];
    const float nrm = sqrtf(tot);
    const float sc = normalize ? 1.0f / fmaxf(nrm, 1e-12f) : 1.0f;
#pragma unroll
    for (int k = 0; k < 2; ++k) {
        const int c = tid + 64 * POOL_WAVES * k;
        if (c < H) emb[(size_t)s * H + c] = e[k] * sc;
    }
}

__global__ __launch_bounds__(64 * POOL_WAVES) void pool_norm_bwd_kernel(const float* demb, const float* pooled, const int64_t* mask,
                                                                        int L, int H, int normalize, float* dtok) {
    __shared__ float red[2 * POOL_WAVES];
    __shared__ float msk[512];
    extern __shared__ __attribute__((aligned(16))) char smem[];   // [H] floats: the gradient of the mean row
    float* dps = (float*)smem;
    const int s = blockIdx.x, tid = threadIdx.x, lane = tid & 63, wave = tid >> 6, nv = H >> 1;
    for (int t = tid; t < L; t += 64 * POOL_WAVES) msk[t] = (float)mask[(size_t)s * L + t];
    __syncthreads();
    float cnt = 0.f;
    for (int t = 0; t < L; ++t) cnt += msk[t];
    const float inv = 1.0f / fmaxf(cnt, 1e-9f);
    float e[2], g[2];
    float sq = 0.f, dot = 0.f;
#pragma unroll
    for (int k = 0; k < 2; ++k) {
        const int c = tid + 64 * POOL_WAVES * k;
        e[k] = g[k] = 0.f;
        if (c < H) {
            e[k] = pooled[(size_t)s * H + c];
            g[k] = demb[(size_t)s * H + c];
            sq += e[k] * e[k];
            dot += e[k] * g[k];
        }
    }
    sq = wave_sum(sq); dot = wave_sum(dot);
    if (lane == 0) { red[wave] = sq; red[POOL_WAVES + wave] = dot; }
    __syncthreads();
    sq = 0.f; dot = 0.f;
#pragma unroll
    for (int w = 0; w < POOL_WAVES; ++w) { sq += red[w]; dot += red[POOL_WAVES + w]; }
    const float nrm = sqrtf(sq);
#pragma unroll
    for (int k = 0; k < 2; ++k) {
        const int c = tid + 64 * POOL_WAVES * k;
        if (c < H) {
            float dp = g[k];
            if (normalize) {
                // y = e / max(n, eps): dy/de = (I - e e^T / n^2) / n for n > eps, I/eps otherwise
                if (nrm > 1e-12f) dp = (g[k] - e[k] * dot / sq) / nrm;
                else dp = g[k] / 1e-12f;
            }
            dps[c] = dp * inv;
        }
    }
    __syncthreads();
    f32x2 d[POOL_VPL];
#pragma unroll
    for (int i = 0; i < POOL_VPL; ++i) {
        const int c = lane + 64 * i;
        d[i][0] = d[i][1] = 0.f;
        if (c < nv) d[i] = *(const f32x2*)(dps + 2 * c);
    }
    for (int t = wave; t < L; t += POOL_WAVES) {
        const float m = msk[t];
        float* row = dtok + ((size_t)s * L + t) * H;
#pragma unroll
        for (int i = 0; i < POOL_VPL; ++i) {
            const int c = lane + 64 * i;
            if (c < nv) { f32x2 o; o[0] = d[i][0] * m; o[1] = d[i][1] * m; *(f32x2*)(row + 2 * c) = o; }
        }
    }
}

// MPNet relative position bias (modeling_mpnet.py:312-348): rel[a,i,j] = table[bucket(j-i)][a]. The bucket of each
// offset comes from a host-built LUT (qst_rel_bucket_host) so the float32 log matches torch's on the boundaries.
__global__ void rel_bias_fwd_kernel(const float* table, const int32_t* lut, int A, int L, float* out) {
    const int idx = blockIdx.x * blockDim.x + threadIdx.x;
    if (idx >= A * L * L) return;
    const int j = idx % L, i = (idx / L) % L, a = idx / (L * L);
    out[idx] = table[lut[j - i + 511] * A + a];
}
__global__ void rel_bias_bwd_kernel(const float* drel, const int32_t* lut, int buckets, int A, int L, float* dtable) {
    // block per (bucket, head): sum drel over the (i, j) that fall in the bucket (deterministic)
    const int b = blockIdx.x % buckets, a = blockIdx.x / buckets;
    float acc = 0.f;
    for (int idx = threadIdx.x; idx < L * L; idx += blockDim.x) {
        const int j = idx % L, i = idx / L;
        if (lut[j - i + 511] == b) acc += drel[(size_t)a * L * L + idx];
    }
    __shared__ float red[4];
    acc = wave_sum(acc);
    if ((threadIdx.x & 63) == 0) red[threadIdx.x >> 6] = acc;
    __syncthreads();
    if (threadIdx.x == 0) dtable[b * A + a] += red[0] + red[1] + red[2] + red[3];
}

// The same bias as relative-position vectors, the form the bf16 attention kernels consume:
// relpos[a][rp + L] = table[bucket(rp)][a] for rp = j - i in (-L, L); entry 0 is unused (zero).
__global__ void rel_pos_fwd_kernel(const float* table, const int32_t* lut, int A, int L, float* out) {
    const int idx = blockIdx.x * blockDim.x + threadIdx.x;
    if (idx >= A * 2 * L) return;
    const int t = idx % (2 * L), a = idx / (2 * L);
    out[idx] = t == 0 ? 0.f : table[lut[(t - L) + 511] * A + a];
}
// dtable[b][a] += sum over the relative positions of bucket b of drelpos[a][rp + L]; block per (bucket, head),
// fixed summation order (deterministic)
__global__ void rel_pos_bwd_kernel(const float* drelpos, const int32_t* lut, int buckets, int A, int L, float* dtable) {
    const int b = blockIdx.x % buckets, a = blockIdx.x / buckets;
    float acc = 0.f;
    for (int t = 1 + threadIdx.x; t < 2 * L; t += blockDim.x)
        if (lut[(t - L) + 511] == b) acc += drelpos[(size_t)a * 2 * L + t];
    __shared__ float red[4];
    acc = wave_sum(acc);
    if ((threadIdx.x & 63) == 0) red[threadIdx.x >> 6] = acc;
    __syncthreads();
    if (threadIdx.x == 0) dtable[b * A + a] += red[0] + red[1] + red[2] + red[3];
}

// bf16 shadow of a [rows, cols] fp32 matrix and its transpose, via a 32x32 LDS tile
__global__ __launch_bounds__(256) void shadow_kernel(const float* src, int rows, int cols, op16* dst, op16* dstT) {
    op_saturate(true);
    __shared__ float tile[32][33];
    const int c0 = blockIdx.x * 32, r0 = blockIdx.y * 32;
    const int tx = threadIdx.x & 31, ty = threadIdx.x >> 5;
    for (int k = 0; k < 4; ++k) {
        const int r = r0 + ty + 8 * k, c = c0 + tx;
        float v = 0.f;
        if (r < rows && c < cols) {
            v = src[(size_t)r * cols + c];
            if (dst) dst[(size_t)r * cols + c] = f2op(v);
        }
        tile[ty + 8 * k][tx] = v;
    }
    __syncthreads();
    if (dstT)
        for (int k = 0; k < 4; ++k) {
            const int c = c0 + ty + 8 * k, r = r0 + tx;
            if (r < rows && c < cols) dstT[(size_t)c * rows + r] = f2op(tile[tx][ty + 8 * k]);
        }
}

// all GEMM weights in ONE launch: table entry = {src offset, rows, cols, dst offset, dstT offset, first block}
// shadow_lo (nullable, QST_PREC_F16W): the low halves of the split weights, lo = op16(w - float(op16(w))), at the offsets of the W
// copies in a second arena of the same layout (W = hi + lo to ~2^-22: mostly SUBNORMAL halves, which the f16 MFMA keeps --
// tools/probe/mfma_f16_denorm_probe.hip)
__global__ __launch_bounds__(256) void shadow_all_kernel(const float* params, op16* shadow, const int64_t* tab, int nseg,
                                                         op16* shadow_lo) {
    op_saturate(true);
    __shared__ float tile[32][33];
    int lo = 0, hi = nseg - 1;
    while (lo < hi) {                                   // last segment whose first block <= blockIdx.x
        const int mid = (lo + hi + 1) >> 1;
        if (tab[mid * 6 + 5] <= (int64_t)blockIdx.x) lo = mid; else hi = mid - 1;
    }
    const int64_t* e = tab + lo * 6;
    const float* src = params + e[0];
    const int rows = (int)e[1], cols = (int)e[2];
    op16* dst = shadow + e[3];
    op16* dstT = shadow + e[4];
    const int b = blockIdx.x - (int)e[5];
    const int tiles_x = (cols + 31) / 32;
    const int c0 = (b % tiles_x) * 32, r0 = (b / tiles_x) * 32;
    const int tx = threadIdx.x & 31, ty = threadIdx.x >> 5;
    for (int k = 0; k < 4; ++k) {
        const int r = r0 + ty + 8 * k, c = c0 + tx;
        float v = 0.f;
        if (r < rows && c < cols) {
            v = src[(size_t)r * cols + c];
            const op16 h = f2op(v);
            dst[(size_t)r * cols + c] = h;
            if (shadow_lo) shadow_lo[e[3] + (size_t)r * cols + c] = f2op(v - op2f(h));
        }
        tile[ty + 8 * k][tx] = v;
    }
    __syncthreads();
    for (int k = 0; k < 4; ++k) {
        const int c = c0 + ty + 8 * k, r = r0 + tx;
        if (r < rows && c < cols) dstT[(size_t)c * rows + r] = f2op(tile[tx][ty + 8 * k]);
    }
}

#define QST_VPL_DISPATCH(H, CALL)                           \
    do {                                                    \
        const int _vpl = ((H) / 2 + 63) / 64;               \
        switch (_vpl) {                                     \
            case 1: { constexpr int VPL = 1; CALL; } break; \
            case 2: { constexpr int VPL = 2; CALL; } break; \
            case 3: { constexpr int VPL = 3; CALL; } break; \
            case 4: { constexpr int VPL = 4; CALL; } break; \
            case 5: case 6: { constexpr int VPL = 6; CALL; } break; \
            case 7: case 8: { constexpr int VPL = 8; CALL; } break; \
            default: return QST_ERR_UNSUPPORTED;            \
        }                                                   \
    } while (0)

}  // namespace

#if !QST_OP_F16
extern "C" int qst_embed_ln_fwd_mx(const int64_t* ids, const int64_t* type_ids, const int32_t* pos_ids,
                                   const float* word_emb, const float* pos_emb, const float* type_emb,
                                   const float* gamma, const float* beta, float eps, int M, int H,
                                   float* y, void* y_bf16, void* yq, void* ys, void* stream);
#endif  // !QST_OP_F16

static const QstDrop kNoDrop = {nullptr, 0u, 0u};
static int drop_ok(const QstDrop* d, int64_t elems) {
    if (!d) return QST_OK;
    if (d->thr16 > 65535u) return QST_ERR_BAD_ARG;
    if (d->thr16 && d->state && elems >= (int64_t)1 << 32) return QST_ERR_UNSUPPORTED;       // 32-bit element counters
    return QST_OK;
}
extern "C" int QST_K(qst_embed_ln_fwd_drop)(const int64_t* ids, const int64_t* type_ids, const int32_t* pos_ids,
                                     const float* word_emb, const float* pos_emb, const float* type_emb,
                                     const float* gamma, const float* beta, float eps, int M, int H,
                                     float* y, void* y_bf16, void* xhat_bf16, float* rstd, const QstDrop* drop, void* stream) {
    if (!ids || !pos_ids || !word_emb || !pos_emb || !gamma || !beta || !y || M <= 0 || H <= 0 || (H & 1))
        return QST_ERR_BAD_ARG;
    if (int rc = drop_ok(drop, (int64_t)M * H)) return rc;
    hipStream_t st = (hipStream_t)stream;
    QST_VPL_DISPATCH(H, (embed_ln_fwd_kernel<VPL><<<(M + 4 * EMB_ROWS - 1) / (4 * EMB_ROWS), 256, 0, st>>>(
                            ids, type_ids, pos_ids, word_emb, pos_emb, type_emb, gamma, beta, eps, M, H, y,
                            (op16*)y_bf16, (op16*)xhat_bf16, rstd, nullptr, nullptr, drop ? *drop : kNoDrop)));
    QST_LAUNCH_CHECK();
    return QST_OK;
}
extern "C" int QST_K(qst_embed_ln_fwd)(const int64_t* ids, const int64_t* type_ids, const int32_t* pos_ids,
                                const float* word_emb, const float* pos_emb, const float* type_emb,
                                const float* gamma, const float* beta, float eps, int M, int H,
                                float* y, void* y_bf16, void* xhat_bf16, float* rstd, void* stream) {
    return QST_K(qst_embed_ln_fwd_drop)(ids, type_ids, pos_ids, word_emb, pos_emb, type_emb, gamma, beta, eps, M, H, y, y_bf16,
                                 xhat_bf16, rstd, nullptr, stream);
}

#if !QST_OP_F16
extern "C" int qst_embed_ln_fwd_mx(const int64_t* ids, const int64_t* type_ids, const int32_t* pos_ids,
                                   const float* word_emb, const float* pos_emb, const float* type_emb,
                                   const float* gamma, const float* beta, float eps, int M, int H,
                                   float* y, void* y_bf16, void* yq, void* ys, void* stream) {
    if (!ids || !pos_ids || !word_emb || !pos_emb || !gamma || !beta || !y || !yq || !ys || M <= 0 || H <= 0) return QST_ERR_BAD_ARG;
    if (H % 64 != 0) return QST_ERR_UNSUPPORTED;
    hipStream_t st = (hipStream_t)stream;
    QST_VPL_DISPATCH(H, (embed_ln_fwd_kernel<VPL><<<(M + 4 * EMB_ROWS - 1) / (4 * EMB_ROWS), 256, 0, st>>>(
                            ids, type_ids, pos_ids, word_emb, pos_emb, type_emb, gamma, beta, eps, M, H, y,
                            (op16*)y_bf16, nullptr, nullptr, (uint8_t*)yq, (uint8_t*)ys, kNoDrop)));
    QST_LAUNCH_CHECK();
    return QST_OK;
}

extern "C" int qst_ln_fwd_mx(const float* s, const float* gamma, const float* beta, float eps, int M, int H,
                             float* y, void* y_bf16, void* yq, void* ys, void* stream) {
    if (!s || !gamma || !beta || !y || !yq || !ys || M <= 0 || H <= 0) return QST_ERR_BAD_ARG;
    if (H % 64 != 0) return QST_ERR_UNSUPPORTED;
    hipStream_t st = (hipStream_t)stream;
    QST_VPL_DISPATCH(H, (ln_fwd_kernel<VPL><<<(M + 3) / 4, 256, 0, st>>>(s, gamma, beta, eps, M, H, y, (op16*)y_bf16, nullptr,
                                                                         nullptr, (uint8_t*)yq, (uint8_t*)ys)));
    QST_LAUNCH_CHECK();
    return QST_OK;
}

// The same two with everything a backward needs as well (fp8 TRAINING forward: the LayerNorm output goes to the next GEMM as
// MXFP8 and to the bf16 backward as y_bf16 / xhat / rstd)
extern "C" int qst_embed_ln_fwd_mx_train(const int64_t* ids, const int64_t* type_ids, const int32_t* pos_ids,
                                         const float* word_emb, const float* pos_emb, const float* type_emb,
                                         const float* gamma, const float* beta, float eps, int M, int H, float* y,
                                         void* y_bf16, void* xhat_bf16, float* rstd, void* yq, void* ys,
                                         const QstDrop* drop, void* stream) {
    if (!ids || !pos_ids || !word_emb || !pos_emb || !gamma || !beta || !y || !y_bf16 || !xhat_bf16 || !rstd || !yq || !ys ||
        M <= 0 || H <= 0)
        return QST_ERR_BAD_ARG;
    if (H % 64 != 0) return QST_ERR_UNSUPPORTED;
    if (int rc = drop_ok(drop, (int64_t)M * H)) return rc;
    hipStream_t st = (hipStream_t)stream;
    QST_VPL_DISPATCH(H, (embed_ln_fwd_kernel<VPL><<<(M + 4 * EMB_ROWS - 1) / (4 * EMB_ROWS), 256, 0, st>>>(
                            ids, type_ids, pos_ids, word_emb, pos_emb, type_emb, gamma, beta, eps, M, H, y,
                            (op16*)y_bf16, (op16*)xhat_bf16, rstd, (uint8_t*)yq, (uint8_t*)ys, drop ? *drop : kNoDrop)));
    QST_LAUNCH_CHECK();
    return QST_OK;
}
extern "C" int qst_ln_fwd_mx_train(const float* s, const float* gamma, const float* beta, float eps, int M, int H, float* y,
                                   void* y_bf16, void* xhat_bf16, float* rstd, void* yq, void* ys, void* stream) {
    if (!s || !gamma || !beta || !y || !y_bf16 || !xhat_bf16 || !rstd || !yq || !ys || M <= 0 || H <= 0) return QST_ERR_BAD_ARG;
    if (H % 64 != 0) return QST_ERR_UNSUPPORTED;
    hipStream_t st = (hipStream_t)stream;
    QST_VPL_DISPATCH(H, (ln_fwd_kernel<VPL><<<(M + 3) / 4, 256, 0, st>>>(s, gamma, beta, eps, M, H, y, (op16*)y_bf16,
                                                                         (op16*)xhat_bf16, rstd, (uint8_t*)yq, (uint8_t*)ys)));
    QST_LAUNCH_CHECK();
    return QST_OK;
}
#endif  // !QST_OP_F16

extern "C" int QST_K(qst_ln_fwd)(const float* s, const float* gamma, const float* beta, float eps, int M, int H,
                          float* y, void* y_bf16, void* xhat_bf16, float* rstd, void* stream) {
    if (!s || !gamma || !beta || !y || M <= 0 || H <= 0 || (H & 1)) return QST_ERR_BAD_ARG;
    hipStream_t st = (hipStream_t)stream;
    QST_VPL_DISPATCH(H, (ln_fwd_kernel<VPL><<<(M + 3) / 4, 256, 0, st>>>(s, gamma, beta, eps, M, H, y, (op16*)y_bf16,
                                                                         (op16*)xhat_bf16, rstd, nullptr, nullptr)));
    QST_LAUNCH_CHECK();
    return QST_OK;
}

#if !QST_OP_F16
// Batched second stage for several LayerNorms at once (one launch per backward call instead of one per LayerNorm)
__global__ __launch_bounds__(256) void ln_bwd_reduce_batch_kernel(QstLnReduceBatch b) {
    const int which = blockIdx.z;
    const int col = blockIdx.x * 256 + threadIdx.x;
    if (col >= 2 * b.H) return;
    const float* partials = b.partials[which];
    float acc = 0.f;
    const int nb = b.nblocks_each[which] > 0 ? b.nblocks_each[which] : b.nblocks;
    // four independent loads in flight per thread: the loop is latency-bound (a row of partials is 3 KB)
    const size_t rs = (size_t)2 * b.H, gs = gridDim.y;
    int r = blockIdx.y;
    float a0 = 0.f, a1 = 0.f, a2 = 0.f, a3 = 0.f;
    for (; r + 3 * (int)gs < nb; r += 4 * (int)gs) {
        a0 += partials[(size_t)r * rs + col];
        a1 += partials[((size_t)r + gs) * rs + col];
        a2 += partials[((size_t)r + 2 * gs) * rs + col];
        a3 += partials[((size_t)r + 3 * gs) * rs + col];
    }
    for (; r < nb; r += (int)gs) acc += partials[(size_t)r * rs + col];
    acc += (a0 + a1) + (a2 + a3);
    atomicAdd(col < b.H ? b.dgamma[which] + col : b.dbeta[which] + (col - b.H), acc);
}

extern "C" int qst_ln_bwd_reduce_batch(const QstLnReduceBatch* b, void* stream) {
    if (!b || b->count <= 0 || b->count > QST_LN_BATCH_MAX || b->H <= 0 || b->nblocks <= 0) return QST_ERR_BAD_ARG;
    ln_bwd_reduce_batch_kernel<<<dim3((2 * b->H + 255) / 256, 32, b->count), 256, 0, (hipStream_t)stream>>>(*b);
    QST_LAUNCH_CHECK();
    return QST_OK;
}

extern "C" size_t qst_ln_bwd_scratch_bytes(int M, int H) {
    const int rows_per_block = 4 * LN_BWD_ROWS_PER_WAVE;
    return (size_t)((M + rows_per_block - 1) / rows_per_block) * 2 * H * sizeof(float);
}
#endif  // !QST_OP_F16

extern "C" int QST_K(qst_ln_bwd)(const float* dy, const void* xhat_bf16, const float* rstd, const float* gamma, int M, int H,
                          float* ds, void* ds_bf16, float* dgamma, float* dbeta, float* scratch, void* stream) {
    return QST_K(qst_ln_bwd_drop)(dy, xhat_bf16, rstd, gamma, M, H, ds, ds_bf16, dgamma, dbeta, scratch, nullptr, nullptr, stream);
}
extern "C" int QST_K(qst_ln_bwd_drop)(const float* dy, const void* xhat_bf16, const float* rstd, const float* gamma, int M, int H,
                               float* ds, void* ds_bf16, float* dgamma, float* dbeta, float* scratch,
                               const QstDrop* drop_in, const QstDrop* drop_out, void* stream) {
    if (int rc = drop_ok(drop_in, (int64_t)M * H)) return rc;
    if (int rc = drop_ok(drop_out, (int64_t)M * H)) return rc;
    // dgamma == dbeta == NULL with a scratch buffer: only write the per-block partials; the caller reduces them later
    // with qst_ln_bwd_reduce_batch (one launch for many LayerNorms)
    const bool deferred = !dgamma && !dbeta && scratch;
    if (!dy || !xhat_bf16 || !rstd || !gamma || !ds || (!deferred && (!dgamma || !dbeta)) || M <= 0 || H <= 0 || (H & 1))
        return QST_ERR_BAD_ARG;
    hipStream_t st = (hipStream_t)stream;
    const int rows_per_block = 4 * LN_BWD_ROWS_PER_WAVE;
    const int grid = (M + rows_per_block - 1) / rows_per_block;
    const size_t lds = (size_t)8 * H * sizeof(float);
    QST_VPL_DISPATCH(H, (ln_bwd_kernel<VPL><<<grid, 256, lds, st>>>(dy, (const op16*)xhat_bf16, rstd, gamma, M, H, ds,
                                                                   (op16*)ds_bf16, dgamma, dbeta, scratch,
                                                                   drop_in ? *drop_in : kNoDrop, drop_out ? *drop_out : kNoDrop)));
    QST_LAUNCH_CHECK();
    if (scratch && !deferred) {
        ln_bwd_reduce_kernel<<<dim3((2 * H + 255) / 256, 32), 256, 0, st>>>(scratch, grid, H, dgamma, dbeta);
        QST_LAUNCH_CHECK();
    }
    return QST_OK;
}

#if !QST_OP_F16
extern "C" int qst_embed_bwd(const float* ds, const int64_t* ids, const int64_t* type_ids, const int32_t* pos_ids,
                             int nseq, int L, int H, int num_types, float* dword, float* dpos, float* dtype_,
                             void* stream) {
    if (!ds || !ids || !pos_ids || !dword || !dpos || nseq <= 0 || L <= 0 || H <= 0) return QST_ERR_BAD_ARG;
    if (num_types > 2) return QST_ERR_UNSUPPORTED;
    if (num_types > 0 && !dtype_) return QST_ERR_BAD_ARG;
    hipStream_t st = (hipStream_t)stream;
    const int M = nseq * L;
    if (H > 1024) return QST_ERR_UNSUPPORTED;
    const int grid = ((L + 3) / 4) * ((nseq + EMB_SEQS_PER_WAVE - 1) / EMB_SEQS_PER_WAVE);
    embed_bwd_kernel<<<grid, 256, (size_t)8 * H * sizeof(float), st>>>(ds, ids, type_ids, pos_ids, nseq, L, H,
                                                                       num_types, dword, dpos, dtype_);
    QST_LAUNCH_CHECK();
    return QST_OK;
}

extern "C" int qst_position_ids(const int64_t* ids, int nseq, int L, int arch, int pad_id, int32_t* pos_ids,
                                void* stream) {
    return qst_forward_prologue(ids, nseq, L, arch, pad_id, pos_ids, nullptr, nullptr, stream);
}
extern "C" int qst_forward_prologue(const int64_t* ids, int nseq, int L, int arch, int pad_id, int32_t* pos_ids,
                                    uint32_t* drop_state, uint32_t* drop_snapshot, void* stream) {
    if (!ids || !pos_ids || nseq <= 0 || L <= 0 || ((drop_state == nullptr) != (drop_snapshot == nullptr))) return QST_ERR_BAD_ARG;
    position_ids_kernel<<<(nseq + 63) / 64, 64, 0, (hipStream_t)stream>>>(ids, nseq, L, arch, pad_id, pos_ids, drop_state,
                                                                          drop_snapshot);
    QST_LAUNCH_CHECK();
    return QST_OK;
}

extern "C" int qst_pool_norm_fwd(const float* tok, const int64_t* mask, int nseq, int L, int H, int normalize,
                                 float* emb, float* pooled, void* stream) {
    if (!tok || !mask || !emb || nseq <= 0 || L <= 0 || H <= 0) return QST_ERR_BAD_ARG;
    if (L > 512 || H > 1024) return QST_ERR_UNSUPPORTED;
    if (H & 1) return QST_ERR_UNSUPPORTED;
    pool_norm_fwd_kernel<<<nseq, 64 * POOL_WAVES, (size_t)POOL_WAVES * H * sizeof(float), (hipStream_t)stream>>>(tok, mask, L, H, normalize,
                                                                                                           emb, pooled);
    QST_LAUNCH_CHECK();
    return QST_OK;
}

extern "C" int qst_pool_norm_bwd(const float* demb, const float* pooled, const int64_t* mask, int nseq, int L, int H,
                                 int normalize, float* dtok, void* stream) {
    if (!demb || !pooled || !mask || !dtok || nseq <= 0 || L <= 0 || H <= 0) return QST_ERR_BAD_ARG;
    if (L > 512 || H > 1024) return QST_ERR_UNSUPPORTED;
    if (H & 1) return QST_ERR_UNSUPPORTED;
    pool_norm_bwd_kernel<<<nseq, 64 * POOL_WAVES, (size_t)H * sizeof(float), (hipStream_t)stream>>>(demb, pooled, mask, L, H, normalize, dtok);
    QST_LAUNCH_CHECK();
    return QST_OK;
}

extern "C" int qst_rel_bias_fwd(const float* table, const int32_t* lut, int A, int L, float* rel_bias, void* stream) {
    if (!table || !lut || !rel_bias || A <= 0 || L <= 0 || L > 512) return QST_ERR_BAD_ARG;
    const int n = A * L * L;
    rel_bias_fwd_kernel<<<(n + 255) / 256, 256, 0, (hipStream_t)stream>>>(table, lut, A, L, rel_bias);
    QST_LAUNCH_CHECK();
    return QST_OK;
}
extern "C" int qst_rel_bias_bwd(const float* drel, const int32_t* lut, int buckets, int A, int L, float* dtable,
                                void* stream) {
    if (!drel || !lut || !dtable || A <= 0 || L <= 0 || L > 512 || buckets <= 0) return QST_ERR_BAD_ARG;
    rel_bias_bwd_kernel<<<buckets * A, 256, 0, (hipStream_t)stream>>>(drel, lut, buckets, A, L, dtable);
    QST_LAUNCH_CHECK();
    return QST_OK;
}

extern "C" int qst_rel_pos_fwd(const float* table, const int32_t* lut, int A, int L, float* rel_pos, void* stream) {
    if (!table || !lut || !rel_pos || A <= 0 || L <= 0 || L > 512) return QST_ERR_BAD_ARG;
    const int n = A * 2 * L;
    rel_pos_fwd_kernel<<<(n + 255) / 256, 256, 0, (hipStream_t)stream>>>(table, lut, A, L, rel_pos);
    QST_LAUNCH_CHECK();
    return QST_OK;
}
extern "C" int qst_rel_pos_bwd(const float* drel_pos, const int32_t* lut, int buckets, int A, int L, float* dtable,
                               void* stream) {
    if (!drel_pos || !lut || !dtable || A <= 0 || L <= 0 || L > 512 || buckets <= 0) return QST_ERR_BAD_ARG;
    rel_pos_bwd_kernel<<<buckets * A, 256, 0, (hipStream_t)stream>>>(drel_pos, lut, buckets, A, L, dtable);
    QST_LAUNCH_CHECK();
    return QST_OK;
}
#endif  // !QST_OP_F16

extern "C" int QST_K(qst_shadow_all)(const float* params, void* shadow, const int64_t* table_dev, int nseg, int nblocks, void* stream) {
    if (!params || !shadow || !table_dev || nseg <= 0 || nblocks <= 0) return QST_ERR_BAD_ARG;
    shadow_all_kernel<<<nblocks, 256, 0, (hipStream_t)stream>>>(params, (op16*)shadow, table_dev, nseg, nullptr);
    QST_LAUNCH_CHECK();
    return QST_OK;
}
#if QST_OP_F16
// ... and the low halves of the split weights into shadow_lo (QST_PREC_F16W; an arena of the same layout as `shadow`)
extern "C" int qst_shadow_all_split_f16(const float* params, void* shadow, void* shadow_lo, const int64_t* table_dev, int nseg,
                                        int nblocks, void* stream) {
    if (!params || !shadow || !shadow_lo || !table_dev || nseg <= 0 || nblocks <= 0) return QST_ERR_BAD_ARG;
    shadow_all_kernel<<<nblocks, 256, 0, (hipStream_t)stream>>>(params, (op16*)shadow, table_dev, nseg, (op16*)shadow_lo);
    QST_LAUNCH_CHECK();
    return QST_OK;
}
#endif

extern "C" int QST_K(qst_shadow_matrix)(const float* src, int rows, int cols, void* dst_bf16, void* dstT_bf16, void* stream) {
    if (!src || rows <= 0 || cols <= 0) return QST_ERR_BAD_ARG;
    shadow_kernel<<<dim3((cols + 31) / 32, (rows + 31) / 32), 256, 0, (hipStream_t)stream>>>(
        src, rows, cols, (op16*)dst_bf16, (op16*)dstT_bf16);
    QST_LAUNCH_CHECK();
    return QST_OK;
}

#if !QST_OP_F16
// ---------------------------------------------------------------- dropout state (include/qst.h, qst_kernels.h: QstDrop)
namespace {
__global__ void drop_init_kernel(uint32_t* st, uint32_t lo, uint32_t hi) { st[0] = lo; st[1] = hi; st[2] = 0u; st[3] = 0u; }
__global__ void drop_advance_kernel(uint32_t* st) { st[2] += 1u; }
__global__ __launch_bounds__(256) void drop_mult_kernel(QstDrop d, int probs, int64_t n, float* out) {
    const int64_t i = ((int64_t)blockIdx.x * 256 + threadIdx.x) * 4;
    if (i >= n) return;
    float m[4] = {1.f, 1.f, 1.f, 1.f};
    if (probs) {
        const DropCtx c = drop_ctx8(d);
        if (c.thr) {
            const uint32_t w = drop_word4(c, (uint32_t)i);
            m[0] = drop_keep_byte<0>(c, w) ? c.scale : 0.f; m[1] = drop_keep_byte<1>(c, w) ? c.scale : 0.f;
            m[2] = drop_keep_byte<2>(c, w) ? c.scale : 0.f; m[3] = drop_keep_byte<3>(c, w) ? c.scale : 0.f;
        }
    } else {
        const DropCtx c = drop_ctx(d);
        if (c.thr) { drop_pair(c, (uint32_t)i, m[0], m[1]); drop_pair(c, (uint32_t)i + 2, m[2], m[3]); }
    }
    for (int k = 0; k < 4 && i + k < n; ++k) out[i + k] = m[k];
}
// out[i] = in[i] * multiplier(i) (+ resid[i]): the hidden-state masks of the fp32 (parity-precision) training path, whose
// GEMMs have no dropout in their epilogues. in may alias out.
__global__ __launch_bounds__(256) void drop_apply_kernel(QstDrop d, const float* in, const float* resid, int64_t n, float* out) {
    const int64_t i = ((int64_t)blockIdx.x * 256 + threadIdx.x) * 4;
    if (i >= n) return;
    const DropCtx c = drop_ctx(d);
    float m[4] = {1.f, 1.f, 1.f, 1.f};
    if (c.thr) { drop_pair(c, (uint32_t)i, m[0], m[1]); drop_pair(c, (uint32_t)i + 2, m[2], m[3]); }
    if (i + 3 < n) {
        f32x4 v = *(const f32x4*)(in + i);
        v[0] *= m[0]; v[1] *= m[1]; v[2] *= m[2]; v[3] *= m[3];
        if (resid) v += *(const f32x4*)(resid + i);
        *(f32x4*)(out + i) = v;
    } else {
        for (int k = 0; k < 4 && i + k < n; ++k) out[i + k] = in[i + k] * m[k] + (resid ? resid[i + k] : 0.f);
    }
}
}  // namespace
extern "C" int qst_dropout_apply_f32(const QstDrop* d, const float* in, const float* resid, int64_t n, float* out, void* stream) {
    if (!d || !in || !out || n <= 0 || (n & 3)) return QST_ERR_BAD_ARG;
    if (int rc = drop_ok(d, n)) return rc;
    drop_apply_kernel<<<(unsigned)((n / 4 + 255) / 256), 256, 0, (hipStream_t)stream>>>(*d, in, resid, n, out);
    QST_LAUNCH_CHECK();
    return QST_OK;
}
extern "C" int qst_dropout_init(uint32_t* state_dev, uint64_t seed, void* stream) {
    if (!state_dev) return QST_ERR_BAD_ARG;
    drop_init_kernel<<<1, 1, 0, (hipStream_t)stream>>>(state_dev, (uint32_t)seed, (uint32_t)(seed >> 32));
    QST_LAUNCH_CHECK();
    return QST_OK;
}
extern "C" int qst_dropout_advance(uint32_t* state_dev, void* stream) {
    if (!state_dev) return QST_ERR_BAD_ARG;
    drop_advance_kernel<<<1, 1, 0, (hipStream_t)stream>>>(state_dev);
    QST_LAUNCH_CHECK();
    return QST_OK;
}
extern "C" int qst_dropout_multipliers(const QstDrop* d, int probs, int64_t n, float* out, void* stream) {
    if (!d || !out || n <= 0) return QST_ERR_BAD_ARG;
    if (int rc = drop_ok(d, n)) return rc;
    drop_mult_kernel<<<(unsigned)((n / 4 + 256) / 256), 256, 0, (hipStream_t)stream>>>(*d, probs, n, out);
    QST_LAUNCH_CHECK();
    return QST_OK;
}
#endif  // !QST_OP_F16

