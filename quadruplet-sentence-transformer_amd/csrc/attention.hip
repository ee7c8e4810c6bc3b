// attention.hip -- masked softmax self-attention forward and backward, LDS-staged, bf16 MFMA 32x32x16.
//
// Replaces BertSelfAttention eager path (modeling_bert.py:111-136: softmax(QK^T/sqrt(d) + mask) V) and
// MPNetSelfAttention (modeling_mpnet.py:149-158: + position_bias before the mask) and their autograd.
// Head dims 32 (MiniLM) and 64 (mpnet/bert-base); L multiple of 32, <= 512.
//
// Orientation (cdna guide section 3 "accumulator tile as the next MFMA's operand"):
//  fwd / dQ : S^T = K.Q^T  -> accumulator rows = keys (registers), column = query (lane): softmax statistics
//             are per lane; P^T feeds the next MFMA as the B operand with no lane movement:
//             O^T = V^T.P^T and dQ^T = K^T.dS^T, the A operand (V^T / K^T) by ds_read_b64_tr_b16.
//  dK/dV    : S = Q.K^T    -> rows = queries (registers), column = key (lane); dV^T = dO^T.P, dK^T = Q^T.dS.
// One workgroup = 4 waves = 128 queries (fwd, dQ) or 128 keys (dK/dV) of one (sequence, head); the other
// side is streamed through LDS in chunks of 128 rows.
#include <stdlib.h>
#include "qst_common.h"
#include "qst_kernels.h"

namespace {

constexpr float kMaskMin = -3.4028234663852886e38f;   // torch.finfo(float32).min, HF's additive mask value
constexpr float kLog2e = 1.4426950408889634f;
// The forward works in log2 units, where kMaskMin * log2(e) would overflow to -inf, and -inf - (-inf) is NaN as soon as
// a whole 32-key tile is masked before any unmasked key has been seen (left padding, an all-padding row). A finite
// constant that absorbs every real score gives what HF's finfo.min gives: masked keys weigh exactly 0 next to any
// unmasked key, and a row with no unmasked key attends uniformly (its pooled embedding is 0 either way: mask sum 0).
constexpr float kMaskLog2 = -3.0e38f;

template <int D> __device__ __forceinline__ uint32_t rr_off(int row, int chunk) {   // image for ds_read_b128 row reads
    if (D == 32) return (uint32_t)(row * 64 + ((chunk ^ ((row >> 2) & 3)) << 4));
    return (uint32_t)(row * 128 + ((chunk ^ ((row >> 1) & 7)) << 4));
}
template <int D> __device__ __forceinline__ uint32_t tr_off(int row, int byte) {     // image for transposed reads
    if (D == 32) return (uint32_t)(row * 64 + byte);
    return (uint32_t)(row * 128 + (byte ^ (((row >> 1) & 1) << 6)));
}
__device__ __forceinline__ op16x4 lds_tr(const char* p) {
    return lds_tr16_op(p);
}
// A-operand fragment (32 rows = dd block `ddb`, k = 16 rows of the image starting at `row0`), k order matching an
// accumulator-sourced B operand: element e of lane-half h <-> image row row0 + 8*(e>>2) + 4h + (e&3)
template <int D> __device__ __forceinline__ op16x8 tr_frag(const char* img, int row0, int ddb, int lane) {
    const int li = lane & 15, q = li >> 2, p = li & 3, gsel = (lane >> 4) & 1, h = lane >> 5;
    const int byte = ddb * 64 + gsel * 32 + 8 * p;
    const op16x4 a = lds_tr(img + tr_off<D>(row0 + 4 * h + q, byte));
    const op16x4 b = lds_tr(img + tr_off<D>(row0 + 8 + 4 * h + q, byte));
    op16x8 f;
#pragma unroll
    for (int e = 0; e < 4; ++e) { f[e] = a[e]; f[4 + e] = b[e]; }
    return f;
}
__device__ __forceinline__ op16x8 acc_frag(const f32x16& acc, int s) {
    op16x8 f;
#pragma unroll
    for (int e = 0; e < 8; ++e) f[e] = (op16)acc[8 * s + e];
    return f;
}
__device__ __forceinline__ int acc_row(int r, int h) { return (r & 3) + 8 * (r >> 2) + 4 * h; }

// Relative-position gradient of one 32 x 32 dS tile (rows = queries in registers, column = key on the lane; rows of
// half h are (r&3) + 8(r>>2) + 4h): the bias gradient only needs the sums along the tile's 63 diagonals j - i. Rotating
// row i by 31 - i lanes lines every diagonal up on one lane: lane t then holds slot t (diagonal t - 31) if t >= 31 - i,
// slot 32 + t (diagonal t + 1) otherwise. 16 ds_bpermute + selects per tile; the first version added every element to an
// LDS array with ds_add_f32, 16 per lane per tile, which more than doubled the kernel (269 -> 656 us at L = 256).
// lo = slots 0..31, hi = slots 32..62 (lane 31: 0); diag_store combines the two half-waves.
__device__ __forceinline__ void diag_add(float v_in, int r, int lane, float& lo, float& hi) {      // one tile element
    const int t = lane & 31, h = lane >> 5;
    const int k = 31 - ((r & 3) + 8 * (r >> 2) + 4 * h);
    const int src = ((t - k) & 31) + 32 * h;
    const float v = __builtin_bit_cast(float, __builtin_amdgcn_ds_bpermute(src * 4, __builtin_bit_cast(int, v_in)));
    lo += (t >= k) ? v : 0.f;
    hi += (t < k) ? v : 0.f;
}
// add a tile's diagonal sums into this wave's private [2L] array: slot s <-> relative position (j0 - i0) + s - 31
__device__ __forceinline__ void diag_store(float* drel_w, int L, int j0, int i0, int lane, float lo, float hi) {
    lo += swap32(lo);
    hi += swap32(hi);
    if (lane < 32) {
        const int base = (j0 - i0) - 31 + L + lane;
        drel_w[base] += lo;
        if (lane < 31) drel_w[base + 32] += hi;
    }
}

// Stage a 128-row chunk (rows valid) of up to three [rows, D] bf16 tensors into LDS images. ALL global loads of the
// chunk are issued before the first LDS store (one round trip instead of one per image: the per-image version spent
// two thirds of the wave's life waiting), and a tensor needed both row-wise and transposed is loaded once.
template <int D>
struct Stager {
    static constexpr int CPR = D / 8;                 // 16-byte chunks per row
    static constexpr int PER = 128 * CPR / 256;       // chunks per thread per tensor (2 for d=32, 4 for d=64)
    u32x4 v[3][PER];
    __device__ __forceinline__ void load(int t, const op16* g, int ld, int rows, int tid) {
#pragma unroll
        for (int k = 0; k < PER; ++k) {
            const int idx = tid + 256 * k, row = idx / CPR, c = idx % CPR;
            const u32x4 z = {0, 0, 0, 0};
            v[t][k] = row < rows ? *(const u32x4*)(g + (size_t)row * ld + c * 8) : z;
        }
    }
    template <bool TR>
    __device__ __forceinline__ void store(int t, char* img, int tid) const {
#pragma unroll
        for (int k = 0; k < PER; ++k) {
            const int idx = tid + 256 * k, row = idx / CPR, c = idx % CPR;
            *(u32x4*)(img + (TR ? tr_off<D>(row, c * 16) : rr_off<D>(row, c))) = v[t][k];
        }
    }
};

// rel / drel: MPNet's position bias and its gradient as RELATIVE-POSITION vectors [A][2L] (entry j - i + L; the bias
// depends on j - i only, modeling_mpnet.py:312-348). A workgroup keeps its head's vector in LDS: the earlier [A, L, L]
// table cost one uncoalesced 4-byte gather per score element (lanes 1 KB apart) and one global float atomic per
// element for the gradient -- forward 227 vs 70 us and backward 677 vs 251 us at L = 256, d = 64.
struct AttnArgs {
    const op16* qkv; const op16* ctx; const op16* dctx; const float* lse_in; const int64_t* mask;
    const float* rel; op16* out; op16* dqkv; float* lse_out; float* drel; float* delta;
    int nseq, L, A, H; float scale;
    // q / k / v (and their gradients) of one (sequence, head) are [L] rows of `ld` = 3H elements of the token-major
    // [M, 3H] tensor (q | k | v, heads concatenated) starting at qkv_base(); the k and v parts follow at + woff = H and
    // + 2 woff. A head's rows are 2d-byte slices of 6H-byte rows (d = 32: half a cache line each).
    int ld; int64_t woff;
    // dropout of the probabilities (training): element ((seq * A + head) * L + query) * L + key of QstDrop's counter space;
    // kernels instantiated with DROP recompute the mask wherever P or dP appears (nothing is stored)
    QstDrop drop;
};
__device__ __forceinline__ size_t qkv_base(const AttnArgs& a, int seq, int head, int D) {
    return (size_t)seq * a.L * a.ld + (size_t)head * D;
}

// ------------------------------------------------------------------ forward
template <int D, bool DROP = false>
__global__ __launch_bounds__(256, D == 32 ? 4 : 2) void attn_fwd_kernel(AttnArgs a) {
    op_saturate(true);
    extern __shared__ __attribute__((aligned(16))) char smem[];
    constexpr int KS = D / 16, DB = D / 32, IMG = 128 * D * 2;
    char* kimg = smem;                  // row-read image of the K chunk
    char* vimg = smem + IMG;            // transposed-read image of the V chunk
    float* madd = (float*)(smem + 2 * IMG);   // additive key mask for the whole sequence [L], in log2 units
    char* ostg = smem + 2 * IMG + ((a.L * 4 + 15) & ~15);   // output staging, 32 rows x D bf16 per wave
    float* relv = (float*)(ostg + 4 * 32 * D * 2);           // this head's relative-position bias [2L], log2 units
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6, h = lane >> 5, fr = lane & 31;
    const int nqb = (a.L + 127) / 128;
    // d = 32: heads 2k and 2k+1 of a token share every 128-byte line of q/k/v. Workgroups b and b+8 run on the same
    // XCD, so give them the two heads of a pair: the line then leaves HBM once and the partner finds it in that L2
    // (consecutive workgroup ids go to different XCDs: both would fetch it).
    int bid = blockIdx.x;
    if (D == 32 && nqb == 1 && (a.A & 1) == 0 && (gridDim.x & 15) == 0)
        bid = 2 * ((bid >> 4) * 8 + (bid & 7)) + ((bid >> 3) & 1);
    const int qb = bid % nqb, head = (bid / nqb) % a.A, seq = bid / (nqb * a.A);
    const int ld = a.ld;
    const op16* base = a.qkv + qkv_base(a, seq, head, D);
    const int i0 = qb * 128 + wave * 32;
    const bool active = i0 < a.L;
    const int qi = i0 + fr;

    // softmax in base 2: scores are scaled by scale*log2(e) and fed to v_exp_f32 directly (one multiply and one
    // exp per element instead of scale, subtract, multiply-by-log2e, exp); masked keys: kMaskLog2 (finite, see above)
    for (int t = tid; t < a.L; t += 256) madd[t] = a.mask[(size_t)seq * a.L + t] ? 0.f : kMaskLog2;
    const float sc2 = a.scale * kLog2e;
    if (a.rel)
        for (int t = tid; t < 2 * a.L; t += 256) relv[t] = kLog2e * a.rel[(size_t)head * 2 * a.L + t];

    op16x8 qf[KS];
    if (active) {
#pragma unroll
        for (int s = 0; s < KS; ++s) qf[s] = *(const op16x8*)(base + (size_t)qi * ld + 16 * s + 8 * h);
    }
    f32x16 o[DB];
#pragma unroll
    for (int b = 0; b < DB; ++b)
#pragma unroll
        for (int r = 0; r < 16; ++r) o[b][r] = 0.f;
    float m = -INFINITY, l = 0.f;
    const DropCtx dc = DROP ? drop_ctx8(a.drop) : DropCtx{0u, 0u, 1.f};
    const uint32_t drow = ((uint32_t)(seq * a.A + head) * a.L + qi) * a.L;      // this lane's query row of the mask

    const int nchunk = (a.L + 127) / 128;
    for (int c = 0; c < nchunk; ++c) {
        const int rows = min(128, a.L - c * 128);
        __syncthreads();
        {
            Stager<D> sg;
            sg.load(0, base + (size_t)c * 128 * ld + a.woff, ld, rows, tid);
            sg.load(1, base + (size_t)c * 128 * ld + 2 * a.woff, ld, rows, tid);
            sg.template store<false>(0, kimg, tid);
            sg.template store<true>(1, vimg, tid);
        }
        __syncthreads();
        if (!active) continue;
        for (int jt = 0; jt < rows / 32; ++jt) {
            f32x16 s;
#pragma unroll
            for (int r = 0; r < 16; ++r) s[r] = 0.f;
#pragma unroll
            for (int ks = 0; ks < KS; ++ks) {
                const op16x8 kf = *(const op16x8*)(kimg + rr_off<D>(jt * 32 + fr, 2 * ks + h));
                s = mfma32_op(kf, qf[ks], s);
            }
            const int j0 = c * 128 + jt * 32;
            float mx = -INFINITY;
#pragma unroll
            for (int r = 0; r < 16; ++r) {
                const int j = j0 + acc_row(r, h);
                float v = s[r] * sc2 + madd[j];
                if (a.rel) v += relv[j - qi + a.L];
                s[r] = v;
                mx = fmaxf(mx, v);
            }
            mx = fmaxf(mx, swap32(mx));
            const float mn = fmaxf(m, mx);
            const float alpha = __builtin_amdgcn_exp2f(m - mn);
            float ps = 0.f;
#pragma unroll
            for (int r = 0; r < 16; ++r) { s[r] = __builtin_amdgcn_exp2f(s[r] - mn); ps += s[r]; }
            ps += swap32(ps);
            l = l * alpha + ps;
            m = mn;
            if (DROP && dc.thr) {
                // dropped probabilities leave the P.V product only: l stays the full softmax denominator, the 1/(1-p)
                // scale joins 1/l at the end. Registers 4g .. 4g+3 are four consecutive keys: the bytes of one random word.
#pragma unroll
                for (int r = 0; r < 16; r += 4) {
                    const uint32_t w = drop_word4(dc, drow + j0 + acc_row(r, h));
                    if (!drop_keep_byte<0>(dc, w)) s[r] = 0.f;
                    if (!drop_keep_byte<1>(dc, w)) s[r + 1] = 0.f;
                    if (!drop_keep_byte<2>(dc, w)) s[r + 2] = 0.f;
                    if (!drop_keep_byte<3>(dc, w)) s[r + 3] = 0.f;
                }
            }
#pragma unroll
            for (int b = 0; b < DB; ++b)
#pragma unroll
                for (int r = 0; r < 16; ++r) o[b][r] *= alpha;
#pragma unroll
            for (int ks = 0; ks < 2; ++ks) {
                const op16x8 pf = acc_frag(s, ks);
#pragma unroll
                for (int b = 0; b < DB; ++b) {
                    const op16x8 vf = tr_frag<D>(vimg, jt * 32 + 16 * ks, b, lane);
                    o[b] = mfma32_op(vf, pf, o[b]);
                }
            }
        }
    }
    if (!active) return;
    const float inv = dc.scale / l;
    // the wave's 32 x D output rows leave through a wave-private LDS staging area as 16-byte stores (four lanes cover
    // a row's 64-byte head slice); 8-byte-per-lane stores cost ~200 cycles each when every wave of the CU issues them
    char* stg = ostg + wave * (32 * D * 2);
#pragma unroll
    for (int b = 0; b < DB; ++b)
#pragma unroll
        for (int g = 0; g < 4; ++g) {
            u32x2 pk;
            pk[0] = pack_op2(o[b][4 * g] * inv, o[b][4 * g + 1] * inv);
            pk[1] = pack_op2(o[b][4 * g + 2] * inv, o[b][4 * g + 3] * inv);
            *(u32x2*)(stg + rr_off<D>(fr, (b * 64 + 16 * g) >> 4) + 8 * h) = pk;
        }
    constexpr int CPR = D / 8;
    op16* obase = a.out + ((size_t)seq * a.L + i0) * a.H + head * D;
#pragma unroll
    for (int kk = 0; kk < 32 * CPR / 64; ++kk) {
        const int idx = lane + 64 * kk, row = idx / CPR, c = idx % CPR;
        const u32x4 v = *(const u32x4*)(stg + rr_off<D>(row, c));
        st_stream((u32x4*)(obase + (size_t)row * a.H + c * 8), v);
    }
    if (h == 0 && a.lse_out) a.lse_out[((size_t)seq * a.A + head) * a.L + qi] = (m + __log2f(l)) * 0.6931471805599453f;
}

// ------------------------------------------------------------------ backward: dQ
template <int D, bool DROP = false>
__global__ __launch_bounds__(256, D == 32 ? 4 : 2) void attn_bwd_dq_kernel(AttnArgs a) {
    extern __shared__ __attribute__((aligned(16))) char smem[];
    constexpr int KS = D / 16, DB = D / 32, IMG = 128 * D * 2;
    char* kimg = smem;                  // K rows (S^T = K.Q^T)
    char* ktr = smem + IMG;             // K transposed reads (dQ^T = K^T.dS^T)
    char* vimg = smem + 2 * IMG;        // V rows (dP^T = V.dO^T)
    float* madd = (float*)(smem + 3 * IMG);
    float* relv = madd + ((a.L + 3) & ~3);      // this head's relative-position bias [2L]
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6, h = lane >> 5, fr = lane & 31;
    const int nqb = (a.L + 127) / 128;
    const int qb = blockIdx.x % nqb, head = (blockIdx.x / nqb) % a.A, seq = blockIdx.x / (nqb * a.A);
    const int ld = a.ld;
    const op16* base = a.qkv + qkv_base(a, seq, head, D);
    const int i0 = qb * 128 + wave * 32;
    const bool active = i0 < a.L;
    const int qi = i0 + fr;

    for (int t = tid; t < a.L; t += 256) madd[t] = a.mask[(size_t)seq * a.L + t] ? 0.f : kMaskMin;
    if (a.rel)
        for (int t = tid; t < 2 * a.L; t += 256) relv[t] = a.rel[(size_t)head * 2 * a.L + t];

    op16x8 qf[KS], dof[KS];
    float lse = 0.f, delta = 0.f;
    if (active) {
        const op16* orow = a.ctx + ((size_t)seq * a.L + qi) * a.H + head * D;
        const op16* drow = a.dctx + ((size_t)seq * a.L + qi) * a.H + head * D;
#pragma unroll
        for (int s = 0; s < KS; ++s) {
            qf[s] = *(const op16x8*)(base + (size_t)qi * ld + 16 * s + 8 * h);
            dof[s] = *(const op16x8*)(drow + 16 * s + 8 * h);
            const op16x8 of = *(const op16x8*)(orow + 16 * s + 8 * h);
#pragma unroll
            for (int e = 0; e < 8; ++e) delta += (float)dof[s][e] * (float)of[e];
        }
        delta += swap32(delta);
        lse = a.lse_in[((size_t)seq * a.A + head) * a.L + qi];
        if (h == 0) a.delta[((size_t)seq * a.A + head) * a.L + qi] = delta;     // reused by the dK/dV kernel
    }
    f32x16 dq[DB];
#pragma unroll
    for (int b = 0; b < DB; ++b)
#pragma unroll
        for (int r = 0; r < 16; ++r) dq[b][r] = 0.f;
    const DropCtx dc = DROP ? drop_ctx8(a.drop) : DropCtx{0u, 0u, 1.f};
    const uint32_t drow = ((uint32_t)(seq * a.A + head) * a.L + qi) * a.L;

    const int nchunk = (a.L + 127) / 128;
    for (int c = 0; c < nchunk; ++c) {
        const int rows = min(128, a.L - c * 128);
        __syncthreads();
        {
            Stager<D> sg;
            sg.load(0, base + (size_t)c * 128 * ld + a.woff, ld, rows, tid);
            sg.load(1, base + (size_t)c * 128 * ld + 2 * a.woff, ld, rows, tid);
            sg.template store<false>(0, kimg, tid);
            sg.template store<true>(0, ktr, tid);
            sg.template store<false>(1, vimg, tid);
        }
        __syncthreads();
        if (!active) continue;
        for (int jt = 0; jt < rows / 32; ++jt) {
            f32x16 s, dp;
#pragma unroll
            for (int r = 0; r < 16; ++r) { s[r] = 0.f; dp[r] = 0.f; }
#pragma unroll
            for (int ks = 0; ks < KS; ++ks) {
                const op16x8 kf = *(const op16x8*)(kimg + rr_off<D>(jt * 32 + fr, 2 * ks + h));
                const op16x8 vf = *(const op16x8*)(vimg + rr_off<D>(jt * 32 + fr, 2 * ks + h));
                s = mfma32_op(kf, qf[ks], s);
                dp = mfma32_op(vf, dof[ks], dp);
            }
            const int j0 = c * 128 + jt * 32;
            if (DROP && dc.thr) {
                // dP = mask * dP~ (dp holds the gradient of the DROPPED probabilities); delta = dO.O is unchanged
#pragma unroll
                for (int r = 0; r < 16; r += 4) {
                    const uint32_t w = drop_word4(dc, drow + j0 + acc_row(r, h));
                    dp[r] *= drop_keep_byte<0>(dc, w) ? dc.scale : 0.f;
                    dp[r + 1] *= drop_keep_byte<1>(dc, w) ? dc.scale : 0.f;
                    dp[r + 2] *= drop_keep_byte<2>(dc, w) ? dc.scale : 0.f;
                    dp[r + 3] *= drop_keep_byte<3>(dc, w) ? dc.scale : 0.f;
                }
            }
#pragma unroll
            for (int r = 0; r < 16; ++r) {
                const int j = j0 + acc_row(r, h);
                float v = s[r] * a.scale;
                if (a.rel) v += relv[j - qi + a.L];
                v += madd[j];
                const float p = __expf(v - lse);
                s[r] = p * (dp[r] - delta) * a.scale;          // dS^T * scale
            }
#pragma unroll
            for (int ks = 0; ks < 2; ++ks) {
                const op16x8 df = acc_frag(s, ks);
#pragma unroll
                for (int b = 0; b < DB; ++b) {
                    const op16x8 kt = tr_frag<D>(ktr, jt * 32 + 16 * ks, b, lane);
                    dq[b] = mfma32_op(kt, df, dq[b]);
                }
            }
        }
    }
    if (!active) return;
    op16* orow = a.dqkv + qkv_base(a, seq, head, D) + (size_t)qi * ld;
#pragma unroll
    for (int b = 0; b < DB; ++b)
#pragma unroll
        for (int g = 0; g < 4; ++g) {
            u32x2 pk;
            pk[0] = pack_op2(dq[b][4 * g], dq[b][4 * g + 1]);
            pk[1] = pack_op2(dq[b][4 * g + 2], dq[b][4 * g + 3]);
            *(u32x2*)(orow + b * 32 + 8 * g + 4 * h) = pk;
        }
}

// ------------------------------------------------------------------ backward: dK, dV
template <int D, bool REL, bool DROP = false>
__global__ __launch_bounds__(256, (D == 32 && !DROP) ? 4 : 2) void attn_bwd_dkv_kernel(AttnArgs a) {
    extern __shared__ __attribute__((aligned(16))) char smem[];
    constexpr int KS = D / 16, DB = D / 32, IMG = 128 * D * 2;
    char* qimg = smem;                  // Q rows   (S = Q.K^T)
    char* qtr = smem + IMG;             // Q^T      (dK^T = Q^T.dS)
    char* dimg = smem + 2 * IMG;        // dO rows  (dP = dO.V^T)
    char* dtr = smem + 3 * IMG;         // dO^T     (dV^T = dO^T.P)
    float* lse_s = (float*)(smem + 4 * IMG);   // [128]
    float* del_s = lse_s + 128;                // [128]
    float* relv = del_s + 128;                 // this head's relative-position bias [2L]
    float* drel_s = relv + 2 * a.L;            // its gradient: one private [2L] array per wave (no atomics)
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6, h = lane >> 5, fr = lane & 31;
    const int nkb = (a.L + 127) / 128;
    const int kb = blockIdx.x % nkb, head = (blockIdx.x / nkb) % a.A, seq = blockIdx.x / (nkb * a.A);
    const int ld = a.ld;
    const op16* base = a.qkv + qkv_base(a, seq, head, D);
    const op16* dbase = a.dctx + (size_t)seq * a.L * a.H + head * D;
    const int j0 = kb * 128 + wave * 32;
    const bool active = j0 < a.L;
    const int kj = j0 + fr;

    if (REL)
        for (int t = tid; t < 2 * a.L; t += 256) {
            relv[t] = a.rel[(size_t)head * 2 * a.L + t];
            drel_s[t] = drel_s[2 * a.L + t] = drel_s[4 * a.L + t] = drel_s[6 * a.L + t] = 0.f;
        }
    op16x8 kf[KS], vf[KS];
    float madd = 0.f;
    if (active) {
#pragma unroll
        for (int s = 0; s < KS; ++s) {
            kf[s] = *(const op16x8*)(base + (size_t)kj * ld + a.woff + 16 * s + 8 * h);
            vf[s] = *(const op16x8*)(base + (size_t)kj * ld + 2 * a.woff + 16 * s + 8 * h);
        }
        madd = a.mask[(size_t)seq * a.L + kj] ? 0.f : kMaskMin;
    }
    f32x16 dk[DB], dv[DB];
#pragma unroll
    for (int b = 0; b < DB; ++b)
#pragma unroll
        for (int r = 0; r < 16; ++r) { dk[b][r] = 0.f; dv[b][r] = 0.f; }
    const DropCtx dc = DROP ? drop_ctx8(a.drop) : DropCtx{0u, 0u, 1.f};
    const uint32_t dhead = (uint32_t)(seq * a.A + head) * a.L;            // mask row of query i: (dhead + i) * L
    // S orientation: the lane's key kj is byte kj & 3 (= lane & 3) of the word that the four lanes of its quad share, one
    // word per query row. Lane q of a quad hashes the rows 4g + q, the others fetch them by DPP: 4 hashes per 16 rows.
    const uint32_t dsh = 8u * (uint32_t)(lane & 3);

    const int nchunk = (a.L + 127) / 128;
    for (int c = 0; c < nchunk; ++c) {
        const int rows = min(128, a.L - c * 128);
        __syncthreads();
        Stager<D> sg;
        sg.load(0, base + (size_t)c * 128 * ld, ld, rows, tid);
        sg.load(1, dbase + (size_t)c * 128 * a.H, a.H, rows, tid);
        if (tid < rows) {      // per-query constants: log-sum-exp from forward, delta_i = dO_i . O_i from the dQ kernel
            const size_t o = ((size_t)seq * a.A + head) * a.L + c * 128 + tid;
            lse_s[tid] = a.lse_in[o];
            del_s[tid] = a.delta[o];
        }
        sg.template store<false>(0, qimg, tid);
        sg.template store<true>(0, qtr, tid);
        sg.template store<false>(1, dimg, tid);
        sg.template store<true>(1, dtr, tid);
        __syncthreads();
        if (!active) continue;
        for (int it = 0; it < rows / 32; ++it) {
            f32x16 s, dp;
#pragma unroll
            for (int r = 0; r < 16; ++r) { s[r] = 0.f; dp[r] = 0.f; }
#pragma unroll
            for (int ks = 0; ks < KS; ++ks) {
                const op16x8 qf = *(const op16x8*)(qimg + rr_off<D>(it * 32 + fr, 2 * ks + h));
                const op16x8 df = *(const op16x8*)(dimg + rr_off<D>(it * 32 + fr, 2 * ks + h));
                s = mfma32_op(qf, kf[ks], s);      // rows i, col j
                dp = mfma32_op(df, vf[ks], dp);
            }
            const int i0 = c * 128 + it * 32;
            f32x16 p;
#pragma unroll
            for (int g = 0; g < 4; ++g) {
                // accumulator registers 4g..4g+3 are query rows 8g + 4h + (0..3): one 16-byte LDS read per constant
                const int il = it * 32 + 8 * g + 4 * h;
                const f32x4 l4 = *(const f32x4*)(lse_s + il), d4 = *(const f32x4*)(del_s + il);
                uint32_t wq[4] = {0u, 0u, 0u, 0u};
                if (DROP && dc.thr) {
                    const uint32_t mine = drop_word4(dc, (dhead + (uint32_t)(i0 + 8 * g + 4 * h + (lane & 3))) * a.L + (uint32_t)(kj & ~3));
                    wq[0] = quad_bcast<0>(mine); wq[1] = quad_bcast<1>(mine); wq[2] = quad_bcast<2>(mine); wq[3] = quad_bcast<3>(mine);
                }
#pragma unroll
                for (int e = 0; e < 4; ++e) {
                    const int r = 4 * g + e;
                    const int i = i0 + 8 * g + 4 * h + e;
                    float v = s[r] * a.scale;
                    if (REL) v += relv[kj - i + a.L];
                    v += madd;
                    const float pr = __expf(v - l4[e]);
                    float mk = 1.f;
                    if (DROP && dc.thr) mk = (((wq[e] >> dsh) & 0xFFu) >= dc.thr) ? dc.scale : 0.f;
                    const float dsr = pr * (dp[r] * mk - d4[e]);               // dS (unscaled) = d(score)
                    p[r] = pr * mk;                                            // dV takes the dropped probabilities
                    s[r] = REL ? dsr : dsr * a.scale;                          // REL: unscaled until the bias gradient is taken
                }
            }
            if (REL && a.drel) {
                float lo = 0.f, hi = 0.f;
#pragma unroll
                for (int r = 0; r < 16; ++r) diag_add(s[r], r, lane, lo, hi);
                diag_store(drel_s + wave * 2 * a.L, a.L, j0, i0, lane, lo, hi);
            }
            if (REL) {
#pragma unroll
                for (int r = 0; r < 16; ++r) s[r] *= a.scale;
            }
#pragma unroll
            for (int ks = 0; ks < 2; ++ks) {
                const op16x8 pf = acc_frag(p, ks), sf = acc_frag(s, ks);
#pragma unroll
                for (int b = 0; b < DB; ++b) {
                    const op16x8 dt = tr_frag<D>(dtr, it * 32 + 16 * ks, b, lane);
                    const op16x8 qt = tr_frag<D>(qtr, it * 32 + 16 * ks, b, lane);
                    dv[b] = mfma32_op(dt, pf, dv[b]);
                    dk[b] = mfma32_op(qt, sf, dk[b]);
                }
            }
        }
    }
    if (active) {
        op16* krow = a.dqkv + qkv_base(a, seq, head, D) + (size_t)kj * ld + a.woff;
        op16* vrow = krow + a.woff;
#pragma unroll
        for (int b = 0; b < DB; ++b)
#pragma unroll
            for (int g = 0; g < 4; ++g) {
                u32x2 pk;
                pk[0] = pack_op2(dk[b][4 * g], dk[b][4 * g + 1]);
                pk[1] = pack_op2(dk[b][4 * g + 2], dk[b][4 * g + 3]);
                *(u32x2*)(krow + b * 32 + 8 * g + 4 * h) = pk;
                pk[0] = pack_op2(dv[b][4 * g], dv[b][4 * g + 1]);
                pk[1] = pack_op2(dv[b][4 * g + 2], dv[b][4 * g + 3]);
                *(u32x2*)(vrow + b * 32 + 8 * g + 4 * h) = pk;
            }
    }
    if (REL && a.drel) {
        __syncthreads();                                 // every wave's LDS adds are done (uniform branch)
        for (int t = tid; t < 2 * a.L; t += 256) {
            const float v = (drel_s[t] + drel_s[2 * a.L + t]) + (drel_s[4 * a.L + t] + drel_s[6 * a.L + t]);
            if (v != 0.f) atomicAdd(a.drel + (size_t)head * 2 * a.L + t, v);
        }
    }
}

// ------------------------------------------------------------------ backward, whole (sequence, head) in one workgroup
// L <= 128: the score matrix of one (sequence, head) is a single 128 x 128 tile, so dQ, dK and dV come out of ONE
// evaluation of S, P, dP and dS (the two-kernel path above evaluates them twice and reads q/k/v/dO twice).
// Orientation as in the dK/dV kernel (S = Q.K^T: rows = queries in registers, column = key on the lane; wave w owns
// keys 32w..32w+31 and accumulates dK^T, dV^T over the query tiles). dQ needs the contraction over keys, i.e. dS^T as
// an operand: every wave drops its dS tiles (bf16, already scaled) into a [key][query] image, and after one barrier
// wave w computes dQ^T = K^T.dS^T for ITS query tile from that image with transposing LDS reads. delta_i = dO_i.O_i is
// computed while dO is being staged.
constexpr int DS_IMG = 128 * 128 * 2;
typedef __attribute__((ext_vector_type(2))) float f32x2;
__device__ __forceinline__ uint32_t ds_off(int row, int byte) { return (uint32_t)(row * 256 + (byte ^ ((row & 3) << 6))); }

// tr_frag on a ROW-READ image of 64-byte rows (rr_off<32>: 16-byte chunk c of row r sits at c ^ ((r >> 2) & 3)). The four
// rows a half-wave reads (row0 + 4h + q, q = 0..3) share r >> 2, so each is still read whole, 256 contiguous bytes per
// half-wave: no bank conflict, and the separate transposed copies of Q and dO (two more 16-byte LDS stores per thread and
// tensor) are not needed.
__device__ __forceinline__ op16x8 tr_frag_rr32(const char* img, int row0, int lane) {
    const int li = lane & 15, q = li >> 2, p = li & 3, gsel = (lane >> 4) & 1, h = lane >> 5;
    const int ra = row0 + 4 * h + q, rb = ra + 8;
    const int chunk = 2 * gsel + (p >> 1), rem = 8 * (p & 1);
    const op16x4 a = lds_tr(img + ra * 64 + ((chunk ^ ((ra >> 2) & 3)) << 4) + rem);
    const op16x4 b = lds_tr(img + rb * 64 + ((chunk ^ ((rb >> 2) & 3)) << 4) + rem);
    op16x8 f;
#pragma unroll
    for (int e = 0; e < 4; ++e) { f[e] = a[e]; f[4 + e] = b[e]; }
    return f;
}

// A-operand fragment that carries one fp32 value per row into a 32 x 32 accumulator tile: row (lane & 31) holds
// x = hi + lo (two bf16) at k = 0, 1; every other k is zero. Multiplied by a B fragment of ones at k = 0, 1.
__device__ __forceinline__ op16x8 row_frag(float x, int h) {
    const float hi = op_lo(pack_op2(x, 0.f));
    u32x4 u = {h == 0 ? pack_op2(x, x - hi) : 0u, 0u, 0u, 0u};
    return __builtin_bit_cast(op16x8, u);
}

template <int D, bool REL, bool DROP = false>
__global__ __launch_bounds__(256, 2) void attn_bwd_fused_kernel(AttnArgs a) {
    extern __shared__ __attribute__((aligned(16))) char smem[];
    constexpr int KS = D / 16, DB = D / 32, IMG = 128 * D * 2, CPR = D / 8, PER = 128 * CPR / 256;
    static_assert(D == 32, "one image per operand: the transposing reads below assume 64-byte rows");
    char* qimg = smem;                  // Q rows   (S = Q.K^T) and, read transposed, Q^T (dK^T = Q^T.dS)
    char* dimg = smem + 2 * IMG;        // dO rows  (dP = dO.V^T) and dO^T (dV^T = dO^T.P)
                                        // (slots 1 and 3 are only the output staging's)
    char* ktr = smem + 4 * IMG;         // K^T      (dQ^T = K^T.dS^T)
    char* dsimg = smem + 5 * IMG;       // dS [key][query] bf16
    float* lse_s = (float*)(smem + 5 * IMG + DS_IMG);   // [128]
    float* del_s = lse_s + 128;                          // [128]
    float* relv = del_s + 128;                           // this head's relative-position bias [2L], log2 units
    float* drel_s = relv + 2 * a.L;                      // its gradient: one private [2L] array per wave
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6, h = lane >> 5, fr = lane & 31;
    const int ld = a.ld, rows = a.L;
    const int j0 = wave * 32;
    const bool active = j0 < rows;
    const int kj = j0 + fr;
    const int nitems = a.nseq * a.A;
    const float sc2 = a.scale * kLog2e, inv_scale = 1.0f / a.scale;
    const f32x16 zero16 = {0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f};
    u32x4 ones_u = {h == 0 ? pack_op2(1.f, 1.f) : 0u, 0u, 0u, 0u};  // B fragment: 1.0 at k = 0, 1 (held by the lanes of half 0)
    const op16x8 ones = __builtin_bit_cast(op16x8, ones_u);

    // Persistent over (sequence, head) items: the global loads of item n+1 are in flight (in registers) while item n
    // is computed -- a workgroup per item spent most of its life waiting for its own loads at two workgroups per CU.
    u32x4 pq[PER], pd[PER], po[PER], pk[PER], pv[PER];
    int64_t nmask = 1;                                   // raw mask word of this lane's key: no use before the next item
    float nlse = 0.f;
    // part 0..3: a quarter of the next item's loads each. Issued between the score tiles so that the CU's address
    // unit sees them spread over the compute phase (issued as one burst, with all eight waves of the CU doing the same,
    // the ten loads took 3700 cycles to get through it while nothing else ran); part < 0 issues everything.
    // Buffer loads: one descriptor per tensor, a per-lane byte offset that never changes (rows past the sequence: an offset
    // that fails the range check, the lane reads zeros -- no branch around the load) and the item's position as the scalar
    // offset. A quarter is then 2-4 instructions with no address arithmetic; as flat loads each one carried two 64-bit adds
    // per lane and an exec-mask branch.
    constexpr uint32_t kOob = 0x7FFFFFF0u;
    const __amdgpu_buffer_rsrc_t rq = make_rsrc(a.qkv, (uint32_t)((size_t)a.nseq * a.L * ld * 2));
    const __amdgpu_buffer_rsrc_t rd = make_rsrc(a.dctx, (uint32_t)((size_t)a.nseq * a.L * a.H * 2));
    const __amdgpu_buffer_rsrc_t ro = make_rsrc(a.ctx, (uint32_t)((size_t)a.nseq * a.L * a.H * 2));
    const __amdgpu_buffer_rsrc_t rw = make_rsrc(a.dqkv, (uint32_t)((size_t)a.nseq * a.L * ld * 2));
    uint32_t vq[PER], vc[PER];
#pragma unroll
    for (int k = 0; k < PER; ++k) {
        const int idx = tid + 256 * k, row = idx / CPR, c = idx % CPR;
        vq[k] = row < rows ? (uint32_t)(row * ld + c * 8) * 2u : kOob;
        vc[k] = row < rows ? (uint32_t)(row * a.H + c * 8) * 2u : kOob;
    }
    auto prefetch = [&](int item, int part) {
        const int head = item % a.A, seq = item / a.A;
        const uint32_t sq = (uint32_t)(qkv_base(a, seq, head, D) * 2);                       // scalar byte offsets
        const uint32_t sc = (uint32_t)(((size_t)seq * a.L * a.H + head * D) * 2);
        const uint32_t wb = (uint32_t)(a.woff * 2);
        if (part < 0 || part == 0) {
#pragma unroll
            for (int k = 0; k < PER; ++k) pq[k] = __builtin_amdgcn_raw_buffer_load_b128(rq, (int)vq[k], (int)sq, 0);
            nmask = a.mask[(size_t)seq * a.L + (active ? kj : 0)];      // (unconditional: a branch here makes the value
                                                                        //  a phi that is waited for at the end of the tiles)
        }
        if (part < 0 || part == 1) {
#pragma unroll
            for (int k = 0; k < PER; ++k) pk[k] = __builtin_amdgcn_raw_buffer_load_b128(rq, (int)vq[k], (int)(sq + wb), 0);
            nlse = a.lse_in[((size_t)seq * a.A + head) * a.L + (tid < rows ? tid : 0)];
        }
        if (part < 0 || part == 2) {
#pragma unroll
            for (int k = 0; k < PER; ++k) pv[k] = __builtin_amdgcn_raw_buffer_load_b128(rq, (int)vq[k], (int)(sq + 2 * wb), 0);
        }
        if (part < 0 || part == 3) {
#pragma unroll
            for (int k = 0; k < PER; ++k) {
                pd[k] = __builtin_amdgcn_raw_buffer_load_b128(rd, (int)vc[k], (int)sc, 0);
                po[k] = __builtin_amdgcn_raw_buffer_load_b128(ro, (int)vc[k], (int)sc, 0);
            }
        }
    };

#ifdef QST_STAMP_ATTN
    // diagnostic build: s_memtime at the phase boundaries of the third item (8 ..: inside its second tile), wave 0 -> a.delta
    // as uint64 [block][16]
#define QST_STAMP(k) do { __builtin_amdgcn_sched_barrier(0); if (iter == 2 && tid == 0) ((unsigned long long*)a.delta)[blockIdx.x * 16 + (k)] = __builtin_readcyclecounter(); __builtin_amdgcn_sched_barrier(0); } while (0)
#define QST_STAMP_ONCE(k) do { if (tid == 0) ((unsigned long long*)a.delta)[blockIdx.x * 16 + (k)] = __builtin_readcyclecounter(); } while (0)
#else
#define QST_STAMP(k) do {} while (0)
#define QST_STAMP_ONCE(k) do {} while (0)
#endif
    // Item order: d = 32 head slices are 64 bytes, so heads 2k and 2k+1 of a token share every 128-byte line.
    // Workgroups b and b + grid/2 sit on the same XCD (grid/2 is a multiple of 8) and walk the two heads of the same
    // pair in step, so each line leaves HBM once and the partner finds it in that XCD's L2.
    const int half = gridDim.x >> 1;
    const bool paired = (a.A % 2 == 0) && (gridDim.x % 16 == 0);
    const int bsub = paired ? ((int)blockIdx.x >= half) : 0;
    const int bp = paired ? (int)blockIdx.x - bsub * half : (int)blockIdx.x;
    const int stride = paired ? half : (int)gridDim.x;
    const int nunits = paired ? nitems / 2 : nitems;
    auto item_of = [&](int unit) { return paired ? 2 * unit + bsub : unit; };
    int unit = bp, iter = 0;
    const DropCtx dc = DROP ? drop_ctx8(a.drop) : DropCtx{0u, 0u, 1.f};
    const uint32_t dsh = 8u * (uint32_t)(lane & 3);       // this lane's byte of its quad's random words (see the dK/dV kernel)
    const uint32_t dlane = (uint32_t)(4 * h + (lane & 3)) * a.L + (uint32_t)(kj & ~3);   // (see dbase below)
    QST_STAMP_ONCE(13);
    if (unit < nunits) prefetch(item_of(unit), -1);
    for (; unit < nunits; unit += stride, ++iter) {
        const int item = item_of(unit);
        const int head = item % a.A, seq = item / a.A;
        // mask element (query i, key j) of this item: dbase + i * L + j. Split into a per-item scalar, a per-tile scalar and
        // a per-lane constant: a per-element multiply compiles to v_mad_u64_u32, whose undefined upper addend register
        // aliased a pending prefetch destination and made every tile wait for vmcnt(0).
        const uint32_t dbase = (uint32_t)(seq * a.A + head) * a.L * a.L;
        __syncthreads();                                 // the previous item's readers of the images are done
        QST_STAMP(0);
        const float madd2 = nmask ? 0.f : -INFINITY;     // kMaskMin in log2 units
        // -lse / scale enters the score accumulator (see the tile loop): q.k - lse / scale. A sequence of padding only has
        // lse ~ -2e38 (every key carries the finite mask constant of the forward): clamped, so that the value stays finite
        // through the bf16 hi/lo split; with every key masked the probabilities are exp2(-inf) = 0 whatever it is.
        // (f16 operands: the clamp sits inside half's range; real rows have |lse / scale| of a few hundred)
        if (tid < rows) lse_s[tid] = fminf(-nlse * inv_scale, QST_OP_F16 ? 6.0e4f : 1.0e30f);
        if (REL)
            for (int t = tid; t < 2 * a.L; t += 256) {
                relv[t] = kLog2e * a.rel[(size_t)head * 2 * a.L + t];
                drel_s[t] = drel_s[2 * a.L + t] = drel_s[4 * a.L + t] = drel_s[6 * a.L + t] = 0.f;
            }
        // delta_i = sum_dd dO[i][dd] * O[i][dd]: 8 elements per thread, CPR adjacent threads per row
#pragma unroll
        for (int k = 0; k < PER; ++k) {
            float part = 0.f;
#pragma unroll
            for (int e = 0; e < 4; ++e)
                part += op_lo(pd[k][e]) * op_lo(po[k][e]) + op_hi(pd[k][e]) * op_hi(po[k][e]);
#pragma unroll
            for (int o = 1; o < CPR; o <<= 1) part += __shfl_xor(part, o);
            const int idx = tid + 256 * k, row = idx / CPR, c = idx % CPR;
            if (c == 0) del_s[row] = -part;
            *(u32x4*)(qimg + rr_off<D>(row, c)) = pq[k];             // one image each serves the row reads AND the
            *(u32x4*)(dimg + rr_off<D>(row, c)) = pd[k];             // transposing reads (tr_frag_rr32)
            *(u32x4*)(ktr + tr_off<D>(row, c * 16)) = pk[k];
            *(u32x4*)(dsimg + ds_off(row, c * 16)) = pv[k];          // V rows borrow the head of the dS image rows
        }
        QST_STAMP(7);
        const bool more = unit + stride < nunits;
        const int next_item = more ? item_of(unit + stride) : 0;
        QST_STAMP(1);
        __syncthreads();
        QST_STAMP(2);

        f32x16 dk[DB], dv[DB];
#pragma unroll
        for (int b = 0; b < DB; ++b)
#pragma unroll
            for (int r = 0; r < 16; ++r) { dk[b][r] = 0.f; dv[b][r] = 0.f; }
        {
            // this wave's K / V operand fragments (lane = key row). Image rows of a wave's own keys are touched by no
            // other wave before the next barrier, so the V rows parked in the dS image are read before this wave's
            // first dS tile overwrites them (program order).
            op16x8 kf[KS], vf[KS];
#pragma unroll
            for (int s = 0; s < KS; ++s) {
                kf[s] = *(const op16x8*)(ktr + tr_off<D>(active ? kj : 0, (16 * s + 8 * h) * 2));
                vf[s] = *(const op16x8*)(dsimg + ds_off(active ? kj : 0, (16 * s + 8 * h) * 2));
            }
            // Unrolled over the (at most four) query tiles, prefetch quarter `it` issued by EVERY wave at one place in the
            // code: `it` is a constant in each copy and each prefetch register has a single definition, so the loads go
            // straight into their registers and nothing waits for them before the next item's staging. As a loop (and
            // with a second copy of the prefetch for waves without keys) the compiler routed the loads through temporaries
            // and phi copies behind vmcnt(0): three of the four quarters exposed a full HBM round trip per item -- 2,700
            // cycles per tile for 256 cycles of MFMA (in-kernel stamps).
            const int nit = active ? rows / 32 : 0;
#pragma unroll
            for (int it = 0; it < 4; ++it) {
                if (it == 1) QST_STAMP(8);
                if (more) prefetch(next_item, it);
                if (it == 1) QST_STAMP(9);
                if (it >= nit) continue;
                // The per-query constants enter through the ACCUMULATORS: -lse_i / scale and -delta_i are split into
                // bf16 hi + lo (16 mantissa bits; the products with 1.0 are exact in the fp32 accumulator) and a
                // one-k-step MFMA against a fragment of ones lays them out as [query row][any key] -- the layout the score
                // tile has. Reading them as per-row scalars cost 8 ds_read_b128 per tile (half-wave broadcasts, 45% of the
                // kernel's LDS traffic, and the LDS pipe shared by the CU's 8 waves is what bounds this kernel) plus 8 packed
                // adds; this way it is 2 ds_read_b32 and 2 of the idle matrix core's cycles.
                const op16x8 la = row_frag(lse_s[it * 32 + fr], h), da = row_frag(del_s[it * 32 + fr], h);
                f32x16 s = mfma32_op(la, ones, zero16);
                f32x16 nd = mfma32_op(da, ones, zero16);
                f32x16 dp = DROP ? zero16 : nd;                     // dropout: mask * dP~ - delta needs -delta on its own
#pragma unroll
                for (int ks = 0; ks < KS; ++ks) {
                    const op16x8 qf = *(const op16x8*)(qimg + rr_off<D>(it * 32 + fr, 2 * ks + h));
                    const op16x8 df = *(const op16x8*)(dimg + rr_off<D>(it * 32 + fr, 2 * ks + h));
                    s = mfma32_op(qf, kf[ks], s);      // rows i, col j
                    dp = mfma32_op(df, vf[ks], dp);
                }
                if (it == 1) QST_STAMP(10);
                // Elementwise part, trimmed because it and the LDS pipe (not the MFMAs) bound this kernel: s already holds
                // q.k - lse / scale, so one packed fma (softmax scale * log2(e), key mask) feeds v_exp_f32 directly; dS
                // stays unscaled (dK and dQ are scaled once at the end); P and dS are rounded to bf16 once and the packed
                // words serve both as MFMA fragments and as the dS image rows.
                uint32_t pw[4][2], sw[4][2];
                float dlo = 0.f, dhi = 0.f;
#pragma unroll
                for (int g = 0; g < 4; ++g) {
                    const int il = it * 32 + 8 * g + 4 * h;     // accumulator registers 4g..4g+3 = query rows il..il+3
                    uint32_t wq[4] = {0u, 0u, 0u, 0u};
                    if (DROP) {
                        const uint32_t mine = drop_word4(dc, dbase + (uint32_t)(it * 32 + 8 * g) * a.L + dlane);
                        wq[0] = quad_bcast<0>(mine); wq[1] = quad_bcast<1>(mine); wq[2] = quad_bcast<2>(mine); wq[3] = quad_bcast<3>(mine);
                    }
#pragma unroll
                    for (int e = 0; e < 4; e += 2) {
                        const int r = 4 * g + e;
                        f32x2 sv, dpv;
                        sv[0] = s[r]; sv[1] = s[r + 1];
                        dpv[0] = dp[r]; dpv[1] = dp[r + 1];
                        f32x2 v = sv * sc2 + madd2;                               // log2 of the probability
                        if (REL) {
                            v[0] += relv[kj - (il + e) + a.L];
                            v[1] += relv[kj - (il + e + 1) + a.L];
                        }
                        f32x2 pr;
                        pr[0] = __builtin_amdgcn_exp2f(v[0]);
                        pr[1] = __builtin_amdgcn_exp2f(v[1]);
                        if (DROP) {
                            // dP = mask * dP~ and dV takes the dropped probabilities
                            f32x2 mk, ndv;
                            mk[0] = (((wq[e] >> dsh) & 0xFFu) >= dc.thr) ? dc.scale : 0.f;
                            mk[1] = (((wq[e + 1] >> dsh) & 0xFFu) >= dc.thr) ? dc.scale : 0.f;
                            ndv[0] = nd[r]; ndv[1] = nd[r + 1];
                            dpv = dpv * mk + ndv;
                            pw[g][e >> 1] = pack_op2(pr[0] * mk[0], pr[1] * mk[1]);
                        } else {
                            pw[g][e >> 1] = pack_op2(pr[0], pr[1]);
                        }
                        const f32x2 dsr = pr * dpv;                               // dS (unscaled) = d(score)
                        if (REL && a.drel) { diag_add(dsr[0], r, lane, dlo, dhi); diag_add(dsr[1], r + 1, lane, dlo, dhi); }
                        sw[g][e >> 1] = pack_op2(dsr[0], dsr[1]);
                    }
                    u32x2 pkd;
                    pkd[0] = sw[g][0]; pkd[1] = sw[g][1];
                    *(u32x2*)(dsimg + ds_off(kj, il * 2)) = pkd;
                }
                if (REL && a.drel) diag_store(drel_s + wave * 2 * a.L, a.L, j0, it * 32, lane, dlo, dhi);
                if (it == 1) QST_STAMP(11);
#pragma unroll
                for (int ks = 0; ks < 2; ++ks) {
                    u32x4 pu, su;
                    pu[0] = pw[2 * ks][0]; pu[1] = pw[2 * ks][1]; pu[2] = pw[2 * ks + 1][0]; pu[3] = pw[2 * ks + 1][1];
                    su[0] = sw[2 * ks][0]; su[1] = sw[2 * ks][1]; su[2] = sw[2 * ks + 1][0]; su[3] = sw[2 * ks + 1][1];
                    const op16x8 pf = __builtin_bit_cast(op16x8, pu), sf = __builtin_bit_cast(op16x8, su);
#pragma unroll
                    for (int b = 0; b < DB; ++b) {
                        const op16x8 dt = tr_frag_rr32(dimg, it * 32 + 16 * ks, lane);
                        const op16x8 qt = tr_frag_rr32(qimg, it * 32 + 16 * ks, lane);
                        dv[b] = mfma32_op(dt, pf, dv[b]);
                        dk[b] = mfma32_op(qt, sf, dk[b]);
                    }
                }
                if (it == 1) QST_STAMP(12);
            }
#pragma unroll
            for (int b = 0; b < DB; ++b)
#pragma unroll
                for (int r = 0; r < 16; ++r) dk[b][r] *= a.scale;
        }
        QST_STAMP(3);
        __syncthreads();                                 // every wave's dS tiles are in the image
        QST_STAMP(4);
        if (REL && a.drel)
            for (int t = tid; t < 2 * a.L; t += 256) {
                const float v = (drel_s[t] + drel_s[2 * a.L + t]) + (drel_s[4 * a.L + t] + drel_s[6 * a.L + t]);
                if (v != 0.f) atomicAdd(a.drel + (size_t)head * 2 * a.L + t, v);
            }
        if (active) {
            // dQ^T[dd][i] for this wave's query tile i = 32*wave + (lane & 31): contraction over all keys
            f32x16 dq[DB];
#pragma unroll
            for (int b = 0; b < DB; ++b)
#pragma unroll
                for (int r = 0; r < 16; ++r) dq[b][r] = 0.f;
            const int li = lane & 15, q = li >> 2, pp = li & 3, gsel = (lane >> 4) & 1;
            const int byte = wave * 64 + gsel * 32 + 8 * pp;
            for (int ks = 0; ks < rows / 16; ++ks) {
                const op16x4 lo = lds_tr(dsimg + ds_off(16 * ks + 4 * h + q, byte));
                const op16x4 hi = lds_tr(dsimg + ds_off(16 * ks + 8 + 4 * h + q, byte));
                op16x8 df;
#pragma unroll
                for (int e = 0; e < 4; ++e) { df[e] = lo[e]; df[4 + e] = hi[e]; }
#pragma unroll
                for (int b = 0; b < DB; ++b) {
                    const op16x8 kt = tr_frag<D>(ktr, 16 * ks, b, lane);
                    dq[b] = mfma32_op(kt, df, dq[b]);
                }
            }
            QST_STAMP(5);
            // Outputs leave through LDS so that every global store is 16 bytes and four lanes cover a row's whole
            // 64-byte head slice (stamps showed ~200 cycles per 8-byte-per-lane store instruction: 12 of them cost as
            // much as the dQ product). The Q / dO images are dead after the barrier above; wave w stages in its 8 KB.
            char* stg = smem + wave * (4 * IMG / 4);
#pragma unroll
            for (int b = 0; b < DB; ++b)
#pragma unroll
                for (int g = 0; g < 4; ++g) {
                    u32x2 o2;
                    const uint32_t off = rr_off<D>(fr, (b * 64 + 16 * g) >> 4) + 8 * h;
                    o2[0] = pack_op2(dq[b][4 * g] * a.scale, dq[b][4 * g + 1] * a.scale);
                    o2[1] = pack_op2(dq[b][4 * g + 2] * a.scale, dq[b][4 * g + 3] * a.scale);
                    *(u32x2*)(stg + off) = o2;
                    o2[0] = pack_op2(dk[b][4 * g], dk[b][4 * g + 1]);
                    o2[1] = pack_op2(dk[b][4 * g + 2], dk[b][4 * g + 3]);
                    *(u32x2*)(stg + 32 * D * 2 + off) = o2;
                    o2[0] = pack_op2(dv[b][4 * g], dv[b][4 * g + 1]);
                    o2[1] = pack_op2(dv[b][4 * g + 2], dv[b][4 * g + 3]);
                    *(u32x2*)(stg + 2 * 32 * D * 2 + off) = o2;
                }
            // wave-private staging: no workgroup barrier, the LDS queue is in order within a wave
            const uint32_t so = (uint32_t)((qkv_base(a, seq, head, D) + (size_t)j0 * ld) * 2);
#pragma unroll
            for (int t = 0; t < 3; ++t)
#pragma unroll
                for (int k = 0; k < 32 * CPR / 64; ++k) {
                    const int idx = lane + 64 * k, row = idx / CPR, c = idx % CPR;
                    const u32x4 v = *(const u32x4*)(stg + t * 32 * D * 2 + rr_off<D>(row, c));
                    __builtin_amdgcn_raw_buffer_store_b128(v, rw, (int)((uint32_t)(row * ld + c * 8) * 2u),
                                                           (int)(so + (uint32_t)t * (uint32_t)(a.woff * 2)), QST_STREAM_STORES ? 2 : 0);   // aux bit 1 = nt
                }
            QST_STAMP(6);
        }
    }
    QST_STAMP_ONCE(14);
}


// tr_frag on the ROW-READ image of 128-byte rows (rr_off<64>): correct for any swizzle that keeps 16-byte chunks whole;
// the four rows a half-wave reads collide pairwise in the banks (2-way), the price of one image instead of two
__device__ __forceinline__ op16x8 tr_frag_rr64(const char* img, int row0, int ddb, int lane) {
    const int li = lane & 15, q = li >> 2, p = li & 3, gsel = (lane >> 4) & 1, h = lane >> 5;
    const int chunk = ddb * 4 + gsel * 2 + (p >> 1), rem = 8 * (p & 1);
    const int ra = row0 + 4 * h + q, rb = ra + 8;
    const op16x4 a = lds_tr(img + rr_off<64>(ra, chunk) + rem);
    const op16x4 b = lds_tr(img + rr_off<64>(rb, chunk) + rem);
    op16x8 f;
#pragma unroll
    for (int e = 0; e < 4; ++e) { f[e] = a[e]; f[4 + e] = b[e]; }
    return f;
}

// ------------------------------------------------------------------ backward, d = 64, L <= 512: one workgroup per (sequence, head)
// The two-kernel path above evaluates S, P, dP and dS twice (7 contractions, two passes over q / k / v / dO, two exp per
// score). Here a 512-thread workgroup (8 waves) owns one (sequence, head): keys are walked in passes of 256 (wave w owns
// the 32 keys 256 p + 32 w .. and accumulates dK^T, dV^T for them, orientation as in the dK/dV kernel: S = Q.K^T, rows =
// queries in registers, column = key on the lane), queries in chunks of 128. Per (pass, chunk): every wave with keys
// evaluates its four 32 x 32 score tiles once, adds into dK / dV and drops dS (bf16, unscaled) into a [key][query] LDS
// image; after one barrier wave w = 2 qt + b multiplies K^T (transposing reads of the pass's K image) with that image
// for query tile qt, d-half b: dQ^T tile += K^T.dS^T over the pass's keys. The dQ tiles of ALL chunks stay in registers
// across the passes (NC x 16 registers per wave), so every output is written once, in bf16, with no atomics.
// delta_i = dO_i.O_i is computed while the first pass stages dO. One image per operand serves row reads and
// transposing reads (the row-read swizzle makes the transposing reads 2-way conflicted; two more images would not fit).
template <int NC, bool REL, bool DROP>
__global__ __launch_bounds__(512, 1) void attn_bwd_one64_kernel(AttnArgs a) {
    extern __shared__ __attribute__((aligned(16))) char smem[];
    constexpr int D = 64, KS = 4, DB = 2, CPR = 8;
    char* qimg = smem;                        // [128 q][128 B]
    char* dimg = smem + 16384;                // dO chunk
    char* kimg = smem + 32768;                // [256 keys][128 B] of this pass
    char* dsimg = smem + 65536;               // dS [256 keys][128 queries] bf16, ds_off
    float* lse_s = (float*)(smem + 131072);   // [L]
    float* del_s = lse_s + a.L;               // [L]
    float* relv = del_s + a.L;                // [2L]   (REL)
    float* drel_s = relv + 2 * a.L;           // [8][2L] (REL)
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6, h = lane >> 5, fr = lane & 31;
    const int head = blockIdx.x % a.A, seq = blockIdx.x / a.A;
    const int ld = a.ld, L = a.L;
    const op16* base = a.qkv + qkv_base(a, seq, head, D);
    const op16* dbase_p = a.dctx + (size_t)seq * L * a.H + head * D;
    const op16* obase_p = a.ctx + (size_t)seq * L * a.H + head * D;
    const int nchunk = (L + 127) / 128, npass = (L + 255) / 256;

    for (int t = tid; t < L; t += 512) lse_s[t] = a.lse_in[((size_t)seq * a.A + head) * L + t];
    if (REL)
        for (int t = tid; t < 2 * L; t += 512) {
            relv[t] = a.rel[(size_t)head * 2 * L + t];
#pragma unroll
            for (int w = 0; w < 8; ++w) drel_s[w * 2 * L + t] = 0.f;
        }
    const DropCtx dc = DROP ? drop_ctx8(a.drop) : DropCtx{0u, 0u, 1.f};
    const uint32_t dhead = (uint32_t)(seq * a.A + head) * L;            // mask row of query i: (dhead + i) * L
    const uint32_t dsh = 8u * (uint32_t)(lane & 3);
    const int qt_mine = wave >> 1, b_mine = wave & 1;                    // this wave's dQ^T tile of every chunk

    f32x16 dq[NC];
#pragma unroll
    for (int c = 0; c < NC; ++c)
#pragma unroll
        for (int r = 0; r < 16; ++r) dq[c][r] = 0.f;

    for (int p = 0; p < npass; ++p) {
        const int nkeys = min(256, L - 256 * p);
        const int j0 = 256 * p + 32 * wave;
        const bool active = j0 < L;
        const int kj = j0 + fr;
        __syncthreads();                                     // the previous pass's readers of kimg are done
#pragma unroll
        for (int k = 0; k < 4; ++k) {                        // K rows of this pass: 256 x 8 chunks of 16 bytes
            const int idx = tid + 512 * k, row = idx / CPR, c = idx % CPR;
            const u32x4 z = {0, 0, 0, 0};
            const u32x4 v = row < nkeys ? *(const u32x4*)(base + (size_t)(256 * p + row) * ld + a.woff + c * 8) : z;
            *(u32x4*)(kimg + rr_off<D>(row, c)) = v;
        }
        op16x8 vf[KS];
        float madd = 0.f;
        if (active) {
#pragma unroll
            for (int s = 0; s < KS; ++s) vf[s] = *(const op16x8*)(base + (size_t)kj * ld + 2 * a.woff + 16 * s + 8 * h);
            madd = a.mask[(size_t)seq * L + kj] ? 0.f : kMaskMin;
        }
        f32x16 dk[DB], dv[DB];
#pragma unroll
        for (int b = 0; b < DB; ++b)
#pragma unroll
            for (int r = 0; r < 16; ++r) { dk[b][r] = 0.f; dv[b][r] = 0.f; }

#pragma unroll 1
        for (int c = 0; c < nchunk; ++c) {
            const int rows = min(128, L - c * 128);
            __syncthreads();                                 // the previous chunk's readers of qimg / dimg / dsimg are done
            {
                u32x4 vq[2], vd[2], vo[2];
                const u32x4 z = {0, 0, 0, 0};
#pragma unroll
                for (int k = 0; k < 2; ++k) {
                    const int idx = tid + 512 * k, row = idx / CPR, cc = idx % CPR;
                    const bool ok = row < rows;
                    vq[k] = ok ? *(const u32x4*)(base + (size_t)(c * 128 + row) * ld + cc * 8) : z;
                    vd[k] = ok ? *(const u32x4*)(dbase_p + (size_t)(c * 128 + row) * a.H + cc * 8) : z;
                    if (p == 0) vo[k] = ok ? *(const u32x4*)(obase_p + (size_t)(c * 128 + row) * a.H + cc * 8) : z;
                }
#pragma unroll
                for (int k = 0; k < 2; ++k) {
                    const int idx = tid + 512 * k, row = idx / CPR, cc = idx % CPR;
                    *(u32x4*)(qimg + rr_off<D>(row, cc)) = vq[k];
                    *(u32x4*)(dimg + rr_off<D>(row, cc)) = vd[k];
                    if (p == 0) {                            // delta_i = sum_dd dO[i][dd] O[i][dd]: 8 adjacent lanes per row
                        float part = 0.f;
#pragma unroll
                        for (int e = 0; e < 4; ++e)
                            part += op_lo(vd[k][e]) * op_lo(vo[k][e]) + op_hi(vd[k][e]) * op_hi(vo[k][e]);
                        part += dpp_mov<0xB1>(part);         // quad_perm [1,0,3,2]
                        part += dpp_mov<0x4E>(part);         // quad_perm [2,3,0,1]
                        part += dpp_mov<0x141>(part);        // row_half_mirror: the other quad of the 8 lanes
                        if (cc == 0 && row < rows) del_s[c * 128 + row] = part;
                    }
                }
            }
            __syncthreads();
            if (active) {
#pragma unroll 1
                for (int it = 0; it < rows / 32; ++it) {
                    f32x16 s, dp;
#pragma unroll
                    for (int r = 0; r < 16; ++r) { s[r] = 0.f; dp[r] = 0.f; }
#pragma unroll
                    for (int ks = 0; ks < KS; ++ks) {
                        // (this wave's K fragments are re-read from the pass image per tile: 16 registers fewer to keep alive
                        // through the elementwise part, where the kernel is at the register limit)
                        const op16x8 kf = *(const op16x8*)(kimg + rr_off<D>(32 * wave + fr, 2 * ks + h));
                        const op16x8 qf = *(const op16x8*)(qimg + rr_off<D>(it * 32 + fr, 2 * ks + h));
                        const op16x8 df = *(const op16x8*)(dimg + rr_off<D>(it * 32 + fr, 2 * ks + h));
                        s = mfma32_op(qf, kf, s);          // rows i, col j
                        dp = mfma32_op(df, vf[ks], dp);
                    }
                    const int i0 = c * 128 + it * 32;
#pragma unroll
                    for (int g = 0; g < 4; ++g) {
                        const int il = i0 + 8 * g + 4 * h;              // accumulator registers 4g..4g+3 = query rows il..il+3
                        const f32x4 l4 = *(const f32x4*)(lse_s + il), d4 = *(const f32x4*)(del_s + il);
                        uint32_t wq[4] = {0u, 0u, 0u, 0u};
                        if (DROP && dc.thr) {
                            const uint32_t mine = drop_word4(dc, (dhead + (uint32_t)(il + (lane & 3))) * L + (uint32_t)(kj & ~3));
                            wq[0] = quad_bcast<0>(mine); wq[1] = quad_bcast<1>(mine); wq[2] = quad_bcast<2>(mine); wq[3] = quad_bcast<3>(mine);
                        }
                        uint32_t sw[2];
#pragma unroll
                        for (int e = 0; e < 4; ++e) {
                            const int r = 4 * g + e;
                            float v = s[r] * a.scale;
                            if (REL) v += relv[kj - (il + e) + L];
                            v += madd;
                            const float pr = __expf(v - l4[e]);
                            float mk = 1.f;
                            if (DROP && dc.thr) mk = (((wq[e] >> dsh) & 0xFFu) >= dc.thr) ? dc.scale : 0.f;
                            const float dsr = pr * (dp[r] * mk - d4[e]);               // dS (unscaled) = d(score)
                            dp[r] = pr * mk;                      // dP is spent: its register takes the (dropped) probability dV needs
                            s[r] = dsr;
                        }
                        sw[0] = pack_op2(s[4 * g], s[4 * g + 1]);
                        sw[1] = pack_op2(s[4 * g + 2], s[4 * g + 3]);
                        u32x2 pkd; pkd[0] = sw[0]; pkd[1] = sw[1];
                        *(u32x2*)(dsimg + ds_off(32 * wave + fr, (it * 32 + 8 * g + 4 * h) * 2)) = pkd;
                    }
                    if (REL && a.drel) {
                        float lo = 0.f, hi = 0.f;
#pragma unroll
                        for (int r = 0; r < 16; ++r) diag_add(s[r], r, lane, lo, hi);
                        diag_store(drel_s + wave * 2 * L, L, j0, i0, lane, lo, hi);
                    }
#pragma unroll
                    for (int ks = 0; ks < 2; ++ks) {
                        const op16x8 pf = acc_frag(dp, ks), sf = acc_frag(s, ks);
#pragma unroll
                        for (int b = 0; b < DB; ++b) {
                            const op16x8 dt = tr_frag_rr64(dimg, it * 32 + 16 * ks, b, lane);
                            const op16x8 qt = tr_frag_rr64(qimg, it * 32 + 16 * ks, b, lane);
                            dv[b] = mfma32_op(dt, pf, dv[b]);
                            dk[b] = mfma32_op(qt, sf, dk[b]);
                        }
                    }
                }
            }
            __syncthreads();                                 // every wave's dS tiles of this chunk are in the image
            if (qt_mine * 32 < rows) {
                // dQ^T[dd half b][query tile qt] += sum over this pass's keys K^T . dS^T
                const int li = lane & 15, q = li >> 2, pp = li & 3, gsel = (lane >> 4) & 1;
                const int byte = qt_mine * 64 + gsel * 32 + 8 * pp;
                // (the chunk loop is a real loop -- one copy of the score-tile code -- so the accumulator of chunk c is picked by a
                // uniform switch around the k loop: dq[] stays in registers only while every index is a constant)
                auto accumulate = [&](f32x16& acc) {
                    for (int ks = 0; ks < nkeys / 16; ++ks) {
                        const op16x4 lo = lds_tr(dsimg + ds_off(16 * ks + 4 * h + q, byte));
                        const op16x4 hi = lds_tr(dsimg + ds_off(16 * ks + 8 + 4 * h + q, byte));
                        op16x8 df;
#pragma unroll
                        for (int e = 0; e < 4; ++e) { df[e] = lo[e]; df[4 + e] = hi[e]; }
                        const op16x8 kt = tr_frag_rr64(kimg, 16 * ks, b_mine, lane);
                        acc = mfma32_op(kt, df, acc);
                    }
                };
                if (c == 0) accumulate(dq[0]);
                if (NC > 1 && c == 1) accumulate(dq[1]);
                if (NC > 2 && c == 2) accumulate(dq[2]);
                if (NC > 3 && c == 3) accumulate(dq[3]);
            }
        }
        if (active) {
            op16* krow = a.dqkv + qkv_base(a, seq, head, D) + (size_t)kj * ld + a.woff;
            op16* vrow = krow + a.woff;
#pragma unroll
            for (int b = 0; b < DB; ++b)
#pragma unroll
                for (int g = 0; g < 4; ++g) {
                    u32x2 pk;
                    pk[0] = pack_op2(dk[b][4 * g] * a.scale, dk[b][4 * g + 1] * a.scale);
                    pk[1] = pack_op2(dk[b][4 * g + 2] * a.scale, dk[b][4 * g + 3] * a.scale);
                    *(u32x2*)(krow + b * 32 + 8 * g + 4 * h) = pk;
                    pk[0] = pack_op2(dv[b][4 * g], dv[b][4 * g + 1]);
                    pk[1] = pack_op2(dv[b][4 * g + 2], dv[b][4 * g + 3]);
                    *(u32x2*)(vrow + b * 32 + 8 * g + 4 * h) = pk;
                }
        }
    }
    // dQ: this wave's tile of every chunk (rows = d inside half b, column = query on the lane)
#pragma unroll
    for (int c = 0; c < NC; ++c) {
        const int qi = c * 128 + qt_mine * 32 + fr;
        if (c >= nchunk || qi >= L) continue;
        op16* orow = a.dqkv + qkv_base(a, seq, head, D) + (size_t)qi * ld + b_mine * 32;
#pragma unroll
        for (int g = 0; g < 4; ++g) {
            u32x2 pk;
            pk[0] = pack_op2(dq[c][4 * g] * a.scale, dq[c][4 * g + 1] * a.scale);
            pk[1] = pack_op2(dq[c][4 * g + 2] * a.scale, dq[c][4 * g + 3] * a.scale);
            *(u32x2*)(orow + 8 * g + 4 * h) = pk;
        }
    }
    if (REL && a.drel) {
        __syncthreads();
        for (int t = tid; t < 2 * L; t += 512) {
            float v = 0.f;
#pragma unroll
            for (int w = 0; w < 8; ++w) v += drel_s[w * 2 * L + t];
            if (v != 0.f) atomicAdd(a.drel + (size_t)head * 2 * L + t, v);
        }
    }
}

template <typename K>
int set_lds(K kern, size_t bytes) {
    QST_HIP_CHECK(hipFuncSetAttribute((const void*)kern, hipFuncAttributeMaxDynamicSharedMemorySize, (int)bytes));
    return QST_OK;
}

}  // namespace

static int check_attn(int nseq, int L, int A, int d) {
    if (nseq <= 0 || L <= 0 || A <= 0) return QST_ERR_BAD_ARG;
    if ((d != 32 && d != 64) || (L % 32) != 0 || L > 512) return QST_ERR_UNSUPPORTED;
    return QST_OK;
}

static int fill_args(const QstAttnDesc* q, AttnArgs& a) {
    if (!q || !q->qkv || !q->mask || !q->ctx) return QST_ERR_BAD_ARG;
    if (int rc = check_attn(q->nseq, q->L, q->A, q->d)) return rc;
    a.qkv = (const op16*)q->qkv; a.mask = q->mask; a.rel = q->rel_pos;
    a.nseq = q->nseq; a.L = q->L; a.A = q->A; a.H = q->A * q->d; a.scale = 1.0f / sqrtf((float)q->d);
    a.ld = 3 * a.H; a.woff = a.H;
    a.drop = q->drop;
    if (a.drop.thr16 > 65535u) return QST_ERR_BAD_ARG;
    if (!a.drop.state) a.drop.thr16 = 0u;
    if (a.drop.thr16 && (int64_t)q->nseq * q->A * q->L * q->L >= ((int64_t)1 << 32)) return QST_ERR_UNSUPPORTED;
    return QST_OK;
}

extern "C" int QST_K(qst_attention_fwd_ex)(const QstAttnDesc* q, void* stream) {
    AttnArgs a{};
    if (int rc = fill_args(q, a)) return rc;
    a.out = (op16*)q->ctx; a.lse_out = q->lse;
    const int nseq = a.nseq, L = a.L, A = a.A, d = q->d;
    const bool rel = a.rel != nullptr, drop = a.drop.thr16 != 0u;
    int rc;
    const int grid = nseq * A * ((L + 127) / 128);
    const size_t lds = (size_t)2 * 128 * d * 2 + (((size_t)L * 4 + 15) & ~(size_t)15) + (size_t)4 * 32 * d * 2 +
                       (rel ? (size_t)2 * L * 4 : 0);
    hipStream_t st = (hipStream_t)stream;
#define QST_FWD(D_, DR_) do { if ((rc = set_lds(attn_fwd_kernel<D_, DR_>, lds))) return rc; \
                              attn_fwd_kernel<D_, DR_><<<grid, 256, lds, st>>>(a); } while (0)
    if (d == 32) { if (drop) QST_FWD(32, true); else QST_FWD(32, false); }
    else         { if (drop) QST_FWD(64, true); else QST_FWD(64, false); }
#undef QST_FWD
    QST_LAUNCH_CHECK();
    return QST_OK;
}

extern "C" int QST_K(qst_attention_bwd_ex)(const QstAttnDesc* q, void* stream) {
    AttnArgs a{};
    if (int rc = fill_args(q, a)) return rc;
    if (!q->dctx || !q->lse || !q->dqkv || !q->delta_scratch) return QST_ERR_BAD_ARG;
    if (q->drel && !q->rel_pos) return QST_ERR_BAD_ARG;
    a.ctx = (const op16*)q->ctx; a.dctx = (const op16*)q->dctx; a.lse_in = q->lse;
    a.dqkv = (op16*)q->dqkv; a.drel = q->drel; a.delta = q->delta_scratch;
    const int nseq = a.nseq, L = a.L, A = a.A, d = q->d;
    const bool rel = a.rel != nullptr, drop = a.drop.thr16 != 0u;
    int rc;
    const int grid = nseq * A * ((L + 127) / 128);
    const size_t lds_q = (size_t)3 * 128 * d * 2 + (((size_t)L + 3) & ~(size_t)3) * 4 + (rel ? (size_t)2 * L * 4 : 0);
    const size_t lds_kv = (size_t)4 * 128 * d * 2 + 256 * 4 + (rel ? (size_t)10 * L * 4 : 0);
    hipStream_t st = (hipStream_t)stream;
#define QST_RUN(K_, G_, LDS_) do { if ((rc = set_lds(K_, LDS_))) return rc; K_<<<G_, 256, LDS_, st>>>(a); QST_LAUNCH_CHECK(); } while (0)
    // (the single-workgroup kernel addresses q/k/v through 32-bit buffer offsets)
    const bool fits32 = (int64_t)nseq * L * 3 * A * d * 2 < 0x7FFFFF00LL;
    if (L <= 128 && d == 32 && !q->force_split && fits32) {
        // one workgroup per (sequence, head) computes dQ, dK and dV from a single evaluation of the score tile
        const size_t lds_f = (size_t)5 * 128 * d * 2 + DS_IMG + 256 * 4 + (rel ? (size_t)10 * L * 4 : 0);
        const int gf = min(nseq * A, 512);                                            // two per CU, persistent
        if (rel) { if (drop) QST_RUN((attn_bwd_fused_kernel<32, true, true>), gf, lds_f); else QST_RUN((attn_bwd_fused_kernel<32, true, false>), gf, lds_f); }
        else     { if (drop) QST_RUN((attn_bwd_fused_kernel<32, false, true>), gf, lds_f); else QST_RUN((attn_bwd_fused_kernel<32, false, false>), gf, lds_f); }
        return QST_OK;
    }
    // d = 64: one 512-thread workgroup per (sequence, head), a single evaluation of the scores, dQ kept in registers across the
    // key passes (attn_bwd_one64_kernel); needs 128 KB of images + the per-query constants (+ 18 L floats with the position
    // bias): every L <= 512 without the bias, L <= 384 with it. Measured against the two-kernel path (tools/one_attn_bwd.py,
    // round 4, us): mpnet shape 128 x 256 x 12 heads with bias + dropout 259 vs 273, dropout only 205 vs 225, neither 186 vs
    // 204; bert-base configs[4] shape 512 x 384 with dropout 1,560 vs 1,665, without 1,366 vs 1,441. (The first version, with
    // the chunk loop unrolled so that dq[c] had a constant index, spilled 100-500 bytes per lane in the masked variants and lost
    // to the pair by 8-16%; as a real loop with the accumulator picked by a uniform switch it is 0-44 bytes at L <= 256, 32 at
    // L = 384 with dropout.) force_split: 1 = the pair, 2 = this kernel.
    // (QST_ATTN_BWD_PAIR=1 in the environment: the kernel pair everywhere -- for A/B runs of the whole step, read once)
    static const bool env_pair = [] { const char* e = getenv("QST_ATTN_BWD_PAIR"); return e && e[0] == '1'; }();
    const bool one64_auto = !q->force_split && !env_pair;
    if (d == 64 && (q->force_split == 2 || one64_auto)) {
        const size_t lds_o = (size_t)131072 + (size_t)2 * L * 4 + (rel ? (size_t)18 * L * 4 : 0);
        if (lds_o <= 163840) {
            const int go = nseq * A;
#define QST_ONE(NC_) do { \
            if (rel) { if (drop) QST_RUN512((attn_bwd_one64_kernel<NC_, true, true>), go, lds_o); else QST_RUN512((attn_bwd_one64_kernel<NC_, true, false>), go, lds_o); } \
            else     { if (drop) QST_RUN512((attn_bwd_one64_kernel<NC_, false, true>), go, lds_o); else QST_RUN512((attn_bwd_one64_kernel<NC_, false, false>), go, lds_o); } } while (0)
#define QST_RUN512(K_, G_, LDS_) do { if ((rc = set_lds(K_, LDS_))) return rc; K_<<<G_, 512, LDS_, st>>>(a); QST_LAUNCH_CHECK(); } while (0)
            if (L <= 256) QST_ONE(2); else if (L <= 384) QST_ONE(3); else QST_ONE(4);
#undef QST_RUN512
#undef QST_ONE
            return QST_OK;
        }
    }
    if (d == 32) {
        if (drop) QST_RUN((attn_bwd_dq_kernel<32, true>), grid, lds_q); else QST_RUN((attn_bwd_dq_kernel<32, false>), grid, lds_q);
        if (rel) { if (drop) QST_RUN((attn_bwd_dkv_kernel<32, true, true>), grid, lds_kv); else QST_RUN((attn_bwd_dkv_kernel<32, true, false>), grid, lds_kv); }
        else     { if (drop) QST_RUN((attn_bwd_dkv_kernel<32, false, true>), grid, lds_kv); else QST_RUN((attn_bwd_dkv_kernel<32, false, false>), grid, lds_kv); }
    } else {
        if (drop) QST_RUN((attn_bwd_dq_kernel<64, true>), grid, lds_q); else QST_RUN((attn_bwd_dq_kernel<64, false>), grid, lds_q);
        if (rel) { if (drop) QST_RUN((attn_bwd_dkv_kernel<64, true, true>), grid, lds_kv); else QST_RUN((attn_bwd_dkv_kernel<64, true, false>), grid, lds_kv); }
        else     { if (drop) QST_RUN((attn_bwd_dkv_kernel<64, false, true>), grid, lds_kv); else QST_RUN((attn_bwd_dkv_kernel<64, false, false>), grid, lds_kv); }
    }
#undef QST_RUN
    return QST_OK;
}

extern "C" int QST_K(qst_attention_fwd)(const void* qkv, const int64_t* mask, const float* rel_bias, int nseq, int L, int A,
                                 int d, void* ctx, float* lse, void* stream) {
    QstAttnDesc q{};
    q.qkv = qkv; q.mask = mask; q.rel_pos = rel_bias; q.nseq = nseq; q.L = L; q.A = A; q.d = d; q.ctx = ctx; q.lse = lse;
    return QST_K(qst_attention_fwd_ex)(&q, stream);
}
extern "C" int QST_K(qst_attention_bwd)(const void* qkv, const void* ctx, const void* dctx, const float* lse,
                                 const int64_t* mask, const float* rel_bias, int nseq, int L, int A, int d,
                                 void* dqkv, float* drel, float* delta_scratch, void* stream) {
    QstAttnDesc q{};
    q.qkv = qkv; q.mask = mask; q.rel_pos = rel_bias; q.nseq = nseq; q.L = L; q.A = A; q.d = d; q.ctx = (void*)ctx;
    q.lse = (float*)lse; q.dctx = dctx; q.dqkv = dqkv; q.drel = drel; q.delta_scratch = delta_scratch;
    return QST_K(qst_attention_bwd_ex)(&q, stream);
}

#if !QST_OP_F16
// Diagnostic (not declared in the public headers): resident workgroups per CU the runtime reports for the d=32 kernels.
extern "C" int qst_debug_attn_occupancy(int which, int lds_bytes) {
    int n = -1;
    if (which == 0) (void)hipOccupancyMaxActiveBlocksPerMultiprocessor(&n, attn_fwd_kernel<32>, 256, lds_bytes);
    if (which == 1) (void)hipOccupancyMaxActiveBlocksPerMultiprocessor(&n, attn_bwd_dq_kernel<32>, 256, lds_bytes);
    if (which == 2) (void)hipOccupancyMaxActiveBlocksPerMultiprocessor(&n, attn_bwd_dkv_kernel<32, false>, 256, lds_bytes);
    return n;
}
#endif
