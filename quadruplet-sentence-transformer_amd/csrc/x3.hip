// x3.hip -- the parity-precision forward path (QST_PREC_BF16X3): fp32 activations and weights, every MFMA operand
// split on the fly into hi + lo bf16 (hi = bf16(x), lo = bf16(x - hi)) and every product issued as three MFMAs
// (hi*hi + hi*lo + lo*hi; the dropped lo*lo term is ~2^-18 relative). Accumulation stays fp32, so the
// contractions are fp32-class (~2^-17) and the embeddings meet rtol 1e-3 / atol 1e-4 against the fp32 CPU
// reference, which single-rounded bf16 operands cannot. Forward only: encode() and
// evaluation use it; training runs the bf16 path.
//
//   gemm_nt_x3 : C[M,N] (fp32) = A[M,K] (fp32) . B[N,K]^T (fp32) + bias [+ resid] [GELU]
//   attn_fwd_x3: softmax(Q K^T / sqrt(d) + rel + mask) V on fp32 qkv -> fp32 ctx
#include <algorithm>
#include "qst_common.h"
#include "qst_kernels.h"

namespace {

typedef __attribute__((ext_vector_type(2))) float f32x2;

__device__ __forceinline__ void split4(const f32x4 v, u32x2& hi, u32x2& lo) {
    float r[4];
#pragma unroll
    for (int e = 0; e < 4; ++e) r[e] = v[e] - bf2f(f2bf(v[e]));
    hi[0] = pack_bf16x2(v[0], v[1]); hi[1] = pack_bf16x2(v[2], v[3]);
    lo[0] = pack_bf16x2(r[0], r[1]); lo[1] = pack_bf16x2(r[2], r[3]);
}
__device__ __forceinline__ f32x16 mfma3(bf16x8 ah, bf16x8 al, bf16x8 bh, bf16x8 bl, f32x16 c) {
    c = __builtin_amdgcn_mfma_f32_32x32x16_bf16(al, bh, c, 0, 0, 0);
    c = __builtin_amdgcn_mfma_f32_32x32x16_bf16(ah, bl, c, 0, 0, 0);
    c = __builtin_amdgcn_mfma_f32_32x32x16_bf16(ah, bh, c, 0, 0, 0);
    return c;
}

// LDS image geometry shared by the attention kernels and the TN GEMM: rr = rows read as rows (one 16-byte chunk = 8
// k-values of a row), tr = rows read through ds_read_b64_tr_b16 (a fragment whose k runs DOWN the image rows)
template <int D> __device__ __forceinline__ uint32_t rr_off(int row, int chunk) {
    if (D == 32) return (uint32_t)(row * 64 + ((chunk ^ ((row >> 2) & 3)) << 4));
    return (uint32_t)(row * 128 + ((chunk ^ ((row >> 1) & 7)) << 4));
}
template <int D> __device__ __forceinline__ uint32_t tr_off(int row, int byte) {
    if (D == 32) return (uint32_t)(row * 64 + byte);
    return (uint32_t)(row * 128 + (byte ^ (((row >> 1) & 1) << 6)));
}
__device__ __forceinline__ bf16x4 lds_tr(const char* p) {
    return __builtin_amdgcn_ds_read_tr16_b64_v4bf16((__attribute__((address_space(3))) bf16x4*)(p));
}
template <int D> __device__ __forceinline__ bf16x8 tr_frag(const char* img, int row0, int ddb, int lane) {
    const int li = lane & 15, q = li >> 2, p = li & 3, gsel = (lane >> 4) & 1, h = lane >> 5;
    const int byte = ddb * 64 + gsel * 32 + 8 * p;
    const bf16x4 a = lds_tr(img + tr_off<D>(row0 + 4 * h + q, byte));
    const bf16x4 b = lds_tr(img + tr_off<D>(row0 + 8 + 4 * h + q, byte));
    bf16x8 f;
#pragma unroll
    for (int e = 0; e < 4; ++e) { f[e] = a[e]; f[4 + e] = b[e]; }
    return f;
}

// ---------------------------------------------------------------- GEMM
constexpr int XBK = 32;
// per-operand image [128 rows][32 bf16] (64-byte rows), chunk c of row r at position c ^ ((r>>2)&3)
__device__ __forceinline__ uint32_t x_off(int row, int chunk) {
    return (uint32_t)(row * 64 + ((chunk ^ ((row >> 2) & 3)) << 4));
}

// EPI 0: +bias ; 1: +bias +resid ; 2: gelu(+bias) ; 3: C += A . B^T over this workgroup's share of K (fp32 atomics) ;
// 4: C = +bias and C2 = gelu(C) ; 5: C = (A . B^T) * gelu'(aux).
// TN (EPI 3 only): the operands are given with the reduction index as their ROW index -- A [K, M], B [K, N] row-major, i.e.
// dW[out, in] += dY[tokens, out]^T . X[tokens, in] straight from the activations -- staged as [32 k][64 column] sub-images
// and read with transposing LDS reads; column sums of A (the bias gradient) ride along in the tiles of the first column panel.
// BK = K-tile depth. 32 is what runs: 64-deep tiles (two 64 KB ring slots, one workgroup per CU) measured slower -- weight
// gradients of a MiniLM layer 320 vs 228 us, the bf16x3 training step 18.5 vs 14.7 ms.
template <int EPI, bool TN = false, int BK = 32>
__global__ __launch_bounds__(256, 2) void gemm_nt_x3_kernel(QstGemmArgs g) {
    extern __shared__ __attribute__((aligned(16))) char smem[];   // A hi | A lo | B hi | B lo (IMG bytes each); epilogue 34 KB
    constexpr int XBK = BK, IMG = 128 * BK * 2, C4 = BK / 4, NP = TN ? BK / 8 : 128 * C4 / 256, SUB = BK * 128;
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int wm = wave >> 1, wn = wave & 1;
    // Workgroup -> work, XCD-aware (consecutive workgroup ids go round the eight XCDs, each with its own L2): the workgroups
    // that read the same rows sit on ONE XCD. NT: the ntn tiles of a 128-row A panel (panel = xcd + 8 * ...). Shared reduction
    // (EPI 3): all tiles of one share of K, whose operand rows every tile of the share reads. Padding workgroups return.
    const int ntn = (g.N + 127) / 128, ntm = (g.M + 127) / 128;
    const int xcd = blockIdx.x & 7, jx = blockIdx.x >> 3;
    const int group = xcd + 8 * (EPI == 3 ? jx / (ntm * ntn) : jx / ntn);       // A panel (NT) or share of K (EPI 3)
    const int tile = EPI == 3 ? jx % (ntm * ntn) : group * ntn + jx % ntn;
    if (EPI != 3 && group >= ntm) return;
    const int m0 = (tile / ntn) * 128, n0 = (tile % ntn) * 128;
    const float* A = (const float*)g.A;
    const float* B = (const float*)g.B;
    // EPI 3 (weight gradients: K = the token rows): gridDim.y workgroups share the reduction, g.splits K-tiles each
    const int kt0 = EPI == 3 ? group * g.splits : 0;
    const int nk = EPI == 3 ? min(g.K / XBK, kt0 + g.splits) : g.K / XBK;
    if (kt0 >= nk) return;
    const __amdgpu_buffer_rsrc_t ra = TN
        ? make_rsrc(A + (size_t)kt0 * XBK * g.lda, (uint32_t)min((size_t)(g.K - kt0 * XBK) * g.lda * 4u, (size_t)0x7FFFFF00u))
        : make_rsrc(A + (size_t)m0 * g.lda, (uint32_t)min(128, g.M - m0) * g.lda * 4u);
    const __amdgpu_buffer_rsrc_t rb = TN
        ? make_rsrc(B + (size_t)kt0 * XBK * g.ldb, (uint32_t)min((size_t)(g.K - kt0 * XBK) * g.ldb * 4u, (size_t)0x7FFFFF00u))
        : make_rsrc(B + (size_t)n0 * g.ldb, (uint32_t)min(128, g.N - n0) * g.ldb * 4u);
    // staging, NT: a tile is 128 rows x BK floats; thread t takes rows t / C4 + (256 / C4) i, float4 column t % C4
    //          TN: a tile is BK k-rows x 128 floats; thread t takes k-rows t/32 + 8 i, float4 column t%32
    const int srow = TN ? tid >> 5 : tid / C4, sc4 = TN ? tid & 31 : tid % C4;
    const bool do_bias = TN && g.colsum && n0 == 0;
    f32x4 bsum = {0.f, 0.f, 0.f, 0.f};
    u32x4 sa0[NP], sb0[NP], sa1[NP], sb1[NP];          // two K-tiles of operand rows in flight (global -> registers)
    auto gload = [&](int kt, u32x4 (&sa)[NP], u32x4 (&sb)[NP]) {
        const bool live = kt < nk;                          // (uniform) past the end: out-of-range offsets, zeros come back
#pragma unroll
        for (int i = 0; i < NP; ++i) {
            if (TN) {
                const uint32_t r = (uint32_t)((kt - kt0) * XBK + srow + 8 * i);
                // columns past the matrix must not alias the next row: those lanes ask for an out-of-range offset (zeros)
                sa[i] = buf_load16(ra, live && m0 + sc4 * 4 < g.M ? r * g.lda * 4u + (uint32_t)(m0 + sc4 * 4) * 4u : 0x80000000u);
                sb[i] = buf_load16(rb, live && n0 + sc4 * 4 < g.N ? r * g.ldb * 4u + (uint32_t)(n0 + sc4 * 4) * 4u : 0x80000000u);
            } else {
                const int r = srow + (256 / C4) * i;
                sa[i] = buf_load16(ra, live ? (uint32_t)r * g.lda * 4u + (uint32_t)(kt * XBK + sc4 * 4) * 4u : 0x80000000u);
                sb[i] = buf_load16(rb, live ? (uint32_t)r * g.ldb * 4u + (uint32_t)(kt * XBK + sc4 * 4) * 4u : 0x80000000u);
            }
        }
    };
    auto lstore = [&](char* smem, u32x4 (&sa)[NP], u32x4 (&sb)[NP]) {   // (smem: the ring slot being filled)
#pragma unroll
        for (int i = 0; i < NP; ++i) {
            const int r = TN ? srow + 8 * i : srow + (256 / C4) * i;
            const uint32_t o = TN ? (uint32_t)(sc4 >> 4) * SUB + tr_off<64>(r, (sc4 & 15) * 8)
                                  : (BK == 32 ? x_off(r, sc4 >> 1) : rr_off<64>(r, sc4 >> 1)) + (sc4 & 1) * 8;
            u32x2 hi, lo;
            split4(__builtin_bit_cast(f32x4, sa[i]), hi, lo);
            *(u32x2*)(smem + o) = hi;
            *(u32x2*)(smem + IMG + o) = lo;
            split4(__builtin_bit_cast(f32x4, sb[i]), hi, lo);
            *(u32x2*)(smem + 2 * IMG + o) = hi;
            *(u32x2*)(smem + 3 * IMG + o) = lo;
            if (do_bias) bsum += __builtin_bit_cast(f32x4, sa[i]);
        }
    };
    f32x16 acc[2][2];
#pragma unroll
    for (int i = 0; i < 2; ++i)
#pragma unroll
        for (int j = 0; j < 2; ++j)
#pragma unroll
            for (int r = 0; r < 16; ++r) acc[i][j][r] = 0.f;
    const int fr = lane & 31, fh = lane >> 5;
    // Two ring slots of operand images: K-tile kt + 1 is split and written into the other slot after this wave has issued the
    // MFMAs of K-tile kt, so one wave's conversion work runs under the others' MFMAs and a K-tile costs ONE barrier. The rows
    // of K-tile kt + 2 are requested before the MFMAs of kt: one K-tile of MFMAs (~1,500 cycles) does not cover a load's
    // round trip, two register sets do.
    char* const ring = smem;
    auto compute = [&](const char* smem) {
#pragma unroll
        for (int ks = 0; ks < BK / 16; ++ks) {
            bf16x8 ah[2], al[2], bh[2], bl[2];
#pragma unroll
            for (int i = 0; i < 2; ++i) {
                if (TN) {
                    ah[i] = tr_frag<64>(smem + wm * SUB, ks * 16, i, lane);
                    al[i] = tr_frag<64>(smem + IMG + wm * SUB, ks * 16, i, lane);
                    bh[i] = tr_frag<64>(smem + 2 * IMG + wn * SUB, ks * 16, i, lane);
                    bl[i] = tr_frag<64>(smem + 3 * IMG + wn * SUB, ks * 16, i, lane);
                    continue;
                }
                const int ra_ = wm * 64 + i * 32 + fr, rb_ = wn * 64 + i * 32 + fr;
                const uint32_t oa = BK == 32 ? x_off(ra_, ks * 2 + fh) : rr_off<64>(ra_, ks * 2 + fh);
                const uint32_t ob = BK == 32 ? x_off(rb_, ks * 2 + fh) : rr_off<64>(rb_, ks * 2 + fh);
                ah[i] = *(const bf16x8*)(smem + oa);
                al[i] = *(const bf16x8*)(smem + IMG + oa);
                bh[i] = *(const bf16x8*)(smem + 2 * IMG + ob);
                bl[i] = *(const bf16x8*)(smem + 3 * IMG + ob);
            }
#pragma unroll
            for (int i = 0; i < 2; ++i)
#pragma unroll
                for (int j = 0; j < 2; ++j) acc[i][j] = mfma3(bh[j], bl[j], ah[i], al[i], acc[i][j]);   // D rows = n, col = m
        }
    };
    // Branch-free: K-tiles past the end are requested at an out-of-range offset (zeros: they add nothing), so the loop always
    // runs whole pairs. With the loads under `if (kt + 2 < nk)` hipcc's waitcnt pass drained the queue before every request.
    gload(kt0, sa0, sb0);
    gload(kt0 + 1, sa1, sb1);
    lstore(ring, sa0, sb0);
    __syncthreads();
    for (int kt = kt0; kt < nk; kt += 2) {
        gload(kt + 2, sa0, sb0);         // slot 0 holds kt, set 1 holds kt + 1 (in flight), set 0 is free
        __builtin_amdgcn_sched_barrier(0);       // (the scheduler otherwise sinks the requests below the MFMAs and the stores)
        compute(ring);
        lstore(ring + 4 * IMG, sa1, sb1);
        __syncthreads();                 // slot 1 is complete; every wave has left slot 0
        gload(kt + 3, sa1, sb1);         // slot 1 holds kt + 1, set 0 holds kt + 2 (in flight), set 1 is free
        __builtin_amdgcn_sched_barrier(0);
        compute(ring + 4 * IMG);
        lstore(ring, sa0, sb0);
        __syncthreads();
    }
    if (do_bias) {                                 // 8 k-row phases x 128 columns of partial sums -> one atomic per column
        float* red = (float*)smem;
        *(f32x4*)(red + srow * 128 + sc4 * 4) = bsum;
        __syncthreads();
        if (tid < 128 && m0 + tid < g.M) {
            float t = 0.f;
#pragma unroll
            for (int k = 0; k < 8; ++k) t += red[k * 128 + tid];
            atomicAdd(g.colsum + m0 + tid, t);
        }
        __syncthreads();
    }
    float* stg = (float*)smem + wave * (32 * 68);
    const int c4 = lane & 15, rsub = lane >> 4;
    const int n = n0 + wn * 64 + c4 * 4;
    f32x4 bias = {0.f, 0.f, 0.f, 0.f};
    if (g.bias && n < g.N) bias = *(const f32x4*)(g.bias + n);
#pragma unroll
    for (int i = 0; i < 2; ++i) {
#pragma unroll
        for (int j = 0; j < 2; ++j)
#pragma unroll
            for (int g4 = 0; g4 < 4; ++g4) {
                f32x4 v;
#pragma unroll
                for (int e = 0; e < 4; ++e) v[e] = acc[i][j][4 * g4 + e];
                *(f32x4*)(stg + fr * 68 + j * 32 + 8 * g4 + 4 * fh) = v;
            }
        if (n < g.N) {
#pragma unroll 4
            for (int t = 0; t < 8; ++t) {
                const int row = t * 4 + rsub;
                const int m = m0 + wm * 64 + i * 32 + row;
                if (m >= g.M) continue;
                f32x4 v = *(const f32x4*)(stg + row * 68 + c4 * 4);
                if (EPI == 3) {
#pragma unroll
                    for (int e = 0; e < 4; ++e) atomicAdd((float*)g.C + (size_t)m * g.ldc + n + e, v[e]);
                    continue;
                }
                v += bias;
                if (EPI == 1 && g.resid) v += *(const f32x4*)(g.resid + (size_t)m * g.ldr + n);
                if (EPI == 2) {
#pragma unroll
                    for (int e = 0; e < 4; ++e) v[e] = 0.5f * v[e] * (1.0f + erff(v[e] * 0.70710678118654752f));   // exact erf here
                }
                if (EPI == 4) {                         // training FFN-1: u = A W^T + b to C, h = gelu(u) to C2
                    f32x4 hh;
#pragma unroll
                    for (int e = 0; e < 4; ++e) hh[e] = 0.5f * v[e] * (1.0f + erff(v[e] * 0.70710678118654752f));
                    *(f32x4*)((float*)g.C2 + (size_t)m * g.ldc + n) = hh;
                }
                if (EPI == 5) {                         // FFN-2 dgrad: du = (dY W) * gelu'(u), u = aux (fp32, the layout of C)
                    const f32x4 u = *(const f32x4*)((const float*)g.aux + (size_t)m * g.ldc + n);
#pragma unroll
                    for (int e = 0; e < 4; ++e)
                        v[e] *= 0.5f * (1.0f + erff(u[e] * 0.70710678118654752f)) + u[e] * 0.3989422804014327f * expf(-0.5f * u[e] * u[e]);
                }
                *(f32x4*)((float*)g.C + (size_t)m * g.ldc + n) = v;
            }
        }
    }
}

// ---------------------------------------------------------------- attention
constexpr float kMaskMin = -3.4028234663852886e38f;
// stage rows x D fp32 (row stride ld floats) into hi and lo bf16 LDS images
template <int D, bool TR>
__device__ __forceinline__ void stage_x3(char* hi_img, char* lo_img, const float* g, int ld, int rows, int tid) {
    constexpr int C4 = D / 4;
    for (int idx = tid; idx < rows * C4; idx += 256) {
        const int row = idx / C4, c = idx % C4;
        const f32x4 v = *(const f32x4*)(g + (size_t)row * ld + c * 4);
        u32x2 hi, lo;
        split4(v, hi, lo);
        const uint32_t off = (TR ? tr_off<D>(row, (c >> 1) * 16) : rr_off<D>(row, c >> 1)) + (c & 1) * 8;
        *(u32x2*)(hi_img + off) = hi;
        *(u32x2*)(lo_img + off) = lo;
    }
}
__device__ __forceinline__ void load_frag_x3(const float* p, bf16x8& hi, bf16x8& lo) {
    const f32x4 a = *(const f32x4*)p, b = *(const f32x4*)(p + 4);
#pragma unroll
    for (int e = 0; e < 4; ++e) {
        hi[e] = f2bf(a[e]); lo[e] = f2bf(a[e] - bf2f(hi[e]));
        hi[4 + e] = f2bf(b[e]); lo[4 + e] = f2bf(b[e] - bf2f(hi[4 + e]));
    }
}

struct AttnX3Args {
    const float* qkv; const int64_t* mask; const float* rel; float* out;
    int nseq, L, A, H; float scale;
    QstDrop drop;          // dropout of the probabilities (training), the 8-bit generator of the bf16 kernels
};

template <int D>
__global__ __launch_bounds__(256) void attn_fwd_x3_kernel(AttnX3Args a) {
    extern __shared__ __attribute__((aligned(16))) char smem[];
    constexpr int KS = D / 16, DB = D / 32, IMG = 128 * D * 2;
    char* khi = smem; char* klo = smem + IMG; char* vhi = smem + 2 * IMG; char* vlo = smem + 3 * IMG;
    float* madd = (float*)(smem + 4 * IMG);
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6, h = lane >> 5, fr = lane & 31;
    const int nqb = (a.L + 127) / 128;
    const int qb = blockIdx.x % nqb, head = (blockIdx.x / nqb) % a.A, seq = blockIdx.x / (nqb * a.A);
    const int ld = 3 * a.H;
    const float* base = a.qkv + (size_t)seq * a.L * ld + head * D;
    const int i0 = qb * 128 + wave * 32;
    const bool active = i0 < a.L;
    const int qi = i0 + fr;
    for (int t = tid; t < a.L; t += 256) madd[t] = a.mask[(size_t)seq * a.L + t] ? 0.f : kMaskMin;
    bf16x8 qh[KS], ql[KS];
    if (active) {
#pragma unroll
        for (int s = 0; s < KS; ++s) load_frag_x3(base + (size_t)qi * ld + 16 * s + 8 * h, qh[s], ql[s]);
    }
    f32x16 o[DB];
#pragma unroll
    for (int b = 0; b < DB; ++b)
#pragma unroll
        for (int r = 0; r < 16; ++r) o[b][r] = 0.f;
    float m = -INFINITY, l = 0.f;
    const DropCtx dc = drop_ctx8(a.drop);
    const uint32_t drow = ((uint32_t)(seq * a.A + head) * a.L + qi) * a.L;      // this lane's query row of the mask
    const int nchunk = (a.L + 127) / 128;
    for (int c = 0; c < nchunk; ++c) {
        const int rows = min(128, a.L - c * 128);
        __syncthreads();
        stage_x3<D, false>(khi, klo, base + (size_t)c * 128 * ld + a.H, ld, rows, tid);
        stage_x3<D, true>(vhi, vlo, base + (size_t)c * 128 * ld + 2 * a.H, ld, rows, tid);
        __syncthreads();
        if (!active) continue;
        for (int jt = 0; jt < rows / 32; ++jt) {
            f32x16 s;
#pragma unroll
            for (int r = 0; r < 16; ++r) s[r] = 0.f;
#pragma unroll
            for (int ks = 0; ks < KS; ++ks) {
                const uint32_t off = rr_off<D>(jt * 32 + fr, 2 * ks + h);
                s = mfma3(*(const bf16x8*)(khi + off), *(const bf16x8*)(klo + off), qh[ks], ql[ks], s);
            }
            const int j0 = c * 128 + jt * 32;
            float mx = -INFINITY;
#pragma unroll
            for (int r = 0; r < 16; ++r) {
                const int j = j0 + (r & 3) + 8 * (r >> 2) + 4 * h;
                float v = s[r] * a.scale;
                if (a.rel) v += a.rel[((size_t)head * a.L + qi) * a.L + j];
                v += madd[j];
                s[r] = v;
                mx = fmaxf(mx, v);
            }
            mx = fmaxf(mx, swap32(mx));
            const float mn = fmaxf(m, mx);
            const float alpha = expf(m - mn);
            float ps = 0.f;
#pragma unroll
            for (int r = 0; r < 16; ++r) { s[r] = expf(s[r] - mn); ps += s[r]; }
            ps += swap32(ps);
            l = l * alpha + ps;
            m = mn;
            if (dc.thr) {
                // dropped probabilities leave the P.V product only (l stays the softmax denominator, the 1 / (1 - p) scale joins
                // 1 / l at the end); registers 4g .. 4g+3 are four consecutive keys: the bytes of one random word
#pragma unroll
                for (int r = 0; r < 16; r += 4) {
                    const uint32_t w = drop_word4(dc, drow + j0 + 8 * (r >> 2) + 4 * h);
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
                bf16x8 ph, pl;
#pragma unroll
                for (int e = 0; e < 8; ++e) { ph[e] = f2bf(s[8 * ks + e]); pl[e] = f2bf(s[8 * ks + e] - bf2f(ph[e])); }
#pragma unroll
                for (int b = 0; b < DB; ++b)
                    o[b] = mfma3(tr_frag<D>(vhi, jt * 32 + 16 * ks, b, lane), tr_frag<D>(vlo, jt * 32 + 16 * ks, b, lane), ph, pl, o[b]);
            }
        }
    }
    if (!active) return;
    const float inv = dc.scale / l;
    float* orow = a.out + ((size_t)seq * a.L + qi) * a.H + head * D;
#pragma unroll
    for (int b = 0; b < DB; ++b)
#pragma unroll
        for (int g = 0; g < 4; ++g) {
            f32x4 v;
#pragma unroll
            for (int e = 0; e < 4; ++e) v[e] = o[b][4 * g + e] * inv;
            *(f32x4*)(orow + b * 32 + 8 * g + 4 * h) = v;
        }
}

// ---------------------------------------------------------------- attention backward on the x3 products
// Two kernels, both shaped like the forward (a wave owns 32 rows of one (sequence, head); the other side streams through LDS
// in chunks of 128 rows as hi / lo bf16 images), every contraction three MFMAs per product, everything elementwise in fp32:
//   dq kernel : wave = 32 queries. Pass 1 over the keys: row maximum m_i and l_i = sum_j exp(s_ij - m_i) (kept apart, as the
//               fp32 kernel of x3_bwd.hip does: a row with every key masked has m_i = -3.4e38); delta_i = dO_i . O_i from the
//               rows the lane loads anyway; all three go to `stats` for the second kernel. Pass 2: S^T = K Q^T and
//               dP^T = V dO^T (rows = keys in registers, column = the lane's query), dS = P (dP mask - delta),
//               dQ^T += K^T dS^T with the dS^T registers as the B operand (the forward's P V step).
//   dkv kernel: wave = 32 keys. S = Q K^T and dP = dO V^T (rows = queries in registers, column = the lane's key), then
//               dK^T += Q^T dS, dV^T += dO^T (P mask); the position-bias gradient leaves as one atomic per score.
// Scores are evaluated twice (once per kernel): the parity path trades those MFMAs for having no dS round trip.
struct AttnBwdX3Args {
    const float* qkv; const float* ctx; const float* dctx; const int64_t* mask; const float* rel;
    float* dqkv; float* drel; float* stats;       // stats: [3][nseq * A * L] = m, 1 / l, delta
    int nseq, L, A, H; float scale;
    QstDrop drop;
};

template <int D>
__global__ __launch_bounds__(256) void attn_bwd_x3_dq_kernel(AttnBwdX3Args a) {
    extern __shared__ __attribute__((aligned(16))) char smem[];
    constexpr int KS = D / 16, DB = D / 32, IMG = 128 * D * 2;
    char* khi = smem; char* klo = smem + IMG; char* kthi = smem + 2 * IMG; char* ktlo = smem + 3 * IMG;
    char* vhi = smem + 4 * IMG; char* vlo = smem + 5 * IMG;
    float* madd = (float*)(smem + 6 * IMG);
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6, h = lane >> 5, fr = lane & 31;
    const int nqb = (a.L + 127) / 128;
    const int qb = blockIdx.x % nqb, head = (blockIdx.x / nqb) % a.A, seq = blockIdx.x / (nqb * a.A);
    const int ld = 3 * a.H;
    const float* base = a.qkv + (size_t)seq * a.L * ld + head * D;
    const int i0 = qb * 128 + wave * 32;
    const bool active = i0 < a.L;
    const int qi = i0 + fr;
    for (int t = tid; t < a.L; t += 256) madd[t] = a.mask[(size_t)seq * a.L + t] ? 0.f : kMaskMin;
    bf16x8 qh[KS], ql[KS], doh[KS], dol[KS];
    float delta = 0.f;
    if (active) {
        const float* drow = a.dctx + ((size_t)seq * a.L + qi) * a.H + head * D;
        const float* orow = a.ctx + ((size_t)seq * a.L + qi) * a.H + head * D;
#pragma unroll
        for (int s = 0; s < KS; ++s) {
            load_frag_x3(base + (size_t)qi * ld + 16 * s + 8 * h, qh[s], ql[s]);
            load_frag_x3(drow + 16 * s + 8 * h, doh[s], dol[s]);
            const f32x4 d0 = *(const f32x4*)(drow + 16 * s + 8 * h), d1 = *(const f32x4*)(drow + 16 * s + 8 * h + 4);
            const f32x4 o0 = *(const f32x4*)(orow + 16 * s + 8 * h), o1 = *(const f32x4*)(orow + 16 * s + 8 * h + 4);
#pragma unroll
            for (int e = 0; e < 4; ++e) delta += d0[e] * o0[e] + d1[e] * o1[e];
        }
        delta += swap32(delta);
    }
    const int nchunk = (a.L + 127) / 128;
    auto scores = [&](int c, int jt, f32x16& s) {
#pragma unroll
        for (int r = 0; r < 16; ++r) s[r] = 0.f;
#pragma unroll
        for (int ks = 0; ks < KS; ++ks) {
            const uint32_t off = rr_off<D>(jt * 32 + fr, 2 * ks + h);
            s = mfma3(*(const bf16x8*)(khi + off), *(const bf16x8*)(klo + off), qh[ks], ql[ks], s);
        }
        const int j0 = c * 128 + jt * 32;
#pragma unroll
        for (int r = 0; r < 16; ++r) {
            const int j = j0 + (r & 3) + 8 * (r >> 2) + 4 * h;
            float v = s[r] * a.scale;
            if (a.rel) v += a.rel[((size_t)head * a.L + qi) * a.L + j];
            s[r] = v + madd[j];
        }
    };
    // ---- pass 1: m_i, l_i
    float m = -INFINITY, l = 0.f;
    for (int c = 0; c < nchunk; ++c) {
        const int rows = min(128, a.L - c * 128);
        __syncthreads();
        stage_x3<D, false>(khi, klo, base + (size_t)c * 128 * ld + a.H, ld, rows, tid);
        __syncthreads();
        if (!active) continue;
        for (int jt = 0; jt < rows / 32; ++jt) {
            f32x16 s;
            scores(c, jt, s);
            float mx = -INFINITY;
#pragma unroll
            for (int r = 0; r < 16; ++r) mx = fmaxf(mx, s[r]);
            mx = fmaxf(mx, swap32(mx));
            const float mn = fmaxf(m, mx);
            float ps = 0.f;
#pragma unroll
            for (int r = 0; r < 16; ++r) ps += expf(s[r] - mn);
            ps += swap32(ps);
            l = l * expf(m - mn) + ps;
            m = mn;
        }
    }
    const float inv_l = 1.0f / l;
    if (active && h == 0) {
        const size_t n = (size_t)a.nseq * a.A * a.L, row = ((size_t)seq * a.A + head) * a.L + qi;
        a.stats[row] = m; a.stats[n + row] = inv_l; a.stats[2 * n + row] = delta;
    }
    // ---- pass 2: dQ
    f32x16 dq[DB];
#pragma unroll
    for (int b = 0; b < DB; ++b)
#pragma unroll
        for (int r = 0; r < 16; ++r) dq[b][r] = 0.f;
    const DropCtx dc = drop_ctx8(a.drop);
    const uint32_t drow_idx = ((uint32_t)(seq * a.A + head) * a.L + qi) * a.L;
    for (int c = 0; c < nchunk; ++c) {
        const int rows = min(128, a.L - c * 128);
        __syncthreads();
        stage_x3<D, false>(khi, klo, base + (size_t)c * 128 * ld + a.H, ld, rows, tid);
        stage_x3<D, true>(kthi, ktlo, base + (size_t)c * 128 * ld + a.H, ld, rows, tid);
        stage_x3<D, false>(vhi, vlo, base + (size_t)c * 128 * ld + 2 * a.H, ld, rows, tid);
        __syncthreads();
        if (!active) continue;
        for (int jt = 0; jt < rows / 32; ++jt) {
            f32x16 s, dp;
            scores(c, jt, s);
#pragma unroll
            for (int r = 0; r < 16; ++r) dp[r] = 0.f;
#pragma unroll
            for (int ks = 0; ks < KS; ++ks) {
                const uint32_t off = rr_off<D>(jt * 32 + fr, 2 * ks + h);
                dp = mfma3(*(const bf16x8*)(vhi + off), *(const bf16x8*)(vlo + off), doh[ks], dol[ks], dp);
            }
            const int j0 = c * 128 + jt * 32;
#pragma unroll
            for (int r = 0; r < 16; r += 4) {
                float mk[4] = {1.f, 1.f, 1.f, 1.f};
                if (dc.thr) {                     // registers 4g .. 4g+3 are four consecutive keys: the bytes of one random word
                    const uint32_t w = drop_word4(dc, drow_idx + j0 + 8 * (r >> 2) + 4 * h);
                    mk[0] = drop_keep_byte<0>(dc, w) ? dc.scale : 0.f;
                    mk[1] = drop_keep_byte<1>(dc, w) ? dc.scale : 0.f;
                    mk[2] = drop_keep_byte<2>(dc, w) ? dc.scale : 0.f;
                    mk[3] = drop_keep_byte<3>(dc, w) ? dc.scale : 0.f;
                }
#pragma unroll
                for (int e = 0; e < 4; ++e) {
                    const float p = expf(s[r + e] - m) * inv_l;
                    s[r + e] = p * (dp[r + e] * mk[e] - delta);                   // dS
                }
            }
#pragma unroll
            for (int ks = 0; ks < 2; ++ks) {
                bf16x8 sh, sl;
#pragma unroll
                for (int e = 0; e < 8; ++e) { sh[e] = f2bf(s[8 * ks + e]); sl[e] = f2bf(s[8 * ks + e] - bf2f(sh[e])); }
#pragma unroll
                for (int b = 0; b < DB; ++b)
                    dq[b] = mfma3(tr_frag<D>(kthi, jt * 32 + 16 * ks, b, lane), tr_frag<D>(ktlo, jt * 32 + 16 * ks, b, lane), sh, sl, dq[b]);
            }
        }
    }
    if (!active) return;
    float* out = a.dqkv + ((size_t)seq * a.L + qi) * ld + head * D;
#pragma unroll
    for (int b = 0; b < DB; ++b)
#pragma unroll
        for (int g = 0; g < 4; ++g) {
            f32x4 v;
#pragma unroll
            for (int e = 0; e < 4; ++e) v[e] = dq[b][4 * g + e] * a.scale;
            *(f32x4*)(out + b * 32 + 8 * g + 4 * h) = v;
        }
}

template <int D>
__global__ __launch_bounds__(256) void attn_bwd_x3_dkv_kernel(AttnBwdX3Args a) {
    extern __shared__ __attribute__((aligned(16))) char smem[];
    constexpr int KS = D / 16, DB = D / 32, IMG = 128 * D * 2;
    char* qhi = smem; char* qlo = smem + IMG; char* qthi = smem + 2 * IMG; char* qtlo = smem + 3 * IMG;
    char* dhi = smem + 4 * IMG; char* dlo = smem + 5 * IMG; char* dthi = smem + 6 * IMG; char* dtlo = smem + 7 * IMG;
    float* sm = (float*)(smem + 8 * IMG);                  // [3][L]: m, 1 / l, delta of this (sequence, head)
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6, h = lane >> 5, fr = lane & 31;
    const int L = a.L, nkb = (L + 127) / 128;
    const int kb = blockIdx.x % nkb, head = (blockIdx.x / nkb) % a.A, seq = blockIdx.x / (nkb * a.A);
    const int ld = 3 * a.H;
    const float* base = a.qkv + (size_t)seq * L * ld + head * D;
    const float* dbase = a.dctx + (size_t)seq * L * a.H + head * D;
    const int j0 = kb * 128 + wave * 32;
    const bool active = j0 < L;
    const int kj = j0 + fr;
    {
        const size_t n = (size_t)a.nseq * a.A * L, row0 = ((size_t)seq * a.A + head) * L;
        for (int t = tid; t < 3 * L; t += 256) { const int w = t / L, i = t - w * L; sm[t] = a.stats[w * n + row0 + i]; }
    }
    bf16x8 kh[KS], kl[KS], vh[KS], vl[KS];
    float maddj = 0.f;
    if (active) {
#pragma unroll
        for (int s = 0; s < KS; ++s) {
            load_frag_x3(base + (size_t)kj * ld + a.H + 16 * s + 8 * h, kh[s], kl[s]);
            load_frag_x3(base + (size_t)kj * ld + 2 * a.H + 16 * s + 8 * h, vh[s], vl[s]);
        }
        maddj = a.mask[(size_t)seq * L + kj] ? 0.f : kMaskMin;
    }
    f32x16 dk[DB], dv[DB];
#pragma unroll
    for (int b = 0; b < DB; ++b)
#pragma unroll
        for (int r = 0; r < 16; ++r) { dk[b][r] = 0.f; dv[b][r] = 0.f; }
    const DropCtx dc = drop_ctx8(a.drop);
    const int nchunk = (L + 127) / 128;
    for (int c = 0; c < nchunk; ++c) {
        const int rows = min(128, L - c * 128);
        __syncthreads();
        stage_x3<D, false>(qhi, qlo, base + (size_t)c * 128 * ld, ld, rows, tid);
        stage_x3<D, true>(qthi, qtlo, base + (size_t)c * 128 * ld, ld, rows, tid);
        stage_x3<D, false>(dhi, dlo, dbase + (size_t)c * 128 * a.H, a.H, rows, tid);
        stage_x3<D, true>(dthi, dtlo, dbase + (size_t)c * 128 * a.H, a.H, rows, tid);
        __syncthreads();
        if (!active) continue;
        for (int it = 0; it < rows / 32; ++it) {
            f32x16 s, dp;
#pragma unroll
            for (int r = 0; r < 16; ++r) { s[r] = 0.f; dp[r] = 0.f; }
#pragma unroll
            for (int ks = 0; ks < KS; ++ks) {
                const uint32_t off = rr_off<D>(it * 32 + fr, 2 * ks + h);
                s = mfma3(*(const bf16x8*)(qhi + off), *(const bf16x8*)(qlo + off), kh[ks], kl[ks], s);
                dp = mfma3(*(const bf16x8*)(dhi + off), *(const bf16x8*)(dlo + off), vh[ks], vl[ks], dp);
            }
            const int ibase = c * 128 + it * 32;
#pragma unroll
            for (int r = 0; r < 16; ++r) {
                const int i = ibase + (r & 3) + 8 * (r >> 2) + 4 * h;           // the query of register r
                float v = s[r] * a.scale;
                if (a.rel) v += a.rel[((size_t)head * L + i) * L + kj];
                v += maddj;
                const float p = expf(v - sm[i]) * sm[L + i];
                float mk = 1.f;
                if (dc.thr) {
                    const uint32_t idx = ((uint32_t)(seq * a.A + head) * L + (uint32_t)i) * L + (uint32_t)kj;
                    mk = (((drop_word4(dc, idx) >> (8u * (idx & 3u))) & 0xFFu) >= dc.thr) ? dc.scale : 0.f;
                }
                const float ds = p * (dp[r] * mk - sm[2 * L + i]);
                if (a.drel) atomicAdd(a.drel + ((size_t)head * L + i) * L + kj, ds);
                s[r] = ds;
                dp[r] = p * mk;                                                 // what dV contracts with
            }
#pragma unroll
            for (int ks = 0; ks < 2; ++ks) {
                bf16x8 sh, sl, ph, pl;
#pragma unroll
                for (int e = 0; e < 8; ++e) {
                    sh[e] = f2bf(s[8 * ks + e]); sl[e] = f2bf(s[8 * ks + e] - bf2f(sh[e]));
                    ph[e] = f2bf(dp[8 * ks + e]); pl[e] = f2bf(dp[8 * ks + e] - bf2f(ph[e]));
                }
#pragma unroll
                for (int b = 0; b < DB; ++b) {
                    dk[b] = mfma3(tr_frag<D>(qthi, it * 32 + 16 * ks, b, lane), tr_frag<D>(qtlo, it * 32 + 16 * ks, b, lane), sh, sl, dk[b]);
                    dv[b] = mfma3(tr_frag<D>(dthi, it * 32 + 16 * ks, b, lane), tr_frag<D>(dtlo, it * 32 + 16 * ks, b, lane), ph, pl, dv[b]);
                }
            }
        }
    }
    if (!active) return;
    float* ok = a.dqkv + ((size_t)seq * L + kj) * ld + a.H + head * D;
    float* ov = ok + a.H;
#pragma unroll
    for (int b = 0; b < DB; ++b)
#pragma unroll
        for (int g = 0; g < 4; ++g) {
            f32x4 v, w;
#pragma unroll
            for (int e = 0; e < 4; ++e) { v[e] = dk[b][4 * g + e] * a.scale; w[e] = dv[b][4 * g + e]; }
            *(f32x4*)(ok + b * 32 + 8 * g + 4 * h) = v;
            *(f32x4*)(ov + b * 32 + 8 * g + 4 * h) = w;
        }
}

}  // namespace

// K-tiles per share of a shared reduction: about `target` workgroups in all, a multiple of eight shares (one XCD each), every
// share at least 8 K-tiles deep where the reduction is long enough
static int x3_share(int nkt, int tiles, int target) {
    int shares = std::max(1, std::min((target + tiles - 1) / tiles, std::max(1, nkt / 8)));
    shares = std::max(8, (shares + 4) / 8 * 8);
    return std::max(1, (nkt + shares - 1) / shares);
}

extern "C" int qst_gemm_nt_x3(const QstGemmArgs* a, int epi, void* stream) {
    if (!a || !a->A || !a->B || !a->C || a->M <= 0 || a->N <= 0 || a->K <= 0) return QST_ERR_BAD_ARG;
    if (a->K % XBK != 0 || a->lda % 4 != 0 || a->ldb % 4 != 0 || a->N % 4 != 0 || a->ldc % 4 != 0) return QST_ERR_UNSUPPORTED;
    const int tiles = ((a->M + 127) / 128) * ((a->N + 127) / 128);
    const int grid = (((a->M + 127) / 128 + 7) / 8 * 8) * ((a->N + 127) / 128);      // A panels padded to the eight XCDs
    hipStream_t st = (hipStream_t)stream;
    const size_t lds = 65536;                           // two ring slots of four 8 KB operand images (the epilogue's 34816 bytes fit)
    if (epi == 3) {
        // a few output tiles and a long reduction (dW = dY^T X): share K among enough workgroups to fill the chip, each
        // at least 8 K-tiles deep; partial tiles meet in C through fp32 atomics
        const int nkt = a->K / XBK;
        const int per = x3_share(nkt, tiles, 256);
        QstGemmArgs g = *a;
        g.splits = per;
        gemm_nt_x3_kernel<3><<<((nkt + per - 1) / per + 7) / 8 * 8 * tiles, 256, lds, st>>>(g);
        QST_LAUNCH_CHECK();
        return QST_OK;
    }
    switch (epi) {
        case 0: gemm_nt_x3_kernel<0><<<grid, 256, lds, st>>>(*a); break;
        case 1: gemm_nt_x3_kernel<1><<<grid, 256, lds, st>>>(*a); break;
        case 2: gemm_nt_x3_kernel<2><<<grid, 256, lds, st>>>(*a); break;
        case 4: if (!a->C2) return QST_ERR_BAD_ARG; gemm_nt_x3_kernel<4><<<grid, 256, lds, st>>>(*a); break;
        case 5: if (!a->aux) return QST_ERR_BAD_ARG; gemm_nt_x3_kernel<5><<<grid, 256, lds, st>>>(*a); break;
        default: return QST_ERR_BAD_ARG;
    }
    QST_LAUNCH_CHECK();
    return QST_OK;
}

// C[N, K] += A[M, N]^T . B[M, K], colsum[n] += sum_m A[m, n]  (the argument roles of qst_gemm_tn; fp32 operands, x3 products)
extern "C" int qst_gemm_tn_x3(const QstGemmArgs* a, void* stream) {
    if (!a || !a->A || !a->B || !a->C || a->M <= 0 || a->N <= 0 || a->K <= 0) return QST_ERR_BAD_ARG;
    if (a->M % XBK != 0 || a->lda % 4 != 0 || a->ldb % 4 != 0 || a->N % 4 != 0 || a->K % 4 != 0 || a->ldc % 4 != 0) return QST_ERR_UNSUPPORTED;
    if (((uintptr_t)a->A | (uintptr_t)a->B) & 15) return QST_ERR_UNSUPPORTED;
    QstGemmArgs g = *a;
    g.M = a->N; g.N = a->K; g.K = a->M;            // the kernel's names: C[M, N] over a reduction of K rows
    const int tiles = ((g.M + 127) / 128) * ((g.N + 127) / 128);
    const int nkt = g.K / XBK;
    const int per = x3_share(nkt, tiles, a->splits > 0 ? a->splits : 256);   // tools/x3_wgrad_sweep.py: 128 - 512 workgroups are the fastest
    // a share's row offsets are 32-bit and its buffer range is clamped to 0x7FFFFF00: a share spanning 2 GiB would read zeros
    // past the clamp and return silently wrong gradients -- step-sized shapes are two orders below it
    if ((int64_t)per * XBK * (int64_t)(a->lda > a->ldb ? a->lda : a->ldb) * 4 >= 0x7FFFFF00LL) return QST_ERR_UNSUPPORTED;
    g.splits = per;
    gemm_nt_x3_kernel<3, true><<<((nkt + per - 1) / per + 7) / 8 * 8 * tiles, 256, 65536, (hipStream_t)stream>>>(g);
    QST_LAUNCH_CHECK();
    return QST_OK;
}

extern "C" int qst_attention_fwd_x3(const float* qkv, const int64_t* mask, const float* rel_bias, int nseq, int L, int A,
                                    int d, float* ctx, void* stream) {
    return qst_attention_fwd_x3_drop(qkv, mask, rel_bias, nseq, L, A, d, ctx, nullptr, stream);
}
extern "C" int qst_attention_fwd_x3_drop(const float* qkv, const int64_t* mask, const float* rel_bias, int nseq, int L, int A,
                                         int d, float* ctx, const QstDrop* drop, void* stream) {
    if (!qkv || !mask || !ctx || nseq <= 0 || L <= 0 || A <= 0) return QST_ERR_BAD_ARG;
    if ((d != 32 && d != 64) || (L % 32) != 0 || L > 512) return QST_ERR_UNSUPPORTED;
    if (drop && drop->thr16 && drop->state && (drop->thr16 > 65535u || (int64_t)nseq * A * L * L >= ((int64_t)1 << 32)))
        return QST_ERR_UNSUPPORTED;
    AttnX3Args a{};
    if (drop && drop->thr16 && drop->state) a.drop = *drop;
    a.qkv = qkv; a.mask = mask; a.rel = rel_bias; a.out = ctx;
    a.nseq = nseq; a.L = L; a.A = A; a.H = A * d; a.scale = 1.0f / sqrtf((float)d);
    const int grid = nseq * A * ((L + 127) / 128);
    const size_t lds = (size_t)4 * 128 * d * 2 + (size_t)L * 4;
    hipStream_t st = (hipStream_t)stream;
    if (d == 32) {
        attn_fwd_x3_kernel<32><<<grid, 256, lds, st>>>(a);
    } else {
        static QstLdsAttr attr;
        if (int rc = qst_ensure_lds(attr, (const void*)attn_fwd_x3_kernel<64>, 70000)) return rc;
        attn_fwd_x3_kernel<64><<<grid, 256, lds, st>>>(a);
    }
    QST_LAUNCH_CHECK();
    return QST_OK;
}

extern "C" size_t qst_attention_bwd_x3_scratch_bytes(int nseq, int L, int A) { return (size_t)3 * nseq * A * L * sizeof(float); }
extern "C" int qst_attention_bwd_x3(const float* qkv, const float* ctx, const float* dctx, const int64_t* mask, const float* rel_bias,
                                    int nseq, int L, int A, int d, float* dqkv, float* drel_bias, void* scratch,
                                    const QstDrop* drop, void* stream) {
    if (!qkv || !ctx || !dctx || !mask || !dqkv || !scratch || nseq <= 0 || L <= 0 || A <= 0) return QST_ERR_BAD_ARG;
    if (drel_bias && !rel_bias) return QST_ERR_BAD_ARG;
    if ((d != 32 && d != 64) || (L % 32) != 0 || L > 512) return QST_ERR_UNSUPPORTED;
    const bool dropping = drop && drop->thr16 && drop->state;
    if (dropping && (drop->thr16 > 65535u || (int64_t)nseq * A * L * L >= ((int64_t)1 << 32))) return QST_ERR_UNSUPPORTED;
    AttnBwdX3Args a{};
    if (dropping) a.drop = *drop;
    a.qkv = qkv; a.ctx = ctx; a.dctx = dctx; a.mask = mask; a.rel = rel_bias; a.dqkv = dqkv; a.drel = drel_bias;
    a.stats = (float*)scratch;
    a.nseq = nseq; a.L = L; a.A = A; a.H = A * d; a.scale = 1.0f / sqrtf((float)d);
    const int grid = nseq * A * ((L + 127) / 128);
    const size_t lds_q = (size_t)6 * 128 * d * 2 + (size_t)L * 4, lds_k = (size_t)8 * 128 * d * 2 + (size_t)3 * L * 4;
    hipStream_t st = (hipStream_t)stream;
    if (d == 32) {
        static QstLdsAttr attr;
        if (int rc = qst_ensure_lds(attr, (const void*)attn_bwd_x3_dkv_kernel<32>, 8 * 128 * 32 * 2 + 3 * 512 * 4)) return rc;
        attn_bwd_x3_dq_kernel<32><<<grid, 256, lds_q, st>>>(a);
        attn_bwd_x3_dkv_kernel<32><<<grid, 256, lds_k, st>>>(a);
    } else {
        static QstLdsAttr attr_q, attr_k;
        if (int rc = qst_ensure_lds(attr_q, (const void*)attn_bwd_x3_dq_kernel<64>, 6 * 128 * 64 * 2 + 512 * 4)) return rc;
        if (int rc = qst_ensure_lds(attr_k, (const void*)attn_bwd_x3_dkv_kernel<64>, 8 * 128 * 64 * 2 + 3 * 512 * 4)) return rc;
        attn_bwd_x3_dq_kernel<64><<<grid, 256, lds_q, st>>>(a);
        attn_bwd_x3_dkv_kernel<64><<<grid, 256, lds_k, st>>>(a);
    }
    QST_LAUNCH_CHECK();
    return QST_OK;
}
