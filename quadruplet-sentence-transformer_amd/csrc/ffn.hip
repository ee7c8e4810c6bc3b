// ffn.hip -- the feed-forward block of a BERT layer as ONE kernel, forward and backward (gfx950, H = 384).
//
//   forward  (MODE 0): x = LayerNorm2( gelu(y1 . W1^T + b1) . W2^T + b2 + y1 )           modeling_bert.py:325-351
//   backward (MODE 1): du = (ds2 . W2) * gelu'(u) ; ds1 = LayerNorm1'( du . W1 + ds2 )    (its autograd)
//
// Both are the same chain  T = A . B1^T  ->  elementwise  ->  acc += T . B2^T  ->  full-row LayerNorm epilogue:
// a workgroup owns 128 token rows and walks the intermediate dimension in chunks of 192 columns; a chunk of T lives
// only in registers and in one LDS image (the A operand of the second product), so the [M, I] tensor between the two
// GEMMs is never re-read from HBM (forward: h; backward: du) and -- in inference -- never written either.
// Against the two-kernel form (gemm_nt_kernel<GELU> + gemm_nt_ln_kernel<0>, or <GELU_BWD> + <1>) that removes 100 MB
// of reads per call at M = 32768, one launch and one epilogue, and it moves the K loops from "first touch of a cold
// activation panel per tile" to "weights re-read from L2 by every workgroup in step": all 256 workgroups stream the
// same W1 / W2 stages at the same time, the only HBM operand is the 128 x 384 activation panel (re-read from L2 for
// chunks 1..7).
//
// Geometry: 8 waves as 4 (M) x 2 (N). Product 1: 128 x 192 tile, wave tile 32 x 96, K = 384 in six 64-deep stages
// (A 16 KB + B1 24 KB). Product 2: 128 x 384 tile, wave tile 32 x 192, K = 192 in three stages (B2 48 KB; A = the
// chunk image, 3 x [128][64] bf16). One 2-slot ring of 48 KB slots carries both kinds of stage: 72 stages per
// workgroup, stage s+1 in flight while stage s is computed, one barrier per stage (as gemm.hip's nt_mainloop).
// Training-time side outputs leave as follows: gelu'(u) straight from the accumulator registers (8-byte stores),
// h / du row-wise from the chunk image (16-byte stores) at the start of product 2; their completion is kept off the
// K loop's critical path with COUNTED vmcnt waits (stores and LDS-DMA share one in-order counter).
#include "qst_common.h"
#include "qst_kernels.h"

namespace {

typedef __attribute__((address_space(3))) void lds_void;
typedef __attribute__((ext_vector_type(2))) float f32x2;

__device__ __forceinline__ void dma16(__amdgpu_buffer_rsrc_t r, char* lds_wave_base, uint32_t voff, uint32_t soff) {
    __builtin_amdgcn_raw_ptr_buffer_load_lds(r, (lds_void*)lds_wave_base, 16, (int)voff, (int)soff, 0, 0);
}
// [rows][64 bf16] image (128-byte rows): 16-byte chunk c of row r at chunk position c ^ ((r >> 1) & 7) -- the layout
// of gemm.hip's NT stages (conflict-free ds_read_b128 fragment reads)
__device__ __forceinline__ uint32_t img_off(int row, int chunk) {
    return (uint32_t)(row * 128 + ((chunk ^ ((row >> 1) & 7)) << 4));
}
__device__ __forceinline__ int xcd_remap(int b, int nwg) {
    const int q = nwg >> 3, r = nwg & 7, x = b & 7;
    return (x < r ? x * (q + 1) : r * (q + 1) + (x - r) * q) + (b >> 3);
}
// wait for all but the N youngest vector-memory operations of this wave, and for its LDS accesses
template <int N> __device__ __forceinline__ void wait_vm_lds() {
    if (N == 0) asm volatile("s_waitcnt vmcnt(0) lgkmcnt(0)" ::: "memory");
    else if (N == 1) asm volatile("s_waitcnt vmcnt(1) lgkmcnt(0)" ::: "memory");
    else if (N == 7) asm volatile("s_waitcnt vmcnt(7) lgkmcnt(0)" ::: "memory");
    else if (N == 13) asm volatile("s_waitcnt vmcnt(13) lgkmcnt(0)" ::: "memory");
    else static_assert(N == 0 || N == 1 || N == 7 || N == 13, "unsupported count");
}
constexpr uint32_t kOOB = 0x7FFFFFF0u;     // voffset that fails every buffer range check: the lane moves nothing

constexpr int FH = 384;                    // model width = full output rows
constexpr int FBM = 128;                   // token rows per workgroup
constexpr int FIC = 192;                   // intermediate columns per chunk
constexpr int FBK = 64;                    // K depth of a stage
constexpr int F_SLOT = FH * FBK * 2;       // 48 KB: the larger stage (product 2: B2 [384][64])
constexpr int F_A1 = FBM * FBK * 2;        // 16 KB: product-1 A stage; B1 [192][64] (24 KB) follows it in the slot
constexpr int F_IMG = FBM * FBK * 2;       // 16 KB: one [128][64] image of the chunk
constexpr int F_HB = 2 * F_SLOT;           // chunk images at 96 KB (3 x 16 KB)
constexpr int F_VEC = F_HB + 3 * F_IMG;    // fp32 vectors at 144 KB: bias1 [I], then bias2 / gamma / beta [384]
constexpr int LN_LD = 388;                 // epilogue slab row stride in floats (1552 B: ds_write_b128 conflict-free)

template <int NB> struct Frag { op16x8 a; op16x8 b[NB]; };     // one k-step's MFMA operands of a wave (b[] statically indexed)

struct FfnArgs {
    const op16* A; const op16* B1; const op16* B2;
    const float* bias1; const float* bias2; const float* resid;
    const op16* aux; op16* save_gp; op16* save_h;
    float* C; op16* C2;
    int M, I;
    int diag;        // timing experiments (QstFfnArgs.diag): 1 = drop the A loads, 2 = drop the weight loads, 4 = L2 touch-ahead
};

template <int MODE, bool SAVE>
__global__ __launch_bounds__(512, 1) void ffn_chain_kernel(FfnArgs g, QstLnEpi e) {
    op_saturate(MODE == 0);
    extern __shared__ __attribute__((aligned(16))) char smem[];
    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int wm = wave >> 1, wn = wave & 1;
    const int fr = lane & 31, fh = lane >> 5;
    const int ntm = (g.M + FBM - 1) / FBM;
    const int m0 = xcd_remap(blockIdx.x, ntm) * FBM;
    const int rows_a = min(FBM, g.M - m0);
    const int I = g.I, NC = I / FIC, total = NC * 9;
    char* hbuf = smem + F_HB;
    float* bias1_s = (float*)(smem + F_VEC);
    float* vec_s = bias1_s + I;                               // [3][384]: bias2, gamma, beta

    const __amdgpu_buffer_rsrc_t ra = make_rsrc(g.A + (size_t)m0 * FH, (g.diag & 1) ? 0u : (uint32_t)rows_a * FH * 2u);
    const __amdgpu_buffer_rsrc_t rb1 = make_rsrc(g.B1, (g.diag & 2) ? 0u : (uint32_t)I * FH * 2u);
    const __amdgpu_buffer_rsrc_t rb2 = make_rsrc(g.B2, (g.diag & 2) ? 0u : (uint32_t)FH * I * 2u);
    const __amdgpu_buffer_rsrc_t rx = make_rsrc(MODE == 1 ? g.aux + (size_t)m0 * I : g.A, MODE == 1 ? (uint32_t)rows_a * I * 2u : 0u);
    // side outputs go through range-checked buffer stores: rows past M are dropped by the hardware while the store
    // INSTRUCTION is still issued by every wave -- the counted vmcnt waits below rely on fixed instruction counts
    const __amdgpu_buffer_rsrc_t rgp = make_rsrc((SAVE && MODE == 0) ? g.save_gp + (size_t)m0 * I : g.A,
                                                 (SAVE && MODE == 0) ? (uint32_t)rows_a * I * 2u : 0u);
    const __amdgpu_buffer_rsrc_t rsh = make_rsrc(SAVE ? g.save_h + (size_t)m0 * I : g.A, SAVE ? (uint32_t)rows_a * I * 2u : 0u);

    // per-lane source offsets of this wave's LDS-DMA instructions (8 rows x 128 B each; swizzle on the SOURCE chunk)
    const int r8 = lane >> 3, c8 = lane & 7;
    uint32_t va[2], vb1[3], vb2[6], vx[6];
#pragma unroll
    for (int t = 0; t < 2; ++t) {
        const int row = (wave * 2 + t) * 8 + r8;
        va[t] = (uint32_t)row * FH * 2u + (uint32_t)((c8 ^ ((row >> 1) & 7)) * 16);
    }
#pragma unroll
    for (int t = 0; t < 3; ++t) {
        const int row = (wave * 3 + t) * 8 + r8;
        vb1[t] = (uint32_t)row * FH * 2u + (uint32_t)((c8 ^ ((row >> 1) & 7)) * 16);
    }
#pragma unroll
    for (int t = 0; t < 6; ++t) {
        const int row = (wave * 6 + t) * 8 + r8;
        vb2[t] = (uint32_t)row * (uint32_t)I * 2u + (uint32_t)((c8 ^ ((row >> 1) & 7)) * 16);
        const int idx = wave * 6 + t, im = idx >> 4, rowx = (idx & 15) * 8 + r8;
        vx[t] = (uint32_t)rowx * (uint32_t)I * 2u + (uint32_t)(im * 128) + (uint32_t)((c8 ^ ((rowx >> 1) & 7)) * 16);
    }
    // This wave's LDS-DMA pieces [p0, p1) of stage s (9-per-chunk sequence) into slot s & 1. Product-1 stages have five
    // pieces per wave (2 of A, 3 of B1), product-2 stages six (B2). The pieces are dealt out BETWEEN the MFMA groups of
    // the running stage instead of as one burst after the barrier: all eight waves queueing 48 KB at once keep the CU's
    // load path busy for ~1200 cycles during which an in-order wave can do nothing else.
    // (Macros, not lambdas: a closure over the buffer descriptors ended up in scratch memory.)
#define issue_pieces(s_, p0_, p1_) do {                                                                                  \
        const int s__ = (s_);                                                                                            \
        if (s__ < total) {                                                                                               \
            char* st = smem + (s__ & 1) * F_SLOT;                                                                        \
            const int c__ = s__ / 9, r__ = s__ - 9 * c__;                                                                \
            if (r__ < 6) {                                                                                               \
                const uint32_t ko = (uint32_t)r__ * (FBK * 2);                                                           \
                const uint32_t bo = (uint32_t)c__ * (FIC * FH * 2) + ko;                                                 \
                _Pragma("unroll") for (int t = (p0_); t < (p1_); ++t) {                                                  \
                    if (t < 2) dma16(ra, st + (wave * 2 + t) * 1024, va[t], ko);                                         \
                    else if (t < 5) dma16(rb1, st + F_A1 + (wave * 3 + (t - 2)) * 1024, vb1[t - 2], bo);                 \
                }                                                                                                        \
            } else {                                                                                                     \
                const uint32_t bo = (uint32_t)(c__ * FIC + (r__ - 6) * FBK) * 2u;                                        \
                _Pragma("unroll") for (int t = (p0_); t < (p1_); ++t) dma16(rb2, st + (wave * 6 + t) * 1024, vb2[t], bo); \
            }                                                                                                            \
        }                                                                                                                \
    } while (0)
#define issue(s_) issue_pieces(s_, 0, 6)

    // L2 touch-ahead: one 4-byte LDS-DMA per lane into a junk area, each lane on a different 128-byte line of stage t
    // (product 1: 128 A rows on waves 0-1, 192 B1 rows on waves 2-4; product 2: 384 B2 rows on waves 0-5). With a 2-slot
    // ring a stage's loads are issued only one stage ahead, so a line that misses L2 costs its whole miss latency; the
    // touch, issued three stages ahead, has the line in L2 when the real load asks for it. Every wave issues exactly
    // one touch per stage (the counted waits depend on it); lanes with nothing to touch go out of range.
    char* junk = smem + F_VEC + (I + 3 * FH) * 4 + wave * 256;
    const int li = wave * 64 + lane;
    const bool touch_on = (g.diag & 4) != 0;
#define touch(t_) do {                                                                                                   \
        const int t__ = (t_);                                                                                            \
        uint32_t voff = kOOB;                                                                                            \
        int so = 0, which = 0;                                                                                           \
        if (touch_on && t__ < total) {                                                                                   \
            const int c__ = t__ / 9, r__ = t__ - 9 * c__;                                                                \
            if (r__ < 6) {                                                                                               \
                if (wave < 2) { which = 0; voff = (uint32_t)li * FH * 2u; so = r__ * 128; }                              \
                else if (wave < 5) { which = 1; voff = (uint32_t)(li - 128) * FH * 2u; so = c__ * (FIC * FH * 2) + r__ * 128; } \
            } else if (wave < 6) { which = 2; voff = (uint32_t)li * (uint32_t)I * 2u; so = (c__ * FIC + (r__ - 6) * FBK) * 2; } \
        }                                                                                                                \
        so = __builtin_amdgcn_readfirstlane(so);                                                                         \
        which = __builtin_amdgcn_readfirstlane(which);                                                                   \
        if (which == 0) __builtin_amdgcn_raw_ptr_buffer_load_lds(ra, (lds_void*)junk, 4, (int)voff, so, 0, 0);           \
        else if (which == 1) __builtin_amdgcn_raw_ptr_buffer_load_lds(rb1, (lds_void*)junk, 4, (int)voff, so, 0, 0);     \
        else __builtin_amdgcn_raw_ptr_buffer_load_lds(rb2, (lds_void*)junk, 4, (int)voff, so, 0, 0);                     \
    } while (0)

    for (int c = tid; c < I + 3 * FH; c += 512) {
        float v = 0.f;
        if (c < I) v = (MODE == 0 && g.bias1) ? g.bias1[c] : 0.f;
        else {
            const int which = (c - I) / FH, n = (c - I) - which * FH;
            if (which == 0) v = g.bias2 ? g.bias2[n] : 0.f;
            else if (which == 1) v = e.gamma[n];
            else v = (MODE == 0) ? e.beta[n] : 0.f;
        }
        bias1_s[c] = v;
    }

    f32x16 acc2[6];
#pragma unroll
    for (int j = 0; j < 6; ++j)
#pragma unroll
        for (int r = 0; r < 16; ++r) acc2[j][r] = 0.f;

    int s = 0;
    issue(0);
    touch(1);
#pragma unroll 1
    for (int c = 0; c < NC; ++c) {
        f32x16 acc1[3];
#pragma unroll
        for (int j = 0; j < 3; ++j)
#pragma unroll
            for (int r = 0; r < 16; ++r) acc1[j][r] = 0.f;
        // ---------------------------------------------------------------- product 1: T = A . B1[chunk]^T
#pragma unroll 1
        for (int kt = 0; kt < 6; ++kt, ++s) {
            // the youngest operation is always last stage's touch; the chunk's aux DMAs (issued last stage) may stay too
            if (MODE == 1 && kt == 1) wait_vm_lds<7>();
            else wait_vm_lds<1>();
            __builtin_amdgcn_s_barrier();
            const char* pa = smem + (s & 1) * F_SLOT;
            const char* pb = pa + F_A1;
            // Fragment reads run one k-step ahead of the MFMAs in their own registers, and the first reads of the stage
            // go out BEFORE the next stage's DMAs are issued: an in-order wave otherwise pays the LDS latency of every
            // k-step and the DMA issue cost (60-185 cycles each) in series with its MFMAs. The scheduling fences keep
            // the compiler from folding the groups back together.
            Frag<3> f0, f1;                                      // two named fragment sets (no runtime-indexed arrays)
            const int arow = wm * 32 + fr, brow = wn * 96 + fr;
#define load1(ks, f) do { (f).a = *(const op16x8*)(pa + img_off(arow, (ks) * 2 + fh));                                   \
                          (f).b[0] = *(const op16x8*)(pb + img_off(brow, (ks) * 2 + fh));                                \
                          (f).b[1] = *(const op16x8*)(pb + img_off(brow + 32, (ks) * 2 + fh));                           \
                          (f).b[2] = *(const op16x8*)(pb + img_off(brow + 64, (ks) * 2 + fh)); } while (0)
#define mma1(f) do { acc1[0] = mfma32_op((f).b[0], (f).a, acc1[0]);               \
                     acc1[1] = mfma32_op((f).b[1], (f).a, acc1[1]);               \
                     acc1[2] = mfma32_op((f).b[2], (f).a, acc1[2]); } while (0)   /* D rows = n, col = m */
#define FENCE __builtin_amdgcn_sched_barrier(0)
            load1(0, f0); FENCE;
            load1(1, f1); FENCE; mma1(f0); FENCE; issue_pieces(s + 1, 0, 2); FENCE;
            load1(2, f0); FENCE; mma1(f1); FENCE; issue_pieces(s + 1, 2, 4); FENCE;
            load1(3, f1); FENCE; mma1(f0); FENCE; issue_pieces(s + 1, 4, 6); FENCE;
            mma1(f1); FENCE;
            touch(s + 3);
            if (MODE == 1 && kt == 0) {
                // gelu'(u) of this chunk -> chunk images (free since the barrier above: product 2 of the previous chunk
                // is finished); landed and published by the wait + barrier that open stage kt = 2
                const uint32_t xo = (uint32_t)c * (FIC * 2);
#pragma unroll
                for (int t = 0; t < 6; ++t) dma16(rx, hbuf + (wave * 6 + t) * 1024, vx[t], xo);
            }
            FENCE;
        }
        // ---------------------------------------------------------------- elementwise: registers -> chunk image
        {
            const int ml = wm * 32 + fr;                         // row of the 128-row panel on this lane
#pragma unroll
            for (int j = 0; j < 3; ++j) {
                const int nl0 = wn * 96 + j * 32;                // first chunk column of this MFMA tile
                char* ib = hbuf + (nl0 >> 6) * F_IMG;
#pragma unroll
                for (int g4 = 0; g4 < 4; ++g4) {
                    const int nl = nl0 + 8 * g4 + 4 * fh;        // 4 consecutive chunk columns nl .. nl+3
                    char* p = ib + img_off(ml, ((nl0 & 63) >> 3) + g4) + 8 * fh;
                    u32x2 out;
                    if (MODE == 0) {
                        const f32x4 b4 = *(const f32x4*)(bias1_s + c * FIC + nl);
                        u32x2 pg;
#pragma unroll
                        for (int h2 = 0; h2 < 2; ++h2) {
                            qst_f32x2 x2, cdf, pdf;
                            x2[0] = acc1[j][4 * g4 + 2 * h2] + b4[2 * h2];
                            x2[1] = acc1[j][4 * g4 + 2 * h2 + 1] + b4[2 * h2 + 1];
                            gelu_parts2(x2, cdf, pdf);
                            const qst_f32x2 gg = x2 * pdf + cdf, hh = x2 * cdf;
                            pg[h2] = pack_op2(gg[0], gg[1]);
                            out[h2] = pack_op2(hh[0], hh[1]);
                        }
                        if (SAVE)
                            __builtin_amdgcn_raw_buffer_store_b64(pg, rgp, (int)(((uint32_t)ml * I + c * FIC + nl) * 2u), 0, 0);
                    } else {
                        const u32x2 ax = *(const u32x2*)p;       // gelu'(u) of the same four elements
#pragma unroll
                        for (int h2 = 0; h2 < 2; ++h2)
                            out[h2] = pack_op2(acc1[j][4 * g4 + 2 * h2] * op_lo(ax[h2]),
                                                  acc1[j][4 * g4 + 2 * h2 + 1] * op_hi(ax[h2]));
                    }
                    *(u32x2*)p = out;
                }
            }
        }
        // ---------------------------------------------------------------- product 2: acc2 += T . B2[:, chunk]^T
#pragma unroll 1
        for (int kt = 0; kt < 3; ++kt, ++s) {
            if (kt == 0) { if (MODE == 0 && SAVE) wait_vm_lds<13>(); else wait_vm_lds<1>(); }
            else if (kt == 1) { if (SAVE) wait_vm_lds<7>(); else wait_vm_lds<1>(); }
            else wait_vm_lds<1>();
            __builtin_amdgcn_s_barrier();
            const char* pb = smem + (s & 1) * F_SLOT;
            const char* ph = hbuf + kt * F_IMG;
            Frag<6> f0, f1;
            const int arow = wm * 32 + fr, brow = wn * 192 + fr;
#define load2(ks, f) do { (f).a = *(const op16x8*)(ph + img_off(arow, (ks) * 2 + fh));                                   \
                          _Pragma("unroll") for (int j = 0; j < 6; ++j)                                                  \
                              (f).b[j] = *(const op16x8*)(pb + img_off(brow + j * 32, (ks) * 2 + fh)); } while (0)
#define mma2(f) do { _Pragma("unroll") for (int j = 0; j < 6; ++j)                                                       \
                         acc2[j] = mfma32_op((f).b[j], (f).a, acc2[j]); } while (0)
            load2(0, f0); FENCE;
            load2(1, f1); FENCE; mma2(f0); FENCE; issue_pieces(s + 1, 0, 2); FENCE;
            load2(2, f0); FENCE; mma2(f1); FENCE; issue_pieces(s + 1, 2, 4); FENCE;
            load2(3, f1); FENCE; mma2(f0); FENCE; issue_pieces(s + 1, 4, 6); FENCE;
            mma2(f1); FENCE;
            touch(s + 3);
            if (SAVE && kt == 0) {
                // h (forward) / du (backward) rows of the chunk, 16 bytes per lane, 24 lanes per 384-byte row segment
#pragma unroll
                for (int i = 0; i < 6; ++i) {
                    const int idx = tid + 512 * i, row = idx / 24, ch = idx - 24 * row;
                    const u32x4 v = *(const u32x4*)(hbuf + (ch >> 3) * F_IMG + img_off(row, ch & 7));
                    buf_store16(rsh, ((uint32_t)row * I + c * FIC + ch * 8) * 2u, v);
                }
            }
            FENCE;
        }
    }
    wait_vm_lds<0>();
    __builtin_amdgcn_s_barrier();                                // every wave has left the ring and the chunk images

    // ---------------------------------------------------------------- full-row LayerNorm epilogue (cf. gemm_nt_ln_kernel)
    // Two passes of 64 rows: the waves of row blocks wm = 2p, 2p+1 stage their 32 x 192 sub-tiles into two [32][384]
    // fp32 slabs, then every wave takes 8 complete rows (6 columns per lane).
    float* slab0 = (float*)smem;
    f32x2 ag[3], ab[3];
#pragma unroll
    for (int t = 0; t < 3; ++t) { ag[t][0] = ag[t][1] = ab[t][0] = ab[t][1] = 0.f; }
    const float inv_n = 1.f / (float)FH;
    const int sw = wave >> 2, rw = (wave & 3) * 8;              // slab and first slab row this wave normalises
#pragma unroll 1
    for (int p = 0; p < 2; ++p) {
        f32x2 rv[8][3];
        uint32_t xv[8][3];
        float rs[8];
#pragma unroll
        for (int k = 0; k < 8; ++k) {
            const int m = m0 + p * 64 + sw * 32 + rw + k;
            const bool ok = m < g.M;
#pragma unroll
            for (int t = 0; t < 3; ++t) {
                const int col = 2 * (lane + 64 * t);
                rv[k][t][0] = rv[k][t][1] = 0.f;
                xv[k][t] = 0u;
                if (ok && g.resid) rv[k][t] = *(const f32x2*)(g.resid + (size_t)m * FH + col);
                if (MODE == 1 && ok) xv[k][t] = *(const uint32_t*)((const op16*)e.xhat + (size_t)m * FH + col);
            }
            rs[k] = (MODE == 1 && ok) ? e.rstd[m] : 0.f;
        }
        if (p > 0) __syncthreads();                              // pass 0's slab rows have been read
        if ((wm >> 1) == p) {
            float* slab = slab0 + (wm & 1) * (32 * LN_LD);
#pragma unroll
            for (int j = 0; j < 6; ++j)
#pragma unroll
                for (int g4 = 0; g4 < 4; ++g4) {
                    f32x4 v;
#pragma unroll
                    for (int q = 0; q < 4; ++q) v[q] = acc2[j][4 * g4 + q];
                    *(f32x4*)(slab + fr * LN_LD + wn * 192 + j * 32 + 8 * g4 + 4 * fh) = v;
                }
        }
        __syncthreads();
        const float* slab = slab0 + sw * (32 * LN_LD);
#pragma unroll
        for (int k = 0; k < 8; ++k) {
            const int row = rw + k;
            const int m = m0 + p * 64 + sw * 32 + row;
            f32x2 v[3];
#pragma unroll
            for (int t = 0; t < 3; ++t) v[t] = *(const f32x2*)(slab + row * LN_LD + 2 * (lane + 64 * t));
            if (MODE == 0) {
                float sum = 0.f;
#pragma unroll
                for (int t = 0; t < 3; ++t) {
                    const f32x2 b = *(const f32x2*)(vec_s + 2 * (lane + 64 * t));
                    v[t][0] = (v[t][0] + b[0]) + rv[k][t][0];
                    v[t][1] = (v[t][1] + b[1]) + rv[k][t][1];
                    sum += v[t][0] + v[t][1];
                }
                const float mean = wave_sum(sum) * inv_n;
                float q = 0.f;
#pragma unroll
                for (int t = 0; t < 3; ++t) {
                    const float a0 = v[t][0] - mean, a1 = v[t][1] - mean;
                    q += a0 * a0 + a1 * a1;
                }
                const float rstd = rsqrtf(wave_sum(q) * inv_n + e.eps);
                if (m < g.M) {
                    if (lane == 0 && e.rstd) e.rstd[m] = rstd;
#pragma unroll
                    for (int t = 0; t < 3; ++t) {
                        const int col = 2 * (lane + 64 * t);
                        const f32x2 ga = *(const f32x2*)(vec_s + FH + col), be = *(const f32x2*)(vec_s + 2 * FH + col);
                        const float h0 = (v[t][0] - mean) * rstd, h1 = (v[t][1] - mean) * rstd;
                        f32x2 o;
                        o[0] = h0 * ga[0] + be[0];
                        o[1] = h1 * ga[1] + be[1];
                        *(f32x2*)(g.C + (size_t)m * FH + col) = o;
                        if (g.C2) *(uint32_t*)(g.C2 + (size_t)m * FH + col) = pack_op2(o[0], o[1]);
                        if (e.xhat) *(uint32_t*)((op16*)e.xhat + (size_t)m * FH + col) = pack_op2(h0, h1);
                    }
                }
            } else {
                float s1 = 0.f, s2 = 0.f;
                f32x2 x[3];
#pragma unroll
                for (int t = 0; t < 3; ++t) {
                    const f32x2 ga = *(const f32x2*)(vec_s + FH + 2 * (lane + 64 * t));
                    x[t][0] = op_lo(xv[k][t]); x[t][1] = op_hi(xv[k][t]);
                    v[t][0] += rv[k][t][0];
                    v[t][1] += rv[k][t][1];
                    ag[t][0] += v[t][0] * x[t][0]; ag[t][1] += v[t][1] * x[t][1];
                    ab[t][0] += v[t][0];           ab[t][1] += v[t][1];
                    v[t][0] *= ga[0]; v[t][1] *= ga[1];
                    s1 += v[t][0] + v[t][1];
                    s2 += v[t][0] * x[t][0] + v[t][1] * x[t][1];
                }
                const float m1 = wave_sum(s1) * inv_n, m2 = wave_sum(s2) * inv_n;
                if (m < g.M) {
#pragma unroll
                    for (int t = 0; t < 3; ++t) {
                        const int col = 2 * (lane + 64 * t);
                        f32x2 o;
                        o[0] = rs[k] * (v[t][0] - m1 - x[t][0] * m2);
                        o[1] = rs[k] * (v[t][1] - m1 - x[t][1] * m2);
                        *(f32x2*)(g.C + (size_t)m * FH + col) = o;
                        if (g.C2) *(uint32_t*)(g.C2 + (size_t)m * FH + col) = pack_op2(o[0], o[1]);
                    }
                }
            }
        }
    }
    if (MODE == 1 && e.partials) {
        __syncthreads();
        float* red = (float*)smem;                               // [8 waves][2][384]
#pragma unroll
        for (int t = 0; t < 3; ++t) {
            *(f32x2*)(red + (wave * 2 + 0) * FH + 2 * (lane + 64 * t)) = ag[t];
            *(f32x2*)(red + (wave * 2 + 1) * FH + 2 * (lane + 64 * t)) = ab[t];
        }
        __syncthreads();
        for (int c = tid; c < 2 * FH; c += 512) {
            const int which = c / FH, n = c - which * FH;
            float a = 0.f;
#pragma unroll
            for (int w = 0; w < 8; ++w) a += red[(w * 2 + which) * FH + n];
            e.partials[(size_t)blockIdx.x * 2 * FH + c] = a;
        }
    }
}

}  // namespace

#if !QST_OP_F16
extern "C" int qst_ffn_chain_supported(int H, int I) { return (H == FH && I > 0 && I % FIC == 0 && I <= 4096) ? 1 : 0; }
#endif

extern "C" int QST_K(qst_ffn_chain)(const QstFfnArgs* a, const QstLnEpi* ln, int mode, void* stream) {
    if (!a || !ln || !a->A || !a->B1 || !a->B2 || !a->C || !ln->gamma || a->M <= 0) return QST_ERR_BAD_ARG;
    if (mode != 0 && mode != 1) return QST_ERR_BAD_ARG;
    if (!qst_ffn_chain_supported(a->H, a->I)) return QST_ERR_UNSUPPORTED;
    if (mode == 0 && (!ln->beta || ((a->save_gp == nullptr) != (a->save_h == nullptr)))) return QST_ERR_BAD_ARG;
    if (mode == 1 && (!ln->xhat || !ln->rstd || !a->aux || !a->save_h)) return QST_ERR_BAD_ARG;
    if ((int64_t)128 * a->I * 2 >= 0x7FFFFF00LL) return QST_ERR_UNSUPPORTED;
    FfnArgs g{};
    g.A = (const op16*)a->A; g.B1 = (const op16*)a->B1; g.B2 = (const op16*)a->B2;
    g.bias1 = a->bias1; g.bias2 = a->bias2; g.resid = a->resid;
    g.aux = (const op16*)a->aux; g.save_gp = (op16*)a->save_gp; g.save_h = (op16*)a->save_h;
    g.C = a->C; g.C2 = (op16*)a->C2; g.M = a->M; g.I = a->I; g.diag = (a->diag & 3) | ((a->diag & 4) ? 0 : 4);
    const int lds = F_VEC + (a->I + 3 * FH) * 4 + 8 * 256;
    if (lds > 160 * 1024) return QST_ERR_UNSUPPORTED;
    const int ntm = (a->M + FBM - 1) / FBM;
    hipStream_t st = (hipStream_t)stream;
    static QstLdsAttr at0, at1, at2;
    if (mode == 0 && a->save_h) {
        if (int rc = qst_ensure_lds(at0, (const void*)ffn_chain_kernel<0, true>, 160 * 1024)) return rc;
        ffn_chain_kernel<0, true><<<dim3(ntm), dim3(512), lds, st>>>(g, *ln);
    } else if (mode == 0) {
        if (int rc = qst_ensure_lds(at1, (const void*)ffn_chain_kernel<0, false>, 160 * 1024)) return rc;
        ffn_chain_kernel<0, false><<<dim3(ntm), dim3(512), lds, st>>>(g, *ln);
    } else {
        if (int rc = qst_ensure_lds(at2, (const void*)ffn_chain_kernel<1, true>, 160 * 1024)) return rc;
        ffn_chain_kernel<1, true><<<dim3(ntm), dim3(512), lds, st>>>(g, *ln);
    }
    QST_LAUNCH_CHECK();
    return QST_OK;
}
