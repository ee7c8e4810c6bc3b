// qst_common.h -- shared device helpers for the gfx950 kernels (wave64, MFMA, buffer loads).
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>
#include "../../include/qst.h"
#include "../../include/qst_kernels.h"

typedef __bf16 bf16;
typedef __attribute__((ext_vector_type(2))) __bf16 bf16x2;
typedef __attribute__((ext_vector_type(4))) __bf16 bf16x4;
typedef __attribute__((ext_vector_type(8))) __bf16 bf16x8;
typedef __attribute__((ext_vector_type(4))) float f32x4;
typedef __attribute__((ext_vector_type(16))) float f32x16;
typedef __attribute__((ext_vector_type(4))) uint32_t u32x4;
typedef __attribute__((ext_vector_type(2))) uint32_t u32x2;

#define QST_WAVE 64

// ---------------------------------------------------------------- the 16-bit matrix-core operand type of a translation unit
// gemm.hip, gemm8.hip (gemm8p.h), ffn.hip, attention.hip and rowops.hip are written once on `op16` and compiled twice (csrc/Makefile):
//   QST_OP_F16 = 0 (default): op16 = bfloat16 -- QST_PREC_BF16, v_mfma_f32_32x32x16_bf16 / 16x16x32_bf16; entry points qst_*.
//   QST_OP_F16 = 1: op16 = IEEE half -- QST_PREC_F16, v_mfma_f32_32x32x16_f16 / 16x16x32_f16 (same issue rate, same bytes);
//                   the same entry points with the suffix _f16 (QST_K). Conversions round to nearest even (v_cvt_pk_f16_f32);
//                   forward kernels saturate at +-65,504 instead of overflowing to inf (op_saturate), backward kernels keep IEEE
//                   inf so that an overflowed gradient reaches the global norm, where the loss scaler looks for it (optim.hip).
// LDS-DMA staging, swizzles and transposing LDS reads move 16-bit lanes and do not care which of the two it is.
#ifndef QST_OP_F16
#define QST_OP_F16 0
#endif
#if QST_OP_F16
typedef _Float16 op16;
#define QST_K(name) name##_f16
#else
typedef __bf16 op16;
#define QST_K(name) name
#endif
typedef __attribute__((ext_vector_type(2))) op16 op16x2;
typedef __attribute__((ext_vector_type(4))) op16 op16x4;
typedef __attribute__((ext_vector_type(8))) op16 op16x8;

extern "C" int qst_set_hip_error(int code);

#define QST_HIP_CHECK(expr)                                 \
    do {                                                    \
        hipError_t _e = (expr);                             \
        if (_e != hipSuccess) {                             \
            qst_set_hip_error((int)_e);                     \
            return QST_ERR_HIP;                             \
        }                                                   \
    } while (0)

#define QST_LAUNCH_CHECK() QST_HIP_CHECK(hipGetLastError())

// hipFuncAttributeMaxDynamicSharedMemorySize is a property of (kernel, device): one of these per kernel remembers the
// devices it has been raised on (bit = device ordinal), so a second encoder on another GPU of the same process is
// covered and concurrent callers at worst both set the same value.
#ifdef __cplusplus
#include <atomic>
struct QstLdsAttr { std::atomic<uint64_t> done{0}; };
static inline int qst_ensure_lds(QstLdsAttr& st, const void* fn, int bytes) {
    int dev = 0;
    QST_HIP_CHECK(hipGetDevice(&dev));
    const uint64_t bit = 1ull << (dev & 63);
    if (st.done.load(std::memory_order_acquire) & bit) return QST_OK;
    QST_HIP_CHECK(hipFuncSetAttribute(fn, hipFuncAttributeMaxDynamicSharedMemorySize, bytes));
    st.done.fetch_or(bit, std::memory_order_release);
    return QST_OK;
}
#endif

// MX scale exponent of a block with largest magnitude amax: the smallest e with amax * 2^-e <= 448 (e4m3's largest
// value), from the float's own exponent and mantissa -- integer arithmetic, so the oracle reproduces it bit for bit.
// (The OCP recipe floor(log2 amax) - 8 lets elements in (448, 512) saturate; this one never saturates.) amax = 0 -> -127.
__device__ __forceinline__ int mx_exponent(float amax) {
    const uint32_t u = __builtin_bit_cast(uint32_t, amax);
    if ((u & 0x7F800000u) == 0u) return -127;                       // zero (or a float denormal: quantises to zero)
    int e = (int)((u >> 23) & 0xFF) - 127 - 8 + ((u & 0x7FFFFFu) > 0x600000u ? 1 : 0);
    return e < -127 ? -127 : (e > 126 ? 126 : e);
}
__device__ __forceinline__ float pow2f(int e) { return __builtin_bit_cast(float, (uint32_t)(e + 127) << 23); }   // -126 <= e <= 127

// Wave64 all-reduce on the VALU cross-lane path (DPP + readlane) instead of __shfl_xor, which lowers to
// ds_bpermute: six dependent LDS round trips per reduction made the row kernels latency-bound.
// Butterfly inside each row of 16 lanes (quad_perm xor1, xor2, row_half_mirror, row_mirror: sums are symmetric,
// so mirrors work as butterflies), then the four row totals are combined through SGPRs.
template <int CTRL> __device__ __forceinline__ float dpp_mov(float v) {
    return __builtin_bit_cast(float, __builtin_amdgcn_mov_dpp(__builtin_bit_cast(int, v), CTRL, 0xf, 0xf, true));
}
__device__ __forceinline__ float row16_sum(float v) {
    v += dpp_mov<0xB1>(v);     // quad_perm [1,0,3,2]
    v += dpp_mov<0x4E>(v);     // quad_perm [2,3,0,1]
    v += dpp_mov<0x141>(v);    // row_half_mirror
    v += dpp_mov<0x140>(v);    // row_mirror
    return v;
}
__device__ __forceinline__ float row16_max(float v) {
    v = fmaxf(v, dpp_mov<0xB1>(v));
    v = fmaxf(v, dpp_mov<0x4E>(v));
    v = fmaxf(v, dpp_mov<0x141>(v));
    v = fmaxf(v, dpp_mov<0x140>(v));
    return v;
}
__device__ __forceinline__ float lane_bcast(float v, int lane) {
    return __builtin_bit_cast(float, __builtin_amdgcn_readlane(__builtin_bit_cast(int, v), lane));
}
__device__ __forceinline__ float wave_sum(float v) {
    v = row16_sum(v);
    return (lane_bcast(v, 0) + lane_bcast(v, 16)) + (lane_bcast(v, 32) + lane_bcast(v, 48));
}
__device__ __forceinline__ float wave_max(float v) {
    v = row16_max(v);
    return fmaxf(fmaxf(lane_bcast(v, 0), lane_bcast(v, 16)), fmaxf(lane_bcast(v, 32), lane_bcast(v, 48)));
}
// exchange with the lane 32 away (the other half of an MFMA 32x32 accumulator column)
__device__ __forceinline__ float swap32(float v) {
    // v_permlane32_swap: lanes 32-63 of the first operand trade places with lanes 0-31 of the second (VALU, no LDS)
    const unsigned u = __builtin_bit_cast(unsigned, v);
    const auto r = __builtin_amdgcn_permlane32_swap(u, u, false, false);
    return __builtin_bit_cast(float, (threadIdx.x & 32) ? r[0] : r[1]);
}

__device__ __forceinline__ float bf2f(bf16 x) { return (float)x; }
__device__ __forceinline__ bf16 f2bf(float x) { return (bf16)x; }

// 16-byte buffer load with hardware range check: bytes past `num_bytes` read as zero.
__device__ __forceinline__ __amdgpu_buffer_rsrc_t make_rsrc(const void* base, uint32_t num_bytes) {
    return __builtin_amdgcn_make_buffer_rsrc(const_cast<void*>(base), 0, (int)num_bytes, 0x00020000);
}
__device__ __forceinline__ u32x4 buf_load16(__amdgpu_buffer_rsrc_t r, uint32_t byte_off) {
    return __builtin_amdgcn_raw_buffer_load_b128(r, (int)byte_off, 0, 0);
}
__device__ __forceinline__ void buf_store16(__amdgpu_buffer_rsrc_t r, uint32_t byte_off, u32x4 v) {
    __builtin_amdgcn_raw_buffer_store_b128(v, r, (int)byte_off, 0, 0);
}

// Store of output bytes that THIS kernel never reads again (epilogue outputs, saved activations, gradients): marked
// non-temporal. Measured on the whole training step, same box, alternating builds (DESIGN.md finding 21): MiniLM
// 4.91 -> 4.70 ms with the stores, 4.64 with the once-read loads below as well; per kernel the LayerNorm-fused GEMMs gain
// 8-15% (their consumers' operand rows arrive without the producer's dirty lines still queued behind them), the others
// move by a few percent either way. ONLY for stores in which one wave-instruction covers whole 64-byte sectors: the
// attention dQ / dK,dV kernels write 8 bytes per lane on 32 different rows per instruction and became 1.6-1.8x slower
// with the hint (partial sectors leave L2 one by one), so they keep plain stores; the row kernels of rowops.hip lost
// 4-12% and keep plain accesses too.
#ifndef QST_STREAM_STORES
#define QST_STREAM_STORES 1
#endif
template <typename T>
__device__ __forceinline__ void st_stream(T* p, const T& v) {
#if QST_STREAM_STORES
    __builtin_nontemporal_store(v, p);
#else
    *p = v;
#endif
}


// ... and the matching load for bytes a launch reads exactly once (residual rows, saved gelu'(u), normalised rows)
#ifndef QST_STREAM_LOADS
#define QST_STREAM_LOADS QST_STREAM_STORES
#endif
template <typename T>
__device__ __forceinline__ T ld_stream(const T* p) {
#if QST_STREAM_LOADS
    return __builtin_nontemporal_load(p);
#else
    return *p;
#endif
}

// (static: these differ between the two builds of a file)
static __device__ __forceinline__ float op2f(op16 x) { return (float)x; }
static __device__ __forceinline__ op16 f2op(float x) { return (op16)x; }
static __device__ __forceinline__ uint32_t pack_op2(float lo, float hi) {          // v_cvt_pk_bf16_f32 / v_cvt_pk_f16_f32: RNE
    op16x2 v;
    v[0] = (op16)lo;
    v[1] = (op16)hi;
    return __builtin_bit_cast(uint32_t, v);
}
#if QST_OP_F16
static __device__ __forceinline__ float op_lo(uint32_t u) { return (float)__builtin_bit_cast(op16x2, u)[0]; }
static __device__ __forceinline__ float op_hi(uint32_t u) { return (float)__builtin_bit_cast(op16x2, u)[1]; }    // v_cvt_f32_f16 sdwa WORD_1
// MODE.FP16_OVFL (hwreg 1, bit 23): an overflowed f16 VALU result is clamped to +-MAX_F16 instead of becoming inf (true
// infinities pass). Set at the top of a FORWARD kernel; backward kernels leave it clear.
static __device__ __forceinline__ void op_saturate(bool on) { if (on) __builtin_amdgcn_s_setreg(1 | (23 << 6), 1); }
#else
static __device__ __forceinline__ float op_lo(uint32_t u) { return __builtin_bit_cast(float, u << 16); }
static __device__ __forceinline__ float op_hi(uint32_t u) { return __builtin_bit_cast(float, u & 0xFFFF0000u); }
static __device__ __forceinline__ void op_saturate(bool) {}                        // bf16 has fp32's exponent range
#endif
static __device__ __forceinline__ f32x16 mfma32_op(op16x8 a, op16x8 b, f32x16 c) {
#if QST_OP_F16
    return __builtin_amdgcn_mfma_f32_32x32x16_f16(a, b, c, 0, 0, 0);
#else
    return __builtin_amdgcn_mfma_f32_32x32x16_bf16(a, b, c, 0, 0, 0);
#endif
}
static __device__ __forceinline__ f32x4 mfma16_op(op16x8 a, op16x8 b, f32x4 c) {
#if QST_OP_F16
    return __builtin_amdgcn_mfma_f32_16x16x32_f16(a, b, c, 0, 0, 0);
#else
    return __builtin_amdgcn_mfma_f32_16x16x32_bf16(a, b, c, 0, 0, 0);
#endif
}
// ds_read_b64_tr_b16: a transposing LDS read of four 16-bit values (type-agnostic in hardware; the builtin is typed)
static __device__ __forceinline__ op16x4 lds_tr16_op(const char* p) {
    typedef __attribute__((ext_vector_type(4))) __bf16 raw4;
    return __builtin_bit_cast(op16x4, __builtin_amdgcn_ds_read_tr16_b64_v4bf16((__attribute__((address_space(3))) raw4*)(p)));
}

__device__ __forceinline__ uint32_t pack_bf16x2(float lo, float hi) {
    bf16x2 v;
    v[0] = (bf16)lo;
    v[1] = (bf16)hi;
    return __builtin_bit_cast(uint32_t, v);
}
__device__ __forceinline__ float bf16lo(uint32_t u) { return __builtin_bit_cast(float, u << 16); }
__device__ __forceinline__ float bf16hi(uint32_t u) { return __builtin_bit_cast(float, u & 0xFFFF0000u); }

typedef __attribute__((ext_vector_type(2))) float qst_f32x2;
// Two elements at a time, written on 2-vectors so that the compiler emits v_pk_fma_f32 / v_pk_mul_f32 /
// v_pk_add_f32 (one issue slot per pair): the GELU epilogue of FFN-1 is VALU-bound (the whole [M, I] tensor goes
// through ~22 VALU issue slots per element in the scalar form, more SIMD time than the GEMM's MFMAs).
__device__ __forceinline__ void gelu_parts2(qst_f32x2 x, qst_f32x2& cdf, qst_f32x2& pdf) {
    qst_f32x2 ax;
    ax[0] = fabsf(x[0]); ax[1] = fabsf(x[1]);
    ax = ax * 0.70710678118654752f;
    const qst_f32x2 den = ax * 0.3275911f + 1.0f;
    qst_f32x2 t, e;
    t[0] = __builtin_amdgcn_rcpf(den[0]); t[1] = __builtin_amdgcn_rcpf(den[1]);
    const qst_f32x2 arg = (ax * ax) * -1.4426950408889634f;          // exp(-ax^2) = exp2(-ax^2 log2 e)
    e[0] = __builtin_amdgcn_exp2f(arg[0]); e[1] = __builtin_amdgcn_exp2f(arg[1]);
    qst_f32x2 poly = t * 1.061405429f + -1.453152027f;
    poly = poly * t + 1.421413741f;
    poly = poly * t + -0.284496736f;
    poly = poly * t + 0.254829592f;
    poly = poly * t;
    const qst_f32x2 half_erf = (poly * e) * -0.5f + 0.5f;              // 0.5 * erf(|x|/sqrt2)
    qst_f32x2 sh;
    sh[0] = __builtin_copysignf(half_erf[0], x[0]); sh[1] = __builtin_copysignf(half_erf[1], x[1]);
    cdf = sh + 0.5f;
    pdf = e * 0.39894228040143268f;
}

// erf GELU (HF "gelu": modeling_bert.py:325-337 via ACT2FN) and its derivative.
// erf by Abramowitz-Stegun 7.1.26 (|abs error| <= 1.5e-7, i.e. fp32-level) on v_exp_f32 / v_rcp_f32: ~14 VALU ops
// instead of libm erff's ~50, which made the GELU epilogue cost more than the GEMM's MFMAs.
// Both functions share e = exp(-x^2/2): erf(x/sqrt2) needs exp(-(x/sqrt2)^2), the pdf needs the same value.
__device__ __forceinline__ void gelu_parts(float x, float& cdf, float& pdf) {
    const float ax = fabsf(x) * 0.70710678118654752f;
    const float t = __builtin_amdgcn_rcpf(1.0f + 0.3275911f * ax);
    const float e = __expf(-ax * ax);
    const float poly = t * (0.254829592f + t * (-0.284496736f + t * (1.421413741f + t * (-1.453152027f + t * 1.061405429f))));
    const float erf_abs = 1.0f - poly * e;                 // erf(|x|/sqrt2)
    const float erf_v = x < 0.f ? -erf_abs : erf_abs;
    cdf = 0.5f * (1.0f + erf_v);
    pdf = 0.39894228040143268f * e;
}
__device__ __forceinline__ float gelu_erf(float x) {
    float cdf, pdf;
    gelu_parts(x, cdf, pdf);
    return x * cdf;
}
__device__ __forceinline__ float gelu_erf_grad(float x) {
    float cdf, pdf;
    gelu_parts(x, cdf, pdf);
    return cdf + x * pdf;
}

static inline int64_t qst_align_up(int64_t v, int64_t a) { return (v + a - 1) / a * a; }

// ---------------------------------------------------------------- dropout masks (include/qst_kernels.h: QstDrop)
__device__ __forceinline__ uint32_t qst_hash32(uint32_t x) {
    x ^= x >> 16; x *= 0x7feb352du; x ^= x >> 15; x *= 0x846ca68bu; x ^= x >> 16;
    return x;
}
struct DropCtx { uint32_t key, thr; float scale; };          // thr == 0: no dropout
__device__ __forceinline__ DropCtx drop_ctx(const QstDrop& d) {
    DropCtx c;
    c.thr = d.state ? d.thr16 : 0u; c.key = 0u; c.scale = 1.f;
    if (c.thr) {
        const uint32_t s0 = d.state[0], s1 = d.state[1], step = d.state[2];           // uniform: scalar loads
        c.key = qst_hash32(s0 ^ qst_hash32(step * 0x9E3779B9u + d.site) ^ ((s1 << 16) | (s1 >> 16)));
        c.scale = 65536.0f / (float)(65536u - c.thr);
    }
    return c;
}
// the 2 x 16 random bits of elements idx_even and idx_even + 1
__device__ __forceinline__ uint32_t drop_bits(const DropCtx& c, uint32_t idx_even) { return qst_hash32((idx_even >> 1) ^ c.key); }
__device__ __forceinline__ bool drop_keep_lo(const DropCtx& c, uint32_t bits) { return (bits & 0xFFFFu) >= c.thr; }
__device__ __forceinline__ bool drop_keep_hi(const DropCtx& c, uint32_t bits) { return (bits >> 16) >= c.thr; }
// multipliers (0 or scale) of elements idx_even and idx_even + 1
__device__ __forceinline__ void drop_pair(const DropCtx& c, uint32_t idx_even, float& m0, float& m1) {
    const uint32_t b = drop_bits(c, idx_even);
    m0 = drop_keep_lo(c, b) ? c.scale : 0.f;
    m1 = drop_keep_hi(c, b) ? c.scale : 0.f;
}

// Attention probabilities ([nseq, A, L, L], the largest masked tensor by far, regenerated in three kernels): 8 random bits
// per element, FOUR elements (keys 4t .. 4t+3 of one query) per hash word -- byte b of hash32((idx >> 2) ^ key) belongs to
// element (idx & ~3) + b, dropped when byte < thr8 = (thr16 + 128) >> 8, kept ones scaled by 256 / (256 - thr8). So the
// rate is quantised to 1/256 (p = 0.1 -> 26/256 = 0.1016, the scale matching it exactly); a quarter of the hashing.
__device__ __forceinline__ DropCtx drop_ctx8(const QstDrop& d) {
    DropCtx c = drop_ctx(d);
    if (c.thr) {
        c.thr = min(255u, (c.thr + 128u) >> 8);
        c.scale = 256.0f / (float)(256u - c.thr);
    }
    return c;
}
__device__ __forceinline__ uint32_t drop_word4(const DropCtx& c, uint32_t idx4) { return qst_hash32((idx4 >> 2) ^ c.key); }
template <int BYTE> __device__ __forceinline__ bool drop_keep_byte(const DropCtx& c, uint32_t w) {
    return ((w >> (8 * BYTE)) & 0xFFu) >= c.thr;
}
// value of `v` in lane (lane & ~3) + SRC of this lane's quad
template <int SRC> __device__ __forceinline__ uint32_t quad_bcast(uint32_t v) {
    return (uint32_t)__builtin_amdgcn_mov_dpp((int)v, SRC | (SRC << 2) | (SRC << 4) | (SRC << 6), 0xf, 0xf, true);
}
