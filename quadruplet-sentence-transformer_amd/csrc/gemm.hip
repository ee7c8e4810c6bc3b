// gemm.hip -- bf16 MFMA GEMMs for the encoder (gfx950, wave64, v_mfma_f32_32x32x16_bf16).
//
//   gemm_nt : C[M,N] = A[M,K] . B[N,K]^T  (+ fused epilogue)   forward Linear and dgrad (B = W^T shadow)
//   gemm_tn : C[N,K] += A[M,N]^T . B[M,K]  (fp32 atomics, split over M)  wgrad; optional column sums (bias grad)
//
// These replace nn.Linear forward/backward inside BertLayer (transformers modeling_bert.py:154-156,
// 282-293, 325-351). Tiles: 128x128 output per 256-thread workgroup (4 waves,
// 2x2, 64x64 per wave = 2x2 MFMA 32x32 tiles), K-step 64, LDS double-buffered through registers
// (16-byte buffer loads with hardware range check -> ragged M/N need no branches in the main loop).
// LDS images are XOR-swizzled so ds_read_b128 (nt) and ds_read_b64_tr_b16 (tn) are conflict-free.
#include "qst_common.h"
#include "qst_kernels.h"

namespace {

constexpr int BM = 128, BN = 128, BK = 64;

__device__ __forceinline__ int xcd_remap(int b, int nwg) {
    // bijective XCD-contiguous remap (blocks b and b+8 share an XCD): neighbours in the remapped id
    // share an A row-panel in the same L2
    const int q = nwg >> 3, r = nwg & 7, x = b & 7;
    return (x < r ? x * (q + 1) : r * (q + 1) + (x - r) * q) + (b >> 3);
}

// ---------------------------------------------------------------- NT
// LDS image per operand tile: [128 rows][64 bf16] = 128-byte rows, 16-byte chunk c of row r stored at
// chunk (c ^ ((r >> 1) & 7)).
__device__ __forceinline__ uint32_t nt_off(int row, int chunk) {
    return (uint32_t)(row * 128 + ((chunk ^ ((row >> 1) & 7)) << 4));
}

template <int EPI>
__global__ __launch_bounds__(256, 2) void gemm_nt_kernel(QstGemmArgs g) {
    extern __shared__ __attribute__((aligned(16))) char smem[];   // 2 x (16K A + 16K B); reused by the epilogue (4 x 17K)
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int wm = wave >> 1, wn = wave & 1;
    const int ntn = (g.N + BN - 1) / BN, ntm = (g.M + BM - 1) / BM;
    const int wg = xcd_remap(blockIdx.x, ntm * ntn);
    const int m0 = (wg / ntn) * BM, n0 = (wg % ntn) * BN;

    const int rows_a = min(BM, g.M - m0), rows_b = min(BN, g.N - n0);
    const bf16* Ab = (const bf16*)g.A + (size_t)m0 * g.lda;
    const bf16* Bb = (const bf16*)g.B + (size_t)n0 * g.ldb;
    const __amdgpu_buffer_rsrc_t ra = make_rsrc(Ab, (uint32_t)rows_a * g.lda * 2u);
    const __amdgpu_buffer_rsrc_t rb = make_rsrc(Bb, (uint32_t)rows_b * g.ldb * 2u);

    // staging map: 4 chunks of A and 4 of B per thread; row = tid/8 + 32*i, chunk = tid%8
    const int srow = tid >> 3, sch = tid & 7;
    uint32_t goff_a[4], goff_b[4], loff[4];
#pragma unroll
    for (int i = 0; i < 4; ++i) {
        const int r = srow + 32 * i;
        goff_a[i] = (uint32_t)r * g.lda * 2u + sch * 16u;
        goff_b[i] = (uint32_t)r * g.ldb * 2u + sch * 16u;
        loff[i] = nt_off(r, sch);
    }
    u32x4 sa[4], sb[4];
    auto gload = [&](int kt) {
        const uint32_t kb = (uint32_t)kt * BK * 2u;
#pragma unroll
        for (int i = 0; i < 4; ++i) {
            sa[i] = buf_load16(ra, goff_a[i] + kb);
            sb[i] = buf_load16(rb, goff_b[i] + kb);
        }
    };
    auto lstore = [&](int buf) {
        char* pa = smem + buf * 32768;
        char* pb = pa + 16384;
#pragma unroll
        for (int i = 0; i < 4; ++i) {
            *(u32x4*)(pa + loff[i]) = sa[i];
            *(u32x4*)(pb + loff[i]) = sb[i];
        }
    };

    f32x16 acc[2][2];
#pragma unroll
    for (int i = 0; i < 2; ++i)
#pragma unroll
        for (int j = 0; j < 2; ++j)
#pragma unroll
            for (int r = 0; r < 16; ++r) acc[i][j][r] = 0.f;

    const int nk = g.K / BK;
    gload(0);
    lstore(0);
    __syncthreads();
    const int fr = lane & 31, fh = lane >> 5;
    for (int kt = 0; kt < nk; ++kt) {
        const int cur = kt & 1;
        if (kt + 1 < nk) gload(kt + 1);
        const char* pa = smem + cur * 32768;
        const char* pb = pa + 16384;
#pragma unroll
        for (int ks = 0; ks < 4; ++ks) {
            bf16x8 fa[2], fb[2];
#pragma unroll
            for (int i = 0; i < 2; ++i) {
                fa[i] = *(const bf16x8*)(pa + nt_off(wm * 64 + i * 32 + fr, ks * 2 + fh));
                fb[i] = *(const bf16x8*)(pb + nt_off(wn * 64 + i * 32 + fr, ks * 2 + fh));
            }
#pragma unroll
            for (int i = 0; i < 2; ++i)
#pragma unroll
                for (int j = 0; j < 2; ++j)
                    acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(fb[j], fa[i], acc[i][j], 0, 0, 0);   // D rows = n, col = m
        }
        if (kt + 1 < nk) lstore(cur ^ 1);
        __syncthreads();
    }

    // ---- epilogue. The MFMA operands were swapped (D rows = n in registers, D column = m on the lane), so each
    // lane holds 4 consecutive n per register group: stage the wave's 64x64 fp32 sub-tile through LDS as [m][n]
    // (row stride 68 floats: conflict-free ds_write_b128) and read it back row-wise, so bias / residual / GELU
    // and the global stores all run on 16-byte row-contiguous vectors (16 lanes = one 256-byte row segment).
    float* stg = (float*)smem + wave * (64 * 68);
#pragma unroll
    for (int i = 0; i < 2; ++i)
#pragma unroll
        for (int j = 0; j < 2; ++j)
#pragma unroll
            for (int g4 = 0; g4 < 4; ++g4) {
                f32x4 v;
#pragma unroll
                for (int e = 0; e < 4; ++e) v[e] = acc[i][j][4 * g4 + e];
                *(f32x4*)(stg + (i * 32 + fr) * 68 + j * 32 + 8 * g4 + 4 * fh) = v;
            }
    // same wave reads what it wrote: no workgroup barrier needed (the K loop ended with one)
    const int c4 = lane & 15, rsub = lane >> 4;
    const int n = n0 + wn * 64 + c4 * 4;
    if (n < g.N) {
        f32x4 bias = {0.f, 0.f, 0.f, 0.f};
        if (g.bias) bias = *(const f32x4*)(g.bias + n);
#pragma unroll 4
        for (int t = 0; t < 16; ++t) {
            const int row = t * 4 + rsub;
            const int m = m0 + wm * 64 + row;
            if (m >= g.M) continue;
            f32x4 v = *(const f32x4*)(stg + row * 68 + c4 * 4);
            v += bias;
            const size_t o = (size_t)m * g.ldc + n;
            if (EPI == QST_EPI_BF16) {
                u32x2 pk; pk[0] = pack_bf16x2(v[0], v[1]); pk[1] = pack_bf16x2(v[2], v[3]);
                *(u32x2*)((bf16*)g.C + o) = pk;
            } else if (EPI == QST_EPI_F32_RESID || EPI == QST_EPI_F32_RESID_BF16) {
                if (g.resid) v += *(const f32x4*)(g.resid + (size_t)m * g.ldr + n);
                *(f32x4*)((float*)g.C + o) = v;
                if (EPI == QST_EPI_F32_RESID_BF16) {
                    u32x2 pk; pk[0] = pack_bf16x2(v[0], v[1]); pk[1] = pack_bf16x2(v[2], v[3]);
                    *(u32x2*)((bf16*)g.C2 + o) = pk;
                }
            } else if (EPI == QST_EPI_GELU) {
                u32x2 pk; pk[0] = pack_bf16x2(v[0], v[1]); pk[1] = pack_bf16x2(v[2], v[3]);
                *(u32x2*)((bf16*)g.C + o) = pk;                                   // u (pre-activation), saved for backward
                pk[0] = pack_bf16x2(gelu_erf(v[0]), gelu_erf(v[1])); pk[1] = pack_bf16x2(gelu_erf(v[2]), gelu_erf(v[3]));
                *(u32x2*)((bf16*)g.C2 + o) = pk;                                  // h
            } else if (EPI == QST_EPI_GELU_BWD) {
                const u32x2 ua = *(const u32x2*)((const bf16*)g.aux + o);
                u32x2 pk;
                pk[0] = pack_bf16x2(v[0] * gelu_erf_grad(bf16lo(ua[0])), v[1] * gelu_erf_grad(bf16hi(ua[0])));
                pk[1] = pack_bf16x2(v[2] * gelu_erf_grad(bf16lo(ua[1])), v[3] * gelu_erf_grad(bf16hi(ua[1])));
                *(u32x2*)((bf16*)g.C + o) = pk;
            }
        }
    }
}

// ---------------------------------------------------------------- TN
// LDS image per operand tile: [64 m-rows][128 bf16] = 256-byte rows; chunk c (16 B) of row r at
// chunk c ^ (((r&3)<<2) | ((r>>2)&3))  -- conflict-free for the 32x32x16 transposed reads.
__device__ __forceinline__ uint32_t tn_off(int row, int chunk) {
    return (uint32_t)(row * 256 + ((chunk ^ (((row & 3) << 2) | ((row >> 2) & 3))) << 4));
}
__device__ __forceinline__ bf16x4 lds_tr16(const char* p) {
    return __builtin_amdgcn_ds_read_tr16_b64_v4bf16((__attribute__((address_space(3))) bf16x4*)(p));
}

__global__ __launch_bounds__(256, 2) void gemm_tn_kernel(QstGemmArgs g) {
    extern __shared__ __attribute__((aligned(16))) char smem[];   // 2 x (16K A + 16K B)
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int wm = wave >> 1, wn = wave & 1;
    // output C[N, K]; reduction over M split into g.splits contiguous ranges
    const int ntn = (g.N + BM - 1) / BM, ntk = (g.K + BN - 1) / BN;
    const int tile = blockIdx.x % (ntn * ntk), split = blockIdx.x / (ntn * ntk);
    const int n0 = (tile / ntk) * BM, k0 = (tile % ntk) * BN;
    const int per = (((g.M + g.splits - 1) / g.splits) + BK - 1) / BK * BK;
    const int mbeg = split * per, mend = min(g.M, mbeg + per);
    if (mbeg >= mend) return;

    const bf16* Ab = (const bf16*)g.A + (size_t)mbeg * g.lda + n0;
    const bf16* Bb = (const bf16*)g.B + (size_t)mbeg * g.ldb + k0;
    // range = rows [mbeg, mend); the last row's tail past the allocation reads as zero
    const uint32_t bytes_a = (uint32_t)min((size_t)(mend - mbeg) * g.lda * 2u - (size_t)n0 * 2u, (size_t)0xFFFFFFF0u);
    const uint32_t bytes_b = (uint32_t)min((size_t)(mend - mbeg) * g.ldb * 2u - (size_t)k0 * 2u, (size_t)0xFFFFFFF0u);
    const __amdgpu_buffer_rsrc_t ra = make_rsrc(Ab, bytes_a);
    const __amdgpu_buffer_rsrc_t rb = make_rsrc(Bb, bytes_b);

    // staging: 64 rows x 16 chunks per operand = 1024 chunks -> 4 per thread: row = tid/16 + 16*i, chunk = tid%16
    const int srow = tid >> 4, sch = tid & 15;
    uint32_t goff_a[4], goff_b[4], loff[4];
#pragma unroll
    for (int i = 0; i < 4; ++i) {
        const int r = srow + 16 * i;
        goff_a[i] = (uint32_t)r * g.lda * 2u + sch * 16u;
        goff_b[i] = (uint32_t)r * g.ldb * 2u + sch * 16u;
        loff[i] = tn_off(r, sch);
    }
    // columns beyond the matrix width must not alias the next row: mask those chunks to zero
    const bool a_ok = (n0 + sch * 8) < g.N, b_ok = (k0 + sch * 8) < g.K;
    u32x4 sa[4], sb[4];
    const u32x4 z = {0, 0, 0, 0};
    auto gload = [&](int mt) {
        const uint32_t ra_off = (uint32_t)mt * BK * g.lda * 2u, rb_off = (uint32_t)mt * BK * g.ldb * 2u;
#pragma unroll
        for (int i = 0; i < 4; ++i) {
            sa[i] = a_ok ? buf_load16(ra, goff_a[i] + ra_off) : z;
            sb[i] = b_ok ? buf_load16(rb, goff_b[i] + rb_off) : z;
        }
    };
    auto lstore = [&](int buf) {
        char* pa = smem + buf * 32768;
        char* pb = pa + 16384;
#pragma unroll
        for (int i = 0; i < 4; ++i) {
            *(u32x4*)(pa + loff[i]) = sa[i];
            *(u32x4*)(pb + loff[i]) = sb[i];
        }
    };

    f32x16 acc[2][2], bacc[2];
#pragma unroll
    for (int i = 0; i < 2; ++i) {
#pragma unroll
        for (int r = 0; r < 16; ++r) bacc[i][r] = 0.f;
#pragma unroll
        for (int j = 0; j < 2; ++j)
#pragma unroll
            for (int r = 0; r < 16; ++r) acc[i][j][r] = 0.f;
    }
    const bool do_bias = (g.colsum != nullptr) && (k0 == 0) && (wn == 0);
    bf16x8 ones;
#pragma unroll
    for (int e = 0; e < 8; ++e) ones[e] = (bf16)1.0f;

    const int nm = (mend - mbeg + BK - 1) / BK;
    gload(0);
    lstore(0);
    __syncthreads();
    // transposed-read lane geometry (cdna guide T10): 16-lane group gq, lane i=4q+p supplies row q, cols 4p..4p+3
    const int li = lane & 15, q = li >> 2, p = li & 3, gsel = (lane >> 4) & 1, fh = lane >> 5;
    for (int mt = 0; mt < nm; ++mt) {
        const int cur = mt & 1;
        if (mt + 1 < nm) gload(mt + 1);
        const char* pa = smem + cur * 32768;
        const char* pb = pa + 16384;
#pragma unroll
        for (int ks = 0; ks < 4; ++ks) {
            bf16x8 fa[2], fb[2];
#pragma unroll
            for (int i = 0; i < 2; ++i) {
#pragma unroll
                for (int jj = 0; jj < 2; ++jj) {
                    const int row = ks * 16 + 8 * fh + 4 * jj + q;
                    const int ca = (wm * 64 + i * 32 + gsel * 16) / 8 + (p >> 1);
                    const int cb = (wn * 64 + i * 32 + gsel * 16) / 8 + (p >> 1);
                    const bf16x4 ta = lds_tr16(pa + tn_off(row, ca) + 8 * (p & 1));
                    const bf16x4 tb = lds_tr16(pb + tn_off(row, cb) + 8 * (p & 1));
#pragma unroll
                    for (int e = 0; e < 4; ++e) { fa[i][jj * 4 + e] = ta[e]; fb[i][jj * 4 + e] = tb[e]; }
                }
            }
#pragma unroll
            for (int i = 0; i < 2; ++i)
#pragma unroll
                for (int j = 0; j < 2; ++j)
                    acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(fa[i], fb[j], acc[i][j], 0, 0, 0);
            if (do_bias) {
#pragma unroll
                for (int i = 0; i < 2; ++i)
                    bacc[i] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(fa[i], ones, bacc[i], 0, 0, 0);
            }
        }
        if (mt + 1 < nm) lstore(cur ^ 1);
        __syncthreads();
    }

    const int fr = lane & 31;
    float* C = (float*)g.C;
#pragma unroll
    for (int j = 0; j < 2; ++j) {
        const int k = k0 + wn * 64 + j * 32 + fr;
        if (k >= g.K) continue;
#pragma unroll
        for (int i = 0; i < 2; ++i) {
#pragma unroll
            for (int r = 0; r < 16; ++r) {
                const int n = n0 + wm * 64 + i * 32 + (r & 3) + 8 * (r >> 2) + 4 * fh;
                if (n < g.N) atomicAdd(&C[(size_t)n * g.ldc + k], acc[i][j][r]);
            }
        }
    }
    if (do_bias && fr == 0) {
#pragma unroll
        for (int i = 0; i < 2; ++i)
#pragma unroll
            for (int r = 0; r < 16; ++r) {
                const int n = n0 + wm * 64 + i * 32 + (r & 3) + 8 * (r >> 2) + 4 * fh;
                if (n < g.N) atomicAdd(&g.colsum[n], bacc[i][r]);
            }
    }
}

}  // namespace

extern "C" int qst_gemm_nt(const QstGemmArgs* a, int epi, void* stream) {
    if (!a || !a->A || !a->B || !a->C || a->M <= 0 || a->N <= 0 || a->K <= 0) return QST_ERR_BAD_ARG;
    if (a->K % BK != 0 || a->lda % 8 != 0 || a->ldb % 8 != 0 || a->N % 4 != 0 || a->ldc % 4 != 0) return QST_ERR_UNSUPPORTED;
    const int ntm = (a->M + BM - 1) / BM, ntn = (a->N + BN - 1) / BN;
    dim3 grid(ntm * ntn), block(256);
    hipStream_t st = (hipStream_t)stream;
    const size_t lds = 4 * 64 * 68 * sizeof(float);   // 69632: epilogue staging is the larger user
#define QST_NT_CASE(E)                                                                                  \
    case E: {                                                                                           \
        static bool attr_set = false;                                                                   \
        if (!attr_set) {                                                                                \
            QST_HIP_CHECK(hipFuncSetAttribute((const void*)gemm_nt_kernel<E>,                           \
                                              hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds));   \
            attr_set = true;                                                                            \
        }                                                                                               \
        gemm_nt_kernel<E><<<grid, block, lds, st>>>(*a);                                                \
    } break;
    switch (epi) {
        QST_NT_CASE(QST_EPI_BF16)
        QST_NT_CASE(QST_EPI_F32_RESID)
        QST_NT_CASE(QST_EPI_GELU)
        QST_NT_CASE(QST_EPI_GELU_BWD)
        QST_NT_CASE(QST_EPI_F32_RESID_BF16)
        default: return QST_ERR_BAD_ARG;
    }
#undef QST_NT_CASE
    QST_LAUNCH_CHECK();
    return QST_OK;
}

extern "C" int qst_gemm_tn(const QstGemmArgs* a, void* stream) {
    if (!a || !a->A || !a->B || !a->C || a->M <= 0 || a->N <= 0 || a->K <= 0) return QST_ERR_BAD_ARG;
    if (a->lda % 8 != 0 || a->ldb % 8 != 0 || a->N % 8 != 0 || a->K % 8 != 0) return QST_ERR_UNSUPPORTED;
    QstGemmArgs g = *a;
    const int tiles = ((g.N + BM - 1) / BM) * ((g.K + BN - 1) / BN);
    if (g.splits <= 0) {
        // fill ~2 waves of workgroups over 256 CUs, keep >= 512 reduction rows per split
        int s = (512 + tiles - 1) / tiles;
        const int max_s = (g.M + 511) / 512;
        g.splits = s < 1 ? 1 : (s > max_s ? max_s : s);
    }
    static bool attr_set = false;
    if (!attr_set) {
        QST_HIP_CHECK(hipFuncSetAttribute((const void*)gemm_tn_kernel, hipFuncAttributeMaxDynamicSharedMemorySize, 65536));
        attr_set = true;
    }
    gemm_tn_kernel<<<dim3(tiles * g.splits), dim3(256), 65536, (hipStream_t)stream>>>(g);
    QST_LAUNCH_CHECK();
    return QST_OK;
}
