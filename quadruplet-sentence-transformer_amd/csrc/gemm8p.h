// gemm8p.h -- the 8-wave, 8-phase K loop for long-reduction bf16 GEMMs on gfx950 (device code, header-only).
//
// Structure (cdna_hip_programming.md, "The 256^2 8-phase template"; written from that description): 512 threads = 8 waves
// as 2 (rows) x 4 (columns), each wave a block of TM x TN tiles of v_mfma_f32_16x16x32_bf16 (4 TM TN accumulator
// registers): TM, TN = 8, 4 is the guide's 256 x 256 workgroup tile; 4, 6 a 128 x 384 tile, which divides every GEMM
// dimension of the three model families (384 / 768 / 1152 / 1536 / 2304 / 3072) and gives whole tiles per CU at the step's M.
// The reduction is walked in 64-deep K-tiles; a K-tile's operands sit in LDS as four half-tiles
//     A0 / A1 : what quadrant row    qm = 0 / 1 of every wave's block reads      (NA LDS-DMAs per thread)
//     B0 / B1 : what quadrant column qn = 0 / 1 reads                            (NB LDS-DMAs per thread)
// and a K-tile is four phases, one quadrant each, in the order (0,0) (0,1) (1,1) (1,0): the fragments a phase needs that
// are not yet in registers are read at its start, and every phase stages ONE half-tile of a later K-tile. Two buffers
// (K-tile parity) x 64 KB. Each phase is   { fragment reads ; stage ; s_barrier ; MFMAs ; s_barrier }   and waves 4-7
// run one barrier behind waves 0-3: on every SIMD one wave multiplies while its partner reads fragments and issues DMAs.
//
// Hazards, by the count (phases p = 4t + 1 .. 4t + 4 belong to K-tile t, which lives in buffer t & 1):
//   issue : A1(t+1) at 4t+1, B0(t+2) at 4t+2, A0(t+2) at 4t+3, B1(t+2) at 4t+4      (three half-tiles in flight)
//   RAW   : `s_waitcnt vmcnt(2 NB + NA)` in phase 4t+4, before its first barrier, retires everything up to A1(t+1), i.e. all
//           of K-tile t+1, first read in phase 4t+5 -- one phase after the wait (the staggered half reaches its wait half a
//           phase later; its partners' phase-4t+5 reads come after the barrier that follows)
//   WAR   : last reads  B0(t), A0(t): 4t+1;  B1(t): 4t+2;  A1(t): 4t+3.  Restaged at 4t+2 (B0: one phase later -- legal
//           because phase 4t+1 issues its B reads FIRST and retires them with a counted lgkmcnt before its first barrier),
//           4t+3 (A0), 4t+4 (B1), 4t+5 (A1): two phases later.
// K-tiles past the end are staged from a zero-length buffer (hardware zero fill, no memory traffic), so the DMA count per
// phase -- and with it every vmcnt -- is the same in the tail as in steady state. The loop runs K-tiles in pairs; an odd
// count is padded with one such zero K-tile.
//
// Two operand forms share the skeleton (struct *Ops below: LDS image, DMA source addresses, fragment reads):
//   NtOps : C = A[M,K] . B[N,K]^T, both operands K-contiguous; [rows][64 k] images of 128-byte rows, XOR-swizzled on
//           the DMA SOURCE address, fragments by ds_read_b128                         (forward Linear, dgrad)
//   TnOps : C[n,k] = sum_m A[m,n] B[m,k], both operands row-major over the reduction index m; [64 m][64 col] images,
//           fragments by two ds_read_b64_tr_b16 each                                  (wgrad)
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>

namespace g8p {

// the 16-bit operand type of the translation unit (qst_common.h has the same switch; this header stands alone for the probes)
#ifndef QST_OP_F16
#define QST_OP_F16 0
#endif
#if QST_OP_F16
typedef _Float16 op16;
#else
typedef __bf16 op16;
#endif
typedef __attribute__((ext_vector_type(8))) op16 op16x8;
typedef __attribute__((ext_vector_type(4))) op16 op16x4;
typedef __attribute__((ext_vector_type(4))) float f32x4_t;
typedef __attribute__((address_space(3))) void lds_void_t;

constexpr int BUF_BYTES = 65536;                 // one K-tile of both operands
constexpr int LDS_BYTES = 2 * BUF_BYTES;
constexpr uint32_t kOOB = 0x7FFFFFF0u;           // voffset that always fails the buffer range check -> zero fill

__device__ __forceinline__ int xcd_remap(int b, int nwg) {
    // bijective XCD-contiguous remap (blocks b and b + 8 share an XCD): neighbours in the remapped id share an L2
    const int q = nwg >> 3, r = nwg & 7, x = b & 7;
    return (x < r ? x * (q + 1) : r * (q + 1) + (x - r) * q) + (b >> 3);
}
__device__ __forceinline__ uint32_t pack2(float lo, float hi) {
    typedef __attribute__((ext_vector_type(2))) op16 v2;
    v2 v; v[0] = (op16)lo; v[1] = (op16)hi;
    return __builtin_bit_cast(uint32_t, v);
}
__device__ __forceinline__ f32x4_t mfma16(op16x8 a, op16x8 b, f32x4_t c) {        // v_mfma_f32_16x16x32_{bf16,f16}
#if QST_OP_F16
    return __builtin_amdgcn_mfma_f32_16x16x32_f16(a, b, c, 0, 0, 0);
#else
    return __builtin_amdgcn_mfma_f32_16x16x32_bf16(a, b, c, 0, 0, 0);
#endif
}
__device__ __forceinline__ __amdgpu_buffer_rsrc_t rsrc(const void* base, uint32_t bytes) {
    return __builtin_amdgcn_make_buffer_rsrc(const_cast<void*>(base), 0, (int)bytes, 0x00020000);
}
__device__ __forceinline__ void dma16(__amdgpu_buffer_rsrc_t r, char* lds_wave_base, uint32_t voff, uint32_t soff) {
    // one wave-instruction writes 64 x 16 B = 1 KB at lds_wave_base + lane * 16 (base must be wave-uniform)
    __builtin_amdgcn_raw_ptr_buffer_load_lds(r, (lds_void_t*)lds_wave_base, 16, (int)voff, (int)soff, 0, 0);
}
template <int N> __device__ __forceinline__ void wait_vm() { asm volatile("s_waitcnt vmcnt(%0)" ::"n"(N) : "memory"); }
template <int N> __device__ __forceinline__ void wait_lgk() { asm volatile("s_waitcnt lgkmcnt(%0)" ::"n"(N) : "memory"); }
__device__ __forceinline__ void bar() {
    __builtin_amdgcn_sched_barrier(0);
    __builtin_amdgcn_s_barrier();
    __builtin_amdgcn_sched_barrier(0);
}

// ---------------------------------------------------------------- the phase skeleton
// OPS: NA, NB, wr, stage_a(kt, h), stage_b(kt, h), rd_a<BUF, QM, PART>(), rd_b<BUF, QN>(), mm<QM, QN>(); the A reads of a
//      quadrant row come in two parts: part 0 = LGK LDS reads (<= 15, the width of the lgkmcnt field), issued right after the B
//      reads of a K-tile's first phase and followed by `s_waitcnt lgkmcnt(LGK)`, which therefore retires the B reads; part 1 the
//      rest (empty when all fit).
// nk: K-tiles that hold data (any count >= 1). Returns behind a workgroup barrier with no DMA outstanding.
template <class OPS>
__device__ __forceinline__ void kloop8(OPS& o, int nk) {
    constexpr int NA = OPS::NA, NB = OPS::NB, LGK = OPS::LGK;
    // vector-memory operations issued after A1(t+1) when phase 4t+4 waits. OPS::touch(kt) / NTOUCH: NTOUCH extra
    // vector-memory instructions per K-tile and wave that travel with the B0 stage of K-tile kt (the fp8 form stages the
    // K-tile's E8M0 block scales there; round 4 also tried an L2 touch-ahead of the operand rows two and four K-tiles early:
    // 167 -> 185 / 191 us on the MiniLM weight gradients, removed)
    constexpr int NW = 2 * NB + NA + OPS::NTOUCH;
    static_assert(LGK <= 15 && NW <= 63, "counter fields");
    o.stage_b(0, 0); o.touch(0); o.stage_a(0, 0); o.stage_b(0, 1); o.stage_a(0, 1);
    o.stage_b(1, 0); o.touch(1); o.stage_a(1, 0); o.stage_b(1, 1);
    wait_vm<NW>();                                      // K-tile 0 has landed (this wave's part)
    bar();
    if (o.wr == 1) bar();                               // waves 4-7 run one barrier behind
#pragma unroll 1
    for (int t = 0; t < nk; t += 2) {
        o.begin_iter(t);
        // ---- K-tile t (buffer 0)
        o.template rd_b<0, 0>();
        __builtin_amdgcn_sched_barrier(0);
        o.template rd_a<0, 0, 0>();
        o.stage_a(t + 1, 1);
        wait_lgk<LGK>();                                // the B0 reads have returned: B0 may be restaged next phase
        o.template rd_a<0, 0, 1>();
        bar(); o.template mm<0, 0>(); bar();

        o.template rd_b<0, 1>();
        o.stage_b(t + 2, 0);
        o.touch(t + 2);
        bar(); o.template mm<0, 1>(); bar();

        o.template rd_a<0, 1, 0>();
        o.template rd_a<0, 1, 1>();
        o.stage_a(t + 2, 0);
        bar(); o.template mm<1, 1>(); bar();

        o.stage_b(t + 2, 1);
        wait_vm<NW>();                                  // K-tile t+1 has landed (this wave's part)
        bar(); o.template mm<1, 0>(); bar();

        // ---- K-tile t + 1 (buffer 1)
        o.template rd_b<1, 0>();
        __builtin_amdgcn_sched_barrier(0);
        o.template rd_a<1, 0, 0>();
        o.stage_a(t + 2, 1);
        wait_lgk<LGK>();
        o.template rd_a<1, 0, 1>();
        bar(); o.template mm<0, 0>(); bar();

        o.template rd_b<1, 1>();
        o.stage_b(t + 3, 0);
        o.touch(t + 3);
        bar(); o.template mm<0, 1>(); bar();

        o.template rd_a<1, 1, 0>();
        o.template rd_a<1, 1, 1>();
        o.stage_a(t + 3, 0);
        bar(); o.template mm<1, 1>(); bar();

        o.stage_b(t + 3, 1);
        wait_vm<NW>();
        bar(); o.template mm<1, 0>(); bar();
    }
    if (o.wr == 0) bar();                               // waves 0-3 wait for the staggered half's last phase
    wait_vm<0>();                                       // the tail's zero-fill DMAs
    bar();
}

// ---------------------------------------------------------------- NT operands
// Workgroup tile BM x BN = 32 TM x 64 TN. LDS buffer: A [BM][64 k] then B [BN][64 k], 128-byte rows; 16-byte chunk c of
// row r sits at chunk position c ^ ((r >> 1) & 7) (conflict-free for the 16x16x32 fragment ds_read_b128). Wave (wr, wc)
// owns rows wr BM/2 + [0, BM/2) and columns wc BN/4 + [0, BN/4); its quadrants halve both.
//   acc[i][j][r] = C[m0 + wr BM/2 + 16 i + (lane & 15)][n0 + wc BN/4 + 16 j + 4 (lane >> 4) + r]      (D rows = n)
template <int TM, int TN>
struct NtOps {
    static constexpr int BM = 32 * TM, BN = 64 * TN;
    static constexpr int NA = BM / 128, NB = BN / 128, LGK = TM, NTOUCH = 0;
    static_assert((BM + BN) * 128 == BUF_BYTES, "a K-tile of both operands fills one 64 KB buffer");
    static_assert(BM % 128 == 0 && BN % 128 == 0 && TM % 2 == 0 && TN % 2 == 0, "tile geometry");
    char* smem;
    __amdgpu_buffer_rsrc_t ra, rb, rz;
    uint32_t va[NA], vb[NB];          // per-lane source offsets (h = 0)
    uint32_t la[NA], lb[NB];          // wave-uniform LDS destinations (buffer 0, h = 0)
    uint32_t ha, hb;                  // source byte offset of half-tile 1
    uint32_t ao0, ao1, bo0, bo1;      // per-lane fragment read offsets (k-step 0 / 1)
    int nk, wr, wc;
    f32x4_t acc[TM][TN];
    op16x8 fa[TM / 2][2], fb[2][TN / 2][2];

    __device__ __forceinline__ void init(const op16* A, int lda, int rows_a, const op16* B, int ldb, int rows_b, int K,
                                         char* smem_) {
        const int tid = threadIdx.x, lane = tid & 63;
        const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
        smem = smem_; wr = wave >> 2; wc = wave & 3; nk = K >> 6;
        ra = rsrc(A, (uint32_t)rows_a * (uint32_t)lda * 2u);
        rb = rsrc(B, (uint32_t)rows_b * (uint32_t)ldb * 2u);
        rz = rsrc(A, 0u);
        // one DMA instruction = 8 tile rows x 128 B; a half-tile of A is BM/16 of them, of B BN/16
#pragma unroll
        for (int t = 0; t < NA; ++t) {
            const int jb = wave * NA + t;
            const int row0 = (jb / (BM / 32)) * (BM / 2) + (jb % (BM / 32)) * 8, row = row0 + (lane >> 3);
            va[t] = (uint32_t)row * lda * 2u + (uint32_t)(((lane & 7) ^ ((row >> 1) & 7)) << 4);
            la[t] = (uint32_t)row0 * 128u;
        }
#pragma unroll
        for (int t = 0; t < NB; ++t) {
            const int jb = wave * NB + t;
            const int row0 = (jb / (BN / 64)) * (BN / 4) + (jb % (BN / 64)) * 8, row = row0 + (lane >> 3);
            vb[t] = (uint32_t)row * ldb * 2u + (uint32_t)(((lane & 7) ^ ((row >> 1) & 7)) << 4);
            lb[t] = (uint32_t)(BM + row0) * 128u;
        }
        ha = (uint32_t)(BM / 4) * lda * 2u; hb = (uint32_t)(BN / 8) * ldb * 2u;
        const uint32_t lo0 = (uint32_t)((lane & 15) * 128 + (((lane >> 4) ^ ((lane & 15) >> 1)) << 4));
        ao0 = lo0 + wr * (BM / 2) * 128; ao1 = ao0 ^ 64u;
        bo0 = lo0 + (BM + wc * (BN / 4)) * 128; bo1 = bo0 ^ 64u;
#pragma unroll
        for (int i = 0; i < TM; ++i)
#pragma unroll
            for (int j = 0; j < TN; ++j) acc[i][j] = f32x4_t{0.f, 0.f, 0.f, 0.f};
    }
    __device__ __forceinline__ void stage_a(int kt, int h) {
        const __amdgpu_buffer_rsrc_t r = kt < nk ? ra : rz;
        const uint32_t so = (uint32_t)kt * 128u + (h ? ha : 0u);
        char* d = smem + (kt & 1) * BUF_BYTES + h * (BM / 4) * 128;
#pragma unroll
        for (int t = 0; t < NA; ++t) dma16(r, d + la[t], va[t], so);
    }
    __device__ __forceinline__ void stage_b(int kt, int h) {
        const __amdgpu_buffer_rsrc_t r = kt < nk ? rb : rz;
        const uint32_t so = (uint32_t)kt * 128u + (h ? hb : 0u);
        char* d = smem + (kt & 1) * BUF_BYTES + h * (BN / 8) * 128;
#pragma unroll
        for (int t = 0; t < NB; ++t) dma16(r, d + lb[t], vb[t], so);
    }
    __device__ __forceinline__ void touch(int) {}
    __device__ __forceinline__ void begin_iter(int) {}
    template <int BUF, int QM, int PART> __device__ __forceinline__ void rd_a() {
        if (PART == 1) return;
#pragma unroll
        for (int i = 0; i < TM / 2; ++i) {
            fa[i][0] = *(const op16x8*)(smem + BUF * BUF_BYTES + (QM * (BM / 4) + i * 16) * 128 + ao0);
            fa[i][1] = *(const op16x8*)(smem + BUF * BUF_BYTES + (QM * (BM / 4) + i * 16) * 128 + ao1);
        }
    }
    template <int BUF, int QN> __device__ __forceinline__ void rd_b() {
#pragma unroll
        for (int j = 0; j < TN / 2; ++j) {
            fb[QN][j][0] = *(const op16x8*)(smem + BUF * BUF_BYTES + (QN * (BN / 8) + j * 16) * 128 + bo0);
            fb[QN][j][1] = *(const op16x8*)(smem + BUF * BUF_BYTES + (QN * (BN / 8) + j * 16) * 128 + bo1);
        }
    }
    template <int QM, int QN> __device__ __forceinline__ void mm() {
        __builtin_amdgcn_s_setprio(1);
#pragma unroll
        for (int s = 0; s < 2; ++s)
#pragma unroll
            for (int i = 0; i < TM / 2; ++i)
#pragma unroll
                for (int j = 0; j < TN / 2; ++j)
                    acc[QM * (TM / 2) + i][QN * (TN / 2) + j] = mfma16(fb[QN][j][s], fa[i][s], acc[QM * (TM / 2) + i][QN * (TN / 2) + j]);
        __builtin_amdgcn_s_setprio(0);
    }
};

// ---------------------------------------------------------------- NT operands, MXFP8 x MXFP8 (v_mfma_scale_f32_16x16x128_f8f6f4)
// The same tile geometry, LDS images, swizzle and DMA map as NtOps with a K-tile of 128 e4m3 elements = the same 128-byte rows;
// one block-scaled MFMA per 16 x 16 tile and K-tile (32 cycles: twice the bf16 rate). Operand layout of the instruction
// (tools/probe/mfma_f8_16x16_probe.hip): lane (r = l & 15, g = l >> 4) holds bytes k = 16 g .. 16 g + 15 of row r in registers
// 0-3 and k = 64 + 16 g .. in registers 4-7 -- 16-byte chunks g and 4 + g of the row's 128 bytes -- and supplies the E8M0 scale
// of MX block g (k = 32 g .. 32 g + 31) of row r in byte `opsel` of its scale register. Scales (uint8 E8M0, stage-major as
// qst_quant_mx writes them: the four of a row and K-tile are one dword at (kt * rows + r) * 4) travel with the B0 stage of
// their K-tile: every wave issues ONE 4-byte-per-lane LDS-DMA (waves 0-3: 64 A rows each, waves 4-7: 64 B rows; the tile has
// at most 256 + 256 or 128 + 384 rows = 8 x 64) into a 4-slot ring of 2 KB, so all waves carry the same vmcnt.
typedef __attribute__((ext_vector_type(8))) int i32x8_t;
typedef __attribute__((ext_vector_type(4))) uint32_t u32x4_t;
template <int TM, int TN>
struct NtOpsF8 {
    static constexpr int BM = 32 * TM, BN = 64 * TN;
    static constexpr int NA = BM / 128, NB = BN / 128, LGK = TM, NTOUCH = 1;
    static constexpr int SCALE_OFF = LDS_BYTES;            // [4 slots][BM + BN rows] dwords behind the two K-tile buffers
    static constexpr int LDS_TOTAL = LDS_BYTES + 4 * (BM + BN) * 4;
    static_assert((BM + BN) * 128 == BUF_BYTES && (BM + BN) == 512, "tile geometry");
    char* smem;
    __amdgpu_buffer_rsrc_t ra, rb, rz, rs;     // rs: this wave's 64 scale rows (A or B)
    uint32_t va[NA], vb[NB], la[NA], lb[NB], ha, hb, ao0, ao1, bo0, bo1;
    uint32_t vs, ss;                           // scale DMA: per-lane byte offset, bytes per K-tile of the scale matrix
    uint32_t sao, sbo;                         // per-lane LDS offsets of this lane's A / B scale dwords (slot 0, tile 0)
    uint32_t gsh;                              // 8 * (lane >> 4): this lane's block inside the scale dword
    int nk, wr, wc;
    f32x4_t acc[TM][TN];
    i32x8_t fa[TM / 2], fb[2][TN / 2];
    int sa_[TM / 2], sb_[2][TN / 2];

    // A: e4m3 [rows_a, K] (lda bytes) at the tile's first row, As: its scale matrix (all M rows: row index m0 + r); B likewise
    __device__ __forceinline__ void init(const uint8_t* A, int lda, int rows_a, const uint8_t* As, int M, int m0,
                                         const uint8_t* B, int ldb, int rows_b, const uint8_t* Bs, int N, int n0, int K, char* smem_) {
        const int tid = threadIdx.x, lane = tid & 63;
        const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
        smem = smem_; wr = wave >> 2; wc = wave & 3; nk = K >> 7;
        ra = rsrc(A, (uint32_t)rows_a * (uint32_t)lda);
        rb = rsrc(B, (uint32_t)rows_b * (uint32_t)ldb);
        rz = rsrc(A, 0u);
#pragma unroll
        for (int t = 0; t < NA; ++t) {
            const int jb = wave * NA + t;
            const int row0 = (jb / (BM / 32)) * (BM / 2) + (jb % (BM / 32)) * 8, row = row0 + (lane >> 3);
            va[t] = (uint32_t)row * lda + (uint32_t)(((lane & 7) ^ ((row >> 1) & 7)) << 4);
            la[t] = (uint32_t)row0 * 128u;
        }
#pragma unroll
        for (int t = 0; t < NB; ++t) {
            const int jb = wave * NB + t;
            const int row0 = (jb / (BN / 64)) * (BN / 4) + (jb % (BN / 64)) * 8, row = row0 + (lane >> 3);
            vb[t] = (uint32_t)row * ldb + (uint32_t)(((lane & 7) ^ ((row >> 1) & 7)) << 4);
            lb[t] = (uint32_t)(BM + row0) * 128u;
        }
        ha = (uint32_t)(BM / 4) * lda; hb = (uint32_t)(BN / 8) * ldb;
        // fragment chunks g and 4 + g: positions differ in bit 2 of the chunk index whatever the row's swizzle
        const uint32_t lo0 = (uint32_t)((lane & 15) * 128 + (((lane >> 4) ^ ((lane & 15) >> 1)) << 4));
        ao0 = lo0 + wr * (BM / 2) * 128; ao1 = ao0 ^ 64u;
        bo0 = lo0 + (BM + wc * (BN / 4)) * 128; bo1 = bo0 ^ 64u;
        // scale rows of the tile in LDS order: A rows 0 .. BM-1, then B rows; wave w stages rows 64 w .. 64 w + 63 of that order
        {
            const int srow = wave * 64 + lane;                        // 0 .. 511
            const bool isA = wave * 64 < BM;                          // wave-uniform (BM is a multiple of 64): scalar descriptor
            const int r = isA ? srow : srow - BM;
            const int rows = isA ? rows_a : rows_b, total = isA ? M : N, first = isA ? m0 : n0;
            // one descriptor per wave is only possible when a wave's 64 rows are all A or all B: BM is a multiple of 64
            const uint8_t* base = (isA ? As : Bs) + (size_t)first * 4;
            rs = rsrc(base, (uint32_t)(((size_t)nk * total - first) * 4));
            vs = r < rows ? (uint32_t)r * 4u : kOOB;
            ss = (uint32_t)total * 4u;
        }
        sao = (uint32_t)(SCALE_OFF + (wr * (BM / 2) + (lane & 15)) * 4);
        sbo = (uint32_t)(SCALE_OFF + (BM + wc * (BN / 4) + (lane & 15)) * 4);
        gsh = 8u * (uint32_t)(lane >> 4);
#pragma unroll
        for (int i = 0; i < TM; ++i)
#pragma unroll
            for (int j = 0; j < TN; ++j) acc[i][j] = f32x4_t{0.f, 0.f, 0.f, 0.f};
    }
    __device__ __forceinline__ void stage_a(int kt, int h) {
        const __amdgpu_buffer_rsrc_t r = kt < nk ? ra : rz;
        const uint32_t so = (uint32_t)kt * 128u + (h ? ha : 0u);
        char* d = smem + (kt & 1) * BUF_BYTES + h * (BM / 4) * 128;
#pragma unroll
        for (int t = 0; t < NA; ++t) dma16(r, d + la[t], va[t], so);
    }
    __device__ __forceinline__ void stage_b(int kt, int h) {
        const __amdgpu_buffer_rsrc_t r = kt < nk ? rb : rz;
        const uint32_t so = (uint32_t)kt * 128u + (h ? hb : 0u);
        char* d = smem + (kt & 1) * BUF_BYTES + h * (BN / 8) * 128;
#pragma unroll
        for (int t = 0; t < NB; ++t) dma16(r, d + lb[t], vb[t], so);
    }
    __device__ __forceinline__ void touch(int kt) {          // the K-tile's block scales: 64 rows x 4 bytes per wave
        const __amdgpu_buffer_rsrc_t r = kt < nk ? rs : rz;
        __builtin_amdgcn_raw_ptr_buffer_load_lds(r, (lds_void_t*)(smem + SCALE_OFF + (kt & 3) * (BM + BN) * 4 + (threadIdx.x >> 6) * 256),
                                                 4, (int)vs, (int)((uint32_t)kt * ss), 0, 0);
    }
    static __device__ __forceinline__ i32x8_t frag(const char* p0, const char* p1) {
        const u32x4_t lo = *(const u32x4_t*)p0, hi = *(const u32x4_t*)p1;
        i32x8_t f;
        f[0] = (int)lo[0]; f[1] = (int)lo[1]; f[2] = (int)lo[2]; f[3] = (int)lo[3];
        f[4] = (int)hi[0]; f[5] = (int)hi[1]; f[6] = (int)hi[2]; f[7] = (int)hi[3];
        return f;
    }
    // BUF doubles as the K-tile's parity; its scale slot is (kt & 3) = BUF or BUF + 2: the skeleton's loop variable t is even in
    // the first half of an iteration and both t and t + 1 alternate slots 0/1 and 2/3 from one iteration to the next
    int slot2;                                              // 0 or 2: added to BUF for the scale slot of the current iteration
    __device__ __forceinline__ void begin_iter(int t) { slot2 = t & 2; }
    template <int BUF, int QM, int PART> __device__ __forceinline__ void rd_a() {
        if (PART == 1) return;
        const char* sc = smem + (BUF + slot2) * (BM + BN) * 4;
#pragma unroll
        for (int i = 0; i < TM / 2; ++i) {
            fa[i] = frag(smem + BUF * BUF_BYTES + (QM * (BM / 4) + i * 16) * 128 + ao0, smem + BUF * BUF_BYTES + (QM * (BM / 4) + i * 16) * 128 + ao1);
            sa_[i] = (int)(*(const uint32_t*)(sc + (QM * (BM / 4) + i * 16) * 4 + sao) >> gsh);
        }
    }
    template <int BUF, int QN> __device__ __forceinline__ void rd_b() {
        const char* sc = smem + (BUF + slot2) * (BM + BN) * 4;
#pragma unroll
        for (int j = 0; j < TN / 2; ++j) {
            fb[QN][j] = frag(smem + BUF * BUF_BYTES + (QN * (BN / 8) + j * 16) * 128 + bo0, smem + BUF * BUF_BYTES + (QN * (BN / 8) + j * 16) * 128 + bo1);
            sb_[QN][j] = (int)(*(const uint32_t*)(sc + (QN * (BN / 8) + j * 16) * 4 + sbo) >> gsh);
        }
    }
    template <int QM, int QN> __device__ __forceinline__ void mm() {
        __builtin_amdgcn_s_setprio(1);
#pragma unroll
        for (int i = 0; i < TM / 2; ++i)
#pragma unroll
            for (int j = 0; j < TN / 2; ++j)
                acc[QM * (TM / 2) + i][QN * (TN / 2) + j] = __builtin_amdgcn_mfma_scale_f32_16x16x128_f8f6f4(
                    fb[QN][j], fa[i], acc[QM * (TM / 2) + i][QN * (TN / 2) + j], 0, 0, 0, sb_[QN][j], 0, sa_[i]);
        __builtin_amdgcn_s_setprio(0);
    }
};

// ---------------------------------------------------------------- TN operands (weight gradients)
// C[n, k] = sum_m A[m, n] B[m, k]: tile BMn x BNk = 32 TM x 64 TN of C, a K-tile = 64 reduction rows m. Both operands are
// read with ds_read_b64_tr_b16 from [64 m][64 col] sub-images of 128-byte rows (8 KB; one LDS-DMA = 8 m-rows, each a whole
// 128-byte line of the source); chunk c (8 columns) of row m sits at chunk position c ^ f(m), f(m) = 2 ((m >> 1) & 1) +
// 4 ((m >> 3) & 1): the 32 lanes of a half-wave (two blocks of 4 rows x 16 columns, 8 rows apart) then cover all 64 banks
// once. Half-tile h of A = the columns n0 + h BMn/2 + [0, BMn/2) (BMn/128 sub-images), of B the columns k0 + h BNk/2 +
// [0, BNk/2); wave (wr, wc) owns, inside half h, the A columns wr BMn/4 + [0, BMn/4) and the B columns wc BNk/8 + [0, BNk/8):
//   acc[qm TM/2 + i][qn TN/2 + j][r] = C[n0 + qm BMn/2 + wr BMn/4 + 16 i + 4 (lane >> 4) + r][k0 + qn BNk/2 + wc BNk/8 + 16 j + (lane & 15)]
// HOOK::after_a<QM>(ops): called once per K-tile and quadrant row, after the last multiplication that uses the A fragments of
// that quadrant row (quadrant order (0,0) (0,1) (1,1) (1,0)): column sums of A = bias gradients.
struct NoTnHook { template <int QM, class O> __device__ __forceinline__ void after_a(O&) const {} };
template <int TM, int TN, class HOOK = NoTnHook>
struct TnOps {
    static constexpr int BMn = 32 * TM, BNk = 64 * TN;
    static constexpr int SA = BMn / 128, SB = BNk / 128;           // sub-images per half-tile
    static constexpr int NA = SA, NB = SB;
    static constexpr int I0 = (2 * TM <= 15) ? TM / 2 : TM / 4;     // A tiles read in part 0 (four transposed reads per tile)
    static constexpr int LGK = 4 * I0;
    static constexpr int NTOUCH = 0;
    static constexpr int BASE_B = 2 * SA * 8192;
    static_assert(2 * (SA + SB) * 8192 == BUF_BYTES, "a K-tile of both operands fills one 64 KB buffer");
    char* smem;
    __amdgpu_buffer_rsrc_t ra, rb, rz;
    uint32_t va[2][NA], vb[2][NB];    // per-lane source offsets by half-tile (kOOB where the column is outside the matrix)
    uint32_t la[NA], lb[NB];          // wave-uniform LDS destinations (buffer 0, h = 0)
    uint32_t ta[TM / 2], tb[TN / 2];  // per-lane transposed-read offsets of this wave's tiles (buffer 0, h = 0, k-step 0)
    uint32_t sa, sb;                  // source bytes per K-tile (64 rows)
    int nk, wr, wc;
    HOOK hook;
    f32x4_t acc[TM][TN];
    op16x8 fa[TM / 2][2], fb[2][TN / 2][2];

    // A, B: first reduction row of this piece; rows: reduction rows in the piece; n0 / k0: first column of the tile
    __device__ __forceinline__ void init(const op16* A, int lda, int N, int n0, const op16* B, int ldb, int K, int k0,
                                         int rows, char* smem_) {
        const int tid = threadIdx.x, lane = tid & 63;
        const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
        smem = smem_; wr = wave >> 2; wc = wave & 3; nk = (rows + 63) >> 6;
        ra = rsrc(A, (uint32_t)rows * (uint32_t)lda * 2u);
        rb = rsrc(B, (uint32_t)rows * (uint32_t)ldb * 2u);
        rz = rsrc(A, 0u);
        sa = 64u * lda * 2u; sb = 64u * ldb * 2u;
#pragma unroll
        for (int t = 0; t < NA; ++t) {
            const int jb = wave * NA + t, sub = jb >> 3, blk = jb & 7;
            const int row = blk * 8 + (lane >> 3);
            const int ch = (lane & 7) ^ ((((row >> 1) & 1) << 1) | (((row >> 3) & 1) << 2));
            la[t] = (uint32_t)(sub * 8192 + blk * 1024);
#pragma unroll
            for (int h = 0; h < 2; ++h) {
                const int col = n0 + h * (BMn / 2) + sub * 64 + ch * 8;
                va[h][t] = col < N ? (uint32_t)row * lda * 2u + (uint32_t)col * 2u : kOOB;
            }
        }
#pragma unroll
        for (int t = 0; t < NB; ++t) {
            const int jb = wave * NB + t, sub = jb >> 3, blk = jb & 7;
            const int row = blk * 8 + (lane >> 3);
            const int ch = (lane & 7) ^ ((((row >> 1) & 1) << 1) | (((row >> 3) & 1) << 2));
            lb[t] = (uint32_t)(BASE_B + sub * 8192 + blk * 1024);
#pragma unroll
            for (int h = 0; h < 2; ++h) {
                const int col = k0 + h * (BNk / 2) + sub * 64 + ch * 8;
                vb[h][t] = col < K ? (uint32_t)row * ldb * 2u + (uint32_t)col * 2u : kOOB;
            }
        }
        // transposed reads (guide T10): lane 4q + p of a 16-lane group g supplies row q, 8-byte piece p of the block's
        // 4 rows x 16 columns; the block of k-step s, half jj is rows 32 s + 8 g + 4 jj + [0, 4)
        const int g = lane >> 4, q = (lane >> 2) & 3, p = lane & 3;
        const int f = ((q >> 1) << 1) | ((g & 1) << 2);
        const uint32_t rowpart = (uint32_t)(128 * (8 * g + q) + 8 * (p & 1));
#pragma unroll
        for (int i = 0; i < TM / 2; ++i) {
            const int col = wr * (BMn / 4) + i * 16;                      // inside the half-tile
            ta[i] = (uint32_t)((col >> 6) * 8192) + rowpart + (uint32_t)((((((col >> 4) & 3) << 1) | (p >> 1)) ^ f) << 4);
        }
#pragma unroll
        for (int j = 0; j < TN / 2; ++j) {
            const int col = wc * (BNk / 8) + j * 16;
            tb[j] = (uint32_t)(BASE_B + (col >> 6) * 8192) + rowpart + (uint32_t)((((((col >> 4) & 3) << 1) | (p >> 1)) ^ f) << 4);
        }
#pragma unroll
        for (int i = 0; i < TM; ++i)
#pragma unroll
            for (int j = 0; j < TN; ++j) acc[i][j] = f32x4_t{0.f, 0.f, 0.f, 0.f};
    }
    __device__ __forceinline__ void stage_a(int kt, int h) {
        const __amdgpu_buffer_rsrc_t r = kt < nk ? ra : rz;
        const uint32_t so = (uint32_t)kt * sa;
        char* d = smem + (kt & 1) * BUF_BYTES + h * SA * 8192;
#pragma unroll
        for (int t = 0; t < NA; ++t) dma16(r, d + la[t], h ? va[1][t] : va[0][t], so);
    }
    __device__ __forceinline__ void stage_b(int kt, int h) {
        const __amdgpu_buffer_rsrc_t r = kt < nk ? rb : rz;
        const uint32_t so = (uint32_t)kt * sb;
        char* d = smem + (kt & 1) * BUF_BYTES + h * SB * 8192;
#pragma unroll
        for (int t = 0; t < NB; ++t) dma16(r, d + lb[t], h ? vb[1][t] : vb[0][t], so);
    }
    __device__ __forceinline__ void touch(int) {}
    __device__ __forceinline__ void begin_iter(int) {}
    static __device__ __forceinline__ op16x8 tr_frag(const char* p) {
        typedef __attribute__((ext_vector_type(4))) __bf16 raw4;           // (ds_read_b64_tr_b16 moves 16-bit lanes of either type)
        typedef __attribute__((address_space(3))) raw4 lds_v4;
        const op16x4 lo = __builtin_bit_cast(op16x4, __builtin_amdgcn_ds_read_tr16_b64_v4bf16((lds_v4*)(p)));
        const op16x4 hi = __builtin_bit_cast(op16x4, __builtin_amdgcn_ds_read_tr16_b64_v4bf16((lds_v4*)(p + 512)));
        op16x8 v;
#pragma unroll
        for (int e = 0; e < 4; ++e) { v[e] = lo[e]; v[4 + e] = hi[e]; }
        return v;
    }
    template <int BUF, int QM, int PART> __device__ __forceinline__ void rd_a() {
#pragma unroll
        for (int i = (PART ? I0 : 0); i < (PART ? TM / 2 : I0); ++i) {
            fa[i][0] = tr_frag(smem + BUF * BUF_BYTES + QM * SA * 8192 + ta[i]);
            fa[i][1] = tr_frag(smem + BUF * BUF_BYTES + QM * SA * 8192 + 4096 + ta[i]);
        }
    }
    template <int BUF, int QN> __device__ __forceinline__ void rd_b() {
#pragma unroll
        for (int j = 0; j < TN / 2; ++j) {
            fb[QN][j][0] = tr_frag(smem + BUF * BUF_BYTES + QN * SB * 8192 + tb[j]);
            fb[QN][j][1] = tr_frag(smem + BUF * BUF_BYTES + QN * SB * 8192 + 4096 + tb[j]);
        }
    }
    template <int QM, int QN> __device__ __forceinline__ void mm() {
        __builtin_amdgcn_s_setprio(1);
#pragma unroll
        for (int s = 0; s < 2; ++s)
#pragma unroll
            for (int i = 0; i < TM / 2; ++i)
#pragma unroll
                for (int j = 0; j < TN / 2; ++j)
                    acc[QM * (TM / 2) + i][QN * (TN / 2) + j] = mfma16(fa[i][s], fb[QN][j][s], acc[QM * (TM / 2) + i][QN * (TN / 2) + j]);
        __builtin_amdgcn_s_setprio(0);
        if ((QM == 0 && QN == 1) || (QM == 1 && QN == 0)) hook.template after_a<QM>(*this);   // last use of fa of row QM
    }
};

// C tile = A[m0 .., :] . B[n0 .., :]^T over K (K % 64 == 0), the guide's 256 x 256 form -- kept for tools/probe/gemm8p_probe.hip
__device__ __forceinline__ void kloop_nt(const op16* A, int lda, int rows_a, const op16* B, int ldb, int rows_b, int K,
                                         int m0, int n0, char* smem, f32x4_t (&acc)[8][4]) {
    NtOps<8, 4> o;
    o.init(A + (size_t)m0 * lda, lda, rows_a, B + (size_t)n0 * ldb, ldb, rows_b, K, smem);
    kloop8(o, o.nk);
#pragma unroll
    for (int i = 0; i < 8; ++i)
#pragma unroll
        for (int j = 0; j < 4; ++j) acc[i][j] = o.acc[i][j];
}

}  // namespace g8p
