// qst_kernels.h -- kernel-level C entry points of libqst.so (the building blocks of qst_encoder_forward/backward;
// exported so that tests can check each kernel against the oracle and tools can time it alone). Self-contained C:
// needs only <stddef.h> / <stdint.h>. The sources under csrc/ include this file (csrc/qst_kernels.h forwards here).
#pragma once
#include <stddef.h>
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

enum {
    QST_EPI_BF16 = 0,            // C(bf16) = acc + bias
    QST_EPI_F32_RESID = 1,       // C(f32)  = acc + bias + resid
    QST_EPI_GELU = 2,            // u = acc + bias ; C(bf16) = gelu'(u) (saved for backward) ; C2(bf16) = gelu(u)
    QST_EPI_GELU_BWD = 3,        // C(bf16) = acc * aux   (aux = the gelu'(u) saved by QST_EPI_GELU)
    QST_EPI_F32_RESID_BF16 = 4,  // C(f32) = acc + bias + resid ; C2(bf16) = same
    QST_EPI_GELU_MX = 5,         // qst_gemm_nt_f8 only: gelu(acc + bias) as MXFP8: C = e4m3 [M, ldc], C2 = E8M0 [M, ldc/32]
    QST_EPI_GELU_MX_TRAIN = 6    // qst_gemm_nt_f8 only: C / C2 as QST_EPI_GELU (bf16 gelu'(u), bf16 h) AND the bf16-rounded h as
                                 // MXFP8: C3 = e4m3 [M, N], C4 = E8M0 scales (fp8 training forward)
};

/* Dropout masks are a pure function of (state, site, element index): nothing is stored between forward and backward,
 * both recompute the same mask. state = device uint32[4] {seed lo, seed hi, step, 0} (qst_dropout_init / _advance in
 * qst.h); site tells the tensors of one step apart (QST_DROP_SITE_*); an element of flat index i is DROPPED when its 16
 * random bits -- the low (i even) or high (i odd) half of hash32((i >> 1) ^ key), key = hash32(seed lo ^
 * hash32(step * 0x9E3779B9 + site) ^ rotl16(seed hi)), hash32 = x ^= x >> 16; x *= 0x7feb352d; x ^= x >> 15;
 * x *= 0x846ca68b; x ^= x >> 16 -- are < thr16; kept elements are multiplied by 65536 / (65536 - thr16). So the
 * effective rate is thr16 / 65536 (p = 0.1: thr16 = 6554). oracle/dropout_ref.py restates this (and the 8-bit form the
 * attention probabilities use, below) in numpy.
 * Flat indices: hidden states [M, H]: m * H + n; attention probabilities: ((seq * A + head) * L + query) * L + key. */
typedef struct {
    const uint32_t* state;   /* NULL = no dropout */
    uint32_t site;
    uint32_t thr16;          /* 0 = no dropout */
} QstDrop;
#define QST_DROP_SITE_EMBED 0xE0u
#define QST_DROP_SITE_ATTN_OUT(layer) (4u * (uint32_t)(layer) + 0u)    /* output of the attention projection, [M, H]  */
#define QST_DROP_SITE_FFN_OUT(layer) (4u * (uint32_t)(layer) + 1u)     /* output of the second feed-forward GEMM      */
#define QST_DROP_SITE_PROBS(layer) (4u * (uint32_t)(layer) + 2u)       /* softmax probabilities, [nseq, A, L, L]      */
/* The attention probabilities (QST_DROP_SITE_PROBS) use a cheaper form of the same generator, 8 random bits per element and
 * four elements per hash word: byte b of hash32((i >> 2) ^ key) belongs to element (i & ~3) + b, dropped when byte < thr8 =
 * min(255, (thr16 + 128) >> 8), kept ones multiplied by 256 / (256 - thr8): their rate is quantised to 1/256 (p = 0.1 ->
 * 26/256 = 0.1016, with the scale that matches it exactly). That tensor is 4 A L / H times the size of a hidden state and its
 * mask is rebuilt in the forward, dQ and dK/dV kernels; hashing per pair cost the fused backward 37 us on 60.
 * test / debug: out[i] = multiplier of element i, i < n (probs != 0: the 8-bit form). */
int qst_dropout_multipliers(const QstDrop* d, int probs, int64_t n, float* out, void* stream);

typedef struct {
    const void* A;        // bf16
    const void* B;        // bf16
    void* C;
    void* C2;
    const void* aux;      // bf16, same shape/ld as C (GELU_BWD)
    const float* bias;    // [N] or NULL
    const float* resid;   // f32 [M, ldr] or NULL
    float* colsum;        // tn only: f32 [N] += column sums of A (bias gradient), or NULL
    int32_t M, N, K;
    int32_t lda, ldb, ldc, ldr;
    int32_t splits;       // tn only: reduction splits over M (0 = auto); nt: tile-variant selector (tests / tuning)
    const void* bscale;   // qst_gemm_nt_f8 only: the E8M0 block scales of B (uint8, layout of qst_quant_mx)
    // Dropout (training; drop.thr16 == 0 / drop.state == NULL = none), element index m * N + n of the [M, N] result:
    //   drop_where 1: the projection output, before the residual: C = (acc + bias) * mask + resid   (QST_EPI_F32_RESID*,
    //                 qst_gemm_nt_ln mode 0) -- BertSelfOutput / BertOutput: LayerNorm(dropout(dense(x)) + input)
    //   drop_where 2: qst_gemm_nt_ln mode 1: the bf16 copy C2 = ds * mask (the gradient that enters the dgrad / wgrad of the
    //                 dropped projection; the fp32 C, which continues down the residual path, stays unmasked)
    //   drop_where 3: qst_gemm_nt_ln mode 1: dy = (acc + resid) * mask before the LayerNorm backward (embedding dropout
    //                 sits AFTER its LayerNorm)
    QstDrop drop;
    int32_t drop_where;
    void* C3;             // QST_EPI_GELU_MX_TRAIN only (see the enum)
    void* C4;
    const void* B2;       // qst_gemm_nt / qst_gemm_nt_ln (tiled kernels): C = A . (B + B2)^T -- a second pass over K against B2, same
                          // shape and ldb as B (QST_PREC_F16W: the low halves of split-f16 weights); NULL = none
    int32_t b2_n0;        // with B2: only output tiles that reach columns >= b2_n0 take the second pass (the fused QKV GEMM: the V
                          // third -- rounding of the q / k weights does not reach the pooled embedding, DESIGN.md finding 34); 0 = all
    int32_t sat16;        // f16 twins only (qst_gemm_nt_f16 / qst_gemm_nt8_f16): != 0 = 16-bit outputs saturate at +-65,504
                          // instead of overflowing to inf (forward launches); 0 = IEEE overflow (backward launches)
} QstGemmArgs;

/* C[M,N] = A[M,K] . B[N,K]^T with epilogue `epi`. K % 64 == 0, lda/ldb % 8 == 0. */
int qst_gemm_nt(const QstGemmArgs* a, int epi, void* stream);
/* NT GEMM on the fp8 matrix cores, both operands MXFP8 (OCP e4m3 elements + one E8M0 power-of-two scale per 32
 * consecutive K elements of a row; v_mfma_scale_f32_32x32x64_f8f6f4): A = e4m3 [M, K] (lda bytes), aux = its scales;
 * B = e4m3 [N, K] (ldb bytes), bscale = its scales (uint8 arrays in the layout qst_quant_mx writes). K % 128 == 0.
 * epi: QST_EPI_BF16, QST_EPI_F32_RESID (with `drop` / drop_where 1: dropout of the projection output, as qst_gemm_nt),
 * QST_EPI_GELU_MX (inference), QST_EPI_GELU / QST_EPI_GELU_MX_TRAIN (fp8 training forward: gelu'(u) and h as bf16, what the
 * bf16 backward reads; the latter with h as MXFP8 too). QST_PREC_FP8, BASELINE configs[4]. */
int qst_gemm_nt_f8(const QstGemmArgs* a, int epi, void* stream);
/* MXFP8 quantisation of a contiguous [rows, K] matrix (src f32, or bf16 when src_is_bf16): per 32-element block the
 * scale exponent e = the smallest with amax * 2^-e <= 448 (-127 for an all-zero block), elements = RNE(x * 2^-e) to
 * e4m3; q = uint8 [rows, K]; scales = uint8 holding e + 127, STAGE-MAJOR: block kb of row r at
 * ((kb / 4) * rows + r) * 4 + kb % 4, ceil(K / 128) * rows * 4 bytes in all -- the four scales a GEMM stage (128 K) needs
 * from a row are one aligned dword, and 32 consecutive rows one 128-byte line. K % 32 == 0. */
int qst_quant_mx(const void* src, int src_is_bf16, int64_t rows, int K, void* q, void* scales, void* stream);
/* NT GEMM with a LayerNorm fused into the epilogue (N = 384: one 128 x 384 tile spans whole rows; N = 512 / 768 / 1024:
 * qst_gemm_nt8_ln below, which this entry point forwards to; qst_gemm_nt_ln_supported(N) tells). xhat is bf16 [M, N]
 * contiguous, rstd f32 [M].
 *  mode 0 (forward):  v = A.B^T + bias + resid ; y = LayerNorm(v) -> C (f32), C2 (bf16, nullable);
 *                     xhat, rstd (nullable) are written for the backward pass.
 *  mode 1 (backward): dy = A.B^T + resid ; ds = rstd*(g*dy - mean(g*dy) - xhat*mean(g*dy*xhat)) -> C (f32),
 *                     C2 (bf16, nullable); partials (nullable) f32 [ceil(M / qst_gemm_nt_ln_block_rows_m(N, M))][2][N] receives each tile's
 *                     sum(dy*xhat) and sum(dy) rows (reduce with qst_ln_bwd_reduce_batch). */
typedef struct {
    const float* gamma;
    const float* beta;
    float eps;
    void* xhat;
    float* rstd;
    float* partials;
} QstLnEpi;
int qst_gemm_nt_ln_supported(int N);
int qst_gemm_nt_ln_block_rows(int N);    /* rows per `partials` block of mode 1: 128 at N = 384; above, see the _m form */
int qst_gemm_nt_ln_block_rows_m(int N, int M);   /* ... exactly: above N = 384 the tile (256 x 256 or 128 x 384) depends on M */
int qst_gemm_nt_ln(const QstGemmArgs* a, const QstLnEpi* ln, int mode, void* stream);
/* The same with rows wider than one tile, N = 512 / 768 / 1024 (csrc/gemm8.hip: 256 x 256 tiles on the 8-phase loop, the
 * N / 256 workgroups of a 256-row panel exchange the row statistics inside the launch); qst_gemm_nt_ln forwards here.
 * partials f32 [ceil(M / qst_gemm_nt8_ln_block_rows(M, N))][2][N]; no B2. qst_gemm_nt8_ln_timeouts(): 0 unless an exchange of this process ever gave up. */
int qst_gemm_nt8_ln_supported(int N);
int qst_gemm_nt8_ln_block_rows(int M, int N);   /* 256, or 128 where the 128 x 384 tile is taken (N = 768, 32,768 <= M < 43,691) */
int qst_gemm_nt8_ln(const QstGemmArgs* a, const QstLnEpi* ln, int mode, void* stream);
int qst_gemm_nt8_ln_timeouts(void);
/* Mode 0 of the same on the fp8 matrix cores: operands as qst_gemm_nt_f8 (A, B e4m3; a->aux / a->bscale their E8M0 scales;
 * K % 128 == 0, ldc == N); outputs as above plus, when a->C3 / a->C4 are given, the normalised rows as MXFP8 (e4m3 [M, N] +
 * scales in qst_quant_mx's layout, quantised from the 16-bit-rounded values as qst_ln_fwd_mx_train does). QST_PREC_FP8 forward. */
int qst_gemm_nt8_f8_ln(const QstGemmArgs* a, const QstLnEpi* ln, void* stream);
/* The feed-forward block of a layer as one kernel (H = 384 token rows complete per tile; csrc/ffn.hip):
 *  mode 0 (forward):  u = A.B1^T + bias1 ; h = gelu(u) ; v = h.B2^T + bias2 + resid ; y = LayerNorm(v) -> C (f32),
 *                     C2 (bf16, nullable), ln->xhat / ln->rstd (nullable). save_gp / save_h (both or neither) receive
 *                     gelu'(u) and h as bf16 [M, I] for the backward pass; NULL = inference, the [M, I] tensor never
 *                     leaves the chip.
 *  mode 1 (backward): du = (A.B1^T) * aux -> save_h (bf16 [M, I], required: the weight gradients read it) ;
 *                     dy = du.B2^T + resid ; C / C2 / ln->partials as qst_gemm_nt_ln mode 1.
 * A bf16 [M, H]; B1 bf16 [I, H]; B2 bf16 [H, I]; aux bf16 [M, I]; resid f32 [M, H]. All contiguous. */
typedef struct {
    const void* A;
    const void* B1;
    const void* B2;
    const float* bias1;
    const float* bias2;
    const float* resid;
    const void* aux;
    void* save_gp;
    void* save_h;
    float* C;
    void* C2;
    int32_t M, H, I;
    int32_t diag;          /* 0. Timing experiments only (tools/ffn_diag.py): 1 = drop the A loads, 2 = drop the weight loads
                              (both give wrong results), 4 = no L2 touch-ahead of the next weight slice */
} QstFfnArgs;
int qst_ffn_chain_supported(int H, int I);
int qst_ffn_chain(const QstFfnArgs* a, const QstLnEpi* ln, int mode, void* stream);
/* C[N,K] (f32, atomically accumulated) += A[M,N]^T . B[M,K]; colsum[N] += sum_m A[m,:]. */
int qst_gemm_tn(const QstGemmArgs* a, void* stream);
/* Several such products over the same M in ONE launch (all weight gradients of a layer). */
#define QST_TN_MAX_PROB 8
typedef struct {
    int32_t nprob;
    int32_t splits;               /* reduction ranges over M, rounded up to a multiple of 8; 0 = auto */
    int32_t total_tiles;          /* filled by the library */
    int32_t ranges_per_xcd;       /* filled by the library */
    int32_t tiles[QST_TN_MAX_PROB];
    QstGemmArgs prob[QST_TN_MAX_PROB];
} QstTnGroup;
int qst_gemm_tn_group(const QstTnGroup* grp, void* stream);

/* The same GEMMs on the 8-wave, 8-phase K loop (csrc/gemm8p.h, csrc/gemm8.hip): one 512-thread workgroup per CU, 128 KB of
 * LDS, K-tiles of 64 with three half-tiles of LDS-DMA in flight across the barriers. qst_gemm_nt / qst_gemm_tn_group take
 * this path by themselves where it is faster (long reductions); these entry points force it (tests, tools).
 * qst_gemm_nt8: tile 0 = 128 x 384 (8 waves of 64 x 96), 1 = 256 x 256 (8 waves of 128 x 64); needs K % 64 == 0,
 * N % 8 == 0, lda / ldb / ldc % 8 == 0 (qst_gemm_nt8_supported tells). Epilogues and dropout as qst_gemm_nt.
 * qst_gemm8_mode(mode): -1 = the library chooses per call (default); otherwise bit 0 = every supported NT GEMM, bit 1 =
 * every weight-gradient launch on this path, 0 = none. Returns the previous mode; mode < -1 only reads it. Process-wide. */
/* qst_gemm8_stagger(cycles): the first round of workgroups of an 8-phase NT launch (more than two tiles per CU) starts spread
 * over `cycles` clock cycles, so that the CUs do not all store at the same time; 0 = together; -1 (default) = the library's
 * choice (together for the plain GEMMs, 40,000 cycles for the GEMM + LayerNorm launches). Returns the previous value; an
 * argument below -1 only reads. Process-wide. */
int qst_gemm8_stagger(int cycles);
/* fp32 row stores of the GEMM + LayerNorm launches: 0 = 64 contiguous bytes per row and instruction, 1 = 16-byte fragments,
 * -1 (default) = by tile (fragments on the 256 x 256 tile, whose shapes stream from HBM; contiguous on 128 x 384). */
int qst_gemm8_ln_store(int mode);
int qst_gemm_nt8_supported(const QstGemmArgs* a, int epi);
int qst_gemm_nt8(const QstGemmArgs* a, int epi, int tile, void* stream);
int qst_gemm_tn8_group(const QstTnGroup* grp, void* stream);
/* qst_gemm_nt_f8's operands (MXFP8 x MXFP8, block scales in qst_quant_mx's layout) on the 8-phase loop with
 * v_mfma_scale_f32_16x16x128_f8f6f4: epi QST_EPI_BF16, QST_EPI_F32_RESID (+ dropout), QST_EPI_GELU; tile as qst_gemm_nt8. */
int qst_gemm_nt8_f8(const QstGemmArgs* a, int epi, int tile, void* stream);
int qst_gemm8_mode(int mode);

/* Embedding gather + LayerNorm (BertEmbeddings / MPNetEmbeddings forward).
 * pos_ids: int32 [M] position row per token. type_emb may be NULL. Outputs: y f32, y bf16, xhat bf16, rstd f32. */
int qst_embed_ln_fwd(const int64_t* ids, const int64_t* type_ids, const int32_t* pos_ids,
                     const float* word_emb, const float* pos_emb, const float* type_emb,
                     const float* gamma, const float* beta, float eps, int M, int H,
                     float* y, void* y_bf16, void* xhat_bf16, float* rstd, void* stream);
/* ... followed by dropout of y / y_bf16 (BertEmbeddings: dropout(LayerNorm(e))); xhat / rstd describe the undropped row. */
int qst_embed_ln_fwd_drop(const int64_t* ids, const int64_t* type_ids, const int32_t* pos_ids,
                          const float* word_emb, const float* pos_emb, const float* type_emb,
                          const float* gamma, const float* beta, float eps, int M, int H,
                          float* y, void* y_bf16, void* xhat_bf16, float* rstd, const QstDrop* drop, void* stream);
/* LayerNorm over rows of s f32 [M,H]. */
int qst_ln_fwd(const float* s, const float* gamma, const float* beta, float eps, int M, int H,
               float* y, void* y_bf16, void* xhat_bf16, float* rstd, void* stream);
/* The same two kernels with the output ALSO as MXFP8 (yq e4m3 [M, H], ys scales in qst_quant_mx's layout), quantised
 * from the bf16-rounded values -- what qst_quant_mx over y_bf16 would give, without the extra pass. y_bf16 may be NULL.
 * H % 64 == 0. QST_PREC_FP8 forward. */
int qst_ln_fwd_mx(const float* s, const float* gamma, const float* beta, float eps, int M, int H,
                  float* y, void* y_bf16, void* yq, void* ys, void* stream);
int qst_embed_ln_fwd_mx(const int64_t* ids, const int64_t* type_ids, const int32_t* pos_ids,
                        const float* word_emb, const float* pos_emb, const float* type_emb,
                        const float* gamma, const float* beta, float eps, int M, int H,
                        float* y, void* y_bf16, void* yq, void* ys, void* stream);
/* ... and with the bf16 copy, xhat and rstd the bf16 backward reads (fp8 TRAINING forward: QST_PREC_FP8 with training != 0) */
int qst_ln_fwd_mx_train(const float* s, const float* gamma, const float* beta, float eps, int M, int H, float* y,
                        void* y_bf16, void* xhat_bf16, float* rstd, void* yq, void* ys, void* stream);
int qst_embed_ln_fwd_mx_train(const int64_t* ids, const int64_t* type_ids, const int32_t* pos_ids,
                              const float* word_emb, const float* pos_emb, const float* type_emb,
                              const float* gamma, const float* beta, float eps, int M, int H, float* y,
                              void* y_bf16, void* xhat_bf16, float* rstd, void* yq, void* ys,
                              const QstDrop* drop /* nullable: dropout of the output, as qst_embed_ln_fwd_drop */, void* stream);
/* LayerNorm backward: ds = rstd*(g*dy - mean(g*dy) - xhat*mean(g*dy*xhat)); dgamma += sum dy*xhat; dbeta += sum dy. */
/* scratch: qst_ln_bwd_scratch_bytes(M, H) of per-block partial sums reduced in a fixed order (deterministic);
 * NULL falls back to float atomics on dgamma/dbeta. */
size_t qst_ln_bwd_scratch_bytes(int M, int H);
int qst_ln_bwd(const float* dy, const void* xhat_bf16, const float* rstd, const float* gamma, int M, int H,
               float* ds, void* ds_bf16, float* dgamma, float* dbeta, float* scratch, void* stream);
/* The same with dropout masks recomputed: drop_in (nullable) multiplies dy first (a dropout that followed this
 * LayerNorm: the embeddings); drop_out (nullable) multiplies ds_bf16 only (the dropout that preceded the residual add in
 * front of this LayerNorm: ds_bf16 is the gradient of the dropped projection output, ds of the residual). */
int qst_ln_bwd_drop(const float* dy, const void* xhat_bf16, const float* rstd, const float* gamma, int M, int H,
                    float* ds, void* ds_bf16, float* dgamma, float* dbeta, float* scratch,
                    const QstDrop* drop_in, const QstDrop* drop_out, void* stream);
/* Deferred second stage: qst_ln_bwd called with dgamma = dbeta = NULL only writes its partials into `scratch`;
 * this reduces up to QST_LN_BATCH_MAX such buffers (same M, H) into their dgamma/dbeta in one launch. */
#define QST_LN_BATCH_MAX 32
typedef struct {
    int32_t count, H, nblocks;      /* nblocks = qst_ln_bwd_scratch_bytes(M, H) / (2 * H * 4) */
    const float* partials[QST_LN_BATCH_MAX];
    float* dgamma[QST_LN_BATCH_MAX];
    float* dbeta[QST_LN_BATCH_MAX];
    int32_t nblocks_each[QST_LN_BATCH_MAX];   /* per-entry row count of partials; 0 = nblocks */
} QstLnReduceBatch;
int qst_ln_bwd_reduce_batch(const QstLnReduceBatch* b, void* stream);
/* Embedding backward: ds f32 [nseq*L, H] rows are ADDED into the word / position / token-type gradient tables (float
 * atomics for the word rows; one pass over ds). */
int qst_embed_bwd(const float* ds, const int64_t* ids, const int64_t* type_ids, const int32_t* pos_ids,
                  int nseq, int L, int H, int num_types, float* dword, float* dpos, float* dtype_, void* stream);
/* MPNet position ids (cumsum of non-pad) or BERT arange -> int32 [nseq*L]. */
int qst_position_ids(const int64_t* ids, int nseq, int L, int arch, int pad_id, int32_t* pos_ids, void* stream);
/* The same launch as the whole prologue of a training forward with dropout: it also advances the dropout step counter
 * (qst_dropout_advance) and copies the four state words to drop_snapshot (kept beside that forward's activations; the
 * backward rebuilds its masks from it). drop_state and drop_snapshot: both device pointers, or both NULL. */
int qst_forward_prologue(const int64_t* ids, int nseq, int L, int arch, int pad_id, int32_t* pos_ids,
                         uint32_t* drop_state, uint32_t* drop_snapshot, void* stream);

/* ST Pooling(mean) + optional Normalize. tok f32 [nseq,L,H]; pooled f32 [nseq,H] (pre-normalize, saved). */
int qst_pool_norm_fwd(const float* tok, const int64_t* mask, int nseq, int L, int H, int normalize,
                      float* emb, float* pooled, void* stream);
int qst_pool_norm_bwd(const float* demb, const float* pooled, const int64_t* mask, int nseq, int L, int H,
                      int normalize, float* dtok, void* stream);

/* Self-attention forward: qkv bf16 [nseq*L, 3H] token-major (q | k | v, heads concatenated),
 * mask int64 [nseq, L], rel_pos f32 [A, 2L] (qst_rel_pos_fwd: bias of relative position j - i at entry j - i + L)
 * or NULL -> ctx bf16 [nseq*L, H], lse f32 [nseq, A, L]. */
int qst_attention_fwd(const void* qkv, const int64_t* mask, const float* rel_pos, int nseq, int L, int A, int d,
                      void* ctx, float* lse, void* stream);
/* Backward: dctx bf16 [nseq*L, H] -> dqkv bf16 [nseq*L, 3H]; drel_pos f32 [A, 2L] += (or NULL; needs rel_pos).
 * delta_scratch: f32 [nseq, A, L] (dO.O per query, written by the dQ kernel, read by the dK/dV kernel). */
int qst_attention_bwd(const void* qkv, const void* ctx, const void* dctx, const float* lse, const int64_t* mask,
                      const float* rel_bias, int nseq, int L, int A, int d, void* dqkv, float* drel,
                      float* delta_scratch, void* stream);
/* The same two kernels taking a description; drop: dropout of the softmax probabilities (BertSelfAttention: dropout(softmax(s)) . v), element index
 * ((seq * A + head) * L + query) * L + key. lse stays the log-sum-exp of the undropped scores. */
typedef struct {
    const void* qkv; const int64_t* mask; const float* rel_pos;
    int32_t nseq, L, A, d;
    void* ctx; float* lse;                       /* forward: outputs; backward: inputs */
    const void* dctx; void* dqkv; float* drel; float* delta_scratch;       /* backward only */
    int32_t force_split;   /* backward: 0 = the library chooses (a one-workgroup-per-(sequence, head) kernel where there is one: d = 32 at
                              L <= 128; d = 64 wherever its LDS fits, i.e. L <= 512, or <= 384 with the position bias); 1 = the
                              dQ + dK/dV kernel pair; 2 = the d = 64 one-workgroup kernel (same as 0 today; tests name it) */
    QstDrop drop;
} QstAttnDesc;
int qst_attention_fwd_ex(const QstAttnDesc* a, void* stream);
int qst_attention_bwd_ex(const QstAttnDesc* a, void* stream);

/* MPNet relative position bias: rel_bias[a, i, j] = table[lut[(j - i) + 511]][a]; lut = int32 [1023] device
 * table of qst_rel_bucket_host(j - i). Backward accumulates drel [A, L, L] into dtable [buckets, A]. */
int qst_rel_bucket_host(int rel, int num_buckets, int max_distance);
int qst_rel_bias_fwd(const float* table, const int32_t* lut, int A, int L, float* rel_bias, void* stream);
int qst_rel_bias_bwd(const float* drel, const int32_t* lut, int buckets, int A, int L, float* dtable, void* stream);
/* The same bias as relative-position vectors [A][2L] (entry j - i + L; entry 0 unused): the form the bf16 attention
 * kernels take (each workgroup keeps its head's 2L values in LDS instead of gathering from an [A, L, L] table), and
 * the reduction of their gradient back onto the bucket table. */
int qst_rel_pos_fwd(const float* table, const int32_t* lut, int A, int L, float* rel_pos, void* stream);
int qst_rel_pos_bwd(const float* drel_pos, const int32_t* lut, int buckets, int A, int L, float* dtable, void* stream);

/* Parity-precision (QST_PREC_BF16X3) forward kernels: fp32 operands split into hi+lo bf16 on the fly, three MFMAs
 * per product, fp32 out. epi: 0 = +bias, 1 = +bias +resid, 2 = gelu(+bias), 3 = C += A . B^T with the reduction shared among
 * workgroups (fp32 atomics; no bias), 4 = C = +bias and C2 = gelu(C) (both fp32, ldc), 5 = C = (A . B^T) * gelu'(aux) with aux
 * fp32 in the layout of C. K % 32 == 0. */
int qst_gemm_nt_x3(const QstGemmArgs* a, int epi, void* stream);
/* Weight gradient at parity precision, the argument roles of qst_gemm_tn: C[N, K] (f32) += A[M, N]^T . B[M, K] with fp32
 * row-major operands (A = dY, B = the layer input; M = token rows, M % 32 == 0), colsum[n] += sum_m A[m, n] when colsum is
 * set. The reduction over M is shared among workgroups (about a->splits of them when splits > 0, 256 otherwise); partial
 * tiles meet in C through fp32 atomics. */
int qst_gemm_tn_x3(const QstGemmArgs* a, void* stream);
int qst_attention_fwd_x3(const float* qkv, const int64_t* mask, const float* rel_bias, int nseq, int L, int A, int d,
                         float* ctx, void* stream);
/* ... with dropout of the probabilities (drop nullable; QST_DROP_SITE_PROBS, the 8-bit form): parity-precision TRAINING */
int qst_attention_fwd_x3_drop(const float* qkv, const int64_t* mask, const float* rel_bias, int nseq, int L, int A, int d,
                              float* ctx, const QstDrop* drop, void* stream);
/* Parity-precision BACKWARD building blocks (csrc/x3_bwd.hip; the contractions themselves are qst_gemm_nt_x3 calls on
 * transposed copies). All tensors fp32, contiguous unless a leading dimension is given.
 *   qst_transpose_f32    : dst[C, R] (ld_dst) = src[R, C]^T (ld_src)
 *   qst_gelu_f32 / qst_gelu_bwd_f32 : h = gelu(u), du = dh * gelu'(u) (exact erf), n elements
 *   qst_colsum_f32       : out[n] += sum_m x[m, n]
 *   qst_embed_sum_f32    : s[m, :] = (word[ids[m]] + type[type_ids[m]]) + pos[pos_ids[m]]  (pre-norm embedding sum)
 *   qst_ln_bwd_f32       : LayerNorm backward from the pre-norm rows (mean / rstd recomputed): ds; dgamma, dbeta += (atomics)
 *   qst_attention_bwd_f32: qkv [nseq*L, 3H], ctx and dctx [nseq*L, H], rel_bias [A, L, L] or NULL -> dqkv [nseq*L, 3H];
 *                          drel_bias [A, L, L] += (or NULL). L <= 512, d in {32, 64}. */
int qst_transpose_f32(const float* src, int R, int C, int ld_src, float* dst, int ld_dst, void* stream);
int qst_gelu_f32(const float* u, int64_t n, float* h, void* stream);
int qst_gelu_bwd_f32(const float* dh, const float* u, int64_t n, float* du, void* stream);
int qst_colsum_f32(const float* x, int M, int N, int ld, float* out, void* stream);
int qst_embed_sum_f32(const int64_t* ids, const int64_t* type_ids, const int32_t* pos_ids, const float* word_emb,
                      const float* pos_emb, const float* type_emb, int M, int H, float* s, void* stream);
int qst_ln_bwd_f32(const float* dy, const float* prenorm, const float* gamma, float eps, int M, int H, float* ds,
                   float* dgamma, float* dbeta, void* stream);
int qst_attention_bwd_f32(const float* qkv, const float* ctx, const float* dctx, const int64_t* mask, const float* rel_bias,
                          int nseq, int L, int A, int d, float* dqkv, float* drel_bias, void* stream);
int qst_attention_bwd_f32_drop(const float* qkv, const float* ctx, const float* dctx, const int64_t* mask, const float* rel_bias,
                               int nseq, int L, int A, int d, float* dqkv, float* drel_bias, const QstDrop* drop, void* stream);
/* The same gradients on the matrix cores (csrc/x3.hip): every contraction as three split-bf16 MFMAs per product, softmax,
 * dropout mask and dS in fp32; agrees with qst_attention_bwd_f32 to fp32 rounding (tests/test_gpu_encoder.py). `scratch`:
 * qst_attention_bwd_x3_scratch_bytes(nseq, L, A) bytes (row maxima, 1 / row sums and dO . O of every query). L % 32 == 0,
 * L <= 512, d in {32, 64}. This is what the QST_PREC_BF16X3 backward of qst_encoder_backward runs. */
size_t qst_attention_bwd_x3_scratch_bytes(int nseq, int L, int A);
int qst_attention_bwd_x3(const float* qkv, const float* ctx, const float* dctx, const int64_t* mask, const float* rel_bias,
                         int nseq, int L, int A, int d, float* dqkv, float* drel_bias, void* scratch, const QstDrop* drop,
                         void* stream);
/* out[i] = in[i] * mask multiplier(i) (+ resid[i], nullable), i < n (n % 4 == 0; in may be out): the hidden-state dropout of
 * the parity-precision training path (element index = flat index, the 16-bit generator of QstDrop). */
int qst_dropout_apply_f32(const QstDrop* d, const float* in, const float* resid, int64_t n, float* out, void* stream);

/* k best entries of every row of scores f32 [nrows, ld] (first n columns), sorted by descending score (ties: ascending
 * index). index_map (nullable, int64, same ld) translates column numbers into caller ids -- used to merge the
 * per-chunk results of qst_topk_scores. k <= min(n, 1024). */
int qst_topk_rows(const float* scores, int64_t ld, const int64_t* index_map, int nrows, int n, int k,
                  float* out_scores, int64_t* out_index, void* stream);

/* All GEMM weights of the arena in one launch; table_dev = int64 [nseg][6] {src off, rows, cols, dst off, dstT off,
 * first block} (built by qst_encoder_create). */
int qst_shadow_all(const float* params, void* shadow, const int64_t* table_dev, int nseg, int nblocks, void* stream);
/* bf16 shadow: dst[i] = bf16(src[i]) and dstT = transpose for a [rows, cols] matrix. */
int qst_shadow_matrix(const float* src, int rows, int cols, void* dst_bf16, void* dstT_bf16, void* stream);

/* ---- f16-operand twins (QST_PREC_F16, round 5) ----
 * Every entry point above whose matrix-core operands or 16-bit outputs are bf16 exists a second time with IEEE half in
 * their place: same arguments, same layouts, `void*` 16-bit tensors holding f16 instead of bf16 (the sources are compiled
 * twice on one operand type, csrc/qst_common.h: op16). v_mfma_f32_32x32x16_f16 / 16x16x32_f16 issue at the bf16 rate on
 * the same bytes; f16 carries 11 significand bits against bf16's 8, at the price of a 5-bit exponent: forward kernels
 * saturate at +-65,504 (QstGemmArgs.sat16 for qst_gemm_nt_f16 / qst_gemm_nt8_f16; the LayerNorm, GELU, fused-LayerNorm mode 0,
 * feed-forward-chain mode 0 and attention-forward kernels always), backward kernels keep IEEE inf so that an overflowed
 * gradient shows up in the global norm (qst_clip_adamw_step_amp). The reference's counterpart is torch.cuda.amp.autocast
 * (fp16) + GradScaler: /root/reference/training/main.py:142, models/evaluators.py:92-94. */
int qst_gemm_nt_f16(const QstGemmArgs* a, int epi, void* stream);
int qst_gemm_nt_ln_f16(const QstGemmArgs* a, const QstLnEpi* ln, int mode, void* stream);
int qst_gemm_nt8_ln_supported_f16(int N);
int qst_gemm_nt8_ln_block_rows_f16(int M, int N);
int qst_gemm_nt8_ln_f16(const QstGemmArgs* a, const QstLnEpi* ln, int mode, void* stream);
int qst_gemm_nt8_ln_timeouts_f16(void);
int qst_ffn_chain_f16(const QstFfnArgs* a, const QstLnEpi* ln, int mode, void* stream);
int qst_gemm_tn_f16(const QstGemmArgs* a, void* stream);
int qst_gemm_tn_group_f16(const QstTnGroup* grp, void* stream);
int qst_gemm_nt8_supported_f16(const QstGemmArgs* a, int epi);
int qst_gemm_nt8_f16(const QstGemmArgs* a, int epi, int tile, void* stream);
int qst_gemm_tn8_group_f16(const QstTnGroup* grp, void* stream);
int qst_embed_ln_fwd_f16(const int64_t* ids, const int64_t* type_ids, const int32_t* pos_ids,
                         const float* word_emb, const float* pos_emb, const float* type_emb,
                         const float* gamma, const float* beta, float eps, int M, int H,
                         float* y, void* y_f16, void* xhat_f16, float* rstd, void* stream);
int qst_embed_ln_fwd_drop_f16(const int64_t* ids, const int64_t* type_ids, const int32_t* pos_ids,
                              const float* word_emb, const float* pos_emb, const float* type_emb,
                              const float* gamma, const float* beta, float eps, int M, int H,
                              float* y, void* y_f16, void* xhat_f16, float* rstd, const QstDrop* drop, void* stream);
int qst_ln_fwd_f16(const float* s, const float* gamma, const float* beta, float eps, int M, int H,
                   float* y, void* y_f16, void* xhat_f16, float* rstd, void* stream);
int qst_ln_bwd_f16(const float* dy, const void* xhat_f16, const float* rstd, const float* gamma, int M, int H,
                   float* ds, void* ds_f16, float* dgamma, float* dbeta, float* scratch, void* stream);
int qst_ln_bwd_drop_f16(const float* dy, const void* xhat_f16, const float* rstd, const float* gamma, int M, int H,
                        float* ds, void* ds_f16, float* dgamma, float* dbeta, float* scratch,
                        const QstDrop* drop_in, const QstDrop* drop_out, void* stream);
int qst_attention_fwd_f16(const void* qkv, const int64_t* mask, const float* rel_pos, int nseq, int L, int A, int d,
                          void* ctx, float* lse, void* stream);
int qst_attention_bwd_f16(const void* qkv, const void* ctx, const void* dctx, const float* lse, const int64_t* mask,
                          const float* rel_bias, int nseq, int L, int A, int d, void* dqkv, float* drel,
                          float* delta_scratch, void* stream);
int qst_attention_fwd_ex_f16(const QstAttnDesc* a, void* stream);
int qst_attention_bwd_ex_f16(const QstAttnDesc* a, void* stream);
int qst_shadow_all_f16(const float* params, void* shadow, const int64_t* table_dev, int nseg, int nblocks, void* stream);
int qst_shadow_matrix_f16(const float* src, int rows, int cols, void* dst_f16, void* dstT_f16, void* stream);
/* QST_PREC_F16W: qst_shadow_all_f16 plus the low halves of the split weights, lo = f16(w - f16(w)), written at the offsets of the
 * W copies in shadow_lo, an arena of the same layout as `shadow` (W = hi + lo to about 2^-22 relative; the halves are mostly
 * subnormal, which v_mfma_f32_*_f16 keeps). The forward GEMMs take them as QstGemmArgs.B2. */
int qst_shadow_all_split_f16(const float* params, void* shadow, void* shadow_lo, const int64_t* table_dev, int nseg, int nblocks,
                             void* stream);

/* sizeof() of the argument structs above as this library was compiled, for bindings to check their mirror of the layout:
 * which = 0 QstGemmArgs, 1 QstLnEpi, 2 QstFfnArgs, 3 QstTnGroup, 4 QstLnReduceBatch, 5 QstDrop, 6 QstAttnDesc. */
int64_t qst_abi_sizeof(int which);

#ifdef __cplusplus
}
#endif
