/*
 * qst.h -- C-ABI of libqst.so: the MI355X (gfx950) quadruplet fine-tuning hot path.
 *
 * This is the drop-in boundary. The reference is pure
 * Python with no FFI of its own, so each entry point names the reference
 * interface whose arithmetic it replaces. All pointers are DEVICE pointers
 * unless the parameter name ends in _host; the library never allocates, frees
 * or retains caller memory (handles own only small device tables that depend
 * on the config). Every call enqueues on `stream` (a hipStream_t passed as
 * void*) and returns without a host sync. Return value: 0 = OK, negative =
 * qst_status (see qst_strerror). Nothing throws or aborts.
 *
 * Numeric contract: fp32 parameters, fp32 residual stream / LayerNorm /
 * softmax / pooling / loss; GEMM and attention contractions take bf16 MFMA
 * operands with fp32 accumulation. precision = QST_PREC_BF16X3 splits every
 * fp32 operand into hi+lo bf16 and issues three MFMAs (fp32-class accuracy,
 * the parity mode); QST_PREC_BF16 rounds operands once (the throughput mode);
 * QST_PREC_FP8 (BASELINE configs[4] "fp8 weights ... CDNA4 fp8 MFMA GEMMs"; inference, and with training != 0 the forward
 * of a training step whose backward is the bf16 one: qst_encoder_backward* on this handle with the bf16 shadows) runs every Linear on the fp8 matrix
 * cores: weights AND activations as OCP MXFP8 (e4m3 elements, one E8M0 scale per 32 input features;
 * v_mfma_scale_f32_32x32x64_f8f6f4, fp32 accumulation), attention in bf16, residual stream / LayerNorm in fp32.
 * QST_PREC_F16 (round 5) is QST_PREC_BF16 with IEEE half in bf16's place: the same kernels compiled on the other 16-bit
 * operand type (v_mfma_f32_32x32x16_f16: same issue rate, same bytes), 11 significand bits instead of 8 -- embeddings
 * within the north-star tolerance (rtol 1e-3 / atol 1e-4) of the fp32 reference for the six-layer models, where bf16's
 * are not -- and a 5-bit exponent: forward kernels saturate at +-65,504, and a training step runs under a loss scale
 * (qst_amp_scaler_init / qst_clip_adamw_step_amp), which is what the reference's own reduced precision does:
 * torch.cuda.amp.autocast + GradScaler, /root/reference/training/main.py:142 (`use_amp`), models/evaluators.py:92-94.
 * QST_PREC_F16W is QST_PREC_F16 with SPLIT WEIGHTS in the forward: every Linear multiplies its f16 activations by hi + lo
 * of the weight (two f16 values, W to ~2^-22), as a second pass over K in the same kernel -- the rounding of the weights is
 * the same for every token of a sequence and does not average out in the pooled embedding, the rounding of the activations
 * does: measured, it is 85-95% of plain f16's embedding error. With it the twelve-layer mpnet-base case is inside the
 * north-star tolerance too. Backward as QST_PREC_F16 (gradients need no more). Its shadow is qst_shadow_elems() long: the f16
 * arena and behind it a second one with the low halves.
 * The shadow of a QST_PREC_F16 handle holds IEEE half (qst_refresh_shadow on that handle), its activation arena f16
 * tensors of the bf16 arena's sizes; its backward is refused on another precision's arena and vice versa.
 */
#ifndef QST_H
#define QST_H

#include <stddef.h>
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

typedef enum {
    QST_OK = 0,
    QST_ERR_BAD_ARG = -1,      /* null pointer, non-positive size, bad enum           */
    QST_ERR_UNSUPPORTED = -2,  /* dims the kernels are not built for (see qst_encoder_create) */
    QST_ERR_WORKSPACE = -3,    /* workspace/saved arena smaller than qst_*_bytes() says */
    QST_ERR_HIP = -4,          /* a HIP runtime call failed; qst_last_hip_error() has the code */
    QST_ERR_NO_DEVICE = -5,
    QST_ERR_COMM = -6,         /* RCCL could not be loaded, or one of its calls failed; qst_comm_last_error() has the text */
    QST_ERR_NO_FORWARD = -7    /* backward over an arena that no training forward of this precision and shape has filled */
} qst_status;

enum { QST_ARCH_BERT = 0, QST_ARCH_MPNET = 1 };
enum { QST_PREC_BF16 = 0, QST_PREC_BF16X3 = 1, QST_PREC_FP8 = 3, QST_PREC_F16 = 4, QST_PREC_F16W = 5 };   /* 2 was an fp8-weights-only mode (removed) */
enum { QST_REDUCE_NONE = 0, QST_REDUCE_SUM = 1, QST_REDUCE_MEAN = 2 };

/* Encoder architecture. Mirrors HF BertConfig / MPNetConfig fields that the
 * reference selects by model name (training/main.py:114,242). */
typedef struct {
    int32_t arch;              /* QST_ARCH_*                                           */
    int32_t vocab_size;
    int32_t hidden_size;       /* H; multiple of 64                                    */
    int32_t num_layers;
    int32_t num_heads;         /* H / num_heads must be 32 or 64                       */
    int32_t intermediate_size; /* I; multiple of 64                                    */
    int32_t max_position;
    int32_t type_vocab_size;   /* 0 for MPNet                                          */
    float   layer_norm_eps;
    int32_t normalize;         /* ST Normalize module present                          */
    int32_t rel_buckets;       /* MPNet relative_attention_num_buckets (32)            */
    int32_t rel_max_distance;  /* MPNet (128)                                          */
    int32_t pad_token_id;      /* MPNet position ids (1)                               */
    int32_t precision;         /* QST_PREC_*                                           */
} qst_config;

typedef struct qst_encoder qst_encoder;

const char* qst_strerror(int status);
int  qst_last_hip_error(void);
int  qst_version(void);

/* ---- parameter arena layout (contract with config.build_layout on the Python side) ---- */
/* Total elements of the flat fp32 arena (params, grads, exp_avg, exp_avg_sq all share it). */
int64_t qst_arena_elems(const qst_config* cfg);
/* Number of segments and per-segment description. name_out receives a static string. */
int     qst_arena_num_segments(const qst_config* cfg);
int     qst_arena_segment(const qst_config* cfg, int idx, const char** name_out, int64_t* offset_out,
                          int64_t* numel_out, int32_t* decay_out, int32_t* gemm_out);
/* Elements of the bf16 shadow arena: [W | W^T] copies of every GEMM weight. */
int64_t qst_shadow_elems(const qst_config* cfg);

/* Bytes of the MXFP8 weight shadow of QST_PREC_FP8: per GEMM weight the e4m3 matrix and its E8M0 block scales. */
int64_t qst_shadow8_bytes(const qst_config* cfg);

/* ---- encoder handle ---- */
int  qst_encoder_create(const qst_config* cfg, qst_encoder** out);
void qst_encoder_destroy(qst_encoder* enc);

/* Bytes of the activation arena `saved` for nseq sequences of length L.
 * training != 0 also reserves what backward needs. */
size_t qst_encoder_saved_bytes(const qst_encoder* enc, int nseq, int L, int training);
/* Bytes of scratch for backward (gradient activations). */
size_t qst_encoder_bwd_workspace_bytes(const qst_encoder* enc, int nseq, int L);

/* Refresh the bf16 shadows (W and W^T of every GEMM weight) from the fp32 arena.
 * Must be called after any parameter update and before forward/backward. */
int qst_refresh_shadow(const qst_encoder* enc, const float* params, void* shadow_bf16, void* stream);
/* QST_PREC_FP8: a buffer of qst_shadow8_bytes holds every GEMM weight as MXFP8: e4m3 bytes, then one E8M0
 * scale per 32 input features of each output row (qst_quant_mx). qst_encoder_forward on a QST_PREC_FP8 handle takes it
 * as its `shadow` argument. */
int qst_refresh_shadow_mx(const qst_encoder* enc, const float* params, void* shadow_mx, void* stream);

/*
 * Replaces SentenceTransformer.forward = Sequential(Transformer, Pooling(mean)[, Normalize])
 * (sentence-transformers 2.2.2; call site /root/reference/models/quadruplet_sentence_transformer.py:42-60)
 * over BertModel / MPNetModel.forward (transformers).
 *   ids, mask, type_ids : int64 [nseq, L] (type_ids may be NULL = all zero); L multiple of 32, <= 512
 *   params              : fp32 arena; shadow: bf16 arena from qst_refresh_shadow
 *   out_emb             : fp32 [nseq, H]  ('sentence_embedding')
 *   out_tok             : fp32 [nseq, L, H] token embeddings, or NULL
 *   saved               : activation arena (qst_encoder_saved_bytes)
 */
int qst_encoder_forward(qst_encoder* enc, const int64_t* ids, const int64_t* mask, const int64_t* type_ids,
                        int nseq, int L, const float* params, const void* shadow_bf16,
                        float* out_emb, float* out_tok, void* saved, size_t saved_bytes, int training,
                        void* stream);

/* Dropout for training forwards/backwards of this handle (HF hidden_dropout_prob on the embeddings and on both
 * projection outputs of every layer, attention_probs_dropout_prob on the softmax probabilities; the reference trains
 * in train() mode with HF's defaults 0.1 / 0.1: /root/reference/training/main.py:128). p_* in [0, 1); 0 / 0 turns it off
 * (the state of a new handle). state_dev: device uint32[4] owned by the caller, initialised with qst_dropout_init; every
 * qst_encoder_forward(training != 0) of a handle with dropout on first advances its step counter (on the stream, so the
 * step can sit inside a captured graph), and the backward of that forward recomputes the same masks from it -- no mask
 * is ever stored (include/qst_kernels.h: QstDrop). Inference forwards never drop. Every precision: the fp8 training forward
 * and the parity path (QST_PREC_BF16X3) drop at the same places with the same masks. */
int qst_encoder_set_dropout(qst_encoder* enc, float p_hidden, float p_attn, uint32_t* state_dev);
int qst_dropout_init(uint32_t* state_dev, uint64_t seed, void* stream);
/* Where this handle runs the feed-forward block (dense + GELU + dense + residual + LayerNorm) as ONE kernel instead of two
 * GEMM launches (H = 384 only, no dropout): bit 0 = inference forward (default), bit 1 = training forward, bit 2 = backward.
 * Same results to fp32 summation order; the training variants are slower than the two-kernel path. */
int qst_encoder_set_ffn_chain(qst_encoder* enc, int mask);
/* Where this handle runs a projection and the LayerNorm behind it (forward), or a dgrad and the LayerNorm backward behind
 * it, as ONE kernel (H = 384: full-row tiles; H = 512 / 768 / 1024: the workgroups of a row panel exchange row statistics
 * inside the launch): 0 = by size (default: from 16,384 token rows at H = 384; above, from two tiles per CU -- 256 x 256, or
 * 128 x 384 where only that gives two: H = 768 from 32,768 token rows),
 * 1 = wherever such a kernel exists, 2 = never. Same results to fp32 summation order. */
int qst_encoder_set_ln_fusion(qst_encoder* enc, int mode);
int qst_dropout_advance(uint32_t* state_dev, void* stream);

/*
 * Backward of the call above (replaces autograd through the same modules;
 * reference call site: `loss.backward()` inside SentenceTransformer.fit).
 *   grad_emb   : fp32 [nseq, H]
 *   grads      : fp32 arena; gradients are ACCUMULATED into it (zero it for a fresh step)
 *   workspace  : qst_encoder_bwd_workspace_bytes
 *   saved      : the arena a TRAINING forward of this process filled (any handle of the same model and arena kind: the
 *                dropout rates that forward ran under are remembered per arena, so the masks rebuilt here are its masks
 *                whatever qst_encoder_set_dropout has been told since); an arena without such a forward, one filled for another
 *                (nseq, L), or an arena of another kind -- the fp32 one of QST_PREC_BF16X3, the f16 one of QST_PREC_F16 / F16W,
 *                the bf16 one of QST_PREC_BF16 / FP8 -- is refused with QST_ERR_NO_FORWARD.
 */
int qst_encoder_backward(qst_encoder* enc, const int64_t* ids, const int64_t* mask, const int64_t* type_ids,
                         int nseq, int L, const float* params, const void* shadow_bf16,
                         const float* grad_emb, float* grads, void* saved, size_t saved_bytes,
                         void* workspace, size_t workspace_bytes, void* stream);

/* Same backward pass in stages (head -> layers [layer_lo, layer_hi) top-down -> embeddings) so the caller can start
 * the RCCL all-reduce of a finished layer's gradients while lower layers are still running.
 * Calls must cover head first, then contiguous descending layer ranges, then embeddings, on one stream.
 * Per-layer gradient ranges of the arena: qst_arena_segment(). */
int qst_encoder_backward_partial(qst_encoder* enc, const int64_t* ids, const int64_t* mask, const int64_t* type_ids,
                                 int nseq, int L, const float* params, const void* shadow_bf16,
                                 const float* grad_emb, float* grads, void* saved, size_t saved_bytes,
                                 void* workspace, size_t workspace_bytes,
                                 int do_head, int layer_hi, int layer_lo, int do_embed, void* stream);

/* The staged backward with flags. QST_BWD_HEAD / QST_BWD_EMBED = do_head / do_embed above. For LAYER 0 ALONE
 * (layer_lo == 0, layer_hi == 1): QST_BWD_SKIP_WGRAD leaves out that layer's weight-gradient launch (its operands stay in
 * the workspace) and a later call with QST_BWD_WGRAD_ONLY for the same layer runs just that launch -- a data-parallel
 * step does {layer 0 | SKIP_WGRAD | EMBED}, starts the all-reduce of the embedding gradients (the largest bucket),
 * and runs layer 0's weight gradients underneath it. No other stage may run between the two calls. Any other layer range
 * with either flag is QST_ERR_BAD_ARG: below a layer l > 0 the next stage overwrites the gradients its postponed launch
 * would read. */
enum { QST_BWD_HEAD = 1, QST_BWD_EMBED = 2, QST_BWD_SKIP_WGRAD = 4, QST_BWD_WGRAD_ONLY = 8 };
int qst_encoder_backward_stage(qst_encoder* enc, const int64_t* ids, const int64_t* mask, const int64_t* type_ids,
                               int nseq, int L, const float* params, const void* shadow_bf16,
                               const float* grad_emb, float* grads, void* saved, size_t saved_bytes,
                               void* workspace, size_t workspace_bytes,
                               int flags, int layer_hi, int layer_lo, void* stream);

/*
 * Replaces gamma_quadruplet_loss (/root/reference/models/losses/losses.py:9-69) and its autograd:
 * three triplet_margin_loss terms with pairwise_distance eps=1e-6 inside the norm.
 *   xa,xp,xq,xn : fp32 [B, D] anchor / positive / partially-positive / negative (row stride = D)
 *   out_loss    : fp32 [B] (reduction none) or [1]
 *   grad_*      : fp32 [B, D] d(loss)/dx, or all NULL for forward only. For reduction none the
 *                 upstream gradient is grad_out [B] (NULL = ones); for sum/mean grad_out is [1] or NULL.
 *   scratch     : fp32 [B] used for deterministic two-stage reduction when reduction != none
 */
int qst_quadruplet_loss(const float* xa, const float* xp, const float* xq, const float* xn,
                        int B, int D, float gamma, float margin_pos_neg, float margin_pos_part,
                        float margin_part_neg, float p, int swap, int reduction,
                        float* out_loss, const float* grad_out,
                        float* grad_a, float* grad_p, float* grad_q, float* grad_n,
                        float* scratch, void* stream);

/*
 * Replaces torch.nn.utils.clip_grad_norm_(params, max_grad_norm) + torch.optim.AdamW.step()
 * with ST fit()'s two parameter groups (/root/reference/training/main.py:128-148).
 *   n            : arena elements; decay is applied per segment as the layout says
 *   grad_scale   : multiplies every gradient first (1/world_size after an all-reduce sum)
 *   max_grad_norm: <= 0 disables clipping. norm_out (fp32 [1], device) receives the pre-clip global L2 norm.
 *   step         : 1-based optimiser step for bias correction
 *   scratch      : fp32 [1024] partial sums
 */
int qst_clip_adamw_step(const qst_encoder* enc, float* params, float* grads, float* exp_avg, float* exp_avg_sq,
                        float lr, float beta1, float beta2, float eps, float weight_decay,
                        float max_grad_norm, float grad_scale, int64_t step,
                        float* norm_out, float* scratch, void* stream);

/* The same step with the schedule on the device, so that a whole training step can be captured in a HIP graph and
 * replayed: no per-step host value is a kernel argument. step_dev (int64 [1], device) holds the number of optimiser
 * steps taken so far and is incremented by the call; the learning rate of step t is
 * transformers.get_linear_schedule_with_warmup(base_lr, warmup_steps, total_steps) at t-1 (constant base_lr when
 * total_steps <= 0), as SentenceTransformer.fit's 'WarmupLinear' (reference call site training/main.py:133-134).
 * scratch: fp32 [1024 + 3]. */
int qst_clip_adamw_step_sched(const qst_encoder* enc, float* params, float* grads, float* exp_avg, float* exp_avg_sq,
                              float base_lr, float beta1, float beta2, float eps, float weight_decay,
                              float max_grad_norm, float grad_scale, int64_t warmup_steps, int64_t total_steps,
                              int64_t* step_dev, float* norm_out, float* scratch, void* stream);

/* Mixed-precision optimiser step: what SentenceTransformer.fit(use_amp=True) wraps around AdamW -- scaler.scale(loss)
 * .backward(); scaler.unscale_(optimizer); clip_grad_norm_; scaler.step(optimizer); scaler.update(); and the scheduler
 * step skipped whenever the scale changed (sentence-transformers 2.2.2 fit(); reference call site
 * /root/reference/training/main.py:128-148 with use_amp = True) -- with torch.cuda.amp.GradScaler's defaults as ST builds it
 * (init_scale 65536, growth_factor 2, backoff_factor 0.5, growth_interval 2000), all on the device:
 *   scaler_dev : fp32 [4] {scale, growth tracker, 1.0 if the LAST step was skipped, number of skipped steps so far};
 *                qst_amp_scaler_init writes {init_scale, 0, 0, 0}. scaler_dev[0] is what the backward multiplies the loss
 *                gradient by: pass scaler_dev as qst_quadruplet_loss's grad_out (reduction sum / mean).
 *   step_dev   : int64 [2] {optimiser steps taken, scheduler steps taken}. Bias correction uses the first, the learning rate
 *                (WarmupLinear as qst_clip_adamw_step_sched) the second.
 * The call unscales by 1 / scale (times grad_scale, the 1 / world_size of a data-parallel step), takes the global norm of
 * the unscaled gradients (norm_out), and: if the norm is not finite (an inf / nan anywhere in `grads`: an f16 gradient
 * overflowed) the parameters, the moments and the optimiser step count stay as they are, the gradients are zeroed, scale *=
 * backoff and the tracker restarts; otherwise clip + AdamW + zero_grad as qst_clip_adamw_step, tracker += 1, and at
 * growth_interval scale *= growth. The scheduler count advances only when the scale did not change (ST's rule).
 * growth_interval <= 0: a STATIC scale (no growth; an overflow still skips the step, without backoff).
 * scratch: fp32 [1024 + 8]. Every rank of a data-parallel job sees the same reduced gradients, so the same decisions. */
int qst_amp_scaler_init(float* scaler_dev, float init_scale, void* stream);
int qst_clip_adamw_step_amp(const qst_encoder* enc, float* params, float* grads, float* exp_avg, float* exp_avg_sq,
                            float base_lr, float beta1, float beta2, float eps, float weight_decay,
                            float max_grad_norm, float grad_scale, int64_t warmup_steps, int64_t total_steps,
                            int64_t* step_dev, float* scaler_dev, float growth_factor, float backoff_factor,
                            int32_t growth_interval, float* norm_out, float* scratch, void* stream);

/* Retrieval scoring for the encode()-driven evaluators: what sentence-transformers'
 * InformationRetrievalEvaluator does per corpus chunk -- util.cos_sim / util.dot_score of the query embeddings
 * against the chunk, then torch.topk(k) (reference call sites models/evaluators.py:572-588,
 * ir_evauation_script.py:107-131). queries f32 [nq, dim], corpus f32 [nc, dim] (device, contiguous), dim % 32 == 0,
 * k <= min(nc, 1024). mode: QST_SCORE_DOT (util.dot_score), QST_SCORE_COS (util.cos_sim: both sides normalised first,
 * eps 1e-12) or QST_SCORE_EUCLID (the reference's own euclidean_score, models/evaluators.py:392-405:
 * 1 / (1 + ||q - c||_2), which training/main.py:57 and ir_evauation_script.py:71 pass as 'euclid_score'). Outputs: the
 * k best per query, sorted by descending score (ties: ascending corpus index): out_scores f32 [nq, k], out_index
 * int64 [nq, k]. */
enum { QST_SCORE_DOT = 0, QST_SCORE_COS = 1, QST_SCORE_EUCLID = 2 };
size_t qst_topk_workspace_bytes(int nq, int nc, int dim);
int qst_topk_scores(const float* queries, const float* corpus, int nq, int nc, int dim, int k, int mode,
                    float* out_scores, int64_t* out_index, void* workspace, size_t workspace_bytes, void* stream);

/* The full score matrix of one of those functions (what util.cos_sim / util.dot_score / euclidean_score return):
 * out f32 [nq, ld_out] with ld_out = nc rounded up to a multiple of 4. */
size_t qst_score_workspace_bytes(int nq, int nc, int dim);
int qst_score_matrix(const float* queries, const float* corpus, int nq, int nc, int dim, int mode, float* out,
                     int64_t ld_out, void* workspace, size_t workspace_bytes, void* stream);
/* torch.nn.functional.normalize(x, p=2, dim=1, eps=1e-12) over the rows of x f32 [n, dim] (contiguous): what
 * SentenceTransformer.encode(normalize_embeddings=True) applies. out may alias x. */
int qst_normalize_rows(const float* x, int n, int dim, float* out, void* stream);


/* The same with a ceiling: corpus rows scoring above max_score do not take part. This is the reference's negative
 * selection (dataset/quadruplet_dataset.py:185-270: candidates with SBERT cosine <= 0.2 to the reference caption, then
 * hard_contrastive_sampling = the k highest remaining scores, :31-47), for all reference captions at once
 * (SURVEY.md 8f rank 4). Where fewer than k rows qualify the tail of a result row is score -inf, index -1. */
int qst_topk_scores_capped(const float* queries, const float* corpus, int nq, int nc, int dim, int k, int mode,
                           float max_score, float* out_scores, int64_t* out_index, void* workspace,
                           size_t workspace_bytes, void* stream);

/* ---- data parallelism (the reference itself is single-process, training/main.py:113) ----
 * The exchange step of the data-parallel path is a sum of slices of the contiguous fp32 gradient arena between
 * qst_encoder_backward_stage calls; qst_clip_adamw_step's grad_scale applies the 1/world_size. The Python side of this
 * build drives it through torch.distributed (backend "nccl" = RCCL over xGMI). A caller without torch uses these: one
 * process per GPU; rank 0 obtains the id and ships its QST_COMM_ID_BYTES to the other ranks by its own means; every rank
 * calls qst_comm_init with the HIP device it will use current. RCCL is loaded at run time (the copy already in the
 * process if there is one), libqst.so has no link dependency on it.
 *   qst_allreduce_bucket : in-place SUM over all ranks of `count` elements at `ptr` (device), enqueued on `stream` --
 *                          pass a stream other than the compute stream and order the two with events to overlap the
 *                          exchange with the next backward stage. dtype: QST_COMM_F32 (the gradient arena) or QST_COMM_BF16.
 * Status QST_ERR_COMM: qst_comm_last_error() returns RCCL's text. */
typedef struct qst_comm qst_comm;
#define QST_COMM_ID_BYTES 128
enum { QST_COMM_F32 = 0, QST_COMM_BF16 = 1 };
int qst_comm_unique_id(void* id_out /* QST_COMM_ID_BYTES, host */);
int qst_comm_init(int rank, int world, const void* unique_id, qst_comm** out);
int qst_allreduce_bucket(qst_comm* comm, void* ptr, int64_t count, int dtype, void* stream);
int qst_comm_rank(const qst_comm* comm);
int qst_comm_world(const qst_comm* comm);
void qst_comm_destroy(qst_comm* comm);
const char* qst_comm_last_error(void);

#ifdef __cplusplus
}
#endif
#endif /* QST_H */
