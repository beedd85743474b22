/* b4r.h  --  C ABI of libb4r_hip.so: the MI355X (gfx950) BERT4Rec hot path.
 *
 * The reference (maneymarkus/BERT4Rec, Python on TensorFlow 2.10) has no FFI boundary of its own: its hot path sits
 * behind Keras object interfaces.  Each entry point below replaces the TensorFlow kernels reached
 * from one of those call sites; the reference file:line it replaces is cited on every declaration.  A maintainer of
 * the reference would bind this library with ctypes exactly as bert4rec_amd/_lib.py does.
 *
 * Conventions
 *   - every function returns int: 0 = ok, negative = B4R_E_*; nothing throws; b4r_last_error() returns the
 *     message of the last failure on the calling thread.
 *   - tensor arguments are CALLER-OWNED DEVICE pointers (contiguous row-major unless a leading dimension is given),
 *     float32 values and int64 ids exactly as the reference's batch dict holds them (bert4rec_model.py:15-22).
 *   - the library allocates nothing persistent; scratch comes from a caller workspace whose size is queried first.
 *   - every op only ENQUEUES on the given hipStream_t: no host synchronisation, no internal threads; safe to
 *     capture in a hipGraph.  All step-varying scalars (step counter, dropout seed, loss sums, gradient norm)
 *     live in the device-resident b4r_train_state, never in by-value arguments.
 */
#ifndef B4R_H_
#define B4R_H_

#include <stddef.h>
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

typedef struct ihipStream_t* b4r_stream_t; /* == hipStream_t */

#define B4R_VERSION 100
#define B4R_MAX_LAYERS 32

enum { B4R_OK = 0, B4R_E_BADARG = -1, B4R_E_SHAPE = -2, B4R_E_ALIGN = -3, B4R_E_HIP = -4, B4R_E_NOMEM = -5 };

/* Encoder hyper-parameters: Bert4RecEncoder.__init__ kwargs, bert4rec_encoder.py:62-80, as set by
 * bert4rec/config/bert4rec_train_configs/ *.json.  head_dim = hidden_size / num_heads must be 32 (true for every
 * shipped config). */
typedef struct b4r_model_config {
  int32_t vocab_size;
  int32_t hidden_size;
  int32_t num_layers;
  int32_t num_heads;
  int32_t inner_dim;
  int32_t max_seq_len; /* max_sequence_length: rows of the position table */
  float output_dropout;
  float attention_dropout;
  float ln_eps; /* 1e-12: bert4rec_encoder.py:117, tfm TransformerEncoderBlock norm_epsilon, tfm MaskedLM */
} b4r_model_config;

/* One batch, the dict of bert4rec_model.py:15-22 / bert4rec_preprocessor.py:48-116.  All int64 [B, .] row-major. */
typedef struct b4r_batch {
  const int64_t* input_word_ids;      /* [B,L] */
  const int64_t* input_mask;          /* [B,L] 1 = real token (incl. masked), 0 = pad */
  const int64_t* masked_lm_positions; /* [B,P] or NULL (then no MLM head) */
  const int64_t* masked_lm_ids;       /* [B,P] y_true; 0 = ignored slot (trainer_utils.py:13) ; NULL for inference */
  int32_t B, L, P;
} b4r_batch;

/* AdamWeightDecay + WarmUp/PolynomialDecay: optimizers/__init__.py:7-56, adam_w_optimizer.py:53-76 */
typedef struct b4r_adamw_config {
  float init_lr;            /* 1e-4 */
  float end_lr;             /* 0 */
  int32_t num_train_steps;  /* 400000 */
  int32_t num_warmup_steps; /* 100 */
  float weight_decay_rate;  /* 0.01 */
  float beta_1, beta_2;     /* 0.9, 0.999 */
  float epsilon;            /* 1e-6 */
  float clip_norm;          /* 5.0 (gradient_clip_norm, adam_w_optimizer.py:67); <= 0 disables */
  /* optional DEVICE pointer, one byte per float of the flat parameter buffer, nonzero = this element is weight-decayed: a custom
   * include_in_weight_decay / exclude_from_weight_decay selection (adam_w_optimizer.py:154-168).  NULL = the layout's rule: the
   * first b4r_param_decay_floats() floats (= the reference's default exclusion list ["LayerNorm", "layer_norm", "bias"]). */
  const uint8_t* decay_mask;
} b4r_adamw_config;

/* Device-resident state, 64 bytes, owned by the caller (one torch tensor).  Kernels read and write it; the host reads
 * it back only when it wants the metrics.  Layout is part of the ABI. */
typedef struct b4r_train_state {
  uint32_t seed;        /* [0]  dropout seed                                                         */
  uint32_t step_lo;     /* [1]  low 32 bits of the optimizer iteration (0-based), also the dropout step */
  int64_t step;         /* [2,3] optimizer.iterations                                                */
  float loss_sum;       /* [4]  sum over valid slots of the per-slot CE  (trainer_utils.py:19-22 numerator)   */
  float valid_count;    /* [5]  number of slots with masked_lm_ids != 0  (denominator)                */
  float correct_masked; /* [6]  argmax == y_true over valid slots        (trainer_utils.py:49-60)     */
  float correct_all;    /* [7]  argmax == y_true over all B*P slots      (SparseCategoricalAccuracy)  */
  float slots_all;      /* [8]  B*P of the batch                                                      */
  float grad_sqnorm;    /* [9]  sum g^2 of the (summed, un-normalised) gradient buffer                */
  float grad_norm;      /* [10] global norm of the mean gradient as clip_by_global_norm sees it      */
  float lr;             /* [11] lr_t used by the last optimizer step                                  */
  float reserved[4];    /* [12..15] zero-initialise; [12] is the optimizer kernel's completion ticket          */
} b4r_train_state;

/* ------------------------------------------------------------------------------------------------------------ */
int b4r_version(void);
/* copies the calling thread's last error message (NUL-terminated) and returns its length */
size_t b4r_last_error(char* buf, size_t cap);

/* ---- parameter layout ---------------------------------------------------------------------------------------
 * All trainable variables live in ONE flat fp32 buffer (so that clip / AdamW / the DP all-reduce are one pass).
 * The first b4r_param_decay_floats() floats are the weight-decayed variables (kernels + the two embedding tables),
 * the rest are the biases and LayerNorm gamma/beta (adam_w_optimizer.py:154-168 with optimizers/__init__.py:35-36).
 * Entries are named after the reference's Keras variables ("transformer/layer_0/self_attention/query/kernel", ...).
 * query/key/value kernels are column blocks of one [H,3H] matrix (ld = 3H) so that QKV is a single GEMM.
 * The pooler (bert4rec_encoder.py:149-153) is not on the loss path (gradient None): it lives in a separate buffer. */
int64_t b4r_param_total_floats(const b4r_model_config* cfg);  /* size of the flat buffer (padded to 4) */
int64_t b4r_param_decay_floats(const b4r_model_config* cfg);
int32_t b4r_param_count(const b4r_model_config* cfg);
int b4r_param_info(const b4r_model_config* cfg, int32_t index, char* name, size_t name_cap, int64_t* offset,
                   int32_t* rows, int32_t* cols, int32_t* ld, int32_t* decay);
int64_t b4r_pooler_floats(const b4r_model_config* cfg); /* [H,H] kernel then [H] bias */

/* ---- workspace ----------------------------------------------------------------------------------------------
 * One caller buffer holds every activation saved for backward plus scratch.  Named regions (outputs of
 * BERT4RecModel.call, bert4rec_model.py:110-149) are located with b4r_workspace_region:
 *   "sequence_output" [B*L,H], "encoder_output_<i>" [B*L,H], "mlm_logits" [B*P, ld>=V], "mlm_hidden" [B*P,H],
 *   "pooled_output" [B,H], "embeddings" [B*L,H]. */
int64_t b4r_workspace_bytes(const b4r_model_config* cfg, int32_t B, int32_t L, int32_t P);
/* bytes that an ENCODER-ONLY forward without the pooler needs (b4r_forward with B4R_FLAG_ENCODER_ONLY and not B4R_FLAG_POOLER:
 * BERT4RecModel.rank_items' forward, bert4rec_model.py:215): the encoder's own regions of the (B, L, P) layout -- the same offsets as in
 * the full workspace, so b4r_workspace_region("sequence_output" / "encoder_output_i" / "embeddings") holds -- without the masked-LM
 * head's [B*P, V] logits and the backward area (an evaluation batch of ML-20M shape: 1.2 GB less). */
int64_t b4r_workspace_bytes_encoder(const b4r_model_config* cfg, int32_t B, int32_t L, int32_t P);
int b4r_workspace_region(const b4r_model_config* cfg, int32_t B, int32_t L, int32_t P, const char* name,
                         int64_t* offset_floats, int32_t* rows, int32_t* cols, int32_t* ld);

/* ---- model level --------------------------------------------------------------------------------------------
 * b4r_forward          replaces BERT4RecModel.call            bert4rec_model.py:110-149  (encoder + MaskedLM)
 *                      = Bert4RecEncoder.call                  bert4rec_encoder.py:186-231
 * b4r_loss             replaces MaskedSparseCategoricalCrossentropy.call trainer_utils.py:12-23 and the two metrics
 *                      trainer_utils.py:49-60 / bert4rec_trainer.py:28-33; writes sums into b4r_train_state and
 *                      (with want_grad) overwrites the logits with d(loss_sum)/d(logits)
 * b4r_backward         replaces tape.gradient                  bert4rec_model.py:166-167 (gradient of loss_SUM; the
 *                      1/valid_count factor is applied inside b4r_optimizer_step, so a data-parallel caller can
 *                      all-reduce grads and state sums in between: SURVEY.md §8e)
 * b4r_optimizer_step   replaces AdamWeightDecay.apply_gradients adam_w_optimizer.py:100-137 (+ schedule :22-36)
 * flags: bit0 = training (dropout on), bit1 = also compute pooled_output. */
#define B4R_FLAG_TRAINING 1
#define B4R_FLAG_POOLER 2
/* Train-step variant of the masked-LM head that never materialises the [B*P, V] logits (hidden size 64/128/256, B4R_GEMM_BF16X3;
 * ask b4r_fused_head_supported).  b4r_forward with this flag needs masked_lm_ids and leaves the per-slot loss terms and
 * d loss_sum / d transform in the workspace instead of "mlm_logits"; b4r_loss must then be called with
 * want_grad | B4R_LOSS_FUSED_HEAD and b4r_backward with the same flag.  b4r_train_step uses it whenever it is supported. */
#define B4R_FLAG_FUSED_HEAD 4
/* b4r_backward only: `grads` is followed by 8 more floats (a multiple of 16 bytes) that receive the step's sums
 * [loss_sum, valid_count, correct_masked, correct_all, slots_all, 0, 0, 0] from the state, so that a data-parallel caller
 * all-reduces ONE flat buffer [gradients | sums] and then calls b4r_optimizer_step_reduced, which takes the
 * reduced sums from there.  No copy kernels around the collective. */
#define B4R_FLAG_GRAD_TAIL 8
/* Range of the item-table gradient: the embedding rows' contributions d loss_sum / d (token row) are scatter-added in 64-bit fixed
 * point (units of 2^-36, order-free => bitwise reproducible).  A contribution that is not finite or reaches 2^18 = 262 144 in magnitude
 * cannot be represented: the whole "word_embeddings/embeddings" gradient of that step is then NaN (and with it the step's gradient
 * norm), as an Inf / NaN would have made it with float atomics -- never a silently wrapped finite value. */
/* b4r_forward + b4r_backward of one TRAIN step (both or neither): the last encoder layer's feed-forward half is evaluated only on the
 * rows of the sequence output that the masked-LM head gathers (about P/L of them: 20 % at ML-1M) -- forward and backward.  Loss, metrics
 * and every gradient are unchanged (the other rows of that output reach neither the loss nor, through attention, any row that does);
 * "sequence_output" / "encoder_output_<last>" are then only defined on those rows, which is why the forward / evaluation API never sets
 * the flag.  b4r_train_step uses it.  Hidden size 64: the resident feed-forward block in its slot mode.  Every other hidden size (the
 * tile-product path), when P <= L / 2 and inner_dim >= 3 hidden + 8: the same half as dense products on compact [B*P, .] rows, and
 * for 64 < L <= 224, P <= 64 also the layer's attention half -- the core with the slots as its ONLY queries (keys / values: all
 * tokens), output projection, dropout, residual and LayerNorm on the compact rows; "encoder_output_<last>" is written at the
 * slots' rows only, the layer's ctx / z1 / x1 regions hold compact data.  Ignored otherwise.
 * PRECONDITION: the valid masked-LM slots (masked_lm_ids != 0) of one sequence name distinct positions -- what the reference's
 * preprocessor and b4r_mask_batch produce (dataloader_utils.py:221-226 samples positions without replacement).  Two valid slots on
 * one position would each write that row's gradient (last writer wins) where the scatter-add path sums them. */
#define B4R_FLAG_HEAD_ROWS_ONLY 16
/* b4r_backward only, with B4R_FLAG_FUSED_HEAD: the call begins by SETTING the state's sums (loss_sum, valid_count, correct_masked,
 * correct_all, slots_all; gradient norms to 0) from the loss rows the logits-free head left in the workspace -- what
 * b4r_loss(..., want_grad | B4R_LOSS_FUSED_HEAD | B4R_LOSS_OVERWRITE) does, inside the launch that clears the gradients (the same
 * summation order, bit for bit).  No b4r_state_begin_step / b4r_loss call is then needed between forward and backward. */
#define B4R_FLAG_LOSS_SUMS 32
/* b4r_forward only: stop at the sequence output although the batch carries masked_lm_positions / masked_lm_ids -- they then only
 * name the rows of B4R_FLAG_HEAD_ROWS_ONLY.  What an evaluation wants (BERT4RecModel.rank_items, bert4rec_model.py:203-240, ranks a
 * handful of slots per user): the last layer's feed-forward half on the ranked rows only, no [B*P, V] logits. */
#define B4R_FLAG_ENCODER_ONLY 64
#define B4R_LOSS_FUSED_HEAD 2
#define B4R_LOSS_OVERWRITE 4 /* b4r_loss: set the state's sums instead of adding to them (= b4r_state_begin_step first) */
int32_t b4r_fused_head_supported(const b4r_model_config* cfg);
int b4r_forward(const b4r_model_config* cfg, const b4r_batch* batch, const float* params, const float* pooler,
                void* workspace, int64_t workspace_bytes, b4r_train_state* state, int32_t flags, b4r_stream_t stream);
int b4r_loss(const b4r_model_config* cfg, const b4r_batch* batch, void* workspace, int64_t workspace_bytes,
             b4r_train_state* state, int32_t want_grad, b4r_stream_t stream);
int b4r_backward(const b4r_model_config* cfg, const b4r_batch* batch, const float* params, float* grads,
                 void* workspace, int64_t workspace_bytes, b4r_train_state* state, int32_t flags,
                 b4r_stream_t stream);
int b4r_optimizer_step(const b4r_model_config* cfg, const b4r_adamw_config* hp, float* params, const float* grads,
                       float* adam_m, float* adam_v, void* workspace, int64_t workspace_bytes,
                       b4r_train_state* state, b4r_stream_t stream);
/* b4r_optimizer_step for a gradient buffer that went through the data-parallel all-reduce with its 8-float tail
 * (B4R_FLAG_GRAD_TAIL): the reduced loss / count sums are moved from the tail into the state first */
int b4r_optimizer_step_reduced(const b4r_model_config* cfg, const b4r_adamw_config* hp, float* params, const float* grads,
                               float* adam_m, float* adam_v, void* workspace, int64_t workspace_bytes,
                               b4r_train_state* state, b4r_stream_t stream);
/* zero the per-step sums of the state (loss_sum .. grad_norm); call before b4r_loss */
int b4r_state_begin_step(b4r_train_state* state, b4r_stream_t stream);
/* BERT4RecModel.train_step, bert4rec_model.py:151-173 = begin_step + forward + loss + backward + optimizer_step */
int b4r_train_step(const b4r_model_config* cfg, const b4r_adamw_config* hp, const b4r_batch* batch, float* params,
                   float* grads, float* adam_m, float* adam_v, void* workspace, int64_t workspace_bytes,
                   b4r_train_state* state, b4r_stream_t stream);

/* ---- batch construction ----------------------------------------------------------------------
 * replaces BERT4RecPreprocessor.process_element bert4rec_preprocessor.py:48-116 on already truncated, right-padded token
 * rows for a whole batch: apply_dynamic_masking_task dataloader_utils.py:186-261 (or mask_last_token_only :264-269 when
 * finetune != 0) + the padding of the six [B,.] int64 tensors of bert4rec_model.py:15-22.  Same law as the reference
 * (uniform subset of min(P, max(1, int(n*rate))) of the first n positions, ascending; [MASK] / random id / unchanged by
 * mask_token_rate / random_token_rate), own counter-hash random stream: a dataset can be re-masked every epoch on the
 * device instead of being masked `duplication_factor` times on the host.
 * row_index (optional, [B]): output row r is row row_index[r] of tokens [U, L] -- a dataset's token matrix stays in HBM and a batch
 * is an index list (make_batches' shuffle + batch, dataloader_utils.py:341-346); the random stream is keyed by the dataset row.
 * row_finetune (optional, per dataset row): rows with a non-zero flag get the last-token mask (the 10 % finetuning share that
 * get_data mixes into the training set, bert4rec_dataloader.py:100-108); `finetune` != 0 forces it for every row. */
int b4r_mask_batch(const int64_t* tokens, const int64_t* row_index, const int64_t* row_finetune, int32_t B, int32_t L, int32_t P,
                   int32_t V, double selection_rate, float mask_token_rate, float random_token_rate, int32_t finetune,
                   uint64_t seed, int64_t* input_word_ids, int64_t* input_mask, int64_t* labels, int64_t* masked_lm_positions,
                   int64_t* masked_lm_ids, int64_t* masked_lm_weights, b4r_stream_t stream);

/* host-only probe (no GPU): the uniform in the OPEN interval (0, 1) that b4r_mask_batch and b4r_sample_candidates form from a
 * 32-bit hash word -- (top 23 bits + 0.5) / 2^23, exact in fp32.  python's random.random() of dataloader_utils.py:245-253 is in
 * [0, 1): with mask_token_rate = 1.0 `u < rate` must hold for EVERY word (a 24-bit form rounds its largest value to 1.0f and lets
 * one selected position in 2^24 keep its label).  tests/test_host.py sweeps the extreme words through this entry. */
float b4r_uniform_from_hash(uint32_t hash_word);

/* ---- evaluator negatives -------------------------------------------------------------------
 * replaces the per-slot sampler call of bert4rec_evaluator.py:84-104 (PopularRandomSampler.sample:
 * np.random.choice(vocab, 100 + |without|, replace=False, p=popularity), drop `without`, keep 100) for ALL ranked slots of
 * a batch in one launch.  Same distribution (successive draws proportional to p among the remaining allowed items,
 * realised as Gumbel top-k on log p), own counter-hash random stream (numpy's stream cannot be reproduced).
 * logp [V] = log popularity (-inf where p = 0); exclude [R,E] item ids that must not be drawn for row r (out-of-range
 * entries such as -1 are ignored; gt[r] is excluded as well); cand [R, C+1]: the C draws in draw order, then gt[r] as the
 * last candidate (bert4rec_evaluator.py:103).  A row with fewer than C drawable items gets -1 entries (the reference
 * raises ValueError there; the Python layer does too). */
int b4r_sample_candidates(const float* logp, int32_t V, const int64_t* exclude, int32_t E, const int64_t* gt, int32_t R,
                          int32_t C, uint64_t seed, int64_t* cand, b4r_stream_t stream);
/* the same; short_flag (optional, one device byte, never cleared here) is set to 1 when a row had fewer than C drawable items: an
 * evaluation reads it back once at its end instead of scanning cand for -1 after every batch */
int b4r_sample_candidates_flagged(const float* logp, int32_t V, const int64_t* exclude, int32_t E, const int64_t* gt, int32_t R,
                                  int32_t C, uint64_t seed, int64_t* cand, uint8_t* short_flag, b4r_stream_t stream);

/* ---- ranking ------------------------------------------------------------------------------------------------
 * replaces BERT4RecModel.rank_items bert4rec_model.py:224-239 (gather candidate logits, tf.argsort DESCENDING,
 * gather candidates) and the rank lookup of bert4rec_evaluator.py:113-117.
 * score(r,j) = fma-chain_k(hidden[hidden_row[r]][k] * table[cand[r][j]][k]) + bias[cand[r][j]]  (k ascending, fp32)
 * ranking[r][pos] = cand[r][j] with pos = #{i: s_i > s_j} + #{i<j: s_i == s_j}   (stable descending)
 * gt_rank[r] = 1 + min{pos_j : cand[r][j] == gt[r]}  (0 if gt[r] is not a candidate).  Any output may be NULL.
 * cand == NULL: the candidates of every row are 0 .. C-1 (rank_items(items=None), bert4rec_model.py:236, C = vocab size).
 * Only the scores asked for are formed (the reference computes all B*P*V logits and reads 101 per user).  Up to 8192
 * candidates per row run in one launch; more (the whole vocabulary of Beauty / Reddit) take a radix argsort per row and
 * need `scratch` (b4r_rank_scratch_bytes(R, C) bytes for one pass over all rows; at least C*20, rows are then ranked in
 * groups; 16-byte aligned).  V = rows of `table` / entries of `bias`: a candidate id outside [0, V) scores -inf and reads nothing
 * (the reference raises ValueError for a row without enough drawable items, popular_random_sampler.py:104-109; the device
 * sampler marks such rows with -1 and the evaluator reports them after the batch). */
int64_t b4r_rank_scratch_bytes(int32_t R, int32_t C);
int b4r_rank_candidates(const float* hidden, int32_t hidden_ld, const int64_t* hidden_row, const float* table,
                        const float* bias, int32_t H, int32_t V, const int64_t* cand, int32_t R, int32_t C, const int64_t* gt,
                        int64_t* ranking, int32_t* gt_rank, float* scores, void* scratch, int64_t scratch_bytes,
                        b4r_stream_t stream);
/* replaces the metric loop of bert4rec_evaluator.py:118-120 over evaluation_metrics.py:47-112 for a batch of ranks:
 * gain_sums[m] += sum over gt_rank[i] > 0 of gain_m(gt_rank[i]), users[0] += #{gt_rank[i] > 0}; double / int64 DEVICE
 * accumulators the caller reads once per evaluate().  family[m]: 0 count (gain 1), 1 hit@cutoff (rank <= k), 2 NDCG@cutoff
 * (1 if rank == 1 else 1/log2(rank+1), inside the cut-off), 3 reciprocal rank (MAP with one relevant item).  One workgroup,
 * fixed summation order: bitwise reproducible. */
int b4r_rank_metrics(const int32_t* gt_rank, int32_t R, const int32_t* family, const int32_t* cutoff, int32_t n_metrics,
                     double* gain_sums, int64_t* users, b4r_stream_t stream);
/* tfm MaskedLM's transform (gather -> dense(gelu) -> LayerNorm, bert4rec_model.py:76-81,143) on an explicit list of R rows of
 * the sequence output [n_seq_rows, H]: out[r] = LN(gelu(seq[rows[r]].Wd + bd)).  The evaluation path transforms only the slots
 * it ranks (one per user) instead of all B*P.  scratch: 3*R*H + 2*R floats, 16-byte aligned. */
int b4r_mlm_transform_rows(const b4r_model_config* cfg, const float* params, const float* seq, int64_t n_seq_rows,
                           const int64_t* rows, int32_t R, float* out, float* scratch, b4r_stream_t stream);

/* ---- op level (each is also a stage of the model-level calls; exposed for parity tests and reuse) ------------ */

/* x = dropout(LN(E[ids] + P[pos]))   bert4rec_encoder.py:198-211 */
int b4r_embed_ln_fwd(const int64_t* ids, int32_t B, int32_t L, const float* table, int32_t V, const float* pos_table,
                     const float* gamma, const float* beta, int32_t H, float eps, float* out, float* mean,
                     float* rstd, const uint32_t* rng, float dropout, b4r_stream_t stream);

/* y = LN(z) rows of width H; saves mean / rstd */
int b4r_ln_fwd(const float* z, int32_t rows, int32_t H, const float* gamma, const float* beta, float eps, float* y,
               float* mean, float* rstd, b4r_stream_t stream);
/* dz from dy; dgamma/dbeta [H] (deterministic two-stage column sums; scratch >= b4r_ln_bwd_scratch_floats) */
int64_t b4r_ln_bwd_scratch_floats(int32_t rows, int32_t H);
int b4r_ln_bwd(const float* dy, const float* z, const float* mean, const float* rstd, const float* gamma,
               int32_t rows, int32_t H, float* dz, float* dgamma, float* dbeta, float* scratch,
               b4r_stream_t stream);

/* epilogues of b4r_gemm_f32 */
enum {
  B4R_EPI_NONE = 0,          /* C = acc                                                   */
  B4R_EPI_BIAS = 1,          /* C = acc + bias                                            */
  B4R_EPI_BIAS_QSCALE = 2,   /* C = (acc + bias) * (col < qcols ? qscale : 1)   (Keras MHA query scaling)     */
  B4R_EPI_BIAS_GELU = 3,     /* C2 = acc + bias ; C = gelu_erf(C2)                        */
  B4R_EPI_BIAS_DROP_RES = 4, /* C = R + dropout(acc + bias)                               */
  B4R_EPI_GELU_BWD = 5,      /* C = acc * gelu'(R)                                        */
  B4R_EPI_ADD_RES = 6,       /* C = acc + R                                               */
  B4R_EPI_BIAS_TANH = 7,     /* C = tanh(acc + bias)                     (pooler, bert4rec_encoder.py:149-153) */
  /* C = R + dropout(acc + bias) ; C2 = LayerNorm(C) * ln_gamma + ln_beta ; ln_mean / ln_rstd [M] = row statistics of C:
   * the dense + dropout + residual + LayerNorm tail of both halves of a Keras TransformerEncoderBlock in one launch.
   * Only where a workgroup holds whole rows: N == 64 (see b4r_gemm_ln_supported); otherwise B4R_E_SHAPE */
  B4R_EPI_BIAS_DROP_RES_LN = 8,
  /* dy = acc + R (not stored) ; C = LayerNorm backward of dy through the normalisation that produced ln_mean / ln_rstd
   * from its input ln_z with scale ln_gamma:  C = rstd (g - mean(g) - xhat mean(g xhat)), g = dy gamma, xhat = (z - mean) rstd;
   * ln_dgamma [64] = sum_rows dy xhat and ln_dbeta [64] = sum_rows dy via the partials C2
   * (b4r_gemm_ln_bwd_partial_floats(M) floats) and an ordered reduction.  The input-gradient product in front of a
   * LayerNorm backward + that backward in one launch.  N == 64, B as [N,K] only (b4r_gemm_ln_supported) */
  B4R_EPI_ADD_RES_LN_BWD = 9,
  /* C3 = acc + bias ; C = gelu_erf(C3) ; C2 = LayerNorm(C) * ln_gamma + ln_beta ; ln_mean / ln_rstd: the dense(gelu) ->
   * LayerNorm transform of tfm MaskedLM in one launch.  N == 64 only (b4r_gemm_ln_supported) */
  B4R_EPI_BIAS_GELU_LN = 10
};
typedef struct b4r_gemm_desc {
  const float* A; int32_t lda;   /* [M,K] row-major                                       */
  const float* B; int32_t ldb;   /* b_is_nk == 0: [K,N] ; b_is_nk == 1: [N,K]  (C = A.B^T) */
  float* C; int32_t ldc;         /* [M,N]                                                 */
  int32_t M, N, K;
  int32_t b_is_nk;
  int32_t epilogue;
  const float* bias;             /* [N]                                                   */
  float* C2; int32_t ldc2;
  const float* R; int32_t ldr;
  float qscale; int32_t qcols;
  /* dropout: on the epilogue (B4R_EPI_BIAS_DROP_RES, element index row*N+col) or, with a_dropout=1, on the A operand
   * as it is loaded (element index row*K+col): dY = dz * mask / keep without materialising dY */
  const uint32_t* rng; uint32_t drop_stream; float drop_rate; int32_t a_dropout;
  /* 1: columns [N, roundup(N,4)) of C / C2 (inside ldc) are scratch the kernel may overwrite, and of R may be read */
  int32_t c_pad_scratch;
  /* B4R_EPI_BIAS_DROP_RES_LN only (zero otherwise): scale / offset [N], optional statistics outputs [M], epsilon */
  const float* ln_gamma; const float* ln_beta; float* ln_mean; float* ln_rstd; float ln_eps;
  /* B4R_EPI_ADD_RES_LN_BWD only: ln_gamma, ln_mean, ln_rstd are INPUTS (what the forward wrote), ln_z [M, ln_ldz] the
   * normalisation's input, C2 the partials scratch, ln_dgamma / ln_dbeta the outputs (ln_dbeta == ln_dgamma + 64: the pair
   * is one strip, as in the flat gradient buffer) */
  const float* ln_z; int32_t ln_ldz; float* ln_dgamma; float* ln_dbeta;
  /* ... and for the embedding stage (ln_ids != NULL; ln_z unused): the normalisation's input is recomputed as
   * ln_table[ln_ids[row]] + ln_pos[row % ln_L] (ids outside [0, ln_V) read row 0) and dy first goes back through the
   * dropout that followed the LayerNorm (rng / drop_stream / drop_rate, element index row*64+col)
   * (bert4rec_encoder.py:186-199: embeddings -> LayerNorm -> dropout) */
  const int64_t* ln_ids; const float* ln_table; const float* ln_pos; int32_t ln_L, ln_V;
  float* C3; int32_t ldc3;       /* B4R_EPI_BIAS_GELU_LN: the pre-activation [M, N] */
  /* B4R_EPI_BIAS_GELU_LN only (NULL otherwise): gathered A rows, with the semantics of b4r_gather_rows -- row m of the product
   * reads row clamp(a_gather_idx[m], 0, a_gather_add_per - 1) + (m / a_gather_per) * a_gather_add_per of A (tfm MaskedLM gathers
   * the masked positions of every sequence, bert4rec_model.py:143); a_copy [M, a_copy_ld >= K] (optional) receives the gathered rows */
  const int64_t* a_gather_idx; int64_t a_gather_add_per; int32_t a_gather_per; float* a_copy; int32_t a_copy_ld;
} b4r_gemm_desc;
/* Arithmetic of the dense layers (process-wide switch; default B4R_GEMM_BF16X3):
 *   B4R_GEMM_F32     exact fp32 matrix cores (v_mfma_f32_32x32x2_f32), LDS-tiled
 *   B4R_GEMM_BF16X3  products with K <= 64 run on the bf16 matrix cores with a 3-term hi/lo split of both fp32 operands and
 *                    fp32 accumulation (~1e-5 of the fp32 result at 5.3x the matrix-core throughput, operands loaded
 *                    straight into registers); K > 64 and the weight-gradient products stay on the exact path */
enum { B4R_GEMM_F32 = 0, B4R_GEMM_BF16X3 = 1 };
int b4r_set_gemm_mode(int mode);
int b4r_get_gemm_mode(void);

/* dense layers of the encoder / MLM head (Keras Dense / EinsumDense / MultiHeadAttention projections) on the exact
 * fp32 matrix cores (v_mfma_f32_32x32x2_f32) */
int b4r_gemm_f32(const b4r_gemm_desc* d, b4r_stream_t stream);
/* 1 when b4r_gemm_f32 accepts this descriptor with B4R_EPI_BIAS_DROP_RES_LN, B4R_EPI_BIAS_GELU_LN or B4R_EPI_ADD_RES_LN_BWD (B4R_GEMM_BF16X3
 * mode, N == 64, K a multiple of 64, B as [K,N] for the former and [N,K] for the latter, no operand dropout, 16-byte
 * aligned operands), else 0: callers then issue B4R_EPI_BIAS_DROP_RES + b4r_ln_fwd, or B4R_EPI_ADD_RES + b4r_ln_bwd */
int b4r_gemm_ln_supported(const b4r_gemm_desc* d);
int64_t b4r_gemm_ln_bwd_partial_floats(int32_t M);

/* out[Mo,No] = A[R,Mo]^T . B[R,No]  (weight gradients), optional colsum[No] = sum_r B[r,:] (bias gradients).
 * Deterministic split over R: partial slabs in `scratch` then an ordered reduce.  b_dropout as above (index r*No+c). */
typedef struct b4r_gemm_tn_desc {
  const float* A; int32_t lda;
  const float* B; int32_t ldb;
  float* out; int32_t ldo;
  int32_t R, Mo, No;
  float* colsum;   /* [No] = sum_r B[r,:] (after the optional dropout) or NULL */
  float* colsum_a; /* [Mo] = sum_r A[r,:] or NULL                              */
  const uint32_t* rng; uint32_t drop_stream; float drop_rate; int32_t b_dropout;
  int32_t accumulate; /* 1: out += result (out must hold defined values) */
  /* optional (dgrad_out != NULL; zero otherwise): the input gradient of the same dense layer from the same pass over B,
   *   dgrad_out[R, Mo] = dropout(B) . dgrad_w^T   with dgrad_w [Mo, dgrad_ldw >= No] the forward weight whose gradient `out` is
   * (y = x.W: dW = x^T.dy and dx = dy.W^T read dy once), optionally times gelu'(dgrad_gelu_pre [R, dgrad_ldg >= Mo]) -- the
   * layer's input was gelu(pre).  Only No = 64 and Mo a multiple of 64 in the B4R_GEMM_BF16X3 mode
   * (b4r_gemm_tn_dgrad_supported), otherwise B4R_E_SHAPE: callers then issue b4r_gemm_f32 with a_dropout for dx */
  const float* dgrad_w; int32_t dgrad_ldw; float* dgrad_out; int32_t dgrad_ldo;
  const float* dgrad_gelu_pre; int32_t dgrad_ldg;
} b4r_gemm_tn_desc;
int64_t b4r_gemm_tn_scratch_floats(int32_t R, int32_t Mo, int32_t No);
int b4r_gemm_tn_f32(const b4r_gemm_tn_desc* d, float* scratch, b4r_stream_t stream);
int b4r_gemm_tn_dgrad_supported(const b4r_gemm_tn_desc* d);

/* Keras MultiHeadAttention core for one layer, head_dim 32: ctx = dropout(softmax(q k^T + (1-mask)*-1e9)) v
 * qkv [B*L, 3H] (q pre-scaled by 1/sqrt(d)), ctx [B*L, H], lse [B, heads, L]; input_mask int64 [B,L]. */
int b4r_attn_fwd(const float* qkv, const int64_t* input_mask, int32_t B, int32_t L, int32_t heads, float* ctx,
                 float* lse, const uint32_t* rng, uint32_t drop_stream, float drop_rate, uint32_t* keep_bits,
                 b4r_stream_t stream);
/* dqkv [B*L,3H] from dctx; dq is returned multiplied by qscale (gradient wrt the un-scaled query projection) */
int b4r_attn_bwd(const float* qkv, const int64_t* input_mask, const float* ctx, const float* lse, const float* dctx,
                 int32_t B, int32_t L, int32_t heads, float qscale, float* dqkv, const uint32_t* rng,
                 uint32_t drop_stream, float drop_rate, const uint32_t* keep_bits, b4r_stream_t stream);
/* keep_bits: uint32[b4r_attn_keep_words(B, L, heads)], 16-byte aligned.  In the B4R_GEMM_BF16X3 mode the forward stores
 * its attention-probability dropout decisions there and the backward of the same step reads them back (required when
 * dropout is active); the B4R_GEMM_F32 kernels regenerate the decisions from the hash and ignore the buffer.  Forward and
 * backward of one step must run in the same mode. */
int64_t b4r_attn_keep_words(int32_t B, int32_t L, int32_t heads);

/* ---- fused encoder-layer halves (hidden size 64, B4R_GEMM_BF16X3) ------------------------------------------------------
 * One Keras TransformerEncoderBlock call (bert4rec_encoder.py:136-147 constructs it, :220-222 calls it once per layer) is
 * two blocks here, each one launch forward: the attention block (b4r_attn_block_*) and the feed-forward block below.  The
 * feed-forward block keeps the [N, inner] intermediate on the chip in both directions:
 *   fwd:  z2 = x1 + dropout(gelu_erf(x1.W1 + b1).W2 + b2) ; x2 = LayerNorm(z2) * ln_gamma + ln_beta ; mean2 / rstd2 [N]
 *   bwd:  from dz2 = d loss / d z2:  dW1, db1, dW2, db2 and dz1 = LayerNorm'(dx1) through the LayerNorm that produced
 *         x1 = LN(z1) (statistics mean1 / rstd1, scale ln1_gamma), dx1 = (dropmask(dz2).W2^T * gelu'(x1.W1 + b1)).W1^T + dz2,
 *         plus that LayerNorm's dgamma / dbeta (dln1_gamma[0..63], then dbeta at dln1_gamma + 64, as in the flat gradient buffer).
 *         The pre-activation is recomputed from x1, nothing of size [N, inner] is stored by the forward.
 * Dropout: element index row*64 + col of site drop_stream (rng == NULL: off).  Gradient outputs are overwritten (deterministic
 * ordered sums over per-workgroup partials in `scratch`, b4r_ffn_block_bwd_scratch_floats floats, 16-byte aligned).
 * b4r_ffn_block_supported: hidden 64, inner 256, bf16x3 mode; other shapes get B4R_E_SHAPE (callers then use b4r_gemm_f32). */
/* The attention block, forward: one workgroup per sequence (hidden 64, 2 heads, L <= 256, bf16x3 mode; b4r_attn_block_supported):
 *   q,k,v = x.Wqkv + bqkv (q * 1/sqrt(32)) ; per head ctx = dropout(softmax(q k^T + (1 - mask) * -1e9)) v ;
 *   z1 = x + dropout(ctx.Wo + bo) ; x1 = LayerNorm(z1) * ln_gamma + ln_beta ; mean1 / rstd1
 * in ONE launch (= b4r_gemm_f32 BIAS_QSCALE + b4r_attn_fwd + b4r_gemm_f32 BIAS_DROP_RES_LN).  Saved for the backward:
 * ctx [B*L,H], lse [B,heads,L], keep_bits (b4r_attn_keep_words; required when probs_rate > 0) and, when qkv != NULL, qkv
 * [B*L,3H] in b4r_attn_fwd's layout (b4r_attn_bwd reads it).  Dropout sites: probs_stream on the probabilities (index as
 * b4r_attn_fwd), out_stream on the output projection (index row*64 + col); rng == NULL: off. */
typedef struct b4r_attn_block_desc {
  int32_t B, L, H, heads;
  const float* x;                                     /* [B*L,H] block input */
  const int64_t* input_mask;                          /* [B,L] */
  const float* Wqkv; const float* bqkv;               /* [H,3H] (query | key | value kernels), [3H] */
  const float* Wo; const float* bo;                   /* [H,H] attention_output/kernel, [H] */
  const float* ln_gamma; const float* ln_beta; float ln_eps;   /* self_attention_layer_norm */
  const uint32_t* rng; uint32_t probs_stream; float probs_rate; uint32_t out_stream; float out_rate;
  float* qkv; float* ctx; float* lse; uint32_t* keep_bits;
  float* z1; float* x1; float* mean1; float* rstd1;   /* z1 / mean1 / rstd1 may be NULL; x1 too when z1, mean1, rstd1 are given (the
                                                       * feed-forward block then forms x1 on load, b4r_ffn_desc.ln1_beta) */
  /* FIRST LAYER, optional (emb_ids != NULL; x is then ignored): the block forms its own input, the embedding stage of
   * bert4rec_encoder.py:198-214, x = dropout(LayerNorm(emb_table[id] + emb_pos[position]) * emb_gamma + emb_beta), and writes it to
   * emb_x [B*L,H] with the statistics emb_mean / emb_rstd [B*L] (may be NULL) -- what b4r_embed_ln_fwd does in a launch of its own.
   * emb_ids [B,L] int64 (out-of-range ids read row 0), emb_table [emb_vocab,H], emb_pos [>= L,H]; dropout site emb_stream at rate
   * emb_rate, element index row*64 + col (rng == NULL: off). */
  const int64_t* emb_ids; const float* emb_table; const float* emb_pos; const float* emb_gamma; const float* emb_beta;
  int32_t emb_vocab; float emb_eps; uint32_t emb_stream; float emb_rate;
  float* emb_x; float* emb_mean; float* emb_rstd;
  /* optional (hidden 64, 64 < L <= 224, out_slots <= 64, no emb_ids): only the rows clamp(out_slot_positions[b][j]), j < out_slots, of
   * this block's outputs are wanted (the last layer under B4R_FLAG_HEAD_ROWS_ONLY: every masked-LM slot of the batch, labelled or
   * padded).  ctx / z1 / x1 / mean1 / rstd1 / lse / keep_bits are then written for those tokens ONLY and only those queries are swept
   * (keys and values: every token).  A backward of such a forward must name a subset of these rows in dz1_slot_positions. */
  const int64_t* out_slot_positions; int32_t out_slots;
} b4r_attn_block_desc;
int32_t b4r_attn_block_supported(int32_t hidden_size, int32_t num_heads, int32_t L);
int b4r_attn_block_fwd(const b4r_attn_block_desc* d, b4r_stream_t stream);

/* The attention block, backward: one launch, one workgroup per sequence (b4r_attn_block_bwd_supported: as the forward, L <= 208).
 * From dz1 = d loss / d z1 (z1 = x + dropout(ctx.Wo + bo)), the forward's ctx / lse / keep_bits and the block input x:
 *   dqkv [B*L,3H]   gradient wrt the q | k | v projections (dq already times 1/sqrt(32)): dWqkv = x^T.dqkv and dbqkv = its column
 *                   sums are left to b4r_gemm_tn_f32, like dWo = ctx^T.dropmask(dz1) / dbo
 *   dx_prev [B*L,H] gradient wrt the INPUT of the LayerNorm that produced x:  LN'(dqkv.Wqkv^T + dz1) -- prev_z / prev_mean /
 *                   prev_rstd / prev_gamma describe that LayerNorm (the previous layer's output_layer_norm); for the first layer
 *                   pass emb_ids [B,L], emb_table [emb_vocab,H], emb_pos [>=L,H] instead of prev_z: x = dropout(LN(table[id] + pos))
 *                   (bert4rec_encoder.py:198-211; emb_stream / emb_rate: that dropout, element index row*64 + col)
 *   dprev_gamma     [128]: that LayerNorm's dgamma, then dbeta
 * q, k, v are recomputed from x (the forward need not store qkv), every score block is formed once, dK / dV accumulate in LDS in
 * a fixed order (bitwise reproducible).  scratch: b4r_attn_block_bwd_scratch_floats(B) floats. */
typedef struct b4r_attn_block_bwd_desc {
  int32_t B, L, H, heads;
  const float* x; const float* dz1; const float* ctx; const float* lse; const uint32_t* keep_bits;
  const int64_t* input_mask;
  const float* Wqkv; const float* bqkv; const float* Wo;
  const uint32_t* rng; uint32_t probs_stream; float probs_rate; uint32_t out_stream; float out_rate;
  const float* prev_z; const float* prev_mean; const float* prev_rstd; const float* prev_gamma;
  const int64_t* emb_ids; const float* emb_table; const float* emb_pos; int32_t emb_vocab; uint32_t emb_stream; float emb_rate;
  float* dqkv; float* dx_prev; float* dprev_gamma;
  float* scratch;
  /* optional (L <= 224): the weight gradients of the q | k | v projections formed inside the launch, dWqkv [H,3H] = x^T.dqkv and dbqkv
   * [3H] = its column sums (tape.gradient of bert4rec_encoder.py:220-222 wrt query / key / value kernel and bias) as ordered sums
   * over per-sequence partials in dw_scratch (b4r_attn_block_bwd_dw_scratch_floats(B) floats, 16-byte aligned); dqkv may then be
   * NULL: nothing of size [B*L,3H] is written.  With dWo / dbo given as well (they need the three above) the launch also forms
   * dWo [H,H] = ctx^T.dropmask(dz1) and dbo [H] (attention_output kernel / bias): no weight-gradient launch is left for the
   * attention half. */
  float* dWqkv; float* dbqkv; float* dw_scratch;
  float* dWo; float* dbo;
  /* optional (L <= 224): dz1 is SPARSE -- only the rows b*L + clamp(dz1_slot_positions[b][j]) with dz1_slot_ids[b][j] != 0 (j <
   * dz1_slots) carry a gradient, every other row counts as zero and is never read: the caller need not clear them.  The last encoder
   * layer of a train step (B4R_FLAG_HEAD_ROWS_ONLY): the masked-LM slots of the batch, bert4rec_model.py:76-81. */
  const int64_t* dz1_slot_positions; const int64_t* dz1_slot_ids; int32_t dz1_slots;
} b4r_attn_block_bwd_desc;
int32_t b4r_attn_block_bwd_supported(int32_t hidden_size, int32_t num_heads, int32_t L);
int64_t b4r_attn_block_bwd_scratch_floats(int32_t B);
int64_t b4r_attn_block_bwd_dw_scratch_floats(int32_t B);
/* tuning hook: the shortest sequence length at which the 32-token-tile attention kernels (b4r_attn32.hip) are chosen over the
 * 16-token-tile ones (default 65; environment B4R_ATTN32_MIN_L).  Returns the previous value; a negative argument only reads it. */
int32_t b4r_attn32_set_min_len(int32_t L);
/* tuning hook: b4r_attn_fwd on the 32-token-tile core as well (default 0: the forward stays on 16-token tiles, only b4r_attn_bwd
 * uses the core; environment B4R_ATTN32_CORE_FWD).  Returns the previous value; a negative argument only reads it. */
int32_t b4r_attn32_set_core_fwd(int32_t on);
int b4r_attn_block_bwd(const b4r_attn_block_bwd_desc* d, b4r_stream_t stream);

typedef struct b4r_ffn_desc {
  int32_t N, H, I;
  const float* x1;                                  /* [N,H] block input */
  const float* W1; const float* b1;                 /* [H,I], [I]  intermediate/kernel, bias */
  const float* W2; const float* b2;                 /* [I,H], [H]  output/kernel, bias */
  const float* ln_gamma; const float* ln_beta; float ln_eps;   /* output_layer_norm (forward only) */
  const uint32_t* rng; uint32_t drop_stream; float drop_rate;
  float* z2; float* x2; float* mean2; float* rstd2; /* forward outputs (z2 / mean2 / rstd2 may be NULL) */
  const float* dz2;                                 /* backward input [N,H] */
  const float* z1; const float* mean1; const float* rstd1; const float* ln1_gamma;
  float* dz1;                                       /* [N,H] */
  float* dW1; float* db1; float* dW2; float* db2; float* dln1_gamma;
  float* scratch;
  /* optional ROW LIST (all NULL / 0 otherwise): the block only processes rows rows[j], j < *n_rows (a DEVICE count; max_rows is its
   * host-side bound, used to size the launch).  The last encoder layer of a TRAIN step needs its feed-forward half only on the rows
   * the masked-LM head gathers (b4r_mlm_rows): the forward then writes x2 / z2 / mean2 / rstd2 of those rows only, and the backward
   * takes, instead of dz2, the head's gradient per masked-LM slot slot_grad [B*P,H] and row_slot[j] (the slot whose gradient entry j
   * carries, -1: none; rows may repeat among the entries with -1, never among the others): it forms dz2 = LayerNorm'(gradient of
   * x2) itself from z2 / mean2 / rstd2 / ln_gamma (inputs here), returns that LayerNorm's dgamma | dbeta in dln_gamma [128], writes
   * dz1 ONLY at the rows of entries with a slot (the caller zero-fills the rest: those rows carry no gradient) and uses dz2_rows
   * [max_rows,H] as scratch. */
  const int32_t* rows; const int32_t* n_rows; int32_t max_rows;
  const int32_t* row_slot; const float* slot_grad; float* dln_gamma; float* dz2_rows;
  /* SLOT MODE: the same list in its implicit form, straight from the batch (no b4r_mlm_rows launch): entry j = masked-LM slot j of
   * max_rows = B*P, rows[j] = (j / slots_per_seq) * seq_len + clamp(slot_positions[j]), row_slot[j] = j where slot_ids[j] != 0,
   * else -1.  rows / n_rows / row_slot stay NULL. */
  const int64_t* slot_positions; const int64_t* slot_ids; int32_t slots_per_seq, seq_len;
  /* x1 == NULL: the block input is formed on load, x1 = LayerNorm(z1) * ln1_gamma + ln1_beta from z1 / mean1 / rstd1 (inputs of the
   * forward too, then) -- the attention block need not store x1 at all (b4r_attn_block_desc.x1 = NULL). */
  const float* ln1_beta;
} b4r_ffn_desc;
/* The rows of the sequence output that the masked-LM head of this batch reads, one entry per masked-LM slot m = b*P + p:
 * rows[m] = b*L + clamp(position[m]) (padded slots gather position 0, as tfm MaskedLM does: their entries repeat a row, which the
 * forward then writes more than once with the same values), row_slot[m] = m where masked_lm_ids[m] != 0, else -1 (no gradient:
 * the backward writes nothing for that entry), *n_rows = B*P.  Two valid slots of one sequence never share a position
 * (dataloader_utils.py:221-226 samples positions without replacement), so every row has at most one entry that writes dz1. */
int b4r_mlm_rows(const int64_t* masked_lm_positions, const int64_t* masked_lm_ids, int32_t B, int32_t L, int32_t P, int32_t* rows,
                 int32_t* n_rows, int32_t* row_slot, b4r_stream_t stream);
int32_t b4r_ffn_block_supported(int32_t hidden_size, int32_t inner_dim);
int64_t b4r_ffn_block_bwd_scratch_floats(int32_t N);
int b4r_ffn_block_fwd(const b4r_ffn_desc* d, b4r_stream_t stream);
int b4r_ffn_block_bwd(const b4r_ffn_desc* d, b4r_stream_t stream);
/* The same half at hidden sizes 128 / 256 (the reference's *_128.json / *_256.json configurations; bert4rec_encoder.py:136-147,
 * :220-222), b4r_ffn32w.hip: the weights stream past 32-token row blocks held in registers, [N, inner] stays on the chip.
 *   b4r_ffn_wide_fwd: x1 -> z2 (may be NULL), x2, mean2 / rstd2 (may be NULL).  d->x1 given, no row list, d->scratch =
 *     b4r_ffn_wide_scratch_floats(H, I) floats (the packed weight records, valid until W1 / b1 / W2 change).  f / fpre: both NULL
 *     (inference) or both [N, I]: gelu(x1.W1 + b1) and its argument, the inputs of the backward.
 *   b4r_ffn_wide_bwd: dz2 -> dx1 [N,H] = dL/dx1 INCLUDING the residual path (the caller continues with LayerNorm1's backward),
 *     df [N, I] = dL/d(x1.W1 + b1) (the caller forms dW1 = x1^T.df, dW2 = f^T.dropmask(dz2) with b4r_gemm_tn_f32).  d->scratch as
 *     left by the forward of the same weights (records_ready != 0) or to be packed again (0). */
int32_t b4r_ffn_wide_supported(int32_t hidden_size, int32_t inner_dim);
int64_t b4r_ffn_wide_scratch_floats(int32_t hidden_size, int32_t inner_dim);
int b4r_ffn_wide_fwd(const b4r_ffn_desc* d, float* f, float* fpre, b4r_stream_t stream);
int b4r_ffn_wide_bwd(const b4r_ffn_desc* d, const float* fpre, float* df, float* dx1, int32_t records_ready, b4r_stream_t stream);

/* ---- one whole encoder layer ---------------------------------------------------------------------------------------------
 * One call of the Keras TransformerEncoderBlock (bert4rec_encoder.py:136-147 builds it post-LN, :220-222 calls it once per
 * layer) and its backward, sequence-resident: the two halves above back to back, nothing of size [N, 3H], [N, inner] or
 * [B, heads, L, L] crosses HBM in the forward, and the backward recomputes q / k / v and the pre-activation from x and x1.
 *   b4r_encoder_layer_fwd: attn (x -> ctx, lse, keep_bits, z1, x1, mean1, rstd1) then ffn (x1 -> z2, x2, mean2, rstd2);
 *     attn->x1 must be ffn->x1.  2 launches.
 *   b4r_encoder_layer_bwd: ffn (dz2 -> dz1, dW1, db1, dW2, db2, dln1_gamma), dWo / dbo = ctx^T.dropmask(dz1) and its column sums,
 *     attn (dz1 -> dqkv, dx_prev, dprev_gamma), dWqkv / dbqkv = x^T.dqkv.  attn->dz1 must be ffn->dz1.  5 launches + the ordered
 *     reduction.  tn_scratch: b4r_encoder_layer_bwd_scratch_floats(B * L) floats, 16-byte aligned.
 * b4r_encoder_layer_supported = both halves supported (hidden 64, 2 heads, inner 256, L <= 208, bf16x3 mode). */
int32_t b4r_encoder_layer_supported(int32_t hidden_size, int32_t num_heads, int32_t inner_dim, int32_t L);
int64_t b4r_encoder_layer_bwd_scratch_floats(int32_t N);
int b4r_encoder_layer_fwd(const b4r_attn_block_desc* attn, const b4r_ffn_desc* ffn, b4r_stream_t stream);
int b4r_encoder_layer_bwd(const b4r_ffn_desc* ffn, const b4r_attn_block_bwd_desc* attn, float* dWo, float* dbo, float* dWqkv,
                          float* dbqkv, float* tn_scratch, b4r_stream_t stream);

/* rows gather / scatter-add:  dst[i,:] = src[idx[i],:]   /   dst[idx[i],:] += src[i,:] (fp32 atomics) */
int b4r_gather_rows(const float* src, int32_t src_ld, const int64_t* idx, int64_t idx_add_per, int32_t per,
                    int32_t n, int32_t H, float* dst, b4r_stream_t stream);
int b4r_scatter_add_rows(const float* src, const int64_t* idx, int64_t idx_add_per, int32_t per, int32_t n, int32_t H,
                         float* dst, int32_t dst_ld, const int64_t* skip_if_zero, b4r_stream_t stream);

/* Masked-LM head of a train step without the [M, V] logits (hidden size H = 64, 128 or 256, B4R_GEMM_BF16X3 arithmetic;
 * what b4r_forward / b4r_backward run under B4R_FLAG_FUSED_HEAD).  T [M,H] transform output, E [V,H] tied table, bias [V],
 * y_true [M].
 *   b4r_mlm_head_fused_fwd: row_scratch[4*M] (as b4r_softmax_ce leaves it; b4r_loss then reduces it into the state),
 *     lse[M] (+inf for ignored slots), labels[M] (int32, -1 for ignored slots) and dT [M,H] = d loss_sum / d T.
 *     only_sweep != 0 runs the vocabulary sweep alone (partials in scratch) -- bench.py times that kernel.
 *   b4r_mlm_head_fused_bwd: dE [V,H] and dbias [V] (overwritten) from the forward's lse / labels.
 * scratch: b4r_mlm_head_fused_scratch_floats(M, V, H) floats, 16-byte aligned.  Replaces, for the train step only, the
 * tfm MaskedLM logits + trainer_utils.py:12-23 loss + their autograd (bert4rec_model.py:151-173). */
int64_t b4r_mlm_head_fused_scratch_floats(int32_t M, int32_t V, int32_t H);
int b4r_mlm_head_fused_fwd(const float* T, const float* E, const float* bias, const int64_t* y_true, int32_t M, int32_t V,
                           int32_t H, float* scratch, float* dT, float* row_scratch, float* lse, int32_t* labels,
                           int32_t only_sweep, b4r_stream_t stream);
int b4r_mlm_head_fused_bwd(const float* T, const float* E, const float* bias, const float* lse, const int32_t* labels,
                           int32_t M, int32_t V, int32_t H, float* scratch, float* dE, float* dbias, b4r_stream_t stream);

/* per-row softmax cross entropy over logits [M, ld] (V valid columns) + argmax metrics; row scalars then an ordered
 * single-workgroup reduction into the state.  want_grad: logits <- softmax - onehot for valid rows, 0 otherwise
 * (pad columns V..ld-1 are zeroed).   trainer_utils.py:12-23,49-60 */
int b4r_softmax_ce(float* logits, int32_t M, int32_t V, int32_t ld, const int64_t* y_true, float* row_scratch,
                   b4r_train_state* state, int32_t want_grad, b4r_stream_t stream);

/* sum of squares of a flat buffer into state->grad_sqnorm (deterministic), scratch >= 1024 floats */
int b4r_global_sqnorm(const float* g, int64_t n, float* scratch, b4r_train_state* state, b4r_stream_t stream);
/* fused clip + decoupled decay + Adam over the flat buffers; first n_decay floats are decayed.  Uses
 * state->{step, valid_count, grad_sqnorm}; writes state->{grad_norm, lr} and advances state->step. */
int b4r_adamw_step(const b4r_adamw_config* hp, float* params, const float* grads, float* adam_m, float* adam_v,
                   int64_t n, int64_t n_decay, b4r_train_state* state, b4r_stream_t stream);

/* ---- measurement aid (bench.py's roofline leg; not on any product path) -----------------------------------------------------
 * Between b4r_timing_begin and b4r_timing_end every kernel launch the library enqueues from the calling thread is followed by a
 * hipEvent on `stream`; b4r_timing_end waits for the last one and returns, per launch in enqueue order, the time since the previous
 * event (= the kernel's duration plus its launch boundary, kernels of one stream run back to back) and a label ("b4r_gemm_f32
 * (bf16x3) [M=51200 N=192 K=64 epi=2]").  names: capacity strings of name_stride bytes.  Not for use inside a graph capture. */
int b4r_timing_begin(b4r_stream_t stream, int32_t max_launches);
int b4r_timing_end(int32_t* n_launches, float* micros, char* names, int32_t name_stride, int32_t capacity);

#ifdef __cplusplus
}
#endif
#endif /* B4R_H_ */
