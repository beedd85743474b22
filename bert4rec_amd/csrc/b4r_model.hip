// Model-level entry points: parameter / workspace layout and the launch sequences of one forward, loss, backward and
// optimizer step.  Host code only: every function just enqueues kernels on the caller's stream (hipGraph-capturable).
//
// Follows  BERT4RecModel.call        bert4rec/models/bert4rec_model.py:110-149
//          Bert4RecEncoder.call      bert4rec/models/components/networks/bert4rec_encoder.py:186-231
//          (tfm TransformerEncoderBlock post-LN, Keras MultiHeadAttention, tfm MaskedLM: SURVEY.md §8 a4-a8)
//          train_step                bert4rec_model.py:151-173
#include <stdarg.h>
#include <stdio.h>
#include <string.h>

#include <string>
#include <vector>

#include <algorithm>
#include "b4r_common.h"

// ---- internal launchers defined in the other translation units --------------------------------------------------
int b4r_ln_bwd_launch(const float* dy, const float* z, const float* mean, const float* rstd, const float* gamma,
                      int rows, int H, float* dz, float* dgamma, float* dbeta, float* scratch, const int64_t* ids,
                      const float* table, const float* pos_table, int L, int V, DropArgs drop, hipStream_t stream,
                      const float* gelu_pre = nullptr, const B4rHeadMerge* merge = nullptr);
int b4r_head_rx_fwd_slices(int M, int V, int H);
int b4r_scatter_add_rows_impl(const float* src, const int64_t* idx, int64_t idx_add_per, int per, int n, int H,
                              float* dst, int dst_ld, const int64_t* skip_if_zero, int64_t dst_rows, int hot_rows,
                              float* hot_scratch, hipStream_t stream);
int64_t b4r_scatter_hot_scratch_floats(int hot_rows, int H);
int b4r_batch_colsum(const float* x, int B, int L, int H, float* dpos, float* scratch, hipStream_t stream);
int b4r_gemm_f32_splitk(const b4r_gemm_desc* d, int splits, float* scratch, int k_pad_ok, hipStream_t stream);
// b4r_head_rx.hip: masked-LM head of a train step without materialised logits (hidden size 64, bf16x3 mode)
bool b4r_head_rx_hidden_ok(int H);
int64_t b4r_head_rx_fwd_scratch_floats(int M, int V, int H);
int64_t b4r_head_rx_dE_scratch_floats(int M, int V, int H);
int b4r_head_rx_fwd_launch(const float* T, const float* E, const float* bias, const int64_t* y, int M, int V, int H,
                           float* scratch, float* dT, float* row_out, float* lse, int32_t* ylab, hipStream_t stream);
int b4r_head_rx_dE_launch(const float* T, const float* E, const float* bias, const float* lse, const int32_t* ylab, int M, int V,
                          int H, float* scratch, float* dE, float* db, hipStream_t stream, const float* fwd_part, const int64_t* y,
                          int records_ready);
int b4r_head_rx_dE_pack_job(const float* T, const float* lse, const int32_t* ylab, int M, int V, int H, float* scratch,
                            const float* fwd_part, const int64_t* y, void* out, size_t out_bytes, int* blocks);
bool b4r_head_rx_combine_foldable(int M, int V, int H);
int b4r_head_rx_fwd_launch2(const float* T, const float* E, const float* bias, const int64_t* y, int M, int V, int H,
                            float* scratch, float* dT, float* row_out, float* lse, int32_t* ylab, int only_sweep, hipStream_t stream);
int b4r_ffn_block_bwd_marked(const b4r_ffn_desc* d, hipStream_t stream, hipEvent_t after_dx);
// b4r_rowops.hip: the last layer's feed-forward half on the masked-LM head's rows (compact [B*P, .] operands)
int b4r_slot_rows_gather(const float* a, const float* b, const float* s0, const float* s1, const int64_t* pos, int L, int P, int M, int H,
                         float* ac, float* bc, float* s0c, float* s1c, hipStream_t s);
int b4r_slot_rows_drop(const float* src, const int64_t* pos, int L, int P, int M, int H, const DropArgs& drop, float* dst, hipStream_t s);
int b4r_slot_rows_tail(const float* y, const float* res, const int64_t* pos, int L, int P, int M, int H, const float* gamma, const float* beta,
                       float eps, const DropArgs& drop, float* z, float* mean, float* rstd, float* out, float* outc, hipStream_t s);
// b4r_attn32.hip: the attention core with the queries restricted to the masked-LM slots (compact [B*P, .] outputs)
bool b4r_attn32_slotq_supported(int L, int P);
int64_t b4r_attn32_slotq_keep_words(int B, int L, int heads, int P);
int b4r_attn32_slotq_fwd_launch(const float* qkv, const int64_t* mask, const int64_t* pos, int B, int L, int heads, int P, float* ctx_c,
                                float* lse_c, const DropArgs& drop, uint32_t* bits_c, hipStream_t stream);
int b4r_attn32_slotq_bwd_launch(const float* qkv, const int64_t* mask, const int64_t* pos, const int64_t* ids, const float* ctx_c,
                                const float* lse_c, const float* dctx_c, int B, int L, int heads, int P, float qscale, float* dqkv,
                                const DropArgs& drop, const uint32_t* bits_c, hipStream_t stream);
// b4r_ffn32w.hip: the feed-forward half at hidden sizes 128 / 256 as one launch (forward without a backward to follow)
bool b4r_ffn32w_supported(int H, int I);
int64_t b4r_ffn32w_rec_floats(int H, int I);
int b4r_ffn32w_fwd(const b4r_ffn_desc* d, float* recs, float* f, float* fpre, hipStream_t stream);
int b4r_ffn32w_bwd(const b4r_ffn_desc* d, float* recs, const float* fpre, float* df, float* dx1, bool records_ready, hipStream_t stream);
int b4r_embed_grads(const float* x, const int64_t* ids, int B, int L, int H, float* table_grad, int64_t V, int hot_rows,
                    float* fixed, float* dpos, float* colsum_scratch, hipStream_t stream, const float* fin_rows = nullptr,
                    int fin_M = 0, b4r_train_state* state = nullptr, float* tail = nullptr);
int64_t b4r_embed_fixed_floats(int64_t V, int H, int hot_rows);
int b4r_gemm_tn_pair(const b4r_gemm_tn_desc* d0, float* scratch0, const b4r_gemm_tn_desc* d1, float* scratch1, hipStream_t stream);
bool b4r_attn32_active(int H, int heads, int L);   // b4r_attn_block.hip: the 32-token-tile backward (it can form dWqkv / dbqkv itself)
int b4r_ce_finalize_launch(const float* row_scratch, int M, b4r_train_state* state, int overwrite, hipStream_t stream);
int b4r_attn_bwd_streams(const float* qkv, const int64_t* input_mask, const float* ctx, const float* lse, const float* dctx,
                         int32_t B, int32_t L, int32_t heads, float qscale, float* dqkv, const uint32_t* rng,
                         uint32_t drop_stream, float drop_rate, const uint32_t* keep_bits, hipStream_t stream,
                         hipStream_t stream_dkv);
int b4r_zero2(float* a, int64_t na, float* b, int64_t nb, hipStream_t stream, float* tail = nullptr, b4r_train_state* state = nullptr,
              const float* fin_rows = nullptr, int fin_M = 0, const void* rider = nullptr, int rider_blocks = 0);
int b4r_optimizer_fused(const b4r_adamw_config* hp, float* params, const float* grads, float* adam_m, float* adam_v, int64_t n,
                        int64_t n_decay, float* scratch, b4r_train_state* state, hipStream_t stream, int sums_from_tail = 0,
                        int np_given = 0);
// internal flag of b4r_backward (b4r_train_step sets it): the closing reduce launch also leaves the partial sums of squares of the
// gradients at the start of the workspace (dead by then) and their number in g_norm_np, so that the optimizer needs no norm launch
#define B4R_FLAG_NORM_PARTIALS_INTERNAL (1 << 20)
// internal flag of b4r_forward AND b4r_backward of one train step (b4r_train_step sets it on both; fused head + B4R_FLAG_LOSS_SUMS):
// the forward runs the head's vocabulary sweep only, the backward's dE launch merges the slices (dT, loss rows, lse, labels) in its
// prologue and the loss / metric sums are formed by one extra workgroup of the embedding-gradient launch -- one launch fewer
#define B4R_FLAG_DEFER_COMBINE_INTERNAL (1 << 21)
static thread_local int g_norm_np = 0;

// ---- error message (thread local) ---------------------------------------------------------------------------------
static thread_local char g_err[512] = "";
void b4r_set_error(const char* fmt, ...) {
  va_list ap;
  va_start(ap, fmt);
  vsnprintf(g_err, sizeof(g_err), fmt, ap);
  va_end(ap);
}
extern "C" size_t b4r_last_error(char* buf, size_t cap) {
  const size_t n = strlen(g_err);
  if (buf && cap) {
    const size_t c = n < cap - 1 ? n : cap - 1;
    memcpy(buf, g_err, c);
    buf[c] = 0;
  }
  return n;
}
extern "C" int b4r_version(void) { return B4R_VERSION; }

// ---- launch timing ----------------------------------------------------------------------------------------------------------
namespace {
struct TimingRec {
  bool on = false;
  hipStream_t stream = nullptr;
  std::vector<hipEvent_t> ev;       // ev[0] = begin, ev[i + 1] = after launch i
  std::vector<std::string> names;
  char detail[96] = "";
};
thread_local TimingRec g_tr;
}  // namespace

void b4r_timing_detail(const char* fmt, ...) {
  if (!g_tr.on) return;
  va_list ap;
  va_start(ap, fmt);
  vsnprintf(g_tr.detail, sizeof(g_tr.detail), fmt, ap);
  va_end(ap);
}
void b4r_timing_mark(const char* what) {
  if (!g_tr.on || g_tr.names.size() + 1 >= g_tr.ev.size()) return;
  std::string n(what);
  if (g_tr.detail[0]) { n += " ["; n += g_tr.detail; n += "]"; g_tr.detail[0] = 0; }
  if (hipEventRecord(g_tr.ev[g_tr.names.size() + 1], g_tr.stream) == hipSuccess) g_tr.names.push_back(n);
}
extern "C" int b4r_timing_begin(b4r_stream_t stream, int32_t max_launches) {
  B4R_CHECK_ARG(!g_tr.on, B4R_E_BADARG, "b4r_timing_begin: a recording is already active on this thread");
  B4R_CHECK_ARG(max_launches > 0 && max_launches <= 1 << 16, B4R_E_BADARG, "b4r_timing_begin: bad capacity");
  g_tr.ev.assign((size_t)max_launches + 1, nullptr);
  for (auto& e : g_tr.ev)
    if (hipEventCreate(&e) != hipSuccess) { b4r_set_error("b4r_timing_begin: hipEventCreate failed"); return B4R_E_HIP; }
  g_tr.names.clear();
  g_tr.stream = (hipStream_t)stream;
  g_tr.detail[0] = 0;
  if (hipEventRecord(g_tr.ev[0], g_tr.stream) != hipSuccess) { b4r_set_error("b4r_timing_begin: hipEventRecord failed"); return B4R_E_HIP; }
  g_tr.on = true;
  return B4R_OK;
}
extern "C" int b4r_timing_end(int32_t* n_launches, float* micros, char* names, int32_t name_stride, int32_t capacity) {
  B4R_CHECK_ARG(g_tr.on, B4R_E_BADARG, "b4r_timing_end: no recording is active on this thread");
  g_tr.on = false;
  const int n = (int)g_tr.names.size();
  int rc = B4R_OK;
  if (n > 0 && hipEventSynchronize(g_tr.ev[n]) != hipSuccess) { b4r_set_error("b4r_timing_end: hipEventSynchronize failed"); rc = B4R_E_HIP; }
  const int m = n < capacity ? n : capacity;
  for (int i = 0; i < m && rc == B4R_OK; ++i) {
    float ms = 0.f;
    if (hipEventElapsedTime(&ms, g_tr.ev[i], g_tr.ev[i + 1]) != hipSuccess) { b4r_set_error("b4r_timing_end: hipEventElapsedTime failed"); rc = B4R_E_HIP; break; }
    if (micros) micros[i] = ms * 1e3f;
    if (names && name_stride > 0) snprintf(names + (size_t)i * name_stride, (size_t)name_stride, "%s", g_tr.names[i].c_str());
  }
  for (auto e : g_tr.ev) if (e) (void)hipEventDestroy(e);
  g_tr.ev.clear();
  if (n_launches) *n_launches = m;
  return rc;
}

namespace {

inline int64_t up4(int64_t x) { return (x + 3) & ~(int64_t)3; }
inline int64_t up32(int64_t x) { return (x + 31) & ~(int64_t)31; }

int check_cfg(const b4r_model_config* c) {
  B4R_CHECK_ARG(c != nullptr, B4R_E_BADARG, "null model config");
  B4R_CHECK_ARG(c->vocab_size > 0 && c->num_layers > 0 && c->num_layers <= B4R_MAX_LAYERS && c->num_heads > 0 &&
                    c->inner_dim > 0 && c->max_seq_len > 0,
                B4R_E_SHAPE, "bad model config");
  B4R_CHECK_ARG(c->hidden_size == 32 * c->num_heads, B4R_E_SHAPE,
                "hidden_size %d / num_heads %d: head_dim must be 32", c->hidden_size, c->num_heads);
  const int H = c->hidden_size;
  B4R_CHECK_ARG(H == 32 || H == 64 || H == 128 || H == 256 || H == 512 || H == 1024, B4R_E_SHAPE,
                "hidden_size %d not supported (32,64,128,256,512,1024)", H);
  B4R_CHECK_ARG(c->inner_dim % 4 == 0, B4R_E_SHAPE, "inner_dim must be a multiple of 4");
  B4R_CHECK_ARG(c->output_dropout >= 0.f && c->output_dropout < 1.f && c->attention_dropout >= 0.f && c->attention_dropout < 1.f,
                B4R_E_BADARG, "dropout rates must be in [0,1)");
  return B4R_OK;
}

struct ParamEntry {
  std::string name;
  int64_t offset;
  int rows, cols, ld, decay;
};

struct ParamLayout {
  int64_t total = 0, n_decay = 0;
  int64_t word_emb = 0, pos_emb = 0, emb_ln_g = 0, emb_ln_b = 0;
  int64_t wqkv[B4R_MAX_LAYERS], wo[B4R_MAX_LAYERS], w1[B4R_MAX_LAYERS], w2[B4R_MAX_LAYERS];
  int64_t bqkv[B4R_MAX_LAYERS], bo[B4R_MAX_LAYERS], ln1_g[B4R_MAX_LAYERS], ln1_b[B4R_MAX_LAYERS], b1[B4R_MAX_LAYERS],
      b2[B4R_MAX_LAYERS], ln2_g[B4R_MAX_LAYERS], ln2_b[B4R_MAX_LAYERS];
  int64_t wd = 0, bd = 0, lnm_g = 0, lnm_b = 0, out_bias = 0;
  std::vector<ParamEntry> entries;
};

ParamLayout make_param_layout(const b4r_model_config& c) {
  ParamLayout p;
  const int64_t H = c.hidden_size, I = c.inner_dim, V = c.vocab_size, Lm = c.max_seq_len;
  int64_t off = 0;
  auto take = [&](int64_t n) { int64_t o = off; off += up4(n); return o; };
  auto add = [&](const std::string& name, int64_t o, int rows, int cols, int ld, int decay) {
    p.entries.push_back(ParamEntry{name, o, rows, cols, ld, decay});
  };
  // ---- weight-decayed region: kernels and the two embedding tables
  p.word_emb = take(V * H); add("word_embeddings/embeddings", p.word_emb, (int)V, (int)H, (int)H, 1);
  p.pos_emb = take(Lm * H); add("position_embedding/embeddings", p.pos_emb, (int)Lm, (int)H, (int)H, 1);
  for (int i = 0; i < c.num_layers; ++i) {
    const std::string pre = "transformer/layer_" + std::to_string(i);
    p.wqkv[i] = take(H * 3 * H);
    add(pre + "/self_attention/query/kernel", p.wqkv[i], (int)H, (int)H, (int)(3 * H), 1);
    add(pre + "/self_attention/key/kernel", p.wqkv[i] + H, (int)H, (int)H, (int)(3 * H), 1);
    add(pre + "/self_attention/value/kernel", p.wqkv[i] + 2 * H, (int)H, (int)H, (int)(3 * H), 1);
    p.wo[i] = take(H * H); add(pre + "/self_attention/attention_output/kernel", p.wo[i], (int)H, (int)H, (int)H, 1);
    p.w1[i] = take(H * I); add(pre + "/intermediate/kernel", p.w1[i], (int)H, (int)I, (int)I, 1);
    p.w2[i] = take(I * H); add(pre + "/output/kernel", p.w2[i], (int)I, (int)H, (int)H, 1);
  }
  p.wd = take(H * H); add("cls/predictions/transform/dense/kernel", p.wd, (int)H, (int)H, (int)H, 1);
  p.n_decay = off;
  // ---- not decayed: every bias and LayerNorm gamma/beta
  p.emb_ln_g = take(H); add("embeddings/layer_norm/gamma", p.emb_ln_g, 1, (int)H, (int)H, 0);
  p.emb_ln_b = take(H); add("embeddings/layer_norm/beta", p.emb_ln_b, 1, (int)H, (int)H, 0);
  for (int i = 0; i < c.num_layers; ++i) {
    const std::string pre = "transformer/layer_" + std::to_string(i);
    p.bqkv[i] = take(3 * H);
    add(pre + "/self_attention/query/bias", p.bqkv[i], 1, (int)H, (int)H, 0);
    add(pre + "/self_attention/key/bias", p.bqkv[i] + H, 1, (int)H, (int)H, 0);
    add(pre + "/self_attention/value/bias", p.bqkv[i] + 2 * H, 1, (int)H, (int)H, 0);
    p.bo[i] = take(H); add(pre + "/self_attention/attention_output/bias", p.bo[i], 1, (int)H, (int)H, 0);
    p.ln1_g[i] = take(H); add(pre + "/self_attention_layer_norm/gamma", p.ln1_g[i], 1, (int)H, (int)H, 0);
    p.ln1_b[i] = take(H); add(pre + "/self_attention_layer_norm/beta", p.ln1_b[i], 1, (int)H, (int)H, 0);
    p.b1[i] = take(I); add(pre + "/intermediate/bias", p.b1[i], 1, (int)I, (int)I, 0);
    p.b2[i] = take(H); add(pre + "/output/bias", p.b2[i], 1, (int)H, (int)H, 0);
    p.ln2_g[i] = take(H); add(pre + "/output_layer_norm/gamma", p.ln2_g[i], 1, (int)H, (int)H, 0);
    p.ln2_b[i] = take(H); add(pre + "/output_layer_norm/beta", p.ln2_b[i], 1, (int)H, (int)H, 0);
  }
  p.bd = take(H); add("cls/predictions/transform/dense/bias", p.bd, 1, (int)H, (int)H, 0);
  p.lnm_g = take(H); add("cls/predictions/transform/LayerNorm/gamma", p.lnm_g, 1, (int)H, (int)H, 0);
  p.lnm_b = take(H); add("cls/predictions/transform/LayerNorm/beta", p.lnm_b, 1, (int)H, (int)H, 0);
  p.out_bias = take(V); add("cls/predictions/output_bias/bias", p.out_bias, 1, (int)V, (int)V, 0);
  p.total = off;
  return p;
}

// K splits of dT = dlogits.E: enough workgroups to cover the chip (tiles of 128 rows x 64 columns)
int mlm_dt_splits(int64_t M, int64_t H, int64_t V) {
  const int64_t tiles = ((M + 127) / 128) * ((H + 63) / 64);
  int64_t s = (512 + tiles - 1) / tiles;
  const int64_t max_s = (V + 255) / 256;
  if (s > max_s) s = max_s;
  if (s > 16) s = 16;
  if (s < 1) s = 1;
  return (int)s;
}

struct WsLayout {
  int64_t total = 0;  // floats
  int64_t N = 0, M = 0, Vp = 0;
  int64_t x0, mean0, rstd0;
  int64_t qkv[B4R_MAX_LAYERS], lse[B4R_MAX_LAYERS], keep[B4R_MAX_LAYERS], ctx[B4R_MAX_LAYERS], z1[B4R_MAX_LAYERS], mean1[B4R_MAX_LAYERS],
      rstd1[B4R_MAX_LAYERS], x1[B4R_MAX_LAYERS], fpre[B4R_MAX_LAYERS], f[B4R_MAX_LAYERS], z2[B4R_MAX_LAYERS],
      mean2[B4R_MAX_LAYERS], rstd2[B4R_MAX_LAYERS], x2[B4R_MAX_LAYERS], ffnrec[B4R_MAX_LAYERS];
  int64_t gath, upre, u, meanm, rstdm, t, logits, rowsc, pooled, head_lse, head_ylab;
  int64_t dx, hot, da, db, dctx, dqkv, df, dt, dg;   // dx | hot | db adjacent: one fill clears dx + hot, or hot + db (row-list mode)
  int64_t dz2c, maxrows;   // row-list mode of the last layer's feed-forward half: one entry per masked-LM slot
  int64_t dzc_a, dzc_b;    // the same mode through the dense products (hidden sizes without the resident block): backward temporaries
  int64_t scratch, scratch_floats;
};

WsLayout make_ws_layout(const b4r_model_config& c, int B, int L, int P) {
  WsLayout w;
  const int64_t H = c.hidden_size, I = c.inner_dim, V = c.vocab_size;
  const int64_t N = (int64_t)B * L, M = (int64_t)B * (P > 0 ? P : 0);
  w.N = N; w.M = M; w.Vp = up32(V);
  int64_t off = 0;
  auto take = [&](int64_t n) { int64_t o = off; off += up4(n); return o; };
  w.x0 = take(N * H); w.mean0 = take(N); w.rstd0 = take(N);
  for (int i = 0; i < c.num_layers; ++i) {
    w.qkv[i] = take(N * 3 * H); w.lse[i] = take((int64_t)B * c.num_heads * L);
    w.keep[i] = take(b4r_attn_keep_words(B, L, c.num_heads));   // attention dropout decisions (uint32 words)
    w.ctx[i] = take(N * H);
    w.z1[i] = take(N * H); w.mean1[i] = take(N); w.rstd1[i] = take(N); w.x1[i] = take(N * H);
    w.fpre[i] = take(N * I); w.f[i] = take(N * I);
    w.z2[i] = take(N * H); w.mean2[i] = take(N); w.rstd2[i] = take(N); w.x2[i] = take(N * H);
    w.ffnrec[i] = take(b4r_ffn32w_supported((int)H, (int)I) ? b4r_ffn32w_rec_floats((int)H, (int)I) : 0);   // packed W1 / b1 / W2 records
  }
  w.gath = take(M * H); w.upre = take(M * H); w.u = take(M * H); w.meanm = take(M); w.rstdm = take(M);
  w.t = take(M * H); w.logits = take(M * w.Vp); w.rowsc = take(4 * M); w.pooled = take((int64_t)B * H);
  w.head_lse = take(M); w.head_ylab = take(M);
  w.dx = take(N * H); w.hot = take(b4r_embed_fixed_floats(c.vocab_size, (int)H, 3)); w.db = take(N * H); w.da = take(N * H); w.dctx = take(N * H);
  w.maxrows = M;
  w.dz2c = take(w.maxrows * H);
  w.dzc_a = take(M * H); w.dzc_b = take(M * H);
  w.dqkv = take(N * 3 * H); w.df = take(N * I); w.dt = take(M * H); w.dg = take(M * H);
  // scratch: every two-stage reduction of the backward pass keeps its partials until the single deferred reduce launch,
  // so the regions are summed (not max-ed); the two immediate reductions (split-K dT, position table) have their own
  int64_t s = 4096;
  auto add = [&](int64_t v) { s += up4(v); };
  for (int i = 0; i < c.num_layers; ++i) {
    add(b4r_gemm_tn_scratch_floats((int)N, (int)H, (int)(3 * H)));
    add(b4r_gemm_tn_scratch_floats((int)N, (int)H, (int)H));
    add(b4r_gemm_tn_scratch_floats((int)N, (int)H, (int)I));
    add(b4r_gemm_tn_scratch_floats((int)N, (int)I, (int)H));
    add(2 * std::max(b4r_ln_bwd_scratch_floats((int)N, (int)H), b4r_gemm_ln_bwd_partial_floats((int)N)));
    if (H == 64 && I == 256) add(b4r_ffn_block_bwd_scratch_floats((int)N));   // partial slabs of the fused feed-forward backward
    add(b4r_attn_block_bwd_scratch_floats(B));
    add(b4r_attn_block_bwd_dw_scratch_floats(B));
  }
  add(std::max(b4r_ln_bwd_scratch_floats((int)N, (int)H), b4r_gemm_ln_bwd_partial_floats((int)N)));
  if (M > 0) {
    add(b4r_gemm_tn_scratch_floats((int)M, (int)H, (int)I));   // the last layer's weight gradients over the head's rows only
    add(b4r_gemm_tn_scratch_floats((int)M, (int)I, (int)H));
    add(b4r_gemm_tn_scratch_floats((int)M, (int)V, (int)H));
    add(b4r_gemm_tn_scratch_floats((int)M, (int)H, (int)H));
    add(b4r_ln_bwd_scratch_floats((int)M, (int)H));
    add((int64_t)mlm_dt_splits(M, H, V) * M * H);
    if (b4r_head_rx_hidden_ok((int)H)) add(b4r_head_rx_dE_scratch_floats((int)M, (int)V, (int)H));
  }
  add((int64_t)b4r_cdiv(B, 16) * L * H);  // position-table gradient partials
  // the fused head's forward partials live at the start of the scratch region; a train step's backward merges them itself (its dE
  // launch), so they stay reserved in front of the backward's own regions
  if (M > 0 && b4r_head_rx_hidden_ok((int)H)) add(b4r_head_rx_fwd_scratch_floats((int)M, (int)V, (int)H));
  w.scratch = take(s); w.scratch_floats = s;
  w.total = off;
  return w;
}

int check_batch(const b4r_batch* b, const b4r_model_config* c, bool need_mlm) {
  B4R_CHECK_ARG(b != nullptr, B4R_E_BADARG, "null batch");
  B4R_CHECK_ARG(b->input_word_ids && b->input_mask, B4R_E_BADARG, "batch needs input_word_ids and input_mask");
  B4R_CHECK_ARG(b->B > 0 && b->L > 0 && b->P >= 0, B4R_E_SHAPE, "bad batch shape B=%d L=%d P=%d", b->B, b->L, b->P);
  B4R_CHECK_ARG(b->L <= c->max_seq_len, B4R_E_SHAPE, "sequence length %d exceeds max_sequence_length %d", b->L, c->max_seq_len);
  B4R_CHECK_ARG(b->L <= 256, B4R_E_SHAPE, "sequence length %d > 256 not supported", b->L);
  B4R_CHECK_ARG(!need_mlm || (b->masked_lm_positions && b->P > 0), B4R_E_BADARG, "batch needs masked_lm_positions");
  return B4R_OK;
}

#define RC(x)                 \
  do {                        \
    int rc__ = (x);           \
    if (rc__ != B4R_OK) return rc__; \
  } while (0)

int gemm(const float* A, int lda, const float* Bm, int ldb, float* C, int ldc, int M, int N, int K, int b_is_nk, int epi,
         const float* bias, float* C2, int ldc2, const float* R, int ldr, float qscale, int qcols, const uint32_t* rng,
         uint32_t stream_id, float rate, int a_dropout, hipStream_t s) {
  b4r_gemm_desc d{};
  d.A = A; d.lda = lda; d.B = Bm; d.ldb = ldb; d.C = C; d.ldc = ldc; d.M = M; d.N = N; d.K = K;
  d.b_is_nk = b_is_nk; d.epilogue = epi; d.bias = bias; d.C2 = C2; d.ldc2 = ldc2; d.R = R; d.ldr = ldr;
  d.qscale = qscale; d.qcols = qcols; d.rng = rng; d.drop_stream = stream_id; d.drop_rate = rate; d.a_dropout = a_dropout;
  d.c_pad_scratch = 1;  // every C of the model path is a workspace region whose pad columns are scratch
  return b4r_gemm_f32(&d, (b4r_stream_t)s);
}

// dense + bias + dropout + residual (-> z) + LayerNorm (-> y, mean, rstd): one launch where b4r_gemm_ln_supported (hidden
// size 64 in the bf16x3 mode), else the product with B4R_EPI_BIAS_DROP_RES followed by b4r_ln_fwd.  B4R_FUSE_LN=0: always two.
int dense_res_ln(const float* A, int lda, const float* W, float* z, float* y, float* mean, float* rstd, int M, int H, int K,
                 const float* bias, const float* R, const float* gamma, const float* beta, float eps, const uint32_t* rng,
                 uint32_t stream_id, float rate, hipStream_t s) {
  static const bool fuse = !(getenv("B4R_FUSE_LN") && atoi(getenv("B4R_FUSE_LN")) == 0);
  b4r_gemm_desc d{};
  d.A = A; d.lda = lda; d.B = W; d.ldb = H; d.C = z; d.ldc = H; d.M = M; d.N = H; d.K = K;
  d.epilogue = B4R_EPI_BIAS_DROP_RES_LN; d.bias = bias; d.C2 = y; d.ldc2 = H; d.R = R; d.ldr = H; d.qscale = 1.f;
  d.rng = rng; d.drop_stream = stream_id; d.drop_rate = rate; d.c_pad_scratch = 1;
  d.ln_gamma = gamma; d.ln_beta = beta; d.ln_mean = mean; d.ln_rstd = rstd; d.ln_eps = eps;
  if (fuse && b4r_gemm_ln_supported(&d)) return b4r_gemm_f32(&d, (b4r_stream_t)s);
  d.epilogue = B4R_EPI_BIAS_DROP_RES; d.C2 = nullptr; d.ldc2 = 0;
  RC(b4r_gemm_f32(&d, (b4r_stream_t)s));
  return b4r_ln_fwd(z, M, H, gamma, beta, eps, y, mean, rstd, (b4r_stream_t)s);
}

// input-gradient product + residual gradient + the LayerNorm backward in front of it:  dz = LN'(A.W^T + R)  (W as [N=H, K]).
// One launch where b4r_gemm_ln_supported, else B4R_EPI_ADD_RES into `dz` followed by b4r_ln_bwd in place.
// With `ids` the LayerNorm is the embedding stage's (input recomputed from the tables, dy first through the dropout `drop`).
int dgrad_ln_bwd(const float* A, int lda, const float* W, int K, const float* R, float* dz, int M, int H, const float* z,
                 const float* mean, const float* rstd, const float* gamma, float* dgamma, float* dbeta, float* scratch,
                 hipStream_t s, const int64_t* ids = nullptr, const float* table = nullptr, const float* pos_table = nullptr,
                 int L = 1, int V = 1, const uint32_t* rng = nullptr, uint32_t drop_stream = 0, float drop_rate = 0.f) {
  static const bool fuse = !(getenv("B4R_FUSE_LN") && atoi(getenv("B4R_FUSE_LN")) == 0);
  b4r_gemm_desc d{};
  d.A = A; d.lda = lda; d.B = W; d.ldb = K; d.C = dz; d.ldc = H; d.M = M; d.N = H; d.K = K; d.b_is_nk = 1;
  d.epilogue = B4R_EPI_ADD_RES_LN_BWD; d.R = R; d.ldr = H; d.qscale = 1.f; d.c_pad_scratch = 1;
  d.C2 = scratch; d.ln_gamma = gamma; d.ln_mean = const_cast<float*>(mean); d.ln_rstd = const_cast<float*>(rstd);
  d.ln_z = z; d.ln_ldz = H; d.ln_dgamma = dgamma; d.ln_dbeta = dbeta;
  d.ln_ids = ids; d.ln_table = table; d.ln_pos = pos_table; d.ln_L = L; d.ln_V = V;
  d.rng = rng; d.drop_stream = drop_stream; d.drop_rate = drop_rate;
  if (fuse && dbeta == dgamma + 64 && b4r_gemm_ln_supported(&d)) return b4r_gemm_f32(&d, (b4r_stream_t)s);
  d.epilogue = B4R_EPI_ADD_RES; d.C2 = nullptr; d.rng = nullptr; d.drop_rate = 0.f;
  RC(b4r_gemm_f32(&d, (b4r_stream_t)s));
  return b4r_ln_bwd_launch(dz, z, mean, rstd, gamma, M, H, dz, dgamma, dbeta, scratch, ids, table, pos_table, L, V,
                           b4r_make_drop(rng, drop_stream, drop_rate, 1), s, nullptr);
}

// the feed-forward half of a layer as one launch forward / two backward (b4r_ffn_rx.hip); B4R_FFN_FUSED=0: the separate
// dense launches of round 1 (kept for A/B timing and for shapes / modes the fused block does not cover)
bool ffn_fused(const b4r_model_config* c) {
  static const bool on = !(getenv("B4R_FFN_FUSED") && atoi(getenv("B4R_FFN_FUSED")) == 0);
  return on && b4r_ffn_block_supported(c->hidden_size, c->inner_dim) != 0;
}

// the attention half of a layer as one launch forward (b4r_attn_block.hip); B4R_ATTN_FUSED=0: the three launches of round 1
bool attn_fused(const b4r_model_config* c, int L) {
  static const bool on = !(getenv("B4R_ATTN_FUSED") && atoi(getenv("B4R_ATTN_FUSED")) == 0);
  return on && b4r_attn_block_supported(c->hidden_size, c->num_heads, L) != 0;
}

// ... and one launch backward (b4r_attn_block_bwd; then the forward need not store qkv); B4R_ATTN_BWD_FUSED=0: round 1's kernels
bool attn_bwd_fused(const b4r_model_config* c, int L) {
  static const bool on = !(getenv("B4R_ATTN_BWD_FUSED") && atoi(getenv("B4R_ATTN_BWD_FUSED")) == 0);
  return on && attn_fused(c, L) && b4r_attn_block_bwd_supported(c->hidden_size, c->num_heads, L) != 0;
}

// x1 = LayerNorm(z1) is not stored between the two fused halves of a layer: the feed-forward kernels form it on load (B4R_X1_ON_LOAD=0:
// the attention block writes it as before)
bool x1_on_load() {
  static const bool on = !(getenv("B4R_X1_ON_LOAD") && atoi(getenv("B4R_X1_ON_LOAD")) == 0);
  return on;
}

// B4R_FLAG_HEAD_ROWS_ONLY is honoured where the last layer's feed-forward half runs as the fused block and the row list fits
bool head_rows_ok(const b4r_model_config* c, const b4r_batch* b) {
  static const bool on = !(getenv("B4R_HEAD_ROWS") && atoi(getenv("B4R_HEAD_ROWS")) == 0);
  return on && ffn_fused(c) && b->masked_lm_positions && b->masked_lm_ids && b->P > 0;
}
// ... and, where that half runs as dense products (every hidden size but 64), with those products on compact [B*P, .] operands: the
// rows are gathered first (b4r_slot_rows_*).  Worth it when the head reads a minority of the rows (P = L / 5 at the benchmark shapes).
bool head_rows_dense_ok(const b4r_model_config* c, const b4r_batch* b) {
  static const bool on = !(getenv("B4R_HEAD_ROWS_DENSE") && atoi(getenv("B4R_HEAD_ROWS_DENSE")) == 0);
  return on && !ffn_fused(c) && c->num_layers > 0 && b->masked_lm_positions && b->masked_lm_ids && b->P > 0 && 2 * b->P <= b->L &&
         c->hidden_size % 32 == 0 && c->inner_dim >= 3 * c->hidden_size + 8;
}
// The compact operands of that mode live inside the last layer's own dense regions (an encoder-only forward has nothing else,
// b4r_workspace_bytes_encoder): f / fpre / z2 / mean2 / rstd2 at the start of theirs, and in the unused upper half of fpre [N, I]
// (2 M <= N, 3 H + 8 <= I): x1 rows, z1 rows, the second product's output, mean1, rstd1.
// The one-launch feed-forward pair of b4r_ffn32w.hip inside a TRAIN step (it then also writes f and the pre-activation): measured per
// dense layer at N = 51 200 -- hidden 128: forward 102 us against 121 (two tile products + LayerNorm), backward 103 against 124; hidden
// 256: 323 against 320 and 403 against 313.  So: hidden 128 only.  B4R_FFN32W_TRAIN = 0 never, 2 both sizes.
bool ffn32w_train_ok(const b4r_model_config* c) {
  static const int lv = getenv("B4R_FFN32W_TRAIN") ? atoi(getenv("B4R_FFN32W_TRAIN")) : 1;
  return lv > 0 && b4r_ffn32w_supported(c->hidden_size, c->inner_dim) && (c->hidden_size == 128 || lv > 1);
}
// ... and the attention half of that layer with the slots as its only queries (hidden sizes on the tile products; P <= 64)
bool slotq_layer(const b4r_model_config* c, const b4r_batch* b, uint32_t flags, int layer) {
  return (flags & B4R_FLAG_HEAD_ROWS_ONLY) && layer == c->num_layers - 1 && head_rows_dense_ok(c, b) && !attn_fused(c, b->L) &&
         b4r_attn32_slotq_supported(b->L, b->P) &&
         b4r_attn32_slotq_keep_words(b->B, b->L, c->num_heads, b->P) <= b4r_attn_keep_words(b->B, b->L, c->num_heads);
}
struct CompactRows { int64_t x1c, z1c, yc, mean1c, rstd1c; };
CompactRows compact_rows(const WsLayout& w, int layer, int64_t M, int64_t H, int64_t I) {
  CompactRows c;
  c.x1c = w.fpre[layer] + up4(M * I); c.z1c = c.x1c + M * H; c.yc = c.z1c + M * H; c.mean1c = c.yc + M * H; c.rstd1c = c.mean1c + up4(M);
  return c;
}

// pair kernels (input gradient inside the weight-gradient kernel, b4r_gemm_tn_desc.dgrad_*): B4R_PAIR bit 0 = the 64 x 64 layers
// (attention output, masked-LM transform), bit 1 = the FFN output layer with its GELU' tail
int pair_level() {
  static const int lv = getenv("B4R_PAIR") ? atoi(getenv("B4R_PAIR")) : 3;
  return lv;
}

// ---- a second stream for the branches of the backward pass that nothing downstream waits for -------------------------
// (weight-gradient products, the dE sweep of the fused head, dK/dV next to dQ).  The idea: every kernel of this workload
// leaves part of the chip idle at its start and tail, a concurrent independent kernel could fill those holes.  MEASURED
// (ML-1M step, one MI355X, same box, ms per step): single stream 0.943 | head dE only 0.958 | + dK/dV 0.970 | + all weight
// gradients 0.990 -- the event hand-offs between streams cost more than the overlap returns, and the concurrent kernels
// mostly take each other's CUs.  So it is OFF by default; B4R_SIDE_STREAM=1|2|3 re-enables the three levels.
struct SideStream {
  hipStream_t stream = nullptr;
  hipEvent_t ev[64];
  int next = 0, device = -1;
};
thread_local SideStream g_side;

int side_level() {   // 0: off; 1: everything independent; 2: head dE + dK/dV; 3: head dE only; 4: dWo / dWqkv of the fused-block path
  static const int lv = getenv("B4R_SIDE_STREAM") ? atoi(getenv("B4R_SIDE_STREAM")) : 0;
  return lv;
}

hipStream_t side_stream() {
  if (side_level() == 0) return nullptr;
  int dev = -1;
  if (hipGetDevice(&dev) != hipSuccess) return nullptr;
  if (g_side.stream == nullptr || g_side.device != dev) {
    if (hipStreamCreateWithFlags(&g_side.stream, hipStreamNonBlocking) != hipSuccess) { g_side.stream = nullptr; return nullptr; }
    for (int i = 0; i < 64; ++i)
      if (hipEventCreateWithFlags(&g_side.ev[i], hipEventDisableTiming) != hipSuccess) { g_side.stream = nullptr; return nullptr; }
    g_side.device = dev;
  }
  return g_side.stream;
}

// work enqueued on `to` after this call starts only when everything enqueued on `from` so far has finished
int order_after(hipStream_t from, hipStream_t to) {
  if (from == to) return B4R_OK;
  hipEvent_t e = g_side.ev[(g_side.next++) & 63];
  if (hipEventRecord(e, from) != hipSuccess || hipStreamWaitEvent(to, e, 0) != hipSuccess) {
    b4r_set_error("b4r_backward: stream ordering failed");
    return B4R_E_HIP;
  }
  return B4R_OK;
}

b4r_gemm_tn_desc tn_desc(const float* A, int lda, const float* Bm, int ldb, float* out, int ldo, int R, int Mo, int No, float* colsum,
                         const uint32_t* rng, uint32_t stream_id, float rate, int b_dropout) {
  b4r_gemm_tn_desc d{};
  d.A = A; d.lda = lda; d.B = Bm; d.ldb = ldb; d.out = out; d.ldo = ldo; d.R = R; d.Mo = Mo; d.No = No;
  d.colsum = colsum; d.rng = rng; d.drop_stream = stream_id; d.drop_rate = rate; d.b_dropout = b_dropout; d.accumulate = 0;
  return d;
}

hipEvent_t side_event() { return g_side.ev[(g_side.next++) & 63]; }
int side_mark(hipStream_t on, hipEvent_t* e) {
  *e = side_event();
  if (hipEventRecord(*e, on) != hipSuccess) { b4r_set_error("b4r_backward: event record failed"); return B4R_E_HIP; }
  return B4R_OK;
}
int side_wait(hipStream_t s, hipEvent_t e) {
  if (e != nullptr && hipStreamWaitEvent(s, e, 0) != hipSuccess) { b4r_set_error("b4r_backward: stream wait failed"); return B4R_E_HIP; }
  return B4R_OK;
}

int gemm_tn(const float* A, int lda, const float* Bm, int ldb, float* out, int ldo, int R, int Mo, int No, float* colsum,
            float* colsum_a, const uint32_t* rng, uint32_t stream_id, float rate, int b_dropout, float* scratch,
            hipStream_t s) {
  b4r_gemm_tn_desc d{};
  d.A = A; d.lda = lda; d.B = Bm; d.ldb = ldb; d.out = out; d.ldo = ldo; d.R = R; d.Mo = Mo; d.No = No;
  d.colsum = colsum; d.colsum_a = colsum_a; d.rng = rng; d.drop_stream = stream_id; d.drop_rate = rate;
  d.b_dropout = b_dropout; d.accumulate = 0;
  return b4r_gemm_tn_f32(&d, scratch, (b4r_stream_t)s);
}

}  // namespace

// ===============================================================================================================
// one encoder layer = the attention block + the feed-forward block (include/b4r.h)
extern "C" int32_t b4r_encoder_layer_supported(int32_t hidden_size, int32_t num_heads, int32_t inner_dim, int32_t L) {
  return (b4r_attn_block_supported(hidden_size, num_heads, L) && b4r_attn_block_bwd_supported(hidden_size, num_heads, L) &&
          b4r_ffn_block_supported(hidden_size, inner_dim)) ? 1 : 0;
}
extern "C" int64_t b4r_encoder_layer_bwd_scratch_floats(int32_t N) {
  return b4r_gemm_tn_scratch_floats(N, 64, 64) + b4r_gemm_tn_scratch_floats(N, 64, 192);
}
extern "C" int b4r_encoder_layer_fwd(const b4r_attn_block_desc* attn, const b4r_ffn_desc* ffn, b4r_stream_t stream) {
  B4R_CHECK_ARG(attn && ffn, B4R_E_BADARG, "b4r_encoder_layer_fwd: null descriptor");
  B4R_CHECK_ARG(((attn->x1 != nullptr && attn->x1 == ffn->x1) || (ffn->x1 == nullptr && attn->z1 != nullptr && attn->z1 == ffn->z1)) &&
                    (int64_t)attn->B * attn->L == ffn->N && attn->H == ffn->H,
                B4R_E_BADARG, "b4r_encoder_layer_fwd: the attention half's x1 (or, with x1 == NULL, its z1) [B*L,H] must be the "
                "feed-forward half's input");
  // (both halves behind each other in ONE launch were measured: 68 us against 39 + 25 us -- the feed-forward phase has to wait
  // for the attention half's stores at the barrier that frees the LDS, DESIGN.md section 4.1 -- and not kept)
  RC(b4r_attn_block_fwd(attn, stream));
  return b4r_ffn_block_fwd(ffn, stream);
}
extern "C" int b4r_encoder_layer_bwd(const b4r_ffn_desc* ffn, const b4r_attn_block_bwd_desc* attn, float* dWo, float* dbo, float* dWqkv,
                                     float* dbqkv, float* tn_scratch, b4r_stream_t stream) {
  B4R_CHECK_ARG(attn && ffn && dWo && dbo && dWqkv && dbqkv && tn_scratch, B4R_E_BADARG, "b4r_encoder_layer_bwd: null argument");
  B4R_CHECK_ARG(ffn->dz1 != nullptr && attn->dz1 == ffn->dz1 && (int64_t)attn->B * attn->L == ffn->N && attn->H == ffn->H, B4R_E_BADARG,
                "b4r_encoder_layer_bwd: the feed-forward half's dz1 [B*L,H] must be the attention half's input gradient");
  const int N = ffn->N, H = ffn->H;
  hipStream_t s = (hipStream_t)stream;
  RC(b4r_ffn_block_bwd(ffn, stream));
  RC(b4r_attn_block_bwd(attn, stream));
  if (attn->dWqkv != nullptr) {   // the attention block formed dWqkv / dbqkv itself (its descriptor's dWqkv: same buffers expected)
    B4R_CHECK_ARG(attn->dWqkv == dWqkv && attn->dbqkv == dbqkv && (attn->dWo == nullptr || (attn->dWo == dWo && attn->dbo == dbo)),
                  B4R_E_BADARG, "b4r_encoder_layer_bwd: the attention descriptor's dWqkv / dbqkv / dWo / dbo must be the call's");
    if (attn->dWo != nullptr) return B4R_OK;
    const b4r_gemm_tn_desc d1 = tn_desc(attn->ctx, H, attn->dz1, H, dWo, H, N, H, H, dbo, attn->out_rate > 0.f ? attn->rng : nullptr,
                                        attn->out_stream, attn->out_rate, 1);
    return b4r_gemm_tn_f32(&d1, tn_scratch, stream);
  }
  // dWo = ctx^T . dropmask(dz1) and dWqkv = x^T . dqkv (+ their bias gradients): one launch
  const b4r_gemm_tn_desc d_wo = tn_desc(attn->ctx, H, attn->dz1, H, dWo, H, N, H, H, dbo, attn->out_rate > 0.f ? attn->rng : nullptr,
                                        attn->out_stream, attn->out_rate, 1);
  const b4r_gemm_tn_desc d_wqkv = tn_desc(attn->x, H, attn->dqkv, 3 * H, dWqkv, 3 * H, N, H, 3 * H, dbqkv, nullptr, 0, 0.f, 0);
  return b4r_gemm_tn_pair(&d_wo, tn_scratch, &d_wqkv, tn_scratch + b4r_gemm_tn_scratch_floats(N, H, H), s);
}

// ===============================================================================================================
extern "C" int64_t b4r_param_total_floats(const b4r_model_config* cfg) {
  if (check_cfg(cfg)) return -1;
  return make_param_layout(*cfg).total;
}
extern "C" int64_t b4r_param_decay_floats(const b4r_model_config* cfg) {
  if (check_cfg(cfg)) return -1;
  return make_param_layout(*cfg).n_decay;
}
extern "C" int32_t b4r_param_count(const b4r_model_config* cfg) {
  if (check_cfg(cfg)) return -1;
  return (int32_t)make_param_layout(*cfg).entries.size();
}
extern "C" int b4r_param_info(const b4r_model_config* cfg, int32_t index, char* name, size_t name_cap, int64_t* offset,
                              int32_t* rows, int32_t* cols, int32_t* ld, int32_t* decay) {
  RC(check_cfg(cfg));
  const ParamLayout p = make_param_layout(*cfg);
  B4R_CHECK_ARG(index >= 0 && index < (int)p.entries.size(), B4R_E_BADARG, "b4r_param_info: index %d out of range", index);
  const ParamEntry& e = p.entries[index];
  if (name && name_cap) snprintf(name, name_cap, "%s", e.name.c_str());
  if (offset) *offset = e.offset;
  if (rows) *rows = e.rows;
  if (cols) *cols = e.cols;
  if (ld) *ld = e.ld;
  if (decay) *decay = e.decay;
  return B4R_OK;
}
extern "C" int64_t b4r_pooler_floats(const b4r_model_config* cfg) {
  if (check_cfg(cfg)) return -1;
  return (int64_t)cfg->hidden_size * cfg->hidden_size + cfg->hidden_size;
}
extern "C" int64_t b4r_workspace_bytes(const b4r_model_config* cfg, int32_t B, int32_t L, int32_t P) {
  if (check_cfg(cfg) || B <= 0 || L <= 0 || P < 0) return -1;
  return make_ws_layout(*cfg, B, L, P).total * (int64_t)sizeof(float);
}
extern "C" int64_t b4r_workspace_bytes_encoder(const b4r_model_config* cfg, int32_t B, int32_t L, int32_t P) {
  if (check_cfg(cfg) != B4R_OK || B <= 0 || L <= 0 || P < 0) return -1;
  return make_ws_layout(*cfg, B, L, P).gath * (int64_t)sizeof(float);
}

extern "C" int b4r_workspace_region(const b4r_model_config* cfg, int32_t B, int32_t L, int32_t P, const char* name,
                                    int64_t* offset_floats, int32_t* rows, int32_t* cols, int32_t* ld) {
  RC(check_cfg(cfg));
  B4R_CHECK_ARG(name && B > 0 && L > 0 && P >= 0, B4R_E_BADARG, "b4r_workspace_region: bad argument");
  const WsLayout w = make_ws_layout(*cfg, B, L, P);
  const int H = cfg->hidden_size, nl = cfg->num_layers;
  int64_t off = -1; int r = 0, c = 0, l = 0;
  const std::string n(name);
  if (n == "sequence_output") { off = w.x2[nl - 1]; r = (int)w.N; c = H; l = H; }
  else if (n == "embeddings") { off = w.x0; r = (int)w.N; c = H; l = H; }
  else if (n == "mlm_logits") { off = w.logits; r = (int)w.M; c = cfg->vocab_size; l = (int)w.Vp; }
  else if (n == "mlm_hidden") { off = w.t; r = (int)w.M; c = H; l = H; }
  else if (n == "pooled_output") { off = w.pooled; r = B; c = H; l = H; }
  else if (n == "grad_sequence_output") { off = w.dx; r = (int)w.N; c = H; l = H; }
  else if (n.rfind("encoder_output_", 0) == 0) {
    const int i = atoi(n.c_str() + 15);
    B4R_CHECK_ARG(i >= 0 && i < nl, B4R_E_BADARG, "b4r_workspace_region: no layer %d", i);
    off = w.x2[i]; r = (int)w.N; c = H; l = H;
  } else if (n.rfind("attention_context_", 0) == 0) {
    const int i = atoi(n.c_str() + 18);
    B4R_CHECK_ARG(i >= 0 && i < nl, B4R_E_BADARG, "b4r_workspace_region: no layer %d", i);
    off = w.ctx[i]; r = (int)w.N; c = H; l = H;
  }
  B4R_CHECK_ARG(off >= 0, B4R_E_BADARG, "b4r_workspace_region: unknown region '%s'", name);
  if (offset_floats) *offset_floats = off;
  if (rows) *rows = r;
  if (cols) *cols = c;
  if (ld) *ld = l;
  return B4R_OK;
}

// ===============================================================================================================
extern "C" int32_t b4r_fused_head_supported(const b4r_model_config* cfg) {
  return (cfg != nullptr && b4r_head_rx_hidden_ok(cfg->hidden_size) && b4r_get_gemm_mode() == B4R_GEMM_BF16X3) ? 1 : 0;
}

// The public entry points take the documented flags only: the internal bits (B4R_FLAG_*_INTERNAL) couple a forward and a backward
// of ONE b4r_train_step call and are set there alone.
static int forward_impl(const b4r_model_config* cfg, const b4r_batch* batch, const float* params, const float* pooler, void* workspace,
                        int64_t workspace_bytes, b4r_train_state* state, int32_t flags, b4r_stream_t stream);
static int backward_impl(const b4r_model_config* cfg, const b4r_batch* batch, const float* params, float* grads, void* workspace,
                         int64_t workspace_bytes, b4r_train_state* state, int32_t flags, b4r_stream_t stream);
constexpr int32_t B4R_PUBLIC_FLAGS = 0xFFFF;
extern "C" int b4r_forward(const b4r_model_config* cfg, const b4r_batch* batch, const float* params, const float* pooler,
                           void* workspace, int64_t workspace_bytes, b4r_train_state* state, int32_t flags,
                           b4r_stream_t stream) {
  return forward_impl(cfg, batch, params, pooler, workspace, workspace_bytes, state, flags & B4R_PUBLIC_FLAGS, stream);
}
extern "C" int b4r_backward(const b4r_model_config* cfg, const b4r_batch* batch, const float* params, float* grads,
                            void* workspace, int64_t workspace_bytes, b4r_train_state* state, int32_t flags,
                            b4r_stream_t stream) {
  return backward_impl(cfg, batch, params, grads, workspace, workspace_bytes, state, flags & B4R_PUBLIC_FLAGS, stream);
}

static int forward_impl(const b4r_model_config* cfg, const b4r_batch* batch, const float* params, const float* pooler, void* workspace,
                        int64_t workspace_bytes, b4r_train_state* state, int32_t flags, b4r_stream_t stream) {
  RC(check_cfg(cfg));
  RC(check_batch(batch, cfg, false));
  B4R_CHECK_ARG(params && workspace, B4R_E_BADARG, "b4r_forward: null params/workspace");
  B4R_CHECK_ARG(b4r_aligned16(params) && b4r_aligned16(workspace), B4R_E_ALIGN, "b4r_forward: buffers must be 16-byte aligned");
  const int B = batch->B, L = batch->L, P = batch->masked_lm_positions ? batch->P : 0;
  const ParamLayout pl = make_param_layout(*cfg);
  const WsLayout w = make_ws_layout(*cfg, B, L, batch->P);
  // an encoder-only forward without the pooler touches nothing behind the encoder's own regions (the masked-LM head's buffers -- the
  // [B*P, V] logits above all -- and the whole backward area): b4r_workspace_bytes_encoder is enough for it
  const int64_t ws_need = ((flags & B4R_FLAG_ENCODER_ONLY) && !(flags & B4R_FLAG_POOLER)) ? w.gath : w.total;
  B4R_CHECK_ARG(workspace_bytes >= ws_need * (int64_t)sizeof(float), B4R_E_NOMEM, "b4r_forward: workspace too small (%lld < %lld)",
                (long long)workspace_bytes, (long long)(ws_need * sizeof(float)));
  const int training = (flags & B4R_FLAG_TRAINING) ? 1 : 0;
  B4R_CHECK_ARG(!training || state || (cfg->output_dropout == 0.f && cfg->attention_dropout == 0.f), B4R_E_BADARG,
                "b4r_forward: training with dropout needs a state (rng)");
  const uint32_t* rng = training ? reinterpret_cast<const uint32_t*>(state) : nullptr;
  const float od = training ? cfg->output_dropout : 0.f, adp = training ? cfg->attention_dropout : 0.f;
  float* ws = static_cast<float*>(workspace);
  hipStream_t s = (hipStream_t)stream;
  const int H = cfg->hidden_size, I = cfg->inner_dim, V = cfg->vocab_size, N = B * L, M = B * P;
  const float qscale = 1.0f / sqrtf(32.0f);

  // the embedding stage: inside the first layer's attention block where that runs fused, else a launch of its own
  static const bool emb_in_block = !(getenv("B4R_EMB_FUSED") && atoi(getenv("B4R_EMB_FUSED")) == 0);
  const bool emb_fused = emb_in_block && attn_fused(cfg, L) && cfg->num_layers > 0;
  if (!emb_fused)
    RC(b4r_embed_ln_fwd(batch->input_word_ids, B, L, params + pl.word_emb, V, params + pl.pos_emb, params + pl.emb_ln_g,
                        params + pl.emb_ln_b, H, cfg->ln_eps, ws + w.x0, ws + w.mean0, ws + w.rstd0, rng, od, stream));
  const bool head_rows = (flags & B4R_FLAG_HEAD_ROWS_ONLY) && head_rows_ok(cfg, batch);
  const bool head_rows_dense = (flags & B4R_FLAG_HEAD_ROWS_ONLY) && head_rows_dense_ok(cfg, batch);
  const float* x = ws + w.x0;
  for (int i = 0; i < cfg->num_layers; ++i) {
    const bool layer_fused = attn_fused(cfg, L) && ffn_fused(cfg);
    b4r_attn_block_desc ad{};
    b4r_ffn_desc fd{};
    if (attn_fused(cfg, L)) {
      ad.B = B; ad.L = L; ad.H = H; ad.heads = cfg->num_heads; ad.x = x; ad.input_mask = batch->input_mask;
      ad.Wqkv = params + pl.wqkv[i]; ad.bqkv = params + pl.bqkv[i]; ad.Wo = params + pl.wo[i]; ad.bo = params + pl.bo[i];
      ad.ln_gamma = params + pl.ln1_g[i]; ad.ln_beta = params + pl.ln1_b[i]; ad.ln_eps = cfg->ln_eps;
      ad.rng = (od > 0.f || adp > 0.f) ? rng : nullptr;
      ad.probs_stream = B4R_STREAM_ATTN_PROBS(i); ad.probs_rate = adp; ad.out_stream = B4R_STREAM_ATTN_OUT(i); ad.out_rate = od;
      ad.qkv = attn_bwd_fused(cfg, L) ? nullptr : ws + w.qkv[i];   // only round 1's backward kernels read it
      ad.ctx = ws + w.ctx[i]; ad.lse = ws + w.lse[i]; ad.keep_bits = reinterpret_cast<uint32_t*>(ws + w.keep[i]);
      ad.z1 = ws + w.z1[i]; ad.mean1 = ws + w.mean1[i]; ad.rstd1 = ws + w.rstd1[i];
      ad.x1 = (layer_fused && x1_on_load()) ? nullptr : ws + w.x1[i];   // the fused feed-forward half forms x1 from z1 itself
      if (head_rows && i == cfg->num_layers - 1) {   // nothing but the head's rows leaves the last layer: only those queries are swept
        ad.out_slot_positions = batch->masked_lm_positions; ad.out_slots = batch->P;
      }
      if (i == 0 && emb_fused) {
        ad.emb_ids = batch->input_word_ids; ad.emb_table = params + pl.word_emb; ad.emb_pos = params + pl.pos_emb; ad.emb_vocab = V;
        ad.emb_gamma = params + pl.emb_ln_g; ad.emb_beta = params + pl.emb_ln_b; ad.emb_eps = cfg->ln_eps;
        ad.emb_stream = B4R_STREAM_EMB; ad.emb_rate = od; if (od > 0.f) ad.rng = rng;
        ad.emb_x = ws + w.x0; ad.emb_mean = ws + w.mean0; ad.emb_rstd = ws + w.rstd0;
      }
      if (!layer_fused) RC(b4r_attn_block_fwd(&ad, stream));
    } else {
    RC(gemm(x, H, params + pl.wqkv[i], 3 * H, ws + w.qkv[i], 3 * H, N, 3 * H, H, 0, B4R_EPI_BIAS_QSCALE, params + pl.bqkv[i],
            nullptr, 0, nullptr, 0, qscale, H, nullptr, 0, 0.f, 0, s));
    if (slotq_layer(cfg, batch, flags, i)) {
      // only the rows the head reads leave this layer: the attention core with the slots as its queries (keys: all tokens), then the
      // output projection, dropout, residual and LayerNorm on the compact [M, H] rows.  ctx / lse / decision words: compact, at the
      // start of the layer's dense regions; x1 / z1 / statistics: where the compact feed-forward half below expects them
      const CompactRows cr = compact_rows(w, i, M, H, I);
      float* xc = ws + w.x1[i];              // the layer input's rows (the residual)
      float* yc = ws + w.x1[i] + up4((int64_t)M * H);
      RC(b4r_attn32_slotq_fwd_launch(ws + w.qkv[i], batch->input_mask, batch->masked_lm_positions, B, L, cfg->num_heads, P, ws + w.ctx[i],
                                     ws + w.lse[i], b4r_make_drop(rng, B4R_STREAM_ATTN_PROBS(i), adp, 1),
                                     reinterpret_cast<uint32_t*>(ws + w.keep[i]), s));
      RC(b4r_slot_rows_gather(x, nullptr, nullptr, nullptr, batch->masked_lm_positions, L, P, M, H, xc, nullptr, nullptr, nullptr, s));
      RC(gemm(ws + w.ctx[i], H, params + pl.wo[i], H, yc, H, M, H, H, 0, B4R_EPI_BIAS, params + pl.bo[i], nullptr, 0, nullptr, 0, 1.f, 0,
              nullptr, 0, 0.f, 0, s));
      RC(b4r_slot_rows_tail(yc, xc, batch->masked_lm_positions, L, P, M, H, params + pl.ln1_g[i], params + pl.ln1_b[i], cfg->ln_eps,
                            b4r_make_drop(rng, B4R_STREAM_ATTN_OUT(i), od, 1), ws + cr.z1c, ws + cr.mean1c, ws + cr.rstd1c, nullptr,
                            ws + cr.x1c, s));
    } else {
    RC(b4r_attn_fwd(ws + w.qkv[i], batch->input_mask, B, L, cfg->num_heads, ws + w.ctx[i], ws + w.lse[i], rng,
                    B4R_STREAM_ATTN_PROBS(i), adp, reinterpret_cast<uint32_t*>(ws + w.keep[i]), stream));
    RC(dense_res_ln(ws + w.ctx[i], H, params + pl.wo[i], ws + w.z1[i], ws + w.x1[i], ws + w.mean1[i], ws + w.rstd1[i], N, H, H,
                    params + pl.bo[i], x, params + pl.ln1_g[i], params + pl.ln1_b[i], cfg->ln_eps, rng, B4R_STREAM_ATTN_OUT(i),
                    od, s));
    }
    }
    if (ffn_fused(cfg)) {
      fd.N = N; fd.H = H; fd.I = I; fd.x1 = (layer_fused && x1_on_load()) ? nullptr : ws + w.x1[i];
      fd.z1 = ws + w.z1[i]; fd.mean1 = ws + w.mean1[i]; fd.rstd1 = ws + w.rstd1[i];
      fd.ln1_gamma = params + pl.ln1_g[i]; fd.ln1_beta = params + pl.ln1_b[i];
      fd.W1 = params + pl.w1[i]; fd.b1 = params + pl.b1[i]; fd.W2 = params + pl.w2[i]; fd.b2 = params + pl.b2[i];
      fd.ln_gamma = params + pl.ln2_g[i]; fd.ln_beta = params + pl.ln2_b[i]; fd.ln_eps = cfg->ln_eps;
      fd.rng = od > 0.f ? rng : nullptr; fd.drop_stream = B4R_STREAM_FFN_OUT(i); fd.drop_rate = od;
      fd.z2 = ws + w.z2[i]; fd.x2 = ws + w.x2[i]; fd.mean2 = ws + w.mean2[i]; fd.rstd2 = ws + w.rstd2[i];
      if (head_rows && i == cfg->num_layers - 1) {   // only the rows the head reads: nothing else of this output is looked at
        fd.slot_positions = batch->masked_lm_positions; fd.slot_ids = batch->masked_lm_ids; fd.slots_per_seq = batch->P; fd.seq_len = L;
        fd.max_rows = (int32_t)w.maxrows;
      }
      if (layer_fused) RC(b4r_encoder_layer_fwd(&ad, &fd, stream));
      else RC(b4r_ffn_block_fwd(&fd, stream));
    } else if (head_rows_dense && i == cfg->num_layers - 1) {
      // only the rows the head reads: gather x1 (and, for the backward, z1 and its statistics), the two products on [M, .] operands,
      // then dropout + residual + LayerNorm per compact row with the result scattered to its row of x2.  The compact f / fpre / z2 /
      // mean2 / rstd2 lie at the start of the layer's dense regions (the backward of this mode reads them there).
      const CompactRows cr = compact_rows(w, i, M, H, I);
      if (!slotq_layer(cfg, batch, flags, i))   // (else the attention half above left the compact rows itself)
        RC(b4r_slot_rows_gather(ws + w.x1[i], ws + w.z1[i], ws + w.mean1[i], ws + w.rstd1[i], batch->masked_lm_positions, L, P, M, H,
                                ws + cr.x1c, ws + cr.z1c, ws + cr.mean1c, ws + cr.rstd1c, s));
      RC(gemm(ws + cr.x1c, H, params + pl.w1[i], I, ws + w.f[i], I, M, I, H, 0, B4R_EPI_BIAS_GELU, params + pl.b1[i], ws + w.fpre[i], I,
              nullptr, 0, 1.f, 0, nullptr, 0, 0.f, 0, s));
      RC(gemm(ws + w.f[i], I, params + pl.w2[i], H, ws + cr.yc, H, M, H, I, 0, B4R_EPI_BIAS, params + pl.b2[i], nullptr, 0, nullptr, 0,
              1.f, 0, nullptr, 0, 0.f, 0, s));
      RC(b4r_slot_rows_tail(ws + cr.yc, ws + cr.x1c, batch->masked_lm_positions, L, P, M, H, params + pl.ln2_g[i], params + pl.ln2_b[i],
                            cfg->ln_eps, b4r_make_drop(rng, B4R_STREAM_FFN_OUT(i), od, 1), ws + w.z2[i], ws + w.mean2[i], ws + w.rstd2[i],
                            ws + w.x2[i], nullptr, s));
    } else if (((flags & B4R_FLAG_ENCODER_ONLY) && b4r_ffn32w_supported(H, I)) || ffn32w_train_ok(cfg)) {
      // the one-launch form (b4r_ffn32w.hip).  No backward follows an encoder-only forward: [N, inner] stays on the chip; otherwise the
      // launch also writes f and the pre-activation, where the backward of this step expects them
      const bool keep = !(flags & B4R_FLAG_ENCODER_ONLY);
      fd.N = N; fd.H = H; fd.I = I; fd.x1 = ws + w.x1[i];
      fd.W1 = params + pl.w1[i]; fd.b1 = params + pl.b1[i]; fd.W2 = params + pl.w2[i]; fd.b2 = params + pl.b2[i];
      fd.ln_gamma = params + pl.ln2_g[i]; fd.ln_beta = params + pl.ln2_b[i]; fd.ln_eps = cfg->ln_eps;
      fd.rng = od > 0.f ? rng : nullptr; fd.drop_stream = B4R_STREAM_FFN_OUT(i); fd.drop_rate = od;
      fd.z2 = ws + w.z2[i]; fd.x2 = ws + w.x2[i]; fd.mean2 = ws + w.mean2[i]; fd.rstd2 = ws + w.rstd2[i];
      RC(b4r_ffn32w_fwd(&fd, ws + w.ffnrec[i], keep ? ws + w.f[i] : nullptr, keep ? ws + w.fpre[i] : nullptr, s));
    } else {
    RC(gemm(ws + w.x1[i], H, params + pl.w1[i], I, ws + w.f[i], I, N, I, H, 0, B4R_EPI_BIAS_GELU, params + pl.b1[i],
            ws + w.fpre[i], I, nullptr, 0, 1.f, 0, nullptr, 0, 0.f, 0, s));
    RC(dense_res_ln(ws + w.f[i], I, params + pl.w2[i], ws + w.z2[i], ws + w.x2[i], ws + w.mean2[i], ws + w.rstd2[i], N, H, I,
                    params + pl.b2[i], ws + w.x1[i], params + pl.ln2_g[i], params + pl.ln2_b[i], cfg->ln_eps, rng,
                    B4R_STREAM_FFN_OUT(i), od, s));
    }
    x = ws + w.x2[i];
  }
  if ((flags & B4R_FLAG_POOLER) && pooler) {
    // tanh(x[:,0,:] . Wp + bp): rows b of A are L*H apart
    RC(gemm(x, L * H, pooler, H, ws + w.pooled, H, B, H, H, 0, B4R_EPI_BIAS_TANH, pooler + (int64_t)H * H, nullptr, 0, nullptr,
            0, 1.f, 0, nullptr, 0, 0.f, 0, s));
  }
  if (P > 0 && !(flags & B4R_FLAG_ENCODER_ONLY)) {
    // tfm MaskedLM: gather -> dense(gelu) -> LayerNorm -> . E^T + bias
    {   // gather + dense(gelu) + LayerNorm: one launch where the LayerNorm tail applies (hidden size 64), else three
      static const bool fuse = !(getenv("B4R_FUSE_LN") && atoi(getenv("B4R_FUSE_LN")) == 0);
      b4r_gemm_desc d{};
      d.A = x; d.lda = H; d.B = params + pl.wd; d.ldb = H; d.C = ws + w.u; d.ldc = H; d.M = M; d.N = H; d.K = H;
      d.a_gather_idx = batch->masked_lm_positions; d.a_gather_add_per = L; d.a_gather_per = P;
      d.a_copy = ws + w.gath; d.a_copy_ld = H;   // the gathered rows: A operand of the transform's weight gradient
      d.epilogue = B4R_EPI_BIAS_GELU_LN; d.bias = params + pl.bd; d.C2 = ws + w.t; d.ldc2 = H; d.C3 = ws + w.upre; d.ldc3 = H;
      d.qscale = 1.f; d.c_pad_scratch = 1;
      d.ln_gamma = params + pl.lnm_g; d.ln_beta = params + pl.lnm_b; d.ln_mean = ws + w.meanm; d.ln_rstd = ws + w.rstdm;
      d.ln_eps = cfg->ln_eps;
      if (fuse && b4r_gemm_ln_supported(&d)) {
        RC(b4r_gemm_f32(&d, (b4r_stream_t)s));
      } else {
        RC(b4r_gather_rows(x, H, batch->masked_lm_positions, L, P, M, H, ws + w.gath, stream));
        RC(gemm(ws + w.gath, H, params + pl.wd, H, ws + w.u, H, M, H, H, 0, B4R_EPI_BIAS_GELU, params + pl.bd, ws + w.upre, H,
                nullptr, 0, 1.f, 0, nullptr, 0, 0.f, 0, s));
        RC(b4r_ln_fwd(ws + w.u, M, H, params + pl.lnm_g, params + pl.lnm_b, cfg->ln_eps, ws + w.t, ws + w.meanm, ws + w.rstdm,
                      stream));
      }
    }
    if (flags & B4R_FLAG_FUSED_HEAD) {
      // no [M,V] tensor: loss rows, log-sum-exp and d loss_sum / d T straight from T, E and the bias
      B4R_CHECK_ARG(b4r_fused_head_supported(cfg), B4R_E_BADARG, "b4r_forward: B4R_FLAG_FUSED_HEAD needs hidden size 64 / 128 / 256 and the bf16x3 mode");
      B4R_CHECK_ARG(batch->masked_lm_ids != nullptr, B4R_E_BADARG, "b4r_forward: B4R_FLAG_FUSED_HEAD needs masked_lm_ids");
      RC(b4r_head_rx_fwd_launch2(ws + w.t, params + pl.word_emb, params + pl.out_bias, batch->masked_lm_ids, M, V, H, ws + w.scratch,
                                 ws + w.dt, ws + w.rowsc, ws + w.head_lse, reinterpret_cast<int32_t*>(ws + w.head_ylab),
                                 (flags & B4R_FLAG_DEFER_COMBINE_INTERNAL) ? 1 : 0, s));
    } else {
      RC(gemm(ws + w.t, H, params + pl.word_emb, H, ws + w.logits, (int)w.Vp, M, V, H, 1, B4R_EPI_BIAS, params + pl.out_bias,
              nullptr, 0, nullptr, 0, 1.f, 0, nullptr, 0, 0.f, 0, s));
    }
  }
  return B4R_OK;
}

extern "C" int b4r_mlm_transform_rows(const b4r_model_config* cfg, const float* params, const float* seq, int64_t n_seq_rows,
                                      const int64_t* rows, int32_t R, float* out, float* scratch, b4r_stream_t stream) {
  RC(check_cfg(cfg));
  B4R_CHECK_ARG(params && seq && rows && out && scratch && R > 0 && n_seq_rows > 0, B4R_E_BADARG, "b4r_mlm_transform_rows: bad argument");
  B4R_CHECK_ARG(b4r_aligned16(params) && b4r_aligned16(seq) && b4r_aligned16(out) && b4r_aligned16(scratch), B4R_E_ALIGN,
                "b4r_mlm_transform_rows: buffers must be 16-byte aligned");
  const ParamLayout pl = make_param_layout(*cfg);
  const int H = cfg->hidden_size;
  hipStream_t s = (hipStream_t)stream;
  float* gath = scratch; float* upre = gath + up4((int64_t)R * H); float* u = upre + up4((int64_t)R * H);
  float* mean = u + up4((int64_t)R * H); float* rstd = mean + up4(R);
  b4r_gemm_desc d{};
  d.A = seq; d.lda = H; d.B = params + pl.wd; d.ldb = H; d.C = u; d.ldc = H; d.M = R; d.N = H; d.K = H;
  d.a_gather_idx = rows; d.a_gather_add_per = n_seq_rows; d.a_gather_per = R;   // one group: row m reads seq[rows[m]]
  d.epilogue = B4R_EPI_BIAS_GELU_LN; d.bias = params + pl.bd; d.C2 = out; d.ldc2 = H; d.C3 = upre; d.ldc3 = H;
  d.qscale = 1.f; d.c_pad_scratch = 0;
  d.ln_gamma = params + pl.lnm_g; d.ln_beta = params + pl.lnm_b; d.ln_mean = mean; d.ln_rstd = rstd; d.ln_eps = cfg->ln_eps;
  if (b4r_gemm_ln_supported(&d)) return b4r_gemm_f32(&d, stream);
  RC(b4r_gather_rows(seq, H, rows, n_seq_rows, R, R, H, gath, stream));
  RC(gemm(gath, H, params + pl.wd, H, u, H, R, H, H, 0, B4R_EPI_BIAS_GELU, params + pl.bd, upre, H, nullptr, 0, 1.f, 0, nullptr, 0,
          0.f, 0, s));
  return b4r_ln_fwd(u, R, H, params + pl.lnm_g, params + pl.lnm_b, cfg->ln_eps, out, mean, rstd, stream);
}

extern "C" int b4r_loss(const b4r_model_config* cfg, const b4r_batch* batch, void* workspace, int64_t workspace_bytes,
                        b4r_train_state* state, int32_t want_grad, b4r_stream_t stream) {
  RC(check_cfg(cfg));
  RC(check_batch(batch, cfg, true));
  B4R_CHECK_ARG(batch->masked_lm_ids && workspace && state, B4R_E_BADARG, "b4r_loss: needs masked_lm_ids, workspace, state");
  const WsLayout w = make_ws_layout(*cfg, batch->B, batch->L, batch->P);
  B4R_CHECK_ARG(workspace_bytes >= w.total * (int64_t)sizeof(float), B4R_E_NOMEM, "b4r_loss: workspace too small");
  float* ws = static_cast<float*>(workspace);
  if (want_grad & B4R_LOSS_FUSED_HEAD)   // the forward (B4R_FLAG_FUSED_HEAD) already produced the loss rows and dT
    return b4r_ce_finalize_launch(ws + w.rowsc, (int)w.M, state, (want_grad & B4R_LOSS_OVERWRITE) ? 1 : 0, (hipStream_t)stream);
  return b4r_softmax_ce(ws + w.logits, (int)w.M, cfg->vocab_size, (int)w.Vp, batch->masked_lm_ids, ws + w.rowsc, state,
                        want_grad & (1 | B4R_LOSS_OVERWRITE), stream);
}

static int backward_impl(const b4r_model_config* cfg, const b4r_batch* batch, const float* params, float* grads, void* workspace,
                         int64_t workspace_bytes, b4r_train_state* state, int32_t flags, b4r_stream_t stream) {
  RC(check_cfg(cfg));
  RC(check_batch(batch, cfg, true));
  B4R_CHECK_ARG(params && grads && workspace && batch->masked_lm_ids, B4R_E_BADARG, "b4r_backward: null argument");
  B4R_CHECK_ARG(b4r_aligned16(params) && b4r_aligned16(grads) && b4r_aligned16(workspace), B4R_E_ALIGN,
                "b4r_backward: buffers must be 16-byte aligned");
  const int B = batch->B, L = batch->L, P = batch->P;
  const ParamLayout pl = make_param_layout(*cfg);
  const WsLayout w = make_ws_layout(*cfg, B, L, P);
  B4R_CHECK_ARG(workspace_bytes >= w.total * (int64_t)sizeof(float), B4R_E_NOMEM, "b4r_backward: workspace too small");
  const int training = (flags & B4R_FLAG_TRAINING) ? 1 : 0;
  const uint32_t* rng = training ? reinterpret_cast<const uint32_t*>(state) : nullptr;
  const float od = training ? cfg->output_dropout : 0.f, adp = training ? cfg->attention_dropout : 0.f;
  float* ws = static_cast<float*>(workspace);
  hipStream_t s = (hipStream_t)stream;
  const int H = cfg->hidden_size, I = cfg->inner_dim, V = cfg->vocab_size, N = B * L, M = B * P, Vp = (int)w.Vp;
  const float qscale = 1.0f / sqrtf(32.0f);
  float* scratch_base = ws + w.scratch;
  int64_t scratch_used = 0;
  auto take = [&](int64_t n) { float* ptr = scratch_base + scratch_used; scratch_used += up4(n); return ptr; };
  const DropArgs nodrop = b4r_make_drop(nullptr, 0, 0.f, 0);
  B4rReduceQueue queue;
  b4r_reduce_queue_begin(&queue);   // every ordered reduction below is summed by ONE launch at the end

  // the gradient buffer; dx and the scatter's hot-row slots (row-list mode: the hot-row slots and db, whose rows outside the list
  // carry no gradient); with B4R_FLAG_GRAD_TAIL also the step's sums behind the gradients
  B4R_CHECK_ARG(!(flags & B4R_FLAG_GRAD_TAIL) || state, B4R_E_BADARG, "b4r_backward: B4R_FLAG_GRAD_TAIL needs the state");
  const bool head_rows_dense = (flags & B4R_FLAG_HEAD_ROWS_ONLY) && head_rows_dense_ok(cfg, batch);
  const bool head_rows = ((flags & B4R_FLAG_HEAD_ROWS_ONLY) && head_rows_ok(cfg, batch)) || head_rows_dense;
  const bool loss_sums = (flags & B4R_FLAG_LOSS_SUMS) != 0;
  const bool defer_combine = (flags & B4R_FLAG_DEFER_COMBINE_INTERNAL) && loss_sums;
  B4R_CHECK_ARG(!loss_sums || ((flags & B4R_FLAG_FUSED_HEAD) && state && batch->masked_lm_ids), B4R_E_BADARG,
                "b4r_backward: B4R_FLAG_LOSS_SUMS needs B4R_FLAG_FUSED_HEAD, the state and masked_lm_ids");
  // row-list mode with the 32-token-tile attention backward: that kernel is told which rows of the last layer's dz1 exist and never
  // reads the others -- db need not be cleared (13 MB per step at ML-1M)
  const bool sparse_dz1 = head_rows && ffn_fused(cfg) && attn_bwd_fused(cfg, L) && b4r_attn32_active(H, cfg->num_heads, L) &&
                          side_level() != 4;
  // the scratch regions of the fused head are fixed here already: the records dE sweeps (the transform rows as fp16 images, -lse, labels)
  // are formed by extra workgroups of the clearing launch (b4r_zero2's rider) instead of a launch of their own
  const bool fused_head_early = (flags & B4R_FLAG_FUSED_HEAD) != 0 && b4r_fused_head_supported(cfg);
  const float* fwd_part_early = (fused_head_early && defer_combine) ? take(b4r_head_rx_fwd_scratch_floats(M, V, H)) : nullptr;   // = ws + w.scratch
  float* dE_scratch = fused_head_early ? take(b4r_head_rx_dE_scratch_floats(M, V, H)) : nullptr;
  alignas(8) char rider[128];
  int rider_blocks = 0;
  if (fused_head_early)
    RC(b4r_head_rx_dE_pack_job(ws + w.t, ws + w.head_lse, reinterpret_cast<const int32_t*>(ws + w.head_ylab), M, V, H, dE_scratch,
                               fwd_part_early, batch->masked_lm_ids, rider, sizeof(rider), &rider_blocks));
  RC(b4r_zero2(grads, pl.total, ws + (head_rows ? w.hot : w.dx), head_rows ? (sparse_dz1 ? w.db - w.hot : w.da - w.hot) : w.db - w.dx, s,
               ((flags & B4R_FLAG_GRAD_TAIL) && !defer_combine) ? grads + pl.total : nullptr, state,
               (loss_sums && !defer_combine) ? ws + w.rowsc : nullptr, (int)w.M, rider_blocks > 0 ? rider : nullptr, rider_blocks));

  // ---- masked-LM head (logits buffer holds d loss_sum / d logits, pad columns zero) --------------------------------
  // s2: independent branches (see SideStream); it is ordered after the memsets here, joined before every reuse of a buffer
  // a branch reads (top of each layer) and before the final reductions
  hipStream_t s2 = side_stream();
  if (s2 == nullptr) s2 = s;
  hipStream_t s_tn = side_level() == 1 ? s2 : s;                          // weight-gradient products
  hipStream_t s_kv = (side_level() == 1 || side_level() == 2) ? s2 : s;  // dK/dV
  RC(order_after(s, s2));
  float* dlog = ws + w.logits;
  const bool fused_head = (flags & B4R_FLAG_FUSED_HEAD) != 0;
  const float* fwd_part = nullptr;
  B4R_CHECK_ARG(!fused_head || b4r_fused_head_supported(cfg), B4R_E_BADARG, "b4r_backward: B4R_FLAG_FUSED_HEAD needs hidden size 64 / 128 / 256 and the bf16x3 mode");
  if (fused_head) {
    // dT came with the forward -- or (defer_combine) the forward left its per-slice partials: dE forms the lse it needs from them, the
    // transform's LayerNorm backward below merges them into dT as it reads it; dE / d output_bias recompute the logit tiles (b4r_head_rx.hip)
    fwd_part = fwd_part_early;
    RC(b4r_head_rx_dE_launch(ws + w.t, params + pl.word_emb, params + pl.out_bias, ws + w.head_lse,
                             reinterpret_cast<const int32_t*>(ws + w.head_ylab), M, V, H, dE_scratch,
                             grads + pl.word_emb, grads + pl.out_bias, s2, fwd_part, batch->masked_lm_ids, rider_blocks > 0 ? 1 : 0));
  } else {
  // dT = dlogits . E   (K = V is long and the output small: split K so that the whole chip streams dlogits)
  {
    b4r_gemm_desc d{};
    d.A = dlog; d.lda = Vp; d.B = params + pl.word_emb; d.ldb = H; d.C = ws + w.dt; d.ldc = H;
    d.M = M; d.N = H; d.K = V; d.b_is_nk = 0; d.epilogue = B4R_EPI_NONE;
    // the loss kernel zeroed columns [V, Vp) of dlogits, and the table is followed by the position table in the flat
    // parameter buffer, so the reduction may run over whole chunks of 64 (rows V..Vp-1 of "E" meet zeros)
    const int k_pad_ok = (pl.word_emb + (int64_t)Vp * H <= pl.total) ? 1 : 0;
    RC(b4r_gemm_f32_splitk(&d, mlm_dt_splits(M, H, V), take((int64_t)mlm_dt_splits(M, H, V) * M * H), k_pad_ok, s));
  }
  // dE = dlogits^T . T ; d output_bias = column sums of dlogits
  RC(gemm_tn(dlog, Vp, ws + w.t, H, grads + pl.word_emb, H, M, V, H, nullptr, grads + pl.out_bias, nullptr, 0, 0.f, 0,
             take(b4r_gemm_tn_scratch_floats(M, V, H)), s));
  }
  // LayerNorm of the transform (with the deferred merge: dT, the loss rows, lse and labels are formed here, from the forward's partials)
  const B4rHeadMerge merge{fwd_part, fwd_part ? b4r_head_rx_fwd_slices(M, V, H) : 0, M, V, ws + w.t, params + pl.word_emb,
                           params + pl.out_bias, batch->masked_lm_ids, ws + w.rowsc, ws + w.head_lse,
                           reinterpret_cast<int32_t*>(ws + w.head_ylab)};
  RC(b4r_ln_bwd_launch(ws + w.dt, ws + w.u, ws + w.meanm, ws + w.rstdm, params + pl.lnm_g, M, H, ws + w.dt, grads + pl.lnm_g,
                       grads + pl.lnm_b, take(b4r_ln_bwd_scratch_floats(M, H)), nullptr, nullptr, nullptr, 1, 1, nodrop, s,
                       ws + w.upre, fwd_part ? &merge : nullptr));   // ... and straight through the GELU of the transform's dense layer
  {   // dense layer of the transform: dWd = gath^T . du (+ bias gradient) and dgath = du . Wd^T, one pass over du where the pair
      // kernel applies (hidden size 64)
    b4r_gemm_tn_desc d{};
    d.A = ws + w.gath; d.lda = H; d.B = ws + w.dt; d.ldb = H; d.out = grads + pl.wd; d.ldo = H; d.R = M; d.Mo = H; d.No = H;
    d.colsum = grads + pl.bd;
    d.dgrad_w = params + pl.wd; d.dgrad_ldw = H; d.dgrad_out = ws + w.dg; d.dgrad_ldo = H;
    if ((pair_level() & 1) && b4r_gemm_tn_dgrad_supported(&d)) {
      RC(b4r_gemm_tn_f32(&d, take(b4r_gemm_tn_scratch_floats(M, H, H)), (b4r_stream_t)s));
    } else {
      RC(order_after(s, s_tn));
      RC(gemm_tn(ws + w.gath, H, ws + w.dt, H, grads + pl.wd, H, M, H, H, grads + pl.bd, nullptr, nullptr, 0, 0.f, 0,
                 take(b4r_gemm_tn_scratch_floats(M, H, H)), s_tn));
      RC(gemm(ws + w.dt, H, params + pl.wd, H, ws + w.dg, H, M, H, H, 1, B4R_EPI_NONE, nullptr, nullptr, 0, nullptr, 0, 1.f, 0,
              nullptr, 0, 0.f, 0, s));
    }
  }
  // scatter into d sequence_output (slots with y_true == 0 carry exactly zero gradient and are skipped); in the row-list mode the last
  // layer's feed-forward backward reads the slot gradients directly
  if (!head_rows)
    RC(b4r_scatter_add_rows_impl(ws + w.dg, batch->masked_lm_positions, L, P, M, H, ws + w.dx, H, batch->masked_lm_ids, N, 0, nullptr, s));

  // ---- encoder layers, last to first ---------------------------------------------------------------------------------
  const int64_t ln_scratch = std::max(b4r_ln_bwd_scratch_floats(N, H), b4r_gemm_ln_bwd_partial_floats(N));
  const bool side4 = side_level() == 4 && s2 != s && ffn_fused(cfg) && attn_bwd_fused(cfg, L);
  hipEvent_t ev_dx = nullptr, ev_wo = nullptr, ev_wqkv = nullptr;
  for (int i = cfg->num_layers - 1; i >= 0; --i) {
    const float* x_in = (i == 0) ? ws + w.x0 : ws + w.x2[i - 1];
    RC(order_after(s_tn, s));   // the branches of the previous layer still read da / df / db / dqkv, which this layer rewrites
    if (side4) RC(side_wait(s, ev_wo));   // dWo of the layer above reads db, which this layer's feed-forward backward rewrites
    float* wo_scratch_of_layer = nullptr;
    // output LayerNorm (for every layer but the last its backward rode on the QKV input-gradient product of layer i + 1)
    const bool rows_here = head_rows && i == cfg->num_layers - 1;
    if (i == cfg->num_layers - 1 && !rows_here)
      RC(b4r_ln_bwd_launch(ws + w.dx, ws + w.z2[i], ws + w.mean2[i], ws + w.rstd2[i], params + pl.ln2_g[i], N, H, ws + w.da,
                           grads + pl.ln2_g[i], grads + pl.ln2_b[i], take(ln_scratch), nullptr, nullptr, nullptr, 1, 1, nodrop, s));
    if (ffn_fused(cfg)) {
      // feed-forward block: dz1 (-> db), dW1 / db1 / dW2 / db2 and the attention LayerNorm's gamma / beta gradients from dz2 (da);
      // the [N, inner] pre-activation is recomputed from x1 inside the two kernels
      b4r_ffn_desc fd{};
      fd.N = N; fd.H = H; fd.I = I;
      fd.x1 = (attn_fused(cfg, L) && x1_on_load()) ? nullptr : ws + w.x1[i];   // as the forward of this step left it
      fd.W1 = params + pl.w1[i]; fd.b1 = params + pl.b1[i]; fd.W2 = params + pl.w2[i]; fd.b2 = params + pl.b2[i];
      fd.rng = od > 0.f ? rng : nullptr; fd.drop_stream = B4R_STREAM_FFN_OUT(i); fd.drop_rate = od;
      fd.dz2 = ws + w.da; fd.z1 = ws + w.z1[i]; fd.mean1 = ws + w.mean1[i]; fd.rstd1 = ws + w.rstd1[i];
      fd.ln1_gamma = params + pl.ln1_g[i]; fd.ln1_beta = params + pl.ln1_b[i]; fd.dz1 = ws + w.db;
      fd.dW1 = grads + pl.w1[i]; fd.db1 = grads + pl.b1[i]; fd.dW2 = grads + pl.w2[i]; fd.db2 = grads + pl.b2[i];
      fd.dln1_gamma = grads + pl.ln1_g[i];
      fd.scratch = take(b4r_ffn_block_bwd_scratch_floats(N));
      if (rows_here) {   // the output LayerNorm's backward runs inside, on the rows with a gradient; dz1 elsewhere stays zero
        fd.dz2 = nullptr;
        fd.slot_positions = batch->masked_lm_positions; fd.slot_ids = batch->masked_lm_ids; fd.slots_per_seq = batch->P; fd.seq_len = L;
        fd.max_rows = (int32_t)w.maxrows;
        fd.slot_grad = ws + w.dg; fd.z2 = ws + w.z2[i]; fd.mean2 = ws + w.mean2[i]; fd.rstd2 = ws + w.rstd2[i];
        fd.ln_gamma = params + pl.ln2_g[i]; fd.dln_gamma = grads + pl.ln2_g[i]; fd.dz2_rows = ws + w.dz2c;
      }
      if (side4) {
        ev_dx = side_event();
        RC(b4r_ffn_block_bwd_marked(&fd, s, ev_dx));
      } else {
        RC(b4r_ffn_block_bwd(&fd, stream));
      }
    } else if (rows_here) {
      // the compact form of the chain below: every operand is [M, .], one row per masked-LM slot (slots without a label carry an exactly
      // zero gradient; two labelled slots never share a row).  dz1 (db) was cleared by the opening launch; the compact result is
      // scatter-added into it.
      const CompactRows cr = compact_rows(w, i, M, H, I);
      float* dz2c = ws + w.dzc_a;                 // LayerNorm2 backward of the slot gradients
      float* dz2d = od > 0.f ? ws + w.dzc_b : dz2c;   // ... through the output dropout
      RC(b4r_ln_bwd_launch(ws + w.dg, ws + w.z2[i], ws + w.mean2[i], ws + w.rstd2[i], params + pl.ln2_g[i], M, H, dz2c,
                           grads + pl.ln2_g[i], grads + pl.ln2_b[i], take(ln_scratch), nullptr, nullptr, nullptr, 1, 1, nodrop, s));
      if (od > 0.f)
        RC(b4r_slot_rows_drop(dz2c, batch->masked_lm_positions, L, P, M, H, b4r_make_drop(rng, B4R_STREAM_FFN_OUT(i), od, 1), dz2d, s));
      RC(gemm(dz2d, H, params + pl.w2[i], H, ws + w.df, I, M, I, H, 1, B4R_EPI_GELU_BWD, nullptr, nullptr, 0, ws + w.fpre[i], I, 1.f, 0,
              nullptr, 0, 0.f, 0, s));
      RC(order_after(s, s_tn));
      RC(gemm_tn(ws + w.f[i], I, dz2d, H, grads + pl.w2[i], H, M, I, H, grads + pl.b2[i], nullptr, nullptr, 0, 0.f, 0,
                 take(b4r_gemm_tn_scratch_floats(M, I, H)), s_tn));
      RC(dgrad_ln_bwd(ws + w.df, I, params + pl.w1[i], I, dz2c, ws + w.dz2c, M, H, ws + cr.z1c, ws + cr.mean1c, ws + cr.rstd1c,
                      params + pl.ln1_g[i], grads + pl.ln1_g[i], grads + pl.ln1_b[i], take(ln_scratch), s));
      RC(order_after(s, s_tn));
      RC(gemm_tn(ws + cr.x1c, H, ws + w.df, I, grads + pl.w1[i], I, M, H, I, grads + pl.b1[i], nullptr, nullptr, 0, 0.f, 0,
                 take(b4r_gemm_tn_scratch_floats(M, H, I)), s_tn));
      RC(b4r_scatter_add_rows_impl(ws + w.dz2c, batch->masked_lm_positions, L, P, M, H, ws + w.db, H, batch->masked_lm_ids, N, 0, nullptr, s));
    } else if (ffn32w_train_ok(cfg)) {
      // dF and dX1 (residual included) in one launch from the records the forward packed; LayerNorm1's backward in place; the two
      // weight gradients as before
      b4r_ffn_desc fd{};
      fd.N = N; fd.H = H; fd.I = I;
      fd.W1 = params + pl.w1[i]; fd.b1 = params + pl.b1[i]; fd.W2 = params + pl.w2[i]; fd.b2 = params + pl.b2[i];
      fd.rng = od > 0.f ? rng : nullptr; fd.drop_stream = B4R_STREAM_FFN_OUT(i); fd.drop_rate = od;
      fd.dz2 = ws + w.da;
      RC(b4r_ffn32w_bwd(&fd, ws + w.ffnrec[i], ws + w.fpre[i], ws + w.df, ws + w.db, true, s));
      RC(order_after(s, s_tn));
      RC(gemm_tn(ws + w.f[i], I, ws + w.da, H, grads + pl.w2[i], H, N, I, H, grads + pl.b2[i], nullptr, rng, B4R_STREAM_FFN_OUT(i), od, 1,
                 take(b4r_gemm_tn_scratch_floats(N, I, H)), s_tn));
      RC(b4r_ln_bwd_launch(ws + w.db, ws + w.z1[i], ws + w.mean1[i], ws + w.rstd1[i], params + pl.ln1_g[i], N, H, ws + w.db,
                           grads + pl.ln1_g[i], grads + pl.ln1_b[i], take(ln_scratch), nullptr, nullptr, nullptr, 1, 1, nodrop, s));
      RC(gemm_tn(ws + w.x1[i], H, ws + w.df, I, grads + pl.w1[i], I, N, H, I, grads + pl.b1[i], nullptr, nullptr, 0, 0.f, 0,
                 take(b4r_gemm_tn_scratch_floats(N, H, I)), s_tn));
    } else {
    // FFN: dFpre = (dropmask(dz2) . W2^T) * gelu'(fpre) and dW2 = f^T . dropmask(dz2) (+ bias gradient): one pass over dz2
    // where the pair kernel applies (B4R_PAIR bit 1), else two products
    {
      b4r_gemm_tn_desc d{};
      d.A = ws + w.f[i]; d.lda = I; d.B = ws + w.da; d.ldb = H; d.out = grads + pl.w2[i]; d.ldo = H; d.R = N; d.Mo = I; d.No = H;
      d.colsum = grads + pl.b2[i]; d.rng = rng; d.drop_stream = B4R_STREAM_FFN_OUT(i); d.drop_rate = od; d.b_dropout = 1;
      d.dgrad_w = params + pl.w2[i]; d.dgrad_ldw = H; d.dgrad_out = ws + w.df; d.dgrad_ldo = I;
      d.dgrad_gelu_pre = ws + w.fpre[i]; d.dgrad_ldg = I;
      if ((pair_level() & 2) && b4r_gemm_tn_dgrad_supported(&d)) {
        RC(b4r_gemm_tn_f32(&d, take(b4r_gemm_tn_scratch_floats(N, I, H)), (b4r_stream_t)s));
      } else {
        RC(gemm(ws + w.da, H, params + pl.w2[i], H, ws + w.df, I, N, I, H, 1, B4R_EPI_GELU_BWD, nullptr, nullptr, 0, ws + w.fpre[i],
                I, 1.f, 0, rng, B4R_STREAM_FFN_OUT(i), od, 1, s));
        RC(order_after(s, s_tn));
        RC(gemm_tn(ws + w.f[i], I, ws + w.da, H, grads + pl.w2[i], H, N, I, H, grads + pl.b2[i], nullptr, rng,
                   B4R_STREAM_FFN_OUT(i), od, 1, take(b4r_gemm_tn_scratch_floats(N, I, H)), s_tn));
      }
    }
    // dz1 = attention LayerNorm backward of dX1 = dFpre . W1^T + dz2
    RC(dgrad_ln_bwd(ws + w.df, I, params + pl.w1[i], I, ws + w.da, ws + w.db, N, H, ws + w.z1[i], ws + w.mean1[i], ws + w.rstd1[i],
                    params + pl.ln1_g[i], grads + pl.ln1_g[i], grads + pl.ln1_b[i], take(ln_scratch), s));
    RC(order_after(s, s_tn));
    RC(gemm_tn(ws + w.x1[i], H, ws + w.df, I, grads + pl.w1[i], I, N, H, I, grads + pl.b1[i], nullptr, nullptr, 0, 0.f, 0,
               take(b4r_gemm_tn_scratch_floats(N, H, I)), s_tn));
    }
    const bool dw_folded = attn_bwd_fused(cfg, L) && b4r_attn32_active(H, cfg->num_heads, L);
    if (attn_bwd_fused(cfg, L)) {
      // dWo = ctx^T . dropmask(dz1) (+ bias gradient); then the attention block's backward in one launch: dqkv and, through the
      // LayerNorm in front of this layer, da (for layer 0: through the embedding stage's dropout and LayerNorm)
      float* wo_scratch = wo_scratch_of_layer = take(b4r_gemm_tn_scratch_floats(N, H, H));
      if (side4) {
        RC(side_wait(s2, ev_dx));   // next to the feed-forward weight-gradient kernel, which leaves room on every CU
        RC(gemm_tn(ws + w.ctx[i], H, ws + w.db, H, grads + pl.wo[i], H, N, H, H, grads + pl.bo[i], nullptr, rng, B4R_STREAM_ATTN_OUT(i),
                   od, 1, wo_scratch, s2));
        RC(side_mark(s2, &ev_wo));
        RC(side_wait(s, ev_wqkv));   // dWqkv of the layer above reads dqkv, which this launch rewrites
      }
      b4r_attn_block_bwd_desc bd{};
      bd.B = B; bd.L = L; bd.H = H; bd.heads = cfg->num_heads;
      bd.x = x_in; bd.dz1 = ws + w.db; bd.ctx = ws + w.ctx[i]; bd.lse = ws + w.lse[i];
      bd.keep_bits = reinterpret_cast<const uint32_t*>(ws + w.keep[i]); bd.input_mask = batch->input_mask;
      bd.Wqkv = params + pl.wqkv[i]; bd.bqkv = params + pl.bqkv[i]; bd.Wo = params + pl.wo[i];
      bd.rng = (od > 0.f || adp > 0.f) ? rng : nullptr;
      bd.probs_stream = B4R_STREAM_ATTN_PROBS(i); bd.probs_rate = adp; bd.out_stream = B4R_STREAM_ATTN_OUT(i); bd.out_rate = od;
      if (i > 0) {
        bd.prev_z = ws + w.z2[i - 1]; bd.prev_mean = ws + w.mean2[i - 1]; bd.prev_rstd = ws + w.rstd2[i - 1];
        bd.prev_gamma = params + pl.ln2_g[i - 1]; bd.dprev_gamma = grads + pl.ln2_g[i - 1];
      } else {
        bd.prev_mean = ws + w.mean0; bd.prev_rstd = ws + w.rstd0; bd.prev_gamma = params + pl.emb_ln_g;
        bd.dprev_gamma = grads + pl.emb_ln_g;
        bd.emb_ids = batch->input_word_ids; bd.emb_table = params + pl.word_emb; bd.emb_pos = params + pl.pos_emb; bd.emb_vocab = V;
        bd.emb_stream = B4R_STREAM_EMB; bd.emb_rate = od;
      }
      bd.dqkv = ws + w.dqkv; bd.dx_prev = ws + w.da;
      bd.scratch = take(b4r_attn_block_bwd_scratch_floats(B));
      if (dw_folded) {   // dWqkv / dbqkv inside the launch: no [N, 3H] round trip, no weight-gradient launch for them
        bd.dqkv = nullptr; bd.dWqkv = grads + pl.wqkv[i]; bd.dbqkv = grads + pl.bqkv[i];
        bd.dw_scratch = take(b4r_attn_block_bwd_dw_scratch_floats(B));
        if (!side4) { bd.dWo = grads + pl.wo[i]; bd.dbo = grads + pl.bo[i]; }   // ... nor for dWo / dbo
      }
      if (sparse_dz1 && i == cfg->num_layers - 1) {
        bd.dz1_slot_positions = batch->masked_lm_positions; bd.dz1_slot_ids = batch->masked_lm_ids; bd.dz1_slots = batch->P;
      }
      RC(b4r_attn_block_bwd(&bd, stream));
    } else {
    if (slotq_layer(cfg, batch, flags, i)) {
      // compact: dz1 of the slots (left in dz2c by the feed-forward half above) -> dropout of the output projection -> dWo / dbo and
      // dctx on [M, H] rows -> the attention core's backward with the slots as its queries (dq rows of the labelled slots, dk / dv of
      // every token)
      float* dz1d = od > 0.f ? ws + w.dctx + up4((int64_t)M * H) : ws + w.dz2c;   // (the dense dctx region is free in this mode)
      float* dctx_c = ws + w.dctx;
      if (od > 0.f)
        RC(b4r_slot_rows_drop(ws + w.dz2c, batch->masked_lm_positions, L, P, M, H, b4r_make_drop(rng, B4R_STREAM_ATTN_OUT(i), od, 1), dz1d, s));
      RC(order_after(s, s_tn));
      RC(gemm_tn(ws + w.ctx[i], H, dz1d, H, grads + pl.wo[i], H, M, H, H, grads + pl.bo[i], nullptr, nullptr, 0, 0.f, 0,
                 take(b4r_gemm_tn_scratch_floats(N, H, H)), s_tn));
      RC(gemm(dz1d, H, params + pl.wo[i], H, dctx_c, H, M, H, H, 1, B4R_EPI_NONE, nullptr, nullptr, 0, nullptr, 0, 1.f, 0, nullptr, 0,
              0.f, 0, s));
      RC(b4r_attn32_slotq_bwd_launch(ws + w.qkv[i], batch->input_mask, batch->masked_lm_positions, batch->masked_lm_ids, ws + w.ctx[i],
                                     ws + w.lse[i], dctx_c, B, L, cfg->num_heads, P, qscale, ws + w.dqkv,
                                     b4r_make_drop(rng, B4R_STREAM_ATTN_PROBS(i), adp, 1),
                                     reinterpret_cast<const uint32_t*>(ws + w.keep[i]), s));
    } else {
    // attention output projection: dctx = dropmask(dz1) . Wo^T and dWo = ctx^T . dropmask(dz1) (+ bias gradient) read dz1
    // once where the pair kernel applies (hidden size 64), else as two products
    {
      b4r_gemm_tn_desc d{};
      d.A = ws + w.ctx[i]; d.lda = H; d.B = ws + w.db; d.ldb = H; d.out = grads + pl.wo[i]; d.ldo = H; d.R = N; d.Mo = H; d.No = H;
      d.colsum = grads + pl.bo[i]; d.rng = rng; d.drop_stream = B4R_STREAM_ATTN_OUT(i); d.drop_rate = od; d.b_dropout = 1;
      d.dgrad_w = params + pl.wo[i]; d.dgrad_ldw = H; d.dgrad_out = ws + w.dctx; d.dgrad_ldo = H;
      if ((pair_level() & 1) && b4r_gemm_tn_dgrad_supported(&d)) {
        RC(b4r_gemm_tn_f32(&d, take(b4r_gemm_tn_scratch_floats(N, H, H)), (b4r_stream_t)s));
      } else {
        RC(gemm(ws + w.db, H, params + pl.wo[i], H, ws + w.dctx, H, N, H, H, 1, B4R_EPI_NONE, nullptr, nullptr, 0, nullptr, 0, 1.f,
                0, rng, B4R_STREAM_ATTN_OUT(i), od, 1, s));
        RC(order_after(s, s_tn));
        RC(gemm_tn(ws + w.ctx[i], H, ws + w.db, H, grads + pl.wo[i], H, N, H, H, grads + pl.bo[i], nullptr, rng,
                   B4R_STREAM_ATTN_OUT(i), od, 1, take(b4r_gemm_tn_scratch_floats(N, H, H)), s_tn));
      }
    }
    // attention core: dQ on the main stream, dK/dV on the other (both need dctx; the QKV product below needs both)
    RC(order_after(s, s_kv));
    RC(b4r_attn_bwd_streams(ws + w.qkv[i], batch->input_mask, ws + w.ctx[i], ws + w.lse[i], ws + w.dctx, B, L, cfg->num_heads,
                            qscale, ws + w.dqkv, rng, B4R_STREAM_ATTN_PROBS(i), adp,
                            reinterpret_cast<const uint32_t*>(ws + w.keep[i]), s, s_kv));
    RC(order_after(s_kv, s));
    }
    // QKV projection: dX_in = dqkv . Wqkv^T + dz1, and for i > 0 straight on to layer i-1's output LayerNorm backward (-> da)
    if (i > 0) RC(order_after(s_tn, s));   // this layer's side branches still read da, which the fused product rewrites
    if (i > 0)
      RC(dgrad_ln_bwd(ws + w.dqkv, 3 * H, params + pl.wqkv[i], 3 * H, ws + w.db, ws + w.da, N, H, ws + w.z2[i - 1],
                      ws + w.mean2[i - 1], ws + w.rstd2[i - 1], params + pl.ln2_g[i - 1], grads + pl.ln2_g[i - 1],
                      grads + pl.ln2_b[i - 1], take(ln_scratch), s));
    else   // ... and for layer 0 on to the embedding stage: dropout -> LayerNorm of (item row + position row)
      RC(dgrad_ln_bwd(ws + w.dqkv, 3 * H, params + pl.wqkv[i], 3 * H, ws + w.db, ws + w.da, N, H, nullptr, ws + w.mean0, ws + w.rstd0,
                      params + pl.emb_ln_g, grads + pl.emb_ln_g, grads + pl.emb_ln_b, take(ln_scratch), s, batch->input_word_ids,
                      params + pl.word_emb, params + pl.pos_emb, L, V, rng, B4R_STREAM_EMB, od));
    }
    if (dw_folded) continue;   // every weight gradient of the attention half came out of its backward launch
    if (!side4 && attn_bwd_fused(cfg, L)) {   // dWo (inputs ready since the feed-forward backward) and dWqkv: one launch
      const b4r_gemm_tn_desc d_wo = tn_desc(ws + w.ctx[i], H, ws + w.db, H, grads + pl.wo[i], H, N, H, H, grads + pl.bo[i], rng,
                                            B4R_STREAM_ATTN_OUT(i), od, 1);
      const b4r_gemm_tn_desc d_wqkv = tn_desc(x_in, H, ws + w.dqkv, 3 * H, grads + pl.wqkv[i], 3 * H, N, H, 3 * H, grads + pl.bqkv[i],
                                              nullptr, 0, 0.f, 0);
      RC(b4r_gemm_tn_pair(&d_wo, wo_scratch_of_layer, &d_wqkv, take(b4r_gemm_tn_scratch_floats(N, H, 3 * H)), s));
      continue;
    }
    if (side4 && attn_bwd_fused(cfg, L)) {   // next to the feed-forward backward of the layer below
      RC(order_after(s, s2));
      RC(gemm_tn(x_in, H, ws + w.dqkv, 3 * H, grads + pl.wqkv[i], 3 * H, N, H, 3 * H, grads + pl.bqkv[i], nullptr, nullptr, 0, 0.f,
                 0, take(b4r_gemm_tn_scratch_floats(N, H, 3 * H)), s2));
      RC(side_mark(s2, &ev_wqkv));
      continue;
    }
    RC(order_after(s, s_tn));
    RC(gemm_tn(x_in, H, ws + w.dqkv, 3 * H, grads + pl.wqkv[i], 3 * H, N, H, 3 * H, grads + pl.bqkv[i], nullptr, nullptr, 0, 0.f,
               0, take(b4r_gemm_tn_scratch_floats(N, H, 3 * H)), s_tn));
  }
  // ---- embedding stage: its dropout -> LayerNorm backward ran with layer 0's QKV product (da = d(item row + position row));
  // what remains: word table scatter-add, position table batch sum
  // the item-table scatter sums in 64-bit fixed point beside the float gradient (bitwise reproducible; b4r_rowops.hip), so it
  // need not wait for the head's part of that gradient: ONE launch then sums every queued ordered reduction (weight / bias /
  // LayerNorm gradients, the position table) and adds the fixed-point sums to the item table
  RC(order_after(s2, s));
  RC(b4r_embed_grads(ws + w.da, batch->input_word_ids, B, L, H, grads + pl.word_emb, V, 3, ws + w.hot /* zeroed at the top */,
                     grads + pl.pos_emb, take((int64_t)b4r_cdiv(B, 16) * L * H), s, defer_combine ? ws + w.rowsc : nullptr, (int)w.M, state,
                     (defer_combine && (flags & B4R_FLAG_GRAD_TAIL)) ? grads + pl.total : nullptr));
  g_norm_np = 0;
  if (flags & B4R_FLAG_NORM_PARTIALS_INTERNAL) {
    // valid only when the jobs of this launch write EVERY gradient (then each value is squared exactly once, as it is stored)
    const int64_t Hh = H, Ii = I, Vv = V;
    const int64_t expected = Vv * Hh + (int64_t)L * Hh + 2 * Hh /* embedding LayerNorm */ +
                             (int64_t)cfg->num_layers * (Hh * 3 * Hh + Hh * Hh + 2 * Hh * Ii + 3 * Hh + Hh + 2 * Hh + Ii + Hh + 2 * Hh) +
                             Hh * Hh + Hh + 2 * Hh /* transform */ + Vv /* output bias */;
    int np = 0;
    int64_t covered = 0;
    RC(b4r_reduce_queue_flush(s, ws, 4096, &np, &covered));
    if (np > 0 && covered == expected) g_norm_np = np;
  } else {
    RC(b4r_reduce_queue_flush(s));
  }
  B4R_CHECK_ARG(scratch_used <= w.scratch_floats, B4R_E_NOMEM, "b4r_backward: internal scratch overflow");
  return B4R_OK;
}

extern "C" int b4r_optimizer_step(const b4r_model_config* cfg, const b4r_adamw_config* hp, float* params, const float* grads,
                                  float* adam_m, float* adam_v, void* workspace, int64_t workspace_bytes,
                                  b4r_train_state* state, b4r_stream_t stream) {
  RC(check_cfg(cfg));
  B4R_CHECK_ARG(hp && params && grads && adam_m && adam_v && workspace && state, B4R_E_BADARG, "b4r_optimizer_step: null argument");
  B4R_CHECK_ARG(workspace_bytes >= 4096 * (int64_t)sizeof(float), B4R_E_NOMEM, "b4r_optimizer_step: workspace too small");
  const ParamLayout pl = make_param_layout(*cfg);
  // the scratch region sits at the END of the workspace; any 4096-float area works, use the start of the buffer the
  // backward no longer needs: the first floats of the workspace hold x0 which is dead after backward.
  float* scratch = static_cast<float*>(workspace);
  return b4r_optimizer_fused(hp, params, grads, adam_m, adam_v, pl.total, pl.n_decay, scratch, state, (hipStream_t)stream);
}

extern "C" int b4r_optimizer_step_reduced(const b4r_model_config* cfg, const b4r_adamw_config* hp, float* params, const float* grads,
                                          float* adam_m, float* adam_v, void* workspace, int64_t workspace_bytes,
                                          b4r_train_state* state, b4r_stream_t stream) {
  RC(check_cfg(cfg));
  B4R_CHECK_ARG(hp && params && grads && adam_m && adam_v && workspace && state, B4R_E_BADARG, "b4r_optimizer_step_reduced: null argument");
  B4R_CHECK_ARG(workspace_bytes >= 4096 * (int64_t)sizeof(float), B4R_E_NOMEM, "b4r_optimizer_step_reduced: workspace too small");
  const ParamLayout pl = make_param_layout(*cfg);
  return b4r_optimizer_fused(hp, params, grads, adam_m, adam_v, pl.total, pl.n_decay, static_cast<float*>(workspace), state,
                             (hipStream_t)stream, 1);
}

extern "C" int b4r_train_step(const b4r_model_config* cfg, const b4r_adamw_config* hp, const b4r_batch* batch, float* params,
                              float* grads, float* adam_m, float* adam_v, void* workspace, int64_t workspace_bytes,
                              b4r_train_state* state, b4r_stream_t stream) {
  const int fused = b4r_fused_head_supported(cfg) ? 1 : 0;   // the train step never needs the logits themselves
  const int defer = (fused && batch && b4r_head_rx_combine_foldable(batch->B * batch->P, cfg->vocab_size, cfg->hidden_size))
                        ? B4R_FLAG_DEFER_COMBINE_INTERNAL : 0;
  // no b4r_state_begin_step launch: the loss reduction overwrites the sums (B4R_LOSS_OVERWRITE)
  // nothing but the loss, the metrics and the gradients leave a train step: the last layer's feed-forward half runs on the rows the
  // head gathers only (B4R_FLAG_HEAD_ROWS_ONLY; the same flag goes to forward and backward)
  RC(forward_impl(cfg, batch, params, nullptr, workspace, workspace_bytes, state,
                  B4R_FLAG_TRAINING | B4R_FLAG_HEAD_ROWS_ONLY | (fused ? B4R_FLAG_FUSED_HEAD : 0) | defer, stream));
  // with the logits-free head the loss sums are formed inside the backward's first launch (B4R_FLAG_LOSS_SUMS), else by b4r_loss
  if (!fused) RC(b4r_loss(cfg, batch, workspace, workspace_bytes, state, 1 | B4R_LOSS_OVERWRITE, stream));
  RC(backward_impl(cfg, batch, params, grads, workspace, workspace_bytes, state,
                   B4R_FLAG_TRAINING | B4R_FLAG_HEAD_ROWS_ONLY |
                      (fused ? B4R_FLAG_FUSED_HEAD | B4R_FLAG_LOSS_SUMS : 0) | defer |
                      B4R_FLAG_NORM_PARTIALS_INTERNAL, stream));
  const int np = g_norm_np;   // > 0: the backward's last launch left the norm's partial sums at the start of the workspace
  g_norm_np = 0;
  if (np > 0) {
    B4R_CHECK_ARG(hp && adam_m && adam_v, B4R_E_BADARG, "b4r_train_step: null argument");
    const ParamLayout pl = make_param_layout(*cfg);
    return b4r_optimizer_fused(hp, params, grads, adam_m, adam_v, pl.total, pl.n_decay, static_cast<float*>(workspace), state,
                               (hipStream_t)stream, 0, np);
  }
  RC(b4r_optimizer_step(cfg, hp, params, grads, adam_m, adam_v, workspace, workspace_bytes, state, stream));
  return B4R_OK;
}
