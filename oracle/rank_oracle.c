/* CPU restatement of the candidate-ranking step  --  TEST INFRASTRUCTURE ONLY (see oracle/bert4rec_oracle.py).
 *
 * Follows  bert4rec/models/bert4rec_model.py:224-239  (gather logits of the candidates, tf.argsort DESCENDING,
 * gather the candidates in that order)  and  bert4rec/evaluation/bert4rec_evaluator.py:113-120  (rank = 1 + index of
 * the ground truth), bert4rec/evaluation/evaluation_metrics.py:55-96 (HR / NDCG / MAP accumulation).
 *
 * The score arithmetic is the one the HIP rank kernel is specified to use (DESIGN.md "rank kernel"):
 *     acc = 0;  for k in 0..H-1: acc = fmaf(hidden[k], E[c][k], acc);  score = acc + bias[c]
 * so that scores, and therefore ranked indices, are bit-identical between this file and the GPU.
 * tf.argsort(DESCENDING) is top_k(k=n): ties keep the lower index first  => stable sort on descending score.
 * Build: see oracle/Makefile (gcc -O2 -ffp-contract=off; fmaf is called explicitly).
 */
#include <math.h>
#include <stdint.h>
#include <string.h>

void rank_oracle_scores(const float* hidden, const float* table, const float* bias, const int64_t* cand,
                        int64_t R, int64_t C, int64_t H, float* scores) {
  for (int64_t r = 0; r < R; ++r)
    for (int64_t j = 0; j < C; ++j) {
      const int64_t c = cand[r * C + j];
      const float* e = table + c * H;
      const float* h = hidden + r * H;
      float acc = 0.0f;
      for (int64_t k = 0; k < H; ++k) acc = fmaf(h[k], e[k], acc);
      scores[r * C + j] = acc + bias[c];
    }
}

/* position of candidate j in the descending stable order = #{i: s_i > s_j} + #{i < j: s_i == s_j} */
void rank_oracle_rank(const float* scores, const int64_t* cand, int64_t R, int64_t C, int64_t* ranking, int32_t* pos) {
  for (int64_t r = 0; r < R; ++r) {
    const float* s = scores + r * C;
    for (int64_t j = 0; j < C; ++j) {
      int32_t p = 0;
      for (int64_t i = 0; i < C; ++i) p += (s[i] > s[j]) || (s[i] == s[j] && i < j);
      pos[r * C + j] = p;
      ranking[r * C + p] = cand[r * C + j];
    }
  }
}

/* out[0]=count, out[1..3]=NDCG@1/5/10 sums, out[4..6]=HR@1/5/10 sums, out[7]=sum 1/rank   (doubles, like python) */
void rank_oracle_metrics(const int64_t* ranks, int64_t n, double* out) {
  static const int ks[3] = {1, 5, 10};
  memset(out, 0, 8 * sizeof(double));
  for (int64_t i = 0; i < n; ++i) {
    const int64_t r = ranks[i];
    out[0] += 1.0;
    for (int t = 0; t < 3; ++t)
      if (r <= ks[t]) {
        out[4 + t] += 1.0;
        out[1 + t] += (r == 1) ? 1.0 : 1.0 / log2((double)r + 1.0);
      }
    out[7] += 1.0 / (double)r;
  }
}
