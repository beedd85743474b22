"""CPU restatement of the maneymarkus/BERT4Rec hot path  --  TEST INFRASTRUCTURE ONLY.

Only ``tests/``, ``__graft_entry__.smoke()`` and ``bench.py``'s ``cpu_baseline`` leg may import
this module, and only as the checker.  The product (``bert4rec_amd``) never imports it and has no
CPU fallback: it fails loudly when the HIP library is missing.

PARITY STATUS: **parity unpinned** for the floating-point path (logits / loss / gradients /
optimizer update).  The reference's arithmetic lives in third-party packages that are neither under
/root/reference nor installed here (tensorflow==2.10.0, keras==2.10.0, tf-models-official==2.10.1;
Pipfile.lock), and the reference's own tests hold no numerical fixture for it.  This
file restates the published algorithms of those packages, anchored on the reference's call sites:

  embedding stage      bert4rec/models/components/networks/bert4rec_encoder.py:198-214
  attention mask       bert4rec_encoder.py:134-135,216   (tfm SelfAttentionMask: mask[b,i,j]=input_mask[b,j])
  transformer block    bert4rec_encoder.py:136-147,220-222 (tfm TransformerEncoderBlock, post-LN)
  pooler               bert4rec_encoder.py:149-153,224-226
  masked-LM head       bert4rec/models/bert4rec_model.py:76-81,143 (tfm MaskedLM, tied table)
  loss                 bert4rec/trainers/trainer_utils.py:12-23
  metrics              bert4rec/trainers/trainer_utils.py:49-60, bert4rec_trainer.py:28-33
  optimizer            bert4rec/trainers/optimizers/adam_w_optimizer.py:22-36,91-137,154-168
                       bert4rec/trainers/optimizers/__init__.py:7-56
  rank_items           bert4rec/models/bert4rec_model.py:203-240
  eval metrics         bert4rec/evaluation/evaluation_metrics.py:47-112   (PINNED by the reference's
                       known-answer tests tests/evaluators_tests/evaluation_metrics_tests.py:28-104)
  batch contract       bert4rec/dataloaders/preprocessors/bert4rec_preprocessor.py:48-116,
                       bert4rec/dataloaders/dataloader_utils.py:186-269 (PINNED by golden vectors
                       captured from the reference's numpy code, tests/golden/)

Third-party semantics that are assumptions of this restatement (each has its own unit test in
tests/test_oracle.py):  (i) post-LN order and dropout placement, (ii) query scaled by 1/sqrt(d)
after the bias and before QK^T, (iii) additive mask (1-mask)*-1e9 in fp32, (iv) Keras non-fused
LayerNormalization: biased variance, y = x*inv + (beta - mean*inv), inv = rsqrt(var+eps)*gamma,
(v) "gelu" = exact erf form, (vi) MLM head = gather -> dense(gelu) -> LN -> tied E^T + bias,
(vii) tf.argsort(DESCENDING) is stable (ties: lower index first), (viii) Keras-2.10 optimizer_v2 Adam
(ResourceApplyAdam): alpha = lr*sqrt(1-b2^t)/(1-b1^t), m += (g-m)(1-b1), v += (g^2-v)(1-b2),
var -= m*alpha/(sqrt(v)+eps), t = iterations+1, schedule evaluated at 0-based iterations,
(ix) clip_by_global_norm scale = clip/max(norm, clip), (x) MHA kernels are [H,h,d] / [h,d,H].

Float math is torch-CPU fp32 with explicit formulas; gradients come from torch.autograd, which makes
them an independent check of the hand-derived HIP backward kernels.
"""
from __future__ import annotations

import math
import random
import re
from dataclasses import dataclass, field
from typing import Dict, List, Optional, Sequence, Tuple

import numpy as np
import torch

PAD_TOKEN_ID, MASK_TOKEN_ID, UNK_TOKEN_ID = 0, 1, 2  # bert4rec_dataloader.py:38-43 (tokenizer insertion order)


# ----------------------------------------------------------------------------------------------
# configuration
# ----------------------------------------------------------------------------------------------
@dataclass
class OracleConfig:
    """Mirrors the kwargs of Bert4RecEncoder (bert4rec_encoder.py:62-80) that the shipped JSON configs set."""
    vocab_size: int
    hidden_size: int = 64
    num_layers: int = 2
    num_attention_heads: int = 2
    max_sequence_length: int = 200
    inner_dim: int = 256
    output_dropout: float = 0.0
    attention_dropout: float = 0.0
    ln_eps: float = 1e-12

    @property
    def head_dim(self) -> int:
        return self.hidden_size // self.num_attention_heads


# ----------------------------------------------------------------------------------------------
# parameters (named exactly like the reference's Keras variables)
# ----------------------------------------------------------------------------------------------
def param_names_and_shapes(cfg: OracleConfig) -> List[Tuple[str, Tuple[int, ...]]]:
    H, h, d, I, V = cfg.hidden_size, cfg.num_attention_heads, cfg.head_dim, cfg.inner_dim, cfg.vocab_size
    out: List[Tuple[str, Tuple[int, ...]]] = [
        ("word_embeddings/embeddings", (V, H)),
        ("position_embedding/embeddings", (cfg.max_sequence_length, H)),
        ("embeddings/layer_norm/gamma", (H,)),
        ("embeddings/layer_norm/beta", (H,)),
    ]
    for i in range(cfg.num_layers):
        p = f"transformer/layer_{i}"
        out += [
            (f"{p}/self_attention/query/kernel", (H, h, d)), (f"{p}/self_attention/query/bias", (h, d)),
            (f"{p}/self_attention/key/kernel", (H, h, d)), (f"{p}/self_attention/key/bias", (h, d)),
            (f"{p}/self_attention/value/kernel", (H, h, d)), (f"{p}/self_attention/value/bias", (h, d)),
            (f"{p}/self_attention/attention_output/kernel", (h, d, H)),
            (f"{p}/self_attention/attention_output/bias", (H,)),
            (f"{p}/self_attention_layer_norm/gamma", (H,)), (f"{p}/self_attention_layer_norm/beta", (H,)),
            (f"{p}/intermediate/kernel", (H, I)), (f"{p}/intermediate/bias", (I,)),
            (f"{p}/output/kernel", (I, H)), (f"{p}/output/bias", (H,)),
            (f"{p}/output_layer_norm/gamma", (H,)), (f"{p}/output_layer_norm/beta", (H,)),
        ]
    out += [
        ("pooler_transform/kernel", (H, H)), ("pooler_transform/bias", (H,)),
        ("cls/predictions/transform/dense/kernel", (H, H)), ("cls/predictions/transform/dense/bias", (H,)),
        ("cls/predictions/transform/LayerNorm/gamma", (H,)), ("cls/predictions/transform/LayerNorm/beta", (H,)),
        ("cls/predictions/output_bias/bias", (V,)),
    ]
    return out


def is_trainable(name: str) -> bool:
    """The pooler is not on the loss path => its gradient is None => Keras skips it."""
    return not name.startswith("pooler_transform/")


def uses_weight_decay(name: str, exclude: Sequence[str] = ("LayerNorm", "layer_norm", "bias")) -> bool:
    """adam_w_optimizer.py:154-168 with the default exclusion list of optimizers/__init__.py:35-36."""
    for r in exclude:
        if re.search(r, name) is not None:
            return False
    return True


def init_params(cfg: OracleConfig, seed: int = 3) -> Dict[str, torch.Tensor]:
    """TruncatedNormal(0.02) for encoder weights (bert4rec_encoder.py:73-74), glorot_uniform for the MLM dense
    (bert4rec_model.py:42), LN gamma=1 beta=0, biases 0.  Same distribution family as the reference; the
    values are of course not TF's RNG stream."""
    g = torch.Generator().manual_seed(seed)
    params: Dict[str, torch.Tensor] = {}
    for name, shape in param_names_and_shapes(cfg):
        if name.endswith("gamma"):
            t = torch.ones(shape)
        elif name.endswith("beta") or name.endswith("bias"):
            t = torch.zeros(shape)
        elif name == "cls/predictions/transform/dense/kernel":
            lim = math.sqrt(6.0 / (shape[0] + shape[1]))
            t = (torch.rand(shape, generator=g) * 2 - 1) * lim
        else:
            t = torch.empty(shape)
            torch.nn.init.trunc_normal_(t, mean=0.0, std=0.02, a=-0.04, b=0.04, generator=g)
        params[name] = t.to(torch.float32)
    return params


# ----------------------------------------------------------------------------------------------
# counter-hash dropout (the build's own RNG; TF's stream is not reproducible, parity runs use rate 0;
# this restatement of the hash lets train-mode kernels be checked bit-for-bit on the mask)
# ----------------------------------------------------------------------------------------------
_M32 = 0xFFFFFFFF


def _hash32(x: torch.Tensor) -> torch.Tensor:
    x = x & _M32
    x = x ^ (x >> 16)
    x = (x * 0x7FEB352D) & _M32
    x = x ^ (x >> 15)
    x = (x * 0x846CA68B) & _M32
    x = x ^ (x >> 16)
    return x


def _hash32_int(x: int) -> int:
    x &= _M32
    x ^= x >> 16
    x = (x * 0x7FEB352D) & _M32
    x ^= x >> 15
    x = (x * 0x846CA68B) & _M32
    x ^= x >> 16
    return x


ATTN_PITCH = 256   # B4R_ATTN_PITCH: attention-probability rows are indexed with this pitch so that they start a hash group


def dropout_keep_mask(numel_shape: Sequence[int], rate: float, seed: int, step: int, stream: int,
                      row_pitch: Optional[int] = None) -> torch.Tensor:
    """keep[idx] (DESIGN.md 'dropout').  idx is the flat row-major index of a tensor of ``numel_shape`` or, with
    ``row_pitch``, (flat index of the leading dims) * row_pitch + last index (attention probabilities: ATTN_PITCH).
    One 2-round hash per group of 4 consecutive indices; it and one xorshift32 step of it hold four 16-bit uniforms."""
    shape = tuple(int(s) for s in numel_shape)
    n = int(np.prod(shape))
    flat = torch.arange(n, dtype=torch.int64)
    if row_pitch is None:
        idx = flat
    else:
        last = shape[-1]
        assert last <= row_pitch
        idx = (flat // last) * row_pitch + (flat % last)
    grp, e = idx >> 2, idx & 3
    lo, hi = grp & _M32, grp >> 32
    key = _hash32_int((stream * 0x9E3779B9 + step) & _M32)
    h1 = _hash32(lo ^ (seed & _M32))
    h1 = _hash32(((h1 ^ hi) + key) & _M32)
    h2 = h1 ^ ((h1 << 13) & _M32)
    h2 = h2 ^ (h2 >> 17)
    h2 = h2 ^ ((h2 << 5) & _M32)
    w = torch.where(e >= 2, h2, h1)
    u = torch.where((e & 1) == 1, w >> 16, w & 0xFFFF)
    thr = int(rate * 65536.0)
    return (u >= thr).reshape(shape)


def _dropout(x: torch.Tensor, rate: float, training: bool, rng: Optional[Tuple[int, int]], stream: int,
             row_pitch: Optional[int] = None) -> torch.Tensor:
    if not training or rate <= 0.0:
        return x
    assert rng is not None, "training with dropout needs rng=(seed, step)"
    keep = dropout_keep_mask(x.shape, rate, rng[0], rng[1], stream, row_pitch)
    scale = torch.tensor(1.0 / (1.0 - rate), dtype=torch.float32)
    return torch.where(keep, x * scale, torch.zeros((), dtype=x.dtype))


STREAM_EMB = 0


def stream_attn_probs(layer: int) -> int:
    return 1 + 4 * layer


def stream_attn_out(layer: int) -> int:
    return 2 + 4 * layer


def stream_ffn_out(layer: int) -> int:
    return 3 + 4 * layer


# ----------------------------------------------------------------------------------------------
# float path
# ----------------------------------------------------------------------------------------------
def layer_norm(x: torch.Tensor, gamma: torch.Tensor, beta: torch.Tensor, eps: float = 1e-12) -> torch.Tensor:
    """Keras LayerNormalization, non-fused branch (epsilon 1e-12 < 1.001e-5): tf.nn.moments + batch_normalization."""
    mean = x.mean(dim=-1, keepdim=True)
    var = ((x - mean) ** 2).mean(dim=-1, keepdim=True)
    inv = torch.rsqrt(var + eps) * gamma
    return x * inv + (beta - mean * inv)


def gelu_erf(x: torch.Tensor) -> torch.Tensor:
    """tf.keras.activations.gelu(approximate=False)."""
    return 0.5 * x * (1.0 + torch.erf(x / 1.4142135623730951))


def encoder_forward(params: Dict[str, torch.Tensor], input_word_ids: torch.Tensor, input_mask: torch.Tensor,
                    cfg: OracleConfig, training: bool = False, rng: Optional[Tuple[int, int]] = None):
    """Bert4RecEncoder.call (bert4rec_encoder.py:186-231)."""
    B, L = input_word_ids.shape
    H, h, d = cfg.hidden_size, cfg.num_attention_heads, cfg.head_dim
    E = params["word_embeddings/embeddings"]
    x = E[input_word_ids]                                            # OnDeviceEmbedding: tf.gather
    x = x + params["position_embedding/embeddings"][:L].unsqueeze(0)  # PositionEmbedding: slice + broadcast
    x = layer_norm(x, params["embeddings/layer_norm/gamma"], params["embeddings/layer_norm/beta"], cfg.ln_eps)
    x = _dropout(x, cfg.output_dropout, training, rng, STREAM_EMB)
    # SelfAttentionMask: mask[b,i,j] = input_mask[b,j]; Keras Softmax adds (1-mask)*-1e9
    adder = (1.0 - input_mask.to(torch.float32))[:, None, None, :] * torch.tensor(-1e9, dtype=torch.float32)
    encoder_outputs = []
    for i in range(cfg.num_layers):
        p = f"transformer/layer_{i}"
        q = torch.einsum("blH,Hhd->blhd", x, params[f"{p}/self_attention/query/kernel"]) + params[f"{p}/self_attention/query/bias"]
        k = torch.einsum("blH,Hhd->blhd", x, params[f"{p}/self_attention/key/kernel"]) + params[f"{p}/self_attention/key/bias"]
        v = torch.einsum("blH,Hhd->blhd", x, params[f"{p}/self_attention/value/kernel"]) + params[f"{p}/self_attention/value/bias"]
        q = q * torch.tensor(1.0 / math.sqrt(float(d)), dtype=torch.float32)
        s = torch.einsum("bqhd,bkhd->bhqk", q, k) + adder
        a = torch.softmax(s, dim=-1)
        a = _dropout(a, cfg.attention_dropout, training, rng, stream_attn_probs(i), ATTN_PITCH)
        ctx = torch.einsum("bhqk,bkhd->bqhd", a, v)
        y = torch.einsum("bqhd,hdH->bqH", ctx, params[f"{p}/self_attention/attention_output/kernel"]) \
            + params[f"{p}/self_attention/attention_output/bias"]
        y = _dropout(y, cfg.output_dropout, training, rng, stream_attn_out(i))
        x1 = layer_norm(x + y, params[f"{p}/self_attention_layer_norm/gamma"],
                        params[f"{p}/self_attention_layer_norm/beta"], cfg.ln_eps)
        f = gelu_erf(x1 @ params[f"{p}/intermediate/kernel"] + params[f"{p}/intermediate/bias"])
        g = f @ params[f"{p}/output/kernel"] + params[f"{p}/output/bias"]
        g = _dropout(g, cfg.output_dropout, training, rng, stream_ffn_out(i))
        x = layer_norm(g + x1, params[f"{p}/output_layer_norm/gamma"], params[f"{p}/output_layer_norm/beta"], cfg.ln_eps)
        encoder_outputs.append(x)
    pooled = torch.tanh(x[:, 0, :] @ params["pooler_transform/kernel"] + params["pooler_transform/bias"])
    return dict(sequence_output=x, pooled_output=pooled, encoder_outputs=encoder_outputs)


def mlm_transform(params: Dict[str, torch.Tensor], sequence_output: torch.Tensor,
                  masked_lm_positions: torch.Tensor, cfg: OracleConfig) -> torch.Tensor:
    """tfm MaskedLM up to (not including) the vocabulary projection: gather -> dense(gelu) -> LN.  [B,P,H]"""
    B, L, H = sequence_output.shape
    flat = sequence_output.reshape(B * L, H)
    offs = (torch.arange(B, dtype=torch.int64) * L)[:, None]
    g = flat[(masked_lm_positions.to(torch.int64) + offs).reshape(-1)]
    t = gelu_erf(g @ params["cls/predictions/transform/dense/kernel"] + params["cls/predictions/transform/dense/bias"])
    t = layer_norm(t, params["cls/predictions/transform/LayerNorm/gamma"],
                   params["cls/predictions/transform/LayerNorm/beta"], cfg.ln_eps)
    return t.reshape(B, -1, H)


def model_forward(params: Dict[str, torch.Tensor], batch: Dict[str, torch.Tensor], cfg: OracleConfig,
                  training: bool = False, rng: Optional[Tuple[int, int]] = None) -> Dict[str, torch.Tensor]:
    """BERT4RecModel.call (bert4rec_model.py:110-149); prediction_mask is disabled in the reference (:101-102)."""
    out = encoder_forward(params, batch["input_word_ids"], batch["input_mask"], cfg, training, rng)
    if "masked_lm_positions" in batch:
        t = mlm_transform(params, out["sequence_output"], batch["masked_lm_positions"], cfg)
        out["mlm_hidden"] = t  # not a reference output; exposed for the rank-kernel boundary test
        out["mlm_logits"] = t @ params["word_embeddings/embeddings"].t() + params["cls/predictions/output_bias/bias"]
    return out


def masked_sparse_categorical_crossentropy(y_true: torch.Tensor, logits: torch.Tensor, pad_token: int = 0) -> torch.Tensor:
    """trainer_utils.py:12-23: sum(l*mask)/sum(mask), batch-global."""
    mask = (y_true != pad_token)
    lse = torch.logsumexp(logits, dim=-1)
    picked = torch.gather(logits, -1, y_true.to(torch.int64).unsqueeze(-1)).squeeze(-1)
    loss = (lse - picked)
    maskf = mask.to(loss.dtype)
    return (loss * maskf).sum() / maskf.sum()


def masked_accuracy(y_true: torch.Tensor, logits: torch.Tensor) -> torch.Tensor:
    """trainer_utils.py:49-60."""
    pred = torch.argmax(logits, dim=2)
    match = (y_true == pred) & (y_true != 0)
    return match.to(torch.float32).sum() / (y_true != 0).to(torch.float32).sum()


def sparse_categorical_accuracy(y_true: torch.Tensor, logits: torch.Tensor) -> torch.Tensor:
    """tf.keras.metrics.SparseCategoricalAccuracy for one batch: unmasked mean over all B*P slots."""
    pred = torch.argmax(logits, dim=2)
    return (y_true == pred).to(torch.float32).mean()


# ----------------------------------------------------------------------------------------------
# optimizer
# ----------------------------------------------------------------------------------------------
@dataclass
class AdamWConfig:
    """optimizers/__init__.py:7-15 and adam_w_optimizer.py:67."""
    init_lr: float = 1e-4
    num_train_steps: int = 400000
    num_warmup_steps: int = 100
    end_lr: float = 0.0
    weight_decay_rate: float = 0.01
    beta_1: float = 0.9
    beta_2: float = 0.999
    epsilon: float = 1e-6
    gradient_clip_norm: float = 5.0
    exclude_from_weight_decay: Tuple[str, ...] = ("LayerNorm", "layer_norm", "bias")


def learning_rate(step: int, hp: AdamWConfig) -> np.float32:
    """WarmUp.__call__ (adam_w_optimizer.py:22-36) over PolynomialDecay(power=1), all in float32; ``step`` is the
    0-based optimizer.iterations; the decay branch is fed the raw step (warmup not subtracted)."""
    f32 = np.float32
    s = f32(step)
    if hp.num_warmup_steps and s < f32(hp.num_warmup_steps):
        return f32(hp.init_lr) * f32(s / f32(hp.num_warmup_steps))  # power 1.0
    gs = min(s, f32(hp.num_train_steps))
    p = f32(gs / f32(hp.num_train_steps))
    return f32(f32(f32(hp.init_lr) - f32(hp.end_lr)) * f32(f32(1.0) - p) + f32(hp.end_lr))


def adamw_apply(params: Dict[str, torch.Tensor], grads: Dict[str, torch.Tensor], m: Dict[str, torch.Tensor],
                v: Dict[str, torch.Tensor], step: int, hp: AdamWConfig) -> float:
    """AdamWeightDecay.apply_gradients (adam_w_optimizer.py:100-137): clip -> decay -> Adam.  In place.
    Returns the global gradient norm (before clipping)."""
    names = [n for n in params if n in grads and grads[n] is not None]
    f32 = torch.float32
    gnorm = torch.sqrt(sum((grads[n].to(f32) ** 2).sum() for n in names))
    if hp.gradient_clip_norm > 0.0:
        clip = torch.tensor(hp.gradient_clip_norm, dtype=f32)
        scale = clip / torch.maximum(gnorm, clip)   # tf.clip_by_global_norm
    else:
        scale = torch.tensor(1.0, dtype=f32)
    lr_t = torch.tensor(float(learning_rate(step, hp)), dtype=f32)
    t = float(step + 1)
    b1, b2 = torch.tensor(hp.beta_1, dtype=f32), torch.tensor(hp.beta_2, dtype=f32)
    b1p, b2p = torch.pow(b1, torch.tensor(t, dtype=f32)), torch.pow(b2, torch.tensor(t, dtype=f32))
    alpha = lr_t * torch.sqrt(1.0 - b2p) / (1.0 - b1p)
    eps = torch.tensor(hp.epsilon, dtype=f32)
    wd = torch.tensor(hp.weight_decay_rate, dtype=f32)
    for n in names:
        g = grads[n].to(f32) * scale
        if hp.weight_decay_rate != 0 and uses_weight_decay(n, hp.exclude_from_weight_decay):
            params[n].sub_(lr_t * params[n] * wd)
        m[n].add_((g - m[n]) * (1.0 - b1))
        v[n].add_((g * g - v[n]) * (1.0 - b2))
        params[n].sub_((m[n] * alpha) / (torch.sqrt(v[n]) + eps))
    return float(gnorm)


def loss_and_grads(params: Dict[str, torch.Tensor], batch: Dict[str, torch.Tensor], cfg: OracleConfig,
                   training: bool = True, rng: Optional[Tuple[int, int]] = None):
    """forward + loss + autograd gradients wrt every trainable variable (train_step lines :158-167)."""
    leaf = {n: p.detach().clone().requires_grad_(is_trainable(n)) for n, p in params.items()}
    out = model_forward(leaf, batch, cfg, training=training, rng=rng)
    loss = masked_sparse_categorical_crossentropy(batch["masked_lm_ids"], out["mlm_logits"])
    names = [n for n in leaf if is_trainable(n)]
    gs = torch.autograd.grad(loss, [leaf[n] for n in names], allow_unused=True)
    grads = {n: (g if g is not None else torch.zeros_like(leaf[n])) for n, g in zip(names, gs)}
    out = {k: (v.detach() if torch.is_tensor(v) else [t.detach() for t in v]) for k, v in out.items()}
    return loss.detach(), grads, out


def train_step(params, m, v, batch, cfg: OracleConfig, hp: AdamWConfig, step: int,
               training: bool = True, rng: Optional[Tuple[int, int]] = None) -> Dict[str, float]:
    """BERT4RecModel.train_step (bert4rec_model.py:151-173).  Mutates params/m/v; returns the batch metrics."""
    loss, grads, out = loss_and_grads(params, batch, cfg, training, rng)
    gnorm = adamw_apply(params, grads, m, v, step, hp)
    y = batch["masked_lm_ids"]
    return dict(loss=float(loss), masked_accuracy=float(masked_accuracy(y, out["mlm_logits"])),
                sparse_categorical_accuracy=float(sparse_categorical_accuracy(y, out["mlm_logits"])),
                grad_norm=gnorm)


def zeros_like_params(params: Dict[str, torch.Tensor]) -> Dict[str, torch.Tensor]:
    return {n: torch.zeros_like(p) for n, p in params.items() if is_trainable(n)}


# ----------------------------------------------------------------------------------------------
# ranking + evaluation metrics
# ----------------------------------------------------------------------------------------------
def candidate_scores_fma(hidden: np.ndarray, table: np.ndarray, bias: np.ndarray, cand: np.ndarray) -> np.ndarray:
    """Score of each candidate as the k-ordered float32 fma chain the rank kernel is specified to use
    (DESIGN.md 'rank kernel'): acc=0; acc = fma(hidden[k], E[c][k], acc) for k=0..H-1; score = acc + bias[c].
    numpy has no fma: emulate in float64 (an fp32*fp32 product is exact in fp64, and one fp64 add followed by a
    single rounding to fp32 equals the fused result except in vanishingly rare double-rounding cases; the C
    oracle oracle/rank_oracle.c uses fmaf proper and is what the GPU test compares against)."""
    R, C = cand.shape
    out = np.zeros((R, C), dtype=np.float32)
    h64, e64 = hidden.astype(np.float64), table.astype(np.float64)
    for r in range(R):
        acc = np.zeros(C, dtype=np.float32)
        rows = e64[cand[r]]
        for k in range(hidden.shape[1]):
            acc = (h64[r, k] * rows[:, k] + acc.astype(np.float64)).astype(np.float32)
        out[r] = acc + bias[cand[r]].astype(np.float32)
    return out


def rank_candidates(scores: np.ndarray, cand: np.ndarray) -> Tuple[np.ndarray, np.ndarray]:
    """bert4rec_model.py:232-234: order = argsort(DESCENDING) (stable: ties keep the lower index first);
    ranking = cand[order].  Returns (ranking [R,C], position of each candidate [R,C])."""
    order = np.argsort(-scores.astype(np.float64), axis=1, kind="stable")
    # NB negating is exact in fp; -0.0 vs +0.0 compare equal in both orderings
    ranking = np.take_along_axis(cand, order, axis=1)
    pos = np.empty_like(order)
    np.put_along_axis(pos, order, np.arange(scores.shape[1])[None, :].repeat(scores.shape[0], 0), axis=1)
    return ranking, pos


def rank_of_ground_truth(ranking: np.ndarray, gt: np.ndarray) -> np.ndarray:
    """bert4rec_evaluator.py:114-117: 1 + first index where ranking == gt."""
    return np.array([1 + int(np.where(ranking[r] == gt[r])[0][0]) for r in range(ranking.shape[0])], dtype=np.int64)


class EvalMetrics:
    """evaluation_metrics.py:47-112 with the default metric set of bert4rec_evaluator.py:12-21."""
    KS = (1, 5, 10)

    def __init__(self):
        self.reset()

    def reset(self):
        self.count = 0
        self.hr = {k: 0.0 for k in self.KS}
        self.ndcg = {k: 0.0 for k in self.KS}
        self.ap = 0.0

    def update(self, rank: int):
        self.count += 1
        for k in self.KS:
            if rank <= k:
                self.hr[k] += 1
                self.ndcg[k] += 1 if rank == 1 else 1 / np.log2(rank + 1)
        self.ap += 1 / rank

    def results(self) -> Dict[str, float]:
        n = float(self.count)
        res = {"Valid Ranks": self.count}
        for k in self.KS:
            res[f"NDCG@{k}"] = self.ndcg[k] / n
        for k in self.KS:
            res[f"HR@{k}"] = self.hr[k] / n
        res["MAP"] = self.ap / n
        return res


# ----------------------------------------------------------------------------------------------
# batch contract (integer work; numpy / python, follows the reference line by line in behaviour)
# ----------------------------------------------------------------------------------------------
def apply_dynamic_masking_task(sequence: np.ndarray, max_selections_per_seq: int, mask_token_id: int,
                               special_token_ids: Sequence[int], vocab_size: int, selection_rate: float = 0.2,
                               mask_token_rate: float = 0.8, random_token_rate: float = 0.1, seed=None):
    """dataloader_utils.py:186-261.  Uses python's ``random`` exactly like the reference so that equal seeds give
    equal outputs (pinned by tests/golden/reference_goldens.json)."""
    random.seed(seed)
    keep = ~np.isin(sequence, special_token_ids)
    n_nonspecial = int(keep.sum())
    num_to_predict = min(max_selections_per_seq, max(1, int(n_nonspecial * selection_rate)))
    selectable_vocab = [i for i in range(vocab_size) if i not in special_token_ids]
    pos_indexes = list(range(n_nonspecial))
    random.shuffle(pos_indexes)
    pos_indexes = sorted(pos_indexes[:num_to_predict])
    masked_token_ids = sequence.copy()
    ids, positions = [], []
    for index in pos_indexes:
        if len(ids) >= num_to_predict:
            break
        replaced = sequence[index]
        rn = random.random()
        if rn < mask_token_rate + random_token_rate:
            replaced = random.choice(selectable_vocab)
        if rn < mask_token_rate:
            replaced = mask_token_id
        masked_token_ids[index] = replaced
        ids.append(sequence[index])
        positions.append(index)
    return masked_token_ids, np.array(positions, dtype=sequence.dtype), np.array(ids, dtype=sequence.dtype)


def mask_last_token_only(sequence: np.ndarray, mask_token_id: int):
    """dataloader_utils.py:264-269."""
    seq = np.array(sequence, dtype=np.int64)
    ids = np.array([seq[-1]], dtype=np.int64)
    seq[-1] = mask_token_id
    return seq, np.array([len(seq) - 1], dtype=np.int64), ids


def process_element(tokens: Sequence[int], max_seq_len: int, max_predictions_per_seq: int, vocab_size: int,
                    apply_mlm: bool, finetuning: bool, masked_lm_rate: float = 0.2, mask_token_rate: float = 1.0,
                    random_token_rate: float = 0.0) -> Dict[str, np.ndarray]:
    """bert4rec_preprocessor.py:48-116 on already-tokenized input."""
    tokens = list(tokens)
    if finetuning or len(tokens) <= max_seq_len:
        segments = tokens[-max_seq_len:]
    else:
        start = random.randint(0, len(tokens) - max_seq_len)
        segments = tokens[start:start + max_seq_len]
    ids = np.array(segments, dtype=np.int64)
    input_mask = np.ones_like(ids)
    labels = ids.copy()
    out: Dict[str, np.ndarray] = {}
    if apply_mlm:
        if not finetuning:
            ids, pos, mids = apply_dynamic_masking_task(ids, max_predictions_per_seq, MASK_TOKEN_ID,
                                                        [UNK_TOKEN_ID, PAD_TOKEN_ID], vocab_size,
                                                        selection_rate=masked_lm_rate, mask_token_rate=mask_token_rate,
                                                        random_token_rate=random_token_rate)
        else:
            ids, pos, mids = mask_last_token_only(ids, MASK_TOKEN_ID)
        w = np.ones_like(mids)
        padn = max_predictions_per_seq - mids.shape[0]
        if padn > 0:
            mids, pos, w = (np.pad(a, (0, padn), constant_values=PAD_TOKEN_ID) for a in (mids, pos, w))
        out.update(masked_lm_ids=mids.astype(np.int64), masked_lm_positions=pos.astype(np.int64),
                   masked_lm_weights=w.astype(np.int64))
    padn = max_seq_len - ids.shape[0]
    if padn > 0:
        ids, input_mask, labels = (np.pad(a, (0, padn), constant_values=PAD_TOKEN_ID) for a in (ids, input_mask, labels))
    out.update(labels=labels, input_word_ids=ids, input_mask=input_mask)
    return out


def synthetic_batch(B: int, L: int, P: int, V: int, rate: float = 0.2, seed: int = 0, ragged: bool = False,
                    finetune: bool = False) -> Dict[str, torch.Tensor]:
    """SURVEY §8(d) S-full / S-ragged / S-eval rows, deterministic (numpy default_rng(seed))."""
    rng = np.random.default_rng(seed)
    keys = ["input_word_ids", "input_mask", "labels", "masked_lm_positions", "masked_lm_ids", "masked_lm_weights"]
    arrs = {k: np.zeros((B, L if k in keys[:3] else P), dtype=np.int64) for k in keys}
    for b in range(B):
        n = int(rng.integers(5, L + 1)) if ragged else L
        seq = rng.integers(3, V, size=n).astype(np.int64)
        arrs["labels"][b, :n] = seq
        arrs["input_mask"][b, :n] = 1
        ids = seq.copy()
        if finetune:
            pos = np.array([n - 1])
        else:
            k = min(P, max(1, int(n * rate)))
            pos = np.sort(rng.choice(n, size=k, replace=False))
        arrs["masked_lm_positions"][b, :len(pos)] = pos
        arrs["masked_lm_ids"][b, :len(pos)] = seq[pos]
        arrs["masked_lm_weights"][b, :len(pos)] = 1
        ids[pos] = MASK_TOKEN_ID
        arrs["input_word_ids"][b, :n] = ids
    return {k: torch.from_numpy(v) for k, v in arrs.items()}
