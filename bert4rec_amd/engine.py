"""Host-side driver of the HIP hot path: owns the flat parameter / gradient / Adam buffers, the workspace and the
device-resident train state, and calls the C ABI (include/b4r.h) on the current torch stream.

PyTorch is used here only as the owner of device memory and streams.  No compute happens in torch and nothing falls back
to CPU: without the HIP library and a GPU every compute method raises.
"""
from __future__ import annotations

import ctypes as C
import math
from typing import Dict, Iterable, List, Optional, Tuple

import torch

from . import _lib
from ._lib import AdamWConfig, B4RError, Batch, ModelConfig

BATCH_KEYS = ("input_word_ids", "input_mask", "masked_lm_positions", "masked_lm_ids")


def _ptr(t: Optional[torch.Tensor]) -> Optional[int]:
    return None if t is None else t.data_ptr()


def _stream(device: torch.device) -> int:
    if device.type != "cuda":
        raise B4RError("bert4rec_amd computes only on an AMD GPU (device 'cuda' under ROCm); got device '%s'" % device)
    return torch.cuda.current_stream(device).cuda_stream


def make_model_config(vocab_size: int, hidden_size: int, num_layers: int, num_attention_heads: int,
                      max_sequence_length: int, inner_dim: int, output_dropout: float = 0.1,
                      attention_dropout: float = 0.1, ln_eps: float = 1e-12) -> ModelConfig:
    return ModelConfig(int(vocab_size), int(hidden_size), int(num_layers), int(num_attention_heads), int(inner_dim),
                       int(max_sequence_length), float(output_dropout), float(attention_dropout), float(ln_eps))


def make_adamw_config(init_lr: float = 1e-4, num_train_steps: int = 400000, num_warmup_steps: int = 100,
                      end_lr: float = 0.0, weight_decay_rate: float = 0.01, beta_1: float = 0.9, beta_2: float = 0.999,
                      epsilon: float = 1e-6, gradient_clip_norm: float = 5.0, decay_mask: torch.Tensor = None) -> AdamWConfig:
    """Defaults of create_adam_w_optimizer, bert4rec/trainers/optimizers/__init__.py:7-15, and of
    AdamWeightDecay.gradient_clip_norm, adam_w_optimizer.py:67.  decay_mask: optional uint8 DEVICE tensor, one byte per float of
    the flat parameter buffer (a custom weight-decay selection); the config keeps it alive."""
    hp = AdamWConfig(float(init_lr), float(end_lr), int(num_train_steps), int(num_warmup_steps or 0),
                     float(weight_decay_rate), float(beta_1), float(beta_2), float(epsilon), float(gradient_clip_norm), None)
    if decay_mask is not None:
        if decay_mask.dtype != torch.uint8 or not decay_mask.is_cuda or not decay_mask.is_contiguous():
            raise ValueError("decay_mask must be a contiguous uint8 tensor on the GPU")
        hp.decay_mask = decay_mask.data_ptr()
        hp._decay_mask_tensor = decay_mask
    return hp


class ParamInfo:
    __slots__ = ("name", "offset", "rows", "cols", "ld", "decay")

    def __init__(self, name, offset, rows, cols, ld, decay):
        self.name, self.offset, self.rows, self.cols, self.ld, self.decay = name, offset, rows, cols, ld, decay


def param_table(cfg: ModelConfig) -> List[ParamInfo]:
    """Named layout of the flat parameter buffer (names = the reference's Keras variable names)."""
    lib = _lib.load()
    n = lib.b4r_param_count(C.byref(cfg))
    if n < 0:
        raise B4RError("invalid model config: " + _lib.last_error())
    out = []
    name = C.create_string_buffer(256)
    off, rows, cols, ld, dec = C.c_int64(), C.c_int32(), C.c_int32(), C.c_int32(), C.c_int32()
    for i in range(n):
        _lib.check(lib.b4r_param_info(C.byref(cfg), i, name, 256, C.byref(off), C.byref(rows), C.byref(cols),
                                      C.byref(ld), C.byref(dec)), "b4r_param_info")
        out.append(ParamInfo(name.value.decode(), off.value, rows.value, cols.value, ld.value, dec.value))
    return out


def keras_shape(name: str, info: ParamInfo, cfg: ModelConfig) -> Tuple[int, ...]:
    """Shape the reference's Keras variable of that name has (MHA kernels are [H,h,d] / [h,d,H])."""
    h, d = cfg.num_heads, cfg.hidden_size // cfg.num_heads
    if name.endswith(("self_attention/query/kernel", "self_attention/key/kernel", "self_attention/value/kernel")):
        return (info.rows, h, d)
    if name.endswith(("self_attention/query/bias", "self_attention/key/bias", "self_attention/value/bias")):
        return (h, d)
    if name.endswith("self_attention/attention_output/kernel"):
        return (h, d, info.cols)
    if info.rows == 1:
        return (info.cols,)
    return (info.rows, info.cols)


def device_mask_batch(tokens: torch.Tensor, max_predictions: int, vocab_size: int, selection_rate: float = 0.2,
                      mask_token_rate: float = 1.0, random_token_rate: float = 0.0, finetune: bool = False, seed: int = 0,
                      rows: Optional[torch.Tensor] = None, row_finetune: Optional[torch.Tensor] = None,
                      device=None) -> Dict[str, torch.Tensor]:
    """b4r_mask_batch on the current stream (include/b4r.h): the masked-LM task + padding of the six int64 tensors for a whole
    batch on the GPU.  No model is involved: the dataloader calls this for every batch of an epoch."""
    device = torch.device(device) if device is not None else torch.as_tensor(tokens).device
    st = _stream(device)   # raises on a non-GPU device: there is no host path behind this call
    tokens = torch.as_tensor(tokens).to(device=device, dtype=torch.int64).contiguous()
    if tokens.dim() != 2:
        raise ValueError(f"tokens must be rank 2 [rows, length], got shape {tuple(tokens.shape)}")
    L, P = tokens.shape[1], int(max_predictions)
    if rows is not None:
        rows = torch.as_tensor(rows).to(device=device, dtype=torch.int64).contiguous()
    if row_finetune is not None:
        row_finetune = torch.as_tensor(row_finetune).to(device=device, dtype=torch.int64).contiguous()
        if row_finetune.numel() != tokens.shape[0]:
            raise ValueError("row_finetune needs one flag per row of tokens")
    B = int(rows.numel()) if rows is not None else tokens.shape[0]
    out = {k: torch.empty((B, L), dtype=torch.int64, device=device) for k in ("input_word_ids", "input_mask", "labels")}
    out.update({k: torch.empty((B, P), dtype=torch.int64, device=device)
                for k in ("masked_lm_positions", "masked_lm_ids", "masked_lm_weights")})
    _lib.check(_lib.load().b4r_mask_batch(_ptr(tokens), _ptr(rows), _ptr(row_finetune), B, L, P, int(vocab_size),
                                          float(selection_rate), float(mask_token_rate), float(random_token_rate),
                                          1 if finetune else 0, int(seed) & 0xFFFFFFFFFFFFFFFF, _ptr(out["input_word_ids"]),
                                          _ptr(out["input_mask"]), _ptr(out["labels"]), _ptr(out["masked_lm_positions"]),
                                          _ptr(out["masked_lm_ids"]), _ptr(out["masked_lm_weights"]), st), "b4r_mask_batch")
    return out


# hipGraph captures only contain this library's launches on the capturing thread; "thread_local" keeps runtime calls made by
# other threads of the process during the capture (the RCCL watchdog polls events) from invalidating it
_CAPTURE_MODE = "thread_local"


class Engine:
    """One replica of the model on one GPU."""

    def __init__(self, cfg: ModelConfig, device="cuda", seed: int = 0):
        self.lib = _lib.load()
        self.cfg = cfg
        self.device = torch.device(device)
        total = self.lib.b4r_param_total_floats(C.byref(cfg))
        if total < 0:
            raise ValueError("invalid encoder configuration: " + _lib.last_error())
        self.n_params = int(total)
        self.n_decay = int(self.lib.b4r_param_decay_floats(C.byref(cfg)))
        self.table = param_table(cfg)
        self.params = torch.zeros(self.n_params, dtype=torch.float32, device=self.device)
        self.pooler = torch.zeros(int(self.lib.b4r_pooler_floats(C.byref(cfg))), dtype=torch.float32, device=self.device)
        self.grads: Optional[torch.Tensor] = None
        self.adam_m: Optional[torch.Tensor] = None
        self.adam_v: Optional[torch.Tensor] = None
        # b4r_train_state: 16 words (seed, step_lo, step(int64), 8 floats, 4 reserved)
        self.state = torch.zeros(_lib.STATE_WORDS, dtype=torch.int32, device=self.device)
        self._ws: Dict[Tuple[int, int, int], torch.Tensor] = {}
        self.rehearse_collectives = False   # True: dp_train_step issues its all-reduce even in a process group of one
        self.set_seed(seed)

    # ---- parameters -----------------------------------------------------------------------------------------------
    def view(self, name: str, buf: Optional[torch.Tensor] = None) -> torch.Tensor:
        """Strided view of one named variable inside a flat buffer (params by default)."""
        buf = self.params if buf is None else buf
        for e in self.table:
            if e.name == name:
                return torch.as_strided(buf, (e.rows, e.cols), (e.ld, 1), e.offset)
        if name == "pooler_transform/kernel":
            H = self.cfg.hidden_size
            return self.pooler[: H * H].view(H, H)
        if name == "pooler_transform/bias":
            H = self.cfg.hidden_size
            return self.pooler[H * H:]
        raise KeyError(name)

    def named_parameters(self, buf: Optional[torch.Tensor] = None) -> Iterable[Tuple[str, torch.Tensor]]:
        for e in self.table:
            yield e.name, self.view(e.name, buf)

    def variable_names(self) -> List[str]:
        return [e.name for e in self.table] + ["pooler_transform/kernel", "pooler_transform/bias"]

    def init_parameters(self, seed: int = 3, mlm_initializer: str = "glorot_uniform") -> None:
        """TruncatedNormal(0.02) everywhere (bert4rec_encoder.py:73-74), glorot_uniform for the MLM dense
        (bert4rec_model.py:42,76-81), LayerNorm gamma 1 / beta 0, biases 0."""
        g = torch.Generator().manual_seed(seed)
        host = torch.zeros(self.n_params, dtype=torch.float32)
        for e in self.table:
            v = torch.as_strided(host, (e.rows, e.cols), (e.ld, 1), e.offset)
            if e.name.endswith("gamma"):
                v.fill_(1.0)
            elif e.name.endswith(("beta", "bias")):
                v.zero_()
            elif e.name == "cls/predictions/transform/dense/kernel" and mlm_initializer == "glorot_uniform":
                lim = math.sqrt(6.0 / (e.rows + e.cols))
                v.copy_((torch.rand((e.rows, e.cols), generator=g) * 2 - 1) * lim)
            else:
                t = torch.empty((e.rows, e.cols))
                torch.nn.init.trunc_normal_(t, mean=0.0, std=0.02, a=-0.04, b=0.04, generator=g)
                v.copy_(t)
        self.params.copy_(host)
        H = self.cfg.hidden_size
        pk = torch.empty((H, H))
        torch.nn.init.trunc_normal_(pk, mean=0.0, std=0.02, a=-0.04, b=0.04, generator=g)
        self.pooler.zero_()
        self.pooler[: H * H].copy_(pk.reshape(-1))

    def load_named(self, tensors: Dict[str, torch.Tensor]) -> None:
        """Copy variables given under the reference's names/shapes into the flat buffer."""
        for name, t in tensors.items():
            v = self.view(name)
            v.copy_(t.detach().to(torch.float32).reshape(v.shape).to(self.device))

    def export_named(self, buf: Optional[torch.Tensor] = None) -> Dict[str, torch.Tensor]:
        out = {}
        for e in self.table:
            out[e.name] = self.view(e.name, buf).detach().cpu().clone().reshape(keras_shape(e.name, e, self.cfg))
        if buf is None:
            out["pooler_transform/kernel"] = self.view("pooler_transform/kernel").detach().cpu().clone()
            out["pooler_transform/bias"] = self.view("pooler_transform/bias").detach().cpu().clone()
        return out

    # ---- state ----------------------------------------------------------------------------------------------------
    def set_seed(self, seed: int) -> None:
        s = int(seed) & 0xFFFFFFFF
        self.state[_lib.ST_SEED] = s - (1 << 32) if s >= (1 << 31) else s

    def set_step(self, step: int) -> None:
        lo = int(step) & 0xFFFFFFFF
        self.state[_lib.ST_STEP_LO] = lo - (1 << 32) if lo >= (1 << 31) else lo
        self.state[_lib.ST_STEP:_lib.ST_STEP + 2].view(torch.int64)[0] = int(step)

    def read_state(self) -> Dict[str, float]:
        """Synchronising read of the device state (metrics of the last step)."""
        host = self.state.cpu()
        f = host.view(torch.float32)
        step = int(host[_lib.ST_STEP:_lib.ST_STEP + 2].view(torch.int64)[0])
        return dict(step=step, seed=int(host[_lib.ST_SEED]) & 0xFFFFFFFF, loss_sum=float(f[_lib.ST_LOSS_SUM]), valid_count=float(f[_lib.ST_VALID]),
                    correct_masked=float(f[_lib.ST_CORRECT_MASKED]), correct_all=float(f[_lib.ST_CORRECT_ALL]),
                    slots_all=float(f[_lib.ST_SLOTS_ALL]), grad_sqnorm=float(f[_lib.ST_SQNORM]),
                    grad_norm=float(f[_lib.ST_GRAD_NORM]), lr=float(f[_lib.ST_LR]))

    def ensure_training_buffers(self) -> None:
        if self.grads is None:
            from .distributed import alloc_grad_buffer
            # gradients + a small tail that carries the per-step sums through the data-parallel all-reduce
            self.grad_ext = alloc_grad_buffer(self.n_params, self.device)
            self.grads = self.grad_ext[: self.n_params]
            self.adam_m = torch.zeros_like(self.params)
            self.adam_v = torch.zeros_like(self.params)

    # ---- workspace ------------------------------------------------------------------------------------------------
    def workspace(self, B: int, L: int, P: int, encoder_only: bool = False) -> torch.Tensor:
        """encoder_only: a buffer that only has to hold the encoder's regions of the (B, L, P) layout (b4r_workspace_bytes_encoder: no
        [B*P, V] logits, no backward area) -- what the evaluation path's forward needs.  A larger buffer of the same key serves it too;
        a later full request of that key replaces an encoder-only buffer."""
        key = (B, L, P)
        ws = self._ws.get(key)
        sizes = self.__dict__.setdefault("_ws_bytes", {})
        nbytes = sizes.get((B, L, P, encoder_only))
        if nbytes is None:
            query = self.lib.b4r_workspace_bytes_encoder if encoder_only else self.lib.b4r_workspace_bytes
            nbytes = query(C.byref(self.cfg), B, L, P)
            if nbytes < 0:
                raise B4RError("b4r_workspace_bytes: " + _lib.last_error())
            sizes[(B, L, P, encoder_only)] = nbytes
        if ws is not None and ws.numel() * 4 < nbytes:
            ws = None
        if ws is None:
            # the workspace is pure scratch (nothing in it outlives a call sequence on one batch) and the library lays its regions out
            # from (B, L, P) alone, so a shape may use any buffer that is large enough: batches trimmed to their longest sequence
            # (dataloader_utils.make_batches(trim_padding=True)) share the buffer of the longest shape instead of owning one each
            fits = [w for w in self._ws.values() if w.numel() * 4 >= nbytes]
            if fits:
                ws = min(fits, key=lambda w: w.numel())
            else:
                distinct = {id(w) for w in self._ws.values()}
                if len(distinct) > 4:
                    # drop the least recently created workspaces, never one a captured hipGraph has its pointers baked into
                    # (train_step_graphed / dp_train_step_graphed pin theirs): a replay would read and write freed memory
                    pinned = self.__dict__.setdefault("_ws_pinned", set())
                    keep = {id(self._ws[k]) for k in pinned if k in self._ws}
                    for k in [k for k in self._ws if id(self._ws[k]) not in keep][: max(0, len(self._ws) - 4)]:
                        del self._ws[k]
                ws = torch.empty(nbytes // 4, dtype=torch.float32, device=self.device)
            self._ws[key] = ws
        return ws

    def _pin_workspace(self, B: int, L: int, P: int) -> None:
        self.workspace(B, L, P)
        self.__dict__.setdefault("_ws_pinned", set()).add((B, L, P))

    def region(self, name: str, B: int, L: int, P: int, encoder_only: bool = False) -> torch.Tensor:
        off, rows, cols, ld = C.c_int64(), C.c_int32(), C.c_int32(), C.c_int32()
        _lib.check(self.lib.b4r_workspace_region(C.byref(self.cfg), B, L, P, name.encode(), C.byref(off), C.byref(rows),
                                                 C.byref(cols), C.byref(ld)), "b4r_workspace_region")
        return torch.as_strided(self.workspace(B, L, P, encoder_only), (rows.value, cols.value), (ld.value, 1), off.value)

    # ---- batches --------------------------------------------------------------------------------------------------
    def prepare_batch(self, batch: Dict[str, torch.Tensor]) -> Tuple[Batch, Dict[str, torch.Tensor]]:
        """Move the int64 batch dict (bert4rec_model.py:15-22) to the device; returns the C struct and the tensors that
        must stay alive while the kernels run."""
        keep: Dict[str, torch.Tensor] = {}
        for k in BATCH_KEYS:
            if k in batch and batch[k] is not None:
                t = torch.as_tensor(batch[k])
                if t.dim() != 2:
                    raise ValueError(f"batch['{k}'] must be rank 2 [batch, length], got shape {tuple(t.shape)}")
                keep[k] = t.to(device=self.device, dtype=torch.int64).contiguous()
        if "input_word_ids" not in keep or "input_mask" not in keep:
            raise ValueError("batch needs 'input_word_ids' and 'input_mask'")
        B, L = keep["input_word_ids"].shape
        if keep["input_mask"].shape != (B, L):
            raise ValueError("input_mask shape differs from input_word_ids")
        P = 0
        if "masked_lm_positions" in keep:
            if keep["masked_lm_positions"].shape[0] != B:
                raise ValueError("masked_lm_positions batch size differs")
            P = keep["masked_lm_positions"].shape[1]
            if "masked_lm_ids" in keep and keep["masked_lm_ids"].shape != (B, P):
                raise ValueError("masked_lm_ids shape differs from masked_lm_positions")
        cb = Batch(_ptr(keep["input_word_ids"]), _ptr(keep["input_mask"]), _ptr(keep.get("masked_lm_positions")),
                   _ptr(keep.get("masked_lm_ids")), B, L, P)
        return cb, keep

    # ---- compute --------------------------------------------------------------------------------------------------
    def fused_head_supported(self) -> bool:
        """train-step masked-LM head that never materialises the logits (include/b4r.h, B4R_FLAG_FUSED_HEAD)"""
        return bool(self.lib.b4r_fused_head_supported(C.byref(self.cfg)))

    def forward(self, cb: Batch, training: bool = False, pooler: bool = True, fused_head: bool = False,
                head_rows_only: bool = False, encoder_only: bool = False) -> None:  # noqa: D401
        """fused_head: the loss / backward of the same step must be called with fused_head=True as well, and the
        "mlm_logits" region is not written.  head_rows_only (train steps: forward AND backward): the last layer's feed-forward half
        only on the rows the masked-LM head gathers; "sequence_output" is then defined on those rows only."""
        ws = self.workspace(cb.B, cb.L, cb.P, encoder_only=encoder_only and not pooler)
        flags = (_lib.FLAG_TRAINING if training else 0) | (_lib.FLAG_POOLER if pooler else 0) | \
                (_lib.FLAG_FUSED_HEAD if fused_head else 0) | (_lib.FLAG_HEAD_ROWS_ONLY if head_rows_only else 0) | \
                (_lib.FLAG_ENCODER_ONLY if encoder_only else 0)
        _lib.check(self.lib.b4r_forward(C.byref(self.cfg), C.byref(cb), _ptr(self.params), _ptr(self.pooler), _ptr(ws),
                                        ws.numel() * 4, _ptr(self.state), flags, _stream(self.device)), "b4r_forward")

    def begin_step(self) -> None:
        _lib.check(self.lib.b4r_state_begin_step(_ptr(self.state), _stream(self.device)), "b4r_state_begin_step")

    def loss(self, cb: Batch, want_grad: bool, fused_head: bool = False) -> None:
        ws = self.workspace(cb.B, cb.L, cb.P)
        _lib.check(self.lib.b4r_loss(C.byref(self.cfg), C.byref(cb), _ptr(ws), ws.numel() * 4, _ptr(self.state),
                                     (1 if want_grad else 0) | (_lib.LOSS_FUSED_HEAD if fused_head else 0),
                                     _stream(self.device)), "b4r_loss")

    def backward(self, cb: Batch, training: bool = True, fused_head: bool = False, grad_tail: bool = False,
                 head_rows_only: bool = False, loss_sums: bool = False) -> None:
        """grad_tail: also write the step's sums behind the gradients (data-parallel steps all-reduce grad_ext as one buffer);
        head_rows_only: as given to the forward of the same step; loss_sums (fused head only): the call also sets the state's loss /
        metric sums (B4R_FLAG_LOSS_SUMS: no begin_step() / loss() calls in front of it)"""
        self.ensure_training_buffers()
        ws = self.workspace(cb.B, cb.L, cb.P)
        flags = (_lib.FLAG_TRAINING if training else 0) | (_lib.FLAG_FUSED_HEAD if fused_head else 0) | \
                (_lib.FLAG_GRAD_TAIL if grad_tail else 0) | (_lib.FLAG_HEAD_ROWS_ONLY if head_rows_only else 0) | \
                (_lib.FLAG_LOSS_SUMS if loss_sums else 0)
        _lib.check(self.lib.b4r_backward(C.byref(self.cfg), C.byref(cb), _ptr(self.params), _ptr(self.grads), _ptr(ws),
                                         ws.numel() * 4, _ptr(self.state), flags, _stream(self.device)), "b4r_backward")

    def optimizer_step(self, hp: AdamWConfig, cb: Batch, reduced: bool = False) -> None:
        """reduced: the gradient buffer (with its tail of sums) went through the data-parallel all-reduce"""
        self.ensure_training_buffers()
        ws = self.workspace(cb.B, cb.L, cb.P)
        fn = self.lib.b4r_optimizer_step_reduced if reduced else self.lib.b4r_optimizer_step
        _lib.check(fn(C.byref(self.cfg), C.byref(hp), _ptr(self.params), _ptr(self.grads), _ptr(self.adam_m), _ptr(self.adam_v),
                      _ptr(ws), ws.numel() * 4, _ptr(self.state), _stream(self.device)), "b4r_optimizer_step")

    def train_step(self, hp: AdamWConfig, cb: Batch) -> None:
        """BERT4RecModel.train_step (bert4rec_model.py:151-173) as one enqueue; metrics stay on the device."""
        self.ensure_training_buffers()
        ws = self.workspace(cb.B, cb.L, cb.P)
        _lib.check(self.lib.b4r_train_step(C.byref(self.cfg), C.byref(hp), C.byref(cb), _ptr(self.params), _ptr(self.grads),
                                           _ptr(self.adam_m), _ptr(self.adam_v), _ptr(ws), ws.numel() * 4, _ptr(self.state),
                                           _stream(self.device)), "b4r_train_step")

    def train_step_graphed(self, hp: AdamWConfig, cb: Batch, max_graphs: int = 64) -> None:
        """train_step replayed from a captured hipGraph (one graph per distinct batch = per set of input pointers; the
        batch tensors must stay alive and in place).  Everything step-varying lives in the device state, so a replay is a
        full new step (new dropout masks, next lr).  GPU time is the same as eager (no launch gaps to remove: DESIGN.md
        §4); the host cost per step drops from ~0.72 ms of launches to ~0.10 ms.  The first step on a batch runs eagerly
        (one-time initialisations must not happen inside a capture), the second captures, later ones replay."""
        key = (cb.input_word_ids, cb.input_mask, cb.masked_lm_positions, cb.masked_lm_ids, cb.B, cb.L, cb.P,
               bytes(memoryview(hp)))
        graphs = self.__dict__.setdefault("_graphs", {})
        seen = self.__dict__.setdefault("_graph_seen", set())
        g = graphs.get(key)
        if g is not None:
            g.replay()
            return
        if key not in seen or len(graphs) >= max_graphs:
            seen.add(key)
            self.train_step(hp, cb)
            return
        torch.cuda.synchronize(self.device)
        self._pin_workspace(cb.B, cb.L, cb.P)   # the graph keeps this workspace's addresses
        g = torch.cuda.CUDAGraph()
        with torch.cuda.graph(g, capture_error_mode=_CAPTURE_MODE):
            self.train_step(hp, cb)        # capture only enqueues: the step itself runs at the replay below
        graphs[key] = g
        g.replay()

    def dp_train_step_graphed(self, hp: AdamWConfig, cb: Batch, group=None, max_graphs: int = 64) -> None:
        """dp_train_step with the two local halves of the step replayed from captured hipGraphs and the all-reduce (RCCL
        is not captured) between them; same first-eager / second-capture / then-replay protocol as train_step_graphed."""
        from .distributed import allreduce_step
        key = (cb.input_word_ids, cb.input_mask, cb.masked_lm_positions, cb.masked_lm_ids, cb.B, cb.L, cb.P,
               bytes(memoryview(hp)))
        graphs = self.__dict__.setdefault("_dp_graphs", {})
        seen = self.__dict__.setdefault("_dp_graph_seen", set())
        pair = graphs.get(key)
        if pair is None:
            if key not in seen or len(graphs) >= max_graphs:
                seen.add(key)
                self.dp_train_step(hp, cb, group)
                return
            self.ensure_training_buffers()
            fused = self.fused_head_supported()
            torch.cuda.synchronize(self.device)
            self._pin_workspace(cb.B, cb.L, cb.P)
            g_pre, g_post = torch.cuda.CUDAGraph(), torch.cuda.CUDAGraph()
            with torch.cuda.graph(g_pre, capture_error_mode=_CAPTURE_MODE):
                if not fused:
                    self.begin_step()
                self.forward(cb, training=True, pooler=False, fused_head=fused, head_rows_only=True)
                if not fused:
                    self.loss(cb, want_grad=True, fused_head=False)
                self.backward(cb, training=True, fused_head=fused, grad_tail=True, head_rows_only=True, loss_sums=fused)
            with torch.cuda.graph(g_post, capture_error_mode=_CAPTURE_MODE):
                self.optimizer_step(hp, cb, reduced=True)
            pair = graphs[key] = (g_pre, g_post)
        pair[0].replay()
        allreduce_step(self.grad_ext, group, self.rehearse_collectives)
        pair[1].replay()

    def dp_train_step(self, hp: AdamWConfig, cb: Batch, group=None) -> None:
        """Data-parallel train step: local forward/backward of the loss SUM, one all-reduce (RCCL over xGMI) of
        [grads | loss sums], then the clip + AdamW step on the reduced buffer (identical on every rank)."""
        from .distributed import allreduce_step
        self.ensure_training_buffers()
        fused = self.fused_head_supported()
        if not fused:
            self.begin_step()
        self.forward(cb, training=True, pooler=False, fused_head=fused, head_rows_only=True)
        if not fused:
            self.loss(cb, want_grad=True, fused_head=False)
        self.backward(cb, training=True, fused_head=fused, grad_tail=True, head_rows_only=True, loss_sums=fused)
        allreduce_step(self.grad_ext, group, self.rehearse_collectives)
        self.optimizer_step(hp, cb, reduced=True)

    def dp_idle_step(self, hp: AdamWConfig, group=None) -> None:
        """A data-parallel round in which THIS rank has no batch (the last round of an epoch whose batch count the world does not
        divide): zero gradients and zero sums go into the round's all-reduce, then the same clip + AdamW step on the reduced buffer
        as on every other rank -- parameters stay identical everywhere and no batch is dropped anywhere."""
        from .distributed import allreduce_step
        self.ensure_training_buffers()
        self.grad_ext.zero_()
        allreduce_step(self.grad_ext, group, self.rehearse_collectives)
        ws = self.__dict__.get("_idle_ws")
        if ws is None:
            ws = self._idle_ws = torch.empty(4096, dtype=torch.float32, device=self.device)   # the optimizer's norm partials
        _lib.check(self.lib.b4r_optimizer_step_reduced(C.byref(self.cfg), C.byref(hp), _ptr(self.params), _ptr(self.grads),
                                                       _ptr(self.adam_m), _ptr(self.adam_v), _ptr(ws), ws.numel() * 4,
                                                       _ptr(self.state), _stream(self.device)), "b4r_optimizer_step_reduced")

    def mask_batch(self, tokens: torch.Tensor, max_predictions: int, selection_rate: float = 0.2,
                   mask_token_rate: float = 1.0, random_token_rate: float = 0.0, finetune: bool = False,
                   seed: int = 0, rows: Optional[torch.Tensor] = None,
                   row_finetune: Optional[torch.Tensor] = None) -> Dict[str, torch.Tensor]:
        """b4r_mask_batch: tokens [B,L] int64 (right-padded with 0) -> the six-tensor batch dict on the device
        (bert4rec_preprocessor.py:48-116 for a whole batch; defaults of BERT4RecPreprocessor: rate 0.2, always [MASK]).
        rows [B]: batch = those rows of a dataset matrix tokens [U,L]; row_finetune [U]: per-row last-token-mask flags."""
        return device_mask_batch(tokens, int(max_predictions), self.cfg.vocab_size, selection_rate, mask_token_rate,
                                 random_token_rate, finetune, seed, rows, row_finetune, self.device)

    def sample_candidates(self, logp: torch.Tensor, exclude: torch.Tensor, gt: torch.Tensor, n_samples: int,
                          seed: int, short_flag: Optional[torch.Tensor] = None) -> torch.Tensor:
        """b4r_sample_candidates: exclude [R,E] int64 (-1 padded), gt [R] int64 -> cand [R, n_samples+1] int64 (device);
        raises ValueError when a row has fewer than n_samples drawable items (popular_random_sampler.py:56-58).
        short_flag: a device bool tensor [1] -- that condition is OR-ed into it instead of being read back here (the caller checks it
        once, e.g. at the end of an evaluation: no host synchronisation per batch)."""
        logp = logp.to(device=self.device, dtype=torch.float32).contiguous()
        exclude = exclude.to(device=self.device, dtype=torch.int64).contiguous()
        gt = gt.to(device=self.device, dtype=torch.int64).contiguous()
        R, E = exclude.shape
        cand = torch.empty((R, n_samples + 1), dtype=torch.int64, device=self.device)
        if short_flag is not None:   # a device bool / uint8 [1]: the kernel sets it itself (no scan of cand, no launch)
            if short_flag.element_size() != 1 or not short_flag.is_cuda:
                raise ValueError("short_flag must be a one-byte tensor on the GPU")
            _lib.check(self.lib.b4r_sample_candidates_flagged(_ptr(logp), logp.numel(), _ptr(exclude), E, _ptr(gt), R, n_samples,
                                                              int(seed) & 0xFFFFFFFFFFFFFFFF, _ptr(cand), _ptr(short_flag),
                                                              _stream(self.device)), "b4r_sample_candidates_flagged")
            return cand
        _lib.check(self.lib.b4r_sample_candidates(_ptr(logp), logp.numel(), _ptr(exclude), E, _ptr(gt), R, n_samples,
                                                  int(seed) & 0xFFFFFFFFFFFFFFFF, _ptr(cand), _stream(self.device)),
                   "b4r_sample_candidates")
        short = (cand[:, :n_samples] < 0).any()
        if bool(short):
            raise ValueError(f"The exclusion lists reduce the vocab too much to take a sample of size {n_samples} "
                             f"(since no duplicates are allowed).")
        return cand

    def encoder_forward(self, cb: Batch, training: bool = False, ranked_rows_only: bool = False) -> Batch:
        """Encoder only (no masked-LM head): the batch struct without its masked_lm_* pointers -> b4r_forward stops at the
        sequence output.  Returns that struct (its workspace key is (B, L, 0)).
        ranked_rows_only: the struct keeps its masked_lm_* pointers and the last layer's feed-forward half runs on the rows of the
        valid slots (masked_lm_ids != 0) only -- "sequence_output" (workspace key (B, L, P)) is defined on those rows alone."""
        if ranked_rows_only and cb.P > 0 and cb.masked_lm_ids:
            self.forward(cb, training=training, pooler=False, head_rows_only=True, encoder_only=True)
            return cb
        enc = Batch(cb.input_word_ids, cb.input_mask, None, None, cb.B, cb.L, 0)
        self.forward(enc, training=training, pooler=False)
        return enc

    def mlm_transform_rows(self, seq: torch.Tensor, rows: torch.Tensor) -> torch.Tensor:
        """b4r_mlm_transform_rows: tfm MaskedLM's gather -> dense(gelu) -> LayerNorm on the listed rows of seq [N,H] only."""
        rows = rows.to(device=self.device, dtype=torch.int64).contiguous()
        R, H = int(rows.numel()), self.cfg.hidden_size
        out = torch.empty((R, H), dtype=torch.float32, device=self.device)
        scratch = torch.empty(3 * (R * H + 4) + 2 * (R + 4), dtype=torch.float32, device=self.device)
        _lib.check(self.lib.b4r_mlm_transform_rows(C.byref(self.cfg), _ptr(self.params), _ptr(seq), seq.shape[0], _ptr(rows), R,
                                                   _ptr(out), _ptr(scratch), _stream(self.device)), "b4r_mlm_transform_rows")
        return out

    def rank_candidates(self, hidden: torch.Tensor, hidden_rows: Optional[torch.Tensor], cand: Optional[torch.Tensor],
                        gt: Optional[torch.Tensor], want_ranking: bool = True, want_scores: bool = False,
                        n_candidates: Optional[int] = None, n_rows: Optional[int] = None):
        """b4r_rank_candidates on `hidden` [*,H] (ld = stride(0)); cand [R,C] int64, or None = every row ranks the items
        0 .. n_candidates-1 (the whole vocabulary; n_rows rows); gt [R] int64 or None."""
        if cand is None:
            R, Cn = int(n_rows), int(n_candidates)
        else:
            R, Cn = cand.shape
            cand = cand.to(device=self.device, dtype=torch.int64).contiguous()
        gt_d = None if gt is None else gt.to(device=self.device, dtype=torch.int64).contiguous()
        rows_d = None if hidden_rows is None else hidden_rows.to(device=self.device, dtype=torch.int64).contiguous()
        ranking = torch.empty((R, Cn), dtype=torch.int64, device=self.device) if want_ranking else None
        gt_rank = torch.empty((R,), dtype=torch.int32, device=self.device) if gt is not None else None
        scores = torch.empty((R, Cn), dtype=torch.float32, device=self.device) if want_scores else None
        H = self.cfg.hidden_size
        need = int(self.lib.b4r_rank_scratch_bytes(R, Cn))
        scratch = None
        if need > 0:
            # large candidate lists: radix argsort in global memory; cap the scratch at 4 GiB (rows are ranked in groups)
            scratch = torch.empty(min(need, max(Cn * 20, 4 << 30)) // 4 + 4, dtype=torch.float32, device=self.device)
        _lib.check(self.lib.b4r_rank_candidates(_ptr(hidden), hidden.stride(0), _ptr(rows_d),
                                                _ptr(self.view("word_embeddings/embeddings")),
                                                _ptr(self.view("cls/predictions/output_bias/bias")), H, self.cfg.vocab_size, _ptr(cand), R, Cn,
                                                _ptr(gt_d), _ptr(ranking), _ptr(gt_rank), _ptr(scores), _ptr(scratch),
                                                0 if scratch is None else scratch.numel() * 4,
                                                _stream(self.device)), "b4r_rank_candidates")
        return ranking, gt_rank, scores

    def rank_metrics(self, gt_rank: torch.Tensor, families, cutoffs, gain_sums: torch.Tensor, users: torch.Tensor) -> None:
        """b4r_rank_metrics: add this batch's gain sums to the device accumulators (float64 [n], int64 [1])."""
        fam = (C.c_int32 * len(families))(*families)
        cut = (C.c_int32 * len(cutoffs))(*cutoffs)
        _lib.check(self.lib.b4r_rank_metrics(_ptr(gt_rank), int(gt_rank.numel()), fam, cut, len(families), _ptr(gain_sums),
                                             _ptr(users), _stream(self.device)), "b4r_rank_metrics")
