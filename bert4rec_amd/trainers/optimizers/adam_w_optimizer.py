"""AdamWeightDecay + WarmUp (mirror bert4rec/trainers/optimizers/adam_w_optimizer.py).

These objects only DESCRIBE the optimizer; the update itself (clip by global norm -> decoupled decay -> Keras Adam,
adam_w_optimizer.py:100-137) is the fused HIP kernel b4r_adamw_step over the model's flat parameter buffer, with the
learning-rate schedule evaluated on the device from the device-resident iteration counter."""
from __future__ import annotations

import re
from typing import Optional, Sequence

import numpy as np

from ...engine import make_adamw_config


class PolynomialDecay:
    """tf.keras.optimizers.schedules.PolynomialDecay(power=1, cycle=False), float32 arithmetic."""

    def __init__(self, initial_learning_rate: float, decay_steps: int, end_learning_rate: float = 0.0, power: float = 1.0):
        if power != 1.0:
            raise NotImplementedError("only power=1.0 (the reference's setting) is implemented in the optimizer kernel")
        self.initial_learning_rate = initial_learning_rate
        self.decay_steps = decay_steps
        self.end_learning_rate = end_learning_rate
        self.power = power

    def __call__(self, step) -> np.float32:
        f32 = np.float32
        gs = min(f32(step), f32(self.decay_steps))
        p = f32(gs / f32(self.decay_steps))
        return f32(f32(f32(self.initial_learning_rate) - f32(self.end_learning_rate)) * f32(f32(1.0) - p)
                   + f32(self.end_learning_rate))


class WarmUp:
    """adam_w_optimizer.py:6-45: linear warm-up from 0, then the wrapped schedule evaluated at the RAW step."""

    def __init__(self, initial_learning_rate, decay_schedule_fn, warmup_steps, power=1.0, name=None):
        if power != 1.0:
            raise NotImplementedError("only power=1.0 is implemented in the optimizer kernel")
        self.initial_learning_rate = initial_learning_rate
        self.warmup_steps = warmup_steps
        self.power = power
        self.decay_schedule_fn = decay_schedule_fn
        self.name = name

    def __call__(self, step) -> np.float32:
        f32 = np.float32
        if f32(step) < f32(self.warmup_steps):
            return f32(self.initial_learning_rate) * f32(f32(step) / f32(self.warmup_steps))
        return self.decay_schedule_fn(step)

    def get_config(self):
        return {"initial_learning_rate": self.initial_learning_rate, "decay_schedule_fn": self.decay_schedule_fn,
                "warmup_steps": self.warmup_steps, "power": self.power, "name": self.name}


class AdamWeightDecay:
    """adam_w_optimizer.py:48-168."""

    def __init__(self, learning_rate=0.001, beta_1=0.9, beta_2=0.999, epsilon=1e-7, amsgrad=False, weight_decay_rate=0.0,
                 include_in_weight_decay: Optional[Sequence[str]] = None,
                 exclude_from_weight_decay: Optional[Sequence[str]] = None, gradient_clip_norm=5.0,
                 name="AdamWeightDecay", **kwargs):
        if amsgrad:
            raise NotImplementedError("amsgrad is not implemented (the reference never enables it)")
        self.learning_rate = learning_rate
        self.beta_1, self.beta_2, self.epsilon = beta_1, beta_2, epsilon
        self.weight_decay_rate = weight_decay_rate
        self.gradient_clip_norm = gradient_clip_norm
        self._include_in_weight_decay = include_in_weight_decay
        self._exclude_from_weight_decay = exclude_from_weight_decay
        self.name = name
        self.iterations = 0  # host mirror; the authoritative counter lives in the device state

    def _do_use_weight_decay(self, param_name: str) -> bool:
        """adam_w_optimizer.py:154-168"""
        if self.weight_decay_rate == 0:
            return False
        if self._include_in_weight_decay:
            for r in self._include_in_weight_decay:
                if re.search(r, param_name) is not None:
                    return True
        if self._exclude_from_weight_decay:
            for r in self._exclude_from_weight_decay:
                if re.search(r, param_name) is not None:
                    return False
        return True

    def lr(self, step: int) -> float:
        s = self.learning_rate
        return float(s(step)) if callable(s) else float(s)

    def kernel_config(self, variables, n_params: int = None, device=None):
        """Translate into the by-value b4r_adamw_config.  The kernel's built-in rule decays the 'decay region' of the flat buffer
        (kernels + embedding tables) = what the default exclusion list selects.  Any other include / exclude selection
        (adam_w_optimizer.py:154-168) becomes a per-element mask: `variables` are then the engine's table entries (name, decay,
        offset, rows, cols, ld), `n_params` the length of the flat buffer, `device` where the mask goes."""
        entries = [(v.name, v.decay, v) if hasattr(v, "name") else (v[0], v[1], None) for v in variables]
        mask = None
        if self.weight_decay_rate != 0 and any(self._do_use_weight_decay(n) != bool(d) for n, d, _ in entries):
            if n_params is None or device is None or any(e is None for _, _, e in entries):
                raise ValueError("a custom weight-decay selection needs the parameter table (name, offset, rows, cols, ld), the "
                                 "buffer length and the device")
            import numpy as np
            import torch
            m = np.zeros(int(n_params), dtype=np.uint8)
            for name, _, e in entries:
                if self._do_use_weight_decay(name):
                    for r in range(e.rows):
                        m[e.offset + r * e.ld: e.offset + r * e.ld + e.cols] = 1
            mask = torch.from_numpy(m).to(device)
        s = self.learning_rate
        if isinstance(s, WarmUp):
            d = s.decay_schedule_fn
            if not isinstance(d, PolynomialDecay) or d.initial_learning_rate != s.initial_learning_rate:
                raise NotImplementedError("WarmUp must wrap a PolynomialDecay with the same initial learning rate")
            init, end, steps, warm = s.initial_learning_rate, d.end_learning_rate, d.decay_steps, s.warmup_steps
        elif isinstance(s, PolynomialDecay):
            init, end, steps, warm = s.initial_learning_rate, s.end_learning_rate, s.decay_steps, 0
        elif isinstance(s, (int, float)):
            init, end, steps, warm = float(s), float(s), 1, 0   # constant learning rate
        else:
            raise NotImplementedError(f"unsupported learning-rate schedule {type(s).__name__}")
        return make_adamw_config(init, steps, warm, end, self.weight_decay_rate, self.beta_1, self.beta_2, self.epsilon,
                                 self.gradient_clip_norm, decay_mask=mask)

    def get_config(self):
        return {"name": self.name, "beta_1": self.beta_1, "beta_2": self.beta_2, "epsilon": self.epsilon,
                "weight_decay_rate": self.weight_decay_rate, "gradient_clip_norm": self.gradient_clip_norm}
