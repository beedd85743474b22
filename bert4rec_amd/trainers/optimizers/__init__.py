"""Optimizer factory (mirrors bert4rec/trainers/optimizers/__init__.py:7-79)."""
from typing import Union

from .adam_w_optimizer import AdamWeightDecay, PolynomialDecay, WarmUp  # noqa: F401


def create_adam_w_optimizer(init_lr: float = 1e-4, num_train_steps: int = 400000, num_warmup_steps: int = 100,
                            end_lr: float = 0.0, weight_decay_rate: float = 0.01, beta_1: float = 0.9,
                            beta_2: float = 0.999, epsilon: float = 1e-6,
                            exclude_from_weight_decay: list = None) -> AdamWeightDecay:
    if exclude_from_weight_decay is None:
        exclude_from_weight_decay = ["LayerNorm", "layer_norm", "bias"]
    lr_schedule = PolynomialDecay(initial_learning_rate=init_lr, decay_steps=num_train_steps, end_learning_rate=end_lr)
    if num_warmup_steps:
        lr_schedule = WarmUp(initial_learning_rate=init_lr, decay_schedule_fn=lr_schedule, warmup_steps=num_warmup_steps)
    return AdamWeightDecay(learning_rate=lr_schedule, weight_decay_rate=weight_decay_rate, beta_1=beta_1, beta_2=beta_2,
                           epsilon=epsilon, exclude_from_weight_decay=exclude_from_weight_decay)


optimizers_map = {"adamw": create_adam_w_optimizer}


def get(identifier: Union[str, AdamWeightDecay] = "adamw", **kwargs) -> AdamWeightDecay:
    if isinstance(identifier, AdamWeightDecay):
        return identifier
    if identifier in optimizers_map:
        return optimizers_map[identifier](**kwargs)
    raise ValueError(f"{identifier} is an unknown optimizer identifier!")
