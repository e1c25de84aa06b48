"""Training callbacks of the reference's engine (ns/engine/callbacks.py:34-113): the small protocol through which a Model
hands per-iteration hooks to the Trainer (`Model.get_training_callbacks`, ns/models/base_model.py:96-100;
`Trainer.setup` collects them, `train` runs them before / after every iteration: ns/engine/trainer.py:154-160,252-267)."""
from __future__ import annotations

from dataclasses import dataclass
from enum import Enum, auto
from inspect import signature
from typing import Any, Callable, Dict, List, Optional, Tuple


@dataclass
class TrainingCallbackAttributes:
    optimizers: Optional[Any] = None
    grad_scaler: Optional[Any] = None
    pipeline: Optional[Any] = None


class TrainingCallbackLocation(Enum):
    BEFORE_TRAIN_ITERATION = auto()
    AFTER_TRAIN_ITERATION = auto()
    AFTER_TRAIN = auto()


class TrainingCallback:
    """func(*args, **kwargs, step=step) every `update_every_num_iters` iterations (or at the steps in `iters`, or always)"""

    def __init__(self, where_to_run: List[TrainingCallbackLocation], func: Callable, update_every_num_iters: Optional[int] = None,
                 iters: Optional[Tuple[int, ...]] = None, args: Optional[List] = None, kwargs: Optional[Dict] = None):
        if "step" not in signature(func).parameters:
            raise AssertionError(f"'step: int' must be an argument in the callback function 'func': {getattr(func, '__name__', func)}")
        self.where_to_run = where_to_run
        self.update_every_num_iters = update_every_num_iters
        self.iters = iters
        self.func = func
        self.args = args if args is not None else []
        self.kwargs = kwargs if kwargs is not None else {}

    def run_callback(self, step: int) -> None:
        if self.update_every_num_iters is not None:
            if step % self.update_every_num_iters == 0:
                self.func(*self.args, **self.kwargs, step=step)
        elif self.iters is not None:
            if step in self.iters:
                self.func(*self.args, **self.kwargs, step=step)
        else:
            self.func(*self.args, **self.kwargs, step=step)

    def run_callback_at_location(self, step: int, location: TrainingCallbackLocation) -> None:
        if location in self.where_to_run:
            self.run_callback(step=step)
