"""The hook protocol between a Model and its Trainer, API-compatible with the reference's engine
(ns/engine/callbacks.py:34-113; `Model.get_training_callbacks`: ns/models/base_model.py:96-100; the trainer collects the hooks
in `setup` and fires them around every iteration: ns/engine/trainer.py:154-160,252-267).

Same names, constructor arguments and firing rule as the reference -- every `update_every_num_iters`-th step (step 0 included), or
exactly at the steps listed in `iters`, or on every step when neither is given; `func` always receives `step=` as a keyword --
written around one "is this step due" predicate that is chosen once, at construction."""
from __future__ import annotations

import enum
import inspect
from typing import Any, Callable, Dict, Iterable, List, Optional, Sequence


class TrainingCallbackLocation(enum.Enum):
    """the points of the training loop a hook can attach to"""

    BEFORE_TRAIN_ITERATION = 1
    AFTER_TRAIN_ITERATION = 2
    AFTER_TRAIN = 3


class TrainingCallbackAttributes:
    """What a model may look at when it builds its hooks: the optimizers, the gradient scaler and the pipeline (any may be None)."""

    __slots__ = ("optimizers", "grad_scaler", "pipeline")

    def __init__(self, optimizers: Any = None, grad_scaler: Any = None, pipeline: Any = None) -> None:
        self.optimizers, self.grad_scaler, self.pipeline = optimizers, grad_scaler, pipeline

    def __repr__(self) -> str:
        return f"TrainingCallbackAttributes(optimizers={self.optimizers!r}, grad_scaler={self.grad_scaler!r}, pipeline={self.pipeline!r})"


def _schedule(every: Optional[int], at: Optional[Iterable[int]]) -> Callable[[int], bool]:
    """step -> "the hook fires": a period wins over an explicit step list, no schedule means always"""
    if every is not None:
        period = int(every)
        return lambda step: step % period == 0
    if at is not None:
        wanted = frozenset(int(i) for i in at)
        return lambda step: step in wanted
    return lambda step: True


class TrainingCallback:
    def __init__(self, where_to_run: Sequence[TrainingCallbackLocation], func: Callable, update_every_num_iters: Optional[int] = None,
                 iters: Optional[Sequence[int]] = None, args: Optional[List] = None, kwargs: Optional[Dict] = None) -> None:
        if "step" not in inspect.signature(func).parameters:
            raise AssertionError(f"a training callback is called with step=<int>: {getattr(func, '__name__', repr(func))} does not take it")
        self.where_to_run = list(where_to_run)
        self.func = func
        self.update_every_num_iters, self.iters = update_every_num_iters, iters
        self.args: List = list(args or ())
        self.kwargs: Dict = dict(kwargs or {})
        self._due = _schedule(update_every_num_iters, iters)

    def run_callback(self, step: int) -> None:
        """fire if `step` is on the hook's schedule"""
        if self._due(step):
            self.func(*self.args, step=step, **self.kwargs)

    def run_callback_at_location(self, step: int, location: TrainingCallbackLocation) -> None:
        """fire if the hook is attached to `location` and `step` is on its schedule"""
        if location in self.where_to_run:
            self.run_callback(step)
