"""Opt-in HIP-event timing of individual kernel launches on torch's current stream (the stream every
libpresight_hip launch goes to).  Used by bench.py to measure the dominant kernel live inside the timed region."""
from __future__ import annotations

from collections import defaultdict
from contextlib import contextmanager

import torch

_enabled = False
_events = defaultdict(list)


def enable(flag: bool = True):
    global _enabled
    _enabled = flag
    if flag:
        _events.clear()


def enabled() -> bool:
    return _enabled


@contextmanager
def region(name: str):
    if not _enabled:
        yield
        return
    a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    a.record()
    yield
    b.record()
    _events[name].append((a, b))


def summary():
    """name -> (launches, mean ms); synchronises."""
    torch.cuda.synchronize()
    return {k: (len(v), sum(a.elapsed_time(b) for a, b in v) / max(len(v), 1)) for k, v in _events.items()}
