"""Opt-in HIP-event timing of individual kernel launches on torch's current stream (the stream every
libpresight_hip launch goes to).  Used by bench.py to measure the dominant kernel live inside the timed region."""
from __future__ import annotations

from collections import defaultdict
from contextlib import contextmanager

import torch

_enabled = False
_only = None  # None: every region; else the set of region names that are timed
_events = defaultdict(list)


def enable(flag: bool = True, only=None):
    """only: time just these regions (an event pair costs the stream a few microseconds: bench.py times the dominant kernels
    inside its timed steps and everything else in a separate pass)"""
    global _enabled, _only
    _enabled = flag
    _only = None if only is None else set(only)
    if flag:
        _events.clear()


def enabled(name: str = None) -> bool:
    return _enabled and (name is None or _only is None or name in _only)


@contextmanager
def region(name: str, extend: bool = False):
    """extend: this interval belongs to the region's previous entry (one logical launch whose kernels are enqueued in two pieces
    with other work in between); its time is added to that entry"""
    if not enabled(name):
        yield
        return
    a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    a.record()
    yield
    b.record()
    if extend and _events[name]:
        _events[name][-1].append((a, b))
    else:
        _events[name].append([(a, b)])


def summary():
    """name -> (launches, mean ms); synchronises."""
    torch.cuda.synchronize()
    return {k: (len(v), sum(a.elapsed_time(b) for e in v for a, b in e) / max(len(v), 1)) for k, v in _events.items()}
