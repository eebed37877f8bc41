"""Scoped random streams: ``with RNG(seed): ...`` and ``@rng_decorator(seed)`` (same surface as the reference's
rng_util.py, which train_util.py uses to pick the visualisation batch and to sample with a fixed seed).

A ``_Streams`` value is a snapshot of every generator the training code draws from - Python's ``random``, NumPy's
global generator, torch's CPU generator and the generators of all visible GPUs.  ``RNG`` owns one such snapshot (its
private stream): entering swaps it in, leaving swaps the caller's streams back and keeps where the private one got to,
so re-entering the same object continues its sequence.
"""
import contextlib
import functools
import random
from dataclasses import dataclass
from typing import Any, List

import numpy as np
import torch as th


@dataclass
class _Streams:
    py: Any
    np_: Any
    cpu: th.Tensor
    gpus: List[th.Tensor]

    @classmethod
    def capture(cls):
        gpus = th.cuda.get_rng_state_all() if th.cuda.is_available() else []
        return cls(random.getstate(), np.random.get_state(), th.get_rng_state(), gpus)

    def install(self):
        random.setstate(self.py)
        np.random.set_state(self.np_)
        th.set_rng_state(self.cpu)
        if self.gpus and th.cuda.is_available():
            th.cuda.set_rng_state_all(self.gpus)


def set_random_seed(seed):
    """Seed the four generator families with seed, seed+1, seed+2, seed+3 (Python, torch CPU, torch GPUs, NumPy) -
    the reference's convention, so that a given ``--seed`` draws the same host-side numbers."""
    random.seed(seed)
    th.manual_seed(seed + 1)
    if th.cuda.is_available():
        th.cuda.manual_seed_all(seed + 2)
    np.random.seed(seed + 3)


def get_random_state():
    return _Streams.capture()


def set_random_state(state):
    state.install()


class RNG:
    def __init__(self, seed=None, state=None):
        outer = _Streams.capture()
        if seed is not None:
            set_random_seed(seed)
        elif state is not None:
            state.install()
        self._own = _Streams.capture()      # seed=None, state=None: starts where the caller's streams are now
        outer.install()
        self._outer = []

    def __enter__(self):
        self._outer.append(_Streams.capture())
        self._own.install()
        return self

    def __exit__(self, *exc):
        self._own = _Streams.capture()
        self._outer.pop().install()
        return False

    def get_state(self):
        return self._own

    def set_state(self, state):
        self._own = state


def rng_decorator(seed):
    """Run every call of the decorated function on a fresh stream seeded with ``seed``; the caller's streams are
    untouched."""
    def wrap(fn):
        @functools.wraps(fn)
        def run(*args, **kwargs):
            with RNG(seed):
                return fn(*args, **kwargs)
        return run
    return wrap


@contextlib.contextmanager
def preserved():
    """Leave every generator exactly as it was found (used around warm-up work that draws noise)."""
    snap = _Streams.capture()
    try:
        yield
    finally:
        snap.install()
