"""Seed / RNG-state scoping helpers (reference rng_util.py): ``RNG`` context and ``rng_decorator``."""
import random

import numpy as np
import torch as th


def set_random_seed(seed):
    random.seed(seed)
    th.manual_seed(seed + 1)
    if th.cuda.is_available():
        th.cuda.manual_seed_all(seed + 2)
    np.random.seed(seed + 3)


def get_random_state():
    return {"python": random.getstate(), "torch": th.get_rng_state(),
            "cuda": th.cuda.get_rng_state_all() if th.cuda.is_available() else [], "numpy": np.random.get_state()}


def set_random_state(state):
    random.setstate(state["python"])
    th.set_rng_state(state["torch"])
    if th.cuda.is_available() and state["cuda"]:
        th.cuda.set_rng_state_all(state["cuda"])
    np.random.set_state(state["numpy"])


class RNG():
    """``with RNG(seed):`` runs the body on a private random stream and restores the outer one."""

    def __init__(self, seed=None, state=None):
        self.state = get_random_state()
        with self:
            if seed is not None:
                set_random_seed(seed)
            elif state is not None:
                set_random_state(state)

    def __enter__(self):
        self.external_state = get_random_state()
        set_random_state(self.state)

    def __exit__(self, *args):
        self.state = get_random_state()
        set_random_state(self.external_state)

    def get_state(self):
        return self.state

    def set_state(self, state):
        self.state = state


class rng_decorator():
    def __init__(self, seed):
        self.seed = seed

    def __call__(self, f):
        def wrapped(*args, **kwargs):
            with RNG(self.seed):
                return f(*args, **kwargs)
        return wrapped
