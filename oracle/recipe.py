"""Closed-form parameter / input recipe shared by the golden-vector generator and the tests.

TEST INFRASTRUCTURE (see oracle/__init__.py).  Weights are never stored in fixtures:
every state-dict entry is a pure function of (key name, shape), computed with exact
64-bit integer hashing (splitmix64) so it is bit-identical on every machine.

Zero-initialised layers of the reference (unet.py:168-170,402; rpe.py:14-16,112) get
non-zero values here as well, otherwise golden outputs would pin nothing.
"""
import zlib

import numpy as np

_M64 = np.uint64(0xFFFFFFFFFFFFFFFF)


def _splitmix64(x):
    """Vectorised splitmix64 finaliser on uint64 arrays (wrapping arithmetic)."""
    with np.errstate(over="ignore"):
        x = (x + np.uint64(0x9E3779B97F4A7C15)) & _M64
        z = x
        z = ((z ^ (z >> np.uint64(30))) * np.uint64(0xBF58476D1CE4E5B9)) & _M64
        z = ((z ^ (z >> np.uint64(27))) * np.uint64(0x94D049BB133111EB)) & _M64
        z = z ^ (z >> np.uint64(31))
    return z


def uniform_pm1(tag, n):
    """n deterministic float64 values in [-1, 1) derived from the string ``tag``."""
    with np.errstate(over="ignore"):
        seed = np.uint64(zlib.crc32(tag.encode("utf-8"))) * np.uint64(0x100000001B3)
    idx = np.arange(n, dtype=np.uint64)
    with np.errstate(over="ignore"):
        h = _splitmix64(idx * np.uint64(0xD6E8FEB86659FD93) + seed)
    top = (h >> np.uint64(11)).astype(np.float64)  # 53 random bits
    return top * (2.0 / float(1 << 53)) - 1.0


def gaussianish(tag, n):
    """Approximately N(0,1): sum of 4 uniforms, rescaled (exact arithmetic, no libm)."""
    u = sum(uniform_pm1(f"{tag}#{j}", n) for j in range(4))
    return u * (1.0 / np.sqrt(4.0 / 3.0))


def fill_param(name, shape):
    """Value of state-dict entry ``name`` (float32 ndarray of ``shape``)."""
    n = int(np.prod(shape)) if len(shape) else 1
    u = uniform_pm1(name, n)
    leaf = name.rsplit(".", 1)[-1]
    if leaf == "weight" and len(shape) == 1:
        # GroupNorm scale (nn.py:95-102): around 1
        v = 1.0 + 0.2 * u
    elif leaf == "bias":
        v = 0.1 * u
    else:
        fan_in = int(np.prod(shape[1:]))
        v = u * np.sqrt(3.0 / fan_in) * 1.2
    return v.astype(np.float32).reshape(shape)


def fill_state_dict(shapes):
    """shapes: dict name -> tuple.  Returns dict name -> float32 ndarray."""
    return {k: fill_param(k, tuple(s)) for k, s in shapes.items()}


def make_inputs(tag, B, T, C, H, W, num_timesteps=1000, max_index=1000, n_pad=0):
    """Deterministic model inputs (numpy).  Row 0 has contiguous frame indices, the
    other rows sorted sparse indices (train-like, train_util.py:193-241).  The last
    ``n_pad`` frames have obs=latent=0 (the "padding clique" of rpe.py:156-163)."""
    x = gaussianish(tag + "/x", B * T * C * H * W).reshape(B, T, C, H, W).astype(np.float32)
    x0 = (0.8 * gaussianish(tag + "/x0", B * T * C * H * W)).reshape(B, T, C, H, W).astype(np.float32)
    tfrac = (uniform_pm1(tag + "/t", B) + 1.0) * 0.5
    t = np.minimum((tfrac * num_timesteps).astype(np.int64), num_timesteps - 1)
    fi = np.zeros((B, T), dtype=np.int64)
    for b in range(B):
        if b == 0:
            fi[b] = np.arange(T)
        else:
            order = np.argsort(uniform_pm1(f"{tag}/fi{b}", max_index), kind="stable")
            fi[b] = np.sort(order[:T])
    n_obs = max(T // 3, 1)
    obs = np.zeros((B, T, 1, 1, 1), dtype=np.float32)
    lat = np.zeros((B, T, 1, 1, 1), dtype=np.float32)
    obs[:, :n_obs] = 1.0
    lat[:, n_obs:T - n_pad] = 1.0
    if B > 1 and T >= 4:
        # second row: interleave observed frames so the two cliques are not contiguous
        obs[1] = 0.0
        lat[1] = 0.0
        obs[1, 0:T - n_pad:3] = 1.0
        lat[1] = (1.0 - obs[1])
        if n_pad:
            lat[1, T - n_pad:] = 0.0
            obs[1, T - n_pad:] = 0.0
    return dict(x=x, x0=x0, t=t, frame_indices=fi, obs_mask=obs, latent_mask=lat)
