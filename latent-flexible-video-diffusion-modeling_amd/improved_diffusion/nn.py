"""Leaf modules and helpers of the reference's ``nn.py`` surface.

The modules here are parameter containers with the reference's names, shapes and default
initialisation (state-dict ABI); on the hot path they are never called one by
one — ``UNetVideoModel.forward`` hands the whole tree to the native engine (``_engine.py``).
"""
import math

import torch as th
import torch.nn as nn


_checkpoint_warned = [False]


def warn_use_checkpoint(flag):
    """``use_checkpoint=True`` (reference nn.py:126-172, used at unet.py:190-192 and rpe.py:127) is accepted for
    constructor compatibility and has NO effect here: the native backward keeps the activations it needs (results are
    identical to the reference's, peak memory is not reduced).  Said once per process."""
    if flag and not _checkpoint_warned[0]:
        _checkpoint_warned[0] = True
        import warnings
        warnings.warn("use_checkpoint=True is accepted but ignored by the native gfx950 path: activations are kept for "
                      "the backward pass (same results as the reference, no memory saving)", RuntimeWarning, stacklevel=3)
    return bool(flag)


class SiLU(nn.Module):
    """x * sigmoid(x)  (reference nn.py:12-14).  Fused into the conv operand load natively."""

    def forward(self, x):
        return x * th.sigmoid(x)


class GroupNorm32(nn.GroupNorm):
    """GroupNorm computed in fp32 (reference nn.py:17-19)."""

    def forward(self, x):
        return super().forward(x.float()).type(x.dtype)


def conv_nd(dims, *args, **kwargs):
    """Conv module factory (reference nn.py:22-32).  Only dims == 2 is on the native path."""
    if dims == 1:
        return nn.Conv1d(*args, **kwargs)
    if dims == 2:
        return nn.Conv2d(*args, **kwargs)
    if dims == 3:
        return nn.Conv3d(*args, **kwargs)
    raise ValueError(f"unsupported dimensions: {dims}")


def linear(*args, **kwargs):
    return nn.Linear(*args, **kwargs)


def avg_pool_nd(dims, *args, **kwargs):
    if dims == 1:
        return nn.AvgPool1d(*args, **kwargs)
    if dims == 2:
        return nn.AvgPool2d(*args, **kwargs)
    if dims == 3:
        return nn.AvgPool3d(*args, **kwargs)
    raise ValueError(f"unsupported dimensions: {dims}")


def normalization(channels):
    """32-group GroupNorm (reference nn.py:95-102)."""
    return GroupNorm32(32, channels)


def zero_module(module):
    """Zero all parameters of ``module`` (reference nn.py:68-74)."""
    with th.no_grad():
        for p in module.parameters():
            p.zero_()
    return module


def scale_module(module, scale):
    with th.no_grad():
        for p in module.parameters():
            p.mul_(scale)
    return module


def update_ema(target_params, source_params, rate=0.99):
    """targ <- rate*targ + (1-rate)*src (reference nn.py:55-65).  Device tensors go through one
    multi-tensor launch per dtype/device group instead of two launches per tensor."""
    target_params, source_params = list(target_params), list(source_params)
    if not target_params:
        return
    with th.no_grad():
        tg = [t.detach() for t in target_params]
        sr = [s.detach() for s in source_params]
        th._foreach_mul_(tg, rate)
        th._foreach_add_(tg, sr, alpha=1 - rate)


def mean_flat(tensor, mask=None):
    """Mean over all non-batch dims, after an optional mask multiply (reference nn.py:86-92)."""
    if mask is not None:
        tensor = tensor * mask
    return tensor.mean(dim=list(range(1, tensor.dim())))


def timestep_freqs(dim, max_period=10000):
    """Frequency table of the sinusoidal embedding.  The reference evaluates it with torch.exp
    on the HOST and moves it to the device (nn.py:116-118); doing the same keeps the
    device-side arguments t*f bit-identical."""
    half = dim // 2
    return th.exp(-math.log(max_period) * th.arange(start=0, end=half, dtype=th.float32) / half)


def timestep_embedding(timesteps, dim, max_period=10000):
    """[cos(t f) | sin(t f)] (reference nn.py:105-123).  Host-side helper; the native engine
    builds the embedding inside its first row-dot kernel."""
    freqs = timestep_freqs(dim, max_period).to(device=timesteps.device)
    args = timesteps[:, None].float() * freqs[None]
    emb = th.cat([th.cos(args), th.sin(args)], dim=-1)
    if dim % 2:
        emb = th.cat([emb, th.zeros_like(emb[:, :1])], dim=-1)
    return emb


def checkpoint(func, inputs, params, flag):
    """Activation checkpointing hook of the reference (nn.py:126-142).  The native engine
    recomputes normalised operands in its backward kernels anyway; ``flag`` is accepted for API
    compatibility and evaluated without re-materialisation."""
    return func(*inputs)
