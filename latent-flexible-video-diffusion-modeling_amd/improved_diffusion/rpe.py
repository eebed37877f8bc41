"""Relative-position-encoding attention modules (reference rpe.py) as parameter containers.

Compute lives in csrc/attention.hip (``lfvdm_rpe_nets``, ``lfvdm_attn_temporal``,
``lfvdm_attn_spatial``) and csrc/conv_igemm.hip (qkv / proj_out GEMMs); see ``_engine.py``.
"""
import torch.nn as nn

from .nn import normalization, zero_module, warn_use_checkpoint


class RPENet(nn.Module):
    """MLP producing R[b,t,s,h,:] from the frame distance and the diffusion-time embedding
    (reference rpe.py:8-31).  ``out`` starts at zero like the reference (rpe.py:14-16)."""

    def __init__(self, channels, num_heads, time_embed_dim):
        super().__init__()
        self.embed_distances = nn.Linear(3, channels)
        self.embed_diffusion_time = nn.Linear(time_embed_dim, channels)
        self.silu = nn.SiLU()
        self.out = zero_module(nn.Linear(channels, channels))
        self.channels = channels
        self.num_heads = num_heads


class RPE(nn.Module):
    """Holder of one RPENet (reference rpe.py:34-52).  The lookup-table variant of the reference
    is dead code there (``self.beta`` is undefined, rpe.py:50) and is not offered."""

    def __init__(self, channels, num_heads, time_embed_dim, use_rpe_net=False):
        super().__init__()
        if not use_rpe_net:
            raise NotImplementedError("only use_rpe_net=True is constructible (as in the reference)")
        self.num_heads = num_heads
        self.head_dim = channels // num_heads
        self.use_rpe_net = use_rpe_net
        self.rpe_net = RPENet(channels, num_heads, time_embed_dim)


class RPEAttention(nn.Module):
    """Attention over the last axis with optional q/k/v relative-position terms
    (reference rpe.py:99-174).  Registration order qkv, proj_out, norm, rpe_q, rpe_k, rpe_v."""

    def __init__(self, channels, num_heads, use_checkpoint=False, time_embed_dim=None, use_rpe_net=None,
                 use_rpe_q=True, use_rpe_k=True, use_rpe_v=True):
        super().__init__()
        self.num_heads = num_heads
        self.channels = channels
        self.scale = (channels // num_heads) ** -0.5
        self.use_checkpoint = warn_use_checkpoint(use_checkpoint)
        self.qkv = nn.Linear(channels, channels * 3)
        self.proj_out = zero_module(nn.Linear(channels, channels))
        self.norm = normalization(channels)
        if use_rpe_q or use_rpe_k or use_rpe_v:
            assert use_rpe_net is not None

        def make():
            return RPE(channels=channels, num_heads=num_heads, time_embed_dim=time_embed_dim, use_rpe_net=use_rpe_net)

        self.rpe_q = make() if use_rpe_q else None
        self.rpe_k = make() if use_rpe_k else None
        self.rpe_v = make() if use_rpe_v else None
