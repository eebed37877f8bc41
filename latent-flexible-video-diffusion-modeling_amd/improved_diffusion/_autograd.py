"""Autograd bridge: differentiable wrappers around the native forward/backward launch plans."""
import torch as th

from . import _native as nat


class _MaskedMSE(th.autograd.Function):
    """out[b] = mean_{t,c,h,w}((target - pred)^2 * mask[b,t])  (reference gaussian_diffusion.py:787-788,
    nn.py:86-92).  Forward is one HIP reduction; backward is the closed form
    d/dpred = -2 (target - pred) * mask / inner * grad[b]."""

    @staticmethod
    def forward(ctx, target, pred, mask):
        B, T = pred.shape[0], pred.shape[1]
        frame_inner = pred[0, 0].numel()
        target, pred = target.contiguous(), pred.contiguous()
        m = None if mask is None else mask.reshape(B, T).to(th.float32).contiguous()
        out = th.empty(B, device=pred.device, dtype=th.float32)
        nat.masked_mse(target, pred, m, out, B, T, frame_inner)
        ctx.save_for_backward(target, pred, m)
        return out

    @staticmethod
    def backward(ctx, g):
        target, pred, m = ctx.saved_tensors
        B, T = pred.shape[0], pred.shape[1]
        d = th.empty_like(pred)
        nat.check(nat.lib().lfvdm_masked_mse_bwd(nat.ptr(target), nat.ptr(pred), nat.ptr(m), nat.ptr(g.contiguous().float()),
                                                 nat.ptr(d), B, T, pred[0, 0].numel(), nat.stream()), "lfvdm_masked_mse_bwd")
        return None, d, None


def masked_mse(target, pred, mask):
    return _MaskedMSE.apply(target, pred, mask)


def unet_apply(engine, x, x0, timesteps, frame_indices, obs_mask, latent_mask, return_attn_weights):
    from ._backward import UNetFunction
    return UNetFunction.run(engine, x, x0, timesteps, frame_indices, obs_mask, latent_mask, return_attn_weights)
