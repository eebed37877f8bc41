"""Functional fp32 CPU restatement of ``UNetVideoModel.forward`` (TEST INFRASTRUCTURE).

Works on a plain ``dict name -> torch.Tensor`` with the reference's state-dict key names
 and a small config dict; no nn.Module, no autograd requirement (autograd
still works, which the backward-parity tests use).  Every function cites the reference
lines it restates.  Pinned against the real reference by tests/golden (see __init__).
"""
import math

import torch
import torch.nn.functional as F


# --------------------------------------------------------------------------- structure
def layer_plan(cfg):
    """Block structure of the U-Net as a list of (prefix, kind, info) in execution order.

    Follows the constructor loops of unet.py:310-397: one input conv, then per level
    ``num_res_blocks`` x [ResBlock (+attention if ds in attention_resolutions)] and a
    Downsample between levels; middle = Res, Attn, Res; output side mirrors it with
    ``num_res_blocks+1`` blocks per level consuming the skip stack, Upsample on the last
    block of every level but the outermost.
    """
    ch0 = cfg["model_channels"]
    mult = tuple(cfg["channel_mult"])
    nrb = cfg["num_res_blocks"]
    att = set(cfg["attention_resolutions"])
    inputs, skips = [], [ch0]
    inputs.append(("input_blocks.0", [("0", "conv_in", dict(cin=cfg["in_channels"] + 1, cout=ch0))]))
    ch, ds, idx = ch0, 1, 1
    for level, m in enumerate(mult):
        for _ in range(nrb):
            mods = [("0", "res", dict(cin=ch, cout=m * ch0))]
            ch = m * ch0
            if ds in att:
                mods.append(("1", "attn", dict(ch=ch)))
            inputs.append((f"input_blocks.{idx}", mods))
            skips.append(ch)
            idx += 1
        if level != len(mult) - 1:
            inputs.append((f"input_blocks.{idx}", [("0", "down", dict(ch=ch))]))
            skips.append(ch)
            idx += 1
            ds *= 2
    middle = ("middle_block", [("0", "res", dict(cin=ch, cout=ch)),
                               ("1", "attn", dict(ch=ch)),
                               ("2", "res", dict(cin=ch, cout=ch))])
    outputs, idx = [], 0
    for level, m in list(enumerate(mult))[::-1]:
        for i in range(nrb + 1):
            mods = [("0", "res", dict(cin=ch + skips.pop(), cout=m * ch0))]
            ch = m * ch0
            if ds in att:
                mods.append((str(len(mods)), "attn", dict(ch=ch)))
            if level and i == nrb:
                mods.append((str(len(mods)), "up", dict(ch=ch)))
                ds //= 2
            outputs.append((f"output_blocks.{idx}", mods))
            idx += 1
    return inputs, middle, outputs


def param_shapes(cfg):
    """name -> shape for every parameter, in ``named_parameters()`` order of the reference
    (registration order: time_embed, input_blocks, middle_block, output_blocks, out)."""
    ch0 = cfg["model_channels"]
    ted = 4 * ch0
    shapes = {}

    def lin(p, i, o):
        shapes[p + ".weight"] = (o, i)
        shapes[p + ".bias"] = (o,)

    def conv(p, i, o, k):
        shapes[p + ".weight"] = (o, i, k, k)
        shapes[p + ".bias"] = (o,)

    def gn(p, c):
        shapes[p + ".weight"] = (c,)
        shapes[p + ".bias"] = (c,)

    def res(p, cin, cout):
        gn(p + ".in_layers.0", cin)
        conv(p + ".in_layers.2", cin, cout, 3)
        lin(p + ".emb_layers.1", ted, 2 * cout if cfg["use_scale_shift_norm"] else cout)
        gn(p + ".out_layers.0", cout)
        conv(p + ".out_layers.3", cout, cout, 3)
        if cin != cout:
            conv(p + ".skip_connection", cin, cout, 1)

    def attn_one(p, c, rpe):
        lin(p + ".qkv", c, 3 * c)
        lin(p + ".proj_out", c, c)
        gn(p + ".norm", c)
        if rpe:
            for r in ("rpe_q", "rpe_k", "rpe_v"):
                q = f"{p}.{r}.rpe_net"
                lin(q + ".embed_distances", 3, c)
                lin(q + ".embed_diffusion_time", ted, c)
                lin(q + ".out", c, c)

    def attn(p, c):
        # registration order in FactorizedAttentionBlock.__init__ (unet.py:212-221)
        attn_one(p + ".spatial_attention", c, False)
        attn_one(p + ".temporal_attention", c, True)

    lin("time_embed.0", ch0, ted)
    lin("time_embed.2", ted, ted)
    inputs, middle, outputs = layer_plan(cfg)
    for prefix, mods in inputs + [middle] + outputs:
        for sub, kind, info in mods:
            p = f"{prefix}.{sub}"
            if kind == "conv_in":
                conv(p, info["cin"], info["cout"], 3)
            elif kind == "res":
                res(p, info["cin"], info["cout"])
            elif kind == "attn":
                attn(p, info["ch"])
            elif kind == "down":
                conv(p + ".op", info["ch"], info["ch"], 3)
            elif kind == "up":
                conv(p + ".conv", info["ch"], info["ch"], 3)
    gn("out.0", ch0)
    conv("out.2", ch0, cfg["out_channels"], 3)
    return shapes


# --------------------------------------------------------------------------- leaf ops
def silu(x):
    """nn.py:12-14."""
    return x * torch.sigmoid(x)


def group_norm32(x, w, b):
    """nn.py:17-19,95-102: 32 groups, eps 1e-5, statistics over (C/32, *spatial).
    (The reference computes in fp32; the oracle follows the dtype of the weights so that the
    same code gives an fp64 "truth" when handed float64 tensors.)"""
    return F.group_norm(x.to(w.dtype), 32, w, b, 1e-5).type(x.dtype)


def timestep_embedding(t, dim, max_period=10000.0, dtype=torch.float32):
    """nn.py:105-123: [cos(t*f) | sin(t*f)], f_i = exp(-ln(max_period)*i/half)."""
    half = dim // 2
    freqs = torch.exp(-math.log(max_period) * torch.arange(half, dtype=dtype) / half)
    args = t[:, None].to(dtype) * freqs[None]
    emb = torch.cat([torch.cos(args), torch.sin(args)], dim=-1)
    if dim % 2:
        emb = torch.cat([emb, torch.zeros_like(emb[:, :1])], dim=-1)
    return emb


def res_block(sd, p, x, emb, use_scale_shift_norm=True, keep=None):
    """ResBlock._forward, unet.py:194-207.  Dropout (unet.py:166) is the identity at p=0 / eval; in training mode
    it multiplies SiLU(h) by ``keep`` (0 or 1/(1-p) per element, shape of h) - the caller supplies the draw."""
    h = group_norm32(x, sd[p + ".in_layers.0.weight"], sd[p + ".in_layers.0.bias"])
    h = F.conv2d(silu(h), sd[p + ".in_layers.2.weight"], sd[p + ".in_layers.2.bias"], padding=1)
    e = F.linear(silu(emb), sd[p + ".emb_layers.1.weight"], sd[p + ".emb_layers.1.bias"])[:, :, None, None]
    gw, gb = sd[p + ".out_layers.0.weight"], sd[p + ".out_layers.0.bias"]
    if use_scale_shift_norm:
        scale, shift = torch.chunk(e, 2, dim=1)
        h = group_norm32(h, gw, gb) * (1 + scale) + shift
    else:
        h = group_norm32(h + e, gw, gb)
    h = silu(h)
    if keep is not None:
        h = h * keep
    h = F.conv2d(h, sd[p + ".out_layers.3.weight"], sd[p + ".out_layers.3.bias"], padding=1)
    if (p + ".skip_connection.weight") in sd:
        x = F.conv2d(x, sd[p + ".skip_connection.weight"], sd[p + ".skip_connection.bias"])
    return x + h


def rpe_net(sd, p, temb, rel, heads):
    """RPENet.forward, rpe.py:20-31.  rel: (B,T,T) integer frame-index differences."""
    B, T, _ = rel.shape
    relf = rel.to(temb.dtype)
    feats = torch.stack([torch.log1p(relf.clamp(min=0)), torch.log1p((-relf).clamp(min=0)),
                         (rel == 0).to(temb.dtype)], dim=-1)
    C = sd[p + ".out.weight"].shape[0]
    hid = F.linear(temb, sd[p + ".embed_diffusion_time.weight"], sd[p + ".embed_diffusion_time.bias"]).view(B, T, 1, C) \
        + F.linear(feats, sd[p + ".embed_distances.weight"], sd[p + ".embed_distances.bias"])
    out = F.linear(silu(hid), sd[p + ".out.weight"], sd[p + ".out.bias"])
    return out.view(B, T, T, heads, C // heads)


def rpe_attention(sd, p, x, temb, frame_indices, attn_mask, heads, use_rpe):
    """RPEAttention._forward, rpe.py:133-174.  x: (B, D, C, T); attends over the last axis.

    Returns (y, attn) with y of x's shape.  Note the residual is taken on the *normalised*
    x (rpe.py:136,172) and q is pre-scaled for both the content and the rpe_k term
    (rpe.py:143-149) while rpe_q receives k*scale (rpe.py:152).
    """
    B, D, C, T = x.shape
    Fh = C // heads
    scale = Fh ** -0.5
    xn = group_norm32(x.reshape(B * D, C, T), sd[p + ".norm.weight"], sd[p + ".norm.bias"])
    xn = xn.view(B, D, C, T).permute(0, 1, 3, 2)  # B D T C
    qkv = F.linear(xn, sd[p + ".qkv.weight"], sd[p + ".qkv.bias"]).reshape(B, D, T, 3, heads, Fh)
    q, k, v = (qkv[:, :, :, i].permute(0, 1, 3, 2, 4) for i in range(3))  # B D H T F
    q = q * scale
    logits = q @ k.transpose(-2, -1)
    if use_rpe:
        rel = frame_indices.unsqueeze(-1) - frame_indices.unsqueeze(-2)  # B T T
        Rk = rpe_net(sd, p + ".rpe_k.rpe_net", temb, rel, heads)
        Rq = rpe_net(sd, p + ".rpe_q.rpe_net", temb, rel, heads)
        Rv = rpe_net(sd, p + ".rpe_v.rpe_net", temb, rel, heads)
        # rpe.py:72-74 "bdhtf,btshf->bdhts"
        logits = logits + torch.einsum("bdhtf,btshf->bdhts", q, Rk)
        logits = logits + torch.einsum("bdhtf,btshf->bdhts", k * scale, Rq).transpose(-1, -2)
    if attn_mask is not None:
        m = attn_mask.view(B, T)
        same = m[:, None, :] * m[:, :, None] + (1 - m[:, None, :]) * (1 - m[:, :, None])
        neg = torch.zeros_like(same)
        neg[same == 0] = float("inf")
        logits = logits - neg.view(B, 1, 1, T, T)
    attn = torch.softmax(logits, dim=-1)  # reference: softmax in fp32 (rpe.py:163)
    out = attn @ v
    if use_rpe:
        out = out + torch.einsum("bdhts,btshf->bdhtf", attn, Rv)
    out = out.permute(0, 1, 3, 2, 4).reshape(B, D, T, C)
    out = F.linear(out, sd[p + ".proj_out.weight"], sd[p + ".proj_out.bias"])
    y = (xn + out).permute(0, 1, 3, 2)
    return y, attn


def factorized_attention(sd, p, x, temb, attn_mask, T, frame_indices, heads, attn_log=None):
    """FactorizedAttentionBlock.forward, unet.py:223-243: temporal (with RPE + mask) then
    spatial (plain MHA) attention."""
    BT, C, H, W = x.shape
    B = BT // T
    xt = x.view(B, T, C, H, W).permute(0, 3, 4, 2, 1).reshape(B, H * W, C, T)
    xt, a_t = rpe_attention(sd, p + ".temporal_attention", xt, temb, frame_indices,
                            attn_mask.flatten(start_dim=2).squeeze(dim=2), heads, True)
    xs = xt.view(B, H, W, C, T).permute(0, 4, 3, 1, 2).reshape(B, T, C, H * W)
    xs, a_s = rpe_attention(sd, p + ".spatial_attention", xs, temb, None, None, heads, False)
    if attn_log is not None:
        # rpe.py:128-131: mean over heads, abs
        attn_log["temporal"].append(a_t.detach().reshape(B * H * W, -1, T, T).mean(dim=1).abs())
        attn_log["spatial"].append(a_s.detach().reshape(B * T, -1, H * W, H * W).mean(dim=1).abs())
    return xs.reshape(BT, C, H, W)


# --------------------------------------------------------------------------- full forward
def unet_forward(sd, cfg, x, x0, timesteps, frame_indices, obs_mask, latent_mask,
                 return_attn_weights=False, dropout_keep=None):
    """UNetVideoModel.forward, unet.py:428-464.  x,x0: (B,T,C,H,W); timesteps (B,) (already
    rescaled floats or ints); masks (B,T,1,1,1).  Returns (out (B,T,Cout,H,W), attns|None).
    ``dropout_keep``: optional list of per-ResBlock dropout factors (N,C,H,W) in execution order (training mode)."""
    keeps = list(dropout_keep) if dropout_keep is not None else None
    B, T, C, H, W = x.shape
    heads = cfg["num_heads"]
    ssn = cfg["use_scale_shift_norm"]
    ts = timesteps.view(B, 1).expand(B, T).reshape(B * T)
    attn_mask = (obs_mask + latent_mask).clip(max=1)
    h = torch.cat([x * (1 - obs_mask) + x0 * obs_mask, torch.ones_like(x[:, :, :1]) * obs_mask], dim=2)
    h = h.reshape(B * T, C + 1, H, W)
    emb = timestep_embedding(ts, cfg["model_channels"], dtype=sd["time_embed.0.weight"].dtype)
    emb = F.linear(emb, sd["time_embed.0.weight"], sd["time_embed.0.bias"])
    emb = F.linear(silu(emb), sd["time_embed.2.weight"], sd["time_embed.2.bias"])
    attns = {"spatial": [], "temporal": [], "mixed": []} if return_attn_weights else None

    def run(prefix, mods, h):
        for sub, kind, info in mods:
            p = f"{prefix}.{sub}"
            if kind == "conv_in":
                h = F.conv2d(h, sd[p + ".weight"], sd[p + ".bias"], padding=1)
            elif kind == "res":
                h = res_block(sd, p, h, emb, ssn, keep=keeps.pop(0) if keeps is not None else None)
            elif kind == "attn":
                h = factorized_attention(sd, p, h, emb, attn_mask, T, frame_indices, heads, attns)
            elif kind == "down":  # unet.py:108-114
                h = F.conv2d(h, sd[p + ".op.weight"], sd[p + ".op.bias"], stride=2, padding=1)
            elif kind == "up":    # unet.py:78-88: nearest x2 then conv
                h = F.interpolate(h, scale_factor=2, mode="nearest")
                h = F.conv2d(h, sd[p + ".conv.weight"], sd[p + ".conv.bias"], padding=1)
        return h

    inputs, middle, outputs = layer_plan(cfg)
    hs = []
    for prefix, mods in inputs:
        h = run(prefix, mods, h)
        hs.append(h)
    h = run(*middle, h)
    for prefix, mods in outputs:
        h = run(prefix, mods, torch.cat([h, hs.pop()], dim=1))
    h = silu(group_norm32(h, sd["out.0.weight"], sd["out.0.bias"]))
    out = F.conv2d(h, sd["out.2.weight"], sd["out.2.bias"], padding=1)
    return out.view(B, T, cfg["out_channels"], H, W), attns


def make_cfg(in_channels=4, model_channels=64, num_res_blocks=1, channel_mult=(1, 2, 2, 2),
             attention_resolutions=(1, 2), num_heads=4, use_scale_shift_norm=True, use_rpe_net=True,
             out_channels=None):
    return dict(in_channels=in_channels, model_channels=model_channels, num_res_blocks=num_res_blocks,
                channel_mult=tuple(channel_mult), attention_resolutions=tuple(attention_resolutions),
                num_heads=num_heads, use_scale_shift_norm=use_scale_shift_norm, use_rpe_net=use_rpe_net,
                out_channels=in_channels if out_channels is None else out_channels)
