"""Frame-conditioned video U-Net with the reference's module tree and API (reference unet.py).

The tree below only declares parameters (names, shapes, init and registration order are the
checkpoint ABI).  ``UNetVideoModel.forward`` executes on the MI355X through the
native engine (``_engine.py`` -> C ABI -> gfx950 kernels); there is no ATen/CPU path.
"""
from abc import abstractmethod

import torch as th
import torch.nn as nn

from .fp16_util import convert_module_to_f16, convert_module_to_f32
from .nn import SiLU, conv_nd, linear, avg_pool_nd, zero_module, normalization, warn_use_checkpoint
from .rpe import RPEAttention


class TimestepBlock(nn.Module):
    """Marker base: blocks that consume the timestep embedding (reference unet.py:24-34)."""

    @abstractmethod
    def forward(self, x, emb):
        ...


class TimestepEmbedAttnThingsSequential(nn.Sequential, TimestepBlock):
    """Container of one U-Net stage (reference unet.py:37-57); executed by the engine."""

    def forward(self, *a, **k):
        raise RuntimeError("stages are executed by the native engine; call UNetVideoModel.forward")


class Upsample(nn.Module):
    """Nearest x2 then 3x3 conv (reference unet.py:60-88); fused into one implicit-GEMM launch."""

    def __init__(self, channels, use_conv, dims=2):
        super().__init__()
        self.channels, self.use_conv, self.dims = channels, use_conv, dims
        if use_conv:
            self.conv = conv_nd(dims, channels, channels, 3, padding=1)


class Downsample(nn.Module):
    """3x3 stride-2 conv (reference unet.py:91-114)."""

    def __init__(self, channels, use_conv, dims=2):
        super().__init__()
        self.channels, self.use_conv, self.dims = channels, use_conv, dims
        stride = 2 if dims != 3 else (1, 2, 2)
        if use_conv:
            self.op = conv_nd(dims, channels, channels, 3, stride=stride, padding=1)
        else:
            self.op = avg_pool_nd(stride)


class ResBlock(TimestepBlock):
    """GN-SiLU-conv, FiLM from the timestep embedding, GN-SiLU-conv, skip
    (reference unet.py:117-207)."""

    def __init__(self, channels, emb_channels, dropout, out_channels=None, use_conv=False,
                 use_scale_shift_norm=False, dims=2, use_checkpoint=False):
        super().__init__()
        self.channels = channels
        self.emb_channels = emb_channels
        self.dropout = dropout
        self.out_channels = out_channels or channels
        self.use_conv = use_conv
        self.use_checkpoint = warn_use_checkpoint(use_checkpoint)
        self.use_scale_shift_norm = use_scale_shift_norm
        self.in_layers = nn.Sequential(normalization(channels), SiLU(),
                                       conv_nd(dims, channels, self.out_channels, 3, padding=1))
        self.emb_layers = nn.Sequential(
            SiLU(), linear(emb_channels, 2 * self.out_channels if use_scale_shift_norm else self.out_channels))
        self.out_layers = nn.Sequential(
            normalization(self.out_channels), SiLU(), nn.Dropout(p=dropout),
            zero_module(conv_nd(dims, self.out_channels, self.out_channels, 3, padding=1)))
        if self.out_channels == channels:
            self.skip_connection = nn.Identity()
        elif use_conv:
            self.skip_connection = conv_nd(dims, channels, self.out_channels, 3, padding=1)
        else:
            self.skip_connection = conv_nd(dims, channels, self.out_channels, 1)

    def forward(self, x, emb):
        raise RuntimeError("ResBlock is executed by the native engine; call UNetVideoModel.forward")


class FactorizedAttentionBlock(nn.Module):
    """Temporal (RPE, masked) then spatial attention (reference unet.py:210-243)."""

    def __init__(self, channels, num_heads, use_rpe_net, time_embed_dim=None, use_checkpoint=False):
        super().__init__()
        self.channels = channels
        self.num_heads = num_heads
        self.spatial_attention = RPEAttention(channels=channels, num_heads=num_heads, use_checkpoint=use_checkpoint,
                                              use_rpe_q=False, use_rpe_k=False, use_rpe_v=False)
        self.temporal_attention = RPEAttention(channels=channels, num_heads=num_heads, use_checkpoint=use_checkpoint,
                                               time_embed_dim=time_embed_dim, use_rpe_net=use_rpe_net)


class UNetVideoModel(nn.Module):
    """The full video U-Net (reference unet.py:246-464).  Same constructor, parameters and
    ``forward`` contract; executes natively on gfx950."""

    def __init__(self, in_channels, model_channels, out_channels, num_res_blocks, attention_resolutions,
                 image_size=None, dropout=0, channel_mult=(1, 2, 4, 8), conv_resample=True, dims=2,
                 use_checkpoint=False, num_heads=1, num_heads_upsample=-1, use_scale_shift_norm=False,
                 use_rpe_net=False):
        super().__init__()
        if num_heads_upsample == -1:
            num_heads_upsample = num_heads
        self.in_channels = in_channels + 1  # + observed-frame indicator (reference unet.py:290)
        self.model_channels = model_channels
        self.out_channels = out_channels
        self.num_res_blocks = num_res_blocks
        self.attention_resolutions = attention_resolutions
        self.dropout = dropout
        self.channel_mult = channel_mult
        self.conv_resample = conv_resample
        self.use_checkpoint = warn_use_checkpoint(use_checkpoint)
        self.num_heads = num_heads
        self.num_heads_upsample = num_heads_upsample
        self.use_rpe_net = use_rpe_net
        self.use_scale_shift_norm = use_scale_shift_norm
        self.dims = dims
        # False: parameter gradients are returned to autograd (hooks fire; DistributedDataParallel can wrap the model as
        # in reference train_util.py:116-125).  True (set by TrainLoop, which reduces its own flat gradient arena): the
        # backward kernels accumulate straight into p.grad (see _backward._GradMode).
        self.native_grad_accumulation = False

        ted = model_channels * 4
        self.time_embed = nn.Sequential(linear(model_channels, ted), SiLU(), linear(ted, ted))

        def res(cin, cout):
            return ResBlock(cin, ted, dropout, out_channels=cout, dims=dims, use_checkpoint=use_checkpoint,
                            use_scale_shift_norm=use_scale_shift_norm)

        def attn(c, heads):
            return FactorizedAttentionBlock(c, use_checkpoint=use_checkpoint, num_heads=heads,
                                            use_rpe_net=use_rpe_net, time_embed_dim=ted)

        self.input_blocks = nn.ModuleList(
            [TimestepEmbedAttnThingsSequential(conv_nd(dims, self.in_channels, model_channels, 3, padding=1))])
        skip_chans = [model_channels]
        ch, ds = model_channels, 1
        for level, mult in enumerate(channel_mult):
            for _ in range(num_res_blocks):
                layers = [res(ch, mult * model_channels)]
                ch = mult * model_channels
                if ds in attention_resolutions:
                    layers.append(attn(ch, num_heads))
                self.input_blocks.append(TimestepEmbedAttnThingsSequential(*layers))
                skip_chans.append(ch)
            if level != len(channel_mult) - 1:
                self.input_blocks.append(TimestepEmbedAttnThingsSequential(Downsample(ch, conv_resample, dims=dims)))
                skip_chans.append(ch)
                ds *= 2

        self.middle_block = TimestepEmbedAttnThingsSequential(res(ch, ch), attn(ch, num_heads), res(ch, ch))

        self.output_blocks = nn.ModuleList([])
        for level, mult in list(enumerate(channel_mult))[::-1]:
            for i in range(num_res_blocks + 1):
                layers = [res(ch + skip_chans.pop(), model_channels * mult)]
                ch = model_channels * mult
                if ds in attention_resolutions:
                    layers.append(attn(ch, num_heads_upsample))
                if level and i == num_res_blocks:
                    layers.append(Upsample(ch, conv_resample, dims=dims))
                    ds //= 2
                self.output_blocks.append(TimestepEmbedAttnThingsSequential(*layers))

        self.out = nn.Sequential(normalization(ch), SiLU(),
                                 zero_module(conv_nd(dims, model_channels, out_channels, 3, padding=1)))
        self._engine = None

    # -- reference API (fp16 is off by default and not part of the fp32 north star) -------------
    def convert_to_fp16(self):
        self.input_blocks.apply(convert_module_to_f16)
        self.middle_block.apply(convert_module_to_f16)
        self.output_blocks.apply(convert_module_to_f16)

    def convert_to_fp32(self):
        self.input_blocks.apply(convert_module_to_f32)
        self.middle_block.apply(convert_module_to_f32)
        self.output_blocks.apply(convert_module_to_f32)

    @property
    def inner_dtype(self):
        return next(self.input_blocks.parameters()).dtype

    def native_engine(self):
        """The per-model engine (plans, packed weights, workspaces); created lazily."""
        if self._engine is None:
            from ._engine import Engine
            self._engine = Engine(self)
        return self._engine

    def __getstate__(self):
        state = self.__dict__.copy()
        state["_engine"] = None  # device plans are rebuilt on demand (deepcopy / pickle safe)
        return state

    def _apply(self, fn, *args, **kwargs):
        # .to()/.cuda()/.float() re-allocate parameters: drop cached device pointers
        self._engine = None
        return super()._apply(fn, *args, **kwargs)

    def forward(self, x, *, x0, timesteps, frame_indices=None, obs_mask=None, latent_mask=None,
                return_attn_weights=False):
        """x, x0: (B,T,C,H,W); timesteps: (B,); frame_indices: (B,T) int64; masks: (B,T,1,1,1).
        Returns ``(out (B,T,out_channels,H,W), attns)`` exactly like reference unet.py:428-464."""
        if not x.is_cuda:
            raise RuntimeError(
                "UNetVideoModel runs on MI355X only (hand-written gfx950 kernels); move the model and "
                "inputs to the GPU with .to(dist_util.dev()) - there is no CPU fallback")
        if frame_indices is None or obs_mask is None or latent_mask is None:
            raise ValueError("frame_indices, obs_mask and latent_mask are required (the reference "
                             "fails on None in rpe.py:146 / unet.py:441 as well)")
        return self.native_engine().forward(x, x0, timesteps, frame_indices, obs_mask, latent_mask,
                                            return_attn_weights)
