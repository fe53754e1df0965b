"""ADM-style UNet whose forward pass runs as hand-written HIP kernels on an AMD Instinct MI355X (gfx950).

Drop-in surface of the reference's ``nicediffusion/model.py``: the same class names, constructor keywords
(model.py:322-340), attributes and ``state_dict`` keys/shapes (``load_state_dict(strict=True)`` of reference
checkpoints works), and default initialisation consumes the RNG in the same order, so a seeded construction gives the
same weights.  The modules below only HOLD parameters; the arithmetic of ``forward`` is a plan of ``libnd_hip.so``
launches over NHWC buffers (see ``_engine.py``).  There is no CPU path: calling the model on CPU tensors raises.
"""
import os

import torch
import torch.nn as nn

from . import _hip
from ._engine import UNetPlan


class UsesSteps(nn.Module):
    """Marker base class: layers that consume the timestep embedding (reference model.py:31-36)."""


class UsesStepsSequential(nn.Sequential, UsesSteps):
    """Ordered container of layers; the engine routes the embedding to the ``UsesSteps`` children (model.py:40-48)."""

    def forward(self, x, step):
        raise _hip.NdHipError('blocks are executed through DiffusionModel.forward (fused HIP plan), not individually')


class Upsample(nn.Module):
    """Nearest-neighbour 2x upsampling, optionally followed by a 3x3 conv (model.py:51-80)."""

    def __init__(self, in_channels, with_conv, out_channels=None):
        super().__init__()
        self.with_conv = with_conv
        if with_conv:
            self.conv = nn.Conv2d(in_channels, out_channels if out_channels is not None else in_channels,
                                  kernel_size=(3, 3), stride=(1, 1), padding=(1, 1))


class Downsample(nn.Module):
    """2x downsampling by a stride-2 3x3 conv or by 2x2 average pooling (model.py:83-112)."""

    def __init__(self, in_channels, with_conv, out_channels=None):
        super().__init__()
        self.with_conv = with_conv
        if with_conv:
            self.conv = nn.Conv2d(in_channels, out_channels if out_channels is not None else in_channels,
                                  kernel_size=(3, 3), stride=(2, 2), padding=(1, 1))


class ResidualBlock(UsesSteps):
    """GN-SiLU-[resample]-conv3x3, timestep-conditioned GN-SiLU-conv3x3, plus skip (model.py:117-211)."""

    def __init__(self, in_channels, step_channels, dropout, upsample=False, downsample=False, use_conv=False,
                 out_channels=None, use_adaptive_gn=False, use_grad_checkpoints=False):
        super().__init__()
        out_channels = in_channels if out_channels is None else out_channels
        self.use_conv = use_conv
        self.use_adaptive_gn = use_adaptive_gn
        self.use_grad_checkpoints = use_grad_checkpoints
        self.resample_mode = 'up' if upsample else ('down' if downsample else None)
        self.resample = self.resample_mode is not None
        # parameter registration order fixes the state_dict order and the RNG stream of default init
        if out_channels == in_channels:
            self.skip = nn.Identity()
        else:
            k = 3 if use_conv else 1
            self.skip = nn.Conv2d(in_channels, out_channels, kernel_size=(k, k), stride=(1, 1), padding=(k // 2, k // 2))
        self.in_norm = nn.GroupNorm(num_groups=32, num_channels=in_channels, eps=1e-5)
        self.in_conv = nn.Conv2d(in_channels, out_channels, kernel_size=(3, 3), stride=(1, 1), padding=(1, 1))
        self.out_norm = nn.GroupNorm(num_groups=32, num_channels=out_channels, eps=1e-5)
        self.out_conv = zero_module(nn.Conv2d(out_channels, out_channels, kernel_size=(3, 3), stride=(1, 1),
                                              padding=(1, 1)))
        self.step_embedding = nn.Linear(step_channels, 2 * out_channels if use_adaptive_gn else out_channels)
        self.dropout = nn.Dropout(dropout)      # identity at inference


class AttentionBlock(nn.Module):
    """GN, 1x1 qkv, softmax(q k^T / sqrt(d)) v over H*W tokens, 1x1 projection, residual (model.py:214-291)."""

    def __init__(self, channels, num_heads=1, num_head_channels=None, split_qkv_first=True):
        super().__init__()
        if num_head_channels is None:
            self.num_heads = num_heads
        else:
            assert channels % num_head_channels == 0, \
                'channels {} is not divisible by num_head_channels {}'.format(channels, num_head_channels)
            self.num_heads = channels // num_head_channels
        self.split_qkv_first = split_qkv_first
        self.scale = (channels // self.num_heads) ** -0.5
        self.qkv_nin = nn.Conv1d(channels, 3 * channels, kernel_size=(1,), stride=(1,))
        self.norm = nn.GroupNorm(num_groups=32, num_channels=channels, eps=1e-5)
        self.proj_out = zero_module(nn.Conv1d(channels, channels, kernel_size=(1,), stride=(1,)))


class DiffusionModel(nn.Module):
    """UNet epsilon-predictor.  Same constructor as the reference (model.py:322-340)."""

    def __init__(self, resolution, in_channels, model_channels, out_channels, num_res_blocks, attention_resolutions,
                 dropout=0, channel_mult=(1, 2, 4, 8), conv_resample=True, num_classes=None, num_heads=1,
                 num_head_channels=None, resblock_updown=False, use_adaptive_gn=False, split_qkv_first=True,
                 use_grad_checkpoints=False):
        super().__init__()
        self.resolution = resolution
        self.in_channels = in_channels
        self.out_channels = out_channels
        self.model_channels = model_channels
        emb_dim = 4 * model_channels
        self.step_embed = nn.Sequential(nn.Linear(model_channels, emb_dim), nn.SiLU(), nn.Linear(emb_dim, emb_dim))
        if num_classes is not None:
            self.class_embedding = nn.Embedding(num_classes, embedding_dim=emb_dim)
        self.conditional = num_classes is not None
        self.num_classes = num_classes

        def res(cin, cout, **kw):
            return ResidualBlock(in_channels=cin, step_channels=emb_dim, dropout=dropout, out_channels=cout,
                                 use_adaptive_gn=use_adaptive_gn, use_grad_checkpoints=use_grad_checkpoints, **kw)

        def attn(c):
            return AttentionBlock(channels=c, split_qkv_first=split_qkv_first, num_heads=num_heads,
                                  num_head_channels=num_head_channels)

        levels = [int(model_channels * m) for m in channel_mult]
        ch = first = levels[0]
        self._feature_size = ch
        size = resolution
        self.downsampling = nn.ModuleList([UsesStepsSequential(
            nn.Conv2d(in_channels, ch, kernel_size=(3, 3), stride=(1, 1), padding=(1, 1)))])
        skip_channels = [ch]
        for depth, width in enumerate(levels):
            for _ in range(num_res_blocks):
                stage = [res(ch, width)]
                ch = width
                if size in attention_resolutions:
                    stage.append(attn(ch))
                self.downsampling.append(UsesStepsSequential(*stage))
                skip_channels.append(ch)
                self._feature_size += ch
            ch = width
            if depth + 1 < len(levels):
                if resblock_updown:
                    self.downsampling.append(UsesStepsSequential(res(ch, ch, downsample=True)))
                else:
                    self.downsampling.append(UsesStepsSequential(Downsample(ch, conv_resample, ch)))
                skip_channels.append(ch)
                size //= 2
                self._feature_size += ch

        self.middle_block = UsesStepsSequential(res(ch, None), attn(ch), res(ch, None))
        self._feature_size += ch

        self.upsampling = nn.ModuleList([])
        for depth in reversed(range(len(levels))):
            width = levels[depth]
            for i in range(num_res_blocks + 1):
                stage = [res(ch + skip_channels.pop(), width)]
                ch = width
                if size in attention_resolutions:
                    stage.append(attn(ch))
                if depth != 0 and i == num_res_blocks:
                    if resblock_updown:
                        stage.append(res(ch, ch, upsample=True))
                    else:
                        stage.append(Upsample(ch, conv_resample, ch))
                    size *= 2
                self._feature_size += ch
                self.upsampling.append(UsesStepsSequential(*stage))

        self.out = nn.Sequential(nn.GroupNorm(num_groups=32, num_channels=ch), nn.SiLU(),
                                 zero_module(nn.Conv2d(first, out_channels, kernel_size=(3, 3), stride=(1, 1),
                                                       padding=(1, 1))))
        self._plans = {}
        # Two host synchronisations per forward() / denoise() entry, each behind a switch for callers that want the call
        # sync-free (e.g. to capture forward() into their own hipGraph): the digest of the weights against the cached
        # plan's (verify_weights) and the range check of the labels (verify_labels)
        self.verify_weights = True     # digest the weights on the device before reusing a cached plan
        self.verify_labels = True      # nn.Embedding raises on an out-of-range label (model.py:459): so does forward()
        # arithmetic of the forward: 'fp32' = the reference's (exact fp32 MFMA kernels); 'bf16' = bf16 activations and
        # weights with fp32 accumulation (parameters stay fp32; the cast happens in the plan's weight repack)
        self.compute_dtype = os.environ.get('ND_COMPUTE_DTYPE', 'fp32')      # e.g. ND_COMPUTE_DTYPE=bf16 scripts/sample.py ...

    # -------------------------------------------------------------------------------------------- plan management
    def _residual_blocks(self):
        return [m for m in self.modules() if isinstance(m, ResidualBlock)]

    def _weight_signature(self):
        return tuple((p.data_ptr(), p._version) for p in self.parameters())

    def _weight_digest(self):
        """64-bit digest of every parameter's bytes, computed on the device (nd_checksum_segments: one HBM pass, ~0.3 ms
        for 1.2 GB, plus one 8-byte readback).  (data_ptr, _version) alone misses in-place rewrites through ``p.data``
        and EMA dicts that land on recycled allocator addresses; the plan keeps private repacked copies of the conv
        weights, so such a miss would silently mix old and new weights."""
        params = [p for p in self.parameters()]
        key = tuple(p.data_ptr() for p in params)
        seg = self.__dict__.get('_digest_segments')
        if seg is None or seg[0] != key:
            dev = params[0].device
            ptrs = torch.tensor([p.data_ptr() for p in params], dtype=torch.int64).to(dev)
            sizes = torch.tensor([p.numel() * p.element_size() for p in params], dtype=torch.int64).to(dev)
            seg = (key, ptrs, sizes, torch.zeros(1, dtype=torch.int64, device=dev))
            self.__dict__['_digest_segments'] = seg
        _, ptrs, sizes, out = seg
        _hip.check(_hip.load().nd_checksum_segments(ptrs.data_ptr(), sizes.data_ptr(), len(params), out.data_ptr(),
                                                    torch.cuda.current_stream().cuda_stream), 'nd_checksum_segments')
        return int(out.item())

    def invalidate_plans(self):
        """Forget every cached launch plan (and with them the hipGraphs built on them).  Called automatically by
        ``load_state_dict``, ``.to()`` and the EMA swap of ``Diffusion.denoise``; call it after editing weights by
        other in-place means if ``verify_weights`` has been switched off."""
        self._plans = {}

    def load_state_dict(self, *a, **k):
        self.invalidate_plans()
        return super().load_state_dict(*a, **k)

    def _plan(self, batch):
        """Plan for ``batch`` images on the parameters' device; rebuilt if any parameter's storage, version or
        contents changed since it was built."""
        dev = next(self.parameters()).device
        _hip.require_device(next(self.parameters()), 'model parameters')
        with torch.cuda.device(dev):
            sig = self._weight_signature()
            digest = self._weight_digest() if self.verify_weights else None
            pkey = (batch, self.compute_dtype)
            plan = self._plans.get(pkey)
            if plan is not None and (plan.weight_signature != sig or (digest is not None and plan.weight_digest != digest)):
                self.invalidate_plans()          # the other batch sizes' plans hold the same stale copies
                plan = None
            if plan is None:
                for p in self.parameters():
                    if p.dtype != torch.float32:
                        raise _hip.NdHipError('parameters must be fp32')
                with torch.no_grad():
                    plan = UNetPlan(self, batch, self.compute_dtype)
                plan.weight_digest = digest
                if len(self._plans) >= 4:
                    self._plans = {}
                self._plans[pkey] = plan
        return plan

    def _apply(self, fn, *a, **k):
        self._plans = {}
        return super()._apply(fn, *a, **k)

    def _check_labels(self, y):
        """nn.Embedding raises on an out-of-range index (model.py:459); the gather kernel must not clamp silently."""
        # (ONE device -> host sync per call -- torch.aminmax; ``verify_labels = False`` skips it for callers that have
        # validated their labels and want forward() free of host synchronisation)
        if y is not None and y.numel() and getattr(self, 'verify_labels', True):
            lo, hi = (int(v) for v in torch.aminmax(y))
            if lo < 0 or hi >= self.num_classes:
                raise IndexError('class label out of range: got [{}, {}], model has {} classes'.format(
                    lo, hi, self.num_classes))

    # -------------------------------------------------------------------------------------------- forward
    @torch.no_grad()
    def forward(self, x, timestep, y=None):
        """x [B, C, R, R] fp32 on the GPU, timestep [B] (original-scale index), y [B] int labels or None."""
        assert (y is not None) == self.conditional, 'pass y iff class-conditional model'
        assert x.shape[2] == self.resolution and x.shape[3] == self.resolution, \
            'incorrect resolution: {}'.format(x.shape[2:])
        _hip.require_device(x, 'x')
        self._check_labels(y)
        B = x.shape[0]
        plan = self._plan(B)
        lib = plan.lib
        with torch.cuda.device(plan.device):          # launches go to the current device's stream
            st = torch.cuda.current_stream().cuda_stream
            xc = x.to(plan.device).contiguous().float()
            _hip.check(lib.nd_nchw_to_nhwc(xc.data_ptr(), plan.x_in.data_ptr(), B, self.in_channels,
                                           self.resolution ** 2, plan.Cin_p, st), 'nd_nchw_to_nhwc')
            plan.t_in.copy_(timestep.to(torch.int64))
            if y is not None:
                plan.y_in.copy_(y.to(torch.int64))
            plan.run()
            out = torch.empty(B, self.out_channels, self.resolution, self.resolution, dtype=torch.float32,
                              device=plan.device)
            _hip.check(lib.nd_nhwc_to_nchw(plan.out.data_ptr(), out.data_ptr(), B, self.out_channels,
                                           self.resolution ** 2, plan.Cout_p, st), 'nd_nhwc_to_nchw')
        return out


def zero_module(module):
    """Zero every parameter of ``module`` (reference model.py:507-510)."""
    for p in module.parameters():
        p.detach().zero_()
    return module


def timestep_embedding(timesteps, embedding_dim, max_period=10000):
    """Sinusoidal embedding [B, dim]: cos half then sin half, zero pad for odd dim (model.py:514-523), on the GPU."""
    import math
    _hip.require_device(timesteps, 'timesteps')
    lib = _hip.load()
    B = timesteps.shape[0]
    half = embedding_dim // 2
    freqs = torch.exp(torch.arange(half, dtype=torch.float32) * -(math.log(max_period) / half)).to(timesteps.device)
    out = torch.empty(B, embedding_dim, dtype=torch.float32, device=timesteps.device)
    t = timesteps.to(torch.int64).contiguous()
    _hip.check(lib.nd_timestep_embed(t.data_ptr(), freqs.data_ptr(), B, embedding_dim, out.data_ptr(), embedding_dim,
                                     torch.cuda.current_stream().cuda_stream), 'nd_timestep_embed')
    return out
