"""CPU oracle for the UNet forward on the sampling hot path.  TEST INFRASTRUCTURE ONLY.

This file is a plain-PyTorch (CPU, fp32) restatement of the reference's UNet forward, written
functionally over a ``state_dict`` so that it shares no code with the product package.  It may be
imported only by ``tests/``, ``__graft_entry__.smoke()`` and the ``cpu_baseline`` leg of
``bench.py``.  The product path (``nice-diffusion_amd/nicediffusion``) never imports it.

Parity status: PINNED.  ``tools/gen_golden.py`` imports the real reference (``/root/reference``) in
the build container, runs both on the same seeded weights/inputs and commits the reference outputs
under ``tests/golden/``; ``tests/test_oracle_golden.py`` re-checks this file against those vectors.

Reference anchors (paths relative to /root/reference):
  * topology ............ nicediffusion/model.py:363-449
  * forward ............. nicediffusion/model.py:451-476
  * ResidualBlock ....... nicediffusion/model.py:188-211
  * AttentionBlock ...... nicediffusion/model.py:260-291
  * Up/Downsample ....... nicediffusion/model.py:76-80, 107-112
  * timestep_embedding .. nicediffusion/model.py:514-523
"""
import math

import torch
import torch.nn.functional as F

GN_GROUPS = 32
GN_EPS = 1e-5


def default_cfg(**kw):
    """Constructor defaults of the reference model (model.py:322-340)."""
    cfg = dict(dropout=0, channel_mult=(1, 2, 4, 8), conv_resample=True, num_classes=None, num_heads=1,
               num_head_channels=None, resblock_updown=False, use_adaptive_gn=False, split_qkv_first=True)
    cfg.update(kw)
    return cfg


# ----------------------------------------------------------------------------------------------------------------
# topology: a flat description of every block, in state_dict naming (model.py:363-449)
# ----------------------------------------------------------------------------------------------------------------
def topology(cfg):
    """Return (down, middle, up, out_in_channels).

    ``down``/``up`` are lists (one entry per ``downsampling.i`` / ``upsampling.i``) of layer lists;
    each layer is a tuple whose first item is the kind:
      ('conv', cin, cout)                      bare 3x3 conv          (model.py:367)
      ('res', cin, cout, mode)                 mode in {None,'up','down'} (model.py:374,393,420,433)
      ('attn', c)                              (model.py:381,407,426)
      ('down', cin, cout, with_conv)           (model.py:397)
      ('up', cin, cout, with_conv)             (model.py:438)
    """
    mc = cfg['model_channels']
    mult = tuple(cfg['channel_mult'])
    nrb = cfg['num_res_blocks']
    attn_res = tuple(cfg['attention_resolutions'])
    updown = cfg['resblock_updown']
    conv_rs = cfg['conv_resample']

    cur = first = int(mc * mult[0])
    res = cfg['resolution']
    down = [[('conv', cfg['in_channels'], cur)]]
    stack = [cur]
    for level, m in enumerate(mult):
        for _ in range(nrb):
            layers = [('res', cur, int(mc * m), None)]
            cur = int(mc * m)
            if res in attn_res:
                layers.append(('attn', cur))
            stack.append(cur)
            down.append(layers)
        cur = int(mc * m)
        if level != len(mult) - 1:
            if updown:
                down.append([('res', cur, cur, 'down')])
            else:
                down.append([('down', cur, cur, conv_rs)])
            stack.append(cur)
            res //= 2
    middle = [('res', cur, cur, None), ('attn', cur), ('res', cur, cur, None)]
    up = []
    for level, m in list(enumerate(mult))[::-1]:
        for i in range(nrb + 1):
            skip = stack.pop()
            layers = [('res', cur + skip, int(mc * m), None)]
            cur = int(mc * m)
            if res in attn_res:
                layers.append(('attn', cur))
            if level != 0 and i == nrb:
                if updown:
                    layers.append(('res', cur, cur, 'up'))
                else:
                    layers.append(('up', cur, cur, conv_rs))
                res *= 2
            up.append(layers)
    return down, middle, up, first, cur


def timestep_embedding(t, dim, max_period=10000):
    """model.py:514-523 -- cos half first, then sin half; zero-pad odd dims."""
    half = dim // 2
    freqs = torch.exp(torch.arange(half, dtype=torch.float32) * -(math.log(max_period) / half))
    arg = t[:, None].float() * freqs[None]
    emb = torch.cat([torch.cos(arg), torch.sin(arg)], dim=1)
    if dim % 2 == 1:
        emb = F.pad(emb, (0, 1, 0, 0))
    return emb


def _gn(x, sd, p):
    return F.group_norm(x, GN_GROUPS, sd[p + '.weight'], sd[p + '.bias'], GN_EPS)


def _resample(x, mode):
    if mode == 'up':      # model.py:77
        return F.interpolate(x, scale_factor=2.0, mode='nearest')
    if mode == 'down':    # model.py:111
        return F.avg_pool2d(x, kernel_size=(2, 2), stride=(2, 2))
    return x


def res_block(sd, p, x, emb, mode, adaptive):
    """model.py:188-211."""
    h = F.silu(_gn(x, sd, p + '.in_norm'))
    if mode is not None:
        h = _resample(h, mode)
        x = _resample(x, mode)
    h = F.conv2d(h, sd[p + '.in_conv.weight'], sd[p + '.in_conv.bias'], padding=1)
    e = F.linear(F.silu(emb), sd[p + '.step_embedding.weight'], sd[p + '.step_embedding.bias'])[:, :, None, None]
    if adaptive:
        scale, shift = torch.chunk(e, 2, dim=1)
        h = _gn(h, sd, p + '.out_norm') * (1 + scale) + shift
    else:
        h = _gn(h + e, sd, p + '.out_norm')
    h = F.silu(h)
    h = F.conv2d(h, sd[p + '.out_conv.weight'], sd[p + '.out_conv.bias'], padding=1)
    if (p + '.skip.weight') in sd:
        w = sd[p + '.skip.weight']
        x = F.conv2d(x, w, sd[p + '.skip.bias'], padding=w.shape[-1] // 2)
    return h + x


def attn_heads(c, cfg):
    """model.py:236-242."""
    if cfg['num_head_channels'] is None:
        return cfg['num_heads']
    assert c % cfg['num_head_channels'] == 0
    return c // cfg['num_head_channels']


def attn_block(sd, p, x, cfg, return_parts=False):
    """model.py:260-291 (both qkv channel orders)."""
    B, C, H, W = x.shape
    T = H * W
    nh = attn_heads(C, cfg)
    hd = C // nh
    scale = (C // nh) ** -0.5
    x = x.reshape(B, C, T)
    n = F.group_norm(x, GN_GROUPS, sd[p + '.norm.weight'], sd[p + '.norm.bias'], GN_EPS)
    qkv = F.conv1d(n, sd[p + '.qkv_nin.weight'], sd[p + '.qkv_nin.bias'])       # [B, 3C, T]
    if cfg['split_qkv_first']:
        # channel = which*C + head*hd + d
        q, k, v = qkv.reshape(B, 3, nh, hd, T).unbind(1)                       # each [B, nh, hd, T]
    else:
        # channel = head*3hd + which*hd + d
        q, k, v = qkv.reshape(B, nh, 3, hd, T).unbind(2)
    w = torch.einsum('bhdt,bhds->bhts', q, k) * scale
    w = torch.softmax(w, dim=-1)
    h = torch.einsum('bhts,bhds->bhdt', w, v).reshape(B, C, T)
    out = F.conv1d(h, sd[p + '.proj_out.weight'], sd[p + '.proj_out.bias'])
    res = (out + x).reshape(B, C, H, W)
    if return_parts:
        return res, dict(norm=n, qkv=qkv, attn=h)
    return res


def _run_layers(sd, prefix, layers, x, emb, cfg, taps=None):
    for j, layer in enumerate(layers):
        p = '{}.{}'.format(prefix, j)
        kind = layer[0]
        if kind == 'conv':
            x = F.conv2d(x, sd[p + '.weight'], sd[p + '.bias'], padding=1)
        elif kind == 'res':
            x = res_block(sd, p, x, emb, layer[3], cfg['use_adaptive_gn'])
        elif kind == 'attn':
            x = attn_block(sd, p, x, cfg)
        elif kind == 'down':        # model.py:107-112
            if layer[3]:
                x = F.conv2d(x, sd[p + '.conv.weight'], sd[p + '.conv.bias'], stride=2, padding=1)
            else:
                x = F.avg_pool2d(x, kernel_size=(2, 2), stride=(2, 2))
        elif kind == 'up':          # model.py:76-80
            x = F.interpolate(x, scale_factor=2.0, mode='nearest')
            if layer[3]:
                x = F.conv2d(x, sd[p + '.conv.weight'], sd[p + '.conv.bias'], padding=1)
        else:
            raise ValueError(kind)
        if taps is not None:
            taps[p] = x
    return x


def embed(sd, cfg, t, y=None):
    """model.py:456-459."""
    e = timestep_embedding(t, cfg['model_channels'])
    e = F.linear(e, sd['step_embed.0.weight'], sd['step_embed.0.bias'])
    e = F.linear(F.silu(e), sd['step_embed.2.weight'], sd['step_embed.2.bias'])
    if cfg.get('num_classes') is not None:
        e = e + sd['class_embedding.weight'][y]
    return e


@torch.no_grad()
def unet_forward(sd, cfg, x, t, y=None, taps=None):
    """model.py:451-476.  ``x`` [B,C,R,R] fp32 NCHW, ``t`` int64 [B] (original-scale index), ``y`` int64 [B]|None."""
    cfg = default_cfg(**cfg)
    assert (y is not None) == (cfg['num_classes'] is not None), 'pass y iff class-conditional model'
    assert x.shape[2] == cfg['resolution'] and x.shape[3] == cfg['resolution']
    down, middle, up, first, last = topology(cfg)
    emb = embed(sd, cfg, t, y)
    if taps is not None:
        taps['emb'] = emb
    xs = []
    for i, layers in enumerate(down):
        x = _run_layers(sd, 'downsampling.{}'.format(i), layers, x, emb, cfg, taps)
        xs.append(x)
    x = _run_layers(sd, 'middle_block', middle, x, emb, cfg, taps)
    for i, layers in enumerate(up):
        x = torch.cat([x, xs.pop()], dim=1)
        x = _run_layers(sd, 'upsampling.{}'.format(i), layers, x, emb, cfg, taps)
    x = F.silu(F.group_norm(x, GN_GROUPS, sd['out.0.weight'], sd['out.0.bias'], GN_EPS))
    return F.conv2d(x, sd['out.2.weight'], sd['out.2.bias'], padding=1)


# ----------------------------------------------------------------------------------------------------------------
# synthetic, torch-version-independent weights (SURVEY.md 8(c) rule (i), 8(d) "Synthetic inputs")
# ----------------------------------------------------------------------------------------------------------------
def param_shapes(cfg):
    """Ordered {key: shape} in the reference's state_dict order (module registration order, model.py:345-449)."""
    import collections
    cfg = default_cfg(**cfg)
    mc = cfg['model_channels']
    ed = 4 * mc
    out = collections.OrderedDict()

    def conv(p, cin, cout, k):
        out[p + '.weight'] = (cout, cin, k, k)
        out[p + '.bias'] = (cout,)

    def lin(p, cin, cout):
        out[p + '.weight'] = (cout, cin)
        out[p + '.bias'] = (cout,)

    def norm(p, c):
        out[p + '.weight'] = (c,)
        out[p + '.bias'] = (c,)

    def res(p, cin, cout):
        # registration order in ResidualBlock.__init__ (model.py:150-183): skip, in_norm, in_conv, out_norm,
        # out_conv, step_embedding
        if cin != cout:
            conv(p + '.skip', cin, cout, 1)
        norm(p + '.in_norm', cin)
        conv(p + '.in_conv', cin, cout, 3)
        norm(p + '.out_norm', cout)
        conv(p + '.out_conv', cout, cout, 3)
        lin(p + '.step_embedding', ed, 2 * cout if cfg['use_adaptive_gn'] else cout)

    def attn(p, c):
        # model.py:247-254: qkv_nin, norm, proj_out
        out[p + '.qkv_nin.weight'] = (3 * c, c, 1)
        out[p + '.qkv_nin.bias'] = (3 * c,)
        norm(p + '.norm', c)
        out[p + '.proj_out.weight'] = (c, c, 1)
        out[p + '.proj_out.bias'] = (c,)

    def layers_(prefix, layers):
        for j, layer in enumerate(layers):
            p = '{}.{}'.format(prefix, j)
            if layer[0] == 'conv':
                conv(p, layer[1], layer[2], 3)
            elif layer[0] == 'res':
                res(p, layer[1], layer[2])
            elif layer[0] == 'attn':
                attn(p, layer[1])
            elif layer[0] in ('down', 'up') and layer[3]:
                conv(p + '.conv', layer[1], layer[2], 3)

    lin('step_embed.0', mc, ed)
    lin('step_embed.2', ed, ed)
    if cfg['num_classes'] is not None:
        out['class_embedding.weight'] = (cfg['num_classes'], ed)
    down, middle, up, first, last = topology(cfg)
    for i, layers in enumerate(down):
        layers_('downsampling.{}'.format(i), layers)
    layers_('middle_block', middle)
    for i, layers in enumerate(up):
        layers_('upsampling.{}'.format(i), layers)
    norm('out.0', last)
    conv('out.2', first, cfg['out_channels'], 3)
    return out


def is_zero_init(key):
    """Parameters the reference zero-initialises (model.py:177,253,448)."""
    return ('.out_conv.' in key) or ('.proj_out.' in key) or key.startswith('out.2.')


def synth_state_dict(cfg, seed=1234, sigma=0.02, sigma_zero=0.005):
    """numpy default_rng weights in state_dict key order: sigma for ordinary tensors, sigma_zero for the
    reference's zero-initialised tensors, GroupNorm weight = 1 + sigma*n."""
    import collections
    import numpy as np
    rng = np.random.default_rng(seed)
    sd = collections.OrderedDict()
    for key, shape in param_shapes(cfg).items():
        n = rng.standard_normal(shape).astype(np.float32)
        is_norm_w = key.endswith('norm.weight') or key == 'out.0.weight'
        if is_norm_w:
            v = 1.0 + sigma * n
        elif is_zero_init(key):
            v = sigma_zero * n
        else:
            v = sigma * n
        sd[key] = torch.from_numpy(np.ascontiguousarray(v.astype(np.float32)))
    return sd
