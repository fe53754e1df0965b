"""Command-line / configuration glue of the sampling path.

Mirrors the surface of the reference's ``nicediffusion/utils.py`` that ``scripts/sample.py`` uses:
``make_argparser`` (utils.py:12-143), ``get_dicts_from_args`` (utils.py:146-214), ``convert_state_dict``
(utils.py:265-292) and the inference pass-through of ``checkpoint`` (utils.py:220-236).  matplotlib is imported
lazily (the reference imports it at module import time, utils.py:3).
"""
import argparse
from collections import OrderedDict

from .default_args import *  # noqa: F401,F403  (the reference re-exports the presets from utils)
from .default_args import PRESETS

MODEL_KEYS = ('resolution', 'attention_resolutions', 'channel_mult', 'num_res_blocks', 'model_channels', 'num_heads',
              'num_head_channels', 'in_channels', 'out_channels', 'split_qkv_first', 'dropout', 'resblock_updown',
              'use_adaptive_gn', 'num_classes')
DIFFUSION_KEYS = ('rescaled_num_steps', 'original_num_steps', 'use_ddim', 'ddim_eta', 'beta_schedule',
                  'sampling_var_type', 'classifier', 'guidance_method', 'guidance_strength', 'loss_type')


def make_argparser(prog):
    """Argument parser for ``'diff_sample'`` (and, for surface compatibility, ``'diff_train'``)."""
    if prog == 'diff_sample':
        sampling, about = True, 'Draw images from a diffusion model on an AMD Instinct GPU.'
    elif prog == 'diff_train':
        sampling, about = False, 'Train a diffusion model (not part of this build; parser kept for compatibility).'
    else:
        raise NotImplementedError(prog)
    req, opt = '(required)', '(optional)'
    p = argparse.ArgumentParser(prog=prog, description=about)

    if sampling:
        g = p.add_argument_group('sampling arguments', 'what to sample and where to put it')
        g.add_argument('--model_path', type=str, required=True, metavar=req, help='state-dict file of the model')
        g.add_argument('-c', '--custom', required=False, action='store_true', default=False,
                       help='take the architecture / diffusion settings from the flags instead of a preset')
        g.add_argument('--batch_size', type=int, required=True, metavar=req, help='images per batch')
        g.add_argument('--num_samples', type=int, required=True, metavar=req,
                       help='number of batches (total images = num_samples * batch_size)')
        g.add_argument('--upsample', required=False, default=False, action='store_true',
                       help='4x Real-ESRGAN super-resolution of the results (needs basicsr)')
        g.add_argument('--wordy', '-w', dest='wordy', required=False, default=False, action='store_true',
                       help='print progress')
        g.add_argument('--save_path', type=str, required=False, metavar=opt, default=None,
                       help='directory prefix for the JPGs; without it the images are displayed')
        g.add_argument('--labels', type=str, required=False, metavar=opt, default='',
                       help='class labels, one per batch, separated by "/"; random when omitted')
        g.add_argument('--start_img', type=str, required=False, metavar=opt, default=None,
                       help='image to noise and then denoise (img2img); default is pure noise')
        g.add_argument('--steps_to_do', type=int, required=False, metavar=opt, default=None,
                       help='how many (original-scale) steps of noise to apply to start_img')
        g.add_argument('--seed', type=int, required=False, metavar=opt, default=None, help='RNG seed')
        g.add_argument('--cpu', required=False, default=False, action='store_true',
                       help='reference flag; this build has no CPU sampling path and refuses it')
    else:
        g = p.add_argument_group('training arguments', 'training loop settings')
        g.add_argument('--batch_size', type=int, required=True, metavar=req, help='images per batch')
        g.add_argument('--lr', type=float, required=True, metavar=req, help='learning rate')
        g.add_argument('--weight_decay', type=float, required=True, metavar=req, help='weight decay')
        g.add_argument('--iterations', type=int, required=True, metavar=req, help='training iterations')
        g.add_argument('--resume_step', type=int, required=False, metavar=opt, default=0, help='checkpoint step')
        g.add_argument('--wordy', '-w', dest='wordy', required=False, default=False, action='store_true',
                       help='print progress')
        g.add_argument('--save_every', type=int, required=False, metavar=opt, default=None, help='checkpoint period')
        g.add_argument('--sample_every', type=int, required=False, metavar=opt, default=None, help='sampling period')
        g.add_argument('--ema_rate', type=float, required=False, metavar=opt, default=0.9999, help='EMA rate')
        g.add_argument('--use_fp16', required=False, default=False, action='store_true', help='unused')
        g.add_argument('--grad_accumulation', type=int, required=False, metavar=opt, default=1,
                       help='optimizer step period')
        g.add_argument('--seed', type=int, required=False, metavar=opt, default=None, help='RNG seed')

    need = not sampling
    mv = req if need else opt
    m = p.add_argument_group('model arguments', 'UNet architecture (only read with --custom when sampling)')
    m.add_argument('--resolution', type=int, required=need, metavar=mv, default=None, help='image height = width')
    m.add_argument('--model_channels', type=int, required=need, metavar=mv, default=None, help='base channel count')
    m.add_argument('--channel_mult', type=str, required=need, metavar=mv, default=None,
                   help='per-level channel multipliers, "/"-separated')
    m.add_argument('--num_res_blocks', type=int, required=need, metavar=mv, default=None,
                   help='residual blocks per level')
    m.add_argument('--attention_resolutions', type=str, required=need, metavar=mv, default=None,
                   help='resolutions that get attention, "/"-separated')
    m.add_argument('--num_classes', type=int, required=False, default=None, metavar=opt,
                   help='class count of a conditional model')
    m.add_argument('--dropout', type=float, required=need, default=0.0, metavar=mv, help='dropout probability')
    m.add_argument('--in_channels', type=int, required=False, default=3, metavar=opt, help='image channels')
    m.add_argument('--num_heads', type=int, required=False, default=4, metavar=opt, help='attention heads')
    m.add_argument('--num_head_channels', type=int, required=False, default=None, metavar=opt,
                   help='channels per attention head (overrides num_heads)')
    m.add_argument('--split_qkv_first', required=False, default=False, action='store_true',
                   help='qkv channel order: q|k|v blocks first, heads inside')
    m.add_argument('--resblock_updown', required=False, default=False, action='store_true',
                   help='resample inside residual blocks')
    m.add_argument('--use_adaptive_gn', required=False, default=False, action='store_true',
                   help='adaptive GroupNorm (scale/shift from the timestep embedding)')

    d = p.add_argument_group('diffusion arguments', 'noise schedule and sampler')
    d.add_argument('--rescaled_num_steps', type=int, required=need, metavar=mv, default=None,
                   help='number of sampling steps')
    d.add_argument('--beta_schedule', type=str, required=need, metavar=mv, default=None,
                   help="'linear', 'cosine' or 'constant'")
    d.add_argument('--sampling_var_type', type=str, required=need, metavar=mv, default=None,
                   help="'small', 'large', 'learned' or 'learned_interpolation'")
    d.add_argument('--use_ddim', required=False, default=False, action='store_true', help='DDIM sampler')
    d.add_argument('--ddim_eta', type=float, required=False, default=0.0, metavar=opt, help='DDIM eta')
    d.add_argument('--original_num_steps', type=int, required=False, default=1000, metavar=opt,
                   help='steps the model was trained with')
    d.add_argument('--loss_type', type=str, required=need, default='hybrid', metavar=opt if sampling else req,
                   help="'simple', 'KL', 'KL_rescaled' or 'hybrid'")
    d.add_argument('--guidance_method', type=str, required=False, default=None, metavar=opt,
                   help="'classifier' or 'classifier_free'")
    d.add_argument('--guidance_strength', type=float, required=False, default=None, metavar=opt,
                   help='guidance weight')
    d.add_argument('--classifier_path', metavar=opt, type=str, required=False, default=None,
                   help='classifier state dict (classifier guidance is not implemented)')
    return p


def _split_ints(text):
    return [int(tok) for tok in text.split('/')]


def get_dicts_from_args(args):
    """Namespace -> (other_args, model_args, diffusion_args), with the reference's preset / fix-up rules."""
    model_args, diff_args, other_args = {}, {}, {}
    for key, val in vars(args).items():
        (model_args if key in MODEL_KEYS else diff_args if key in DIFFUSION_KEYS else other_args)[key] = val

    assert diff_args['guidance_method'] is None or model_args['num_classes'] is not None, \
        'use guidance only for conditional models'
    assert (diff_args['guidance_method'] == 'classifier') == (other_args['classifier_path'] is not None)
    if other_args['classifier_path'] is not None:
        raise NotImplementedError('classifier guidance needs a noisy classifier, which does not exist')

    have_labels = 'labels' in other_args and len(other_args['labels']) > 0
    if 'custom' in other_args:                       # sampling mode
        if other_args['custom']:
            must = [model_args[k] for k in ('resolution', 'model_channels', 'channel_mult', 'num_res_blocks',
                                           'attention_resolutions')]
            must += [diff_args[k] for k in ('rescaled_num_steps', 'sampling_var_type', 'beta_schedule')]
            if not all(must):
                raise ValueError('--custom needs every architecture and diffusion flag')
        else:
            path = other_args['model_path']
            for tag, m, d in PRESETS:
                if tag in path:
                    model_args.update(m)
                    diff_args.update(d)
                    break
            else:
                raise NotImplementedError(path, 'this is not a default model')
            if have_labels:
                other_args['labels'] = _split_ints(other_args['labels'])
            return other_args, model_args, diff_args

    if have_labels:
        other_args['labels'] = _split_ints(other_args['labels'])
    model_args['attention_resolutions'] = _split_ints(model_args['attention_resolutions'])
    model_args['channel_mult'] = _split_ints(model_args['channel_mult'])
    learned = diff_args['sampling_var_type'] in ('learned', 'learned_interpolation')
    model_args['out_channels'] = model_args['in_channels'] * (2 if learned else 1)
    if diff_args['guidance_method'] == 'classifier_free':
        model_args['num_classes'] += 1               # extra null class
    return other_args, model_args, diff_args


def checkpoint(module, inputs, parameters, use_grad_checkpoints):
    """Inference pass-through of the reference's gradient checkpointing (utils.py:220-250): just call."""
    return module(*inputs)


_OPENAI_RENAMES = (('input_blocks', 'downsampling'), ('output_blocks', 'upsampling'), ('in_layers.0', 'in_norm'),
                   ('in_layers.2', 'in_conv'), ('emb_layers.1', 'step_embedding'), ('out_layers.0', 'out_norm'),
                   ('out_layers.3', 'out_conv'), ('skip_connection', 'skip'), ('time_embed', 'step_embed'),
                   ('qkv', 'qkv_nin'), ('label_emb', 'class_embedding'))


def convert_state_dict(sd):
    """openai/guided-diffusion parameter names -> this model's names; returns a new OrderedDict, input untouched."""
    out = OrderedDict()
    for key, val in sd.items():
        for old, new in _OPENAI_RENAMES:
            key = key.replace(old, new)
        out[key] = val
    return out


def imshow(img, title=None, colormap=None):
    import matplotlib.pyplot as plt
    import numpy as np
    plt.imshow(img.astype(np.uint8), cmap=colormap)
    if title is not None:
        plt.title(title)
    plt.pause(0.001)


def cycle(iterable):
    while True:
        for item in iterable:
            yield item


def override(fn):
    return fn
