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


# Flag tables: (flags, value type | 'flag', default, who must give it, help).  "who": 'always' | 'train' (required when
# training, optional when sampling) | None (never required).  Names, types and defaults are those of the reference's
# parsers (utils.py:28-142); the help texts are this build's.
_SAMPLE_FLAGS = (
    (('--model_path',), str, None, 'always', 'state-dict file of the model'),
    (('-c', '--custom'), 'flag', False, None, 'take the architecture / diffusion settings from the flags, not a preset'),
    (('--batch_size',), int, None, 'always', 'images per batch'),
    (('--num_samples',), int, None, 'always', 'number of batches (total images = num_samples * batch_size)'),
    (('--upsample',), 'flag', False, None, '4x Real-ESRGAN super-resolution of the results (needs basicsr)'),
    (('--wordy', '-w'), 'flag', False, None, 'print progress'),
    (('--save_path',), str, None, None, 'directory prefix for the JPGs; without it the images are displayed'),
    (('--labels',), str, '', None, 'class labels, one per batch, separated by "/"; random when omitted'),
    (('--start_img',), str, None, None, 'image to noise and then denoise (img2img); default is pure noise'),
    (('--steps_to_do',), int, None, None, 'how many (original-scale) steps of noise to apply to start_img'),
    (('--seed',), int, None, None, 'RNG seed'),
    (('--cpu',), 'flag', False, None, 'reference flag; this build has no CPU sampling path and refuses it'),
)
_TRAIN_FLAGS = (
    (('--batch_size',), int, None, 'always', 'images per batch'),
    (('--lr',), float, None, 'always', 'learning rate'),
    (('--weight_decay',), float, None, 'always', 'weight decay'),
    (('--iterations',), int, None, 'always', 'training iterations'),
    (('--resume_step',), int, 0, None, 'checkpoint step'),
    (('--wordy', '-w'), 'flag', False, None, 'print progress'),
    (('--save_every',), int, None, None, 'checkpoint period'),
    (('--sample_every',), int, None, None, 'sampling period'),
    (('--ema_rate',), float, 0.9999, None, 'EMA rate'),
    (('--use_fp16',), 'flag', False, None, 'unused'),
    (('--grad_accumulation',), int, 1, None, 'optimizer step period'),
    (('--seed',), int, None, None, 'RNG seed'),
)
_MODEL_FLAGS = (
    (('--resolution',), int, None, 'train', 'image height = width'),
    (('--model_channels',), int, None, 'train', 'base channel count'),
    (('--channel_mult',), str, None, 'train', 'per-level channel multipliers, "/"-separated'),
    (('--num_res_blocks',), int, None, 'train', 'residual blocks per level'),
    (('--attention_resolutions',), str, None, 'train', 'resolutions that get attention, "/"-separated'),
    (('--num_classes',), int, None, None, 'class count of a conditional model'),
    (('--dropout',), float, 0.0, 'train', 'dropout probability'),
    (('--in_channels',), int, 3, None, 'image channels'),
    (('--num_heads',), int, 4, None, 'attention heads'),
    (('--num_head_channels',), int, None, None, 'channels per attention head (overrides num_heads)'),
    (('--split_qkv_first',), 'flag', False, None, 'qkv channel order: q|k|v blocks first, heads inside'),
    (('--resblock_updown',), 'flag', False, None, 'resample inside residual blocks'),
    (('--use_adaptive_gn',), 'flag', False, None, 'adaptive GroupNorm (scale/shift from the timestep embedding)'),
)
_DIFFUSION_FLAGS = (
    (('--rescaled_num_steps',), int, None, 'train', 'number of sampling steps'),
    (('--beta_schedule',), str, None, 'train', "'linear', 'cosine' or 'constant'"),
    (('--sampling_var_type',), str, None, 'train', "'small', 'large', 'learned' or 'learned_interpolation'"),
    (('--use_ddim',), 'flag', False, None, 'DDIM sampler'),
    (('--ddim_eta',), float, 0.0, None, 'DDIM eta'),
    (('--original_num_steps',), int, 1000, None, 'steps the model was trained with'),
    (('--loss_type',), str, 'hybrid', 'train', "'simple', 'KL', 'KL_rescaled' or 'hybrid'"),
    (('--guidance_method',), str, None, None, "'classifier' or 'classifier_free'"),
    (('--guidance_strength',), float, None, None, 'guidance weight'),
    (('--classifier_path',), str, None, None, 'classifier state dict (classifier guidance is not implemented)'),
)


def _add_flags(group, table, training):
    for flags, kind, default, who, text in table:
        needed = who == 'always' or (who == 'train' and training)
        if kind == 'flag':
            group.add_argument(*flags, action='store_true', default=default, required=False, help=text)
        else:
            group.add_argument(*flags, type=kind, default=default, required=needed,
                               metavar='(required)' if needed else '(optional)', help=text)


def make_argparser(prog):
    """Argument parser for ``'diff_sample'`` (and, for surface compatibility, ``'diff_train'``)."""
    if prog not in ('diff_sample', 'diff_train'):
        raise NotImplementedError(prog)
    training = prog == 'diff_train'
    p = argparse.ArgumentParser(prog=prog, description=(
        'Train a diffusion model (not part of this build; parser kept for compatibility).' if training
        else 'Draw images from a diffusion model on an AMD Instinct GPU.'))
    if training:
        _add_flags(p.add_argument_group('training arguments', 'training loop settings'), _TRAIN_FLAGS, True)
    else:
        _add_flags(p.add_argument_group('sampling arguments', 'what to sample and where to put it'), _SAMPLE_FLAGS, False)
    _add_flags(p.add_argument_group('model arguments', 'UNet architecture (only read with --custom when sampling)'),
               _MODEL_FLAGS, training)
    _add_flags(p.add_argument_group('diffusion arguments', 'noise schedule and sampler'), _DIFFUSION_FLAGS, training)
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
