"""Preset (model, diffusion) argument dictionaries, selected by a substring of ``--model_path``.

Data contract of the reference's ``nicediffusion/default_args.py:4-43``: the values must be identical because real
checkpoints are built against them (pinned by tests/golden/cli_dicts.json).
"""


def _diffusion(beta_schedule, use_ddim, guidance_method):
    return {'rescaled_num_steps': 25, 'original_num_steps': 1000, 'use_ddim': use_ddim, 'ddim_eta': 0.0,
            'beta_schedule': beta_schedule, 'sampling_var_type': 'learned_interpolation', 'classifier': None,
            'guidance_method': guidance_method, 'guidance_strength': 0.8, 'loss_type': 'hybrid'}


def _model(resolution, channel_mult, model_channels, num_res_blocks, in_channels, num_classes, heads):
    d = {'resolution': resolution, 'attention_resolutions': None, 'channel_mult': channel_mult}
    d.update(heads)
    d.update({'in_channels': in_channels, 'out_channels': 2 * in_channels, 'model_channels': model_channels,
              'num_res_blocks': num_res_blocks, 'split_qkv_first': True, 'dropout': 0.05,
              'resblock_updown': True, 'use_adaptive_gn': True, 'num_classes': num_classes})
    return d


# EMNIST letters, 28x28 grayscale, trained with classifier-free guidance (27 = 26 letters + null class)
EMNIST_DIFFUSION_ARGS = _diffusion('cosine', False, 'classifier_free')
EMNIST_MODEL_ARGS = _model(28, (1, 2, 4), 64, 2, 1, 27, {'num_heads': 4})
EMNIST_MODEL_ARGS['attention_resolutions'] = (7, 14)

# OpenAI guided-diffusion ImageNet checkpoints
OPENAI_64_DIFFUSION_ARGS = _diffusion('cosine', True, None)
OPENAI_64_MODEL_ARGS = _model(64, (1, 2, 3, 4), 192, 3, 3, 1000, {'num_head_channels': 64})
OPENAI_64_MODEL_ARGS['attention_resolutions'] = (8, 16, 32)

OPENAI_128_DIFFUSION_ARGS = _diffusion('linear', True, None)
OPENAI_128_MODEL_ARGS = _model(128, (1, 1, 2, 3, 4), 256, 2, 3, 1000, {'num_heads': 4})
OPENAI_128_MODEL_ARGS['attention_resolutions'] = (8, 16, 32)

OPENAI_256_DIFFUSION_ARGS = _diffusion('linear', True, None)
OPENAI_256_MODEL_ARGS = _model(256, (1, 1, 2, 2, 4, 4), 256, 2, 3, 1000, {'num_head_channels': 64})
OPENAI_256_MODEL_ARGS['attention_resolutions'] = (8, 16, 32)

PRESETS = (('64x64', OPENAI_64_MODEL_ARGS, OPENAI_64_DIFFUSION_ARGS),
           ('128x128', OPENAI_128_MODEL_ARGS, OPENAI_128_DIFFUSION_ARGS),
           ('256x256', OPENAI_256_MODEL_ARGS, OPENAI_256_DIFFUSION_ARGS),
           ('EMNIST', EMNIST_MODEL_ARGS, EMNIST_DIFFUSION_ARGS))
