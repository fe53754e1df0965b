"""Case tables shared by tools/gen_golden.py (which runs the real reference on them) and the tests."""
import torch

# --- tiny model configurations: every code path of SURVEY.md 8(c)(vi), < 1M parameters each ---------------------
TINY_CFGS = {
    # AdaGN + ResBlock up/down + split_qkv_first (the shape of all four presets)
    'adagn_updown': dict(resolution=16, in_channels=3, model_channels=32, out_channels=6, num_res_blocks=1,
                         attention_resolutions=(8,), channel_mult=(1, 2), num_classes=10, num_head_channels=32,
                         resblock_updown=True, use_adaptive_gn=True, split_qkv_first=True),
    # plain GN (+emb before norm) + strided-conv / nearest+conv resampling + legacy qkv order, unconditional
    'plain_convres_legacy': dict(resolution=16, in_channels=3, model_channels=32, out_channels=3, num_res_blocks=1,
                                 attention_resolutions=(8, 16), channel_mult=(1, 2), num_classes=None, num_heads=2,
                                 resblock_updown=False, conv_resample=True, use_adaptive_gn=False,
                                 split_qkv_first=False),
    # avg-pool / nearest resampling without conv, single head (head_dim = C)
    'pool_resample': dict(resolution=16, in_channels=3, model_channels=32, out_channels=6, num_res_blocks=2,
                          attention_resolutions=(8,), channel_mult=(1, 2), num_classes=5, num_heads=1,
                          resblock_updown=False, conv_resample=False, use_adaptive_gn=True, split_qkv_first=True),
    # odd spatial sizes 28 -> 14 -> 7 (T = 196, 49), 1-channel input, three levels
    'odd_sizes': dict(resolution=28, in_channels=1, model_channels=32, out_channels=2, num_res_blocks=1,
                      attention_resolutions=(7, 14), channel_mult=(1, 2, 4), num_classes=27, num_head_channels=32,
                      resblock_updown=True, use_adaptive_gn=True, split_qkv_first=True),
}


def labels_for(cfg, B):
    if cfg.get('num_classes') is None:
        return None
    return (torch.arange(B) * 37) % cfg['num_classes']


# --- schedule tables: (original T, rescaled S, beta schedule) ----------------------------------------------------
SCHEDULE_CASES = {}
for _sched in ('cosine', 'linear', 'constant'):
    for _S in (25, 50, 250, 1000):
        SCHEDULE_CASES['{}_{}'.format(_sched, _S)] = (1000, _S, _sched)
SCHEDULE_CASES['cosine_300_nondividing'] = (1000, 300, 'cosine')     # table length != S
SCHEDULE_CASES['linear_T4000_100'] = (4000, 100, 'linear')

# --- sampler cases (10-step loops on tiny models; teacher-forced + free-running) ---------------------------------
SAMPLER_CASES = {
    'ddim_eta0_li': dict(cfg='adagn_updown', ddim=True, eta=0.0, var='learned_interpolation', sched='cosine', S=10),
    'ddim_eta05_learned': dict(cfg='adagn_updown', ddim=True, eta=0.5, var='learned', sched='linear', S=10),
    'ddim_cfg': dict(cfg='adagn_updown', ddim=True, eta=0.0, var='learned_interpolation', sched='cosine', S=10,
                     guidance='classifier_free', w=0.8),
    'ddpm_li': dict(cfg='adagn_updown', ddim=False, var='learned_interpolation', sched='cosine', S=10),
    'ddpm_learned': dict(cfg='adagn_updown', ddim=False, var='learned', sched='linear', S=10),
    'ddpm_small': dict(cfg='plain_convres_legacy', ddim=False, var='small', sched='linear', S=10),
    'ddpm_large': dict(cfg='plain_convres_legacy', ddim=False, var='large', sched='cosine', S=10),
    'ddpm_cfg': dict(cfg='adagn_updown', ddim=False, var='learned_interpolation', sched='linear', S=10,
                     guidance='classifier_free', w=0.8),
}

# --- CLI argv -> (other, model, diffusion) dicts -----------------------------------------------------------------
CLI_CASES = {
    'preset_64': ['--model_path', 'models/64x64_diffusion.pt', '--batch_size', '4', '--num_samples', '2',
                  '--labels', '1/2', '--seed', '0', '-w'],
    'preset_emnist': ['--model_path', 'models/EMNIST_model_params.pt', '--batch_size', '4', '--num_samples', '1',
                      '--save_path', 'out/', '--cpu'],
    'preset_128': ['--model_path', 'x/128x128_diffusion.pt', '--batch_size', '1', '--num_samples', '1'],
    'preset_256': ['--model_path', '256x256_diffusion.pt', '--batch_size', '1', '--num_samples', '1',
                   '--rescaled_num_steps', '77'],
    'custom_config2': ['--model_path', 'W.pt', '--custom', '--batch_size', '64', '--num_samples', '1',
                       '--resolution', '64', '--model_channels', '192', '--channel_mult', '1/2/3/4',
                       '--num_res_blocks', '3', '--attention_resolutions', '8/16/32', '--num_classes', '1000',
                       '--num_head_channels', '64', '--split_qkv_first', '--resblock_updown', '--use_adaptive_gn',
                       '--rescaled_num_steps', '250', '--beta_schedule', 'cosine', '--sampling_var_type',
                       'learned_interpolation', '--use_ddim', '--ddim_eta', '0.0', '--seed', '0'],
    'custom_cfg_small': ['--model_path', 'W.pt', '-c', '--batch_size', '2', '--num_samples', '3',
                         '--resolution', '16', '--model_channels', '32', '--channel_mult', '1/2',
                         '--num_res_blocks', '1', '--attention_resolutions', '8', '--num_classes', '10',
                         '--rescaled_num_steps', '10', '--beta_schedule', 'linear', '--sampling_var_type', 'small',
                         '--guidance_method', 'classifier_free', '--guidance_strength', '0.8', '--labels', '1/2/3',
                         '--in_channels', '1'],
}
