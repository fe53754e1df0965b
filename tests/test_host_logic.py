"""Host-side logic of the product package (runs without a GPU): schedules, CLI glue, state_dict contract,
default-init parity, C-ABI symbol table, fail-loud behaviour on CPU tensors."""
import json
import os
import re
import subprocess

import numpy as np
import pytest
import torch

from nicediffusion import _hip
from nicediffusion import default_args as DA
from nicediffusion.diffusion import Diffusion, get_beta_schedule, VarType
from nicediffusion.model import DiffusionModel
from nicediffusion.utils import make_argparser, get_dicts_from_args, convert_state_dict
from nicediffusion.parallel import shard_slice
from tests.cases import TINY_CFGS, SCHEDULE_CASES, CLI_CASES

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
CPU = torch.device('cpu')


def tiny_model():
    return DiffusionModel(resolution=8, in_channels=1, model_channels=32, out_channels=1, num_res_blocks=1,
                          attention_resolutions=(), channel_mult=(1,))


@pytest.mark.parametrize('name', sorted(SCHEDULE_CASES))
def test_product_schedule_tables_bit_exact(golden_dir, name):
    g = np.load(os.path.join(golden_dir, 'schedules.npz'))
    T, S, sched = SCHEDULE_CASES[name]
    d = Diffusion(tiny_model(), T, S, 'small', 'simple', beta_schedule=sched, device=CPU)
    for attr in ('betas', 'alphas_cumprod', 'alphas_cumprod_prev', 'sqrt_alphas_cumprod',
                 'sqrt_one_minus_alphas_cumprod', 'sqrt_reciprocal_alphas_cumprod',
                 'sqrt_reciprocal_alphas_minus_one_cumprod', 'posterior_mean_coef_x0', 'posterior_mean_coef_xt',
                 'posterior_variance', 'log_posterior_var_clipped'):
        assert np.array_equal(getattr(d, attr), g['{}/{}'.format(name, attr)]), (name, attr)
    assert np.array_equal(d.timestep_map.numpy(), g['{}/timestep_map'.format(name)])
    assert d.timestep_map.dtype == torch.long


def test_coefficient_table_columns():
    d = Diffusion(tiny_model(), 1000, 250, 'learned_interpolation', 'hybrid', beta_schedule='cosine', use_ddim=True,
                  ddim_eta=0.0, device=CPU)
    tab = d.coefficient_table()
    assert tab.shape == (250, 8) and tab.dtype == torch.float32
    assert np.array_equal(tab[:, 0].numpy(), d.sqrt_reciprocal_alphas_cumprod.astype(np.float32))
    assert np.array_equal(tab[:, 3].numpy(), d.alphas_cumprod_prev.astype(np.float32))
    assert np.array_equal(tab[:, 6].numpy(), d.log_posterior_var_clipped.astype(np.float32))
    assert np.array_equal(tab[:, 7].numpy(), np.log(d.betas).astype(np.float32))
    d = Diffusion(tiny_model(), 1000, 50, 'large', 'simple', device=CPU)
    assert np.array_equal(d.coefficient_table()[:, 6].numpy(),
                          np.log(np.append(d.posterior_variance[1], d.betas[1:])).astype(np.float32))
    d = Diffusion(tiny_model(), 1000, 50, 'small', 'simple', device=CPU)
    assert np.array_equal(d.coefficient_table()[:, 6].numpy(),
                          np.log(np.maximum(d.posterior_variance, 1e-20)).astype(np.float32))


def test_constructor_errors():
    m = tiny_model()
    with pytest.raises(NotImplementedError):
        Diffusion(m, 1000, 10, 'small', 'simple', guidance_method='bogus', device=CPU)
    with pytest.raises(AssertionError):
        Diffusion(m, 1000, 10, 'small', 'simple', guidance_method='classifier_free', guidance_strength=1.0, device=CPU)
    with pytest.raises(AssertionError):
        Diffusion(m, 1000, 10, 'small', 'simple', use_ddim=True, device=CPU)
    with pytest.raises(NotImplementedError):
        Diffusion(m, 1000, 10, 'tiny', 'simple', device=CPU)
    with pytest.raises(NotImplementedError):
        Diffusion(m, 1000, 10, 'small', 'L7', device=CPU)
    with pytest.raises(NotImplementedError):
        get_beta_schedule('quadratic', 10, 1e-4, 2e-2)
    with pytest.raises(AssertionError):
        Diffusion(m, 1000, 10, 'small', 'simple', betas=[0.1] * 5, device=CPU)
    assert VarType.get_var_type('learned') is VarType.LEARNED


def test_no_cpu_fallback():
    """CPU tensors must fail loudly, never silently run somewhere else."""
    m = tiny_model()
    with pytest.raises(_hip.NdHipError):
        m(torch.zeros(1, 1, 8, 8), torch.zeros(1, dtype=torch.long))
    d = Diffusion(m, 1000, 10, 'small', 'simple', device=CPU)
    with pytest.raises(_hip.NdHipError):
        d.denoise(x=torch.zeros(1, 1, 8, 8), batch_size=1, progress=False)
    with pytest.raises(AssertionError):      # label iff conditional (diffusion.py:179)
        d.denoise(x=torch.zeros(1, 1, 8, 8), kwargs={'y': torch.zeros(1, dtype=torch.long)}, batch_size=1)
    with pytest.raises(AssertionError):      # model.py:452-454
        m(torch.zeros(1, 1, 8, 8), torch.zeros(1, dtype=torch.long), y=torch.zeros(1, dtype=torch.long))
    with pytest.raises(AssertionError):
        m(torch.zeros(1, 1, 16, 16), torch.zeros(1, dtype=torch.long))
    # the public per-step methods (diffusion.py:232-369) have no CPU path either
    t = torch.zeros(1)
    for call in (lambda: d.ddim_denoising_step(torch.zeros(1, 1, 8, 8), t), lambda: d.denoising_step(torch.zeros(1, 1, 8, 8), t),
                 lambda: d.get_eps_and_log_var(torch.zeros(1, 1, 8, 8), t, {}), lambda: d.diffusion_step(torch.zeros(1, 1, 8, 8), t)):
        with pytest.raises(_hip.NdHipError):
            call()
    # ... and the reference's surface is all there, with its signatures (diffusion.py:232,242,266,318)
    import inspect
    for name, params in (('diffusion_step', ['x_0', 't', 'noise']), ('get_eps_and_log_var', ['x_t', 't', 'kwargs']),
                         ('denoising_step', ['x_t', 't', 'kwargs', 'clip_x']), ('ddim_denoising_step', ['x_t', 't', 'kwargs', 'clip_x'])):
        got = list(inspect.signature(getattr(Diffusion, name)).parameters)[1:]
        assert got[:len(params)] == params, (name, got)
    assert inspect.signature(Diffusion.denoising_step).parameters['clip_x'].default is True


def test_product_never_imports_oracle():
    pkg = os.path.join(ROOT, 'nice-diffusion_amd')
    for dirpath, _, files in os.walk(pkg):
        for f in files:
            if f.endswith('.py'):
                src = open(os.path.join(dirpath, f)).read()
                assert not re.search(r'^\s*(from|import)\s+oracle', src, re.M), f


# ------------------------------------------------------------------------------------------------- CLI (A12)
@pytest.mark.parametrize('name', sorted(CLI_CASES))
def test_cli_dicts_match_reference(golden_dir, name):
    ref = json.load(open(os.path.join(golden_dir, 'cli_dicts.json')))[name]
    args = make_argparser('diff_sample').parse_args(CLI_CASES[name])
    other, margs, dargs = get_dicts_from_args(args)

    def norm(d):
        return json.loads(json.dumps(d, default=lambda o: list(o)))
    assert norm(margs) == ref['model']
    assert norm(dargs) == ref['diff']
    assert norm(other) == ref['other']


def test_cli_errors():
    p = make_argparser('diff_sample')
    with pytest.raises(NotImplementedError):
        get_dicts_from_args(p.parse_args(['--model_path', 'foo.pt', '--batch_size', '1', '--num_samples', '1']))
    with pytest.raises(Exception):
        get_dicts_from_args(p.parse_args(['--model_path', 'foo.pt', '-c', '--batch_size', '1', '--num_samples', '1']))
    with pytest.raises(NotImplementedError):
        make_argparser('nope')
    make_argparser('diff_train')


def test_presets_values():
    assert DA.OPENAI_64_MODEL_ARGS == {'resolution': 64, 'attention_resolutions': (8, 16, 32),
                                      'channel_mult': (1, 2, 3, 4), 'num_head_channels': 64, 'in_channels': 3,
                                      'out_channels': 6, 'model_channels': 192, 'num_res_blocks': 3,
                                      'split_qkv_first': True, 'dropout': 0.05, 'resblock_updown': True,
                                      'use_adaptive_gn': True, 'num_classes': 1000}
    assert DA.EMNIST_DIFFUSION_ARGS['guidance_method'] == 'classifier_free'
    assert DA.OPENAI_128_MODEL_ARGS['num_heads'] == 4 and DA.OPENAI_256_MODEL_ARGS['channel_mult'] == (1, 1, 2, 2, 4, 4)


# ------------------------------------------------------------------------------------------------- state_dict (A10)
@pytest.mark.parametrize('pname,margs', [('EMNIST', DA.EMNIST_MODEL_ARGS), ('OPENAI_64', DA.OPENAI_64_MODEL_ARGS)])
def test_state_dict_contract(golden_dir, pname, margs):
    meta = json.load(open(os.path.join(golden_dir, 'preset_state_dicts.json')))[pname]
    with torch.device('meta'):
        m = DiffusionModel(**margs)
    sd = m.state_dict()
    assert list(sd.keys()) == meta['keys']
    assert [list(v.shape) for v in sd.values()] == meta['shapes']
    assert m.conditional and m.in_channels == margs['in_channels'] and m.resolution == margs['resolution']
    assert m.num_classes == margs['num_classes'] and m.model_channels == margs['model_channels']


@pytest.mark.parametrize('name', sorted(TINY_CFGS))
def test_tiny_state_dict_loads_strict(name):
    from oracle import unet_oracle as UO      # tests may use the oracle's key table
    cfg = TINY_CFGS[name]
    m = DiffusionModel(**cfg)
    m.load_state_dict(UO.synth_state_dict(cfg), strict=True)


def test_default_init_matches_reference_rng_stream(golden_dir):
    g = np.load(os.path.join(golden_dir, 'init_seed0.npz'))
    torch.manual_seed(0)
    m = DiffusionModel(**TINY_CFGS['adagn_updown'])
    sd = m.state_dict()
    for k in g.files:
        assert np.array_equal(sd[k].numpy(), g[k]), k
    # the zero-initialised tensors of model.py:177,253,448
    assert not sd['downsampling.1.0.out_conv.weight'].any() and not sd['out.2.weight'].any()
    assert not sd['middle_block.1.proj_out.weight'].any()


def test_convert_state_dict_names():
    sd = {'input_blocks.1.0.in_layers.0.weight': 1, 'output_blocks.2.1.qkv.bias': 2, 'time_embed.0.weight': 3,
          'label_emb.weight': 4, 'middle_block.0.emb_layers.1.bias': 5, 'input_blocks.4.0.out_layers.3.weight': 6,
          'output_blocks.0.0.skip_connection.weight': 7, 'out.2.bias': 8, 'input_blocks.1.0.out_layers.0.bias': 9,
          'input_blocks.1.0.in_layers.2.bias': 10}
    out = convert_state_dict(dict(sd))
    assert list(out.keys()) == ['downsampling.1.0.in_norm.weight', 'upsampling.2.1.qkv_nin.bias', 'step_embed.0.weight',
                                'class_embedding.weight', 'middle_block.0.step_embedding.bias',
                                'downsampling.4.0.out_conv.weight', 'upsampling.0.0.skip.weight', 'out.2.bias',
                                'downsampling.1.0.out_norm.bias', 'downsampling.1.0.in_conv.bias']
    assert list(out.values()) == list(sd.values())


def test_convert_state_dict_whole_preset_key_tables(golden_dir):
    """Every key of the four presets' state dicts (the REFERENCE's own key tables, preset_state_dicts.json), written with
    openai/guided-diffusion names, comes back under the reference's name, in order, values untouched (utils.py:265-292)."""
    import json as _json
    meta = _json.load(open(os.path.join(golden_dir, 'preset_state_dicts.json')))
    from tests.test_gpu_model import to_openai_names
    for pname, rec in meta.items():
        sd = {k: i for i, k in enumerate(rec['keys'])}
        oa = to_openai_names(sd)
        assert len(oa) == len(sd) and set(oa).isdisjoint(k for k in sd if not k.startswith(('middle_block', 'out.')))
        assert all(k.split('.')[0] in ('input_blocks', 'middle_block', 'output_blocks', 'time_embed', 'label_emb', 'out')
                   for k in oa), pname
        back = convert_state_dict(oa)
        assert list(back.keys()) == rec['keys'] and list(back.values()) == list(range(len(sd))), pname
        assert list(oa.values()) == list(range(len(sd)))          # input untouched


def _built_library_and_llvm_tools():
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    need = [os.path.join(root, 'nice-diffusion_amd', 'nicediffusion', 'libnd_hip.so'), '/opt/rocm/lib/llvm/bin/clang-offload-bundler',
            '/opt/rocm/lib/llvm/bin/llvm-readelf']
    import shutil
    return all(os.path.exists(p) for p in need) and shutil.which('objcopy') is not None


@pytest.mark.skipif(not _built_library_and_llvm_tools(), reason='needs the built libnd_hip.so, objcopy and the ROCm LLVM tools')
def test_hand_scheduled_kernels_have_no_scratch_and_fit_two_waves_per_simd():
    """gemm4_kernel, gemm_bf16q_kernel, conv_wino4_kernel and conv_wf4_kernel issue their run-ahead loads as inline ISA whose
    destination registers are "in flight" until a hand-counted s_waitcnt: a spill or a copy of such a register by a future
    compiler / flag change would read it before the data has landed (the pattern behind the run-to-run race fixed in round
    3).  The built library's own metadata must show no scratch, no spills and <= 256 VGPRs (two blocks of 4 waves per CU)
    for every instantiation of the first three and <= 168 (three waves per SIMD: one 12-wave workgroup per CU)
    for conv_wf4_kernel."""
    import importlib.util
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    spec = importlib.util.spec_from_file_location('kernel_regs', os.path.join(root, 'tools', 'kernel_regs.py'))
    kr = importlib.util.module_from_spec(spec)
    spec.loader.exec_module(kr)
    tab = kr.kernel_table(os.path.join(root, 'nice-diffusion_amd', 'nicediffusion', 'libnd_hip.so'))
    assert len(tab) > 100
    seen = set()
    for name, r in tab.items():
        for k in ('gemm4_kernel', 'gemm_bf16q_kernel', 'conv_wino4_kernel', 'conv_wf4_kernel'):
            if 'nd::' + k in name:
                seen.add(k)
                assert r['scratch'] == 0 and r['sgpr_spill'] == 0 and r['vgpr_spill'] == 0, (name, r)
                assert r['vgpr'] + r['agpr'] <= (168 if k == 'conv_wf4_kernel' else 256), (name, r)
    assert seen == {'gemm4_kernel', 'gemm_bf16q_kernel', 'conv_wino4_kernel', 'conv_wf4_kernel'}, seen


def _kernel_regs_tool():
    import importlib.util
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    spec = importlib.util.spec_from_file_location('kernel_regs', os.path.join(root, 'tools', 'kernel_regs.py'))
    kr = importlib.util.module_from_spec(spec)
    spec.loader.exec_module(kr)
    return kr, os.path.join(root, 'nice-diffusion_amd', 'nicediffusion', 'libnd_hip.so')


def test_hand_counted_waits_see_exactly_the_loads_they_count():
    """conv_wf4_kernel and gemm4_kernel wait with LITERAL vmcnt / lgkmcnt values (wf4_younger, vmcnt(25 - NDMA - LDEARLY),
    LPS + NDMA in gemm4's residual seeding) that assume the compiler puts no other memory operation into the main loop.  The
    machine code of the built library is disassembled (tools/kernel_regs.py loop_table) and every loop that holds MFMAs must
    contain exactly the operations the counts were written for: conv_wf4_kernel -- a loop body = two 16-channel chunks = 144
    MFMAs, 48 fragment loads (global_load_dwordx3: 6 positions x 4 k4-steps x 2 chunks), 2 x NDMA halo DMAs (buffer_load ... lds:
    NDMA = 2 for the 16x16-pixel geometry, 3 for the four-image one), two barriers and NOTHING else that counts on vmcnt or
    lgkmcnt besides its ds_read_b64 (no other load, no store, no atomic, no scalar load, no LDS write); gemm4_kernel -- per
    k-step 8 weight-fragment loads (16 with the GroupNorm coefficients), all global_load_dwordx4, pixel rows by LDS-DMA only.
    A compiler or flag change that rematerialises an argument or reloads a descriptor inside these loops fails here instead
    of reading registers before their data has landed."""
    kr, so = _kernel_regs_tool()
    wf4 = kr.loop_table(so, 'conv_wf4_kernel')
    assert len(wf4) == 8, sorted(wf4)
    for name, loops in wf4.items():
        gw = int(re.search(r'conv_wf4_kernel<(\d+)', name).group(1))
        main = [lp for lp in loops if any(k.startswith('v_mfma') for k in lp['counts'])]
        assert len(main) == 2, (name, loops)                 # waves of transform rows 1-4 and of rows 0 / 5
        for lp in main:
            c = dict(lp['counts'])
            assert c.pop('v_mfma_f32_16x16x4_f32') == 144, (name, lp)
            assert c.pop('global_load_dwordx3') == 48, (name, lp)
            assert c.pop('buffer_load_dwordx4_lds') == 2 * (2 if gw == 5 else 3), (name, lp)
            assert c.pop('s_barrier') == 2, (name, lp)
            assert c.pop('ds_read_b64') in (72, 96), (name, lp)
            c.pop('s_waitcnt')
            assert not c, (name, 'unexpected memory operations in the chunk loop', c)
    g4 = kr.loop_table(so, 'gemm4_kernel')
    assert len(g4) >= 6, sorted(g4)
    for name, loops in g4.items():
        gn = 'gemm4_kernel<true' in name
        main = [lp for lp in loops if any(k.startswith('v_mfma') for k in lp['counts'])]
        assert main, name
        for lp in main:
            c = dict(lp['counts'])
            assert c.pop('global_load_dwordx4') == (16 if gn else 8), (name, lp)
            for k in ('v_mfma_f32_32x32x2_f32', 'buffer_load_dwordx4_lds', 'ds_read_b128', 's_barrier', 's_waitcnt'):
                assert c.pop(k, 0) > 0, (name, k, lp)
            assert not c, (name, 'unexpected memory operations in the k loop', c)


def test_conv_wf4_lds_halo_image_is_what_the_reads_expect():
    """conv_wf4_kernel fills its halo buffers by LDS-DMA (one 16-byte unit per lane and round, a 4-bit XOR key per 4-pixel group)
    and reads them back with per-lane addresses built from four keys: tools/wf4_lds_image.py restates both index maps and checks
    every (tile, patch element, k group) of both block geometries against the unit the DMA put there."""
    import subprocess
    import sys
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    out = subprocess.run([sys.executable, os.path.join(root, 'tools', 'wf4_lds_image.py')], capture_output=True, text=True)
    assert out.returncode == 0, out.stderr
    assert out.stdout.count('image consistent') == 2 and 'False' not in out.stdout, out.stdout


def test_bench_weights_are_the_survey_recipe_the_parity_tests_use():
    """bench.py's synthetic weights (numpy default_rng(1234) in state_dict order, 0.02 / 0.005 / 1 + 0.02 n) are
    bit for bit the oracle's synth_state_dict(seed=1234): the full-size GPU parity tests run the very model BENCH times."""
    import bench
    from nicediffusion import default_args as DA
    from oracle import unet_oracle as UO
    cfg = dict(DA.EMNIST_MODEL_ARGS)
    m = DiffusionModel(**cfg)
    bench.synthetic_weights(m)
    sd = UO.synth_state_dict(cfg, seed=1234)
    got = m.state_dict()
    assert list(got) == list(sd) and all(torch.equal(sd[k], v) for k, v in got.items())


def test_shipped_library_is_the_product_build_of_the_sources_in_the_tree():
    """nd_build_id() of the library in the tree is the SHA-256 the Makefile computes over the sources in the tree (a library
    left over from an earlier state of csrc/ fails here), nd_build_flags() is empty (no timing-only / diagnostic macro was
    defined for any translation unit), csrc/nd_variant_flags.inc lists every macro the sources test, and the stamp of every
    measured artefact starts with the build id."""
    from nicediffusion import _engine
    lib = _hip.load()
    assert lib.nd_build_flags() == b''
    assert re.fullmatch(r'[0-9a-f]{16}', lib.nd_build_id().decode())
    assert lib.nd_build_id().decode() == _hip.source_hash(), 'libnd_hip.so was not built from the sources in the tree: make -C nice-diffusion_amd'
    assert _hip.build_id() == _hip.source_hash()
    assert _engine._tune_stamp().startswith('b' + _hip.source_hash() + ':v')
    import importlib.util
    spec = importlib.util.spec_from_file_location('gen_variant_flags', os.path.join(ROOT, 'tools', 'gen_variant_flags.py'))
    gv = importlib.util.module_from_spec(spec)
    spec.loader.exec_module(gv)
    names = gv.scan()
    assert len(names) >= 40 and 'ND_F4ABL_NOEPI' in names and 'ND_BF_DIAG' in names
    assert open(gv.OUT).read() == gv.render(names), 'run python tools/gen_variant_flags.py'


def test_an_ablation_build_is_refused_by_the_loader(monkeypatch):
    """_hip.load() must not take a library that reports variant flags (simulated here on the loaded handle) unless
    ND_ALLOW_ABLATION=1; with it, the flags become part of build_id() and therefore of every stamp."""
    import ctypes
    real = _hip.load()

    class Fake:
        def __getattr__(self, k):
            return getattr(real, k)

    fake = Fake()
    fake.__dict__['nd_build_flags'] = lambda: b'ND_F4ABL_NOEPI=1 ND_F4_DIAG=1'
    monkeypatch.setattr(_hip, '_LIB', None)
    monkeypatch.setattr(ctypes, 'CDLL', lambda path: fake)
    monkeypatch.delenv('ND_ALLOW_ABLATION', raising=False)
    with pytest.raises(_hip.NdHipError, match='ND_F4ABL_NOEPI'):
        _hip.load()
    monkeypatch.setenv('ND_ALLOW_ABLATION', '1')
    monkeypatch.setattr(_hip, '_LIB', None)
    assert _hip.load() is fake
    assert _hip.build_id() == _hip.source_hash() + '+ND_F4ABL_NOEPI=1,ND_F4_DIAG=1'
    monkeypatch.setattr(_hip, '_LIB', real)


@pytest.mark.skipif(not os.path.exists(os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), 'nice-diffusion_amd',
                                                   'nicediffusion', 'libnd_hip.so')), reason='needs the built libnd_hip.so')
def test_committed_tune_caches_match_the_built_library():
    """profiles/tune_cache_<workload>.json (the kernel choices bench.py and the full-size tests build their plans from) carry
    the stamp of THIS library build -- the hash of its sources (nd_build_id), version, variant tables -- so a kernel change
    that did not regenerate them (tools/tune_all.sh on an MI355X) is flagged here instead of silently falling back to tuning on
    the box: editing one character of any .hip fails this test until the caches are regenerated.  Every choice names an
    existing variant."""
    from nicediffusion import _engine
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    lib = _hip.load()
    for wl in ('config1', 'config2', 'config4', 'config5', 'config4_fp32', 'config5_fp32'):
        path = os.path.join(root, 'profiles', 'tune_cache_{}.json'.format(wl))
        raw = json.load(open(path))
        assert raw.pop('__stamp__') == _engine._tune_stamp(), wl
        assert len(raw) >= 20, (wl, len(raw))
        for k, v in raw.items():
            kind, var = v[0], v[1]
            if kind.startswith('bf16'):
                assert 0 <= var < lib.nd_conv_bf16_num_variants() and not lib.nd_conv_bf16_variant_name(var).startswith(b'(retired)')
            elif kind in ('wino', 'wino+splitk'):
                assert 0 <= var < lib.nd_conv_winograd_num_variants() and \
                    not lib.nd_conv_winograd_variant_name(var).startswith(b'(retired)')
            elif kind in ('wf4', 'wf4+splitk'):
                assert 0 <= var < lib.nd_conv_winograd_f4_num_variants(), (k, v)
            else:
                assert kind in ('direct', 'direct+splitk') and 0 <= var < lib.nd_conv_num_variants(), (k, v)
        saved = dict(_engine._TUNED)
        try:
            _engine._TUNED.clear()
            assert _engine.preload_tune_cache(path, device_index=5) == len(raw)
            assert all(key[0] == 5 for key in _engine._TUNED)
        finally:
            _engine._TUNED.clear()
            _engine._TUNED.update(saved)


# ------------------------------------------------------------------------------------------------- sharding
def test_shard_slice_partitions():
    for n in (1, 7, 64, 512, 513):
        for world in (1, 2, 3, 8):
            rows = []
            for r in range(world):
                s = shard_slice(n, r, world)
                rows.extend(range(s.start, s.stop))
            assert rows == list(range(n))
    assert shard_slice(512, 3, 8) == slice(192, 256)
    with pytest.raises(ValueError):
        shard_slice(8, 8, 8)


# ------------------------------------------------------------------------------------------------- C ABI
def _header_functions():
    src = open(os.path.join(ROOT, 'include', 'nd_hip.h')).read()
    src = re.sub(r'/\*.*?\*/', '', src, flags=re.S)
    return sorted(set(re.findall(r'\b(nd_[a-z0-9_]+)\s*\(', src)))


def test_library_exports_every_header_symbol():
    names = _header_functions()
    assert len(names) >= 20
    assert sorted(_hip.EXPORTS) == names, 'ctypes table out of sync with include/nd_hip.h'
    path = _hip.lib_path()
    assert os.path.exists(path), 'libnd_hip.so not built (run __graft_entry__.build())'
    lib = _hip.load()                                   # loads without a GPU; no compute calls here
    for n in names:
        assert hasattr(lib, n), n
    assert lib.nd_version() >= 100
    assert lib.nd_conv_num_variants() >= 4
    assert lib.nd_conv_weight_floats(192, 192, 3) == (6 + 1) * 6 * 9 * 4 * 256
    out = subprocess.run(['nm', '-D', '--defined-only', path], capture_output=True, text=True).stdout
    exported = set(re.findall(r' T (nd_[a-z0-9_]+)', out))
    assert exported == set(names)


def test_weight_read_ahead_stays_inside_the_packed_tensors():
    """Every kernel that streams packed weights keeps fragment loads in flight past the last chunk it consumes; the packers
    append zero chunks for that.  csrc/nd_weight_stream.h states the contract (static_asserts in every kernel
    instantiation); here the per-variant bound nd_conv*_max_weight_read is checked against the allocation size for every
    variant over a grid of shapes, split-K's shifted pointers included.  (Round 2: a 16x16x32 bf16 1x1 stream read one
    fragment past a single padding chunk and aborted the process.)"""
    lib = _hip.load()
    Ns = (1, 6, 32, 33, 96, 192, 200, 768, 1152, 2304)
    Cs = (4, 32, 36, 64, 70, 96, 128, 192, 200, 576, 960, 1344, 1536, 2048)
    for N in Ns:
        for C in Cs:
            for k in (1, 3):
                size = lib.nd_conv_weight_floats(N, C, k)
                for v in range(-1, lib.nd_conv_num_variants()):
                    r = lib.nd_conv_max_weight_read(v, N, C, k)
                    assert 0 < r <= size, ('fp32', v, N, C, k, r, size)
                size = lib.nd_conv_bf16_weight_elems(N, C, k)
                for v in range(-1, lib.nd_conv_bf16_num_variants()):
                    for splits in (1, 2, 3, 4, 8):
                        r = lib.nd_conv_bf16_max_weight_read(v, N, C, k, splits)
                        assert 0 < r <= size, ('bf16', v, N, C, k, splits, r, size)
            size = lib.nd_conv_winograd_weight_floats(N, C)
            for v in range(-1, lib.nd_conv_winograd_num_variants()):
                r = lib.nd_conv_winograd_max_weight_read(v, N, C)
                assert 0 < r <= size, ('wino', v, N, C, r, size)
            for v in range(lib.nd_conv_winograd_f4_num_variants()):
                size = lib.nd_conv_winograd_f4_weight_floats(v, N, C)
                r = lib.nd_conv_winograd_f4_max_weight_read(v, N, C)
                assert 0 < r <= size, ('wf4', v, N, C, r, size)
    # the bound is tight where it matters: the two-fragments-per-chunk 1x1 stream of the 16x16x32 layout needs both padding chunks
    names = [lib.nd_conv_bf16_variant_name(v) for v in range(lib.nd_conv_bf16_num_variants())]
    vs = [v for v, n in enumerate(names) if n == b'nd::conv_bf16s_kernel']
    assert vs and any(lib.nd_conv_bf16_max_weight_read(v, 64, 128, 1, 1) == lib.nd_conv_bf16_weight_elems(64, 128, 1) for v in vs)
    assert lib.nd_conv_max_weight_read(lib.nd_conv_num_variants(), 32, 32, 3) < 0 and lib.nd_conv_bf16_max_weight_read(0, 32, 32, 3, 0) < 0


def test_argument_validation_without_gpu():
    """Bad arguments are rejected on the host before any launch."""
    lib = _hip.load()
    rc = lib.nd_conv_nhwc(None, 32, 32, None, 0, 0, None, None, None, 0, None, 0, None, 32, 1, 8, 8, 32, 3, 0, -1, None, None, 0,
                          None)
    assert rc == -1 and 'null' in _hip.last_error()
    rc = lib.nd_attention_nhwc(16, 96, 16, 32, 1, 64, 1, 12, 0, 32, 64, 12, 1.0, None)
    assert rc == -1 and 'multiple of 8' in _hip.last_error()
    rc = lib.nd_groupnorm_stats_nhwc(16, 30, 32, None, 0, 0, None, 0, 16, 1, 4, 32, _hip.DT_F32, None)
    assert rc == -1
    rc = lib.nd_groupnorm_stats_nhwc(16, 32, 32, None, 0, 0, None, 0, 16, 1, 4, 32, 7, None)
    assert rc == -1 and 'dtype' in _hip.last_error()
    rc = lib.nd_conv_bf16_nhwc(16, 12, 16, None, 0, 0, 16, None, None, 0, None, 0, 16, 32, 1, 8, 8, 32, 3, 0, -1, None, None, 0, None)
    assert rc == -1 and 'multiples of 8' in _hip.last_error()
    rc = lib.nd_attention_bf16_nhwc(16, 96, 16, 32, 1, 64, 1, 12, 0, 32, 64, 12, 1.0, None)
    assert rc == -1 and 'multiple of 8' in _hip.last_error()
    assert lib.nd_conv_bf16_weight_elems(96, 70, 3) == (2 + 2) * 3 * 9 * 4 * 512      # 70 ch -> 2 chunks of 64 (+2 zero chunks)
    assert 1 <= lib.nd_groupnorm_stats_blocks(64, 4096, 192, _hip.DT_F32) <= 32 and lib.nd_groupnorm_stats_blocks(2, 64, 64, _hip.DT_BF16) >= 1


def test_start_image_resize_follows_cv2_linear_semantics(tmp_path):
    """scripts/sample.py --start_img: cv2.resize's default INTER_LINEAR (half-pixel centres, no antialiasing, exact-2x box
    case), restated for hosts without cv2; checked against torch's non-antialiased bilinear kernel (same sampling
    positions, float arithmetic) to within the fixed-point rounding."""
    import sys
    sys.path.insert(0, os.path.join(ROOT, 'nice-diffusion_amd', 'scripts'))
    import sample
    rng = np.random.default_rng(0)
    for (h, w, r) in ((40, 56, 16), (9, 13, 16), (16, 16, 16), (100, 37, 28)):
        img = rng.integers(0, 256, (h, w, 3), dtype=np.uint8)
        got = sample.resize_linear_u8(img, r, r)
        assert got.shape == (r, r, 3) and got.dtype == np.uint8
        ref = torch.nn.functional.interpolate(torch.from_numpy(img).permute(2, 0, 1)[None].double(), size=(r, r),
                                              mode='bilinear', align_corners=False, antialias=False)[0]
        assert np.abs(got.astype(float) - ref.permute(1, 2, 0).numpy()).max() <= 1.0
        if (h, w) == (r, r):
            assert np.array_equal(got, img)
    img = rng.integers(0, 256, (32, 32, 3), dtype=np.uint8)
    box = (img.reshape(16, 2, 16, 2, 3).astype(int).sum((1, 3)) + 2) >> 2
    assert np.array_equal(sample.resize_linear_u8(img, 16, 16), box.astype(np.uint8))
    assert (sample.resize_linear_u8(np.full((31, 17, 3), 77, np.uint8), 16, 16) == 77).all()
    from PIL import Image
    p = str(tmp_path / 'start.png')
    Image.fromarray(img).save(p)
    t = sample.load_start_image(p, 16)
    assert t.shape == (3, 16, 16) and t.dtype == torch.float32
    assert torch.equal(t, torch.from_numpy(box.astype(np.float64) / 127.5 - 1).permute(2, 0, 1).float())
