"""Worker of the sharded-sampling GPU tests in test_gpu_model.py: one rank of a job whose ranks share cuda:0 (rehearsal of
the one-process-per-GPU path on a one-GPU box).

    shard_worker.py RANK WORLD PORT OUT.pt [BACKEND [CASE]]

BACKEND: gloo (default; the gather is staged through host memory) or nccl (= RCCL; one rank per device, so WORLD must be
1 on a one-GPU box).  CASE: see CASES.  Rank 0 saves the gathered result to OUT.pt; every rank saves the kernel choices it
ran with (``_engine._TUNED`` without the device ordinal) to OUT.pt.rank<r>.json."""
import json
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, 'nice-diffusion_amd'))
sys.path.insert(0, ROOT)
os.environ.setdefault('HSA_ENABLE_IPC_MODE_LEGACY', '0')     # dmabuf IPC: RCCL needs it on this driver
import torch
import torch.distributed as dist

# a model above the autotuner's threshold (every 3x3 conv >= 2e8 flops at 8 forwards), so that the ranks really have measured
# choices to agree on; classifier-free guidance doubles the forward batch (plans are keyed by 2 x rows)
TUNED_CFG = dict(resolution=32, in_channels=3, model_channels=96, out_channels=6, num_res_blocks=1, attention_resolutions=(16,),
                 channel_mult=(1, 2), num_head_channels=32, num_classes=10, use_adaptive_gn=True, resblock_updown=True,
                 split_qkv_first=True)
CASES = {
    # 3 + 2 rows of a tiny model (below the tuner's threshold), DDPM with in-kernel noise
    'ragged_tiny': dict(cfg='adagn_updown', rows=5, seed=9, steps=6, guidance=None),
    # 4 + 4 rows, classifier-free guidance (8 forwards per rank and step), every conv tuned
    'tuned_cfg': dict(cfg=TUNED_CFG, rows=8, seed=9, steps=4, guidance='classifier_free'),
    # 3 + 2 rows of the tuned model: two shard sizes, both measured by rank 0
    'tuned_ragged': dict(cfg=TUNED_CFG, rows=5, seed=9, steps=3, guidance=None),
}


def build_case(case, device):
    from oracle import unet_oracle as UO
    from tests.cases import TINY_CFGS
    from nicediffusion.model import DiffusionModel
    from nicediffusion.diffusion import Diffusion
    c = CASES[case]
    cfg = dict(TINY_CFGS[c['cfg']]) if isinstance(c['cfg'], str) else dict(c['cfg'])
    sd = UO.synth_state_dict(cfg, seed=c['seed'])
    m = DiffusionModel(**cfg)
    m.load_state_dict(sd, strict=True)
    m.to(device).eval()
    d = Diffusion(m, 1000, c['steps'], 'learned_interpolation', 'hybrid', beta_schedule='cosine', use_ddim=False,
                  guidance_method=c['guidance'], guidance_strength=0.8 if c['guidance'] else None, device=torch.device(device))
    d.seed = 4242                                     # in-kernel Philox noise, same stream on every rank
    torch.manual_seed(0)
    R = cfg['resolution']
    x = torch.randn(c['rows'], 3, R, R)
    y = (torch.arange(c['rows']) * 3) % (cfg['num_classes'] - 1) + 1          # label 0 is the null class under guidance
    return m, d, x, y


def main():
    rank, world, port, out_path = int(sys.argv[1]), int(sys.argv[2]), sys.argv[3], sys.argv[4]
    backend = sys.argv[5] if len(sys.argv) > 5 else 'gloo'
    case = sys.argv[6] if len(sys.argv) > 6 else 'ragged_tiny'
    os.environ['MASTER_ADDR'] = '127.0.0.1'
    os.environ['MASTER_PORT'] = port
    torch.cuda.set_device(0)
    if backend == 'nccl':
        dist.init_process_group('nccl', rank=rank, world_size=world, device_id=torch.device('cuda:0'))
    else:
        dist.init_process_group('gloo', rank=rank, world_size=world)
    from nicediffusion import _engine
    m, d, x, y = build_case(case, 'cuda:0')
    out = d.denoise_sharded(x, kwargs={'y': y.to('cuda:0')}, progress=False)
    json.dump(sorted([json.dumps(list(k[1:])), list(v)] for k, v in _engine._TUNED.items()),
              open('{}.rank{}.json'.format(out_path, rank), 'w'))
    if rank == 0:
        torch.save(out.cpu(), out_path)
    dist.barrier()
    dist.destroy_process_group()


if __name__ == '__main__':
    main()
