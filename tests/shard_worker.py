"""Worker of test_gpu_model.py::test_sharded_denoise_two_ranks_one_gpu: one rank of a 2-rank gloo job whose ranks share
cuda:0 (rehearsal of the one-process-per-GPU path on a one-GPU box).  Usage: shard_worker.py RANK WORLD PORT OUT.pt"""
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, 'nice-diffusion_amd'))
sys.path.insert(0, ROOT)
import torch
import torch.distributed as dist


def main():
    rank, world, port, out_path = int(sys.argv[1]), int(sys.argv[2]), sys.argv[3], sys.argv[4]
    os.environ['MASTER_ADDR'] = '127.0.0.1'
    os.environ['MASTER_PORT'] = port
    dist.init_process_group('gloo', rank=rank, world_size=world)
    from oracle import unet_oracle as UO
    from tests.cases import TINY_CFGS
    from nicediffusion.model import DiffusionModel
    from nicediffusion.diffusion import Diffusion
    cfg = dict(TINY_CFGS['adagn_updown'])
    sd = UO.synth_state_dict(cfg, seed=9)
    m = DiffusionModel(**cfg)
    m.load_state_dict(sd, strict=True)
    m.to('cuda:0').eval()
    d = Diffusion(m, 1000, 6, 'learned_interpolation', 'hybrid', beta_schedule='cosine', use_ddim=False,
                  device=torch.device('cuda:0'))
    d.seed = 4242                                     # in-kernel Philox noise, same stream on every rank
    torch.manual_seed(0)
    x = torch.randn(5, 3, 16, 16)                     # ragged split: 3 + 2 rows
    y = (torch.arange(5) * 3) % 10
    out = d.denoise_sharded(x, kwargs={'y': y.to('cuda:0')}, progress=False)
    if rank == 0:
        torch.save(out.cpu(), out_path)
    dist.barrier()
    dist.destroy_process_group()


if __name__ == '__main__':
    main()
