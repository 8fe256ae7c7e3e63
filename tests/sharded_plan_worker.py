"""One rank of tests/test_gpu_two_rank_plans.py (launched by
torch.distributed.run; both ranks on cuda:0, gloo rendezvous)."""
import os
import sys

import numpy as np
import torch
import torch.distributed as dist

HERE = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, os.path.dirname(HERE))
sys.path.insert(0, HERE)

import zhusuan_amd as zs                                  # noqa: E402
from zhusuan_amd.distributed import ChainSharding, shard_bounds   # noqa: E402
import helpers_sharded_cases as cases                     # noqa: E402


def main(out_dir):
    rank, world = int(os.environ['RANK']), int(os.environ['WORLD_SIZE'])
    dist.init_process_group('gloo', rank=rank, world_size=world)
    dev = torch.device('cuda', 0)
    out = {}
    for family in ('lntm', 'blr', 'blrb'):
        prob = getattr(cases, family + '_problem')()
        lo, hi = shard_bounds(prob['q0'].shape[0], rank, world)
        for native in (True, False):
            for adapt in (False, True):
                r = cases.run(zs, torch, dev, family, lo, hi, adapt,
                              ChainSharding(), native, rank0_reads=adapt,
                              rank=rank)
                for k, v in r.items():
                    out['%s/%d/%d/%s' % (family, native, adapt, k)] = v
    np.savez(os.path.join(out_dir, 'rank%d.npz' % rank), **out)
    dist.barrier()
    dist.destroy_process_group()


if __name__ == '__main__':
    main(sys.argv[1])
