"""Chain sharding over the GPUs of one node: one process per GPU
(torch.distributed; backend "nccl" is RCCL over xGMI on ROCm).

Chains are independent, so the chain axis is split across ranks with NO
data-path collective.  The only coupling in the reference is through global
adaptation statistics:
  * tf.reduce_mean(acceptance_rate) feeding dual averaging (hmc.py:377) and
    the step-size search (hmc.py:326)      -> all-reduce(sum) of 1 double;
  * the chain-axis means of the EWMV mass estimator (hmc.py:138,143)
                                           -> all-reduce(sum) of 2*D doubles.
Every rank then applies the identical update to its replicated (epsilon,
tuner, mass) state, so no broadcast is needed.  Random numbers are keyed by
the GLOBAL chain index, so results do not depend on the number of ranks.
"""
import torch
import torch.distributed as dist

__all__ = ['ChainSharding', 'shard_bounds']


def shard_bounds(n_chains_global, rank, world_size):
    """[lo, hi) of the chains owned by `rank`: even split, the first
    n % world ranks own one extra chain."""
    base, rem = divmod(int(n_chains_global), int(world_size))
    lo = rank * base + min(rank, rem)
    return lo, lo + base + (1 if rank < rem else 0)


class ChainSharding(object):
    """Describes how this process's chains sit in the global chain axis and
    performs the (tiny) adaptation all-reduces on a process group."""

    def __init__(self, process_group=None, chain_offset=None,
                 n_chains_global=None):
        if not dist.is_initialized():
            raise RuntimeError("torch.distributed is not initialised")
        self.group = process_group
        self.rank = dist.get_rank(process_group)
        self.world_size = dist.get_world_size(process_group)
        self._chain_offset = chain_offset
        self._n_chains_global = n_chains_global

    def layout(self, n_local, device):
        """(chain_offset, n_chains_global) for a shard of `n_local` chains.
        Unless given explicitly, computed with one all-gather of the local
        counts at plan-build time (not on the hot loop)."""
        if self._chain_offset is not None and self._n_chains_global is not None:
            return int(self._chain_offset), int(self._n_chains_global)
        if dist.get_backend(self.group) == 'gloo':
            device = torch.device('cpu')
        mine = torch.tensor([int(n_local)], dtype=torch.int64, device=device)
        counts = [torch.zeros_like(mine) for _ in range(self.world_size)]
        dist.all_gather(counts, mine, group=self.group)
        counts = [int(c.item()) for c in counts]
        return sum(counts[:self.rank]), sum(counts)

    def all_reduce_sum(self, tensor):
        """In-place sum over ranks.  With the RCCL ("nccl") backend the
        collective is enqueued on the device, ordered with the current
        stream; with gloo (CPU tests, or two ranks sharing one GPU in the
        functional test) device tensors are staged through the host."""
        if self.world_size > 1:
            if tensor.is_cuda and dist.get_backend(self.group) == 'gloo':
                host = tensor.cpu()
                dist.all_reduce(host, op=dist.ReduceOp.SUM, group=self.group)
                tensor.copy_(host)
            else:
                dist.all_reduce(tensor, op=dist.ReduceOp.SUM,
                                group=self.group)
        return tensor
