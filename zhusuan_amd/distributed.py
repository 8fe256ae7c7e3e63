"""Chain sharding over the GPUs of one node: one process per GPU; the data
path's one collective goes straight to RCCL over xGMI through the C-ABI
(zshmc_comm_*, csrc/comm.hip), torch.distributed only bootstraps it.

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
import ctypes

import torch
import torch.distributed as dist

from . import _capi

__all__ = ['ChainSharding', 'shard_bounds']


def shard_bounds(n_chains_global, rank, world_size):
    """[lo, hi) of the chains owned by `rank`: even split, the first
    n % world ranks own one extra chain."""
    base, rem = divmod(int(n_chains_global), int(world_size))
    lo = rank * base + min(rank, rem)
    return lo, lo + base + (1 if rank < rem else 0)


class ChainSharding(object):
    """Describes how this process's chains sit in the global chain axis and
    performs the (tiny) adaptation all-reduces.

    backend='rccl' : a communicator of libzshmc.so (ncclCommInitRank; the
        unique id is broadcast over the torch.distributed group, which may be
        a CPU/gloo group), collectives enqueued on the current HIP stream.
        This is the production path: one process per GPU, each with its
        device selected (torch.cuda.set_device) BEFORE the communicator is
        made -- ncclCommInitRank binds to the calling thread's device.  Hosts
        whose driver only supports dmabuf IPC need HSA_ENABLE_IPC_MODE_LEGACY=0
        in the environment before the first HIP call (bench.py sets it).
    backend='torch': torch.distributed collectives on `process_group` -- the
        gloo world_size-2 CPU tests, and the functional test that lets two
        ranks share one GPU (RCCL refuses two ranks on one device)."""

    def __init__(self, process_group=None, chain_offset=None,
                 n_chains_global=None, backend='torch', always_reduce=False):
        if not dist.is_initialized():
            raise RuntimeError("torch.distributed is not initialised")
        if backend not in ('torch', 'rccl'):
            raise ValueError("backend must be 'torch' or 'rccl'")
        self.group = process_group
        self.rank = dist.get_rank(process_group)
        self.world_size = dist.get_world_size(process_group)
        self._chain_offset = chain_offset
        self._n_chains_global = n_chains_global
        self.backend = backend
        # a one-rank communicator normally skips its collectives; tests set
        # this to run the sharded code path (and RCCL) on a 1-GPU box
        self.always_reduce = bool(always_reduce)
        self._comm = None
        if backend == 'rccl':
            self._comm = self._create_rccl_comm()

    def _create_rccl_comm(self):
        ident = [None]
        if self.rank == 0:
            buf = ctypes.create_string_buffer(_capi.COMM_ID_BYTES)
            _capi.call('zshmc_comm_unique_id', buf)
            ident = [buf.raw]
        src = 0 if self.group is None else dist.get_global_rank(self.group, 0)
        dist.broadcast_object_list(ident, src=src, group=self.group)
        comm = ctypes.c_void_p()
        _capi.call('zshmc_comm_create', ident[0], self.rank, self.world_size,
                   ctypes.byref(comm))
        return comm

    @property
    def active(self):
        """Whether the adaptation statistics have to cross ranks."""
        return self.world_size > 1 or self.always_reduce

    @property
    def rccl_ranks(self):
        """Ranks of the direct RCCL communicator (0: torch backend)."""
        if self._comm is None:
            return 0
        return int(_capi.load().zshmc_comm_world_size(self._comm))

    def relayout(self, chain_offset=None, n_chains_global=None):
        """The same communicator for another sampler whose chains sit
        differently in the global chain axis (None, None: derived from the
        ranks' local counts at plan build).  The view does not own the
        communicator: close the original."""
        import copy
        view = copy.copy(self)
        view._chain_offset = chain_offset
        view._n_chains_global = n_chains_global
        view._owner = False
        return view

    def close(self):
        if not getattr(self, '_owner', True):
            return
        if self._comm is not None:
            _capi.call('zshmc_comm_destroy', self._comm)
            self._comm = None

    def layout(self, n_local, device):
        """(chain_offset, n_chains_global) for a shard of `n_local` chains.
        Unless given explicitly, computed with one all-gather of the local
        counts at plan-build time (not on the hot loop)."""
        if self._chain_offset is not None and self._n_chains_global is not None:
            return int(self._chain_offset), int(self._n_chains_global)
        counts = [None] * self.world_size
        dist.all_gather_object(counts, int(n_local), group=self.group)
        return sum(counts[:self.rank]), sum(counts)

    def all_reduce_sum(self, tensor):
        """In-place sum of a contiguous float64 tensor over the ranks.
        'rccl': ncclAllReduce enqueued on the current stream (device tensors
        only).  'torch': dist.all_reduce; with gloo, device tensors are staged
        through the host."""
        if self.backend == 'rccl':
            if not (tensor.is_cuda and tensor.dtype == torch.float64 and
                    tensor.is_contiguous()):
                raise ValueError("rccl all-reduce: contiguous float64 device "
                                 "tensor expected")
            _capi.call('zshmc_comm_all_reduce_sum', self._comm,
                       tensor.data_ptr(), tensor.numel(),
                       _capi.current_stream())
            return tensor
        if self.world_size > 1:
            if tensor.is_cuda and dist.get_backend(self.group) == 'gloo':
                host = tensor.cpu()
                dist.all_reduce(host, op=dist.ReduceOp.SUM, group=self.group)
                tensor.copy_(host)
            else:
                dist.all_reduce(tensor, op=dist.ReduceOp.SUM,
                                group=self.group)
        return tensor
