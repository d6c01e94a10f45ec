"""Multi-GPU form of the matvec: the target-leaf list is cut into contiguous shards, one per rank
(one process per GPU); every rank builds the same tree, owns a slice of the near blocks and of the
M2L/L2L/L2P work, and produces a result vector that is zero outside its rows.  An all-reduce(sum) of the
N-vector over RCCL/xGMI makes the full result available on every rank (what GMRES needs for its next
Arnoldi step).  The upward pass is either repeated by every rank (SURVEY.md section 8e as written: one
collective per matvec) or -- shard_upward, the default -- computed by the owners of the boxes and shared
with ONE all-gather of the multipoles (60 MB at N = 1M, p = 10) in front of M2L, overlapped with the near field.

The reference has no distributed code at all; this is the design of section 8(e).
"""
import os

import torch
import torch.distributed as dist

from .plan import FMM_plan


class ShardedFMM:
    """FMM_plan sharded over the ranks of a torch.distributed process group.

    local_execute: callable(x) -> partial result (zeros outside the owned rows). Defaults to the HIP
    plan's execute_torch; tests inject a CPU stand-in to exercise the partition + collective under gloo.
    local_split: (doubles_per_rank, upward(x, send), downward(recv, y)) stand-ins for the split execute, same purpose.
    """

    def __init__(self, K, panels, opts=None, bc=None, p_max=None, group=None, device=None,
                 host_only=False, local_execute=None, shard_upward=None, local_split=None):
        self.group = group
        self.rank = dist.get_rank(group) if dist.is_initialized() else 0
        self.world = dist.get_world_size(group) if dist.is_initialized() else 1
        if device is None:
            device = torch.cuda.current_device() if torch.cuda.is_available() else 0
        if shard_upward is None:
            shard_upward = os.environ.get("FMMBEM_SHARD_UPWARD", "1") != "0"
        # the owner-sharded upward pass carries at most 8 shards (DevicePlan::xch_ptr): larger groups repeat the upward pass
        self.split = (bool(shard_upward) and 1 < self.world <= 8 and local_execute is None) or local_split is not None
        self._split_fns = local_split
        self.plan = FMM_plan(K, panels, opts, bc=bc, p_max=p_max, device=device,
                             shard=(self.rank, self.world), host_only=host_only,
                             shard_upward=self.split and self.world > 1)
        self.n = self.plan.n
        self._local = local_execute if local_execute is not None else self.plan.execute_torch
        self._xbuf = {}                                   # p -> (send, recv) exchange buffers
        # overlap of the all-gather with the near field: RCCL only (an asynchronous gloo collective on device tensors
        # goes through the host and was measured 20x slower in a one-GPU rehearsal).  Off unless FMMBEM_OVERLAP_GATHER=1:
        # the stream ordering of this path has not run on a multi-GPU node yet (SCALE_r01 was skipped), and the first
        # such run should be a measurement of the plain path, not a debugging session.
        self._overlap = (self.split and dist.is_initialized() and dist.get_backend(group) == "nccl"
                         and os.environ.get("FMMBEM_OVERLAP_GATHER", "0") == "1")

    def kernel(self):
        return self.plan.kernel()

    def options(self):
        return self.plan.options()

    def owned_rows(self):
        """Original-order indices of the result rows this rank computes."""
        s = self.plan.stats()
        return self.plan.perm()[s["owned_row_begin"]:s["owned_row_end"]]

    def execute(self, x, out=None):
        """x: full charge vector, replicated on every rank (torch tensor). Returns the full result."""
        if not self.split:
            y = self._local(x) if out is None else self._local(x, out=out)
        else:
            p = self.plan.kernel().P
            if p not in self._xbuf:
                per = self._split_fns[0] if self._split_fns else self.plan.exchange_doubles(p)
                self._xbuf[p] = (torch.empty(per, dtype=torch.float64, device=x.device),
                                 torch.empty(per * self.world, dtype=torch.float64, device=x.device))
            send, recv = self._xbuf[p]
            y = torch.empty_like(x) if out is None else out
            if self._split_fns:
                self._split_fns[1](x, send)
                dist.all_gather_into_tensor(recv, send, group=self.group)
                self._split_fns[2](recv, y)
            else:
                stream = torch.cuda.current_stream(x.device).cuda_stream
                self.plan.upward_device(x.data_ptr(), send.data_ptr(), stream, p)
                if self._overlap:
                    # the near field (HBM-bound, needs no multipoles) streams while the multipoles travel over xGMI
                    work = dist.all_gather_into_tensor(recv, send, group=self.group, async_op=True)
                    self.plan.near_split_device(y.data_ptr(), stream)
                    work.wait()                           # RCCL: the current stream waits for the collective, the host does not
                else:
                    dist.all_gather_into_tensor(recv, send, group=self.group)
                self.plan.downward_device(recv.data_ptr(), y.data_ptr(), stream, p)
        if self.world > 1:
            dist.all_reduce(y, op=dist.ReduceOp.SUM, group=self.group)
        return y
