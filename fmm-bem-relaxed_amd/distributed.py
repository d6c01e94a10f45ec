"""Multi-GPU form of the matvec: the target-leaf list is cut into contiguous shards, one per rank
(one process per GPU); every rank builds the same tree, replicates the cheap upward pass, owns a
slice of the near blocks and of the M2L/L2L/L2P work, and produces a result vector that is zero
outside its rows.  ONE collective per matvec -- an all-reduce(sum) of the N-vector over RCCL/xGMI --
makes the full result available on every rank (what GMRES needs for its next Arnoldi step).

The reference has no distributed code at all; this is the design of section 8(e).
"""
import torch
import torch.distributed as dist

from .plan import FMM_plan


class ShardedFMM:
    """FMM_plan sharded over the ranks of a torch.distributed process group.

    local_execute: callable(x) -> partial result (zeros outside the owned rows). Defaults to the HIP
    plan's execute_torch; tests inject a CPU stand-in to exercise the partition + collective under gloo.
    """

    def __init__(self, K, panels, opts=None, bc=None, p_max=None, group=None, device=None,
                 host_only=False, local_execute=None):
        self.group = group
        self.rank = dist.get_rank(group) if dist.is_initialized() else 0
        self.world = dist.get_world_size(group) if dist.is_initialized() else 1
        if device is None:
            device = torch.cuda.current_device() if torch.cuda.is_available() else 0
        self.plan = FMM_plan(K, panels, opts, bc=bc, p_max=p_max, device=device,
                             shard=(self.rank, self.world), host_only=host_only)
        self.n = self.plan.n
        self._local = local_execute if local_execute is not None else self.plan.execute_torch

    def kernel(self):
        return self.plan.kernel()

    def options(self):
        return self.plan.options()

    def owned_rows(self):
        """Original-order indices of the result rows this rank computes."""
        s = self.plan.stats()
        return self.plan.perm()[s["owned_row_begin"]:s["owned_row_end"]]

    def execute(self, x):
        """x: full charge vector, replicated on every rank (torch tensor). Returns the full result."""
        y = self._local(x)
        if self.world > 1:
            dist.all_reduce(y, op=dist.ReduceOp.SUM, group=self.group)
        return y
