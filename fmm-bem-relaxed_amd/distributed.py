"""Multi-GPU form of the matvec: the target-leaf list is cut into contiguous shards, one per rank
(one process per GPU); every rank builds the same tree, owns a slice of the near blocks and of the
M2L/L2L/L2P work, and computes the result rows of its own leaves.  The rows of different shards are disjoint and,
in tree order, contiguous, so the result is replicated by ONE all-gather of the shards' slices over RCCL/xGMI
(N * dof * 8 bytes in total; what GMRES needs for its next Arnoldi step) followed by the un-permute to panel order --
or, FMMBEM_Y_COLLECTIVE=allreduce, by the all-reduce(sum) of zero-padded vectors that BASELINE.json's north_star words
(twice the bytes; the two produce the same bits).  The upward pass is either repeated by every rank (SURVEY.md section 8e
as written: one collective per matvec) or -- shard_upward, the default -- computed by the owners of the boxes and shared
with ONE collective in front of M2L: by default an all-to-all in which every shard receives only the multipoles its own lists
read (2-3 MB per shard at N = 1M, p = 10 on 8 shards; FMMBEM_XCH=allgather: all 60 MB to everybody).

The reference has no distributed code at all; this is the design of section 8(e).
"""
import os
import time

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
                 host_only=False, local_execute=None, shard_upward=None, local_split=None, y_collective=None):
        self.group = group
        self.rank = dist.get_rank(group) if dist.is_initialized() else 0
        self.world = dist.get_world_size(group) if dist.is_initialized() else 1
        if device is None:
            device = torch.cuda.current_device() if torch.cuda.is_available() else 0
        if shard_upward is None:
            shard_upward = os.environ.get("FMMBEM_SHARD_UPWARD", "1") != "0"
        # how the multipoles travel when the upward pass is sharded: "alltoall" (default) = every shard sends every other only
        # what that shard's lists read (2-3 thousand boxes per shard at N = 1M on 8 shards), "allgather" = everything to everybody
        self.xch = os.environ.get("FMMBEM_XCH", "alltoall")
        # the owner-sharded upward pass carries at most 8 shards (DevicePlan::xch_ptr): larger groups repeat the upward pass
        self.split = (bool(shard_upward) and 1 < self.world <= 8 and local_execute is None) or local_split is not None
        self._split_fns = local_split
        self.plan = FMM_plan(K, panels, opts, bc=bc, p_max=p_max, device=device,
                             shard=(self.rank, self.world), host_only=host_only,
                             shard_upward=(2 if self.xch == "alltoall" and local_split is None else True) if (self.split and self.world > 1) else False)
        self.n = self.plan.n
        # The launch chains of a shard's matvec are replayed as hipGraphs (one per half of a split execute and order): a rank of
        # eight has 0.45 ms of kernels per matvec at p = 10 and the host must stay ahead of it with two collectives to issue as
        # well -- one graph launch instead of ~16 kernel launches per half (57 -> 18-27 us of host time per matvec on one GPU,
        # profiles/r03i_graph.txt; the same kernels, the same bits: tests/test_gpu_sharded_upward.py).  FMMBEM_GRAPH=0: off.
        if (self.world > 1 and not host_only and local_execute is None and local_split is None
                and os.environ.get("FMMBEM_GRAPH", "1") != "0"):
            self.plan.set_graphs(True)
        self._local = local_execute if local_execute is not None else self.plan.execute_torch
        self._xbuf = {}                                   # p -> (send, recv) exchange buffers
        # overlap of the all-gather with the near field: RCCL only (an asynchronous gloo collective on device tensors
        # goes through the host and was measured 20x slower in a one-GPU rehearsal).  Off unless FMMBEM_OVERLAP_GATHER=1:
        # the stream ordering of this path has not run on a multi-GPU node yet (SCALE_r01 was skipped), and the first
        # such run should be a measurement of the plain path, not a debugging session.
        self._overlap = (self.split and dist.is_initialized() and dist.get_backend(group) == "nccl"
                         and os.environ.get("FMMBEM_OVERLAP_GATHER", "0") == "1")
        # how y is replicated: all-gather of tree-order slices (HIP plans only) or all-reduce of zero-padded vectors
        if y_collective is None:
            y_collective = os.environ.get("FMMBEM_Y_COLLECTIVE", "allgather")
        self.y_collective = "allreduce"
        self._slices = None
        if (y_collective == "allgather" and self.world > 1 and local_execute is None and local_split is None
                and not host_only):
            cut = self.plan.shard_rows(self.world)
            self._cut = cut
            self._chunk = int((cut[1:] - cut[:-1]).max()) * self.plan.dof
            self.plan.set_result_slices(True)
            self.y_collective = "allgather"
        # instrumentation (bench.py): host-side seconds spent waiting in collectives, bytes moved per matvec
        self.profile = False
        self.collective_s = 0.0
        self.calls = 0

    def kernel(self):
        return self.plan.kernel()

    def options(self):
        return self.plan.options()

    def owned_rows(self):
        """Original-order indices of the result rows this rank computes."""
        s = self.plan.stats()
        return self.plan.perm()[s["owned_row_begin"]:s["owned_row_end"]]

    def exchange_bytes(self, p=None):
        """Bytes this rank RECEIVES per matvec in the two collectives: (multipole all-gather, result collective)."""
        p = self.plan.kernel().P if p is None else p
        if self.split and not self._split_fns:
            up = int(self.plan.exchange_counts(p)[1].sum()) * 8 if self.plan.exchange_mode == 2 else self.plan.exchange_doubles(p) * 8 * (self.world - 1)
        else:
            up = 0
        nd = self.n * self.plan.dof * 8
        if self.world == 1:
            return 0, 0
        if self.y_collective == "allgather":
            return up, self._chunk * 8 * (self.world - 1)
        return up, int(2 * nd * (self.world - 1) / self.world)          # ring all-reduce: reduce-scatter + all-gather

    def check_exchange_symmetry(self, p=None):
        """Collective, once, before the first all-to-all: all-gather the (send, recv) count vectors of every rank and check
        that what r sends q is what q expects from r, and that nobody sends to itself.  A mismatch inside
        all_to_all_single would be a hang or a silent overrun; here it is a False.  None when there is no all-to-all."""
        if not (self.split and not self._split_fns and self.plan.exchange_mode == 2 and self.world > 1):
            return None
        p = self.plan.kernel().P if p is None else p
        sc, rc = self.plan.exchange_counts(p)
        dev = torch.device("cuda", self.plan.device) if dist.get_backend(self.group) == "nccl" else torch.device("cpu")
        mine = torch.tensor([list(map(int, sc)), list(map(int, rc))], dtype=torch.int64, device=dev)
        everyone = [torch.zeros_like(mine) for _ in range(self.world)]
        dist.all_gather(everyone, mine, group=self.group)
        t = torch.stack(everyone).cpu()                   # [rank][send|recv][peer]
        send, recv = t[:, 0, :], t[:, 1, :]
        return bool(torch.equal(send, recv.t()) and int(send.diagonal().abs().sum()) == 0)

    def _timed(self, fn, device):
        """Run a collective; with profile on, bracket it with device syncs and add the wall time to collective_s."""
        if not self.profile:
            return fn()
        if device is not None and device.type == "cuda":
            torch.cuda.synchronize(device)
        t0 = time.perf_counter()
        r = fn()
        if device is not None and device.type == "cuda":
            torch.cuda.synchronize(device)
        self.collective_s += time.perf_counter() - t0
        return r

    def _all_to_all(self, recv, send, recv_counts, send_counts):
        """One all-to-all with uneven counts.  RCCL takes device tensors; gloo (CPU rehearsals of the N > 1 path on a one-GPU
        box) has no all-to-all at all: there the segments go through the host, point to point."""
        ns, nr = sum(send_counts), sum(recv_counts)
        if dist.get_backend(self.group) == "nccl":
            dist.all_to_all_single(recv[:nr], send[:ns], recv_counts, send_counts, group=self.group)
            return
        hs, hr = send[:ns].cpu(), torch.empty(nr, dtype=send.dtype)
        outs = list(hr.split(recv_counts)) if nr else [hr[:0] for _ in recv_counts]
        ins = list(hs.split(send_counts)) if ns else [hs[:0] for _ in send_counts]
        peer = (lambda q: q) if self.group is None else (lambda q: dist.get_global_rank(self.group, q))
        reqs = []
        for q in range(self.world):                       # gloo has no all-to-all: point-to-point, receives posted first
            if q != self.rank and recv_counts[q]:
                reqs.append(dist.irecv(outs[q], src=peer(q), group=self.group))
        for q in range(self.world):
            if q != self.rank and send_counts[q]:
                reqs.append(dist.isend(ins[q].contiguous(), dst=peer(q), group=self.group))
        for r in reqs:
            r.wait()
        if nr:
            recv[:nr].copy_(hr)

    def execute(self, x, out=None):
        """x: full charge vector, replicated on every rank (torch tensor). Returns the full result."""
        self.calls += 1
        slices = self.y_collective == "allgather"
        y = (torch.empty_like(x) if out is None else out)
        if slices:
            if self._slices is None:
                self._slices = (torch.empty(self._chunk, dtype=x.dtype, device=x.device),
                                torch.empty(self._chunk * self.world, dtype=x.dtype, device=x.device))
            part = self._slices[0]
        else:
            part = y
        if not self.split:
            if slices:
                self._local(x, out=part)
            elif out is None:
                y = part = self._local(x)
            else:
                self._local(x, out=part)
        else:
            p = self.plan.kernel().P
            a2a = not self._split_fns and self.plan.exchange_mode == 2
            if p not in self._xbuf:
                if a2a:
                    sc, rc = self.plan.exchange_counts(p)
                    self._xbuf[p] = (torch.empty(max(int(sc.sum()), 1), dtype=torch.float64, device=x.device),
                                     torch.empty(max(int(rc.sum()), 1), dtype=torch.float64, device=x.device),
                                     [int(c) for c in sc], [int(c) for c in rc])
                else:
                    per = self._split_fns[0] if self._split_fns else self.plan.exchange_doubles(p)
                    self._xbuf[p] = (torch.empty(per, dtype=torch.float64, device=x.device),
                                     torch.empty(per * self.world, dtype=torch.float64, device=x.device))
            send, recv = self._xbuf[p][0], self._xbuf[p][1]
            if self._split_fns:
                self._split_fns[1](x, send)
                dist.all_gather_into_tensor(recv, send, group=self.group)
                self._split_fns[2](recv, part)
            else:
                stream = torch.cuda.current_stream(x.device).cuda_stream
                self.plan.upward_device(x.data_ptr(), send.data_ptr(), stream, p)
                if self._overlap:
                    # the near field (HBM-bound, needs no multipoles) streams while the multipoles travel over xGMI
                    if a2a:
                        sc, rc = self._xbuf[p][2], self._xbuf[p][3]
                        work = dist.all_to_all_single(recv[:sum(rc)], send[:sum(sc)], rc, sc, group=self.group, async_op=True)
                    else:
                        work = dist.all_gather_into_tensor(recv, send, group=self.group, async_op=True)
                    self.plan.near_split_device(part.data_ptr(), stream)
                    work.wait()                           # RCCL: the current stream waits for the collective, the host does not
                elif a2a:
                    sc, rc = self._xbuf[p][2], self._xbuf[p][3]
                    self._timed(lambda: self._all_to_all(recv, send, rc, sc), x.device)
                else:
                    self._timed(lambda: dist.all_gather_into_tensor(recv, send, group=self.group), x.device)
                self.plan.downward_device(recv.data_ptr(), part.data_ptr(), stream, p)
        if self.world > 1:
            if slices:
                gathered = self._slices[1]
                self._timed(lambda: dist.all_gather_into_tensor(gathered, part, group=self.group), x.device)
                self.plan.assemble_slices_device(gathered.data_ptr(), self._chunk, y.data_ptr(),
                                                 torch.cuda.current_stream(x.device).cuda_stream)
            else:
                self._timed(lambda: dist.all_reduce(y, op=dist.ReduceOp.SUM, group=self.group), x.device)
        return y

    # the solver (solver.gmres) looks for execute_torch first
    execute_torch = execute
