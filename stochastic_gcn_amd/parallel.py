"""Multi-GPU data parallelism for the hot path: one process per GPU, ``torch.distributed``
(backend "nccl" = RCCL over xGMI on the box; "gloo" in the CPU tests).

The reference is single-process (one tf.Session, gcn/train.py:130); this module is new
 and follows BASELINE.json's north star:

  * vertex-range sharding: rank g owns vertices [g*N/W, (g+1)*N/W) -- its slice of the train
    ids and one sampler instance seeded ``seed + g`` (rank 0 of a 1-rank job reproduces the
    reference's exact sample sequence);
  * features, the CSR and the history are replicated per GPU (Reddit: 1.12 GB + 0.2 GB + 119 MB
    of 288 GB), so every gather of the step stays local;
  * ONE collective on the gradient path: all-reduce(sum) of the flat fp32 gradient buffer
    (~0.21 M floats for Reddit = 0.84 MB: latency-bound on xGMI, so a single flat call rather
    than per-tensor buckets), divided by W to keep the mean-loss semantics of
    gcn/models.py:82-83;
  * history consistency (policy H-a): after the optimizer step the ranks all-gather their
    ``(fields[l], new_history rows)`` and every replica applies all W updates in rank order
    (deterministic; a vertex updated by two ranks in one step keeps the higher rank's row).
"""
import os

import numpy as np
import torch
import torch.distributed as dist


class DataParallel(object):
    HISTORY_FIXED_LIMIT_BYTES = 16 << 20     # per-rank payload above which the bound is too loose

    def __init__(self, backend=None, device=None, init=True):
        self.world = int(os.environ.get("WORLD_SIZE", "1"))
        self.rank = int(os.environ.get("RANK", "0"))
        self.local_rank = int(os.environ.get("LOCAL_RANK", "0"))
        self.device = device
        self.history_cap = None          # upper bound of |fields[l]| per step (rows), or None
        self._hist_bufs = {}
        if self.world > 1 and init and not dist.is_initialized():
            os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
            os.environ.setdefault("MASTER_PORT", "29511")
            backend = backend or os.environ.get("SGCN_DIST_BACKEND") or \
                ("nccl" if (device is not None and device.type == "cuda") else "gloo")
            kw = {}
            if backend == "nccl":
                kw["device_id"] = device
            dist.init_process_group(backend, rank=self.rank, world_size=self.world, **kw)

    @property
    def active(self):
        return self.world > 1

    # ---- sharding ---------------------------------------------------------------------------
    def vertex_range(self, n):
        lo = (n * self.rank) // self.world
        hi = (n * (self.rank + 1)) // self.world
        return lo, hi

    def shard_ids(self, ids, n):
        """The ids of this rank's vertex range, order preserved."""
        lo, hi = self.vertex_range(n)
        ids = np.asarray(ids)
        return ids[(ids >= lo) & (ids < hi)]

    def sampler_seed(self, seed):
        return int(seed) + self.rank

    # ---- collectives ------------------------------------------------------------------------
    def allreduce_mean_(self, flat):
        if self.active:
            dist.all_reduce(flat, op=dist.ReduceOp.SUM)
            flat.div_(self.world)
        return flat

    def broadcast_(self, flat, src=0):
        if self.active:
            dist.broadcast(flat, src=src)
        return flat

    def max_scalar(self, x):
        if not self.active:
            return x
        t = torch.tensor([x], dtype=torch.float64, device=self.device or "cpu")
        dist.all_reduce(t, op=dist.ReduceOp.MAX)
        return float(t.item())

    def barrier(self):
        if self.active:
            dist.barrier()

    def attach(self, model):
        """Install the gradient all-reduce into a model's step and align the replicas' weights."""
        model.grad_hook = self.allreduce_mean_
        model.history_hook = self.sync_history
        self.broadcast_(model.theta)

    def sync_history(self, history, idx, rows, scatter_fn):
        """All-gather this step's (idx[n], rows[n x d]) and apply every rank's update to the local
        replica ``history`` in rank order.

        Fixed-capacity path (``history_cap`` rows, set by the trainer from batch size and
        degrees): ONE all-gather of a preallocated int32 buffer ``[cap ids | cap x d row bits]``
        per rank, ids padded with -1 (the scatter kernel skips them) -- no size exchange, no
        host synchronisation, so the step stays asynchronous.  Without a bound: a size exchange
        (one host sync) followed by a padded all-gather."""
        if not self.active:
            scatter_fn(history, idx, rows)
            return
        dev = rows.device
        n, d = int(rows.shape[0]), int(rows.shape[1])
        cap = self.history_cap
        if cap is not None and n > cap:      # a rank-local branch here would desynchronise the ranks
            raise RuntimeError("history exchange: %d rows exceed history_cap=%d" % (n, cap))
        if cap is None or cap * (d + 1) * 4 > self.HISTORY_FIXED_LIMIT_BYTES:
            sizes = [torch.zeros(1, dtype=torch.int64, device=dev) for _ in range(self.world)]
            dist.all_gather(sizes, torch.tensor([n], dtype=torch.int64, device=dev))
            cap = max(int(s.item()) for s in sizes)
            bufs = None
        else:
            bufs = self._hist_bufs.get((cap, d, dev))
        if bufs is None:
            send = torch.empty(cap * (d + 1), dtype=torch.int32, device=dev)
            recv = torch.empty(self.world * cap * (d + 1), dtype=torch.int32, device=dev)
            bufs = (send, recv)
            if self.history_cap is not None and cap == self.history_cap:
                self._hist_bufs[(cap, d, dev)] = bufs
        send, recv = bufs
        send[:n] = idx
        send[n:cap] = -1
        send[cap:].view(torch.float32).view(cap, d)[:n] = rows
        per = cap * (d + 1)
        dist.all_gather([recv[r * per:(r + 1) * per] for r in range(self.world)], send)
        for r in range(self.world):
            blk = recv[r * per:(r + 1) * per]
            scatter_fn(history, blk[:cap], blk[cap:].view(torch.float32).view(cap, d))

    def shutdown(self):
        if self.active and dist.is_initialized():
            dist.destroy_process_group()
