"""Data-parallel training over RCCL/xGMI (one process per GPU).

The reference has no distributed code; the semantics defined here are: every rank
runs forward/backward on its own shard of frames with LOCAL BatchNorm statistics, parameter
gradients are summed across ranks (and divided by the world size inside the Adam kernel), every
rank applies the identical update.  Gradients live in one flat buffer, so the exchange is a few
large contiguous all-reduces; buckets are launched as soon as the backward tape has produced all
gradients they cover (reverse parameter order), which overlaps the xGMI traffic with the rest of
the backward pass.
"""
import os

import torch
import torch.distributed as dist
import torch.utils.data


def init_from_env(backend=None):
    """torchrun-style bootstrap; returns (rank, local_rank, world)."""
    world = int(os.environ.get("WORLD_SIZE", "1"))
    rank = int(os.environ.get("RANK", "0"))
    local = int(os.environ.get("LOCAL_RANK", "0"))
    if world > 1 and not dist.is_initialized():
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        os.environ.setdefault("MASTER_PORT", "29500")
        if backend is None:
            backend = os.environ.get("CATSEG_DIST_BACKEND") or ("nccl" if torch.cuda.is_available() else "gloo")
        if backend == "nccl":
            torch.cuda.set_device(local)
            dist.init_process_group(backend, rank=rank, world_size=world, device_id=torch.device("cuda", local))
        else:
            dist.init_process_group(backend, rank=rank, world_size=world)
    return rank, local, world


class GradSync:
    """Bucketed all-reduce(sum) of a flat gradient buffer, driven by per-parameter 'ready' events."""

    def __init__(self, bucket_bytes=64 << 20, group=None):
        self.bucket_bytes = bucket_bytes
        self.group = group
        self.world = dist.get_world_size(group) if dist.is_initialized() else 1
        self._key = None
        self.buckets = []       # [(start, end)] over the flat buffer, in REVERSE parameter order
        self.bucket_of = {}
        self.pending = []
        self.handles = []
        self.launched = []
        self.bytes_reduced = 0

    def _plan(self, params, offsets, numel, align):
        self.buckets, self.bucket_of = [], {}
        order = sorted(params, key=lambda p: offsets[id(p)], reverse=True)
        cur_end, cur_start, members = None, None, []
        for p in order:
            o = offsets[id(p)]
            e = o + (p.numel() + align - 1) // align * align
            if cur_end is None:
                cur_end = e
            cur_start = o
            members.append(p)
            if (cur_end - cur_start) * 4 >= self.bucket_bytes:
                self._close(cur_start, cur_end, members)
                cur_end, members = None, []
        if members:
            self._close(cur_start, cur_end, members)

    def _close(self, start, end, members):
        for p in members:
            self.bucket_of[id(p)] = len(self.buckets)
        self.buckets.append((start, end, len(members)))

    def begin(self, fp):
        """fp: FlatParams-like (params, offsets, grad, ALIGN)"""
        key = (fp.grad.data_ptr(), fp.grad.numel())
        if key != self._key:
            self._plan(fp.params, fp.offsets, fp.grad.numel(), fp.ALIGN)
            self._key = key
        self.grad = fp.grad
        self.pending = [n for (_, _, n) in self.buckets]
        self.launched = [False] * len(self.buckets)
        self.handles = []

    def _launch(self, b):
        if self.launched[b]:
            return
        self.launched[b] = True
        s, e, _ = self.buckets[b]
        if self.world > 1:
            self.handles.append(dist.all_reduce(self.grad[s:e], op=dist.ReduceOp.SUM, group=self.group, async_op=True))
            self.bytes_reduced += (e - s) * 4

    def param_ready(self, p):
        b = self.bucket_of.get(id(p))
        if b is None:
            return
        self.pending[b] -= 1
        if self.pending[b] == 0:
            self._launch(b)

    def finish(self):
        for b in range(len(self.buckets)):   # parameters that received no gradient this step
            self._launch(b)
        for h in self.handles:
            h.wait()
        self.handles = []


def attach(model, bucket_bytes=64 << 20):
    """Enable data-parallel gradient averaging on an EngineNet; returns 1/world for FusedAdam.grad_scale."""
    sync = GradSync(bucket_bytes)
    model._grad_sync = sync
    return 1.0 / sync.world


def broadcast_parameters(model, src=0):
    """identical initial weights / BN running stats on every rank"""
    if not dist.is_initialized() or dist.get_world_size() == 1:
        return
    fp = model.flat()
    dist.broadcast(fp.flat, src)
    for b in model.buffers():
        dist.broadcast(b, src)


def shard_indices(indices, rank, world, drop_last=True):
    """rank-strided shard of a (shared-seed) epoch index list, equal length on every rank"""
    n = len(indices) // world * world if drop_last else len(indices)
    return list(indices[rank:n:world])


class ShardedSampler(torch.utils.data.Sampler):
    """Rank shard of an epoch's index stream.  ``source`` is either a dataset length (a fresh shared-seed permutation is
    drawn every epoch: ``seed + epoch``) or another sampler whose stream is identical on every rank (the repeat-factor
    sampler draws from a private ``Generator(seed=1)``, utils/repeat_factor_sampling.py:74-77 of the reference).  Every rank
    takes ``indices[rank::world]`` of the stream truncated to a multiple of ``world * batch``: disjoint frames, equal step
    counts on every rank (a rank that ran one step more would dead-lock the gradient all-reduce)."""

    def __init__(self, source, rank, world, batch_size=1, seed=0):
        self.source, self.rank, self.world, self.batch, self.seed = source, rank, world, max(int(batch_size), 1), seed
        self.epoch = 0

    def set_epoch(self, epoch):
        self.epoch = int(epoch)

    def _stream(self):
        if isinstance(self.source, int):
            g = torch.Generator().manual_seed(self.seed + self.epoch)
            return torch.randperm(self.source, generator=g).tolist()
        return list(iter(self.source))

    def __iter__(self):
        idx = self._stream()
        n = len(idx) // (self.world * self.batch) * (self.world * self.batch)
        self.epoch += 1                      # a loader that never calls set_epoch still reshuffles every epoch
        return iter(idx[self.rank:n:self.world])

    def __len__(self):
        total = self.source if isinstance(self.source, int) else len(self.source)
        return total // (self.world * self.batch) * self.batch


def sync_bn_stats(model, how="mean"):
    """BatchNorm running statistics drift apart across ranks (each rank normalises with its LOCAL batch statistics; the
    reference has no SyncBN).  Before validation / checkpointing every rank takes the mean over ranks
    (``how='mean'``) or rank 0's values (``how='rank0'``), so that the sharded validation pass scores exactly the weights
    that rank 0 saves.  One all-reduce of a flat buffer of all running means / variances."""
    if not dist.is_initialized() or dist.get_world_size() == 1:
        return
    bufs = [b for n, b in model.named_buffers() if b.dtype.is_floating_point]
    if not bufs:
        return
    flat = torch.cat([b.reshape(-1) for b in bufs])
    if how == "rank0":
        dist.broadcast(flat, 0)
    else:
        dist.all_reduce(flat)
        flat /= dist.get_world_size()
    o = 0
    for b in bufs:
        b.copy_(flat[o:o + b.numel()].view_as(b))
        o += b.numel()
