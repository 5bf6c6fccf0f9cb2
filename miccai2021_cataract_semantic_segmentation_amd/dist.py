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


try:
    from . import plan as _plan
except ImportError:          # this file loaded on its own (tests/test_dist_cpu.py: gloo workers without the HIP library): plan.py by path
    import importlib.util as _ilu
    _spec = _ilu.spec_from_file_location("catseg_plan", os.path.join(os.path.dirname(os.path.abspath(__file__)), "plan.py"))
    _plan = _ilu.module_from_spec(_spec)
    _spec.loader.exec_module(_plan)

COMM_HIGH_PRIORITY = _plan.get("comm_priority") != "normal"


def init_group(backend, rank, world, local=0):
    """creates the default process group.  RCCL ('nccl'): bound to this rank's device, and with a HIGH-PRIORITY communicator stream
    (CATSEG_COMM_PRIORITY=normal: the default priority) -- the all-reduce kernels of a gradient bucket are launched while the trunk
    kernels of the remaining backward pass hold every CU with persistent blocks (csrc/dconv3_pl.hip: 512 blocks, all registers of their
    SIMDs); the hardware scheduler serves a high-priority queue first whenever a workgroup slot frees, so a bucket's few-CU ring kernel
    starts beside them instead of behind the queue of the compute streams."""
    if backend == "nccl":
        # the host driver of the target pool only supports dmabuf IPC: without this RCCL's peer mappings fail with
        # `hipIpcGetMemHandle: invalid argument`.  Set here, in front of this process's first GPU call of the group, so that a rank
        # started by an external torchrun or through managers.* gets it as bench.self_launch's children do.
        os.environ.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")
        torch.cuda.set_device(local)
        opts = None
        if COMM_HIGH_PRIORITY:
            opts = dist.ProcessGroupNCCL.Options(is_high_priority_stream=True)
        dist.init_process_group(backend, rank=rank, world_size=world, device_id=torch.device("cuda", local), pg_options=opts)
    else:
        dist.init_process_group(backend, rank=rank, world_size=world)


def init_from_env(backend=None):
    """torchrun-style bootstrap; returns (rank, local_rank, world).  A world of one creates no group unless CATSEG_DIST_SINGLE=1 asks
    for it (tests/_rccl_worker.py: RCCL itself exercised on a one-GPU box)."""
    world = int(os.environ.get("WORLD_SIZE", "1"))
    rank = int(os.environ.get("RANK", "0"))
    local = int(os.environ.get("LOCAL_RANK", "0"))
    if (world > 1 or os.environ.get("CATSEG_DIST_SINGLE") == "1") and not dist.is_initialized():
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        os.environ.setdefault("MASTER_PORT", "29500")
        init_group(resolve_backend(backend), rank, world, local)
    return rank, local, world


def resolve_backend(backend=None, cuda=None):
    """backend of the data-parallel process group: the argument, else CATSEG_DIST_BACKEND, else 'nccl' (= RCCL) on GPUs / 'gloo' without.
    gloo on a machine with GPUs is refused, however it was asked for, unless CATSEG_DIST_BACKEND=gloo says so in the environment: it
    stages every gradient bucket through the host (1.3 s of host wait per HRNet-W48 step in profiles/r03_bench_2rank_gloo_1gpu.json)."""
    env = os.environ.get("CATSEG_DIST_BACKEND")
    cuda = torch.cuda.is_available() if cuda is None else cuda
    backend = backend or env or ("nccl" if cuda else "gloo")
    if backend == "gloo" and cuda and env != "gloo":
        raise RuntimeError("data parallel on GPUs runs over RCCL (backend 'nccl'); gloo stages every bucket through the host -- "
                           "set CATSEG_DIST_BACKEND=gloo to ask for it explicitly (functional tests)")
    return backend


def default_bucket_bytes():
    """CATSEG_BUCKET_MB (default 32): size of the gradient buckets.  xGMI is point to point (7 links x ~153 GB/s per GPU): a ring
    all-reduce of a 32 MB bucket over 8 GPUs moves 2 * 7/8 * 32 MB per link ~ 0.4 ms -- long enough to run at link rate, short
    enough that several buckets are in flight under the backward pass"""
    return int(_plan.get("bucket_mb") * (1 << 20))


class GradSync:
    """Bucketed all-reduce(sum) of a flat gradient buffer, driven by per-parameter 'ready' events.  The bucket that becomes ready
    LAST (the lowest offsets: stem and first stage, whose gradients the backward pass writes at its very end) is kept small
    (tail_bytes), because its all-reduce cannot overlap with anything."""

    def __init__(self, bucket_bytes=None, group=None, tail_bytes=None, force=None):
        # force: launch the all-reduces even in a world of one (CATSEG_FORCE_ALLREDUCE=1): the collective then runs through the whole
        # RCCL path -- communicator, its stream, the event hand-over from and to the launch stream -- with nothing to add
        self.force = (os.environ.get("CATSEG_FORCE_ALLREDUCE") == "1") if force is None else bool(force)
        self.bucket_bytes = int(bucket_bytes) if bucket_bytes else default_bucket_bytes()
        self.tail_bytes = int(tail_bytes) if tail_bytes else min(self.bucket_bytes, int(_plan.get("tail_bucket_mb") * (1 << 20)))
        self.group = group
        self.world = dist.get_world_size(group) if dist.is_initialized() else 1
        self._key = None
        self.buckets = []       # [(start, end, members)] over the flat buffer
        self.bucket_of = {}
        self.pending = []
        self.handles = []
        self.launched = []
        self.bytes_reduced = 0
        self.bytes_early = 0    # ... of which launched while the backward pass still had kernels to enqueue (from the tape / between graphs)
        self.launches_early = 0
        self.steps = 0
        self.exposed_ms = 0.0   # GPU time the launch stream spent waiting for the exchange after its last backward kernel
        self.host_wait_ms = 0.0
        self._ev = None

    def _plan(self, params, offsets, numel, align):
        self.buckets, self.bucket_of = [], {}
        order = sorted(params, key=lambda p: offsets[id(p)])          # ascending offsets: the first bucket is the tail bucket
        cur_start, cur_end, members = None, None, []
        limit = self.tail_bytes
        for p in order:
            o = offsets[id(p)]
            e = o + (p.numel() + align - 1) // align * align
            if cur_start is None:
                cur_start = o
            cur_end = e
            members.append(p)
            if (cur_end - cur_start) * 4 >= limit:
                self._close(cur_start, cur_end, members)
                cur_start, members = None, []
                limit = self.bucket_bytes
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

    def _launch(self, b, early=True):
        if self.launched[b]:
            return
        self.launched[b] = True
        s, e, _ = self.buckets[b]
        if self.world > 1 or (self.force and dist.is_initialized()):
            self.handles.append(dist.all_reduce(self.grad[s:e], op=dist.ReduceOp.SUM, group=self.group, async_op=True))
            self.bytes_reduced += (e - s) * 4
            if early:
                self.bytes_early += (e - s) * 4
                self.launches_early += 1

    def param_ready(self, p):
        b = self.bucket_of.get(id(p))
        if b is None:
            return
        self.pending[b] -= 1
        if self.pending[b] == 0:
            self._launch(b)

    def finish(self):
        import time
        for b in range(len(self.buckets)):   # parameters that received no gradient this step
            self._launch(b, early=False)
        cuda = self.grad.is_cuda and bool(self.handles)
        if cuda:
            if self._ev is not None:         # (read last step's pair here: no synchronisation on the hot path)
                e0, e1 = self._ev
                if e1.query():
                    self.exposed_ms += e0.elapsed_time(e1)
            e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            e0.record()
        t0 = time.perf_counter()
        for h in self.handles:
            h.wait()
        self.host_wait_ms += (time.perf_counter() - t0) * 1e3
        if cuda:
            e1.record()
            self._ev = (e0, e1)
        self.handles = []
        self.steps += 1

    def stats(self):
        """what a scaling run needs to interpret its number: which backend carried how many bytes in how many buckets, and how long the
        launch stream waited for the exchange after the backward pass had nothing left to overlap it with"""
        if self._ev is not None and self.grad.is_cuda:
            torch.cuda.synchronize()
            self.exposed_ms += self._ev[0].elapsed_time(self._ev[1])
            self._ev = None
        n = max(self.steps, 1)
        return {"backend": dist.get_backend(self.group) if dist.is_initialized() else "none",
                "world_seen_by_backend": self.world, "buckets": len(self.buckets), "bucket_MB": self.bucket_bytes / (1 << 20),
                "tail_bucket_MB": self.tail_bytes / (1 << 20),
                "bucket_sizes_MB": [round((e - s) * 4 / (1 << 20), 2) for (s, e, _) in self.buckets],
                "bytes_reduced_per_step": self.bytes_reduced // n, "exposed_wait_ms": self.exposed_ms / n,
                "bytes_launched_under_backward_per_step": self.bytes_early // n,
                "buckets_launched_under_backward_per_step": self.launches_early / n,
                "host_wait_ms": self.host_wait_ms / n, "steps": self.steps,
                "comm_stream_priority": "high" if (COMM_HIGH_PRIORITY and dist.is_initialized() and dist.get_backend(self.group) == "nccl")
                else "default", "forced_in_world_of_one": bool(self.force and self.world == 1)}


def attach(model, bucket_bytes=None, force=None):
    """Enable data-parallel gradient averaging on an EngineNet; returns 1/world for FusedAdam.grad_scale."""
    sync = GradSync(bucket_bytes, force=force)
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


def check_unsharded(source):
    """a sampler that already shards itself (utils.RepeatFactorSampler(rank=, world=)) must not be sharded again: every rank would
    train on 1 / world^2 of the epoch and the step counts would be derived from the wrong total"""
    if getattr(source, "world", 1) > 1:
        raise ValueError("the sampler passed to a rank-sharding wrapper already shards its stream (world=%d): construct it with "
                         "rank=0, world=1 and let ShardedSampler / PinnedFrameLoader / the manager take the rank's slice"
                         % getattr(source, "world"))


class ShardedSampler(torch.utils.data.Sampler):
    """Rank shard of an epoch's index stream.  ``source`` is either a dataset length (a fresh shared-seed permutation is
    drawn every epoch: ``seed + epoch``) or another sampler whose stream is identical on every rank (the repeat-factor
    sampler draws from a private ``Generator(seed=1)``, utils/repeat_factor_sampling.py:74-77 of the reference).  Every rank
    takes ``indices[rank::world]`` of the stream truncated to a multiple of ``world * batch``: disjoint frames, equal step
    counts on every rank (a rank that ran one step more would dead-lock the gradient all-reduce)."""

    def __init__(self, source, rank, world, batch_size=1, seed=0):
        check_unsharded(source)
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
    reference has no SyncBN).  Before validation / checkpointing every rank takes the POOLED statistics over ranks
    (``how='mean'``: mean of the running means; variance = mean of the running variances + the variance of the running means, i.e.
    the variance of the pooled population for equally sized shards) or rank 0's values (``how='rank0'``), so that the sharded
    validation pass scores exactly the weights that rank 0 saves.  Only the running_mean / running_var buffers of BatchNorm
    modules take part; one all-reduce of a flat buffer."""
    if not dist.is_initialized() or dist.get_world_size() == 1:
        return
    bns = [m for m in model.modules() if isinstance(m, torch.nn.modules.batchnorm._BatchNorm) and m.running_mean is not None]
    if not bns:
        return
    means = torch.cat([m.running_mean.reshape(-1) for m in bns])
    vars_ = torch.cat([m.running_var.reshape(-1) for m in bns])
    n = means.numel()
    if how == "rank0":
        flat = torch.cat([means, vars_])
        dist.broadcast(flat, 0)
        means, vars_ = flat[:n], flat[n:]
    else:
        flat = torch.cat([means, vars_, means * means])
        dist.all_reduce(flat)
        flat /= dist.get_world_size()
        means = flat[:n]
        vars_ = flat[n:2 * n] + (flat[2 * n:] - means * means).clamp_min(0)
    o = 0
    for m in bns:
        k = m.running_mean.numel()
        m.running_mean.copy_(means[o:o + k].view_as(m.running_mean))
        m.running_var.copy_(vars_[o:o + k].view_as(m.running_var))
        o += k
