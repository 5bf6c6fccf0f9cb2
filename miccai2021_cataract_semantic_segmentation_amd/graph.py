"""The whole training step as ONE hipGraph -- or, data parallel, as a short chain of hipGraphs with the gradient exchange between them.

A step of the reference's loop (managers/OCRNet_Manager.py:80-113: zero_grad -> forward -> loss -> backward -> Adam -> confusion matrix
of the batch) is ~3000 kernel launches through ctypes on up to five HIP streams: 56-60 ms of host time per step for an HRNet-W48 step
whose GPU time is ~110 ms (profiles/r04_host_enqueue_time.txt).  On one GPU the host keeps ahead; on a node whose host cores are shared
by eight rank processes, or after the next round of kernel gains, it is the wall.  Every shape in a training epoch is static (the
loaders drop the ragged last batch), no kernel of the step needs a host decision (the Lovasz loss takes its class-presence decisions on
the device), so the step is recorded ONCE -- the engine's tape, its fork / join events across the branch streams, the loss, Adam -- into a
hipGraph through stream capture and replayed with one launch per step:

    step = GraphedTrainStep(model, lambda out, lbl: criterion(*out, lbl), optimiser, img0, lbl0, confusion=cm)
    for img, lbl in loader:
        loss = step(img, lbl)          # device tensor; step.outputs are the (static) logits of the replayed step

What a replay must not freeze is kept out of the graph's kernel arguments: the Adam kernel reads {lr, bias corrections, gradient scale}
from device memory (catseg_adam_step_dev; uploaded in front of each replay: LambdaLR and the step count keep working), inputs are copied
into static buffers.  The arithmetic is the eager step's, launch for launch: results are bit-identical (tests/test_graph_gpu.py).

Data-parallel runs (model._grad_sync attached) -- SEGMENTED replay.  The bucketed all-reduce of dist.GradSync has to start while the
backward pass is still running (the xGMI traffic hides under the remaining backward kernels), and a collective is kept out of the
capture (RCCL launches it on its own communicator stream as an ordinary asynchronous call: the very path the eager loop uses and the
gloo / world-1 RCCL tests exercise).  So the capture is CUT wherever the backward tape has released enough gradient bytes
(CATSEG_SEGMENT_MB, default 48: complete buckets only, and only at points where every branch stream of the engine has joined the launch
stream -- the tape holds a parallel region's 'gradient ready' signals back until its join, engine.Ctx.done):

    graph 0: zero_grad, forward, loss, backward of the heads ... first cut      -> all_reduce(buckets ready so far)   [comm stream]
    graph 1: backward of the next HRNet modules ... second cut                  -> all_reduce(...)                    [comm stream]
    ...
    graph n: rest of the backward pass (stem)                                   -> all_reduce(tail bucket), wait for all of them
    tail graph: Adam (reads the reduced gradients; 1/world folded in), confusion matrix

Each replay is one hipGraphLaunch on the launch stream; each all_reduce is ordered behind the launch stream's position at its call (the
process group's event hand-over) and runs beside the next segment.  All segments allocate from ONE private pool and are replayed in
capture order, so an activation written by segment i is read by segment j > i exactly as in the single graph.  The backward pass of the
network is driven on the CALLING thread (EngineNet.backward_from: the gradients of the logits come from torch.autograd.grad over the loss,
the tape is popped here), because a stream capture must end on the thread that began it and autograd runs a device's nodes on its own.
"""
import torch

from . import engine
from . import plan as _plan
from .optim import FusedAdam


def default_segment_bytes():
    """CATSEG_SEGMENT_MB (default 48): gradient bytes the backward tape must have released since the last cut before the captured step is
    cut again.  A cut costs one more hipGraphLaunch per step (~0.1 ms of host time, and the GPU-side hand-over between two graphs); a
    segment that is too long delays the start of its buckets' all-reduce.  HRNet-W48 (292.7 MB of gradients): 5 graphs + the tail graph."""
    return int(_plan.get("segment_mb") * (1 << 20))


class _CutRecorder:
    """stands in for dist.GradSync on the tape WHILE THE STEP IS BEING CAPTURED: same begin / param_ready / finish surface, launches
    nothing (a capture executes nothing: there is no gradient to reduce yet), decides where the capture is cut and which buckets each
    segment releases."""

    def __init__(self, owner, sync, segment_bytes):
        self.owner, self.sync, self.segment_bytes = owner, sync, segment_bytes
        self.released = []          # buckets complete since the last cut, in the order the tape completed them
        self.bytes = 0

    def begin(self, fp):
        self.sync.begin(fp)         # (plans the buckets; resets the per-step state of the real reducer)
        self.pending = list(self.sync.pending)

    def param_ready(self, p):
        b = self.sync.bucket_of.get(id(p))
        if b is None:
            return
        self.pending[b] -= 1
        if self.pending[b] == 0:
            s, e, _ = self.sync.buckets[b]
            self.released.append(b)
            self.bytes += (e - s) * 4

    def quiet_point(self):
        """(engine.Ctx.backward, between two tape entries on the launch stream, every parallel region joined) -- the cut is taken here
        and not inside param_ready: a region's join releases its held-back signals in one burst that may complete several buckets"""
        if 0 < self.segment_bytes <= self.bytes and self.owner._open is not None:
            self.owner._cut(self.take())

    def take(self):
        out, self.released, self.bytes = self.released, [], 0
        return out

    def finish(self):               # end of the backward pass: whatever is complete now rides behind the last backward segment
        pass


class GraphedTrainStep:
    def __init__(self, model, loss_fn, optimiser, img, lbl, confusion=None, warmup=2, keep_state=True, forward_loss=None,
                 segment_bytes=None):
        """model: an EngineNet in training mode; loss_fn(model_output, labels) -> scalar loss tensor (or forward_loss(img, labels) ->
        (loss, final logits), the managers' method, instead of model + loss_fn); optimiser: FusedAdam over the model;
        img / lbl: a batch of the shapes every later call will have; confusion: optional int32 [K, K] matrix the step accumulates its
        batch's confusion matrix into (the reference's per-step training metric).
        Warm-up: `warmup` eager steps on the capture stream (workspaces, weight images, allocator pools reach their steady state);
        keep_state restores parameters, Adam moments, step count and every module buffer afterwards, so that the first call of this
        object IS step 1 of the run.
        segment_bytes: data parallel only -- see default_segment_bytes(); 0 = never cut (the whole exchange behind the backward pass)."""
        if not isinstance(optimiser, FusedAdam):
            raise TypeError("GraphedTrainStep needs optim.FusedAdam (its kernel reads the step-dependent scalars from device memory)")
        if not model.training:
            raise RuntimeError("GraphedTrainStep captures a TRAINING step: call model.train() first")
        self.model, self.loss_fn, self.opt, self.confusion = model, loss_fn, optimiser, confusion
        self.forward_loss = forward_loss
        # data parallel: the reducer is taken off the tape (a recorder takes its place during the capture); the replays drive it
        self.sync, model._grad_sync = model._grad_sync, None
        self.split = self.sync is not None
        self.segment_bytes = default_segment_bytes() if segment_bytes is None else int(segment_bytes)
        dev = img.device
        self.img = torch.empty_like(img).copy_(img)
        self.lbl = torch.empty_like(lbl).copy_(lbl)
        optimiser.device_hyper()
        fp = model.flat()
        m, v = optimiser._moments(fp)
        saved = None
        if keep_state:
            saved = (fp.flat.clone(), m.clone(), v.clone(), [b.clone() for b in model.buffers()], optimiser._steps,
                     [(mod, mod._pending_batches) for mod in model.modules() if isinstance(mod, engine.BatchNorm2d)],
                     None if confusion is None else confusion.clone())
        self.stream = torch.cuda.Stream(device=dev)
        self.stream.wait_stream(torch.cuda.current_stream(dev))
        self.graph = self.tail_graph = None
        self.graphs, self.releases = [], []          # split: the backward-cut chain and the buckets each of its graphs releases
        self._keep0 = model._keep_pass
        try:
            with torch.cuda.stream(self.stream):
                for _ in range(max(int(warmup), 1)):
                    self._body()
        except BaseException:
            self._restore_model()
            raise
        torch.cuda.current_stream(dev).wait_stream(self.stream)
        torch.cuda.synchronize(dev)
        if saved is not None:
            with torch.no_grad():
                fp.flat.copy_(saved[0])
                m.copy_(saved[1])
                v.copy_(saved[2])
                for b, s in zip(model.buffers(), saved[3]):
                    b.copy_(s)
                if confusion is not None:
                    confusion.copy_(saved[6])
            optimiser._steps = saved[4]
            for mod, n in saved[5]:
                mod._pending_batches = n
        self._bns = [mod for mod in model.modules() if isinstance(mod, engine.BatchNorm2d)]
        pend = [mod._pending_batches for mod in self._bns]
        try:
            if self.split:
                self._capture_segments(dev)
            else:
                self.graph = torch.cuda.CUDAGraph()
                with torch.cuda.graph(self.graph, stream=self.stream):
                    self.loss, self.outputs = self._body()
        except BaseException:
            self._restore_model()                    # (hand the reducer back: the caller may go on with the eager loop)
            self.graph = self.tail_graph = None
            self.graphs = []
            raise
        for mod, n in zip(self._bns, pend):          # (the capture ran the Python side of one step: undo its host-side counters)
            mod._pending_batches = n
        self.replays = 0
        self.launch_log = []                         # split: (index of the graph just replayed, bucket) of the last step, in call order

    # ------------------------------------------------------------------------------------------------ capture
    def _restore_model(self):
        self.model._grad_sync = self.sync
        self.model._keep_pass = self._keep0
        self.model._last_pass = self.model._last_outputs = None

    def _begin(self):
        g = torch.cuda.CUDAGraph()
        g.capture_begin(pool=self._pool)
        self._open = g

    def _end(self, released):
        self._open.capture_end()
        self.graphs.append(self._open)
        self.releases.append(list(released))
        self._open = None

    def _cut(self, released):
        """(called from the tape through _CutRecorder, on this thread, on the launch stream, outside every parallel region)"""
        self._end(released)
        self._begin()

    def _capture_segments(self, dev):
        import gc
        model = self.model
        torch.cuda.synchronize(dev)
        gc.collect()
        torch.cuda.empty_cache()
        self._pool = torch.cuda.graph_pool_handle()
        self._open = None
        rec = _CutRecorder(self, self.sync, self.segment_bytes)
        model._grad_sync = rec
        try:
            with torch.cuda.stream(self.stream):
                self._begin()
                try:
                    self.loss, self.outputs = self._body()
                    self._end(rec.take() + [b for b in range(len(self.sync.buckets)) if rec.pending[b] > 0])
                    # (a bucket with a parameter that received no gradient never completes on the tape: reduced last, as
                    #  GradSync.finish() does in the eager loop)
                    self.tail_graph = torch.cuda.CUDAGraph()
                    self.tail_graph.capture_begin(pool=self._pool)
                    self._open = self.tail_graph
                    self._tail(self.outputs)
                    self.tail_graph.capture_end()
                    self._open = None
                finally:
                    if self._open is not None:       # an exception inside a capture: close it so that the stream is usable again
                        try:
                            self._open.capture_end()
                        except Exception:            # noqa: BLE001
                            pass
                        self._open = None
        finally:
            model._grad_sync = None
        seen = sorted(b for r in self.releases for b in r)
        if seen != list(range(len(self.sync.buckets))):
            raise RuntimeError("segmented capture: the cuts release buckets %s of %d" % (seen, len(self.sync.buckets)))

    def release(self):
        """gives the reducer back to the model (eager steps may follow) and drops the graph(s) with their memory pool"""
        self._restore_model()
        self.graph = self.tail_graph = None
        self.graphs = []
        self.loss = self.outputs = None

    def _body(self):
        self.opt.zero_grad()
        self.model._keep_pass = self.split
        try:
            if self.forward_loss is not None:
                loss, out = self.forward_loss(self.img, self.lbl)
            else:
                out = self.model(self.img)
                loss = self.loss_fn(out, self.lbl)
            if self.split:
                # the loss's own backward through autograd (its device thread); the network's tape on THIS thread
                outs = self.model._last_outputs
                if outs is None:
                    raise RuntimeError("GraphedTrainStep: the step's forward pass did not run through the model it was given")
                gouts = torch.autograd.grad(loss, outs, allow_unused=True)
                self.model.backward_from(gouts)
            else:
                loss.backward()
        finally:
            self.model._keep_pass = self._keep0
        if not self.split:
            self._tail(out)
        return loss.detach(), out

    def _tail(self, out):
        self.opt.step()
        if self.confusion is not None:
            from .utils.metrics import t_get_confusion_matrix
            final = out[-1] if isinstance(out, (tuple, list)) else out
            t_get_confusion_matrix(final.detach(), self.lbl, self.confusion)

    # ------------------------------------------------------------------------------------------------ replay
    def __call__(self, img, lbl):
        """one training step on (img, lbl); returns the loss (a static device tensor: read or copy it before the next call)"""
        if img is not self.img:
            self.img.copy_(img, non_blocking=True)
        if lbl is not self.lbl:
            self.lbl.copy_(lbl, non_blocking=True)
        self.opt._steps += 1
        self.opt.upload_hyper()                      # (on the current stream, in front of the replays)
        if self.split:
            sync = self.sync
            sync.begin(self.model.flat())
            log = []
            for i, (g, rel) in enumerate(zip(self.graphs, self.releases)):
                g.replay()
                for b in rel:                        # behind graph i on the launch stream, beside graph i + 1
                    sync._launch(b, early=i + 1 < len(self.graphs))
                    log.append((i, b))
            self.launch_log = log
            sync.finish()                            # the launch stream waits for every bucket (exposed_wait_ms = that wait)
            self.tail_graph.replay()
        else:
            self.graph.replay()                      # (launched on the current stream, behind the copies above)
        for mod in self._bns:
            mod._pending_batches += 1
        self.model._grads_pending = True
        self.replays += 1
        return self.loss

    def overlap_report(self):
        """what the bench line's comm.overlap says about the replayed step"""
        if not self.split:
            return None
        n = len(self.graphs)
        sizes = [round((e - s) * 4 / (1 << 20), 2) for (s, e, _) in self.sync.buckets]
        early = [b for r in self.releases[:-1] for b in r]
        return {"mode": "segmented graph replay: the captured step is cut where the backward tape has released >= %.0f MB of complete "
                        "buckets; their all_reduce is launched between two hipGraphLaunches and runs beside the next graph"
                        % (self.segment_bytes / (1 << 20)),
                "graphs_per_step": n + 1, "backward_cuts": n - 1,
                "buckets_launched_before_the_last_backward_graph": len(early),
                "MB_launched_before_the_last_backward_graph": round(sum(sizes[b] for b in early), 1),
                "buckets_behind_the_backward_pass": len(self.releases[-1]) if self.releases else 0,
                "MB_behind_the_backward_pass": round(sum(sizes[b] for b in self.releases[-1]), 1) if self.releases else 0.0,
                "release_plan": [list(r) for r in self.releases]}
