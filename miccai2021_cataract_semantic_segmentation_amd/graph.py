"""The whole training step as ONE hipGraph.

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
Data-parallel runs (model._grad_sync attached): the graph ends with the backward pass; the bucketed all-reduce of the flat gradient
buffer, Adam and the confusion matrix follow as ordinary launches (three to a dozen calls) -- the collectives stay OUTSIDE the capture
(a captured RCCL collective cannot be validated on the one-GPU boxes this was built on), at the price of an exchange that no longer
overlaps the backward pass: 293 MB over xGMI, ~2 ms at 8 GPUs, against ~3000 launches less host work per step.
"""
import torch

from . import engine
from .optim import FusedAdam


class GraphedTrainStep:
    def __init__(self, model, loss_fn, optimiser, img, lbl, confusion=None, warmup=2, keep_state=True, forward_loss=None):
        """model: an EngineNet in training mode; loss_fn(model_output, labels) -> scalar loss tensor (or forward_loss(img, labels) ->
        (loss, final logits), the managers' method, instead of model + loss_fn); optimiser: FusedAdam over the model;
        img / lbl: a batch of the shapes every later call will have; confusion: optional int32 [K, K] matrix the step accumulates its
        batch's confusion matrix into (the reference's per-step training metric).
        Warm-up: `warmup` eager steps on the capture stream (workspaces, weight images, allocator pools reach their steady state);
        keep_state restores parameters, Adam moments, step count and every module buffer afterwards, so that the first call of this
        object IS step 1 of the run."""
        if not isinstance(optimiser, FusedAdam):
            raise TypeError("GraphedTrainStep needs optim.FusedAdam (its kernel reads the step-dependent scalars from device memory)")
        if not model.training:
            raise RuntimeError("GraphedTrainStep captures a TRAINING step: call model.train() first")
        self.model, self.loss_fn, self.opt, self.confusion = model, loss_fn, optimiser, confusion
        self.forward_loss = forward_loss
        # data parallel: the reducer is taken off the tape; the exchange runs behind the replayed graph
        self.sync, model._grad_sync = model._grad_sync, None
        self.split = self.sync is not None
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
        try:
            with torch.cuda.stream(self.stream):
                for _ in range(max(int(warmup), 1)):
                    self._body()
        except BaseException:
            model._grad_sync = self.sync
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
        self.graph = torch.cuda.CUDAGraph()
        try:
            with torch.cuda.graph(self.graph, stream=self.stream):
                self.loss, self.outputs = self._body()
        except BaseException:
            model._grad_sync = self.sync            # (hand the reducer back: the caller may go on with the eager loop)
            self.graph = None
            raise
        for mod, n in zip(self._bns, pend):          # (the capture ran the Python side of one step: undo its host-side counters)
            mod._pending_batches = n
        self.replays = 0

    def release(self):
        """gives the reducer back to the model (eager steps may follow) and drops the graph with its memory pool"""
        if self.sync is not None:
            self.model._grad_sync = self.sync
        self.graph = None
        self.loss = self.outputs = None

    def _body(self):
        self.opt.zero_grad()
        if self.forward_loss is not None:
            loss, out = self.forward_loss(self.img, self.lbl)
        else:
            out = self.model(self.img)
            loss = self.loss_fn(out, self.lbl)
        loss.backward()
        if not self.split:
            self._tail(out)
        return loss.detach(), out

    def _tail(self, out):
        self.opt.step()
        if self.confusion is not None:
            from .utils.metrics import t_get_confusion_matrix
            final = out[-1] if isinstance(out, (tuple, list)) else out
            t_get_confusion_matrix(final.detach(), self.lbl, self.confusion)

    def __call__(self, img, lbl):
        """one training step on (img, lbl); returns the loss (a static device tensor: read or copy it before the next call)"""
        cur = torch.cuda.current_stream(self.img.device)
        if img is not self.img:
            self.img.copy_(img, non_blocking=True)
        if lbl is not self.lbl:
            self.lbl.copy_(lbl, non_blocking=True)
        if self.split:
            self.graph.replay()
            self.sync.begin(self.model.flat())       # every bucket, highest offsets first as the tape would have released them
            for b in reversed(range(len(self.sync.buckets))):
                self.sync._launch(b)
            self.sync.finish()
            self._tail(self.outputs)                 # (eager Adam: advances the step count and uploads its scalars itself)
        else:
            self.opt._steps += 1
            self.opt.upload_hyper()
            self.graph.replay()                      # (launched on the current stream, behind the copies above)
        for mod in self._bns:
            mod._pending_batches += 1
        self.model._grads_pending = True
        self.replays += 1
        del cur
        return self.loss
