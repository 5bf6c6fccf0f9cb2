"""Adam on the flat parameter buffer: one HIP kernel per step for the whole network
(torch.optim.Adam(lr) semantics as built at managers/BaseManager.py:441 of the reference)."""
import torch

from . import ops
from .engine import FlatParams


class FusedAdam(torch.optim.Optimizer):
    """Drop-in for ``torch.optim.Adam(model.parameters(), lr=...)`` on an EngineNet.
    Works with ``torch.optim.lr_scheduler.LambdaLR`` (reads ``param_groups[0]['lr']``).

    ``state_dict()`` / ``load_state_dict()`` speak ``torch.optim.Adam``'s format (per-parameter ``step`` / ``exp_avg`` /
    ``exp_avg_sq`` in the parameters' logical OIHW shape, one param group with ``params = [0 .. n-1]`` in
    ``model.parameters()`` order), so an ``optimiser_state_dict`` written by the reference (BaseManager.py:471-495) resumes
    here and a checkpoint written here resumes under ``torch.optim.Adam``.  Internally the moments are two flat fp32
    buffers laid out like the parameters (conv weights OHWI)."""

    def __init__(self, model, lr=1e-3, betas=(0.9, 0.999), eps=1e-8, grad_scale=1.0):
        self.model = model
        super().__init__(list(model.parameters()), dict(lr=lr, betas=betas, eps=eps, weight_decay=0, amsgrad=False))
        self._m = self._v = None
        self._steps = 0
        self.grad_scale = grad_scale
        self._hyper_dev = self._hyper_host = None      # device_hyper(): the step-dependent scalars live in device memory

    def zero_grad(self, set_to_none=True):
        """The backward tape OVERWRITES each parameter's slice of the flat gradient buffer (it does not accumulate over
        several backward passes: a second backward() before zero_grad() raises); clearing the buffer here matters for
        parameters that receive no gradient in a step (one memset of the flat buffer, ~0.1 ms)."""
        self.model.zero_grad()
        return None

    def _moments(self, fp):
        if self._m is None or self._m.numel() != fp.flat.numel() or self._m.device != fp.flat.device:
            m, v = torch.zeros_like(fp.flat), torch.zeros_like(fp.flat)
            if self._m is not None and self._m.numel() == fp.flat.numel():     # the model moved: carry the state along
                m.copy_(self._m)
                v.copy_(self._v)
            self._m, self._v = m, v
        return self._m, self._v

    @torch.no_grad()
    def step(self, closure=None):
        fp = self.model.flat()
        m, v = self._moments(fp)
        g = self.param_groups[0]
        if g.get("weight_decay", 0) or g.get("amsgrad", False):
            raise NotImplementedError("FusedAdam implements torch.optim.Adam(lr, betas, eps) without weight decay / amsgrad "
                                      "(managers/BaseManager.py:441 of the reference)")
        if self._hyper_dev is not None:
            # device-scalar form (graph.GraphedTrainStep): a launch recorded into a hipGraph reads {lr, bias corrections, gradient scale}
            # from device memory; outside a capture this call advances the step and uploads them itself
            if not torch.cuda.is_current_stream_capturing():
                self._steps += 1
                self.upload_hyper()
            ops.adam_step_dev(fp.flat, fp.grad, m, v, self._hyper_dev, g["betas"][0], g["betas"][1], g["eps"])
            return
        self._steps += 1
        ops.adam_step(fp.flat, fp.grad, m, v, float(g["lr"]), self._steps, g["betas"][0], g["betas"][1], g["eps"],
                      self.grad_scale)

    def device_hyper(self):
        """switch to the device-scalar kernel form (include/catseg.h: catseg_adam_step_dev); bit-identical updates"""
        if self._hyper_dev is None:
            fp = self.model.flat()
            self._hyper_host = torch.zeros((16, 4), dtype=torch.float32).pin_memory()     # a ring: the host may run steps ahead of the GPU
            self._hyper_events = [None] * 16
            self._hyper_dev = torch.zeros(4, dtype=torch.float32, device=fp.flat.device)
        return self

    def upload_hyper(self):
        """{lr, 1 - beta1^step, sqrt(1 - beta2^step), grad_scale} of the CURRENT step count -> device (asynchronous copy from pinned memory
        on the current stream: ordered in front of the next launch / graph replay on it)"""
        g = self.param_groups[0]
        slot = self._steps % 16
        if self._hyper_events[slot] is not None:
            self._hyper_events[slot].synchronize()         # (the copy that last read this pinned slot, 16 steps ago)
        row = self._hyper_host[slot]
        ops.lib.catseg_adam_hyper(float(g["lr"]), g["betas"][0], g["betas"][1], max(self._steps, 1), self.grad_scale, row.data_ptr())
        self._hyper_dev.copy_(row, non_blocking=True)
        ev = torch.cuda.Event()
        ev.record()
        self._hyper_events[slot] = ev

    # ------------------------------------------------------------------ torch.optim.Adam-compatible (de)serialisation
    def state_dict(self):
        fp = self.model.flat()
        groups = []
        for g in self.param_groups:
            d = {k: v for k, v in g.items() if k != "params"}
            d["params"] = list(range(len(fp.params)))
            groups.append(d)
        state = {}
        if self._steps > 0 and self._m is not None:
            step = torch.tensor(float(self._steps))
            for i, p in enumerate(fp.params):
                o = fp.offsets[id(p)]
                state[i] = {"step": step.clone(),
                            "exp_avg": FlatParams._view(self._m, o, p).detach().clone().contiguous(),
                            "exp_avg_sq": FlatParams._view(self._v, o, p).detach().clone().contiguous()}
        return {"state": state, "param_groups": groups}

    def load_state_dict(self, sd):
        if "state" not in sd or "param_groups" not in sd:
            if {"steps", "m", "v"} <= set(sd):     # round-1 flat format of this package
                self._steps, self._m, self._v = sd["steps"], sd["m"], sd["v"]
                for g, s in zip(self.param_groups, sd.get("param_groups", [])):
                    g.update({k: v for k, v in s.items() if k != "params"})
                return
            raise ValueError("unrecognised optimiser state: expected torch.optim.Adam's {'state', 'param_groups'} dictionary")
        fp = self.model.flat()
        order = [i for g in sd["param_groups"] for i in g["params"]]
        if len(order) != len(fp.params):
            raise ValueError("optimiser state has %d parameters, the model has %d" % (len(order), len(fp.params)))
        for g, s in zip(self.param_groups, sd["param_groups"]):
            g.update({k: v for k, v in s.items() if k != "params"})
        m, v = self._moments(fp)
        m.zero_()
        v.zero_()
        steps = set()
        for p, idx in zip(fp.params, order):
            st = sd["state"].get(idx)
            if st is None:
                continue
            if tuple(st["exp_avg"].shape) != tuple(p.shape):
                raise ValueError("optimiser state %d has shape %s, parameter has %s" % (idx, tuple(st["exp_avg"].shape), tuple(p.shape)))
            o = fp.offsets[id(p)]
            FlatParams._view(m, o, p).copy_(st["exp_avg"])
            FlatParams._view(v, o, p).copy_(st["exp_avg_sq"])
            steps.add(int(st["step"]))
        if len(steps) > 1:
            raise ValueError("per-parameter step counts differ (%s): not representable by the fused single-step kernel" % sorted(steps))
        self._steps = steps.pop() if steps else 0
