"""Adam on the flat parameter buffer: one HIP kernel per step for the whole network
(torch.optim.Adam(lr) semantics as built at managers/BaseManager.py:441 of the reference)."""
import torch

from . import ops


class FusedAdam(torch.optim.Optimizer):
    """Drop-in for ``torch.optim.Adam(model.parameters(), lr=...)`` on an EngineNet.
    Works with ``torch.optim.lr_scheduler.LambdaLR`` (reads ``param_groups[0]['lr']``)."""

    def __init__(self, model, lr=1e-3, betas=(0.9, 0.999), eps=1e-8, grad_scale=1.0):
        self.model = model
        super().__init__(list(model.parameters()), dict(lr=lr, betas=betas, eps=eps))
        self._m = self._v = None
        self._steps = 0
        self.grad_scale = grad_scale

    def zero_grad(self, set_to_none=True):
        """The backward tape OVERWRITES each parameter's slice of the flat gradient buffer (it does not
        accumulate over several backward passes); clearing the buffer here only matters for parameters
        that receive no gradient in a step (one memset of the flat buffer, ~0.1 ms)."""
        fp = self.model.flat()
        if fp.grad is not None:
            fp.grad.zero_()
        return None

    @torch.no_grad()
    def step(self, closure=None):
        fp = self.model.flat()
        if self._m is None or self._m.data_ptr() == 0 or self._m.numel() != fp.flat.numel() or self._m.device != fp.flat.device:
            self._m = torch.zeros_like(fp.flat)
            self._v = torch.zeros_like(fp.flat)
        g = self.param_groups[0]
        self._steps += 1
        ops.adam_step(fp.flat, fp.grad, self._m, self._v, float(g["lr"]), self._steps, g["betas"][0], g["betas"][1], g["eps"],
                      self.grad_scale)

    def state_dict(self):
        return {"steps": self._steps, "m": self._m, "v": self._v,
                "param_groups": [{k: v for k, v in g.items() if k != "params"} for g in self.param_groups]}

    def load_state_dict(self, sd):
        self._steps, self._m, self._v = sd["steps"], sd["m"], sd["v"]
        for g, s in zip(self.param_groups, sd["param_groups"]):
            g.update(s)
