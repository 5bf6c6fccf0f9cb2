"""Learning-rate multiplier schedule of the reference (utils/lr_functions.py:5-99) for
torch.optim.lr_scheduler.LambdaLR: static / exponential (gamma 0.98 when lr_params is None — the value the
OCRNet / DeepLab managers really run with) / polynomial / cosine, with optional restarts."""
import numpy as np


class LRFcts:
    def __init__(self, config, lr_restart_steps, lr_total_steps):
        self.lr_fct = config["lr_fct"]
        self.batchwise = config["lr_batchwise"]
        restarts = list(lr_restart_steps)
        if 0 not in restarts:
            restarts.insert(0, 0)
        vals = [1]
        rv = config["lr_restart_vals"]
        if isinstance(rv, (int, float)):
            for i in range(1, len(restarts)):
                vals.append(vals[i - 1] * rv)
        else:
            assert len(rv) == len(config["lr_restarts"]) - 1, "lr_restart_vals must have len(lr_restarts) - 1 entries"
            vals.extend(rv)
        if lr_total_steps not in restarts:
            restarts.append(lr_total_steps)
            vals.append(0)
        self.lr_restarts = np.array(restarts)
        self.lr_restart_vals = np.array(vals)
        self.restart_lengths = np.ones_like(self.lr_restarts)
        self.restart_lengths[:-1] = self.lr_restarts[1:] - self.lr_restarts[:-1]
        steps = np.arange(lr_total_steps + 1)
        self.curr_restart = np.searchsorted(self.lr_restarts, steps, side="right") - 1
        self.lr_params = config["lr_params"]
        if self.lr_fct == "piecewise_static":
            sched = self.lr_params["piecewise_static_schedule"]
            assert sched[-1][0] == config["epochs"]
            self.piecewise = [(int(a), float(b)) for a, b in sched]

    def __call__(self, step):
        r = self.curr_restart[step]
        since = step - self.lr_restarts[r]
        base = self.lr_restart_vals[r]
        if self.lr_fct == "static":
            return base
        if self.lr_fct == "piecewise_static":
            for end, lr in self.piecewise:
                if step <= end:
                    return lr
        if self.lr_fct == "exponential":
            gamma = .98 if self.lr_params is None else self.lr_params
            return base * gamma ** since
        length = self.restart_lengths[r]
        if self.lr_fct == "polynomial":
            power = .9 if self.lr_params is None else self.lr_params
            return base * (1 - since / length) ** power
        if self.lr_fct == "cosine":
            return base * 0.5 * (1. + np.cos(np.pi * since / length))
        raise ValueError("Learning rate schedule '{}' not recognised.".format(self.lr_fct))
