"""fp64-calibrated gradient check shared by the whole-network GPU tests.

An absolute tolerance on the gradients of a random-weight 60-300 layer network is meaningless: the reference's own fp32
CPU gradients sit 1e-3 ... 4e-2 (relative L2, per parameter) away from an fp64 evaluation of the same graph, depending on
how badly conditioned the case is.  So every HIP gradient is compared with the fp64 oracle gradient g64 and the error is
required to be of the size of the CPU fp32 path's own error:

    r_p = ||g_hip - g64|| / (||g_cpu32 - g64|| + 1e-4 ||g64||)        median_p r_p < 2,  95th percentile < 4,  max_p r_p < 16

(the 1e-4 floor covers parameters on which the CPU path happens to be exact to ~1e-7; the maximum over several hundred
tensors is a ratio of two noise samples, hence the wider bar on it than on the percentiles)."""
import numpy as np
import torch


def calibrated_grad_check(model, spec, seed, forward, loss_of, x, lbl, med=2.0, p95=4.0, mx=16.0, label=""):
    """model: HIP model with .grad filled for (x, lbl); spec/seed: its fill_state; forward(S, x) -> oracle output(s);
    loss_of(outputs, lbl) -> oracle loss.  Returns (median ratio, max ratio, worst relative HIP error)."""
    from oracle.state import fill_state
    grads = {}
    for dt in (torch.float32, torch.float64):
        S = {k: (v.to(dt) if v.dtype.is_floating_point else v.clone()) for k, v in fill_state(spec, seed).items()}
        params = [k for k, v in S.items() if v.dtype.is_floating_point and "running" not in k]
        for k in params:
            S[k].requires_grad_()
        loss_of(forward(S, x.to(dt)), lbl).backward()
        grads[dt] = {k: S[k].grad.double() for k in params if S[k].grad is not None}
    P = dict(model.named_parameters())
    ratios, worst = [], 0.0
    for k, g64 in grads[torch.float64].items():
        n64 = float(g64.norm())
        if n64 < 1e-7 or P[k].grad is None:
            continue
        e32 = float((grads[torch.float32][k] - g64).norm())
        eh = float((P[k].grad.detach().cpu().double() - g64).norm())
        ratios.append(eh / (e32 + 1e-4 * n64))
        worst = max(worst, eh / n64)
    ratios = np.array(ratios)
    print("%s grad error vs fp64, hip / cpu32 ratio over %d tensors: median %.2f p95 %.2f max %.2f; worst hip relative error %.3g"
          % (label, len(ratios), np.median(ratios), np.percentile(ratios, 95), ratios.max(), worst))
    assert len(ratios) > 0.9 * len(grads[torch.float64])
    assert np.median(ratios) < med and np.percentile(ratios, 95) < p95 and ratios.max() < mx, (np.median(ratios), ratios.max())
    return float(np.median(ratios)), float(ratios.max()), worst
