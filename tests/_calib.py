"""fp64-calibrated gradient check shared by the whole-network GPU tests.

An absolute tolerance on the gradients of a random-weight 60-300 layer network is meaningless: fp32 evaluations of the same
graph sit 1e-5 ... 4e-2 (relative L2, per parameter) away from an fp64 evaluation, depending on how badly conditioned the
case is -- and on the implementation: the SAME CPU oracle run with 1 thread instead of 8, or with oneDNN disabled, lands
4-7x (median; up to 200x on single tensors) further from fp64 than the default run on the HRNetv2 fixture (measured; the
summation order inside the convolutions is all that changes).  So every HIP gradient is compared with the fp64 oracle
gradient g64, and its error is required to be of the size of the fp32 CPU evaluations' own errors:

    e_p   = max over three fp32 CPU variants (default threads, 2 threads, oneDNN off) of ||g_cpu32 - g64||
    r_p   = ||g_hip - g64|| / (e_p + 1e-4 ||g64||)        median_p r_p < 2,  95th percentile < 16,  max_p r_p < 64

(the 1e-4 floor covers parameters on which every CPU variant happens to be exact to ~1e-7; percentile and maximum over several
hundred tensors are ratios of two noise samples).  Measured on the MI355X box (128 cores): the three CPU variants differ from
EACH OTHER by a median factor of 15 ... 670 per tensor on the tiny HRNetv2 fixture (BatchNorm over 12 positions in its fourth
branch); the HIP gradients -- exact-fp32 MFMA kernels or the bf16x3 split-precision kernels forced onto every layer -- sit at
median 0.1 ... 0.7, p95 1 ... 13, max 1.3 ... 44 of the worst CPU variant, moving inside that band with every change of a rounding
order (e.g. a re-vectorised bilinear kernel).  A kernel with a wrong tap, stride or scale gives ratios of 1e3 and more."""
import numpy as np
import torch


_CPU_CACHE = {}


def _cpu_grads(spec, seed, forward, loss_of, x, lbl, dt, threads=None, mkldnn=True):
    from oracle.state import fill_state
    old = torch.get_num_threads()
    if threads:
        torch.set_num_threads(threads)
    try:
        S = {k: (v.to(dt) if v.dtype.is_floating_point else v.clone()) for k, v in fill_state(spec, seed).items()}
        params = [k for k, v in S.items() if v.dtype.is_floating_point and "running" not in k]
        for k in params:
            S[k].requires_grad_()
        with torch.backends.mkldnn.flags(enabled=mkldnn):
            loss_of(forward(S, x.to(dt)), lbl).backward()
        return {k: S[k].grad.double() for k in params if S[k].grad is not None}
    finally:
        torch.set_num_threads(old)


def calibrated_grad_check(model, spec, seed, forward, loss_of, x, lbl, med=2.0, p95=None, mx=None, label=""):
    """model: HIP model with .grad filled for (x, lbl); spec/seed: its fill_state; forward(S, x) -> oracle output(s);
    loss_of(outputs, lbl) -> oracle loss.  Returns (median ratio, max ratio, worst relative HIP error)."""
    from miccai2021_cataract_semantic_segmentation_amd import ops
    split = ops.PRECISION == "bf16x3"
    p95 = p95 if p95 is not None else 16.0
    mx = mx if mx is not None else 64.0
    # the four CPU evaluations do not depend on the GPU arithmetic under test: the second precision parameter of a test (same label,
    # same inputs) reuses them (they are most of the whole-network tests' time)
    key = (label, seed, tuple(x.shape), float(x.double().sum()), int(lbl.sum()))
    if key not in _CPU_CACHE:
        g64 = _cpu_grads(spec, seed, forward, loss_of, x, lbl, torch.float64)
        variants = [_cpu_grads(spec, seed, forward, loss_of, x, lbl, torch.float32),
                    _cpu_grads(spec, seed, forward, loss_of, x, lbl, torch.float32, threads=2),
                    _cpu_grads(spec, seed, forward, loss_of, x, lbl, torch.float32, mkldnn=False)]
        _CPU_CACHE.clear()          # (one entry: the fp64 gradients of a 65 M-parameter network are 0.5 GB)
        _CPU_CACHE[key] = (g64, {k: [float((v[k] - g).norm()) for v in variants] for k, g in g64.items()})
    g64, errs = _CPU_CACHE[key]
    P = dict(model.named_parameters())
    ratios, worst, spread = [], 0.0, []
    for k, g in g64.items():
        n64 = float(g.norm())
        if n64 < 1e-7 or P[k].grad is None:
            continue
        es = errs[k]
        eh = float((P[k].grad.detach().cpu().double() - g).norm())
        ratios.append(eh / (max(es) + 1e-4 * n64))
        spread.append(max(es) / (min(es) + 1e-30))
        worst = max(worst, eh / n64)
    ratios = np.array(ratios)
    print("%s grad error vs fp64, hip / worst-of-3 cpu32 over %d tensors: median %.2f p95 %.2f max %.2f; worst hip relative error %.3g "
          "(cpu32 variants differ from each other by a median factor %.1f)" % (label, len(ratios), np.median(ratios), np.percentile(ratios, 95),
                                                                                ratios.max(), worst, float(np.median(spread))))
    assert len(ratios) > 0.9 * len(g64)
    assert np.median(ratios) < med and np.percentile(ratios, 95) < p95 and ratios.max() < mx, (np.median(ratios), ratios.max())
    return float(np.median(ratios)), float(ratios.max()), worst
