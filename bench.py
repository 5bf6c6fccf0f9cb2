#!/usr/bin/env python3
"""Headline benchmark: training frames/s of OCRNet (25 classes) with the TwoScale Lovasz-Softmax loss and
Adam, batch 8 per GPU at 3x544x960 (a 540x960 frame after the reference's 'pad' transform), synthetic
data, random-init weights, fp32.  --model ocrnet_hrnet48 (default; the configuration BASELINE.json's
metric names: HRNetV2-W48 trunk + the reference's OCR heads) or ocrnet_r50 (the OCRNet the reference ships,
configs/OCRNet_rf_lvsz.json: ResNet50, output stride 8).

    python bench.py --gpus 1 --steps 10 --warmup 3
    python bench.py --gpus N --steps K --warmup W          (no torchrun environment: launches the N ranks itself, see self_launch)
    python -m torch.distributed.run --nnodes=1 --nproc-per-node N --master-addr 127.0.0.1 --master-port P \
        bench.py --gpus N --steps K --warmup W

Prints ONE JSON line on rank 0 (see DESIGN.md "Measurement").
"""
import argparse
import json
import os
import sys
import time

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)


def _gpus_arg(argv):
    for i, a in enumerate(argv):
        if a == "--gpus" and i + 1 < len(argv):
            return int(argv[i + 1])
        if a.startswith("--gpus="):
            return int(a.split("=", 1)[1])
    return 1


def self_launch(argv):
    """`python bench.py --gpus N` with N > 1 and no torchrun environment: this process -- which has made NO GPU call and has not even
    imported torch -- starts `python -m torch.distributed.run --nnodes=1 --nproc-per-node N --master-addr 127.0.0.1 --master-port P
    bench.py <the same arguments>` as a fresh child (one fresh rank process per GPU), relays rank 0's single JSON line on its own stdout
    (everything else the job printed goes to stderr) and exits with the child's return code."""
    import socket
    import subprocess
    n = _gpus_arg(argv)
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    port = s.getsockname()[1]
    s.close()
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", str(n), "--master-addr", "127.0.0.1",
           "--master-port", str(port), os.path.abspath(__file__)] + list(argv)
    env = dict(os.environ)
    env.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")       # dmabuf IPC: RCCL's peer mappings fail without it on this driver
    print("bench: --gpus %d without WORLD_SIZE: launching %s" % (n, " ".join(cmd)), file=sys.stderr, flush=True)
    p = subprocess.run(cmd, stdout=subprocess.PIPE, env=env)
    line = None
    for ln in p.stdout.decode(errors="replace").splitlines():
        if ln.startswith('{"metric"'):
            line = ln
        elif ln.strip():
            print(ln, file=sys.stderr)
    if line is not None:
        print(line, flush=True)
    elif p.returncode == 0:
        print("bench: the %d-rank job printed no result line" % n, file=sys.stderr)
        return 1
    return p.returncode


if __name__ == "__main__" and _gpus_arg(sys.argv[1:]) > 1 and "WORLD_SIZE" not in os.environ:
    sys.exit(self_launch(sys.argv[1:]))

import torch  # noqa: E402


def synth_batch(B, H, W, K, seed, device):
    g = torch.Generator().manual_seed(seed)
    img = torch.rand(B, 3, H, W, generator=g)
    # blob-structured labels: 32x32 patches, a few classes absent, some ignore-labelled patches
    lbl = torch.randint(0, K + 1, (B, H // 32, W // 32), generator=g)
    lbl[lbl == 5] = 0
    lbl[lbl == 11] = 4
    lbl[lbl == 19] = 4
    lbl = lbl.repeat_interleave(32, 1).repeat_interleave(32, 2).contiguous()
    return img.to(device), lbl.to(device)


MODELS = {
    "ocrnet_hrnet48": ({"backbone": "hrnet48", "pretrained": False}, "OCRNet-HRNetV2-W48 (stride-4 720-ch concat + OCR heads)"),
    "ocrnet_r50": ({"backbone": "resnet50", "out_stride": 8, "pretrained": False}, "OCRNet-ResNet50-OS8"),
    # BASELINE config 2: DeepLabv3+ ResNet50, 17-class (task 2), cross entropy with the ignore label
    "deeplabv3plus_r50": ({"backbone": "resnet50", "out_stride": 8, "pretrained": False}, "DeepLabv3+-ResNet50-OS8"),
}
IS_DEEPLAB = lambda name: name.startswith("deeplab")   # noqa: E731
PROFILE_ROUND, PREVIOUS_PROFILE_ROUND = "r06", "r05_s2"   # profiles/<round>_pmc_traffic_<model>.json feeds roofline.traffic


def host_cpu_info():
    """(model name, physical cores, logical cpus available to this process)"""
    model, cores = "unknown", set()
    try:
        phys = core = None
        for line in open("/proc/cpuinfo"):
            k, _, v = line.partition(":")
            k, v = k.strip(), v.strip()
            if k == "model name":
                model = v
            elif k == "physical id":
                phys = v
            elif k == "core id":
                core = v
            elif not line.strip():
                if core is not None:
                    cores.add((phys, core))
                phys = core = None
    except OSError:
        pass
    try:
        avail = len(os.sched_getaffinity(0))
    except AttributeError:
        avail = os.cpu_count() or 1
    return model, (len(cores) or avail), avail


def cpu_baseline(H, W, K, model_name, threads=0, all_core_figure=True):
    """the CPU oracle (port of the reference path) on this box's host cores -- the identical synthetic
    train step (zero_grad -> forward -> loss -> backward -> Adam), fp32, batch 2: 1 warm-up + 3 timed steps with anomaly
    detection off, then 1 step with torch.autograd.set_detect_anomaly(True) as the reference's main.py:8 sets it."""
    from oracle import nets as ON, losses as OL
    from oracle.state import fill_state, spec_of
    from miccai2021_cataract_semantic_segmentation_amd.models import DeepLabv3Plus, OCRNet
    cpu_model, physical, avail = host_cpu_info()
    # threads: --cpu-threads; default 32.  The sweep of this very step on the GPU box's host (128-core EPYC 9575F, tools/cpu_thread_sweep.py,
    # profiles/r03_cpu_thread_sweep.json): 16 threads 0.260 frames/s, 32: 0.249, 48: 0.160, 64: 0.129, 128 (all physical cores): 0.052 --
    # beyond 32 threads the oracle's torch CPU ops (sort, BatchNorm, small convolutions) get SLOWER, so all cores would understate the CPU
    cores = max(1, min(avail, physical, threads if threads > 0 else 32))
    torch.set_num_threads(cores)
    torch.manual_seed(0)
    deeplab = IS_DEEPLAB(model_name)
    net = DeepLabv3Plus(dict(MODELS[model_name][0]), 2) if deeplab else OCRNet(dict(MODELS[model_name][0]), 3)
    spec = spec_of(net.state_dict())
    fwd = ON.deeplabv3plus_forward if deeplab else (ON.ocrnet_hrnet_forward if "hrnet" in model_name else ON.ocrnet_forward)
    S = fill_state(spec, 0)
    params = [k for k, v in S.items() if v.dtype.is_floating_point and "running" not in k]
    for k in params:
        S[k].requires_grad_()
    nb = 2
    img, lbl = synth_batch(nb, H, W, K, 0, "cpu")
    m = {k: torch.zeros_like(S[k]) for k in params}
    v = {k: torch.zeros_like(S[k]) for k in params}

    def step(i):
        for k in params:
            S[k].grad = None
        if deeplab:
            loss = OL.cross_entropy(fwd(S, img, train=True), lbl, 2)
        else:
            oi, of = fwd(S, img, train=True)
            loss = OL.two_scale_lovasz(oi, of, lbl)
        loss.backward()
        with torch.no_grad():
            for k in params:
                OL.adam_step(S[k], S[k].grad, m[k], v[k], i, 1e-4)

    step(1)                                     # warm-up (allocator, thread pool, oneDNN primitive caches)
    timed = 3
    t0 = time.perf_counter()
    for i in range(timed):
        step(2 + i)
    dt = (time.perf_counter() - t0) / timed
    with torch.autograd.set_detect_anomaly(True):
        t0 = time.perf_counter()
        step(5)
        dt_anom = time.perf_counter() - t0
    # SURVEY 8(d) names "all physical cores": the same step once more with every physical core (one untimed + one timed step; on the
    # 128-core EPYC of the GPU boxes this is ~5x SLOWER than 32 threads, which is why `value` above is the 32-thread figure)
    all_cores = None
    used = torch.get_num_threads()
    if all_core_figure and min(avail, physical) > used:
        torch.set_num_threads(min(avail, physical))
        step(6)
        t0 = time.perf_counter()
        step(7)
        dta = time.perf_counter() - t0
        all_cores = {"value": nb / dta, "unit": "frames/s", "cores": torch.get_num_threads(), "sample": "1 untimed + 1 timed step, %.1f s" % dta}
        torch.set_num_threads(used)
    return {"value": nb / dt, "unit": "frames/s", "cores": used, "kind": "port", "all_physical_cores": all_cores,
            "cpu_model": cpu_model, "physical_cores": physical, "logical_cpus_available": avail,
            "value_anomaly_mode_on": nb / dt_anom,
            "sample": "1 warm-up + %d timed train steps (fwd + %s + bwd + Adam) of the CPU oracle, batch %d, 3x%dx%d, K=%d, fp32, "
                      "%.1f s per step; + 1 step with torch.autograd.set_detect_anomaly(True) (reference main.py:8): %.1f s"
                      % (timed, "cross entropy" if deeplab else "TwoScale-Lovasz", nb, H, W, K, dt, dt_anom)}


def DTYPE_STRING():
    from miccai2021_cataract_semantic_segmentation_amd import ops
    if ops.PRECISION == "fp32":
        return "f32"
    heads = ("forward / backward-data of those with > 192 output columns: operands scaled by a per-tensor power of two and split into 2 fp16 "
             "planes (22 significant bits), 3 fp16 MFMA products; " if ops.HEADS == "f16x2" else "")
    trunk = ("forward / backward-data of the trunk convolutions likewise (prescale from the producer's amax record, split in the kernel); "
             if ops.TRUNK == "f16x2" else "")
    return ("f32 (convolutions with K >= 2048 and >= 192 output columns, and the 3x3 trunk convolutions of the HRNet widths: " + heads + trunk +
            "otherwise fp32 operands split exactly into 3 bf16 planes, 6 bf16 MFMA products; fp32 accumulate everywhere)")


def infer_traffic(dom, calls_per_step, shape):
    """fabric bytes per call of the dominant forward kernel from the committed rocprofv3 --pmc passes of `bench.py --infer`
    (tools/pmc_traffic_run.sh infer): (bytes or None, source)"""
    for rnd in (PROFILE_ROUND, PREVIOUS_PROFILE_ROUND):
        tpath = os.path.join(ROOT, "profiles", "%s_pmc_traffic_infer.json" % rnd)
        if os.path.exists(tpath) and shape == (4, 1088, 1920):
            key = {"fwd_h2": "h2w", "fwd_b3": "b3w"}.get(dom, dom)
            per_step = json.load(open(tpath))["kernels"].get(key, {}).get("hbm_bytes_per_step")
            if per_step and calls_per_step:
                return per_step / calls_per_step, "profiles/" + os.path.basename(tpath) + " (rocprofv3 --pmc FETCH_SIZE / WRITE_SIZE passes, not measured in this run)"
    return None, None


def _check_world(args, world, dist):
    """a multi-GPU line must be an RCCL line over `--gpus` ranks: anything else (a gloo fallback, a 1-rank group) exits non-zero instead
    of printing a number that could be read as a scaling figure.  CATSEG_DIST_BACKEND=gloo (functional artefact on one GPU) is the one
    explicit exception and is labelled as such in the line's `comm` block."""
    if world != args.gpus:
        raise SystemExit("--gpus %d but WORLD_SIZE=%d (plain `python bench.py --gpus N` launches the ranks itself; or torch.distributed.run "
                         "--nproc-per-node %d)" % (args.gpus, world, args.gpus))
    if world > 1:
        be, seen = dist.get_backend(), dist.get_world_size()
        if seen != args.gpus or (be != "nccl" and os.environ.get("CATSEG_DIST_BACKEND") != "gloo"):
            raise SystemExit("bench: process group is backend=%s world=%d, expected nccl (RCCL) over %d ranks" % (be, seen, args.gpus))


def infer_cpu_baseline(H, W, threads=0):
    """the CPU oracle's eval-mode forward of ONE frame of config 5 (oracle.upernet.resnext101_upernet_infer + argmax + confusion matrix) on
    this box's host cores: one untimed + two timed frames"""
    from oracle import losses as OL
    from oracle.state import fill_state, spec_of
    from oracle.upernet import resnext101_upernet_infer
    from miccai2021_cataract_semantic_segmentation_amd.models import EncDec
    cpu_model, physical, avail = host_cpu_info()
    cores = max(1, min(avail, physical, threads if threads > 0 else 32))
    torch.set_num_threads(cores)
    torch.manual_seed(0)
    net = EncDec({"encoder": {"model": "ResNeXt101", "pretrained": False}, "decoder": {"model": "UPerNet"}}, 3)
    S = fill_state(spec_of(net.state_dict()), 0)
    img, lbl = synth_batch(1, H, W, 25, 0, "cpu")

    def frame():
        with torch.no_grad():
            return OL.confusion_matrix(resnext101_upernet_infer(S, img), lbl)
    frame()
    t0 = time.perf_counter()
    for _ in range(2):
        frame()
    dt = (time.perf_counter() - t0) / 2
    return {"value": 1.0 / dt, "unit": "frames/s", "cores": cores, "kind": "port", "cpu_model": cpu_model, "physical_cores": physical,
            "logical_cpus_available": avail,
            "sample": "1 untimed + 2 timed frames of 3x%dx%d through the CPU oracle (ResNeXt101_32x8d + UPerNet eval forward, argmax, confusion "
                      "matrix), fp32, %.1f s per frame" % (H, W, dt)}


def infer_bench(args):
    """config 5: frames sharded over ranks, no exchange in the timed region (confusion matrices are summed once at
    the end of a real run); eval-mode BatchNorm, argmax + confusion matrix included in the step"""
    from miccai2021_cataract_semantic_segmentation_amd import dist as D
    rank, local, world = D.init_from_env()
    import torch.distributed as dist
    _check_world(args, world, dist)
    dev = torch.device("cuda", local % max(torch.cuda.device_count(), 1))
    torch.cuda.set_device(dev)
    from miccai2021_cataract_semantic_segmentation_amd.models import EncDec
    from miccai2021_cataract_semantic_segmentation_amd.utils.metrics import t_get_confusion_matrix
    B = 4 if args.batch == 8 else args.batch
    H, W = (1080, 1920) if (args.height, args.width) == (544, 960) else (args.height, args.width)
    H = (H + 31) // 32 * 32   # the encoder strides by 32; 1080 -> 1088 rows (the reference pads as well)
    torch.manual_seed(0)
    model = EncDec({"encoder": {"model": "ResNeXt101", "pretrained": False}, "decoder": {"model": "UPerNet"}}, 3).to(dev).eval()
    model.get_features = False
    img, lbl = synth_batch(B, H, W, 25, 2000 + rank, dev)
    cm = torch.zeros((25, 25), dtype=torch.int32, device=dev)

    def step():
        with torch.no_grad():
            t_get_confusion_matrix(model(img), lbl, cm)

    def barrier():
        torch.cuda.synchronize()
        if world > 1:
            dist.barrier()
        torch.cuda.synchronize()
    for _ in range(args.warmup):
        step()
    barrier()
    t0 = time.perf_counter()
    for _ in range(args.steps):
        step()
    barrier()
    dt = time.perf_counter() - t0
    if world > 1:
        t = torch.tensor([dt], dtype=torch.float64, device=dev)
        dist.all_reduce(t, op=dist.ReduceOp.MAX)
        dt = float(t)
    roof = None
    if not args.no_roofline and rank == 0:
        # HIP events around every convolution launch of two more steps (forward only: the fused conv + folded-BN + ReLU kernels)
        from miccai2021_cataract_semantic_segmentation_amd import ops
        step()
        torch.cuda.synchronize()
        ops.PROFILE = []
        for _ in range(2):
            step()
        torch.cuda.synchronize()
        prof, ops.PROFILE = ops.PROFILE, None
        agg = {}
        for kind, work, e0, e1 in prof:
            a = agg.setdefault(kind, [0.0, 0.0, 0])
            a[0] += work
            a[1] += e0.elapsed_time(e1) * 1e-3
            a[2] += 1
        mm = {k: v for k, v in agg.items() if k in ("fwd", "fwd_b3", "fwd_h2") and v[1] > 0}
        if mm:
            dom = max(mm, key=lambda k: mm[k][1])
            fl, sec, n = mm[dom]
            peak = {"fwd_b3": 2500.0 / 6.0, "fwd_h2": 2500.0 / 3.0}.get(dom, 157.3)
            roof = {"bound": "mfma", "kernel": {"fwd": "igemm_f32_kernel<NT> (conv2d forward + folded BatchNorm / residual / ReLU epilogue, fp32 MFMA)",
                                                "fwd_b3": "igemm_b3w_kernel (conv2d forward, bf16x3 split precision)",
                                                "fwd_h2": "igemm_h2w8_kernel (conv2d forward + fused epilogue, f16x2 split precision: 3 fp16 MFMA products)"}[dom],
                    "achieved": fl / sec / 1e12, "peak": peak, "unit": "TFLOP/s", "frac": fl / sec / 1e12 / peak,
                    "traffic": infer_traffic(dom, n // 2, (B, H, W))[0], "traffic_source": infer_traffic(dom, n // 2, (B, H, W))[1],
                    "launches_per_step": n // 2, "avg_launch_ms": sec / n * 1e3, "algorithmic_gflop_per_launch": fl / n / 1e9,
                    "all_igemm": {k: {"tflops": v[0] / v[1] / 1e12, "ms_per_step": v[1] / 2 * 1e3, "launches_per_step": v[2] // 2}
                                  for k, v in mm.items()},
                    "conv_ms_per_step": sum(v[1] for v in mm.values()) / 2 * 1e3,
                    "algorithmic_tflop_per_step": sum(v[0] for v in mm.values()) / 2 / 1e12}
    cpu = None
    if not args.no_cpu_baseline and rank == 0 and world == 1:
        cpu = infer_cpu_baseline(H, W, args.cpu_threads)
    # the one exchange of a frame-sharded inference run (reference managers/BaseManager.py:640-688 scores one confusion matrix over the whole
    # set): every rank's matrix summed once, after the timed region
    torch.cuda.synchronize()
    if world > 1:
        dist.all_reduce(cm)
    cm_pixels, cm_local = int(cm.sum()), int((lbl != 25).sum())
    if rank == 0:
        _emit(json.dumps({"metric": "inference frames/sec @1080x1920 UPerNet-ResNeXt101", "value": world * B * args.steps / dt,
                          "unit": "frames/s", "n_gpus": world, "steps": args.steps, "warmup": args.warmup,
                          "ms_per_step": dt / args.steps * 1e3, "higher_is_better": True, "scaling": "weak", "vs_baseline": None,
                          "dtype": DTYPE_STRING(), "data": "synthetic",
                          "config": {"workload": "EncDec(ResNeXt101_32x8d + UPerNet), 25-class, bs=%d/GPU @3x%dx%d, eval-mode forward + "
                                                 "argmax + confusion matrix (BASELINE config 5)" % (B, H, W),
                                     "global_batch": world * B, "parallelism": "dp%d (frame sharded)" % world,
                                     "confusion_matrix": {"pixels_scored_all_ranks": cm_pixels, "labelled_pixels_per_step_rank0": cm_local,
                                                          "summed_over_ranks": world > 1}},
                          "roofline": roof, "cpu_baseline": cpu}))
    if world > 1:
        dist.destroy_process_group()


_RESULT_OUT = None


def _claim_stdout():
    """the contract is ONE JSON line on stdout: keep the real stdout for that line and point fd 1 at stderr for everything else
    (gloo prints its connection report to stdout; RCCL does with NCCL_DEBUG set)"""
    global _RESULT_OUT
    if _RESULT_OUT is None:
        sys.stdout.flush()
        _RESULT_OUT = os.fdopen(os.dup(1), "w")
        os.dup2(2, 1)


def _emit(line):
    _claim_stdout()
    _RESULT_OUT.write(line + "\n")
    _RESULT_OUT.flush()


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=10)
    ap.add_argument("--warmup", type=int, default=3)
    ap.add_argument("--batch", type=int, default=8, help="frames per GPU")
    ap.add_argument("--height", type=int, default=544)
    ap.add_argument("--width", type=int, default=960)
    ap.add_argument("--model", default="ocrnet_hrnet48", choices=sorted(MODELS))
    ap.add_argument("--infer", action="store_true",
                    help="BASELINE config 5 instead: EncDec(ResNeXt101_32x8d + UPerNet) inference at 3x1080x1920, 4 frames per GPU")
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--no-roofline", action="store_true")
    ap.add_argument("--no-all-core-cpu", action="store_true", help="skip the all-physical-core repetition of the CPU baseline (~1 min)")
    ap.add_argument("--cpu-threads", type=int, default=0, help="threads of the CPU baseline (0 = 32: the fastest region of the committed sweep, "
                                                                 "profiles/r03_cpu_thread_sweep.json; all 128 cores are 5x slower)")
    ap.add_argument("--no-side-figures", action="store_true",
                    help="skip the side figures measured after the timed region (exact-fp32 arithmetic, unpadded 540x960 frames, through the uint8 loader)")
    ap.add_argument("--eager", action="store_true",
                    help="run the timed steps through the Python launch loop (~3000 ctypes launches per step) instead of replaying the step as "
                         "one hipGraph (graph.GraphedTrainStep; bit-identical results)")
    ap.add_argument("--with-h2d", action="store_true",
                    help="side measurement (never the reported `value` of the contract): every step also copies its batch from "
                         "pinned host memory, float32 image + int64 labels as the reference's loader hands them over")
    args = ap.parse_args()
    _claim_stdout()

    if args.infer:
        return infer_bench(args)
    from miccai2021_cataract_semantic_segmentation_amd import dist as D
    rank, local, world = D.init_from_env()
    import torch.distributed as dist
    _check_world(args, world, dist)
    dev = torch.device("cuda", local % max(torch.cuda.device_count(), 1))   # (gloo smoke runs may share one GPU)
    torch.cuda.set_device(dev)

    from miccai2021_cataract_semantic_segmentation_amd import ops
    from miccai2021_cataract_semantic_segmentation_amd.models import DeepLabv3Plus, OCRNet
    from miccai2021_cataract_semantic_segmentation_amd.losses import CrossEntropyLoss, TwoScaleLoss
    from miccai2021_cataract_semantic_segmentation_amd.optim import FusedAdam

    deeplab = IS_DEEPLAB(args.model)
    K, B, H, W = (17 if deeplab else 25), args.batch, args.height, args.width
    torch.manual_seed(0)
    if deeplab:
        model = DeepLabv3Plus(dict(MODELS[args.model][0]), 2).to(dev).train()
        ce = CrossEntropyLoss(ignore_index=17)
        crit = lambda interm, final, lbl: ce(final, lbl)   # noqa: E731
    else:
        model = OCRNet(dict(MODELS[args.model][0]), 3).to(dev).train()
        crit = TwoScaleLoss({"experiment": 3, "interm": {"name": "LovaszSoftmax", "args": [], "weight": 0.4},
                             "final": {"name": "LovaszSoftmax", "args": [], "weight": 1.0}})
    gscale = 1.0
    if world > 1:
        D.broadcast_parameters(model)
        gscale = D.attach(model)
    opt = FusedAdam(model, lr=1e-4, grad_scale=gscale)
    # FOUR distinct synthetic batches rotate through the warm-up and the timed region (the cost of the Lovasz loss depends on the
    # predictions -- active-set pruning --, so a single fixed batch could be memorised into an unrepresentatively cheap loss)
    NBATCH = 4
    batches = [synth_batch(B, H, W, K, 1000 + rank + 7919 * i, dev) for i in range(NBATCH)]
    img, lbl = batches[0]
    counter = [0]

    host = (img.cpu().pin_memory(), lbl.cpu().pin_memory()) if args.with_h2d else None
    from miccai2021_cataract_semantic_segmentation_amd.utils.metrics import t_get_confusion_matrix
    cm = torch.zeros((K, K), dtype=torch.int32, device=dev)

    # Execution mode of the timed region: the whole step recorded once into a hipGraph and replayed (graph.GraphedTrainStep: the same
    # launches on the same streams, bit-identical results, one host call per step instead of ~3000), or --eager: the Python launch loop.
    # The instrumented roofline steps (HIP events around single launches) always run eagerly.
    from miccai2021_cataract_semantic_segmentation_amd.graph import GraphedTrainStep
    graphed = {"step": None, "key": None, "captures": 0, "capture_s": 0.0}
    use_graph = [not args.eager and not args.with_h2d]

    def drop_graph():
        if graphed["step"] is not None:
            graphed["overlap"] = graphed["step"].overlap_report()
            graphed["step"].release()
            graphed["step"], graphed["key"] = None, None

    def step(batch=None):
        """one training step as the reference's manager runs it (managers/OCRNet_Manager.py:80-113): zero_grad, forward, loss, backward,
        (gradient exchange,) Adam, and the per-step training metric: the confusion matrix of the batch's predictions"""
        if batch is not None:
            x, y = batch
        elif host is not None:
            x, y = img, lbl
            img.copy_(host[0], non_blocking=True)
            lbl.copy_(host[1], non_blocking=True)
        else:
            x, y = batches[counter[0] % NBATCH]
            counter[0] += 1
        if use_graph[0] and ops.PROFILE is None:
            key = (tuple(x.shape), tuple(y.shape), ops.PRECISION)
            if graphed["key"] != key:
                drop_graph()
                t_c = time.perf_counter()
                try:
                    graphed["step"] = GraphedTrainStep(model, (lambda o, l: crit(None, o, l)) if deeplab else (lambda o, l: crit(o[0], o[1], l)),
                                                       opt, x, y, confusion=cm)
                except Exception as e:      # noqa: BLE001  a capture the runtime refuses must not cost the line: the launch loop runs the same kernels
                    graphed["error"] = "%s: %s" % (type(e).__name__, str(e)[:300])
                    print("bench: hipGraph capture failed (%s) -- falling back to the eager launch loop" % graphed["error"], file=sys.stderr)
                    use_graph[0] = False
                    torch.cuda.synchronize()
                    return step(batch if batch is not None else (x, y))
                graphed["key"] = key
                graphed["captures"] += 1
                graphed["capture_s"] += time.perf_counter() - t_c
            return graphed["step"](x, y)
        drop_graph()
        opt.zero_grad()
        out = model(x)
        interm, final = (None, out) if deeplab else out
        loss = crit(interm, final, y)
        loss.backward()
        opt.step()
        t_get_confusion_matrix(final.detach(), y, cm)
        return loss

    def barrier():
        torch.cuda.synchronize()
        if world > 1:
            dist.barrier()
        torch.cuda.synchronize()

    for _ in range(args.warmup):
        step()
    barrier()
    w_start = None if (args.no_side_figures or world > 1) else model.flat().flat.clone()     # the weights the timed region starts from (side figures)
    t0 = time.perf_counter()
    for _ in range(args.steps):
        loss = step()
    barrier()
    dt = time.perf_counter() - t0
    if world > 1:
        t = torch.tensor([dt], dtype=torch.float64, device=dev)
        dist.all_reduce(t, op=dist.ReduceOp.MAX)
        dt = float(t)
    final_loss = float(loss.detach())

    roof = None
    if not args.no_roofline:
        # per-launch HIP-event timing of the implicit-GEMM kernels on the launch stream (2 extra steps, run by
        # EVERY rank because a step contains the gradient all-reduce; only rank 0 records)
        # (the instrumented steps run the HRNet branches one after the other: with the branches overlapped on side streams, as
        #  in the timed region, the HIP events around one launch would also time its concurrent siblings)
        from miccai2021_cataract_semantic_segmentation_amd import engine
        from miccai2021_cataract_semantic_segmentation_amd.losses import two_scale
        par, engine.PARALLEL_BRANCHES = engine.PARALLEL_BRANCHES, False
        conc, two_scale.CONCURRENT = two_scale.CONCURRENT, False
        was_graph, use_graph[0] = use_graph[0], False
        step()
        torch.cuda.synchronize()
        ops.PROFILE = [] if rank == 0 else None
        for _ in range(2):
            step()
        torch.cuda.synchronize()
        engine.PARALLEL_BRANCHES = par
        two_scale.CONCURRENT = conc
        use_graph[0] = was_graph
    if not args.no_roofline and rank == 0:
        prof, ops.PROFILE = ops.PROFILE, None
        agg = {}
        for kind, work, e0, e1 in prof:
            a = agg.setdefault(kind, [0.0, 0.0, 0])
            a[0] += work
            a[1] += e0.elapsed_time(e1) * 1e-3
            a[2] += 1
        # matrix-core operations: three ops x two arithmetics.  "*_b3" = the bf16x3 split-precision kernels (six bf16 MFMA
        # products per fp32-equivalent product: their peak is the dense bf16 peak / 6); the rest = exact fp32 MFMA.
        # "*_h2" = the f16x2 kernels (three fp16 MFMA products per fp32-equivalent product: dense fp16 peak / 3)
        PEAK_F32, PEAK_B3, PEAK_H2 = 157.3, 2500.0 / 6.0, 2500.0 / 3.0

        def peak_of(kind):
            if kind.endswith(("_h2", "_d3h", "_d3p", "_p1", "_s2p")) or kind in ("h2w", "d3h", "wgrad_d3h", "d3p", "p1", "s2p"):
                return PEAK_H2
            return PEAK_B3 if kind.endswith(("_b3", "_d3")) or kind in ("b3w", "d3") else PEAK_F32
        mm = {k: v for k, v in agg.items() if not k.startswith("hbm:") and k != "split3" and v[1] > 0}
        # the dominant KERNEL (as rocprofv3 --stats names it): forward and backward-data of the bf16x3 layers are launches of one
        # kernel (igemm_b3w_kernel); the fp32 operations are the NT / NN / TN layouts of igemm_f32_kernel
        # kernel FAMILIES as rocprofv3 --stats would sum them: the three layouts of igemm_f32_kernel (NT forward, NN backward-data incl. the
        # multi-class strided form, TN backward-weight + wgrad_direct_kernel) are ONE family -- together they are the fp32-MFMA residue of
        # the step; forward and backward-data of a split-precision route are launches of one kernel
        KERNEL_OF = {"fwd_b3": "b3w", "dgrad_b3": "b3w", "wgrad_b3": "wgrad_b3", "fwd": "f32", "dgrad": "f32", "wgrad": "f32",
                     "fwd_d3": "d3", "dgrad_d3": "d3", "wgrad_d3": "wgrad_d3", "fwd_h2": "h2w", "dgrad_h2": "h2w", "wgrad_h2": "wgrad_h2",
                     "fwd_d3h": "d3h", "dgrad_d3h": "d3h", "wgrad_d3h": "wgrad_d3h", "fwd_d3p": "d3p", "dgrad_d3p": "d3p", "wgrad_d3p": "wgrad_d3p",
                     "fwd_s2p": "s2p", "dgrad_s2p": "s2p", "wgrad_s2p": "wgrad_s2p", "fwd_p1": "p1", "dgrad_p1": "p1", "wgrad_p1": "wgrad_p1"}
        groups = {}
        for k, v in mm.items():
            g = groups.setdefault(KERNEL_OF.get(k, k), [0.0, 0.0, 0])
            g[0] += v[0]; g[1] += v[1]; g[2] += v[2]
        ranked = sorted(groups, key=lambda k: -groups[k][1])
        dom = ranked[0]                        # plain largest family time; the runner-up rides beside it (`runner_up`)
        fl, sec, n = groups[dom]
        peak = peak_of(dom)
        traffic = traffic_src = None   # HBM bytes per launch from committed rocprofv3 --pmc passes of this same command (tools/pmc_traffic.py)
        tpath = os.path.join(ROOT, "profiles", "%s_pmc_traffic_%s.json" % (PROFILE_ROUND, args.model))
        if not os.path.exists(tpath):          # (no pass of this round yet: the previous round's, named as such in traffic_source)
            tpath = os.path.join(ROOT, "profiles", "%s_pmc_traffic_%s.json" % (PREVIOUS_PROFILE_ROUND, args.model))
        def traffic_kernels():
            kern = json.load(open(tpath))["kernels"]
            f32 = [kern[k]["hbm_bytes_per_step"] for k in ("fwd", "dgrad", "wgrad") if kern.get(k, {}).get("hbm_bytes_per_step")]
            if f32:
                kern["f32"] = {"hbm_bytes_per_step": sum(f32)}
            return kern
        if os.path.exists(tpath) and (B, H, W) == (8, 544, 960):
            per_step = traffic_kernels().get(dom, {}).get("hbm_bytes_per_step")
            if per_step:
                traffic = per_step / (n // 2)          # per C-ABI call, like `achieved`
                traffic_src = "profiles/" + os.path.basename(tpath) + " (rocprofv3 --pmc FETCH_SIZE / WRITE_SIZE passes of this command, not measured in this run)"
        traffic_all = None
        if os.path.exists(tpath) and (B, H, W) == (8, 544, 960):
            kern = traffic_kernels()
            traffic_all = {k: {"hbm_bytes_per_call": kern[k]["hbm_bytes_per_step"] / (groups[k][2] // 2),
                               "algorithmic_gflop_per_call": groups[k][0] / groups[k][2] / 1e9}
                           for k in groups if k in kern and kern[k].get("hbm_bytes_per_step") and groups[k][2] >= 2}
        LABELS = {"f32": "igemm_f32_kernel / igemm_f32_multi_kernel (every layer outside the split-precision routes: NT forward, NN backward-data, "
                         "TN backward-weight + wgrad_direct_kernel incl. slab reduction; exact fp32 MFMA chains)",
                 "p1": "p1_kernel (pointwise 1x1 conv2d forward and backward-data: fp32 rows split into two fp16 planes in registers straight from "
                       "global memory, weight image through LDS, three MFMA products; csrc/pconv1.hip)",
                 "wgrad_p1": "p1t_kernel (pointwise 1x1 conv2d backward-weight, in-register split, transposed LDS reads, incl. slab reduction)",
                 "b3w": "igemm_b3w_kernel (conv2d forward and backward-data of the large layers, bf16x3 split precision)",
                 "h2w": "igemm_h2w8_kernel (conv2d forward and backward-data of the large layers, f16x2 split precision: two fp16 planes "
                        "per operand, three MFMA products; 256 x 256 block tile, two waves per SIMD)",
                 "wgrad_b3": "igemm_b3t_kernel (conv2d backward-weight, bf16x3 split precision, incl. slab reduction)",
                 "wgrad_h2": "igemm_h2t_kernel (conv2d backward-weight of the large layers, f16x2 split precision, incl. slab reduction)",
                 "wgrad_d3h": "dwgrad3_h2_kernel (direct 3x3 conv2d backward-weight of the HRNet trunk, f16x2 split precision, incl. slab reduction)",
                 "wgrad_d3": "dwgrad3_b3_kernel (direct 3x3 conv2d backward-weight of the HRNet trunk, bf16x3 split precision, incl. slab reduction)",
                 "d3p": "dconv3_pl_kernel (direct 3x3 conv2d forward and backward-data of the HRNet trunk on PRODUCER-WRITTEN fp16 x 2 planes: halo tiles "
                        "and weights stream into LDS by LDS-DMA from helper waves, three MFMA products)",
                 "wgrad_d3p": "dwgrad3_pl_kernel (direct 3x3 conv2d backward-weight of the HRNet trunk on producer-written planes of x and dy, incl. slab reduction)",
                 "d3h": "dconv3_h2_kernel (direct 3x3 conv2d forward and backward-data of the HRNet trunk, f16x2 split precision: the fp32 halo "
                        "tile scaled by its producer's amax record and split into two fp16 planes in registers, three MFMA products)",
                 "d3": "dconv3_b3_kernel (direct 3x3 conv2d forward and backward-data of the HRNet trunk, bf16x3 split precision, "
                       "in-kernel split of the fp32 halo tile)"}
        label = LABELS.get(dom, dom)
        tot_fl = sum(v[0] for v in mm.values())
        tot_s = sum(v[1] for v in mm.values()) + agg.get("split3", [0, 0, 0])[1]
        hbm = {k[4:]: {"achieved_GBps": v[0] / v[1] / 1e9, "frac_of_8TBps": v[0] / v[1] / 8e12, "ms_per_step": v[1] / 2 * 1e3,
                       "calls_per_step": v[2] // 2, "algorithmic_GB_per_step": v[0] / 2 / 1e9}
               for k, v in agg.items() if k.startswith("hbm:") and v[1] > 0}
        roof = {"bound": "mfma", "kernel": label, "achieved": fl / sec / 1e12, "peak": peak,
                "unit": "TFLOP/s", "frac": fl / sec / 1e12 / peak, "traffic": traffic, "traffic_source": traffic_src,
                "launches_per_step": n // 2, "traffic_all_kernels": traffic_all,
                "selection": "the kernel family with the largest summed event time in the instrumented steps (no override)",
                "runner_up": (None if len(ranked) < 2 else
                              {"kernel": LABELS.get(ranked[1], ranked[1]), "ms_per_step": groups[ranked[1]][1] / 2 * 1e3,
                               "achieved": groups[ranked[1]][0] / groups[ranked[1]][1] / 1e12, "peak": peak_of(ranked[1]),
                               "frac": groups[ranked[1]][0] / groups[ranked[1]][1] / 1e12 / peak_of(ranked[1]),
                               "launches_per_step": groups[ranked[1]][2] // 2}),
                "families_ms_per_step": {k: round(groups[k][1] / 2 * 1e3, 3) for k in ranked},
                "ms_per_step": sec / 2 * 1e3,
                "avg_launch_ms": sec / n * 1e3, "algorithmic_gflop_per_launch": fl / n / 1e9,
                "peak_note": "fp32 MFMA 157.3 TFLOP/s; bf16x3 kernels: dense bf16 MFMA 2500 TFLOP/s / 6 products = 416.7 TFLOP/s-equivalent; "
                             "f16x2 kernels: dense fp16 MFMA 2500 TFLOP/s / 3 products = 833.3 "
                             "(achieved counts algorithmic fp32-equivalent FLOPs 2MNK; x6 / x3 for the MFMA FLOPs issued)",
                "all_igemm": {k: {"tflops": v[0] / v[1] / 1e12, "frac": v[0] / v[1] / 1e12 / peak_of(k),
                                  "ms_per_step": v[1] / 2 * 1e3, "launches_per_step": v[2] // 2}
                              for k, v in mm.items()},
                "all_matrix_ops": {"algorithmic_tflop_per_step": tot_fl / 2 / 1e12, "ms_per_step": tot_s / 2 * 1e3,
                                   "tflops_equivalent": tot_fl / tot_s / 1e12, "frac_of_fp32_matrix_peak": tot_fl / tot_s / 1e12 / PEAK_F32,
                                   "split3_ms_per_step": agg.get("split3", [0, 0, 0])[1] / 2 * 1e3},
                "hbm_kernels": hbm}
        # whole-step view: the time the step's algorithmic work would take at the stated peaks (matrix work per arithmetic, the HBM-bound
        # kernels' algorithmic bytes at 8 TB/s), against the measured step
        lb_ms = (sum(v[0] / (peak_of(k) * 1e12) for k, v in mm.items())
                 + sum(v[0] for k, v in agg.items() if k.startswith("hbm:")) / 8e12) / 2 * 1e3
        roof["whole_step"] = {"lower_bound_ms": lb_ms, "measured_ms": dt / args.steps * 1e3, "frac": lb_ms / (dt / args.steps * 1e3),
                              "f16x2_tflop_per_step": sum(v[0] for k, v in mm.items() if k.endswith(("_h2", "_d3h", "_d3p", "_p1", "_s2p"))) / 2 / 1e12,
                              "bf16x3_tflop_per_step": sum(v[0] for k, v in mm.items() if k.endswith(("_b3", "_d3"))) / 2 / 1e12,
                              "fp32_tflop_per_step": sum(v[0] for k, v in mm.items()
                                                         if not k.endswith(("_b3", "_d3", "_h2", "_d3h", "_d3p", "_p1", "_s2p"))) / 2 / 1e12}
    comm = None
    if world > 1:
        drop_graph()                             # (hands the reducer back to the model)
        overlap = graphed.get("overlap") if (not args.eager and not args.with_h2d and "error" not in graphed) else None
        comm = model._grad_sync.stats()          # every rank (it synchronises its device); rank 0 prints
        # how the exchange sits against the backward pass in the mode the timed region ran in: the replayed step is cut into a chain of
        # hipGraphs at bucket boundaries and each all_reduce is launched between two of them (graph.GraphedTrainStep), or --eager: the
        # buckets are launched from the backward tape as their last gradient is written.  exposed_wait_ms = what the launch stream still
        # waited for after its last backward kernel
        comm["overlap"] = overlap if overlap is not None else {
            "mode": "eager launch loop: buckets are launched from the backward tape as their last gradient is written"}
        comm["payload_MB_per_step"] = round(comm["bytes_reduced_per_step"] / 1e6, 1)
        if comm["backend"] != "nccl":
            comm["note"] = "FUNCTIONAL ARTEFACT: backend %s asked for through CATSEG_DIST_BACKEND -- not an RCCL / xGMI measurement, not a scaling figure" % comm["backend"]

    def timed_steps(n, batches=None, warm=1):
        for i in range(warm):
            step(None if batches is None else next(batches))
        barrier()
        t0_ = time.perf_counter()
        for i in range(n):
            step(None if batches is None else next(batches))
        barrier()
        return (time.perf_counter() - t0_) / n

    side = {}
    SIDE_STEPS = 10
    if not args.no_side_figures and (B, H, W) == (8, 544, 960) and world == 1:
        # (N = 1 only: a scaling run measures the step, nothing else)
        # Every side figure runs from the weights the timed region started from, with lr = 0.  The cost of the Lovasz loss depends on the
        # predictions (active-set pruning, DESIGN.md 4.2: ONE confidently predicted foreground pixel decides whether a class's whole pixel
        # set enters the sort), so figures taken after 30 more training steps -- along trajectories that differ in the last bits between
        # arithmetics -- jumped by +-20 ... 40 ms for that reason alone (f16x2 trunk: 121.6 ms step, 166 ms "through the loader";
        # bf16x3 trunk: 130.7 and 130.1).
        def side_state():
            model.flat().flat.copy_(w_start)
            opt.param_groups[0]["lr"] = 0.0
        side_state()
        # (1) the same step with exact fp32 MFMA chains everywhere (CATSEG_PRECISION=fp32: no split-precision kernel)
        if ops.PRECISION != "fp32":
            saved = ops.PRECISION
            ops.PRECISION = "fp32"
            dt32 = timed_steps(SIDE_STEPS, warm=2)
            ops.PRECISION = saved
            side["exact_fp32"] = {"frames_per_s": world * B / dt32, "ms_per_step": dt32 * 1e3, "steps": SIDE_STEPS}
        # (2) frames as the camera delivers them: 540 rows, no 'pad' transform
        try:
            g540 = torch.Generator().manual_seed(3000 + rank)
            img540 = torch.rand(B, 3, 540, W, generator=g540).to(dev)
            lbl540 = torch.randint(0, K + 1, (B, 18, W // 30), generator=g540).repeat_interleave(30, 1).repeat_interleave(30, 2).contiguous().to(dev)
            side_state()
            dt540 = timed_steps(SIDE_STEPS, iter(lambda: (img540, lbl540), None), warm=2)
            side_state()
            dt544 = timed_steps(SIDE_STEPS, warm=2)      # the headline shape under the same conditions (same weights, lr = 0, same step count)
            side["unpadded_540x960"] = {"frames_per_s": world * B / dt540, "ms_per_step": dt540 * 1e3, "steps": SIDE_STEPS,
                                        "padded_544x960_same_conditions_ms_per_step": dt544 * 1e3,
                                        "note": "labels 540 rows; the odd feature-map heights (135 / 68 / 34 / 17) run the same kernels"}
        except Exception as e:   # noqa: BLE001
            side["unpadded_540x960"] = {"error": "%s: %s" % (type(e).__name__, e)}
        # (3) through the uint8 loader: pinned host frames -> side-stream copy -> GpuIngest(remap, flip, pad, blur, colour jitter) -> step
        if True:
            from miccai2021_cataract_semantic_segmentation_amd.utils.loader import PinnedFrameLoader

            class _Frames(torch.utils.data.Dataset):
                """64 raw uint8 frames 540 x 960 in host memory (the decode stays on the host in the reference too: datasets/Dataset_from_df.py:31-69)"""
                def __init__(self):
                    g = torch.Generator().manual_seed(7 + rank)
                    self.img = torch.randint(0, 256, (64, 540, 960, 3), dtype=torch.uint8, generator=g).numpy()
                    self.lbl = torch.randint(0, 36, (64, 540 // 30, 960 // 30), dtype=torch.uint8, generator=g).repeat_interleave(30, 1).repeat_interleave(30, 2).numpy()
                def __len__(self):
                    return 64
                def __getitem__(self, i):
                    return self.img[i], self.lbl[i]
            def make_loader():
                return PinnedFrameLoader(_Frames(), batch_size=B, experiment=3, flip_probability=(0.0, 0.5), pad=(2, 2), normalise=False,
                                         device=dev, blur=True, colorjitter=True, seed=rank, worker_processes=4)

            def forever(loader):
                while True:
                    for b_ in loader:
                        yield b_
            # The loader's frames carry other labels than the synthetic batch (random 30 x 30 blocks of raw class ids) and every batch its own
            # flips / blur / jitter: the Lovasz work of a step depends on them.  What the PIPELINE costs is the difference between the SAME ten
            # batches (two warm-up + eight timed, in the same order) already resident in HBM and delivered by a second, identically seeded loader.
            # The SAME 2 + 2 x HALF batches, in the same order, resident in HBM and delivered by an identically seeded loader, INTERLEAVED:
            # resident (first half) / through the loader (all) / resident (second half) -- a drift of the box between two long blocks cannot
            # make the loader "beat" resident data; >= 10 timed steps on each side.
            HALF = SIDE_STEPS // 2
            ld = make_loader()
            gen = forever(ld)
            resident = [tuple(t.clone() for t in next(gen)) for _ in range(2 + 2 * HALF)]
            gen.close()
            ld.close()
            side_state()
            dtr1 = timed_steps(HALF, iter(resident[:2 + HALF]), warm=2)
            side_state()
            ld = make_loader()
            gen = forever(ld)
            dtl = timed_steps(2 * HALF, gen, warm=2)
            gen.close()
            ld.close()
            side_state()
            dtr2 = timed_steps(HALF, iter(resident[HALF:]), warm=2)      # (warm-up: the two batches in front of the second half)
            dtr = 0.5 * (dtr1 + dtr2)
            side["through_loader"] = {"frames_per_s": world * B / dtl, "ms_per_step": dtl * 1e3, "steps": 2 * HALF,
                                      "same_batches_resident_ms_per_step": dtr * 1e3,
                                      "resident_blocks_ms_per_step": [dtr1 * 1e3, dtr2 * 1e3],
                                      "order": "resident (batches 1-%d) / loader (all %d) / resident (batches %d-%d)" % (HALF, 2 * HALF, HALF + 1, 2 * HALF),
                                      "pipeline": "PinnedFrameLoader (64 host uint8 frames 540x960, stacked by 4 forked worker processes into shared pinned staging slots) -> "
                                                  "copy on a side stream -> GpuIngest(label remap, flip, pad to 544, BlurPIL, ColorJitter, ToTensor) -> train step",
                                      "sustains_step_rate": bool(dtl <= 1.05 * dtr)}      # (within 5 %: the ingest kernels themselves are on the step's stream)
    cpu = None
    if not args.no_cpu_baseline and rank == 0 and world == 1:
        cpu = cpu_baseline(H, W, K, args.model, args.cpu_threads, all_core_figure=not args.no_all_core_cpu)
    if world > 1:
        dist.barrier()

    if rank == 0:
        out = {
            "metric": "train frames/sec @540x960 %s" % {"ocrnet_hrnet48": "OCRNet-HRNetw48", "ocrnet_r50": "OCRNet-ResNet50",
                                                       "deeplabv3plus_r50": "DeepLabv3+-ResNet50"}[args.model],
            "value": world * B * args.steps / dt, "unit": "frames/s", "n_gpus": world, "steps": args.steps,
            "warmup": args.warmup, "ms_per_step": dt / args.steps * 1e3, "higher_is_better": True, "scaling": "weak",
            "vs_baseline": None,
            # the headline arithmetic rounds its matrix operands to 22 significant bits (f16x2) where the reference computes in fp32: the
            # same workload with exact fp32 MFMA chains everywhere (CATSEG_PRECISION=fp32) rides beside `value` (side figure, 10 steps)
            "value_exact_fp32": side.get("exact_fp32", {}).get("frames_per_s"),
            "dtype": DTYPE_STRING(),
            "data": "synthetic",
            "config": {"workload": ("%s, 17-class (task 2), bs=%d/GPU @3x%dx%d, cross entropy (ignore 17), Adam lr 1e-4 (BASELINE config 2)"
                                    if deeplab else
                                    "%s, 25-class (task 3), bs=%d/GPU @3x%dx%d, TwoScale Lovasz-Softmax (0.4 interm + 1.0 final), "
                                    "Adam lr 1e-4, loss/optimiser of reference configs/OCRNet_rf_lvsz.json") % (MODELS[args.model][1], B, H, W),
                       "global_batch": world * B, "parallelism": "dp%d" % world, "final_loss": final_loss,
                       "peak_hbm_GB": round(torch.cuda.max_memory_allocated(dev) / 1e9, 1),
                       "inputs": "host (pinned) -> device copy inside every step" if args.with_h2d else "resident in HBM"},
            "roofline": roof, "cpu_baseline": cpu,
        }
        if comm is not None:
            out["comm"] = comm
        if side:
            out["side_figures"] = side
        out["config"]["step"] = "zero_grad, forward, loss, backward, Adam, confusion matrix of the batch (the reference's per-step training metric)"
        # the execution plan the timed region ran under: every route / threshold switch of the host layer with its live value
        # (miccai2021_cataract_semantic_segmentation_amd/plan.py documents the fields); `non_default` is empty for the shipped plan
        from miccai2021_cataract_semantic_segmentation_amd import plan as P
        out["config"]["plan"] = {"non_default": P.non_default(), "fields": ops.plan()}
        out["config"]["execution"] = (("the step recorded once by stream capture and replayed as one hipGraph per step (graph.GraphedTrainStep: same "
                                       "launches, same streams, bit-identical results; %d capture(s), %.1f s, outside the timed region)%s"
                                       % (graphed["captures"], graphed["capture_s"],
                                          "; data parallel: the capture is cut into a chain of hipGraphs at gradient-bucket boundaries, each bucket's "
                                          "RCCL all_reduce is launched between two replays and overlaps the rest of the backward pass; Adam + "
                                          "confusion matrix are the chain's last graph (comm.overlap)" if world > 1 else ""))
                                      if (not args.eager and not args.with_h2d and "error" not in graphed) else
                                      ("eager: one ctypes launch per kernel" + (" (hipGraph capture failed: %s)" % graphed["error"] if "error" in graphed else " (--eager)")))
        _emit(json.dumps(out))
    if world > 1:
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
