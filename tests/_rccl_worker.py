"""Child process of tests/test_00_a_launch_gpu.py: RCCL itself on a one-GPU box.

A process group with backend 'nccl' (= RCCL) and world_size 1 is created exactly as a data-parallel rank creates it
(dist.init_group: device-bound, high-priority communicator stream), and the bucketed gradient exchange is FORCED to launch
(GradSync(force=True)) although a world of one has nothing to add: every bucket goes through ncclAllReduce on RCCL's stream,
ordered behind the launch stream by the event hand-over the real multi-GPU run uses, while the rest of the backward pass --
the persistent 512-block trunk kernels among it -- keeps running.  Checked by the parent:
  * the flat gradient after the forced exchange is BIT-IDENTICAL to the same backward without a reducer (sum over one rank),
  * several buckets were launched, bytes_reduced = the whole flat buffer,
  * the exposed wait of the launch stream is recorded,
  * an out-of-place collective (all_gather_into_tensor) returns its input: RCCL kernels really ran on this device.
"""
import argparse
import json
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--out")
    ap.add_argument("--steps", type=int, default=3)
    a = ap.parse_args()
    import torch
    import torch.distributed as dist
    import bench
    from miccai2021_cataract_semantic_segmentation_amd import dist as D
    from miccai2021_cataract_semantic_segmentation_amd.losses import TwoScaleLoss
    from miccai2021_cataract_semantic_segmentation_amd.models import OCRNet

    dev = torch.device("cuda", 0)
    torch.cuda.set_device(dev)
    os.environ.update(RANK="0", LOCAL_RANK="0", WORLD_SIZE="1", CATSEG_DIST_SINGLE="1")
    rank, local, world = D.init_from_env()
    assert dist.is_initialized() and dist.get_backend() == "nccl" and dist.get_world_size() == 1
    crit = TwoScaleLoss({"experiment": 3, "interm": {"name": "LovaszSoftmax", "args": [], "weight": 0.4},
                         "final": {"name": "LovaszSoftmax", "args": [], "weight": 1.0}})
    # the bench model (HRNetV2-W48 trunk + OCR heads) on a map large enough for the trunk's persistent-block launches
    torch.manual_seed(0)
    model = OCRNet(dict(bench.MODELS["ocrnet_hrnet48"][0]), 3).to(dev).train()
    x, lbl = bench.synth_batch(2, 256, 384, 25, 5, dev)

    def backward_pass():
        model.zero_grad()
        loss = crit(*model(x), lbl)
        loss.backward()
        torch.cuda.synchronize()
        return model.flat().grad.clone(), float(loss.detach())

    g_plain, l_plain = backward_pass()                       # no reducer
    g_plain2, _ = backward_pass()                            # (the step itself is deterministic)
    scale = D.attach(model, bucket_bytes=16 << 20, force=True)
    sync = model._grad_sync
    res = []
    for _ in range(a.steps):
        res.append(backward_pass())
    st = sync.stats()
    # an out-of-place collective: in a world of one RCCL copies send -> recv on its stream
    src = torch.arange(1 << 20, dtype=torch.float32, device=dev)
    dst = torch.zeros_like(src)
    dist.all_gather_into_tensor(dst, src)
    torch.cuda.synchronize()
    # the step in the DEFAULT execution mode of `bench.py --gpus N` with the RCCL group alive (its watchdog thread included): the
    # captured step cut into a chain of hipGraphs at bucket boundaries, every bucket's forced ncclAllReduce launched between two replays
    # on RCCL's high-priority stream, the launch stream waiting for them in front of the tail graph (Adam)
    from miccai2021_cataract_semantic_segmentation_amd.graph import GraphedTrainStep
    from miccai2021_cataract_semantic_segmentation_amd.optim import FusedAdam
    opt = FusedAdam(model, lr=0.0, grad_scale=scale)
    step = GraphedTrainStep(model, lambda o, l: crit(*o, l), opt, x, lbl, segment_bytes=32 << 20)
    assert step.split and model._grad_sync is None
    g_graph = []
    early0, bytes0 = sync.launches_early, sync.bytes_reduced
    for _ in range(2):
        lg = float(step(x, lbl))
        torch.cuda.synchronize()
        g_graph.append((model.flat().grad.clone(), lg))
    graph_overlap = step.overlap_report()
    graph_log = list(step.launch_log)
    n_graphs = len(step.graphs)
    step.release()
    assert model._grad_sync is sync
    st2 = sync.stats()
    out = {"deterministic": bool(torch.equal(g_plain, g_plain2)),
           "graph_bit_identical": all(bool(torch.equal(g, g_plain)) and l == l_plain for g, l in g_graph),
           "graph_steps_reduced": st2["steps"] - st["steps"], "graph_overlap": graph_overlap, "graph_launch_log": graph_log,
           "graph_backward_graphs": n_graphs, "graph_early_launches_per_step": (sync.launches_early - early0) / 2,
           "graph_bytes_per_step": (sync.bytes_reduced - bytes0) / 2,
           "graph_exposed_wait_ms": st2["exposed_wait_ms"],
           "bit_identical": all(bool(torch.equal(g, g_plain)) for g, _ in res),
           "loss_equal": all(l == l_plain for _, l in res), "scale": scale, "stats": st,
           "flat_bytes": int(model.flat().grad.numel() * 4), "gather_ok": bool(torch.equal(dst, src)),
           "nccl_version": list(torch.cuda.nccl.version()) if hasattr(torch.cuda, "nccl") else None,
           "grad_norm": float(g_plain.norm())}
    with open(a.out, "w") as f:
        json.dump(out, f)
    dist.destroy_process_group()


if __name__ == "__main__":
    main()
