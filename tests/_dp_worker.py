"""Child process of tests/test_00_dp2_gpu.py: one data-parallel rank (gloo rendezvous, all ranks on GPU 0).

Phase A (no communicator): this rank's shard alone -> flat gradient g_r (the single-process reference).
Phase B (data parallel): same initial state, dist.attach + FusedAdam(grad_scale = 1/world): one step; the flat gradient
        buffer after the bucketed all-reduce and the updated parameters are saved.
Phase C: a two-epoch OCRNetManager run under WORLD_SIZE = 2 (rank-sharded loader, BN-statistics sync, sharded validation).
"""
import argparse
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--out")
    a = ap.parse_args()
    import torch
    import torch.distributed as dist
    from miccai2021_cataract_semantic_segmentation_amd import dist as D
    from miccai2021_cataract_semantic_segmentation_amd.losses import TwoScaleLoss
    from miccai2021_cataract_semantic_segmentation_amd.managers import OCRNetManager, SyntheticCataractDataset
    from miccai2021_cataract_semantic_segmentation_amd.models import OCRNet
    from miccai2021_cataract_semantic_segmentation_amd.optim import FusedAdam
    from oracle.state import fill_state, spec_of

    rank, world = int(os.environ["RANK"]), int(os.environ["WORLD_SIZE"])
    dev = torch.device("cuda", 0)
    torch.cuda.set_device(dev)
    crit = TwoScaleLoss({"experiment": 3, "interm": {"name": "LovaszSoftmax", "args": [], "weight": 0.4},
                         "final": {"name": "LovaszSoftmax", "args": [], "weight": 1.0}})
    g = torch.Generator().manual_seed(50 + rank)                                  # a different shard of frames per rank
    x = torch.rand(2, 3, 64, 96, generator=g).to(dev)
    lbl = torch.randint(0, 26, (2, 8, 12), generator=g).repeat_interleave(8, 1).repeat_interleave(8, 2).to(dev)

    def fresh():
        torch.manual_seed(0)
        m = OCRNet({"backbone": "resnet50", "out_stride": 8, "pretrained": False}, 3)
        m.load_state_dict(fill_state(spec_of(m.state_dict()), 77))
        return m.to(dev).train()

    # ---- A: single-process gradient of this rank's shard
    model = fresh()
    loss = crit(*model(x), lbl)
    loss.backward()
    g_single = model.flat().grad.clone()
    init_flat = model.flat().flat.clone()
    # ---- B: the data-parallel step
    rk, local, w = D.init_from_env()                                              # CATSEG_DIST_BACKEND=gloo
    assert (rk, w) == (rank, world) and dist.get_backend() == "gloo"
    model = fresh()
    D.broadcast_parameters(model)
    scale = D.attach(model, bucket_bytes=16 << 20)                                # several buckets on a 39 M-parameter net
    opt = FusedAdam(model, lr=1e-3, grad_scale=scale)
    opt.zero_grad()
    loss = crit(*model(x), lbl)
    loss.backward()
    g_reduced = model.flat().grad.clone()
    nb = len(model._grad_sync.buckets)
    opt.step()
    torch.cuda.synchronize()
    out = {"g_single": g_single.cpu(), "g_reduced": g_reduced.cpu(), "flat_after": model.flat().flat.detach().cpu().clone(),
           "init_flat": init_flat.cpu(), "buckets": nb, "scale": scale, "bn_before_sync": model.backbone["bn1"].running_mean.cpu().clone()}
    D.sync_bn_stats(model)
    out["bn_after_sync"] = model.backbone["bn1"].running_mean.cpu().clone()
    # a second backward without zero_grad must be refused (it would race the bucket launches)
    try:
        crit(*model(x), lbl).backward()
        out["second_backward_raised"] = False
    except RuntimeError:
        out["second_backward_raised"] = True
    dist.barrier()
    # ---- C: the manager loop under WORLD_SIZE = 2
    seen = []

    class Spy(SyntheticCataractDataset):
        def __getitem__(self, i):
            seen.append(int(i))
            return super().__getitem__(i)

    cfg = {"name": "dp", "mode": "training", "manager": "OCRNet", "log_path": os.path.join(a.out, "logs"),
           "graph": {"model": "OCRNet", "backbone": "resnet50", "out_stride": 8, "pretrained": False},
           "data": {"experiment": 2, "batch_size": 2},
           "loss": {"name": "TwoScaleLoss", "interm": {"name": "LovaszSoftmax", "args": [], "weight": 0.4},
                    "final": {"name": "LovaszSoftmax", "args": [], "weight": 1.0}},
           "train": {"learning_rate": 1e-3, "epochs": 2}, "log_every_n_epochs": 1, "seed": 0}
    mgr = OCRNetManager(cfg, Spy(10, 64, 96, 17, seed=1), SyntheticCataractDataset(3, 64, 96, 17, seed=2))
    mgr.train()
    out.update({"train_indices": seen, "history": mgr.history, "metrics": {k: v for k, v in mgr.metrics.items() if isinstance(v, (int, float))},
                "mgr_flat": mgr.model.flat().flat.detach().cpu().clone(),
                "mgr_bn": mgr.model.backbone["bn1"].running_var.cpu().clone(), "steps": mgr.global_step})
    torch.save(out, os.path.join(a.out, "rank%d.pt" % rank))
    dist.barrier()
    dist.destroy_process_group()


if __name__ == "__main__":
    main()
