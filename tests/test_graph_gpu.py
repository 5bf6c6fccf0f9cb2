"""The whole training step replayed as one hipGraph (graph.GraphedTrainStep) against the eager launch loop: the same kernels in the same
order on the same streams -> bit-identical losses, parameters, Adam moments, BatchNorm statistics and confusion matrices over several
steps with changing batches and a changing learning rate (reference loop: managers/OCRNet_Manager.py:80-113)."""
import pytest
import torch

pytestmark = pytest.mark.gpu


def _need_gpu():
    if not torch.cuda.is_available():
        pytest.skip("needs a GPU")


def _setup(name, B, H, W):
    import bench
    from miccai2021_cataract_semantic_segmentation_amd.losses import TwoScaleLoss
    from miccai2021_cataract_semantic_segmentation_amd.models import OCRNet
    from miccai2021_cataract_semantic_segmentation_amd.optim import FusedAdam
    dev = torch.device("cuda")
    torch.manual_seed(0)
    model = OCRNet(dict(bench.MODELS[name][0]), 3).to(dev).train()
    crit = TwoScaleLoss({"experiment": 3, "interm": {"name": "LovaszSoftmax", "args": [], "weight": 0.4},
                         "final": {"name": "LovaszSoftmax", "args": [], "weight": 1.0}})
    opt = FusedAdam(model, lr=1e-3)
    batches = [bench.synth_batch(B, H, W, 25, 300 + i, dev) for i in range(4)]
    return model, crit, opt, batches


@pytest.mark.parametrize("name,shape", [("ocrnet_hrnet48", (2, 128, 192)), ("ocrnet_r50", (2, 64, 96))])
def test_graphed_step_is_bit_identical_to_the_eager_step(name, shape):
    _need_gpu()
    from miccai2021_cataract_semantic_segmentation_amd.graph import GraphedTrainStep
    from miccai2021_cataract_semantic_segmentation_amd.utils.metrics import t_get_confusion_matrix
    B, H, W = shape
    model, crit, opt, batches = _setup(name, B, H, W)
    fp = model.flat()
    w0 = fp.flat.clone()
    bufs0 = [b.clone() for b in model.buffers()]
    lrs = [1e-3, 1e-3, 5e-4, 2.5e-4, 2.5e-4]
    # ---- eager
    cm_e = torch.zeros((25, 25), dtype=torch.int32, device="cuda")
    losses_e = []
    for i, lr in enumerate(lrs):
        opt.param_groups[0]["lr"] = lr
        x, y = batches[i % 4]
        opt.zero_grad()
        out = model(x)
        loss = crit(*out, y)
        loss.backward()
        opt.step()
        t_get_confusion_matrix(out[1].detach(), y, cm_e)
        losses_e.append(float(loss.detach()))
    torch.cuda.synchronize()
    w_e, m_e, v_e = fp.flat.clone(), opt._m.clone(), opt._v.clone()
    bufs_e = [b.clone() for b in model.buffers()]
    logits_e = out[1].detach().clone()
    # ---- the same five steps from the same state through the graph
    with torch.no_grad():
        fp.flat.copy_(w0)
        opt._m.zero_()
        opt._v.zero_()
        for b, s in zip(model.buffers(), bufs0):
            b.copy_(s)
    opt._steps = 0
    cm_g = torch.zeros((25, 25), dtype=torch.int32, device="cuda")
    step = GraphedTrainStep(model, lambda o, l: crit(*o, l), opt, *batches[0], confusion=cm_g)
    assert opt._steps == 0 and torch.equal(fp.flat, w0) and int(cm_g.sum()) == 0          # the warm-up left no trace
    losses_g = []
    for i, lr in enumerate(lrs):
        opt.param_groups[0]["lr"] = lr
        losses_g.append(float(step(*batches[i % 4])))
    torch.cuda.synchronize()
    assert losses_g == losses_e
    assert torch.equal(fp.flat, w_e) and torch.equal(opt._m, m_e) and torch.equal(opt._v, v_e)
    for b, s in zip(model.buffers(), bufs_e):
        assert torch.equal(b, s)
    assert torch.equal(cm_g, cm_e) and int(cm_e.sum()) > 0
    assert torch.equal(step.outputs[1].detach(), logits_e)
    assert opt._steps == len(lrs) and step.replays == len(lrs)
    sd = model.state_dict()                                     # lazily flushed BatchNorm step counters follow the replays
    k = [k for k in sd if k.endswith("num_batches_tracked")][0]
    assert int(sd[k]) == 2 * len(lrs)
    # an eager step after the graph keeps working (and needs its zero_grad, as after any backward)
    opt.zero_grad()
    loss = crit(*model(batches[1][0]), batches[1][1])
    loss.backward()
    opt.step()
    torch.cuda.synchronize()
    assert torch.isfinite(loss)


def test_segmented_replay_cuts_at_bucket_boundaries_and_is_bit_identical():
    """data parallel, default execution mode: with a reducer attached the captured step is CUT where the backward tape has released whole
    gradient buckets (graph.GraphedTrainStep, segmented replay) so that each bucket's all_reduce can be launched between two hipGraph
    replays, under the rest of the backward pass.  Here: no process group (a world of one: GradSync launches nothing), so this pins the
    chain itself -- several graphs from one private pool, the backward tape driven on the calling thread (EngineNet.backward_from) --
    against the eager loop: losses, parameters, Adam moments, BatchNorm statistics, confusion matrix bit for bit over changing batches
    and learning rates; every bucket is released exactly once, most of them before the last backward graph."""
    _need_gpu()
    from miccai2021_cataract_semantic_segmentation_amd import dist as D
    from miccai2021_cataract_semantic_segmentation_amd.graph import GraphedTrainStep
    from miccai2021_cataract_semantic_segmentation_amd.utils.metrics import t_get_confusion_matrix
    model, crit, opt, batches = _setup("ocrnet_hrnet48", 2, 128, 192)
    fp = model.flat()
    w0 = fp.flat.clone()
    bufs0 = [b.clone() for b in model.buffers()]
    lrs = [1e-3, 5e-4, 2.5e-4]
    cm_e = torch.zeros((25, 25), dtype=torch.int32, device="cuda")
    losses_e = []
    for i, lr in enumerate(lrs):
        opt.param_groups[0]["lr"] = lr
        x, y = batches[i % 4]
        opt.zero_grad()
        out = model(x)
        loss = crit(*out, y)
        loss.backward()
        opt.step()
        t_get_confusion_matrix(out[1].detach(), y, cm_e)
        losses_e.append(float(loss.detach()))
    torch.cuda.synchronize()
    w_e, m_e, v_e = fp.flat.clone(), opt._m.clone(), opt._v.clone()
    bufs_e = [b.clone() for b in model.buffers()]
    g_e = fp.grad.clone()
    with torch.no_grad():
        fp.flat.copy_(w0)
        opt._m.zero_()
        opt._v.zero_()
        for b, s in zip(model.buffers(), bufs0):
            b.copy_(s)
    opt._steps = 0
    assert D.attach(model, bucket_bytes=16 << 20) == 1.0
    sync = model._grad_sync
    cm_g = torch.zeros((25, 25), dtype=torch.int32, device="cuda")
    step = GraphedTrainStep(model, lambda o, l: crit(*o, l), opt, *batches[0], confusion=cm_g, segment_bytes=32 << 20)
    assert step.split and model._grad_sync is None and step.graph is None
    nb = len(sync.buckets)
    assert nb >= 10 and len(step.graphs) >= 4 and step.tail_graph is not None
    assert sorted(b for r in step.releases for b in r) == list(range(nb))
    rep = step.overlap_report()
    assert rep["graphs_per_step"] == len(step.graphs) + 1 and rep["buckets_launched_before_the_last_backward_graph"] >= nb // 2
    assert 0 in step.releases[-1]                       # the tail bucket (stem: lowest offsets) is ready last
    losses_g = []
    for i, lr in enumerate(lrs):
        opt.param_groups[0]["lr"] = lr
        losses_g.append(float(step(*batches[i % 4])))
    torch.cuda.synchronize()
    assert [b for _, b in step.launch_log] == [b for r in step.releases for b in r]
    assert [i for i, _ in step.launch_log] == sorted(i for i, _ in step.launch_log)
    assert losses_g == losses_e
    assert torch.equal(fp.grad, g_e)
    assert torch.equal(fp.flat, w_e) and torch.equal(opt._m, m_e) and torch.equal(opt._v, v_e)
    for b, s in zip(model.buffers(), bufs_e):
        assert torch.equal(b, s)
    assert torch.equal(cm_g, cm_e) and int(cm_e.sum()) > 0
    assert opt._steps == len(lrs) and sync.steps == len(lrs)
    step.release()
    assert model._grad_sync is sync and not model._keep_pass
    opt.zero_grad()                                     # the eager loop goes on (reducer back on the tape)
    loss = crit(*model(batches[1][0]), batches[1][1])
    loss.backward()
    opt.step()
    torch.cuda.synchronize()
    assert torch.isfinite(loss) and sync.steps == len(lrs) + 1


def test_manager_epochs_through_the_graph_equal_the_eager_loop(tmp_path):
    """config['train']['hip_graph'] = True: the manager records its step once per epoch and replays it -- two epochs (the second at the
    scheduler's next learning rate) give the history, the weights and the BatchNorm statistics of the eager loop, bit for bit"""
    _need_gpu()
    from miccai2021_cataract_semantic_segmentation_amd.managers import OCRNetManager, SyntheticCataractDataset

    def run(hip_graph, sub):
        cfg = {"name": "t", "mode": "training", "manager": "OCRNet", "log_path": str(tmp_path / sub),
               "graph": {"model": "OCRNet", "backbone": "resnet50", "out_stride": 8, "pretrained": False},
               "data": {"experiment": 2, "batch_size": 2},
               "loss": {"name": "TwoScaleLoss", "interm": {"name": "LovaszSoftmax", "args": [], "weight": 0.4},
                        "final": {"name": "LovaszSoftmax", "args": [], "weight": 1.0}},
               "train": {"learning_rate": 1e-3, "epochs": 2, "hip_graph": hip_graph}, "log_every_n_epochs": 1, "seed": 0}
        torch.manual_seed(123)                                  # (the model's default initialisation draws from the global generator)
        m = OCRNetManager(cfg, SyntheticCataractDataset(8, 64, 96, 17, seed=1), SyntheticCataractDataset(2, 64, 96, 17, seed=2))
        m.train()
        torch.cuda.synchronize()
        return m.history, m.model.flat().flat.clone(), [b.clone() for b in m.model.buffers()], m.optimiser._steps
    h_e, w_e, b_e, s_e = run(False, "eager")
    h_g, w_g, b_g, s_g = run(True, "graph")
    assert s_e == s_g == 8
    assert [(r["train_loss"], r["train_miou"], r["lr"]) for r in h_e] == [(r["train_loss"], r["train_miou"], r["lr"]) for r in h_g]
    assert [r.get("valid_miou") for r in h_e] == [r.get("valid_miou") for r in h_g]
    assert torch.equal(w_e, w_g)
    for a, b in zip(b_e, b_g):
        assert torch.equal(a, b)
