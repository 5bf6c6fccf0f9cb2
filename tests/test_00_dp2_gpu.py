"""Data parallelism on real hardware without an 8-GPU node: two FRESH child processes (one rank each, gloo rendezvous on
127.0.0.1) share GPU 0 and run tests/_dp_worker.py.  The file sorts first so that the children are started before this
pytest process has initialised the GPU (a GPU-initialised process must not exec; only device_count() is called here).

Checked: the bucketed all-reduce launched from the backward tape (EngineNet._grad_sync) + FusedAdam(grad_scale = 1/world):
  * the reduced flat gradient is bit-identical on both ranks and equals g_0 + g_1 of the two single-process shard runs,
  * the parameters after the Adam step are bit-identical on both ranks and equal to a single-process Adam step on the
    mean gradient (same kernel, same inputs),
  * the manager loop under WORLD_SIZE = 2: ranks train on disjoint frames with equal step counts, logged metrics are
    global, BatchNorm running statistics are synchronised before validation / checkpointing, parameters stay identical."""
import os
import socket
import subprocess
import sys

import pytest
import torch

pytestmark = pytest.mark.gpu
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def test_two_rank_data_parallel_step_on_one_gpu(tmp_path):
    if torch.cuda.device_count() == 0:
        pytest.skip("needs a GPU")
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    port = s.getsockname()[1]
    s.close()
    procs = []
    for rank in range(2):
        env = dict(os.environ, RANK=str(rank), LOCAL_RANK="0", WORLD_SIZE="2", MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port),
                   CATSEG_DIST_BACKEND="gloo", HSA_ENABLE_IPC_MODE_LEGACY="0")
        procs.append(subprocess.Popen([sys.executable, os.path.join(ROOT, "tests", "_dp_worker.py"), "--out", str(tmp_path)],
                                      env=env, stdout=subprocess.PIPE, stderr=subprocess.STDOUT))
    logs = []
    for p in procs:
        try:
            o, _ = p.communicate(timeout=600)
        except subprocess.TimeoutExpired:
            for q in procs:
                q.kill()
            raise
        logs.append(o.decode(errors="replace"))
    assert all(p.returncode == 0 for p in procs), "\n----\n".join(l[-3000:] for l in logs)
    r0 = torch.load(tmp_path / "rank0.pt", weights_only=False)
    r1 = torch.load(tmp_path / "rank1.pt", weights_only=False)
    # ---- gradient exchange
    assert r0["buckets"] >= 3 and r0["scale"] == 0.5
    assert torch.equal(r0["init_flat"], r1["init_flat"])
    assert not torch.equal(r0["g_single"], r1["g_single"])                       # different shards
    want = r0["g_single"] + r1["g_single"]
    assert torch.equal(r0["g_reduced"], r1["g_reduced"])                         # bit-for-bit across ranks
    assert torch.equal(r0["g_reduced"], want)                                    # = sum of the two single-process shard runs
    assert torch.equal(r0["flat_after"], r1["flat_after"])
    assert r0["second_backward_raised"] and r1["second_backward_raised"]
    # BatchNorm running statistics: local (different) before the sync, the mean afterwards
    assert not torch.equal(r0["bn_before_sync"], r1["bn_before_sync"])
    assert torch.equal(r0["bn_after_sync"], r1["bn_after_sync"])
    assert torch.allclose(r0["bn_after_sync"], 0.5 * (r0["bn_before_sync"] + r1["bn_before_sync"]), atol=1e-7)
    # ---- the same Adam step in THIS process on the mean gradient (children are done: the GPU may be used now)
    from miccai2021_cataract_semantic_segmentation_amd import ops
    p = r0["init_flat"].cuda()
    m, v = torch.zeros_like(p), torch.zeros_like(p)
    ops.adam_step(p, want.cuda(), m, v, 1e-3, 1, grad_scale=0.5)
    assert torch.equal(p.cpu(), r0["flat_after"])
    # ---- manager loop
    i0, i1 = r0["train_indices"], r1["train_indices"]
    assert len(i0) == len(i1) == 2 * 4                                           # 10 frames -> 8 per epoch over 2 ranks x bs 2, 2 epochs
    assert not set(i0[:4]) & set(i1[:4]) and not set(i0[4:]) & set(i1[4:])       # disjoint frames within each epoch
    assert r0["steps"] == r1["steps"] == 4
    assert torch.equal(r0["mgr_flat"], r1["mgr_flat"]) and torch.equal(r0["mgr_bn"], r1["mgr_bn"])
    for h0, h1 in zip(r0["history"], r1["history"]):                             # global (all-reduced) metrics on every rank
        assert h0["train_loss"] == h1["train_loss"] and h0["train_miou"] == h1["train_miou"] and h0["valid_miou"] == h1["valid_miou"]
    assert r0["metrics"]["best_miou"] == r1["metrics"]["best_miou"]
    assert os.path.exists(os.path.join(str(tmp_path), "logs"))
