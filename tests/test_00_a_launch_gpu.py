"""The launch contract of `bench.py --gpus N` and RCCL on real hardware (one MI355X is what the test box has).

Sorts before every other GPU test: both tests start FRESH child processes and this pytest process must not have initialised the
GPU before (a GPU-initialised process must not exec on this pool; only torch.cuda.device_count() is called here).

(a) `python bench.py --gpus 2 ...` with NO torchrun environment launches its two ranks by itself (bench.self_launch) and relays ONE JSON
    line; both ranks share GPU 0 over gloo (CATSEG_DIST_BACKEND=gloo, the labelled functional artefact): world_seen_by_backend == 2.
(b) backend 'nccl' (RCCL) with world_size 1: the bucketed gradient exchange forced to launch through a real HRNet-W48 backward
    (tests/_rccl_worker.py) -- gradients bit-identical to the run without a reducer, exposed wait recorded.
"""
import json
import os
import subprocess
import sys

import pytest
import torch

pytestmark = pytest.mark.gpu
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _env(**kw):
    env = {k: v for k, v in os.environ.items() if k not in ("RANK", "LOCAL_RANK", "WORLD_SIZE", "MASTER_ADDR", "MASTER_PORT")}
    env.update(HSA_ENABLE_IPC_MODE_LEGACY="0", **kw)
    return env


def test_bench_gpus2_launches_its_own_ranks():
    if torch.cuda.device_count() == 0:
        pytest.skip("needs a GPU")
    p = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--gpus", "2", "--steps", "2", "--warmup", "1", "--batch", "2",
                        "--no-cpu-baseline", "--no-side-figures"], env=_env(CATSEG_DIST_BACKEND="gloo"),
                       stdout=subprocess.PIPE, stderr=subprocess.PIPE, timeout=1500)
    out, err = p.stdout.decode(errors="replace"), p.stderr.decode(errors="replace")
    assert p.returncode == 0, err[-4000:]
    lines = [ln for ln in out.splitlines() if ln.strip()]
    assert len(lines) == 1, out[-2000:]                                   # ONE JSON line on stdout, nothing else
    r = json.loads(lines[0])
    assert r["n_gpus"] == 2 and r["steps"] == 2 and r["warmup"] == 1 and r["unit"] == "frames/s" and r["scaling"] == "weak"
    assert r["config"]["global_batch"] == 4 and r["config"]["parallelism"] == "dp2"
    assert r["comm"]["world_seen_by_backend"] == 2 and r["comm"]["backend"] == "gloo" and "FUNCTIONAL ARTEFACT" in r["comm"]["note"]
    assert r["comm"]["buckets"] >= 3 and r["comm"]["bytes_reduced_per_step"] > 250e6           # the whole flat gradient (292.7 MB)
    assert r["value"] > 0 and abs(r["value"] - 4 / (r["ms_per_step"] * 1e-3)) < 1e-6 * r["value"]
    assert r["roofline"] is not None and r["cpu_baseline"] is None        # (the CPU baseline is an N = 1 figure)


def test_bench_infer_gpus2_shards_frames_and_sums_the_confusion_matrices():
    """BASELINE config 5's launch path: `bench.py --infer --gpus 2` starts its two ranks itself, each rank scores its own frames
    (frame-sharded, no exchange in the timed region), the confusion matrices are summed once at the end (the reference scores ONE
    matrix over the whole set, managers/BaseManager.py:640-688).  Two ranks on the one GPU over gloo = the labelled functional artefact."""
    if torch.cuda.device_count() == 0:
        pytest.skip("needs a GPU")
    p = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--infer", "--gpus", "2", "--steps", "2", "--warmup", "1", "--batch", "2",
                        "--height", "256", "--width", "384", "--no-cpu-baseline"], env=_env(CATSEG_DIST_BACKEND="gloo"),
                       stdout=subprocess.PIPE, stderr=subprocess.PIPE, timeout=1200)
    out, err = p.stdout.decode(errors="replace"), p.stderr.decode(errors="replace")
    assert p.returncode == 0, err[-4000:]
    lines = [ln for ln in out.splitlines() if ln.strip()]
    assert len(lines) == 1, out[-2000:]
    r = json.loads(lines[0])
    assert r["n_gpus"] == 2 and r["unit"] == "frames/s" and r["scaling"] == "weak" and r["steps"] == 2
    assert r["config"]["global_batch"] == 4 and r["config"]["parallelism"] == "dp2 (frame sharded)"
    cmx = r["config"]["confusion_matrix"]
    # warm-up + timed + (rank 0 only) three instrumented steps all score into the matrix: at least (1 + 2) steps x 2 ranks of labelled pixels
    assert cmx["summed_over_ranks"] and cmx["pixels_scored_all_ranks"] >= 2 * 3 * cmx["labelled_pixels_per_step_rank0"] * 0.8
    assert cmx["pixels_scored_all_ranks"] > 3 * cmx["labelled_pixels_per_step_rank0"] * 1.5      # ... more than one rank's share
    assert r["value"] > 0 and abs(r["value"] - 4 / (r["ms_per_step"] * 1e-3)) < 1e-6 * r["value"]


def test_bench_refuses_a_group_that_is_not_rccl():
    """a multi-GPU line must be an RCCL line: without the explicit CATSEG_DIST_BACKEND=gloo the two-ranks-on-one-GPU job must fail
    (RCCL refuses two ranks on one device) instead of printing a number, and the launcher must pass the failure on"""
    if torch.cuda.device_count() != 1:
        pytest.skip("needs exactly one GPU (on a multi-GPU box the same command is a valid RCCL run)")
    p = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--gpus", "2", "--steps", "1", "--warmup", "0", "--batch", "1",
                        "--no-cpu-baseline", "--no-side-figures", "--no-roofline"], env=_env(NCCL_DEBUG="WARN"),
                       stdout=subprocess.PIPE, stderr=subprocess.PIPE, timeout=900)
    assert p.returncode != 0
    assert not [ln for ln in p.stdout.decode(errors="replace").splitlines() if ln.startswith("{")]


def test_rccl_world1_forced_gradient_exchange(tmp_path):
    if torch.cuda.device_count() == 0:
        pytest.skip("needs a GPU")
    out = tmp_path / "rccl.json"
    import socket
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    port = s.getsockname()[1]
    s.close()
    p = subprocess.run([sys.executable, os.path.join(ROOT, "tests", "_rccl_worker.py"), "--out", str(out)],
                       env=_env(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port)), stdout=subprocess.PIPE, stderr=subprocess.STDOUT, timeout=1200)
    assert p.returncode == 0, p.stdout.decode(errors="replace")[-4000:]
    r = json.loads(out.read_text())
    st = r["stats"]
    assert r["deterministic"] and r["bit_identical"] and r["loss_equal"] and r["gather_ok"]
    assert st["backend"] == "nccl" and st["world_seen_by_backend"] == 1 and st["forced_in_world_of_one"]
    assert st["comm_stream_priority"] == "high"
    assert st["buckets"] >= 5 and st["bytes_reduced_per_step"] >= 0.99 * r["flat_bytes"]
    assert st["steps"] == 3 and st["exposed_wait_ms"] >= 0.0 and st["host_wait_ms"] >= 0.0
    assert r["grad_norm"] > 0 and r["scale"] == 1.0
    # the graph replay with the RCCL group (and its watchdog thread) alive: same gradients, one forced exchange per replayed step
    assert r["graph_bit_identical"] and r["graph_steps_reduced"] == 2
    # ... in the segmented form: >= 4 buckets went to RCCL before the LAST backward graph was replayed (they overlap the rest of the
    # backward pass), every bucket exactly once per step, the whole flat gradient per step
    ov, log = r["graph_overlap"], r["graph_launch_log"]
    last = r["graph_backward_graphs"] - 1
    assert r["graph_backward_graphs"] >= 4 and ov["graphs_per_step"] == r["graph_backward_graphs"] + 1
    assert sum(1 for i, _ in log if i < last) >= 4 and ov["buckets_launched_before_the_last_backward_graph"] >= 4
    assert sorted(b for _, b in log) == list(range(st["buckets"]))
    assert r["graph_early_launches_per_step"] >= 4 and r["graph_bytes_per_step"] >= 0.99 * r["flat_bytes"]
    assert r["graph_exposed_wait_ms"] >= 0.0
    print("segmented graph replay over RCCL:", json.dumps(ov))
    print("RCCL world-1 forced exchange:", json.dumps(st))
