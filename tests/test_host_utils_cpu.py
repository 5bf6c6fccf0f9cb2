"""Host-side utilities against fixtures generated from the reference (CPU only)."""
import os

import numpy as np
import torch

from miccai2021_cataract_semantic_segmentation_amd.utils import LRFcts, RepeatFactorSampler


def test_repeat_factor_sampler_matches_reference(golden):
    g = golden("rfs")
    for exp in (1, 2, 3):
        s = RepeatFactorSampler(g["e%d_presence" % exp], g["e%d_cmap" % exp], g["e%d_class_keys" % exp], 0.15)
        for k in g["e%d_class_keys" % exp]:
            assert abs(s.class_repeat_factors[int(k)] - g["e%d_class_rf" % exp][int(k)]) < 1e-6
        np.testing.assert_allclose(s.repeat_factors.numpy(), g["e%d_image_rf" % exp], rtol=1e-6)
        n1 = len(s)
        e1 = list(s)
        n2 = len(s)
        e2 = list(s)
        assert [n1, n2] == list(g["e%d_lens" % exp])
        assert e1 == list(g["e%d_epoch1" % exp]) and e2 == list(g["e%d_epoch2" % exp])


def test_repeat_factor_sampler_rank_shards():
    rng = np.random.default_rng(0)
    pres = rng.random((200, 6)) < 0.3
    pres[:, 0] = True
    shards = []
    for r in range(2):
        s = RepeatFactorSampler(pres, [0, 1, 1, 2, 3, 3], [0, 1, 2, 3], 0.15, rank=r, world=2)
        n = len(s)
        e = list(s)
        assert len(e) == n
        shards.append(e)
    full = RepeatFactorSampler(pres, [0, 1, 1, 2, 3, 3], [0, 1, 2, 3], 0.15)
    len(full)
    order = list(full)
    m = len(order) // 2 * 2
    assert shards[0] == order[0:m:2] and shards[1] == order[1:m:2]


def test_lr_schedule_matches_reference(golden):
    g = golden("metrics")
    f = LRFcts({"epochs": 50, "learning_rate": 1e-4, "lr_fct": "exponential", "lr_params": None, "lr_restarts": [],
                "lr_restart_vals": 1, "lr_batchwise": False}, [], 50)
    np.testing.assert_allclose([f(e) for e in range(50)], g["lr_mult"], rtol=1e-12)
    opt = torch.optim.SGD([torch.nn.Parameter(torch.zeros(1))], lr=1e-4)
    sch = torch.optim.lr_scheduler.LambdaLR(opt, lr_lambda=f)
    opt.step()
    sch.step()
    assert abs(opt.param_groups[0]["lr"] - 0.98e-4) < 1e-12
    cos = LRFcts({"epochs": 10, "lr_fct": "cosine", "lr_params": None, "lr_restarts": [5], "lr_restart_vals": 0.5,
                  "lr_batchwise": False}, [5], 10)
    assert abs(cos(0) - 1) < 1e-12 and abs(cos(5) - 0.5) < 1e-12 and cos(4) < cos(1)


def test_pretrained_trunk_loading(tmp_path, monkeypatch):
    """config['pretrained'] (default True, models/OCR.py:44 of the reference) reads a torchvision-format checkpoint from disk;
    without one it warns loudly instead of silently training from scratch"""
    import pytest
    from oracle import resnet_tv
    from miccai2021_cataract_semantic_segmentation_amd.models import DeepLabv3, EncDec, OCRNet
    monkeypatch.delenv("CATSEG_PRETRAINED_DIR", raising=False)
    with pytest.warns(RuntimeWarning, match="RANDOMLY INITIALISED"):
        OCRNet({"backbone": "resnet50", "out_stride": 8}, 3)                # 'pretrained' absent -> True
    r50 = resnet_tv.resnet50()
    torch.save(r50.state_dict(), tmp_path / "resnet50.pth")
    monkeypatch.setenv("CATSEG_PRETRAINED_DIR", str(tmp_path))
    m = DeepLabv3({"backbone": "resnet50", "out_stride": 8, "pretrained": True}, 2)
    assert torch.equal(m.backbone["layer4"][2].conv3.weight, r50.layer4[2].conv3.weight)
    assert torch.equal(m.backbone["bn1"].running_var, r50.bn1.running_var)
    r18 = resnet_tv.resnet18()
    torch.save(r18.state_dict(), tmp_path / "r18.pth")
    e = EncDec({"encoder": {"model": "ResNet18", "pretrained": True, "pretrained_path": str(tmp_path / "r18.pth")},
                "decoder": {"model": "UPerNet"}}, 1)
    assert torch.equal(e.enc_model.layer2[0].conv1.weight, r18.layer2[0].conv1.weight)
    torch.save({"conv1.weight": r18.conv1.weight}, tmp_path / "bad.pth")
    with pytest.raises(RuntimeError, match="lacks"):
        EncDec({"encoder": {"model": "ResNet18", "pretrained": True, "pretrained_path": str(tmp_path / "bad.pth")},
                "decoder": {"model": "UPerNet"}}, 1)


def test_fused_adam_state_dict_is_torch_adam_format():
    """'optimiser_state_dict' of a checkpoint (BaseManager.py:471-495) in torch.optim.Adam's format, both directions"""
    from miccai2021_cataract_semantic_segmentation_amd.engine import FlatParams
    from miccai2021_cataract_semantic_segmentation_amd.models import EncDec
    from miccai2021_cataract_semantic_segmentation_amd.optim import FusedAdam
    net = EncDec({"encoder": {"model": "ResNet18", "pretrained": False}, "decoder": {"model": "UPerNet"}}, 1)
    ref = torch.optim.Adam(net.parameters(), lr=1e-4)
    for p in net.parameters():
        p.grad = torch.randn_like(p)
    ref.step()
    ref.step()
    sd_ref = ref.state_dict()                                   # written by the reference's optimiser
    opt = FusedAdam(net, lr=1.0)
    opt.load_state_dict(sd_ref)
    assert opt._steps == 2 and opt.param_groups[0]["lr"] == 1e-4
    fp = net.flat()
    for i, p in enumerate(fp.params):
        assert torch.equal(FlatParams._view(opt._m, fp.offsets[id(p)], p), sd_ref["state"][i]["exp_avg"])
        assert torch.equal(FlatParams._view(opt._v, fp.offsets[id(p)], p), sd_ref["state"][i]["exp_avg_sq"])
    sd = opt.state_dict()                                       # ... and one written here loads under torch.optim.Adam
    back = torch.optim.Adam(net.parameters(), lr=5.0)
    back.load_state_dict(sd)
    assert back.param_groups[0]["lr"] == 1e-4
    w = net.enc_model.conv1.weight
    assert torch.equal(back.state[w]["exp_avg"], ref.state[w]["exp_avg"]) and float(back.state[w]["step"]) == 2
    import pytest
    with pytest.raises(ValueError, match="unrecognised"):
        opt.load_state_dict({"foo": 1})


def test_manager_registry_resolves_every_shipped_config_name():
    """main.py:46 resolves config['manager'] + 'Manager'; configs/*.json of the reference name these five"""
    from miccai2021_cataract_semantic_segmentation_amd import managers
    for name in ("OCRNet", "DeepLabv3Plus", "DeepLabv3", "EncDec"):
        assert issubclass(getattr(managers, name + "Manager"), managers.BaseManager)


def test_second_backward_and_shared_module_are_rejected():
    from miccai2021_cataract_semantic_segmentation_amd import engine
    cx = engine.Ctx(train=True, record=True)
    p = torch.nn.Parameter(torch.zeros(3))
    cx.claim(p, None)
    import pytest
    with pytest.raises(NotImplementedError, match="applied twice"):
        cx.claim(p)
    engine.Ctx(train=False, record=False).claim(p, p)           # nothing is recorded in inference: no restriction


def test_bf16x3_layer_selection():
    """which convolutions take the split-precision kernels, and which of those read blocked planes (ops._b3_eligible /
    _b3_blocked_ok: host logic of DESIGN 4.1c)"""
    from miccai2021_cataract_semantic_segmentation_amd import ops
    assert ops.PRECISION == "bf16x3" and ops.B3_BLOCKED
    rows = 8 * 136 * 240
    assert ops._b3_eligible(rows, 512, 9, 720)                         # OCRNet-HRNet head 3x3 720 -> 512
    assert ops._b3_blocked_ok(512, 720, rows, 512, 9)
    assert not ops._b3_eligible(rows, 48, 9, 48)                       # HRNet branch 3x3 48 -> 48: fp32 kernels
    assert not ops._b3_eligible(8 * 17 * 30, 384, 9, 384)              # 384-channel branch: too few tiles
    assert ops._b3_eligible(rows, 512, 1, 1024)                        # the wide 1x1 of the OCR head at stride 4 ...
    assert ops._b3_eligible(rows, 1024, 1, 512)                        # ... and its backward-data (N = 1024, K = 512)
    assert ops._b3_eligible(8 * 68 * 120, 512, 1, 1024)                # the same layer at stride 8 (OCRNet-R50; 65 280 pixels): since round 3
    assert ops._b3_eligible(8 * 68 * 120, 2048, 1, 512)                # ResNet50 layer 4, 1x1 512 -> 2048 on the stride-8 map
    assert not ops._b3_eligible(4 * 68 * 120, 512, 1, 1024)            # half the batch: fp32
    assert not ops._b3_eligible(8 * 68 * 120, 256, 1, 1024)            # an extent of 256 (ResNet50 layer 3): fp32
    assert not ops._b3_eligible(rows, 256, 1, 512)                     # smaller 1x1 layers: fp32
    assert ops._b3_eligible(8 * 68 * 120, 512, 9, 2048)                # OCRNet-R50 conv_high_map
    assert not ops._b3_eligible(8 * 68 * 120, 512, 9, 2044)            # Cin % 8
    assert not ops._b3_blocked_ok(192, 720, rows, 192, 9)              # <= 192 columns: 8-wave tiles, planar planes
    assert not ops._b3_blocked_ok(512, 712, rows, 512, 9)              # Cin % 16
    big = 4 * 272 * 480
    assert ops._b3_eligible(big, 512, 9, 2048)
    assert not ops._b3_blocked_ok(512, 2048, big, 512, 9)              # three blocked planes = 6.4 GB > 4 GB: planar / cut batch
    assert ops._b3_blocked_ok(512, 2048, big // 2, 512, 9)


def test_gaussian_box_parameters_match_the_oracle():
    """host side of the GPU blur: (radius, ww, fw) of Pillow's BoxBlur.c for ImageFilter.GaussianBlur(3..6)"""
    from oracle import augment as A
    from miccai2021_cataract_semantic_segmentation_amd.utils.augment import gaussian_box_params
    for r in (1, 2, 3, 4, 5, 6, 9):
        assert gaussian_box_params(r) == A.box_weights(A.gaussian_box_radius(r)), r


def test_amax_record_scopes():
    """ops.AmaxScope: zeroed 2 KB records handed out in order from chunks that are replaced, not recycled -- a record stays
    valid for as long as a tensor refers to it, whatever later forward passes allocate"""
    import torch
    from miccai2021_cataract_semantic_segmentation_amd import ops
    saved = ops.AMAX_SCOPE_RECORDS
    try:
        ops.AMAX_SCOPE_RECORDS = 4
        dev = torch.device("cpu")
        a = ops.AmaxScope(dev)
        recs = [a.new() for _ in range(6)]                     # rolls over into a second chunk after four
        assert all(r.numel() == ops.AMAX_WORDS == 512 and r.dtype == torch.int32 and int(r.abs().sum()) == 0 for r in recs)
        assert len({r.data_ptr() for r in recs}) == 6
        assert recs[3].data_ptr() - recs[2].data_ptr() == 2048       # include/catseg.h: CATSEG_AMAX_RECORD_BYTES
        recs[0][0] = 7
        b = ops.AmaxScope(dev)                                 # a second forward pass: its own chunks
        ops.set_amax_scope(b)
        r = ops.new_amax(dev)
        assert int(r.abs().sum()) == 0 and int(recs[0][0]) == 7
        ops.set_amax_scope(a)                                  # the first pass's backward continues in its own scope
        assert ops.new_amax(dev).data_ptr() == recs[5].data_ptr() + 2048
    finally:
        ops.AMAX_SCOPE_RECORDS = saved
        ops.set_amax_scope(None)


def test_roctx_ranges_are_optional_and_balanced():
    """CATSEG_ROCTX=1: every timed C-ABI call is bracketed by roctxRangePush(kind) / roctxRangePop(); off by default"""
    import subprocess
    import sys
    code = ("import os, sys; sys.path.insert(0, %r); from miccai2021_cataract_semantic_segmentation_amd import ops\n"
            "assert (ops._roctx is not None) == (os.environ.get('CATSEG_ROCTX') == '1')\n"
            "if ops._roctx is not None:\n"
            "    with ops._Timed('fwd', 1.0):\n"
            "        pass\n"
            "print('ok')\n") % os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    for flag in ("0", "1"):
        env = dict(os.environ, CATSEG_ROCTX=flag)
        out = subprocess.run([sys.executable, "-c", code], env=env, capture_output=True, text=True, timeout=120)
        assert out.returncode == 0 and "ok" in out.stdout, out.stderr[-500:]


def test_tn_split_selection():
    """ops.tn_splits: the OCR head's reductions over all pixels (bench shape 136 x 240 = 32 640 rows -> 32 chunks of 1 020; the ResNet50 models'
    68 x 120 = 8 160 -> 15 of 544); short reductions, large results and row counts without a suitable divisor stay one launch"""
    from miccai2021_cataract_semantic_segmentation_amd import ops
    assert ops.tn_splits(25, 512, 32640) == 32 and 32640 // 32 == 1020
    assert ops.tn_splits(25, 256, 32640) == 32
    assert ops.tn_splits(17, 512, 8160) == 15 and 8160 // 15 == 544
    assert ops.tn_splits(25, 512, 4095) == 1            # short
    assert ops.tn_splits(25, 512, 4099) == 1            # prime: no divisor
    assert ops.tn_splits(512, 512, 32640) == 1          # a large result is already output-parallel
    for K in range(4096, 40000, 977):
        s = ops.tn_splits(25, 512, K)
        assert s == 1 or (K % s == 0 and K // s >= 512 and (K // s) % 4 == 0)


def test_loader_worker_errors_reach_the_consumer_with_their_traceback():
    """a dataset error inside a forked loader worker is re-raised in the consumer with the worker's traceback (as torch's DataLoader
    does), and a worker that dies without a report is named with its exit code -- never a bare EOFError / a silent exit 0"""
    import os
    import numpy as np
    import pytest
    from miccai2021_cataract_semantic_segmentation_amd.utils.loader import _WorkerPool

    class DS:
        def __len__(self):
            return 8

        def __getitem__(self, i):
            if i == 5:
                raise ValueError("corrupt frame %d" % i)
            if i == 7:
                os._exit(9)
            return np.full((4, 6, 3), i, np.uint8), np.full((4, 6), i, np.uint8)
    slots = [(torch.empty((2, 4, 6, 3), dtype=torch.uint8).share_memory_(), torch.empty((2, 4, 6), dtype=torch.uint8).share_memory_())
             for _ in range(2)]
    pool = _WorkerPool(DS(), slots, 2)
    try:
        pool.submit(0, [1, 2], 100)
        pool.wait(100)
        assert int(slots[0][0][0, 0, 0, 0]) == 1 and int(slots[0][1][1, 0, 0]) == 2
        pool.submit(1, [3, 5], 101)
        with pytest.raises(RuntimeError) as e:
            pool.wait(101)
        assert "corrupt frame 5" in str(e.value) and "ValueError" in str(e.value) and "Traceback" in str(e.value)
    finally:
        pool.close()
    pool = _WorkerPool(DS(), slots, 2)
    try:
        pool.submit(0, [1, 7], 200)
        with pytest.raises(RuntimeError) as e:
            pool.wait(200)
        assert "died without a report" in str(e.value) and "exit code 9" in str(e.value)
    finally:
        pool.close()
    assert pool.pending == {} and pool.conns == []


def test_plan_registry_parses_one_variable_and_reads_live_values_back(monkeypatch):
    """plan.py: every route switch is a field with a default, a parser and an owner; CATSEG_PLAN overrides the field's own variable, an
    unknown field is refused, and active() reports the LIVE module attributes (what bench.py prints as config.plan)"""
    import importlib
    from miccai2021_cataract_semantic_segmentation_amd import plan
    monkeypatch.setenv("CATSEG_PLAN", "heads=bf16x3, planes_widths=96+192,segment_mb=16")
    monkeypatch.setenv("CATSEG_HEADS", "f16x2")
    monkeypatch.setenv("CATSEG_G1_MIN_ROWS", "2048")
    monkeypatch.setattr(plan, "_plan_env", None)
    assert plan.get("heads") == "bf16x3" and plan.get("planes_widths") == (96, 192) and plan.get("segment_mb") == 16.0
    assert plan.get("g1_min_rows") == 2048 and plan.get("trunk") == "f16x2" and plan.get("stem7") is True
    monkeypatch.setenv("CATSEG_PLAN", "no_such_field=1")
    monkeypatch.setattr(plan, "_plan_env", None)
    try:
        plan.get("heads")
        raise AssertionError("an unknown CATSEG_PLAN field must be refused")
    except ValueError as e:
        assert "no_such_field" in str(e)
    monkeypatch.delenv("CATSEG_PLAN")
    monkeypatch.setattr(plan, "_plan_env", None)
    ops = importlib.import_module("miccai2021_cataract_semantic_segmentation_amd.ops")
    act = ops.plan()
    assert set(act) == set(plan.FIELDS) and act["precision"] == ops.PRECISION
    saved = ops.TRUNK
    try:
        ops.TRUNK = "bf16x3"
        assert ops.plan()["trunk"] == "bf16x3" and plan.non_default().get("trunk") == "bf16x3"
    finally:
        ops.TRUNK = saved
    for name, (default, parse, owner, doc) in plan.FIELDS.items():
        assert doc and (owner is None or ":" in owner), name
        assert parse(default) == default, name            # a default survives its own parser
