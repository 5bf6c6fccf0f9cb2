"""Training / validation / inference loop with the reference managers' contract
(managers/BaseManager.py:26-741, managers/OCRNet_Manager.py:12-245) on the HIP engine.

Kept: step order ``zero_grad -> model(img.float()) -> loss -> backward -> step`` (OCRNet_Manager.py:80-90),
Adam(lr) + LambdaLR(LRFcts) stepped once per epoch (BaseManager.py:439-464, OCRNet_Manager.py:132-134),
validation with batch size 1 in eval mode under no_grad (BaseManager.py:305), best-mIoU rule on the mIoU
rounded to 4 decimals (OCRNet_Manager.py:208-223), checkpoint file names and dictionary keys
(BaseManager.py:471-495), inference with ``get_intermediate = False`` from the 'best' checkpoint (:640-688).

Changed on purpose (SURVEY.md "hard parts"): metrics stay on the device (one confusion-matrix kernel per
step, no one-hot matmul, no per-step ``.item()``); TensorBoard / matplotlib logging is out of scope; data
parallelism (absent in the reference) shards frames by rank and averages gradients over RCCL.
The dataset is any ``torch.utils.data.Dataset`` yielding ``(img float[3,H,W], lbl int[H,W], metadata)`` like
datasets/Dataset_from_df.py:69; the CaDIS cv2/PIL pipeline itself is outside the accelerated path.
"""
import datetime
import json
import pathlib

import torch
from torch.utils.data import DataLoader

from .. import dist as D
from .. import losses as _losses
from .. import models as _models
from ..optim import FusedAdam
from ..utils import LRFcts
from ..utils.metrics import t_get_confusion_matrix, t_get_mean_iou, t_get_pixel_accuracy

DEFAULTS = {"mode": "training", "debugging": False, "log_every_n_epochs": 100, "max_valid_imgs": 10, "cuda": True,
            "gpu_device": 0, "seed": 0, "tta": False, "log_path": "logs"}
TRAIN_DEFAULTS = {"epochs": 50, "lr_fct": "exponential", "lr_batchwise": False, "lr_restarts": [], "lr_restart_vals": 1,
                  "lr_params": None, "learning_rate": 1e-4}
DATA_DEFAULTS = {"batch_size": 10, "num_workers": 0, "experiment": 1}


def merge_defaults(config):
    """utils/utils.py:533-542: flat defaults + nested defaults for 'data' / 'train' ('loss' must exist)"""
    cfg = dict(DEFAULTS)
    cfg.update(config)
    for key, dflt in (("train", TRAIN_DEFAULTS), ("data", DATA_DEFAULTS)):
        merged = dict(dflt)
        merged.update(cfg.get(key, {}))
        cfg[key] = merged
    if "loss" not in cfg:
        raise KeyError("config needs a 'loss' entry")
    return cfg


class BaseManager:
    model_registry = _models
    loss_registry = _losses

    def __init__(self, configuration, train_set=None, valid_set=None, train_sampler=None):
        self.config = merge_defaults(configuration)
        self.start_epoch = self.epoch = 0
        self.best_loss = 1e10
        self.metrics = {"best_miou": 0, "best_miou_anatomies": 0, "best_miou_instruments": 0, "best_miou_rare": 0,
                        "best_miou_epoch_step": "n/a"}
        self.global_step = 0
        self.experiment = self.config["data"]["experiment"]
        self.rank, self.local_rank, self.world = D.init_from_env()
        if not torch.cuda.is_available():
            raise RuntimeError("the HIP engine needs an MI355X device (no CPU fallback)")
        self.device = torch.device("cuda", self.local_rank if self.world > 1 else self.config["gpu_device"])
        torch.cuda.set_device(self.device)
        self.run_id = self.config.get("load_checkpoint") if self.config["mode"] != "training" and "load_checkpoint" in self.config \
            else "{:%Y%m%d_%H%M%S}_e{}".format(datetime.datetime.now(), self.experiment) + \
                 ("__" + self.config["name"] if "name" in self.config else "")
        self.log_dir = pathlib.Path(self.config["log_path"]) / self.run_id
        if self.rank == 0:
            self.log_dir.mkdir(parents=True, exist_ok=True)
        # `manager_class(config)` as main.py:61 constructs it: the reference's load_data() (cv2 / pandas decoding, out of this path's scope)
        # is replaced by a callable the configuration names -- config['data']['dataset_factory'](config) -> (train_set, valid_set, sampler)
        factory = self.config["data"].get("dataset_factory")
        if train_set is None and valid_set is None and callable(factory):
            made = tuple(factory(self.config))
            train_set, valid_set, train_sampler = (made + (None, None, None))[:3]
        self.train_set, self.valid_set, self.train_sampler = train_set, valid_set, train_sampler
        self.load_model()
        self.loss = self.optimiser = self.scheduler = None
        if self.config["mode"] == "training":
            self.load_loss()
            torch.manual_seed(self.config["seed"])      # after model construction, as BaseManager.py:104-112
            self.load_optimiser()
        elif self.config["mode"] != "inference":
            raise ValueError("mode: {} is not recognized".format(self.config["mode"]))
        self.history = []

    # ------------------------------------------------------------------ construction
    def load_model(self):
        cls = getattr(self.model_registry, self.config["graph"]["model"])
        self.model = cls(config=self.config["graph"], experiment=self.experiment).to(self.device)
        if self.world > 1:
            D.broadcast_parameters(self.model)
            self.grad_scale = D.attach(self.model)
        else:
            self.grad_scale = 1.0

    def load_loss(self):
        cls = getattr(self.loss_registry, self.config["loss"]["name"])
        self.config["loss"]["experiment"] = self.experiment
        self.config["loss"]["device"] = str(self.device)
        self.loss = cls(self.config["loss"])

    def load_optimiser(self):
        tc = self.config["train"]
        self.optimiser = FusedAdam(self.model, lr=tc["learning_rate"], grad_scale=self.grad_scale)
        self.scheduler = torch.optim.lr_scheduler.LambdaLR(self.optimiser, lr_lambda=LRFcts(tc, list(tc["lr_restarts"]), tc["epochs"]))

    def loaders(self):
        """train loader (batch bs, drop_last) + validation loader (batch 1, BaseManager.py:305).  With WORLD_SIZE > 1 the
        training frames are SHARDED by rank: one shared-seed index stream per epoch (a permutation, or the given sampler's
        stream -- the repeat-factor sampler draws from a private Generator(seed=1) and is identical on every rank), each
        rank taking every world-th index: global batch = world * bs, disjoint frames, equal step counts."""
        bs = self.config["data"]["batch_size"]
        self.shard_sampler = None
        if self.world > 1:
            src = self.train_sampler if self.train_sampler is not None else len(self.train_set)
            if self.train_sampler is not None:
                D.check_unsharded(self.train_sampler)
            self.shard_sampler = D.ShardedSampler(src, self.rank, self.world, bs, seed=self.config["seed"])
            tl = DataLoader(self.train_set, batch_size=bs, sampler=self.shard_sampler, drop_last=True,
                            num_workers=0 if self.train_sampler is not None else self.config["data"]["num_workers"])
        elif self.train_sampler is not None:
            tl = DataLoader(self.train_set, batch_size=bs, sampler=self.train_sampler, drop_last=True, num_workers=0)
        else:
            tl = DataLoader(self.train_set, batch_size=bs, shuffle=True, drop_last=True,
                            num_workers=self.config["data"]["num_workers"])
        vl = DataLoader(self.valid_set, batch_size=1, shuffle=False) if self.valid_set is not None else None
        return tl, vl

    # ------------------------------------------------------------------ hooks for the subclasses
    def forward_loss(self, img, lbl):
        """returns (loss, final logits)"""
        out = self.model(img.float())
        return self.loss(out, lbl.long()), out

    def final_output(self, out):
        return out

    # ------------------------------------------------------------------ loops
    def train(self):
        self.train_loader, self.valid_loader = self.loaders()
        for self.epoch in range(self.config["train"]["epochs"]):
            self.train_one_epoch()
            if self.valid_loader is not None:
                self.validate()
        return self.metrics

    def train_one_epoch(self):
        self.model.train()
        K = self.model.num_classes
        running_cm = torch.zeros((K, K), dtype=torch.int32, device=self.device)
        loss_sum = torch.zeros((), device=self.device)
        n = 0
        if getattr(self, "shard_sampler", None) is not None:
            self.shard_sampler.set_epoch(self.epoch + self.start_epoch)
        # config['train']['hip_graph'] (build-side key, default off): the step -- zero_grad, forward, loss, backward, Adam, confusion matrix --
        # is recorded once per epoch into a hipGraph and replayed (graph.GraphedTrainStep: bit-identical to this loop, one host call per
        # step instead of ~3000 launches); batches of another shape (a ragged last batch) run the loop below
        graphed = None
        use_graph = bool(self.config.get("train", {}).get("hip_graph", False)) and isinstance(self.optimiser, FusedAdam)
        for img, lbl, _ in self.train_loader:
            img, lbl = img.to(self.device, non_blocking=True), lbl.to(self.device, non_blocking=True)
            if use_graph and graphed is None:
                from ..graph import GraphedTrainStep
                img, lbl = img.float().contiguous(), lbl.long().contiguous()
                graphed = GraphedTrainStep(self.model, None, self.optimiser, img, lbl, confusion=running_cm, forward_loss=self.forward_loss)
            if graphed is not None and img.shape == graphed.img.shape and lbl.shape == graphed.lbl.shape:
                loss_sum += graphed(img, lbl)
            else:
                if graphed is not None:
                    graphed.release()
                    graphed, use_graph = None, False
                self.optimiser.zero_grad()
                loss, out = self.forward_loss(img, lbl)
                loss.backward()
                self.optimiser.step()
                t_get_confusion_matrix(out.detach(), lbl, running_cm)      # accumulates on the device
                loss_sum += loss.detach()
            n += 1
            self.global_step += 1
        if graphed is not None:
            graphed.release()
        if self.scheduler is not None:
            self.scheduler.step()
        if self.world > 1:                       # logged training metrics are GLOBAL: one exchange per epoch
            import torch.distributed as dist
            cm64 = running_cm.to(torch.int64)
            dist.all_reduce(cm64)
            running_cm = cm64.to(torch.int32)
            cnt = torch.tensor([float(n)], device=self.device)
            dist.all_reduce(cnt)
            dist.all_reduce(loss_sum)
            n = int(cnt)
        pa, pac = t_get_pixel_accuracy(running_cm)
        miou = t_get_mean_iou(running_cm, self.experiment)
        rec = {"epoch": self.epoch + self.start_epoch, "train_loss": float(loss_sum / max(n, 1)), "train_miou": float(miou),
               "train_pixel_accuracy": float(pa), "lr": self.optimiser.param_groups[0]["lr"]}
        self.history.append(rec)
        if self.rank == 0:
            print("Epoch {epoch:03d} - loss {train_loss:.5f} - miou {train_miou:.4f} - pa {train_pixel_accuracy:.4f}".format(**rec))

    def _eval_pass(self, loader, with_loss):
        K = self.model.num_classes
        cm = torch.zeros((K, K), dtype=torch.int32, device=self.device)
        loss_sum = torch.zeros((), device=self.device)
        n = 0
        with torch.no_grad():
            for i, (img, lbl, _) in enumerate(loader):
                if self.world > 1 and i % self.world != self.rank:   # frames sharded round-robin over ranks
                    continue
                img, lbl = img.to(self.device), lbl.to(self.device)
                if with_loss:
                    loss, out = self.forward_loss(img, lbl)
                    loss_sum += loss
                else:
                    out = self.final_output(self.model(img.float()))
                t_get_confusion_matrix(out, lbl, cm)
                n += 1
        if self.world > 1:                                            # one exchange at the end
            import torch.distributed as dist
            cm64 = cm.to(torch.int64)
            dist.all_reduce(cm64)
            cm = cm64.to(torch.int32)
            cnt = torch.tensor([float(n)], device=self.device)
            dist.all_reduce(cnt)
            dist.all_reduce(loss_sum)
            n = int(cnt)
        return cm, float(loss_sum) / max(n, 1)

    def validate(self):
        self.model.eval()
        D.sync_bn_stats(self.model)      # every rank scores (and rank 0 saves) the same running statistics
        cm, valid_loss = self._eval_pass(self.valid_loader, with_loss=True)
        pa, pac = t_get_pixel_accuracy(cm)
        m_iou, m_ins, m_anat, m_rare = t_get_mean_iou(cm, self.experiment, True, rare=True)
        m_iou, m_ins, m_anat = round(float(m_iou), 4), round(float(m_ins), 4), round(float(m_anat), 4)
        step = [self.epoch + self.start_epoch, self.global_step - 1]
        if self.history:
            self.history[-1].update({"valid_loss": valid_loss, "valid_miou": m_iou})
        if m_iou > self.metrics["best_miou"]:
            self.metrics.update({"best_miou": m_iou, "best_miou_anatomies": m_anat, "best_miou_instruments": m_ins,
                                 "best_miou_rare": float(m_rare), "best_miou_epoch_step": step})
            self.save_checkpoint(is_best=True)
        if valid_loss < self.best_loss:
            self.best_loss = valid_loss
            self.metrics.update({"best_loss_miou": m_iou, "best_loss_miou_anatomies": m_anat,
                                 "best_loss_miou_instruments": m_ins, "best_loss_miou_rare": float(m_rare),
                                 "best_loss_epoch_step": step})
        if (self.epoch % self.config["log_every_n_epochs"] == 0 and self.epoch > 0) or self.epoch == self.config["train"]["epochs"] - 1:
            self.save_checkpoint(is_best=False)
        self.write_info_json()
        if self.rank == 0:
            print("Epoch {:03d} - Validation loss: {:.5f} - miou:{:.3f} - ins:{:.3f} - anat:{:.3f} - rare:{:.4f}".format(
                self.epoch + self.start_epoch, valid_loss, m_iou, m_ins, m_anat, float(m_rare)))
        return m_iou

    def infer(self):
        self.model.eval()
        if hasattr(self.model, "get_intermediate"):
            self.model.get_intermediate = False
        if hasattr(self.model, "get_features"):
            self.model.get_features = False
        self.load_checkpoint("best")
        loader = DataLoader(self.valid_set, batch_size=1, shuffle=False)
        net = self.model
        if self.config["tta"]:                       # BaseManager.py:652-660: hflip x 5 scales, mean-merged
            from ..utils.tta import SegmentationTTA
            self.model = SegmentationTTA(net)
            self.model.num_classes = net.num_classes
        try:
            cm, _ = self._eval_pass(loader, with_loss=False)
        finally:
            self.model = net
        m = t_get_mean_iou(cm, self.experiment, True, rare=True)
        return tuple(float(v) for v in m)

    # ------------------------------------------------------------------ checkpoints (BaseManager.py:471-529)
    def save_checkpoint(self, is_best):
        """BaseManager.py:471-495.  Called by every rank (validate() synchronised the BatchNorm running statistics just
        before); only rank 0 writes."""
        if self.rank != 0:
            return
        base = self.log_dir / "chkpts"
        base.mkdir(exist_ok=True)
        state = {"global_step": self.global_step, "epoch": self.start_epoch + self.epoch,
                 "model_state_dict": {k: v.detach().cpu().contiguous() for k, v in self.model.state_dict().items()},
                 "optimiser_state_dict": self.optimiser.state_dict(), "best_loss": self.best_loss,
                 "best_miou": self.metrics["best_miou"], "is_best": is_best}
        if self.scheduler is not None:
            state["scheduler_state_dict"] = self.scheduler.state_dict()
        name = "chkpt_best.pt" if is_best else "chkpt_epoch_{:03d}.pt".format(state["epoch"])
        torch.save(state, base / name)

    def load_checkpoint(self, chkpt_type):
        names = sorted(f.name for f in (self.log_dir / "chkpts").iterdir())
        name = "chkpt_best.pt"
        if chkpt_type == "best":
            if name not in names:
                raise ValueError("No checkpoint of type 'best' found.")
        elif chkpt_type == "last":
            if "chkpt_epoch_" not in names[-1]:
                raise ValueError("No checkpoint of type 'last' found.")
            name = names[-1]
        ck = torch.load(str(self.log_dir / "chkpts" / name), map_location=self.device, weights_only=False)
        self.model.load_state_dict(ck["model_state_dict"], strict=False)
        if self.config["mode"] == "training":
            self.optimiser.load_state_dict(ck["optimiser_state_dict"])
            if "scheduler_state_dict" in ck:
                self.scheduler.load_state_dict(ck["scheduler_state_dict"])   # (the reference loads the optimiser dict here: a bug)
            self.start_epoch, self.global_step = ck["epoch"], ck["global_step"]
        self.best_loss = ck["best_loss"]
        self.metrics["best_miou"] = ck["best_miou"]

    def write_info_json(self):
        if self.rank != 0:
            return
        cfg = json.loads(json.dumps(self.config, default=str))
        with open(self.log_dir / "info.json", "w") as f:
            json.dump({"config": cfg, "metrics": self.metrics, "history": self.history}, f, indent=1, default=str)
