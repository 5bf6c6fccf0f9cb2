"""what a batch costs in PinnedFrameLoader alone (no training step beside it): host time per next(), GPU time of the ingest"""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from miccai2021_cataract_semantic_segmentation_amd.utils.loader import PinnedFrameLoader
dev = torch.device("cuda")
class Frames(torch.utils.data.Dataset):
    def __init__(self):
        g = torch.Generator().manual_seed(7)
        self.img = torch.randint(0, 256, (64, 540, 960, 3), dtype=torch.uint8, generator=g).numpy()
        self.lbl = torch.randint(0, 36, (64, 18, 32), dtype=torch.uint8, generator=g).repeat_interleave(30, 1).repeat_interleave(30, 2).numpy()
    def __len__(self): return 64
    def __getitem__(self, i): return self.img[i], self.lbl[i]
for workers in (4, 1):
    loader = PinnedFrameLoader(Frames(), batch_size=8, experiment=3, flip_probability=(0.0, 0.5), pad=(2, 2), normalise=False, device=dev,
                               blur=True, colorjitter=True, seed=0, workers=workers)
    def forever():
        while True:
            for b in loader:
                yield b
    gen = forever()
    for _ in range(4): next(gen)
    torch.cuda.synchronize()
    ts = []
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for i in range(24):
        t0 = time.perf_counter(); next(gen); ts.append((time.perf_counter() - t0) * 1e3)
    e1.record(); torch.cuda.synchronize()
    print("workers=%d: host ms per next(): mean %.2f, max %.2f (every 8th = epoch restart: %s); GPU time per batch %.2f ms"
          % (workers, sum(ts) / len(ts), max(ts), ["%.1f" % t for t in ts[::8]], e0.elapsed_time(e1) / 24), flush=True)
# GPU time of the ingest alone (resident uint8 batch)
from miccai2021_cataract_semantic_segmentation_amd.utils.loader import sample_flips
import numpy as np
from miccai2021_cataract_semantic_segmentation_amd.utils.augment import sample_blur, sample_color_jitter
fr = Frames()
img = torch.from_numpy(fr.img[:8]).to(dev); lbl = torch.from_numpy(fr.lbl[:8]).to(dev)
rng = np.random.RandomState(1)
fl = sample_flips(8, (0.0, 0.5), rng); br = sample_blur(8, random=rng); jt = sample_color_jitter(8, generator=torch.Generator().manual_seed(1))
for name, kw in (("remap + flip + pad + ToTensor", {}), ("+ blur", {"blur_radii": br}), ("+ blur + colour jitter", {"blur_radii": br, "jitter": jt})):
    for _ in range(3): loader.ingest(img, lbl, fl, nhwc4=False, **kw)
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(10): loader.ingest(img, lbl, fl, nhwc4=False, **kw)
    e1.record(); torch.cuda.synchronize()
    print("ingest %-32s %.2f ms GPU per batch (blur radii %s)" % (name, e0.elapsed_time(e1) / 10, br), flush=True)
