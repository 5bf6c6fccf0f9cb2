"""Rank-sharded, pinned-memory uint8 frame loader feeding ``GpuIngest``.

The reference's loaders (managers/BaseManager.py:286-405) decode and augment on the main thread (the repeat-factor loader is
hard-wired to ``num_workers=0``, :391) and hand float32 tensors to ``img.to(device)`` from pageable memory (OCRNet_Manager.py:75):
62.5 MB of float32 + int64 per 8-frame batch.  Here the host side only stacks RAW uint8 frames (12.4 + 4.1 MB per batch) into
pinned staging buffers on background threads; the copy to the device runs on a side HIP stream while the previous step
computes, and remap / flip / reflect-pad / ToTensor / Normalize happen in one kernel on the device (``GpuIngest``).

    loader = PinnedFrameLoader(dataset, batch_size=8, experiment=3, sampler=RepeatFactorSampler(...))   # or indices / None
    for x, labels in loader:        # x float32 [B,3,H+4,W] (or NHWC-4), labels int64 [B,H+4,W], both on the device

``dataset[i]`` returns ``(img uint8 [H,W,3] RGB, lbl uint8 [H,W] raw CaDIS ids)`` (numpy arrays or tensors; anything after the
second element is ignored) -- e.g. ``cv2.imread`` + BGR->RGB as datasets/Dataset_from_df.py:36-44 does.  With WORLD_SIZE > 1
the index stream is sharded by rank (dist.ShardedSampler).  blur= / colorjitter= apply BlurPIL / ColorJitter on the GPU (utils/augment.py).
"""
import queue
import threading

import numpy as np
import torch

from .augment import sample_blur, sample_color_jitter
from .ingest import GpuIngest, sample_flips


class PinnedFrameLoader:
    def __init__(self, dataset, batch_size, experiment, sampler=None, shuffle=True, drop_last=True, flip_probability=(0.0, 0.5),
                 pad=(2, 2), normalise=False, nhwc4=False, device="cuda", prefetch=3, workers=0, seed=0, rank=0, world=1,
                 blur=False, colorjitter=False):
        """blur / colorjitter: the 'blur' / 'colorjitter' entries of the reference's `transforms` config list (utils/utils.py:412-417):
        BlurPIL(probability=.05, kernel_limits=(3, 7)) and ColorJitter((2/3, 1.5) x 3, hue (-.05, .05)) on the padded uint8 frames,
        as GPU kernels (utils/augment.py)"""
        self.dataset, self.batch, self.drop_last = dataset, int(batch_size), drop_last
        self.device = torch.device(device)
        self.ingest = GpuIngest(experiment, pad=pad, normalise=normalise, device=device)
        self.flip_p, self.nhwc4 = flip_probability, nhwc4
        self.blur, self.colorjitter = bool(blur), bool(colorjitter)
        # workers = 0: no fill thread -- each batch is stacked into its staging slot by the CONSUMER's thread inside next(), i.e. after the
        # previous step's kernels have been enqueued and while the GPU runs them (a fill thread shares the interpreter lock with the
        # launch loop of ~3000 kernels per step: on a busy host that stretched an HRNet-W48 step by 10 - 30 ms; the inline fill costs the
        # host ~9 ms per batch of eight 540 x 960 frames that it has to spare).  workers >= 1: a producer thread (datasets whose
        # __getitem__ releases the interpreter lock for long: decoding).
        self.prefetch, self.workers = max(int(prefetch), 1), max(int(workers), 0)
        self.seed, self.epoch, self.rank, self.world = seed, 0, rank, world
        self.sampler, self.shuffle = sampler, shuffle
        if world > 1 and sampler is not None:
            from ..dist import check_unsharded
            check_unsharded(sampler)        # (this loader takes the rank's slice itself)
        self.copy_stream = torch.cuda.Stream(device=self.device)
        self._slots = None

    def set_epoch(self, epoch):
        self.epoch = int(epoch)

    def _indices(self):
        if self.sampler is not None:
            idx = list(iter(self.sampler))
        else:
            n = len(self.dataset)
            idx = torch.randperm(n, generator=torch.Generator().manual_seed(self.seed + self.epoch)).tolist() if self.shuffle else list(range(n))
        if self.world > 1:
            n = len(idx) // (self.world * self.batch) * (self.world * self.batch)
            idx = idx[self.rank:n:self.world]
        return idx

    def __len__(self):
        n = len(self._indices()) if self.sampler is None else len(self.sampler) // max(self.world, 1)
        return n // self.batch if self.drop_last else (n + self.batch - 1) // self.batch

    def _alloc(self, h, w):
        self._slots = [(torch.empty((self.batch, h, w, 3), dtype=torch.uint8).pin_memory(),
                        torch.empty((self.batch, h, w), dtype=torch.uint8).pin_memory()) for _ in range(self.prefetch + 1)]

    def __iter__(self):
        idx = self._indices()
        self.epoch += 1
        batches = [idx[i:i + self.batch] for i in range(0, len(idx), self.batch)]
        if self.drop_last and batches and len(batches[-1]) < self.batch:
            batches.pop()
        rng = np.random.RandomState(self.seed * 1000003 + self.epoch)
        flips = [sample_flips(len(b), self.flip_p, rng) for b in batches]
        gen = torch.Generator().manual_seed(self.seed * 1000003 + self.epoch)
        blurs = [sample_blur(len(b), random=rng) if self.blur else None for b in batches]
        jitters = [sample_color_jitter(len(b), generator=gen) if self.colorjitter else None for b in batches]
        ready = queue.Queue(maxsize=self.prefetch)
        free = queue.Queue()
        first = self.dataset[batches[0][0]] if batches else None
        if first is not None and (self._slots is None or tuple(self._slots[0][0].shape[1:3]) != tuple(np.asarray(first[0]).shape[:2])):
            self._alloc(*np.asarray(first[0]).shape[:2])
        for s in range(len(self._slots or [])):
            free.put(s)
        stop = threading.Event()

        def fill(slot, ids):
            img_buf, lbl_buf = self._slots[slot]
            def one(j):
                item = self.dataset[ids[j]]
                img_buf[j].copy_(torch.as_tensor(np.ascontiguousarray(item[0])))
                lbl_buf[j].copy_(torch.as_tensor(np.ascontiguousarray(item[1])))
            if self.workers > 1 and len(ids) > 1:
                ts = [threading.Thread(target=lambda a=a: [one(j) for j in range(a, len(ids), self.workers)]) for a in range(min(self.workers, len(ids)))]
                for t in ts:
                    t.start()
                for t in ts:
                    t.join()
            else:
                for j in range(len(ids)):
                    one(j)

        def put(q, item):                 # bounded queue: never block for good once the consumer has gone away
            while not stop.is_set():
                try:
                    q.put(item, timeout=0.1)
                    return True
                except queue.Full:
                    pass
            return False

        def producer():
            try:
                for bi, ids in enumerate(batches):
                    slot = None
                    while slot is None and not stop.is_set():
                        try:
                            slot = free.get(timeout=0.1)
                        except queue.Empty:
                            pass
                    if stop.is_set():
                        return
                    fill(slot, ids)
                    if not put(ready, (slot, len(ids), flips[bi], blurs[bi], jitters[bi])):
                        return
            finally:
                put(ready, None)

        # the consumer's thread is the one that launches the training step's ~3000 kernels: a fill thread holding the interpreter lock for a
        # whole default switch interval (5 ms) at a time starves the GPU; hand the lock over at 0.5 ms
        import sys
        if self.workers > 0 and sys.getswitchinterval() > 5e-4:
            sys.setswitchinterval(5e-4)
        inline = self.workers == 0
        th = threading.Thread(target=producer, daemon=True)
        if not inline:
            th.start()
        nxt = 0            # (inline fill: index of the next batch)
        in_copy = []       # (event, slot): staging slots whose host -> device copy may still be reading them
        try:
            while True:
                # a staging slot goes back to the producer once its copy has left it.  The host never WAITS for that (unless every slot
                # is still in a copy): HIP maps streams onto a few hardware queues, the copy may sit behind a whole step of kernels the
                # host has already enqueued, and a blocking wait here stalled the launch loop of the next step by 15 ... 45 ms
                while in_copy and (in_copy[0][0].query() or len(in_copy) >= len(self._slots)):
                    ev, sl = in_copy.pop(0)
                    ev.synchronize()
                    free.put(sl)
                if inline:
                    if nxt >= len(batches):
                        break
                    if free.empty():           # every slot still in a copy: wait for the oldest
                        ev, sl = in_copy.pop(0)
                        ev.synchronize()
                        free.put(sl)
                    sl = free.get()
                    fill(sl, batches[nxt])
                    item = (sl, len(batches[nxt]), flips[nxt], blurs[nxt], jitters[nxt])
                    nxt += 1
                else:
                    item = ready.get()
                if item is None:
                    break
                slot, n, fl, br, jt = item
                img_buf, lbl_buf = self._slots[slot]
                with torch.cuda.stream(self.copy_stream):          # host -> device on the side stream (pinned: truly asynchronous)
                    img_d = img_buf[:n].to(self.device, non_blocking=True)
                    lbl_d = lbl_buf[:n].to(self.device, non_blocking=True)
                    done = torch.cuda.Event()
                    done.record(self.copy_stream)
                torch.cuda.current_stream(self.device).wait_event(done)   # the ingest kernel waits for the copy, the host does not
                img_d.record_stream(torch.cuda.current_stream(self.device))
                lbl_d.record_stream(torch.cuda.current_stream(self.device))
                x, labels = self.ingest(img_d, lbl_d, fl, nhwc4=self.nhwc4, blur_radii=br, jitter=jt)
                in_copy.append((done, slot))
                yield x, labels
        finally:
            # the consumer is done or has abandoned the iterator (break / exception): stop the producer and wait for it, so that it
            # cannot write into the pinned staging slots while the NEXT iteration fills them
            stop.set()
            while True:
                try:
                    ready.get_nowait()
                except queue.Empty:
                    break
            if not inline:
                th.join()
            for ev, _ in in_copy:          # (the next iteration refills the slots: their copies must have left them)
                ev.synchronize()
