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
import multiprocessing
import multiprocessing.connection
import os
import queue
import threading

import numpy as np
import torch

from .augment import sample_blur, sample_color_jitter
from .ingest import GpuIngest, sample_flips


def _worker_main(conn, dataset, slots, widx, nworkers):
    """a loader worker PROCESS (forked, never exec'ed; it never touches the GPU): for every task (slot, ids, token) it reads frames
    widx, widx + nworkers, ... of the batch from the dataset and writes them into the SHARED staging slot, then answers with the token"""
    code = 0
    token = None
    try:
        torch.set_num_threads(1)
        while True:
            task = conn.recv()
            if task is None:
                break
            slot, ids, token = task
            img_buf, lbl_buf = slots[slot]
            for j in range(widx, len(ids), nworkers):
                item = dataset[ids[j]]
                img_buf[j].copy_(torch.as_tensor(np.ascontiguousarray(item[0])))
                lbl_buf[j].copy_(torch.as_tensor(np.ascontiguousarray(item[1])))
            conn.send(token)
    except (EOFError, KeyboardInterrupt, BrokenPipeError):
        pass
    except BaseException:       # noqa: BLE001  a dataset error, a corrupt frame, a shape mismatch: the parent re-raises it with this traceback
        code = 1                # (torch's DataLoader does the same for its workers; a silent exit 0 left the parent with a bare EOFError)
        try:
            import traceback
            conn.send(("error", token, widx, traceback.format_exc()))
        except (BrokenPipeError, OSError):
            pass
    finally:
        os._exit(code)         # (no interpreter shutdown in a forked copy of a process that has initialised the GPU)


class _WorkerPool:
    """worker processes filling shared, pinned staging slots (the reference's DataLoader(num_workers=...),
    managers/BaseManager.py:298-305).  The processes are FORKED from the loader's process (as torch's DataLoader forks its workers): no
    exec of a GPU-initialised process, the dataset needs no pickling; they only read the dataset and write host memory.
    The fork is PERSISTENT (torch's `persistent_workers=True`): the workers hold the copy of the dataset object they inherited when the
    pool was created, so a later change of the parent's dataset state does not reach them -- PinnedFrameLoader.close() (or a frame-shape
    change) ends the pool and the next iteration forks a new one."""

    def __init__(self, dataset, slots, n):
        ctx = multiprocessing.get_context("fork")
        self.conns, self.procs = [], []
        for w in range(n):
            a, b = ctx.Pipe(duplex=True)
            p = ctx.Process(target=_worker_main, args=(b, dataset, slots, w, n), daemon=True)
            p.start()
            b.close()
            self.conns.append(a)
            self.procs.append(p)
        self.pending = {}        # token -> answers still missing

    def submit(self, slot, ids, token):
        self.pending[token] = len(self.conns)
        for c in self.conns:
            c.send((slot, list(ids), token))

    def wait(self, token):
        """blocks (interpreter lock released while polling the pipes) until every worker has answered `token`"""
        while self.pending.get(token, 0) > 0:
            if not self.conns:
                raise RuntimeError("loader worker pool was closed while batch %r was pending" % (token,))
            for w, c in enumerate(self.conns):
                try:
                    while c.poll(0):
                        msg = c.recv()
                        if isinstance(msg, tuple) and msg and msg[0] == "error":
                            raise RuntimeError("loader worker %d failed on batch token %r:\n%s" % (msg[2], msg[1], msg[3]))
                        self.pending[msg] -= 1
                except (EOFError, OSError) as e:
                    p = self.procs[w]
                    p.join(timeout=1.0)
                    raise RuntimeError("loader worker %d (pid %s) died without a report, exit code %s"
                                       % (w, p.pid, p.exitcode)) from e
            if self.pending[token] > 0:
                multiprocessing.connection.wait(self.conns, timeout=0.05)
        self.pending.pop(token, None)

    def close(self):
        for c in self.conns:
            try:
                c.send(None)
            except (BrokenPipeError, OSError):
                pass
        for p in self.procs:
            p.join(timeout=2.0)
            if p.is_alive():
                p.terminate()       # (this exact child)
                p.join(timeout=1.0)
        for c in self.conns:
            c.close()
        self.conns, self.procs = [], []
        self.pending = {}


class PinnedFrameLoader:
    def __init__(self, dataset, batch_size, experiment, sampler=None, shuffle=True, drop_last=True, flip_probability=(0.0, 0.5),
                 pad=(2, 2), normalise=False, nhwc4=False, device="cuda", prefetch=3, workers=0, seed=0, rank=0, world=1,
                 blur=False, colorjitter=False, worker_processes=0):
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
        # worker_processes = N: N forked processes read the dataset and stack the frames into SHARED pinned staging slots; the consumer's
        # thread only enqueues the host -> device copy and the ingest kernels (immune to the interpreter lock and to a slow __getitem__)
        self.nproc = max(int(worker_processes), 0)
        self._pool = None
        self.copy_stream = torch.cuda.Stream(device=self.device)
        self._slots = None

    def close(self):
        if self._pool is not None:
            self._pool.close()
            self._pool = None
        self._unregister()

    def _unregister(self):
        """the shared staging slots were page-locked with hipHostRegister: un-register them (after the copy stream has drained) before the
        tensors are dropped -- a registration that outlives its pages can make a later hipHostRegister of a reused address range fail"""
        regs, self._registered = getattr(self, "_registered", []), []
        if regs:
            self.copy_stream.synchronize()
            rt = torch.cuda.cudart()
            for t in regs:
                rt.cudaHostUnregister(t.data_ptr())

    def __del__(self):
        try:
            self.close()
        except Exception:       # noqa: BLE001
            pass

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
        if self.nproc:
            # shared memory first (the workers inherit the mapping at the fork), then page-locked in THIS process for the asynchronous copies
            self.close()
            self._slots = [(torch.empty((self.batch, h, w, 3), dtype=torch.uint8).share_memory_(),
                            torch.empty((self.batch, h, w), dtype=torch.uint8).share_memory_()) for _ in range(self.prefetch + 1)]
            self._pool = _WorkerPool(self.dataset, self._slots, self.nproc)
            rt = torch.cuda.cudart()
            self._registered = []
            for pair in self._slots:
                for t in pair:
                    err = rt.cudaHostRegister(t.data_ptr(), t.numel(), 0)
                    if int(err) != 0:
                        raise RuntimeError("hipHostRegister of a shared staging slot failed: %s" % (err,))
                    self._registered.append(t)      # (keeps the pages alive until _unregister)
            return
        self._slots = [(torch.empty((self.batch, h, w, 3), dtype=torch.uint8).pin_memory(),
                        torch.empty((self.batch, h, w), dtype=torch.uint8).pin_memory()) for _ in range(self.prefetch + 1)]

    def _emit(self, slot, n, fl, br, jt):
        """host -> device copy of a filled staging slot on the side stream + the ingest kernels; returns (x, labels, copy event)"""
        img_buf, lbl_buf = self._slots[slot]
        with torch.cuda.stream(self.copy_stream):          # pinned: truly asynchronous
            img_d = img_buf[:n].to(self.device, non_blocking=True)
            lbl_d = lbl_buf[:n].to(self.device, non_blocking=True)
            done = torch.cuda.Event()
            done.record(self.copy_stream)
        cur = torch.cuda.current_stream(self.device)
        cur.wait_event(done)                               # the ingest kernel waits for the copy, the host does not
        img_d.record_stream(cur)
        lbl_d.record_stream(cur)
        x, labels = self.ingest(img_d, lbl_d, fl, nhwc4=self.nhwc4, blur_radii=br, jitter=jt)
        return x, labels, done

    def _iter_processes(self, batches, flips, blurs, jitters):
        """the worker processes run `prefetch` batches ahead; a staging slot returns to them when its copy has left it"""
        pool = self._pool
        free = list(range(len(self._slots)))
        in_copy, submitted, slot_of = [], 0, {}
        self._tok = getattr(self, "_tok", 0)
        base = self._tok
        self._tok += len(batches)
        try:
            bi = 0
            while bi < len(batches):
                while in_copy and in_copy[0][0].query():   # copies that have left their slot (never a blocking wait here)
                    free.append(in_copy.pop(0)[1])
                while submitted < min(len(batches), bi + self.prefetch) and free:
                    sl = free.pop(0)
                    slot_of[submitted] = sl
                    pool.submit(sl, batches[submitted], base + submitted)
                    submitted += 1
                if submitted <= bi:                        # every slot is still in a copy: wait for the oldest
                    ev, sl = in_copy.pop(0)
                    ev.synchronize()
                    free.append(sl)
                    continue
                pool.wait(base + bi)
                sl = slot_of.pop(bi)
                x, labels, done = self._emit(sl, len(batches[bi]), flips[bi], blurs[bi], jitters[bi])
                in_copy.append((done, sl))
                bi += 1
                yield x, labels
        finally:
            if pool.conns:                                 # (close() / __del__ may already have ended the pool)
                for t in range(base, base + submitted):   # (abandoned iteration: the workers finish what they were given)
                    if t in pool.pending:
                        try:
                            pool.wait(t)
                        except RuntimeError:
                            break                          # (a dead worker: its error has been / is being raised by the consumer's wait)
            for ev, _ in in_copy:
                ev.synchronize()

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
        if first is not None and (self._slots is None or tuple(self._slots[0][0].shape[1:3]) != tuple(np.asarray(first[0]).shape[:2])
                                  or (self.nproc and self._pool is None)):
            self._alloc(*np.asarray(first[0]).shape[:2])
        if self.nproc and batches:
            yield from self._iter_processes(batches, flips, blurs, jitters)
            return
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
