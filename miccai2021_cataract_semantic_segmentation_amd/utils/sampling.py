"""Repeat-factor sampling (LVIS style) as the reference implements it in
utils/repeat_factor_sampling.py:9-131 — CPU-side, no kernel: it only decides which frames an epoch
visits.  Inputs are explicit instead of a pandas frame: ``presence`` [frames, n_canonical] (bool: class
present in frame, training split, blacklist already applied) and ``cmap`` [n_canonical] -> experiment class.

Kept quirks: a frame that contains several canonical classes of one experiment class counts once per
canonical class in that class's frequency (:26-27); ``len()`` draws the epoch (consumes the generator) and
``iter()`` re-uses that draw (:103-121); the generator is private and seeded with 1 by default (:75-76)."""
import numpy as np
import torch
from torch.utils.data import Sampler


def class_repeat_factors(presence, cmap, class_keys, repeat_thresh):
    """r(c) = max(1, sqrt(t / f(c))), f(c) = sum over canonical classes mapped to c of (frames containing it)/frames"""
    presence = np.asarray(presence, dtype=bool)
    n = presence.shape[0]
    freq = {int(k): 0.0 for k in class_keys}
    for j, c in enumerate(np.asarray(cmap)):
        freq[int(c)] = freq.get(int(c), 0.0) + presence[:, j].sum() / n
    rf = {}
    for c in class_keys:
        f = freq[int(c)] if freq[int(c)] != 0 else repeat_thresh
        rf[int(c)] = float(np.maximum(1, np.sqrt(repeat_thresh / f)))
    return rf


def image_repeat_factors(presence, cmap, cls_rf):
    """r(I) = max_{c in I} r(c)"""
    presence = np.asarray(presence, dtype=bool)
    per_canon = np.array([cls_rf[int(c)] for c in np.asarray(cmap)])
    return torch.tensor(np.where(presence, per_canon[None, :], -np.inf).max(1), dtype=torch.float32)


class RepeatFactorSampler(Sampler):
    def __init__(self, presence, cmap, class_keys, repeat_thresh, seed=None, rank=0, world=1):
        assert 0 <= repeat_thresh < 1
        self.seed = int(1 if seed is None else seed)
        self.shuffle = True
        self.repeat_thresh = repeat_thresh
        self.class_repeat_factors = class_repeat_factors(presence, cmap, class_keys, repeat_thresh)
        self.repeat_factors = image_repeat_factors(presence, cmap, self.class_repeat_factors)
        self._int_part = torch.trunc(self.repeat_factors)
        self._frac_part = self.repeat_factors - self._int_part
        self.g = torch.Generator()
        self.g.manual_seed(self.seed)
        self.indices = None
        self.rank, self.world = rank, world   # data parallel: every rank draws the same epoch, keeps a strided shard

    def _get_epoch_indices(self, generator):
        rands = torch.rand(len(self._frac_part), generator=generator)
        rounded = self._int_part + (rands < self._frac_part).float()
        idx = torch.repeat_interleave(torch.arange(len(rounded)), rounded.long())
        self.indices = idx.tolist()
        return idx

    def __len__(self):
        n = len(self.indices) if self.indices is not None else len(self._get_epoch_indices(self.g))
        return n // self.world if self.world > 1 else n

    def __iter__(self):
        indices = torch.tensor(self.indices, dtype=torch.int64) if self.indices is not None else self._get_epoch_indices(self.g)
        randperm = torch.randperm(len(indices), generator=self.g)
        order = indices[randperm].tolist()
        if self.world > 1:
            n = len(order) // self.world * self.world
            order = order[self.rank:n:self.world]
        self.indices = None
        return iter(order)
