"""Deterministic parameter fill shared by the golden generator, the oracle and the tests
(oracle; test infrastructure only).  Given the ordered (key, shape) list of a model's
state dict it produces the same tensors on every machine with this torch build, so
fixtures only need to store the seed, not 150 MB of weights."""
from collections import OrderedDict

import torch


def fill_state(spec, seed):
    """spec: iterable of (key, shape). Non-trivial BN affine/running stats on purpose."""
    g = torch.Generator().manual_seed(seed)
    S = OrderedDict()
    for key, shape in spec:
        shape = tuple(shape)
        leaf = key.rsplit(".", 1)[-1]
        if leaf == "num_batches_tracked":
            t = torch.zeros((), dtype=torch.int64)
        elif leaf == "running_mean":
            t = 0.1 * torch.randn(shape, generator=g)
        elif leaf == "running_var":
            t = 1.0 + 0.2 * torch.rand(shape, generator=g)
        elif len(shape) == 1 and leaf == "weight":        # BN gamma
            t = 1.0 + 0.1 * torch.randn(shape, generator=g)
        elif len(shape) == 1:                              # BN beta / conv bias
            t = 0.1 * torch.randn(shape, generator=g)
        else:                                              # conv weight (O, I, kh, kw)
            fan_in = shape[1] * shape[2] * shape[3]
            t = torch.randn(shape, generator=g) * (2.0 / fan_in) ** 0.5
        S[key] = t
    return S


def spec_of(state_dict):
    return [(k, tuple(v.shape)) for k, v in state_dict.items()]
