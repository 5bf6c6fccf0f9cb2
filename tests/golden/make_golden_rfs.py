"""Repeat-factor-sampling fixture from the REAL reference + its data/data.csv (utils/repeat_factor_sampling.py).
    PYTHONDONTWRITEBYTECODE=1 python tests/golden/make_golden_rfs.py"""
import os
import sys

import numpy as np
import pandas as pd
import torch

HERE = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, HERE)
sys.path.insert(0, os.path.dirname(os.path.dirname(HERE)))
import ref_harness  # noqa: E402

R = ref_harness.load()
U = R.utils
df = pd.read_csv(os.path.join(ref_harness.REF, "data", "data.csv"))
out = {}
canonical = U.CLASS_NAMES[0]
name_to_num = U.reverse_one_to_many_mapping(U.CLASS_INFO[0][1])
for exp in (1, 2, 3):
    s = U.RepeatFactorSampler(data_source=None, dataframe=df, repeat_thresh=0.15, experiment=exp, split=2, blacklist=True)
    # inputs the product needs: per-frame canonical class presence of the training split + canonical->experiment map
    d = U.get_class_info(df, 0, with_name=True)
    d = d.drop(d[d["blacklisted"] == 1].index)
    train = d.loc[d["vid_num"].isin(U.DATA_SPLITS[2][0])].reset_index()
    pres = np.stack([(train[c] > 0).to_numpy() for c in canonical], 1)
    rev = U.reverse_one_to_many_mapping(U.CLASS_INFO[exp][0])
    cmap = np.array([rev[name_to_num[c]] for c in canonical], dtype=np.int64)
    K = len(U.CLASS_INFO[exp][1])
    crf = np.zeros(max(U.CLASS_INFO[exp][1].keys()) + 1)
    keys = sorted(U.CLASS_INFO[exp][1].keys())
    for k, v in s.class_repeat_factors.items():
        crf[k] = v
    n1 = len(s)
    e1 = [i for i in s]
    n2 = len(s)
    e2 = [i for i in s]
    out.update({"e%d_presence" % exp: pres, "e%d_cmap" % exp: cmap, "e%d_class_keys" % exp: np.array(keys),
                "e%d_class_rf" % exp: crf, "e%d_image_rf" % exp: s.repeat_factors.numpy(),
                "e%d_epoch1" % exp: np.array(e1), "e%d_epoch2" % exp: np.array(e2), "e%d_lens" % exp: np.array([n1, n2])})
    print(exp, pres.shape, n1, n2, float(s.repeat_factors.sum()), float(s.repeat_factors.max()))
np.savez_compressed(os.path.join(HERE, "rfs.npz"), **out)
print("wrote rfs.npz", os.path.getsize(os.path.join(HERE, "rfs.npz")))
