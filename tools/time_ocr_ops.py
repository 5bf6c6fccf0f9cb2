"""Times the pieces of the OCR head's spatial gather / object attention at the bench shape (8 x 136 x 240, 512 / 256 channels, 25 classes).
python3 tools/time_ocr_ops.py"""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from miccai2021_cataract_semantic_segmentation_amd import ops
dev = torch.device("cuda")
B, H, W, C, K, Ck = 8, 136, 240, 512, 25, 256
N = H * W
feats = torch.randn(B, H, W, C, device=dev)
logits = torch.zeros(B, H, W, 32, device=dev); logits[..., :K] = torch.randn(B, H, W, K, device=dev)
lbuf = logits.view(B, N, 32)
q = torch.randn(B, H, W, Ck, device=dev)
key = torch.randn(B, K, 1, Ck, device=dev); val = torch.randn(B, K, 1, Ck, device=dev)
dproxy = torch.randn(B, K, 1, C, device=dev)


def t(name, fn, n=5):
    for _ in range(2): fn()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(n): fn()
    e1.record(); torch.cuda.synchronize()
    print("%-64s %8.3f ms" % (name, e0.elapsed_time(e1) / n), flush=True)


probs = ops.softmax_spatial_fwd(lbuf, K)
proxy = torch.empty(B, K, 1, C, device=dev)
t("softmax_spatial_fwd", lambda: ops.softmax_spatial_fwd(lbuf, K))
t("gather: gemm TN split  [25 x 512] <- 32640 rows", lambda: ops.gemm_tn_split(B, K, C, N, probs, 32, feats, C, proxy))
ops.GEMM_TN_SPLIT = False
t("gather: gemm TN one launch", lambda: ops.gemm_tn_split(B, K, C, N, probs, 32, feats, C, proxy))
ops.GEMM_TN_SPLIT = True
dfe = torch.empty_like(feats)
t("gather bwd: gemm NN dfeats [32640 x 512] = probs[32640 x 25] dproxy", lambda: ops.gemm(ops.NN, B, N, C, K, probs, 32, N * 32, dproxy, C, K * C, dfe, C, N * C))
t("gather bwd: same, accumulate", lambda: ops.gemm(ops.NN, B, N, C, K, probs, 32, N * 32, dproxy, C, K * C, dfe, C, N * C, accumulate=True))
dprobs = torch.empty_like(probs)
t("gather bwd: gemm NT dprobs [32640 x 25] = feats dproxy^T", lambda: ops.gemm(ops.NT, B, N, K, C, feats, C, N * C, dproxy, C, K * C, dprobs, 32, N * 32, zero_to=32))
dl = torch.empty_like(lbuf)
t("gather bwd: softmax_spatial_bwd", lambda: ops.softmax_spatial_bwd(probs, dprobs, dl, K))
sim = torch.empty(B, N, 32, device=dev)
t("attention: gemm NT sim [32640 x 25] = q key^T", lambda: ops.gemm(ops.NT, B, N, K, Ck, q, Ck, N * Ck, key, Ck, K * Ck, sim, 32, N * 32, zero_to=32))
p = ops.softmax_rows_fwd(sim.view(B * N, 32), K, 0.0625)
t("attention: softmax_rows_fwd", lambda: ops.softmax_rows_fwd(sim.view(B * N, 32), K, 0.0625))
ctx = torch.empty(B, H, W, Ck, device=dev)
t("attention: gemm NN ctx [32640 x 256] = p val", lambda: ops.gemm(ops.NN, B, N, Ck, K, p, 32, N * 32, val, Ck, K * Ck, ctx, Ck, N * Ck))
dv = torch.empty(B, K, 1, Ck, device=dev)
t("attention bwd: gemm TN split dv [25 x 256] <- 32640 rows", lambda: ops.gemm_tn_split(B, K, Ck, N, p, 32, ctx, Ck, dv))
