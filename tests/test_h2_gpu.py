"""f16x2 split passes of the head convolutions: max|x| from the producers' amax records instead of a pass over the tensor
(csrc/igemm_f16x2.hip: catseg_split2h_bound)."""
import pytest
import torch

pytestmark = pytest.mark.gpu


def _need_gpu():
    if not torch.cuda.is_available():
        pytest.skip("needs a GPU")


def test_split2h_bound_takes_max_from_producer_records():
    """catseg_split2h_bound: the exponent comes from up to four amax records instead of a pass over x; with the exact maximum in the records the
    planes are bit-identical to catseg_split2h's, with an upper bound one power of two looser they still reproduce x to 2^-21 of the bound"""
    _need_gpu()
    from miccai2021_cataract_semantic_segmentation_amd import ops
    dev = torch.device("cuda")
    g = torch.Generator().manual_seed(5)
    x = (torch.randn(3, 9, 17, 80, generator=g) * torch.exp(torch.randn(80, generator=g))).to(dev)
    blk0, pl0, sc0 = ops.split2h(x, True, True)
    parts = [x[..., :16], x[..., 16:48], x[..., 48:]]
    recs = []
    for p in parts:
        r = ops.new_amax(dev)
        r[32 * (len(recs) + 3)] = p.abs().max().reshape(1).view(torch.int32)[0]      # any slot of the record may hold the maximum
        recs.append(r)
    x._amax_parts = recs
    blk1, pl1, sc1 = ops.split2h(x, True, True)
    assert int(sc1[1]) == int(sc0[1]) and torch.equal(blk0, blk1) and torch.equal(pl0, pl1)
    x._amax_parts = None
    r = ops.new_amax(dev)
    r[0] = (x.abs().max() * 2.5).reshape(1).view(torch.int32)[0]
    x._amax = r
    blk2, pl2, sc2 = ops.split2h(x, True, True)
    e = int(sc2[1])
    assert e in (int(sc0[1]) - 1, int(sc0[1]) - 2)
    hl = pl2.view(torch.float16).float()                       # [2, rows, 80]
    back = (hl[0] + hl[1]).reshape(3, 9, 17, 80) * 2.0 ** (-e)
    assert float((back - x).abs().max()) <= float(x.abs().max()) * 2.5 * 2.0 ** -21
    ops.SPLIT_BOUND = False
    try:
        blk3, _, sc3 = ops.split2h(x, True, False)
    finally:
        ops.SPLIT_BOUND = True
    assert torch.equal(blk3, blk0) and int(sc3[1]) == int(sc0[1])


def test_concat_records_follow_the_engine_and_die_with_in_place_updates():
    """engine.concat_views collects the slices' amax records (copies and bilinear resizes keep their source's); an accumulating launch into the
    buffer drops them (a stale bound would overflow the fp16 planes), and the split pass then takes its own maximum again"""
    _need_gpu()
    from miccai2021_cataract_semantic_segmentation_amd import engine, ops
    dev = torch.device("cuda")
    saved = ops.TRUNK
    ops.TRUNK = "f16x2"
    try:
        scope = ops.AmaxScope(dev, 64)
        ops.set_amax_scope(scope)
        cx = engine.Ctx(train=True, record=False)
        a = torch.randn(2, 8, 12, 16, device=dev) * 3.0
        b = torch.randn(2, 4, 6, 32, device=dev) * 5.0
        for t in (a, b):
            t._amax = ops.new_amax(dev)
            t._amax[0:1] = t.abs().max().reshape(1).view(torch.int32)
        cat = torch.empty(2, 8, 12, 48, device=dev)
        p0 = engine.copy_into(cx, a, cat[..., :16])
        p1 = engine.bilinear(cx, b, 8, 12, False, out=cat[..., 16:])
        engine.concat_views(cx, cat, [(p0, 0, 16), (p1, 16, 48)])
        recs = ops.amax_records_of(cat)
        assert recs is not None and len(recs) == 2
        blk, _, sc = ops.split2h(cat, True, False)
        bound = max(float(a.abs().max()), float(b.abs().max()))
        assert float(sc.view(torch.float32)[0]) == bound                      # the merged records, not a pass over the buffer
        assert float(cat.abs().max()) <= bound
        ops.axpy(torch.full_like(cat, 100.0), cat, 1.0, True)                 # in-place update: the records no longer bound the contents
        assert ops.amax_records_of(cat) is None
        _, _, sc2 = ops.split2h(cat, True, False)
        assert float(sc2.view(torch.float32)[0]) == float(cat.abs().max())
    finally:
        ops.TRUNK = saved
        ops.set_amax_scope(None)


@pytest.mark.parametrize("shape", [(2, 25, 512, 32640), (1, 17, 256, 8160), (3, 25, 256, 4100), (2, 8, 64, 1000)])
def test_gemm_tn_split_matches_fp64(shape):
    """the K-split form of the OCR head's long reductions (ops.gemm_tn_split: batch * splits chunk GEMMs + catseg_sum_slabs) against fp64, with
    and without accumulation, deterministic; shapes that do not split (indivisible or short K) take the one-launch path"""
    _need_gpu()
    from miccai2021_cataract_semantic_segmentation_amd import ops
    B, M, N, K = shape
    dev = torch.device("cuda")
    g = torch.Generator().manual_seed(K + M)
    lda = (M + 7) // 8 * 8
    A = torch.zeros(B, K, lda)
    A[..., :M] = torch.rand(B, K, M, generator=g)              # (softmax-like non-negative weights: no cancellation to hide a wrong chunking)
    Bm = torch.randn(B, K, N, generator=g)
    ref = torch.einsum("bkm,bkn->bmn", A[..., :M].double(), Bm.double())
    Ad, Bd = A.to(dev), Bm.to(dev)
    C0 = torch.full((B, M, N), float("nan"), device=dev)
    ops.gemm_tn_split(B, M, N, K, Ad, lda, Bd, N, C0)
    err = float((C0.cpu().double() - ref).abs().max()) / float(ref.abs().max())
    assert err <= 2e-6, err
    C1 = torch.full((B, M, N), float("nan"), device=dev)
    ops.gemm_tn_split(B, M, N, K, Ad, lda, Bd, N, C1)
    assert torch.equal(C0, C1)
    base = torch.randn(B, M, N, generator=g)
    C2 = base.to(dev)
    ops.gemm_tn_split(B, M, N, K, Ad, lda, Bd, N, C2, accumulate=True)
    err = float((C2.cpu().double() - (ref + base.double())).abs().max()) / float(ref.abs().max())
    assert err <= 2e-6, err
    saved = ops.GEMM_TN_SPLIT
    ops.GEMM_TN_SPLIT = False
    try:
        C3 = torch.empty((B, M, N), device=dev)
        ops.gemm_tn_split(B, M, N, K, Ad, lda, Bd, N, C3)
    finally:
        ops.GEMM_TN_SPLIT = saved
    # (the one-launch form: ONE fp32 chain over all K rows per output -- 32 640 of them at the bench shape: measurably less accurate than the split)
    assert float((C3.cpu().double() - ref).abs().max()) / float(ref.abs().max()) <= 3e-5
