"""GPU tests (-m gpu): the HIP product path, called through the C ABI, against the oracle on the
same seeded inputs and against the golden vectors of the reference.

Tolerances: k-NN indices bit-exact (up to the order of exactly tied distances, which torch.topk
leaves undefined); activations / gradients 1e-4 relative to the tensor's max (north-star bound: 1e-3).
"""
import argparse
import contextlib
import io
import json
import os

import numpy as np
import pytest
import torch

from oracle import knn as oknn
from oracle import params as oparams
from oracle import sv_ref
from tests.common import compare_case, compare_statistical, load_npz
from tests.decisions import decisions_of, tapped
from tests.golden import cases as C
from tests.golden import harness as H

pytestmark = pytest.mark.gpu
RTOL = 1e-4
OUT = os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "gpurun_out")


def _api(dev):
    import svnet_amd.models.sv_layers as L
    import svnet_amd.models.utils.sv_util as U
    return H.ModuleAPI(L, U, dev)


def test_library_is_loaded_and_native(hip_device):
    from svnet_amd import _lib
    assert _lib.lib().svnet_version() == _lib.ABI_VERSION
    assert os.path.exists(_lib.LIB_PATH)


@pytest.mark.parametrize("case", C.KNN_CASES, ids=[c[0] for c in C.KNN_CASES])
def test_knn_bit_exact(case, hip_device):
    from svnet_amd.models.utils.sv_util import knn
    name, B, N, Cc, k, layout = case
    x = C.knn_input(*case)
    xd = x.to(hip_device) if layout == "cn" else x.transpose(-1, -2).contiguous().to(hip_device).transpose(-1, -2)
    assert xd.stride() == x.stride()
    got = knn(xd, k).cpu()
    ref, pd = oknn.knn_exact(x, k, return_pd=True)
    assert oknn.tie_aware_mismatches(ref, got, pd) == 0, "vs oracle"
    gold = torch.from_numpy(load_npz("knn.npz")[name].astype(np.int64))
    assert oknn.tie_aware_mismatches(gold, got, pd) == 0, "vs reference golden"
    assert int((got != ref).sum()) == 0, "tie order must be lowest-index-first like the oracle"


@pytest.mark.parametrize("shape", [(3, 1024, 3, 20, "cn"), (2, 1024, 62, 20, "nc"), (2, 1024, 127, 20, "nc"),
                                   (1, 2048, 80, 40, "nc"), (1, 2048, 136, 40, "nc"), (2, 100, 7, 5, "nc"),
                                   (2, 77, 3, 9, "cn"), (1, 3000, 20, 33, "nc"), (2, 65, 12, 64, "nc"), (3, 40, 6, 40, "cn"),
                                   # B % 8 == 0: the XCD-aware cloud order of the launch (what the B=32 bench runs)
                                   (8, 1024, 62, 20, "nc"), (8, 1024, 127, 20, "nc"), (32, 1024, 62, 20, "nc"), (32, 1024, 127, 20, "nc"),
                                   (32, 1024, 3, 20, "cn"), (16, 2048, 80, 40, "nc"), (8, 512, 62, 20, "nc"),
                                   # N % 16 != 0 at N <= 1024: the staged (non-split) form; N % 4 != 0: the unstaged form
                                   (2, 1000, 62, 20, "nc"), (8, 1000, 127, 20, "nc"), (2, 1001, 30, 20, "nc"), (3, 600, 16, 12, "nc"),
                                   # 512 < N < 1024, N % 16 == 0: the matrix-core form with staged columns past N (clamped DMA sources), a last
                                   # workgroup of 16 queries (N % 32 != 0), k up to 64 (the four-query selection's one-at-a-time fall-back)
                                   (8, 768, 62, 20, "nc"), (2, 528, 30, 20, "nc"), (8, 1008, 127, 20, "nc"), (2, 1024, 5, 64, "nc"),
                                   (3, 1024, 131, 33, "nc")])
def test_knn_bit_exact_more_shapes(shape, hip_device):
    from svnet_amd.models.utils.sv_util import knn
    B, N, Cc, k, layout = shape
    x = C.knn_input("more_%d_%d_%d" % (N, Cc, k), B, N, Cc, k, layout)
    xd = x.to(hip_device) if layout == "cn" else x.transpose(-1, -2).contiguous().to(hip_device).transpose(-1, -2)
    got = knn(xd, k).cpu()
    ref = oknn.knn_exact(x, k)
    assert int((got != ref).sum()) == 0


@pytest.mark.parametrize("N,C_,k,dup", [(256, 12, 20, 100), (1024, 62, 20, 300), (1000, 9, 16, 70), (2048, 20, 40, 129),
                                        (1024, 62, 20, 100), (2048, 20, 40, 300), (2048, 7, 40, 200),
                                        # N = 1024: ties inside the four-query selection's 64 slots, just past them, in a short table
                                        (1024, 62, 20, 60), (1024, 3, 20, 66), (768, 62, 20, 70), (1024, 127, 20, 30)])
def test_knn_heavy_ties_take_the_lowest_index(N, C_, k, dup, hip_device):
    """`dup` copies of one point: more than 64 candidates tie with the k-th distance.  Up to the kernel's candidate slots (64 / 128 /
    256 per query at N <= 512 / 1024 / 2048) they take its sort-and-merge selection, beyond that the fallback (k passes of wave
    arg-max).  Both the oracle and the kernel break exact ties by the lowest index."""
    from svnet_amd.models.utils.sv_util import knn
    feat = C.t("knn_dup/%d_%d" % (N, C_), (2, N, C_), 0.7)
    feat[0, 5:5 + dup] = feat[0, 5]
    feat[1, N - dup:] = feat[1, 3]
    feat[1, 3 + 7] = feat[1, 3]
    x = feat.transpose(-1, -2)
    xd = feat.to(hip_device).transpose(-1, -2)
    got = knn(xd, k).cpu()
    ref = oknn.knn_exact(x, k)
    assert int((got != ref).sum()) == 0


_OPS = H.op_cases()


@pytest.mark.parametrize("name", list(_OPS), ids=list(_OPS))
def test_ops_match_oracle_and_golden(name, hip_device):
    if name in ("stn_bin_train",):
        # six binary layers deep: a 1e-6 difference flips a sign() one layer later, so the HIP run's decisions are replayed into
        # the oracle (exact-sign mode), each one certified as a knife edge, and outputs AND gradient norms are compared element-wise
        with tapped() as tap:
            got = H.to_numpy(_OPS[name](_api(hip_device)))
        dec = decisions_of(tap, tau=2e-4)               # (BatchNorms over 3 per-cloud rows: the PointNet callers' conditioning)
        orc = H.to_numpy(_OPS[name](H.OracleAPI(decisions=dec, exact_ste=True)))
        dec.check()
        assert set(got) == set(orc)
        compare_case(got, orc, 1e-3, name + " vs oracle (decisions replayed)")
        return
    got = H.to_numpy(_OPS[name](_api(hip_device)))
    orc = H.to_numpy(_OPS[name](H.OracleAPI()))
    assert set(got) == set(orc)
    compare_case(got, orc, RTOL, name + " vs oracle")
    gold = load_npz("ops.npz")
    ref = {k.split("/", 1)[1]: gold[k] for k in gold.files if k.startswith(name + "/")}
    compare_case(got, ref, RTOL, name + " vs golden")


def _build(model, binary, k, dev, state):
    import svnet_amd.models as M
    cls, nc = {"sv_dgcnn_cls": (M.SV_DGCNN_CLS, 40), "sv_pointnet_cls": (M.SV_PointNet_CLS, 40),
               "sv_dgcnn_pseg": (M.SV_DGCNN_PSEG, 50), "sv_pointnet_pseg": (M.SV_PointNet_PSEG, 50)}[model]
    with contextlib.redirect_stdout(io.StringIO()):
        m = cls(argparse.Namespace(k=k, binary=binary, dropout=0.0), nc)
    for mod in m.modules():
        if isinstance(mod, torch.nn.Dropout):
            mod.p = 0.0
    m.load_state_dict(state, strict=True)
    return m.to(dev)


def _oracle_forward(model, binary, k, x, l, P, ctx):
    return {"sv_dgcnn_cls": lambda: sv_ref.sv_dgcnn_cls(x, P, k, binary, ctx),
            "sv_pointnet_cls": lambda: sv_ref.sv_pointnet_cls(x, P, k, binary, ctx),
            "sv_dgcnn_pseg": lambda: sv_ref.sv_dgcnn_pseg(x, l, P, k, binary, ctx),
            "sv_pointnet_pseg": lambda: sv_ref.sv_pointnet_pseg(x, l, P, k, binary, ctx)}[model]()


@pytest.mark.parametrize("case", C.MODEL_CASES, ids=[c[0] for c in C.MODEL_CASES])
def test_models_eval_match_golden(case, hip_device):
    """Eval-mode logits of the full model, element-wise at 1e-3 of the logit range:
      * against the oracle with the HIP run's discrete decisions (neighbour lists, binarized signs, max-pool arg-max) replayed,
        every disagreement certified as a knife edge of the oracle's own arithmetic (tests/decisions.py; thresholds include the
        fp32 oracle's measured distance from a float64 run of itself) - fp AND binary models;
      * against the reference's golden logits: fp models element-wise; binary models statistically (the golden run took its own
        decisions at its own knife edges: a handful of logits may sit on the other side of one)."""
    tag, model, binary, B, N, k = case
    gold = load_npz("models.npz")
    P = oparams.synthetic_params(model, binary=binary, seed=C.SEED)
    x, l, y = C.model_inputs(tag, model, B, N)
    m = _build(model, binary, k, hip_device, P).eval()
    with tapped() as tap, torch.no_grad():
        out = (m(x.to(hip_device), l.to(hip_device)) if l is not None else m(x.to(hip_device))).cpu().numpy()
    with torch.no_grad():
        dec64 = decisions_of(tap)
        dec64.value_record = {"knn": [], "signs": [], "pools": []}
        ctx64 = sv_ref.Ctx(train=False)
        ctx64.decisions = dec64
        P64 = {n: (t.double() if t.is_floating_point() else t) for n, t in P.items()}
        _oracle_forward(model, binary, k, x.double(), None if l is None else l.double(), P64, ctx64)
        dec = decisions_of(tap)
        dec.truth = dec64.value_record
        ctx = sv_ref.Ctx(train=False)
        ctx.decisions = dec
        lo = _oracle_forward(model, binary, k, x, l, P, ctx).numpy()
    cert = dec.check()
    os.makedirs(OUT, exist_ok=True)
    with open(os.path.join(OUT, "eval_logits_replay_%s.json" % tag), "w") as f:
        json.dump({"logits_err_vs_oracle_replayed": H.max_rel_err(out, lo), "replayed_decisions": cert}, f, indent=0)
    assert H.max_rel_err(out, lo) < 1e-3, (H.max_rel_err(out, lo), cert)
    ref = gold[tag + "/logits_eval"]
    err = np.abs(out - ref) / np.abs(ref).max()
    if binary:
        assert np.median(err) < 1e-3 and (err > 5e-2).mean() < 0.02, (np.median(err), err.max())
    else:
        assert err.max() < 1e-3, err.max()


def test_rotation_invariance_full_size(hip_device):
    """Size-independent property at BASELINE's full size (B=32,N=1024,k=20): eval logits of the fp model do not
    depend on an SO(3) rotation of the clouds (binary models: identical up to sign-flip chaos, checked statistically)."""
    from svnet_amd import synth
    P = oparams.synthetic_params("sv_dgcnn_cls", binary=False, seed=C.SEED)
    m = _build("sv_dgcnn_cls", False, 20, hip_device, P).eval()
    x = torch.from_numpy(synth.cloud_batch(C.SEED, 3, 0, 32, 1024)).to(hip_device)
    R = torch.from_numpy(synth.random_rotation(11, 5)).float().to(hip_device)
    with torch.no_grad():
        a = m(x)
        b = m(torch.einsum("ij,bjn->bin", R, x).contiguous())
    assert float((a - b).abs().max() / a.abs().max()) < 1e-3


def test_knn_properties_full_size(hip_device):
    """At full size: slot 0 is the point itself, neighbours are distinct and sorted by distance."""
    from svnet_amd import synth
    from svnet_amd.models.utils.sv_util import knn
    x = torch.from_numpy(synth.cloud_batch(C.SEED, 4, 0, 32, 1024)).to(hip_device)
    idx = knn(x, 20)
    assert (idx[..., 0] == torch.arange(1024, device=hip_device).view(1, -1)).all()
    srt = idx.sort(dim=-1)[0]
    assert (srt[..., 1:] != srt[..., :-1]).all()
    pts = x.transpose(1, 2)                                             # [B,N,3]
    nb = torch.gather(pts.unsqueeze(1).expand(-1, 1024, -1, -1), 2, idx.unsqueeze(-1).expand(-1, -1, -1, 3))
    d = ((nb - pts.unsqueeze(2)) ** 2).sum(-1)
    assert (d[..., 1:] >= d[..., :-1] - 1e-6).all()


def test_cpu_tensors_are_refused():
    from svnet_amd.models.utils.sv_util import knn
    with pytest.raises(RuntimeError):
        knn(torch.zeros(1, 3, 8), 2)


# ----------------------------------------------------------------------------- MFMA GEMM paths (large-M shapes)

def _planes_row_sliced(x_b, dev):
    """ternary [M,K] (cpu) -> row-sliced sign / nz planes [ceil(M/64), K] int64 on dev"""
    M, K = x_b.shape
    Mp = (M + 63) // 64 * 64
    pad = torch.zeros(Mp, K)
    pad[:M] = x_b
    bits = (1 << torch.arange(64, dtype=torch.int64)).view(1, 64, 1)
    sg = ((pad > 0).view(-1, 64, K).long() * bits).sum(1)
    nz = ((pad != 0).view(-1, 64, K).long() * bits).sum(1)
    return sg.to(dev), nz.to(dev)


@pytest.mark.parametrize("M,K,N", [(1000, 42, 21), (5000, 128, 254), (777, 20, 10), (4096, 505, 512), (300, 83, 170)])
def test_gemm_rows_mfma_exact_weights(M, K, N, hip_device):
    """rows x sign-weights on bf16 MFMA with the exact 3-way split, incl. a_scale, col_scale, STE mask, column sums."""
    from svnet_amd import _ops
    g = torch.Generator().manual_seed(M + K)
    A = torch.randn(M, K, generator=g)
    Wb = torch.sign(torch.randn(N, K, generator=g))
    Wb[0, 0] = 0.0
    asc = torch.rand(K, generator=g) + 0.5
    csc = torch.rand(N, generator=g) + 0.5
    keep = (torch.rand(M, N, generator=g) > 0.3).float()
    msg, _ = _planes_row_sliced(keep, hip_device)
    ref = ((A * asc).double() @ Wb.t().double() * csc.double()) * keep.double()
    C = torch.empty(M, N, device=hip_device)
    cs = torch.zeros(_ops._sliced_len(N), device=hip_device)             # sliced accumulator (svnet_hip.h): totals by svnet_slices_sum_f32
    _ops.gemm(M, N, K, A=A.to(hip_device), a_rs=K, a_cs=1, a_scale=asc.to(hip_device), B=Wb.to(hip_device), b_rs=1, b_cs=K,
              b_exact=True, C=C, ldc=N, col_scale=csc.to(hip_device), mask=msg, col_sum=cs)
    _ops.call("svnet_slices_sum_f32", _ops._p(cs), N, _ops._stream())
    assert H.max_rel_err(C.cpu().numpy(), ref.numpy()) < 2e-6
    assert H.max_rel_err(cs[:N].cpu().numpy(), ref.sum(0).numpy()) < 1e-5
    # NN orientation (dx = g . w_b): B(k,j) = Wb2[k*N + j]
    Wb2 = torch.sign(torch.randn(K, N, generator=g))
    C2 = torch.empty(M, N, device=hip_device)
    _ops.gemm(M, N, K, A=A.to(hip_device), a_rs=K, a_cs=1, B=Wb2.to(hip_device), b_rs=N, b_cs=1, b_exact=True, C=C2, ldc=N)
    assert H.max_rel_err(C2.cpu().numpy(), (A.double() @ Wb2.double()).numpy()) < 2e-6


@pytest.mark.parametrize("M,K,N,bias,nn", [(32768, 2044, 512, False, False), (4096, 340, 170, True, False), (3000, 127, 512, False, True),
                                            (1024, 21, 170, True, True), (5000, 1022, 40, True, False),
                                            # ragged K / unaligned rows through the LDS-tiled kernel (scalar A loads), many column groups
                                            (8200, 127, 512, True, False), (4100, 1022, 341, False, True), (4096, 512, 2044, False, True),
                                            # every column-group width x A load width of the pre-split kernel (K % 4 == 0 / even / odd), short K, tiny N
                                            (6000, 170, 340, True, False), (4099, 10, 10, False, False), (3000, 21, 170, True, True),
                                            (2048, 170, 21, False, True), (2500, 62, 32, True, False), (4096, 126, 64, False, False),
                                            (1100, 33, 129, True, True), (1024, 8, 8, False, False), (2000, 64, 100, True, True)])
def test_gemm_rows_mfma_general_fp32_weights(M, K, N, bias, nn, hip_device):
    """rows x GENERAL fp32 weights (the fp layers' F.linear and its input gradient, sv_layers.py:30-31): B split exactly into three
    bf16 pieces, A split once per element by the staging threads, the six leading bf16 products - fp32-GEMM accuracy against a float64 product, both orientations
    (x W^T with W [N,K]; g W with W [K,N]), with and without the bias."""
    from svnet_amd import _ops
    g = torch.Generator().manual_seed(M + K + N)
    A = torch.randn(M, K, generator=g)
    W = torch.randn(K, N, generator=g) if nn else torch.randn(N, K, generator=g)
    b = torch.randn(N, generator=g) if bias else None
    C = torch.full((M, N), 7.0, device=hip_device)
    if nn:
        _ops.gemm(M, N, K, A=A.to(hip_device), a_rs=K, a_cs=1, B=W.to(hip_device), b_rs=N, b_cs=1, C=C, ldc=N, bias=None if b is None else b.to(hip_device))
        ref = A.double() @ W.double()
    else:
        _ops.gemm(M, N, K, A=A.to(hip_device), a_rs=K, a_cs=1, B=W.to(hip_device), b_rs=1, b_cs=K, C=C, ldc=N, bias=None if b is None else b.to(hip_device))
        ref = A.double() @ W.double().t()
    if b is not None:
        ref = ref + b.double()
    assert H.max_rel_err(C.cpu().numpy(), ref.numpy()) < 2e-6
    # a column slice of a wider row (lda > K, unaligned start) accumulated into C with a scale
    if K > 9:
        Ad, Wd = A.to(hip_device), W.to(hip_device)
        Ks = K - 5
        C2 = torch.full((M, N), 0.5, device=hip_device)
        if nn:
            _ops.gemm(M, N, Ks, A=Ad[:, 3:], a_rs=K, a_cs=1, B=Wd[3:], b_rs=N, b_cs=1, C=C2, ldc=N, alpha=-0.75, accumulate=True)
            ref2 = 0.5 - 0.75 * (A[:, 3:3 + Ks].double() @ W[3:3 + Ks].double())
        else:
            _ops.gemm(M, N, Ks, A=Ad[:, 3:], a_rs=K, a_cs=1, B=Wd[:, 3:], b_rs=1, b_cs=K, C=C2, ldc=N, alpha=-0.75, accumulate=True)
            ref2 = 0.5 - 0.75 * (A[:, 3:3 + Ks].double() @ W[:, 3:3 + Ks].double().t())
        assert H.max_rel_err(C2.cpu().numpy(), ref2.numpy()) < 2e-6


@pytest.mark.parametrize("R,P,Q", [(3000, 70, 200), (70000, 128, 254), (2049, 10, 20), (5000, 170, 83), (1500, 512, 505),
                                   (4096, 512, 2044)])                 # (a weight gradient wider than 1024 columns: conv_fuse of sv_pointnet_cls)
def test_gemm_tn_mfma(R, P, Q, hip_device):
    """weight-gradient products: reduction over R rows, fp32 x fp32 (6-term split) and fp32 x ternary planes."""
    from svnet_amd import _ops
    g = torch.Generator().manual_seed(R + P)
    G = torch.randn(R, P, generator=g)
    X = torch.randn(R, Q, generator=g)
    out = torch.empty(P, Q, device=hip_device)
    _ops.gemm(P, Q, R, A=G.to(hip_device), a_rs=1, a_cs=P, B=X.to(hip_device), b_rs=Q, b_cs=1, C=out, ldc=Q)
    assert H.max_rel_err(out.cpu().numpy(), (G.double().t() @ X.double()).numpy()) < 5e-6
    Xb = torch.sign(torch.randn(R, Q, generator=g)) * (torch.rand(R, Q, generator=g) > 0.1)
    sg, nz = _planes_row_sliced(Xb, hip_device)
    gx = torch.empty(P, Q, device=hip_device)                    # GX[p,q] = sum_r G[r,p] Xb[r,q], written through C strides
    _ops.gemm(Q, P, R, a_planes=(sg, nz), B=G.to(hip_device), b_rs=P, b_cs=1, C=gx, ldc=1, c_cs=Q)
    assert H.max_rel_err(gx.cpu().numpy(), (G.double().t() @ Xb.double()).numpy()) < 5e-6


@pytest.mark.parametrize("M,K,O", [(1500, 254, 128), (700, 124, 32), (330, 505, 512),
                                   # K > 1024: the WIDE forward and the ternary weight-gradient product with more than 32 column tiles
                                   (1500, 2044, 512), (8, 2044, 512), (2048, 2144, 256), (32, 2044, 512), (70000, 544, 512)])
def test_binlinear_large_train_matches_oracle(M, K, O, hip_device):
    """Linear(bw,ba) at conv-sized rows: XNOR forward, row-sliced planes, MFMA backward; against the oracle."""
    from svnet_amd.models.sv_layers import Linear
    params = H.module_params("Linear", (K, O, False, True, True), "big%d" % M)
    x = C.t("big_lin/%d" % M, (M, K), 1.0)
    x.view(-1)[::11] = 0.0
    params["beta"][:, ::5] = 0.0
    r = C.t("big_lin_r/%d" % M, (M, O))
    m = Linear(K, O, False, bw=True, ba=True)
    m.load_state_dict(params)
    m = m.to(hip_device).train()
    xd = x.to(hip_device).requires_grad_(True)
    y = m(xd)
    (y * r.to(hip_device)).sum().backward()
    P = {"m." + k: v.clone().requires_grad_(True) for k, v in params.items()}
    xo = x.clone().requires_grad_(True)
    yo = sv_ref.linear(xo, P, "m", bw=True, ba=True, ctx=sv_ref.Ctx(train=True))
    (yo * r).sum().backward()
    got = {"out0": y.detach().cpu().numpy(), "dx0": xd.grad.cpu().numpy(), "d:weight": m.weight.grad.cpu().numpy(),
           "d:beta": m.beta.grad.cpu().numpy(), "d:scale": m.scale.grad.cpu().numpy()}
    ref = {"out0": yo.detach().numpy(), "dx0": xo.grad.numpy(), "d:weight": P["m.weight"].grad.numpy(),
           "d:beta": P["m.beta"].grad.numpy(), "d:scale": P["m.scale"].grad.numpy()}
    compare_case(got, ref, RTOL, "binlinear large")


def test_bwlinear_large_train_matches_oracle(hip_device):
    from svnet_amd.models.sv_layers import Linear
    K, O, rows = 42, 42, (40, 33, 3)
    params = H.module_params("Linear", (K, O, False, True, False), "bigbw")
    x = C.t("big_bw/x", rows + (K,), 1.0)
    r = C.t("big_bw/r", rows + (O,))
    m = Linear(K, O, False, bw=True)
    m.load_state_dict(params)
    m = m.to(hip_device).train()
    xd = x.to(hip_device).requires_grad_(True)
    (m(xd) * r.to(hip_device)).sum().backward()
    P = {"m." + k: v.clone().requires_grad_(True) for k, v in params.items()}
    xo = x.clone().requires_grad_(True)
    (sv_ref.linear(xo, P, "m", bw=True, ctx=sv_ref.Ctx(train=True)) * r).sum().backward()
    got = {"dx0": xd.grad.cpu().numpy(), "d:weight": m.weight.grad.cpu().numpy(), "d:scale": m.scale.grad.cpu().numpy()}
    ref = {"dx0": xo.grad.numpy(), "d:weight": P["m.weight"].grad.numpy(), "d:scale": P["m.scale"].grad.numpy()}
    compare_case(got, ref, RTOL, "bwlinear large")


def test_conv1d_binary_partseg_head_shape_matches_oracle(hip_device):
    """Conv1d(2144 -> 256, binary) on channel-first [B,C,N] at N = 2048: the first layer of the part-segmentation head
    (sv_dgcnn_partseg.py:64-66), train mode, against the oracle: output, input gradient, every parameter gradient."""
    from svnet_amd.models.sv_layers import Conv1d
    Cin, Cout, B, N = 2144, 256, 2, 2048
    params = H.module_params("Conv1d", (Cin, Cout, True), "pseg_head")
    x = C.t("pseg_head/x", (B, Cin, N), 1.0)
    x.view(-1)[::13] = 0.0
    params["beta"][:, ::5] = 0.0
    r = C.t("pseg_head/r", (B, Cout, N))
    with contextlib.redirect_stdout(io.StringIO()):
        m = Conv1d(Cin, Cout, binary=True)
    m.load_state_dict(params)
    m = m.to(hip_device).train()
    xd = x.to(hip_device).requires_grad_(True)
    y = m(xd)
    (y * r.to(hip_device)).sum().backward()
    P = {"m." + k: v.clone().requires_grad_(True) for k, v in params.items()}
    xo = x.clone().requires_grad_(True)
    yo = sv_ref.conv1d(xo, P, "m", True, sv_ref.Ctx(train=True, exact_ste=True))
    (yo * r).sum().backward()
    got = {"out0": y.detach().cpu().numpy(), "dx0": xd.grad.cpu().numpy(), "d:weight": m.weight.grad.cpu().numpy(),
           "d:beta": m.beta.grad.cpu().numpy(), "d:scale": m.scale.grad.cpu().numpy()}
    ref = {"out0": yo.detach().numpy(), "dx0": xo.grad.numpy(), "d:weight": P["m.weight"].grad.numpy(),
           "d:beta": P["m.beta"].grad.numpy(), "d:scale": P["m.scale"].grad.numpy()}
    compare_case(got, ref, RTOL, "conv1d binary 2144->256 N=2048")


@pytest.mark.parametrize("mode", ["plain", "first", "cross"])
def test_graph_feature_coordinate_gradients_match_oracle(mode, hip_device):
    """get_graph_feature[_cross] with an input that requires grad (autograd of sv_util.py:51-60 / :81-86 in the reference):
    values and the gradient w.r.t. the coordinates against the oracle, on the oracle's graph."""
    from svnet_amd.models.utils import sv_util as U
    B, m, N, k = 2, 2 if mode != "cross" else 1, 40, 6
    x = C.t("gf_grad/" + mode, (B, 1, 3 * m, N), 0.7)
    idx = oknn.knn_exact(x.view(B, -1, N), k)
    r = C.t("gf_grad_r/" + mode, (B, N, k, 3, (3 if mode == "cross" else 2) * m))
    xd = x.to(hip_device).requires_grad_(True)
    xo = x.clone().requires_grad_(True)
    if mode == "cross":
        got, ref = U.get_graph_feature_cross(xd, k=k, idx=idx.to(hip_device)), sv_ref.graph_feature_cross(xo, k=k, idx=idx)
    else:
        got = U.get_graph_feature(xd, k=k, idx=idx.to(hip_device), first=(mode == "first"))
        ref = sv_ref.graph_feature(xo, k=k, idx=idx, first=(mode == "first"))
    (got * r.to(hip_device)).sum().backward()
    (ref * r).sum().backward()
    compare_case({"out0": got.detach().cpu().numpy(), "dx0": xd.grad.cpu().numpy()}, {"out0": ref.detach().numpy(), "dx0": xo.grad.numpy()},
                 RTOL, "graph feature coordinate gradients (%s)" % mode)


@pytest.mark.parametrize("shape", [(1500, 2044, 512), (2048, 505, 512), (1024, 83, 64), (3000, 1022, 341), (4096, 64, 170), (1100, 300, 600)],
                         ids=lambda s: "x".join(map(str, s)))
def test_binlinear_matrix_core_path_is_bit_identical(shape, hip_device):
    """svnet_binlinear_i8_fwd_f32 (int8 ternary operands on v_mfma_i32_32x32x32_i8, used for >= 1024 rows) against the XNOR-popcount
    kernel svnet_binlinear_fwd_f32 (itself pinned to the oracle by the op cases above) on identical inputs - exact zeros in x, W
    and beta included, ragged K and O: outputs and the three saved row-sliced planes must be bit-identical (the count is the same
    integer; sv_layers.py:35-51)."""
    from svnet_amd import _lib
    from svnet_amd._ops import _p, _stream, _words, call
    M, K, O = shape
    g = torch.Generator().manual_seed(M + K + O)
    x = torch.randn(M, K, generator=g)
    x[::7, ::5] = 0.0
    W = torch.randn(O, K, generator=g)
    W[::3, ::4] = 0.0
    beta = torch.randn(K, generator=g) * 0.1
    beta[::2] = 0.0
    sc, bias = torch.rand(O, generator=g) + 0.5, torch.randn(O, generator=g)
    x, W, beta, sc, bias = (t.to(hip_device) for t in (x, W, beta, sc, bias))
    KW = _words(K)
    ws = torch.empty((O, KW), dtype=torch.int64, device=hip_device)
    wz = torch.empty_like(ws)
    wb = torch.empty((O, K), device=hip_device)
    call("svnet_binweight_prepare_f32", _p(W), None, O, K, _p(ws), _p(wz), _p(wb), None, _stream())
    w8 = torch.empty((_lib.lib().svnet_binweight_i8_bytes(O, K),), dtype=torch.int8, device=hip_device)
    call("svnet_binweight_pack_i8", _p(W), O, K, _p(w8), _stream())
    outs = []
    for matrix_cores in (False, True):
        y = torch.full((M, O), 7.0, device=hip_device)
        pl = [torch.full(((M + 63) // 64, K), -1, dtype=torch.int64, device=hip_device) for _ in range(3)]
        if matrix_cores:
            from svnet_amd._ops import _sliced_len
            sums = torch.zeros(_sliced_len(2 * O), dtype=torch.float64, device=hip_device)       # sliced accumulator: the totals go to the first 2 O
            call("svnet_binlinear_i8_fwd_f32", _p(x), K, _p(beta), _p(w8), _p(sc), _p(bias), M, K, O, _p(y), _p(pl[0]), _p(pl[1]), _p(pl[2]),
                 _p(sums), _stream())
            call("svnet_slices_sum_f64", _p(sums), 2 * O, _stream())      # (in the product the consuming svnet_bn_finalize_f32 adds the slices up)
        else:
            call("svnet_binlinear_fwd_f32", _p(x), K, _p(beta), _p(ws), _p(wz), _p(sc), _p(bias), M, K, O, _p(y), _p(pl[0]), _p(pl[1]),
                 _p(pl[2]), _stream())
        outs.append((y, pl))
    assert torch.equal(outs[0][0], outs[1][0])
    for a, b in zip(outs[0][1], outs[1][1]):
        assert torch.equal(a, b)
    # the column sums the kernel leaves for the BatchNorm that follows: sum y and sum y^2 over the rows, from the integer counts
    yd = outs[1][0].double()
    ref = torch.cat([yd.sum(0), (yd * yd).sum(0)])
    assert float((sums[:2 * O] - ref).abs().max() / ref.abs().max()) < 1e-6
