"""GPU tests (-m gpu) of the fused edge block (csrc/edgeblock.hip) against the layer-by-layer HIP path (tier 1,
itself pinned to the oracle by test_hip_parity.py) and against the oracle, on identical inputs."""
import contextlib
import ctypes
import io

import numpy as np
import pytest
import torch

from tests.decisions import tapped

from oracle import sv_ref
from tests.common import compare_case
from tests.golden import cases as C
from tests.golden import harness as H

pytestmark = pytest.mark.gpu

# (Cs, Cv) point tables -> (Os, Ov); B, N, k
SHAPES = [((32, 10), (32, 10), 2, 96, 6), ((32, 10), (64, 21), 2, 80, 7), ((64, 21), (128, 42), 2, 130, 20),
          ((64, 24), (128, 40), 1, 70, 40),
          # E % 32 == 0 at Os = 128 with all ten column tiles in use: the one-tile-per-workgroup weight-gradient kernel (whole slabs;
          # an odd number of 32-row slabs; k = 8, the smallest its two-point row groups allow)
          ((64, 21), (128, 42), 2, 136, 20), ((64, 21), (128, 42), 1, 72, 20), ((64, 21), (128, 42), 3, 40, 8)]


def _make(shape, dev, train, tag, zero_w=False):
    from svnet_amd.models.sv_layers import SVBlock
    (Cs, Cv), (Os, Ov), B, N, k = shape
    in_dims, out_dims = (2 * Cs, 2 * Cv), (Os, Ov)
    params = H.module_params("SVBlock", (in_dims, out_dims, True), tag)
    params["linear1.beta"][:, ::4] = 0.0
    if zero_w:                                         # exact zeros among the weights: sign(0) = 0 (sv_layers.py:44-45) - the general popcount path
        params["linear1.weight"][::3, ::5] = 0.0
    with contextlib.redirect_stdout(io.StringIO()):
        blk = SVBlock(in_dims, out_dims, binary=True)
    blk.load_state_dict(params)
    blk = blk.to(dev).train(train)
    s, v = C.sv_pair(tag + "/pt", (B, N), Cs, Cv, 1.0)
    s = torch.round(s * 4) / 4                         # discrete scalars like a binary net's: exact zeros / ties
    return blk, params, s, v, (in_dims, out_dims, B, N, k)


def _run(blk, s, v, k, dev, fuse, grad=False):
    from svnet_amd import config
    from svnet_amd.models.utils.sv_util import get_graph_feature_sv, svpool
    old = config.FUSE_EDGE_BLOCKS
    config.FUSE_EDGE_BLOCKS = fuse
    try:
        sd = s.to(dev).requires_grad_(grad)
        vd = v.to(dev).requires_grad_(grad)
        out = svpool(blk(get_graph_feature_sv((sd, vd), k=k)))
    finally:
        config.FUSE_EDGE_BLOCKS = old
    return sd, vd, out


@pytest.mark.parametrize("train", [False, True], ids=["eval", "train"])
@pytest.mark.parametrize("shape", SHAPES, ids=[str(i) for i in range(len(SHAPES))])
def test_fused_forward_matches_layerwise_and_oracle(shape, train, hip_device):
    blk, params, s, v, (in_dims, out_dims, B, N, k) = _make(shape, hip_device, train, "fused_fwd")
    with torch.no_grad():
        _, _, (fs, fv) = _run(blk, s, v, k, hip_device, True)
        bufs_f = {n: b.detach().cpu().numpy().copy() for n, b in blk.named_buffers() if b.is_floating_point()}
        blk.load_state_dict(params)
        _, _, (ls, lv) = _run(blk, s, v, k, hip_device, False)
        bufs_l = {n: b.detach().cpu().numpy() for n, b in blk.named_buffers() if b.is_floating_point()}
    got = {"out0": fs.cpu().numpy(), "out1": fv.cpu().numpy()}
    ref = {"out0": ls.cpu().numpy(), "out1": lv.cpu().numpy()}
    compare_case(got, ref, 1e-4, "fused vs layerwise")
    if train:
        compare_case({"buf:" + n: x for n, x in bufs_f.items()}, {"buf:" + n: x for n, x in bufs_l.items()}, 1e-4, "running stats")
    # oracle on the same inputs
    P = {"m." + n: t.clone() for n, t in params.items()}
    ctx = sv_ref.Ctx(train=train)
    with torch.no_grad():
        os_, ov = sv_ref.svpool(sv_ref.svblock(sv_ref.graph_feature_sv((s, v), k=k), P, "m", True, ctx))
    compare_case(got, {"out0": os_.numpy(), "out1": ov.numpy()}, 1e-4, "fused vs oracle")


@pytest.mark.parametrize("train", [False, True], ids=["eval", "train"])
@pytest.mark.parametrize("shape", [SHAPES[0], SHAPES[2]], ids=["narrow", "wide"])
def test_fused_forward_with_zero_weights_takes_the_general_popcount_path(shape, train, hip_device):
    """The forward edge kernels skip the weights' non-zero plane when the packing kernel found no exact zero in linear1.weight (a device
    flag: config.EDGE_DENSE_WEIGHTS, csrc/edgeblock.hip) - the case of every trained or freshly initialised layer.  With zeros among the
    weights (sign(0) = 0: ternary weights, sv_layers.py:44-45) the general path must run and agree with the layer-wise kernels bit for
    bit on the pooled scalars (same integer counts) - and the dense path must NOT be taken (it would count the zero weights as -1)."""
    from svnet_amd import _ops
    blk, params, s, v, (in_dims, out_dims, B, N, k) = _make(shape, hip_device, train, "fused_zero_w", zero_w=True)
    with torch.no_grad():
        _, _, (fs, fv) = _run(blk, s, v, k, hip_device, True)
        _, _, (ls, lv) = _run(blk, s, v, k, hip_device, False)
    assert float((fs - ls).abs().max()) <= 1e-5 * float(ls.abs().max())
    assert float((fv - lv).abs().max()) <= 1e-5 * float(lv.abs().max())
    blk2, _, _, _, _ = _make(shape, hip_device, train, "fused_zero_w", zero_w=False)
    with torch.no_grad():
        _, _, (ds, _) = _run(blk2, s, v, k, hip_device, True)
    assert float((ds - fs).abs().max()) > 1e-3 * float(fs.abs().max())      # (the zeros do change the result: the comparison above has teeth)


@pytest.mark.parametrize("shape", SHAPES, ids=[str(i) for i in range(len(SHAPES))])
def test_fused_backward_matches_layerwise(shape, hip_device):
    """Every gradient of the fused block (recompute-based single edge pass) against autograd through the layer-wise
    HIP path on identical inputs and upstream gradients."""
    grads = {}
    for fuse in (True, False):
        blk, params, s, v, (in_dims, out_dims, B, N, k) = _make(shape, hip_device, True, "fused_bwd")
        sd, vd, (os_, ov) = _run(blk, s, v, k, hip_device, fuse, grad=True)
        rs = C.t("fused_bwd/rs", tuple(os_.shape)).to(hip_device)
        rv = C.t("fused_bwd/rv", tuple(ov.shape)).to(hip_device)
        ((os_ * rs).sum() + (ov * rv).sum()).backward()
        g = {"dx0": sd.grad.cpu().numpy(), "dx1": vd.grad.cpu().numpy()}
        for n, p in blk.named_parameters():
            g["d:" + n] = p.grad.cpu().numpy()
        grads[fuse] = g
    assert set(grads[True]) == set(grads[False])
    compare_case(grads[True], grads[False], 1e-3, "fused vs layerwise backward")


@pytest.mark.parametrize("chunk", [32, None, 0], ids=["chunk32", "default_chunk", "whole_lists"])
def test_fused_backward_with_hub_points(chunk, hip_device):
    """A graph with hubs: the first few points sit at the origin of feature space, the others far out, so every point's
    neighbours are (itself and) those few - reverse lists of ~N entries, summed in chunks by the gather kernel's second launch
    (svnet_knn_reverse_i32's overflow items), against the layer-wise path on the same inputs.  With config.GATHER_CHUNK = 32 (four
    overflow chunks per hub), the default (one) and 0 (one wave walks the whole list, no second launch)."""
    from svnet_amd import config
    shape = ((32, 10), (64, 21), 2, 150, 8)
    grads, deg = {}, None
    old_chunk = config.GATHER_CHUNK
    if chunk is not None:
        config.GATHER_CHUNK = chunk
    try:
        grads, deg = _hub_grads(shape, hip_device)
    finally:
        config.GATHER_CHUNK = old_chunk
    assert int(deg.max()) > 100, "the construction should make hubs (max in-degree %d)" % int(deg.max())
    compare_case(grads[True], grads[False], 1e-3, "fused vs layerwise backward, hub graph")


def _hub_grads(shape, hip_device):
    grads, deg = {}, None
    for fuse in (True, False):
        blk, params, s, v, (in_dims, out_dims, B, N, k) = _make(shape, hip_device, True, "fused_hub")
        s, v = s.clone(), v.clone()
        s[:, k:] *= 6.0
        v[:, k:] *= 6.0
        s[:, :k] *= 0.05
        v[:, :k] *= 0.05
        sd, vd, (os_, ov) = _run(blk, s, v, k, hip_device, fuse, grad=True)
        if deg is None:
            from svnet_amd import _ops
            idx = _ops.knn_sv(s.to(hip_device), v.to(hip_device), k)
            deg = torch.bincount(idx[0].reshape(-1).cpu(), minlength=N)
        rs = C.t("fused_hub/rs", tuple(os_.shape)).to(hip_device)
        rv = C.t("fused_hub/rv", tuple(ov.shape)).to(hip_device)
        ((os_ * rs).sum() + (ov * rv).sum()).backward()
        g = {"dx0": sd.grad.cpu().numpy(), "dx1": vd.grad.cpu().numpy()}
        for n, p in blk.named_parameters():
            g["d:" + n] = p.grad.cpu().numpy()
        grads[fuse] = g
    return grads, deg


# ----------------------------------------------------------------------------- fused FIRST layer (xyz -> init_scalar -> conv1 -> pool)

class _FirstLayer(torch.nn.Module):
    """nc = 2: get_graph_feature -> Vector2Scalar(2,3) -> SVBlock((6,2), .) (the DGCNN callers' conv1); nc = 3: get_graph_feature_cross
    -> Vector2Scalar(3,3) -> SVBlock((9,3), .) (the PointNet callers' conv_pos, sv_pointnet_cls.py:35-40)."""

    def __init__(self, out_dims, nc=2):
        super().__init__()
        from svnet_amd.models.sv_layers import SVBlock, Vector2Scalar
        self.nc = nc
        with contextlib.redirect_stdout(io.StringIO()):
            self.init_scalar = Vector2Scalar(nc, 3)
            self.conv1 = SVBlock((3 * nc, nc), out_dims)

    def forward(self, x, k):
        from svnet_amd.models.utils.sv_util import get_graph_feature, get_graph_feature_cross, svpool
        v = (get_graph_feature if self.nc == 2 else get_graph_feature_cross)(x.unsqueeze(1), k=k)
        return svpool(self.conv1((self.init_scalar(v), v)))


def _first_layer_params(out_dims, tag, nc=2):
    p = {"init_scalar." + n: t for n, t in H.module_params("Vector2Scalar", (nc, 3, False, False), tag + "/v2s").items()}
    p.update({"conv1." + n: t for n, t in H.module_params("SVBlock", ((3 * nc, nc), out_dims, False), tag + "/blk").items()})
    return p


@pytest.mark.parametrize("train", [False, True], ids=["eval", "train"])
@pytest.mark.parametrize("cfg", [((32, 10), 2, 96, 6, 2), ((32, 16), 1, 150, 40, 2), ((32, 10), 3, 64, 20, 2),
                                 ((32, 10), 2, 96, 6, 3), ((32, 10), 3, 200, 20, 3)], ids=["a", "b", "c", "cross_a", "cross_b"])
def test_fused_first_layer_matches_layerwise_and_oracle(cfg, train, hip_device):
    from svnet_amd import config
    out_dims, B, N, k, nc = cfg
    params = _first_layer_params(out_dims, "first", nc)
    x = C.small_cloud(B, N, 5)
    res = {}
    for fuse in (True, False):
        m = _FirstLayer(out_dims, nc)
        m.load_state_dict(params)
        m = m.to(hip_device).train(train)
        old = config.FUSE_EDGE_BLOCKS
        config.FUSE_EDGE_BLOCKS = fuse
        try:
            if train:
                s, v = m(x.to(hip_device), k)
                rs, rv = C.t("first/rs", tuple(s.shape)).to(hip_device), C.t("first/rv", tuple(v.shape)).to(hip_device)
                ((s * rs).sum() + (v * rv).sum()).backward()
            else:
                with torch.no_grad():
                    s, v = m(x.to(hip_device), k)
        finally:
            config.FUSE_EDGE_BLOCKS = old
        r = {"out0": s.detach().cpu().numpy(), "out1": v.detach().cpu().numpy()}
        if train:
            r.update({"d:" + n: p.grad.cpu().numpy() for n, p in m.named_parameters()})
            r.update({"buf:" + n: b.detach().cpu().numpy() for n, b in m.named_buffers() if b.is_floating_point()})
        res[fuse] = r
    assert set(res[True]) == set(res[False])
    compare_case(res[True], res[False], 1e-4 if not train else 1e-3, "fused first layer vs layerwise")
    # oracle forward on the same input
    P = {n: t.clone() for n, t in params.items()}
    ctx = sv_ref.Ctx(train=train)
    with torch.no_grad():
        ve = (sv_ref.graph_feature if nc == 2 else sv_ref.graph_feature_cross)(x.unsqueeze(1), k=k)
        s0 = sv_ref.vector2scalar(ve, P, "init_scalar")
        os_, ov = sv_ref.svpool(sv_ref.svblock((s0, ve), P, "conv1", False, ctx))
    compare_case({"out0": res[True]["out0"], "out1": res[True]["out1"]}, {"out0": os_.numpy(), "out1": ov.numpy()}, 1e-4, "fused first layer vs oracle")


# ----------------------------------------------------------------------------- pieces of the fused backward, through the C ABI

def test_knn_reverse_lists_are_a_permutation_of_the_edges(hip_device):
    """svnet_knn_reverse_i32: every edge e = i*k + t appears exactly once, in the list of the point idx[e] names
    (cloud-local ids, sv_util.py:19-25), and edges with an out-of-range id appear nowhere."""
    from svnet_amd import _lib
    from svnet_amd._ops import _p, _stream, call
    B, N, k = 3, 200, 7
    g = torch.Generator().manual_seed(5)
    idx = torch.randint(0, N, (B, N, k), generator=g, dtype=torch.int64)
    idx[1, 17, 3] = N + 5                                     # a corrupted id
    idx[2, 0, 0] = -1
    d_idx = idx.to(hip_device)
    P, E = B * N, B * N * k
    rng = torch.empty(2 * P, dtype=torch.int32, device=hip_device)
    red = torch.full((E,), -7, dtype=torch.int32, device=hip_device)
    src = torch.full((E,), -7, dtype=torch.int32, device=hip_device)
    call("svnet_knn_reverse_i32", _p(d_idx), B, N, k, _p(rng), _p(red), _p(src), 0, None, None, _stream())
    rng, red, src = rng.cpu().numpy().reshape(P, 2), red.cpu().numpy(), src.cpu().numpy()
    flat = idx.reshape(-1).numpy()
    seen = np.zeros(E, dtype=np.int64)
    for j in range(P):
        b = j // N
        for e in red[rng[j, 0]:rng[j, 1]]:
            assert 0 <= e < E and e // (N * k) == b            # an edge of the same cloud ...
            assert flat[e] == j - b * N                         # ... that points at j
            seen[e] += 1
        assert np.array_equal(src[rng[j, 0]:rng[j, 1]], red[rng[j, 0]:rng[j, 1]] // k)   # source point of every listed edge
    valid = (flat >= 0) & (flat < N)
    assert np.array_equal(seen, valid.astype(np.int64))
    # the optional overflow items: every chunk after the first of every list longer than `chunk` entries, once each
    chunk = 4
    items = torch.full((2 * (2 * E // chunk + 1),), -1, dtype=torch.int32, device=hip_device)
    count = torch.zeros(1, dtype=torch.int32, device=hip_device)
    rng2 = torch.empty(2 * P, dtype=torch.int32, device=hip_device)
    red2, src2 = torch.empty(E, dtype=torch.int32, device=hip_device), torch.empty(E, dtype=torch.int32, device=hip_device)
    call("svnet_knn_reverse_i32", _p(d_idx), B, N, k, _p(rng2), _p(red2), _p(src2), chunk, _p(items), _p(count), _stream())
    n_items = int(count.item())
    got = sorted(map(tuple, items.cpu().numpy()[:2 * n_items].reshape(n_items, 2).tolist()))
    lens = rng[:, 1] - rng[:, 0]
    want = sorted((j, c) for j in range(P) for c in range(1, (int(lens[j]) - 1) // chunk + 1) if lens[j] > chunk)
    assert np.array_equal(rng2.cpu().numpy().reshape(P, 2), rng) and got == want and n_items > 0


def test_pool_max_mean_matches_torch(hip_device):
    """_ops.PoolMaxMean = cat(max, mean) over the points (sv_dgcnn_cls.py:72-74), first index on ties, one shared backward."""
    from svnet_amd import _ops
    g = torch.Generator().manual_seed(11)
    x = torch.randn(4, 300, 70, generator=g)
    x[1, 5] = x[1, 200]                                        # ties: the first index must receive the gradient
    w = torch.randn(4, 140, generator=g)
    xd = x.to(hip_device).requires_grad_(True)
    out = _ops.PoolMaxMean.apply(xd, 1)
    (out * w.to(hip_device)).sum().backward()
    xr = x.clone().requires_grad_(True)
    ref = torch.cat((xr.max(dim=1).values, xr.mean(dim=1)), dim=1)
    # torch's max backward sends the gradient to ONE arg-max; build the first-index rule explicitly
    arg = torch.zeros(4, 70, dtype=torch.int64)
    for b in range(4):
        for c in range(70):
            arg[b, c] = int(torch.nonzero(x[b, :, c] == x[b, :, c].max())[0])
    gref = (w[:, 70:] / 300.0).unsqueeze(1).expand(4, 300, 70).clone()
    gref.scatter_add_(1, arg.unsqueeze(1), w[:, :70].unsqueeze(1))
    assert torch.allclose(out.detach().cpu(), ref.detach(), atol=1e-6, rtol=1e-6)
    assert torch.allclose(xd.grad.cpu(), gref, atol=1e-6, rtol=1e-6)


@pytest.mark.parametrize("training", [True, False], ids=["train", "eval"])
def test_bn_pool_fused_equals_bn_act_then_pool(training, hip_device):
    """_ops.GlobalMaxMeanPoolBN (conv5's bn1 + LeakyReLU inside the classifier's global [max | mean] pooling pass,
    sv_layers.py:189-190 + sv_dgcnn_cls.py:69-74) against the unfused chain BNAct -> GlobalMaxMeanPool on the same inputs:
    outputs bit-identical, running statistics identical, every gradient to 1e-5 of its largest element."""
    from svnet_amd import _ops
    g = torch.Generator().manual_seed(21)
    B, N, Ca, Cb = 3, 300, 70, 45
    y = torch.randn(B, N, Ca, generator=g) * 2.0
    y[1, 7] = y[1, 211]                                      # ties: the first index takes the max gradient
    b = torch.randn(B, N, Cb, generator=g)
    w = torch.randn(B, 2 * (Ca + Cb), generator=g).to(hip_device)
    res = {}
    for fused in (True, False):
        bn = torch.nn.BatchNorm1d(Ca).to(hip_device).train(training)
        with torch.no_grad():
            bn.weight.copy_(torch.linspace(-1.0, 1.5, Ca))    # both signs of gamma
            bn.bias.copy_(torch.linspace(0.3, -0.3, Ca))
            bn.running_mean.copy_(torch.linspace(-0.2, 0.2, Ca))
            bn.running_var.copy_(torch.linspace(0.5, 2.0, Ca))
        yd, bd = y.to(hip_device).requires_grad_(True), b.to(hip_device).requires_grad_(True)
        nbt = bn.num_batches_tracked if training else None
        if fused:
            out = _ops.GlobalMaxMeanPoolBN.apply(yd, bd, bn.weight, bn.bias, bn.running_mean, bn.running_var, training, 1, 0.2, nbt,
                                                 bn.eps, bn.momentum)
        else:
            a = _ops.BNAct.apply(yd, bn.weight, bn.bias, bn.running_mean, bn.running_var, training, 1, 0.2, nbt, bn.eps, bn.momentum)
            out = _ops.GlobalMaxMeanPool.apply(a, bd)
        (out * w).sum().backward()
        res[fused] = dict(out=out.detach().cpu(), dy=yd.grad.cpu(), db=bd.grad.cpu(), dgamma=bn.weight.grad.cpu(), dbeta=bn.bias.grad.cpu(),
                          rm=bn.running_mean.cpu().clone(), rv=bn.running_var.cpu().clone(), nbt=int(bn.num_batches_tracked))
    assert torch.equal(res[True]["out"], res[False]["out"])
    assert torch.equal(res[True]["rm"], res[False]["rm"]) and torch.equal(res[True]["rv"], res[False]["rv"])
    assert res[True]["nbt"] == res[False]["nbt"] == (1 if training else 0)
    for n in ("dy", "db", "dgamma", "dbeta"):
        a_, b_ = res[True][n], res[False][n]
        assert float((a_ - b_).abs().max()) <= 1e-5 * max(float(b_.abs().max()), 1e-6), n


@pytest.mark.parametrize("recompute", [True, False], ids=["apply_recomputes", "apply_reads_g5"])
@pytest.mark.parametrize("training", [True, False], ids=["train", "eval"])
@pytest.mark.parametrize("cfg", [(3, 300, 70, 45, True), (2, 1024, 64, 170, True), (4, 257, 96, 100, False), (2, 256, 32, 64, True)],
                         ids=["c45", "c170", "c100_fp", "c64"])
def test_vector_tail_fused_equals_vectorbn_v2s_then_pool(cfg, training, recompute, hip_device, monkeypatch):
    """_ops.GlobalMaxMeanPoolBNV (csrc/vtail.hip: conv5's VectorBN + gate, svfuse's Vector2Scalar and the [max | mean] pooling of its
    half in one pass over linear2's product each way; sv_layers.py:86-102,111-129,193-194,206-220, sv_dgcnn_cls.py:68-74) against the
    layer-wise chain VBN -> V2S -> GlobalMaxMeanPoolBN on the same inputs: the scalar half bit-identical, the vector half's pooled
    values / statistics / every gradient to 2e-5 of the tensor's largest element (other contraction and summation orders), arg-max equal
    but for near-ties, first index on exact ties.  Both forms of the backward's second pass: recomputing dL/d(VectorBN's output) per
    point (svnet_vtail_bwd_apply_f32, the default) and reading the copy the first pass stores (svnet_vbn_bwd_apply_f32)."""
    from svnet_amd import _ops, config
    monkeypatch.setattr(config, "FUSE_VTAIL_APPLY", recompute)
    B, N, Ca, C, binary = cfg
    g = torch.Generator().manual_seed(33 + C)
    y = torch.randn(B, N, Ca, generator=g) * 2.0
    v = torch.randn(B, N, 3, C, generator=g)
    v[0, 5] = v[0, 77]                                       # ties: the first index takes the max gradient
    v[1, 3, :, ::7] = 0.0                                    # zero vectors: norm backward at 0 is 0
    gate = torch.rand(B, C, generator=g) * 0.8 + 0.1
    Wz = torch.randn(3, C, generator=g)
    scz = (torch.rand(1, 3, generator=g) + 0.5) if binary else None
    w = torch.randn(B, 2 * (Ca + 3 * C), generator=g).to(hip_device)
    res = {}
    for fused in (True, False):
        bn1 = torch.nn.BatchNorm1d(Ca).to(hip_device).train(training)
        bn2 = torch.nn.BatchNorm1d(C).to(hip_device).train(training)
        with torch.no_grad():
            bn1.weight.copy_(torch.linspace(-1.0, 1.5, Ca)); bn1.bias.copy_(torch.linspace(0.3, -0.3, Ca))
            bn2.weight.copy_(torch.linspace(1.5, -0.5, C)); bn2.bias.copy_(torch.linspace(-0.2, 0.4, C))       # both signs of gamma
            for bn in (bn1, bn2):
                bn.running_mean.copy_(torch.linspace(0.8, 1.6, bn.num_features)); bn.running_var.copy_(torch.linspace(0.5, 2.0, bn.num_features))
        yd, vd, gd = (t.to(hip_device).requires_grad_(True) for t in (y, v, gate))
        Wd = Wz.to(hip_device).requires_grad_(True)
        sd = None if scz is None else scz.to(hip_device).requires_grad_(True)
        n1, n2 = (bn1.num_batches_tracked, bn2.num_batches_tracked) if training else (None, None)
        with tapped() as tap:
            if fused:
                out = _ops.GlobalMaxMeanPoolBNV.apply(yd, vd, gd, bn1.weight, bn1.bias, bn1.running_mean, bn1.running_var, bn2.weight, bn2.bias,
                                                      bn2.running_mean, bn2.running_var, Wd, sd, training, 1, 0.2, n1, n2, bn1.eps, bn1.momentum)
            else:
                v5 = _ops.VBN.apply(vd, bn2.weight, bn2.bias, bn2.running_mean, bn2.running_var, gd, N, training, n2, bn2.eps, bn2.momentum)
                sv, _ = _ops.V2S.apply(v5, Wd, sd, training)
                out = _ops.GlobalMaxMeanPoolBN.apply(yd, sv, bn1.weight, bn1.bias, bn1.running_mean, bn1.running_var, training, 1, 0.2, n1,
                                                     bn1.eps, bn1.momentum)
        (out * w).sum().backward()
        torch.cuda.synchronize()
        res[fused] = dict(out=out.detach().cpu(), dy=yd.grad.cpu(), dv=vd.grad.cpu(), dgate=gd.grad.cpu(), dWz=Wd.grad.cpu(),
                          dscz=None if sd is None else sd.grad.cpu(), dg1=bn1.weight.grad.cpu(), db1=bn1.bias.grad.cpu(),
                          dg2=bn2.weight.grad.cpu(), db2=bn2.bias.grad.cpu(), rm2=bn2.running_mean.cpu().clone(), rv2=bn2.running_var.cpu().clone(),
                          nbt2=int(bn2.num_batches_tracked), arg=tap["pools"][-1].cpu())
    Ct = Ca + 3 * C
    assert torch.equal(res[True]["out"][:, :Ca], res[False]["out"][:, :Ca])                  # max a: the same kernel
    # (max b: the same expressions, but the two kernels' fused multiply-adds are contracted differently: values to rounding, and the
    #  arg-max may differ only where two points' values are that close)
    same = res[True]["arg"] == res[False]["arg"]
    assert float(same.float().mean()) > 0.999
    assert bool(same[0, Ca + 5 * 3 % (3 * C)]) and int(res[True]["arg"][0, Ca:].min()) >= 0
    assert res[True]["nbt2"] == res[False]["nbt2"] == (1 if training else 0)
    for n in ("out", "dy", "dv", "dgate", "dWz", "dscz", "dg1", "db1", "dg2", "db2", "rm2", "rv2"):
        a_, b_ = res[True][n], res[False][n]
        if b_ is None:
            assert a_ is None
            continue
        assert torch.isfinite(a_).all(), n
        assert float((a_ - b_).abs().max()) <= 2e-5 * max(float(b_.abs().max()), 1e-6), (n, float((a_ - b_).abs().max()), float(b_.abs().max()))


# ----------------------------------------------------------------------------- CatSink with a level that is not fused

_SINK_LEVELS = [((32, 10), (32, 10)), ((32, 10), (32, 10)), ((32, 10), (64, 21)), ((64, 21), (128, 42))]


@pytest.mark.parametrize("layerwise_level", [None, 0, 1], ids=["all_fused", "first_layerwise", "second_layerwise"])
def test_cat_sink_with_a_layerwise_level_falls_back_to_cat(layerwise_level, hip_device):
    """A pyramid of four edge levels (widths 32/10, 32/10, 64/21, 128/42) under one CatSink, ONE of them forced onto the layer-wise
    path (a non-default BatchNorm momentum: SVBlock._default_bn).  Second level layer-wise: the third asks for slot 1 with other widths,
    the sink is abandoned, and the fourth used to crash in CatSink.slot() (len() of None).  First level layer-wise: the second level's
    kernel lands in slot 0 (same widths) before the third abandons the sink; result() must return torch.cat.  Every way the
    concatenation and every gradient equal those of the same pyramid run WITHOUT a sink (identical kernels, torch.cat + autograd's adds)."""
    from svnet_amd import _ops
    from svnet_amd.models.utils.sv_util import get_graph_feature_sv, svpool
    k, B, N = 6, 2, 96
    outs = {}
    for use_sink in (True, False):
        blocks = []
        for i, (cin, cout) in enumerate(_SINK_LEVELS):
            blk, _, s, v, _ = _make((cin, cout, B, N, k), hip_device, True, "sink_lvl%d" % i)
            if i == layerwise_level:
                blk.bn1.momentum = 0.2                      # running statistics only: outputs and gradients are unaffected
            if i == 0:
                leaves = (s.to(hip_device).requires_grad_(True), v.to(hip_device).requires_grad_(True))
            blocks.append(blk)
        level, pyramid = leaves, []
        sink = _ops.CatSink([c[1][0] for c in _SINK_LEVELS], [c[1][1] for c in _SINK_LEVELS])
        with (sink if use_sink else contextlib.nullcontext()):
            for blk in blocks:
                level = svpool(blk(get_graph_feature_sv(level, k=k)))
                pyramid.append(level)
        if use_sink:
            s_cat, v_cat = sink.result(pyramid)
            in_place = sink.filled is not None and len(sink.filled) == len(blocks)
            assert in_place == (layerwise_level is None)
        else:
            s_cat, v_cat = torch.cat([x[0] for x in pyramid], -1), torch.cat([x[1] for x in pyramid], -1)
        rs = C.t("sink/rs", tuple(s_cat.shape)).to(hip_device)
        rv = C.t("sink/rv", tuple(v_cat.shape)).to(hip_device)
        ((s_cat * rs).sum() + (v_cat * rv).sum()).backward()
        g = {"out0": s_cat.detach().cpu().numpy(), "out1": v_cat.detach().cpu().numpy(),
             "dx0": leaves[0].grad.cpu().numpy(), "dx1": leaves[1].grad.cpu().numpy()}
        for i, blk in enumerate(blocks):
            for n, p in blk.named_parameters():
                g["d:%d.%s" % (i, n)] = p.grad.cpu().numpy()
        outs[use_sink] = g
    assert np.array_equal(outs[True]["out0"], outs[False]["out0"]) and np.array_equal(outs[True]["out1"], outs[False]["out1"])
    compare_case(outs[True], outs[False], 1e-4, "pyramid with a layer-wise level: sink vs no sink")


# ----------------------------------------------------------------------------- the next level's k-NN table from the apply pass

@pytest.mark.parametrize("xyz", [False, True], ids=["edge_level", "first_level"])
@pytest.mark.parametrize("cfg", [(2, 1024, 32, 10, 20), (2, 1024, 64, 21, 20), (1, 1024, 128, 42, 20), (1, 2048, 32, 16, 40), (1, 2048, 64, 24, 40),
                                 (3, 96, 32, 10, 8), (2, 160, 64, 21, 16), (2, 512, 8, 1, 5)],
                         ids=["conv2", "conv3", "conv4w", "pseg2", "pseg3", "n96", "n160", "narrow"])
def test_apply_pass_that_prepares_the_knn_table_is_bit_identical(cfg, xyz, hip_device):
    """svnet_{edgeblock,xyzblock}_apply_knn_f32 + svnet_knn_from_table_f32 (csrc/apply_knn.h) against svnet_*_apply_f32 +
    svnet_knn_sv_f32 on the same inputs: pooled outputs, their concatenation slices, the WHOLE k-NN workspace (channel-major table, zero
    rows, ||x||^2 by ATen's contiguous-row recipe) and the neighbour lists, bit for bit (sv_util.py:19-25,100-101)."""
    from svnet_amd import _lib
    from svnet_amd._ops import _p, _stream, call
    B, N, Os, Ov, k = cfg
    P, C = B * N, Os + 3 * Ov
    L = _lib.lib()
    assert L.svnet_knn_table_fusable(B, N, C) == 1
    g = torch.Generator().manual_seed(5 + Os + N)
    f32 = dict(dtype=torch.float32, device=hip_device)
    if xyz:
        hi = torch.randn(P, Os, generator=g).to(hip_device)
        lo = (hi.cpu() - torch.rand(P, Os, generator=g)).to(hip_device)
    else:
        hi = torch.randint(-200, 200, (P, Os), generator=g, dtype=torch.int32).to(hip_device)
        lo = (hi.cpu() - torch.randint(0, 50, (P, Os), generator=g, dtype=torch.int32)).to(hip_device)
    mv, mvn = torch.randn(P, 3, Ov, generator=g).to(hip_device), torch.randn(P, 3, Ov, generator=g).to(hip_device)
    coef = torch.randn(4 * Os + 4 * Ov, generator=g)
    coef[:Os] *= 0.05                                          # A1: both signs (max_k n or min_k n is the pooled one)
    coef = coef.to(hip_device)
    gate = torch.rand(B, Ov, generator=g).to(hip_device)
    name = "xyzblock" if xyz else "edgeblock"
    nbytes = L.svnet_knn_workspace_bytes(B, N, C)
    out = {}
    for fused in (False, True):
        s_out, v_out = torch.empty(B, N, Os, **f32), torch.empty(B, N, 3, Ov, **f32)
        s_cat, v_cat = torch.zeros(P, Os + 7, **f32), torch.zeros(P, 3, Ov + 5, **f32)
        ws = torch.zeros(nbytes, dtype=torch.uint8, device=hip_device)
        idx = torch.empty(B, N, k, dtype=torch.int64, device=hip_device)
        args = (_p(hi), _p(lo), _p(mv), _p(mvn), _p(coef), _p(gate), P, N, Os, Ov, 0.2, _p(s_out), _p(v_out), _p(s_cat[:, 3:]), Os + 7,
                _p(v_cat[:, :, 2:]), Ov + 5)
        if fused:
            call("svnet_%s_apply_knn_f32" % name, *args, _p(ws), nbytes, _stream())
            call("svnet_knn_from_table_f32", _p(ws), nbytes, B, N, C, k, _p(idx), _stream())
        else:
            call("svnet_%s_apply_f32" % name, *args, _stream())
            call("svnet_knn_sv_f32", _p(s_out), Os, _p(v_out), 3 * Ov, B, N, k, _p(idx), _p(ws), nbytes, _stream())
        torch.cuda.synchronize()
        out[fused] = (s_out.cpu(), v_out.cpu(), s_cat.cpu(), v_cat.cpu(), ws.cpu(), idx.cpu())
    C8 = (C + 7) // 8 * 8
    wa, wb = (out[f][4][:4 * (P * C8 + P)].view(torch.float32) for f in (True, False))
    bad_t, bad_x = int((wa[:P * C8] != wb[:P * C8]).sum()), int((wa[P * C8:] != wb[P * C8:]).sum())
    assert bad_t == 0 and bad_x == 0, "table entries that differ: %d of %d, ||x||^2: %d of %d" % (bad_t, P * C8, bad_x, P)
    for a, b_, what in zip(out[True], out[False], ("s_out", "v_out", "s_cat", "v_cat", "knn workspace", "idx")):
        assert torch.equal(a, b_), what
    assert torch.equal(out[True][2][:, 3:3 + Os], out[True][0].view(P, Os)) and float(out[True][2][:, :3].abs().max()) == 0.0
    assert int(out[True][5].min()) >= 0 and int(out[True][5].max()) < N


def test_apply_knn_refuses_what_it_cannot_tile(hip_device):
    from svnet_amd import _lib
    from svnet_amd._ops import _p, _stream
    L = _lib.lib()
    assert L.svnet_knn_table_fusable(2, 100, 62) == 0 and L.svnet_knn_table_fusable(2, 1024, 4) == 0 and L.svnet_knn_table_fusable(2, 1024, 62) == 1
    t = torch.zeros(64, device=hip_device)
    rc = L.svnet_edgeblock_apply_knn_f32(_p(t), _p(t), _p(t), _p(t), _p(t), _p(t), 200, 100, 32, 10, 0.2, _p(t), _p(t), None, 0, None, 0, _p(t), 256,
                                          _stream())
    assert rc == -2 and b"not supported" in L.svnet_last_error()


@pytest.mark.parametrize("model_name,B,N,k", [("sv_dgcnn_cls", 2, 128, 8), ("sv_dgcnn_pseg", 2, 64, 6)])
def test_models_give_the_same_bits_with_and_without_the_table_ahead(model_name, B, N, k, hip_device, monkeypatch):
    """The callers with config.KNN_TABLE_AHEAD on and off: logits and loss identical, every parameter gradient to the run-to-run noise
    of its atomics (the prepared table is the table), and the switch does route the k-NN: three calls from a prepared table per step when on, none when off."""
    from svnet_amd import _ops, config
    from oracle import params as oparams
    from tests.test_hip_train_parity import build_model, hip_step
    x, l, y = C.model_inputs("ahead_" + model_name, model_name, B, N)
    state = oparams.synthetic_params(model_name, binary=True, seed=C.SEED)
    taken = []
    real_take = _ops.knn_table_ahead.take
    monkeypatch.setattr(_ops.knn_table_ahead, "take", staticmethod(lambda s_, v_: taken.append(real_take(s_, v_)) or taken[-1]))
    res = {}
    for on in (True, False):
        monkeypatch.setattr(config, "KNN_TABLE_AHEAD", on)
        del taken[:]
        m = build_model(model_name, True, k, hip_device, state).train()
        logits, loss, got, _ = hip_step(m, x, l, y, hip_device)
        torch.cuda.synchronize()
        assert [t is not None for t in taken] == [on] * 3, taken
        res[on] = (logits, loss, got)
    assert np.array_equal(res[True][0], res[False][0]) and res[True][1] == res[False][1]
    # (gradients: the backward never sees the table - the same kernels on the same bits, but their float atomics arrive in another order
    #  from run to run, and cancelling sums such as a binarized layer's scale gradient show that at 1e-2 of their own size)
    gmax = max(float(np.abs(g_).max()) for g_ in res[False][2].values())
    for n, g_ in res[True][2].items():
        ref = res[False][2][n]
        assert float(np.abs(g_ - ref).max()) <= 1e-4 * float(np.abs(ref).max()) + 1e-5 * gmax, n


# ----------------------------------------------------------------------------- coefficients + gate MLP + apply (+ table) as one launch

@pytest.mark.parametrize("table", [True, False], ids=["with_table", "no_table"])
@pytest.mark.parametrize("training", [True, False], ids=["train", "eval"])
@pytest.mark.parametrize("xyz", [False, True], ids=["edge_level", "first_level"])
@pytest.mark.parametrize("cfg", [(2, 1024, 32, 10, 64), (1, 1024, 128, 42, 128), (2, 2048, 64, 24, 64), (3, 96, 32, 10, 6)],
                         ids=["conv2", "conv4", "pseg3", "n96"])
def test_block_tail_is_bit_identical_to_its_three_launches(cfg, xyz, training, table, hip_device):
    """svnet_{edgeblock,xyzblock}_tail_f32 against svnet_*_coeffs_f32 (gate MLP inside) + svnet_*_apply[_knn]_f32: coefficients, running
    statistics, counters, the gate MLP's three outputs, the pooled features, their concatenation slices and the k-NN workspace, bit for
    bit, in train and eval mode (sv_layers.py:172-196: bn1 / bn2 / gate of an SVBlock behind the fused edge pass)."""
    from svnet_amd import _lib
    from svnet_amd._ops import _p, _stream, call, BN_EPS, BN_MOMENTUM, RED_SLICES
    B, N, Os, Ov, Cin = cfg
    P, E, k, H = B * N, B * N * 20, 20, max(Ov // 2, 1)
    L = _lib.lib()
    assert L.svnet_block_tail_supported(P, N, Os, Ov, int(table)) == 1
    g = torch.Generator().manual_seed(11 + Os + N)
    dev = hip_device
    f32 = dict(dtype=torch.float32, device=dev)
    if xyz:
        hi = torch.randn(P, Os, generator=g).to(dev)
        lo = (hi.cpu() - torch.rand(P, Os, generator=g)).to(dev)
        mean, var = torch.randn(Os, generator=g).double(), torch.rand(Os, generator=g).double() + 0.1
        per = torch.stack([mean * E / RED_SLICES, (var + mean * mean) * E / RED_SLICES])           # [2, Os]
        stat1 = (per.unsqueeze(0) * (1.0 + 0.01 * torch.randn(RED_SLICES, 2, Os, generator=g).double())).reshape(-1).to(dev)
        sc1 = None
    else:
        hi = torch.randint(-200, 200, (P, Os), generator=g, dtype=torch.int32).to(dev)
        lo = (hi.cpu() - torch.randint(0, 50, (P, Os), generator=g, dtype=torch.int32)).to(dev)
        mean, var = torch.randn(Os, generator=g).double() * 3, torch.rand(Os, generator=g).double() * 30 + 1
        per = torch.stack([mean * E / RED_SLICES, (var + mean * mean) * E / RED_SLICES])
        stat1 = (per.unsqueeze(0) * (1.0 + 0.01 * torch.rand(RED_SLICES, 2, Os, generator=g).double())).round().long().reshape(-1).to(dev)
        sc1 = (torch.rand(Os, generator=g) * 0.1 + 0.01).to(dev)
    mv_m, mv_v = torch.rand(Ov, generator=g).double() + 0.5, torch.rand(Ov, generator=g).double() * 0.2 + 0.01
    per_v = torch.stack([mv_m * E / RED_SLICES, (mv_v + mv_m * mv_m) * E / RED_SLICES])
    stat_v = (per_v.unsqueeze(0) * (1.0 + 0.01 * torch.rand(RED_SLICES, 2, Ov, generator=g).double())).reshape(-1).to(dev)
    mv, mvn = torch.randn(P, 3, Ov, generator=g).to(dev), torch.randn(P, 3, Ov, generator=g).to(dev)
    g1, b1 = (torch.randn(Os, generator=g)).to(dev), torch.randn(Os, generator=g).to(dev)          # both signs of gamma
    g2, b2 = (torch.rand(Ov, generator=g) + 0.5).to(dev), torch.randn(Ov, generator=g).to(dev)
    gate_sum = (torch.randn(B, Cin, generator=g).double() * N * k).to(dev)
    W0, W2 = (torch.randn(H, Cin, generator=g) * 0.3).to(dev), torch.randn(Ov, H, generator=g).to(dev)
    nbytes = L.svnet_knn_workspace_bytes(B, N, Os + 3 * Ov)
    name = "xyzblock" if xyz else "edgeblock"
    out = {}
    for fused in (False, True):
        rm1, rv1 = torch.linspace(-1, 1, Os).to(dev), torch.linspace(0.5, 2, Os).to(dev)
        rm2, rv2 = torch.linspace(0.2, 1, Ov).to(dev), torch.linspace(0.5, 2, Ov).to(dev)
        nbt1, nbt2 = torch.full((), 3, dtype=torch.int64, device=dev), torch.full((), 5, dtype=torch.int64, device=dev)
        coef = torch.empty(4 * Os + 4 * Ov, **f32)
        h, gate, gin = torch.empty(B, H, **f32), torch.empty(B, Ov, **f32), torch.empty(B, Cin, **f32)
        s_out, v_out = torch.empty(B, N, Os, **f32), torch.empty(B, N, 3, Ov, **f32)
        s_cat, v_cat = torch.zeros(P, Os + 7, **f32), torch.zeros(P, 3, Ov + 5, **f32)
        ws = torch.zeros(nbytes, dtype=torch.uint8, device=dev)
        job = _lib.GateFwdJob(None, _p(gate_sum), _p(gin), 1.0 / float(N * k), _p(W0), _p(W2), B, Cin, H, Ov, _p(h), _p(gate))
        st1, stv = (stat1, stat_v) if training else (None, None)
        if fused:
            d = _lib.BlockTailDesc()
            d.stat1, d.stat_v, d.E, d.Os, d.Ov, d.scale1 = _p(st1), _p(stv), E, Os, Ov, _p(sc1)
            d.gamma1, d.beta1, d.running_mean1, d.running_var1 = _p(g1), _p(b1), _p(rm1), _p(rv1)
            d.gamma2, d.beta2, d.running_mean2, d.running_var2 = _p(g2), _p(b2), _p(rm2), _p(rv2)
            d.training, d.eps, d.momentum = int(training), BN_EPS, BN_MOMENTUM
            d.coef, d.num_batches_tracked1, d.num_batches_tracked2 = _p(coef), _p(nbt1), _p(nbt2)
            d.gate = job
            d.hi, d.lo, d.mv, d.mvn, d.P, d.N, d.slope = _p(hi), _p(lo), _p(mv), _p(mvn), P, N, 0.2
            d.s_out, d.v_out, d.s_cat, d.s_ld, d.v_cat, d.v_ld = _p(s_out), _p(v_out), _p(s_cat[:, 3:]), Os + 7, _p(v_cat[:, :, 2:]), Ov + 5
            if table:
                d.knn_workspace, d.knn_workspace_bytes = _p(ws), nbytes
            call("svnet_%s_tail_f32" % name, ctypes.byref(d), _stream())
        else:
            stats = (_p(st1), _p(stv), E, Os, Ov) + ((_p(sc1),) if not xyz else ())
            call("svnet_%s_coeffs_f32" % name, *stats, _p(g1), _p(b1), _p(rm1), _p(rv1), _p(g2), _p(b2), _p(rm2), _p(rv2), int(training), BN_EPS,
                 BN_MOMENTUM, _p(coef), _p(nbt1), _p(nbt2), ctypes.byref(job), _stream())
            args = (_p(hi), _p(lo), _p(mv), _p(mvn), _p(coef), _p(gate), P, N, Os, Ov, 0.2, _p(s_out), _p(v_out), _p(s_cat[:, 3:]), Os + 7,
                    _p(v_cat[:, :, 2:]), Ov + 5)
            if table:
                call("svnet_%s_apply_knn_f32" % name, *args, _p(ws), nbytes, _stream())
            else:
                call("svnet_%s_apply_f32" % name, *args, _stream())
        torch.cuda.synchronize()
        out[fused] = dict(coef=coef, h=h, gate=gate, gin=gin, s_out=s_out, v_out=v_out, s_cat=s_cat, v_cat=v_cat, ws=ws, rm1=rm1, rv1=rv1, rm2=rm2,
                          rv2=rv2, nbt1=nbt1, nbt2=nbt2)
        out[fused] = {n: t.cpu() for n, t in out[fused].items()}
    for n, t in out[True].items():
        assert torch.equal(t, out[False][n]), n
    assert int(out[True]["nbt1"]) == (4 if training else 3) and int(out[True]["nbt2"]) == (6 if training else 5)
    assert torch.isfinite(out[True]["s_out"]).all() and torch.isfinite(out[True]["v_out"]).all()
    if training:
        assert not torch.equal(out[True]["rm1"], torch.linspace(-1, 1, Os))            # the running statistics moved - once


@pytest.mark.parametrize("model_name,B,N,k", [("sv_dgcnn_cls", 2, 128, 8), ("sv_dgcnn_pseg", 2, 64, 6)])
def test_models_give_the_same_bits_with_and_without_the_block_tail(model_name, B, N, k, hip_device, monkeypatch):
    """The callers with config.FUSE_BLOCK_TAIL on and off: logits, loss and BatchNorm buffers identical, gradients to their atomics' noise."""
    from svnet_amd import config
    from oracle import params as oparams
    from tests.test_hip_train_parity import build_model, hip_step
    x, l, y = C.model_inputs("tail_" + model_name, model_name, B, N)
    state = oparams.synthetic_params(model_name, binary=True, seed=C.SEED)
    res = {}
    for on in (True, False):
        monkeypatch.setattr(config, "FUSE_BLOCK_TAIL", on)
        m = build_model(model_name, True, k, hip_device, state).train()
        logits, loss, got, _ = hip_step(m, x, l, y, hip_device)
        torch.cuda.synchronize()
        res[on] = (logits, loss, got, {n: t.cpu() for n, t in m.state_dict().items() if "running_" in n or "num_batches" in n})
        with torch.no_grad():
            res[on] += (m.eval()(*((x.to(hip_device),) if l is None else (x.to(hip_device), l.to(hip_device)))).cpu(),)
    assert np.array_equal(res[True][0], res[False][0]) and res[True][1] == res[False][1] and torch.equal(res[True][4], res[False][4])
    for n, t in res[True][3].items():
        assert torch.equal(t, res[False][3][n]), n
    gmax = max(float(np.abs(g_).max()) for g_ in res[False][2].values())
    for n, g_ in res[True][2].items():
        ref = res[False][2][n]
        assert float(np.abs(g_ - ref).max()) <= 1e-4 * float(np.abs(ref).max()) + 1e-5 * gmax, n


# ----------------------------------------------------------------------------- conv5's concatenation with the gate's mean inside

@pytest.mark.parametrize("cfg", [(4, 256, 256, 83, True), (2, 1024, 256, 83, True), (3, 64, 200, 40, False), (2, 48, 256, 83, True)],
                         ids=["n256", "n1024", "fp_w200", "n48_fallback"])
def test_cat_with_the_gate_mean_inside_equals_cat_pool_gate(cfg, hip_device):
    """_ops.V2SCat with the block's gate MLP (config.FUSE_CAT_MEAN; svnet_v2s_cat_sum_fwd_f32: per-cloud fp64 column sums of s from the
    concatenation kernel's own copy, the MLP started from them) against V2SCat + the pooling pass + _ops.GateMLP (sv_layers.py:179-188):
    the concatenation bit for bit, the gate to 1e-6, every gradient to 2e-5 of its largest element.  N = 48 is outside the summing
    kernel's shapes (whole 32-row blocks per cloud): the op then pools by its own pass."""
    from svnet_amd import _ops
    B, N, Cs, Cv, binary = cfg
    H, Ov = 85, 170
    g = torch.Generator().manual_seed(3 + N + Cs)
    s0, v0 = torch.randn(B, N, Cs, generator=g), torch.randn(B, N, 3, Cv, generator=g)
    Wz0 = torch.randn(3, Cv, generator=g)
    scz0 = (torch.rand(1, 3, generator=g) + 0.5) if binary else None
    W00, W20 = torch.randn(H, Cs, generator=g) * 0.1, torch.randn(Ov, H, generator=g) * 0.3
    wc, wg = torch.randn(B, N, Cs + 3 * Cv, generator=g).to(hip_device), torch.randn(B, Ov, generator=g).to(hip_device)
    res = {}
    for fused in (True, False):
        s, v, Wz, W0, W2 = (t.clone().to(hip_device).requires_grad_(True) for t in (s0, v0, Wz0, W00, W20))
        scz = None if scz0 is None else scz0.clone().to(hip_device).requires_grad_(True)
        if fused:
            cat, gate = _ops.V2SCat.apply(s, v, Wz, scz, True, B, W0, W2)
        else:
            cat, s_mean = _ops.V2SCat.apply(s, v, Wz, scz, True, B)
            gate = _ops.GateMLP.apply(s_mean, W0, W2)
        ((cat * wc).sum() + (gate * wg).sum()).backward()
        torch.cuda.synchronize()
        res[fused] = dict(cat=cat.detach().cpu(), gate=gate.detach().cpu(), ds=s.grad.cpu(), dv=v.grad.cpu(), dWz=Wz.grad.cpu(), dW0=W0.grad.cpu(),
                          dW2=W2.grad.cpu(), dscz=None if scz is None else scz.grad.cpu())
    assert torch.equal(res[True]["cat"], res[False]["cat"])
    assert float((res[True]["gate"] - res[False]["gate"]).abs().max()) <= 1e-6
    ref_gate = torch.sigmoid(torch.relu(s0.mean(1) @ W00.t()) @ W20.t())
    assert float((res[True]["gate"] - ref_gate).abs().max()) <= 1e-5
    for n in ("ds", "dv", "dWz", "dW0", "dW2", "dscz"):
        a_, b_ = res[True][n], res[False][n]
        if b_ is None:
            assert a_ is None
            continue
        assert float((a_ - b_).abs().max()) <= 2e-5 * max(float(b_.abs().max()), 1e-6), n
