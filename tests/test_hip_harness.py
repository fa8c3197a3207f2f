"""GPU tests (-m gpu) of the pieces around the kernels that round 2's review found untested: the augmentation, the step objects'
scratch arenas under graph replay, the packed-weight cache's lifetime, optimizer / scheduler checkpoints in torch.optim's
layout, and op-level parity of the three ops only the SV-PointNet part-segmentation caller uses."""
import argparse
import contextlib
import gc
import io
import math

import numpy as np
import pytest
import torch

from oracle import params as oparams
from oracle import sv_ref
from tests.common import compare_case
from tests.golden import cases as C
from tests.golden import harness as H

pytestmark = pytest.mark.gpu


def _dgcnn(dev, binary, k=8, seed=0):
    import svnet_amd.models as M
    torch.manual_seed(seed)
    with contextlib.redirect_stdout(io.StringIO()):
        return M.SV_DGCNN_CLS(argparse.Namespace(k=k, binary=binary), 40).to(dev)


def _batch(dev, B, N, t=0):
    from svnet_amd import synth
    return (torch.from_numpy(synth.cloud_batch(1234, t, 0, B, N)).to(dev), torch.from_numpy(synth.class_labels(1234, t, 0, B)).to(dev))


# ----------------------------------------------------------------------------- rotate_clouds (main_cls_dgcnn.py:168-178)

@pytest.mark.parametrize("mode", ["z", "so3"])
def test_rotate_clouds_applies_one_proper_rotation_per_cloud(mode, hip_device):
    """x' = R_b x with R_b^T R_b = I, det R_b = +1, a different R_b per cloud; 'z' leaves the z row alone and mixes x, y by one
    angle; the rotation acts on the coordinate axis of the [B,3,N] layout (R x, not x R)."""
    from svnet_amd.train import rotate_clouds
    x, _ = _batch(hip_device, 6, 256)
    g = torch.Generator(device=hip_device).manual_seed(5)
    xr = rotate_clouds(x, mode, generator=g)
    assert xr.shape == x.shape and xr.data_ptr() != x.data_ptr()
    X, Xr = x.double().cpu(), xr.double().cpu()
    R = Xr @ torch.linalg.pinv(X)                                       # [B,3,3]: the map that was applied, recovered from the points
    eye = torch.eye(3, dtype=torch.float64).expand(6, 3, 3)
    assert float((R.transpose(1, 2) @ R - eye).abs().max()) < 1e-5
    assert float((torch.linalg.det(R) - 1.0).abs().max()) < 1e-5
    assert float((R @ X - Xr).abs().max()) < 1e-5                        # every point of a cloud moved by the SAME matrix
    assert float((R[0] - R[1]).abs().max()) > 1e-2                       # ... and clouds by different ones
    if mode == "z":
        assert torch.equal(xr[:, 2], x[:, 2])
        assert float((R[:, 2, :2].abs().max() + R[:, :2, 2].abs().max())) < 1e-6
        ang = torch.atan2(R[:, 1, 0], R[:, 0, 0])
        assert float((ang.max() - ang.min())) > 0.5                      # angles spread over the circle
    assert torch.equal(rotate_clouds(x, "none"), x)                      # any other mode: untouched (the reference's `else` branch)


def test_rotated_clouds_leave_the_fp_logits_unchanged(hip_device):
    """What the augmentation is for (README: z / SO(3) protocols): the model is rotation-equivariant, so eval logits of the fp model
    on rotate_clouds(x, 'so3') equal those on x (1e-3 of the logit range)."""
    from svnet_amd.train import rotate_clouds
    P = oparams.synthetic_params("sv_dgcnn_cls", binary=False, seed=C.SEED)
    m = _dgcnn(hip_device, False, k=10)
    m.load_state_dict(P)
    m.eval()
    x, _ = _batch(hip_device, 4, 256)
    g = torch.Generator(device=hip_device).manual_seed(9)
    with torch.no_grad():
        a, b = m(x), m(rotate_clouds(x, "so3", generator=g).contiguous())
    assert float((a - b).abs().max() / a.abs().max()) < 1e-3


# ----------------------------------------------------------------------------- step objects

def test_a_captured_step_survives_a_larger_eager_step(hip_device):
    """the zero-filled scratch of a step used to be ONE process-wide buffer that a later, larger step replaced -
    a captured graph then filled and accumulated into freed memory.  Each step object now owns its arena and a captured one is
    pinned: capture at B=2, run a larger eager step (B=6) of the same model and an eval step of another, replay - the loss must be
    bit-identical to the replay before and the gradient bucket equal to float-atomic order."""
    from svnet_amd.train import ForwardStep, TrainStep
    model = _dgcnn(hip_device, True).train()
    x, y = _batch(hip_device, 2, 256)
    small = TrainStep(model, (x,), y).capture()
    l0 = float(small.run(all_reduce=False))
    b0 = small.bucket.flat.clone()
    buf = small._arena.buf
    assert small._arena.pinned and buf is not None
    xb, yb = _batch(hip_device, 6, 512, t=1)
    big = TrainStep(model, (xb,), yb)
    for _ in range(3):                                   # (the second eager step is the one that sizes its arena)
        big.fwd_bwd()
    other = _dgcnn(hip_device, True, seed=1)
    ForwardStep(other, (xb,)).run()
    junk = [torch.full((1 << 20,), float("nan"), device=hip_device) for _ in range(8)]      # whatever was freed gets reused
    torch.cuda.synchronize()
    assert small._arena.buf is buf and big._arena.buf is not buf and big._arena.buf.numel() > 0
    l1 = float(small.run(all_reduce=False))
    torch.cuda.synchronize()
    assert l1 == l0, (l0, l1)
    assert float((small.bucket.flat - b0).abs().max()) <= 5e-5 * float(b0.abs().max())
    del junk


def test_forward_step_restores_the_training_mode(hip_device):
    """ForwardStep.forward() left the shared model in eval(), so an eager TrainStep after it trained with running
    statistics and bare sign() (zero STE gradients)."""
    from svnet_amd.train import ForwardStep, TrainStep
    model = _dgcnn(hip_device, True).train()
    x, y = _batch(hip_device, 2, 128)
    step = TrainStep(model, (x,), y)
    step.fwd_bwd()
    ref = step.bucket.flat.clone()
    ForwardStep(model, (x,)).run()
    assert model.training
    step.fwd_bwd()
    assert float(step.bucket.flat.abs().max()) > 0 and float((step.bucket.flat - ref).abs().max()) <= 5e-5 * float(ref.abs().max())
    model.eval()
    ForwardStep(model, (x,)).run()
    assert not model.training


def test_packed_weight_cache_lets_dead_models_go(hip_device):
    """the cache's re-pack closures held the parameters strongly - every model ever built stayed alive and was re-packed
    after every optimizer step.  They hold weak references now: once a model is dropped its entries disappear."""
    import weakref
    from svnet_amd import _ops
    from svnet_amd.train import TrainStep
    model = _dgcnn(hip_device, True).train()
    x, y = _batch(hip_device, 2, 128)
    TrainStep(model, (x,), y).fwd_bwd()
    assert len(_ops.PLANES.entries) > 0
    probe = weakref.ref(model.conv3.linear1.weight)
    mine = [k for k, e in _ops.PLANES.entries.items() if any(r() is not None and any(r() is p for p in model.parameters()) for r in e[0])]
    assert mine
    del model
    gc.collect()
    assert probe() is None, "a parameter of the deleted model is still referenced"
    _ops.PLANES._stale()
    assert not any(k in _ops.PLANES.entries for k in mine)


@pytest.mark.parametrize("kind", ["adam", "sgd"])
def test_flat_optimizer_state_round_trips_through_torch_optim(kind, hip_device):
    """utils.py:141-171 stores optimizer.state_dict() / scheduler.state_dict() in the checkpoint and main_cls_dgcnn.py:147-148
    resumes with load_state_dict: the flat optimizers speak torch.optim's layout in both directions.  3 flat steps -> state into
    torch.optim -> 2 steps on both -> same weights; then torch's state back into a FRESH flat optimizer -> 2 more steps on both."""
    from svnet_amd.dist import GradBucket
    from svnet_amd.train import CosineLR, FlatAdam, FlatParams, FlatSGD
    torch.manual_seed(3)
    net = torch.nn.Sequential(torch.nn.Linear(37, 53), torch.nn.Linear(53, 11)).to(hip_device)
    ref = torch.nn.Sequential(torch.nn.Linear(37, 53), torch.nn.Linear(53, 11)).to(hip_device)
    fp, bucket = FlatParams(net), GradBucket(net.parameters())

    def flat(lr):
        return FlatAdam(fp, bucket, lr=lr, weight_decay=1e-4) if kind == "adam" else FlatSGD(fp, bucket, lr=lr, momentum=0.9, weight_decay=1e-4)

    def torch_opt():
        return (torch.optim.Adam(ref.parameters(), lr=0.5, weight_decay=0.0) if kind == "adam"
                else torch.optim.SGD(ref.parameters(), lr=0.5, momentum=0.1, weight_decay=0.0))     # (hyper-parameters come from the state)

    def steps(n, opt, topt, seed):
        g = torch.Generator(device=hip_device).manual_seed(seed)
        for _ in range(n):
            xb = torch.randn(16, 37, device=hip_device, generator=g)
            if opt is not None:
                bucket.zero()
                net(xb).pow(2).mean().backward()
                opt.step()
            if topt is not None:
                topt.zero_grad()
                ref(xb).pow(2).mean().backward()
                topt.step()

    def same():
        for (n, a), (_, b) in zip(net.named_parameters(), ref.named_parameters()):
            assert torch.allclose(a, b, rtol=1e-5, atol=1e-6), (kind, n, float((a - b).abs().max()))

    opt = flat(1e-3 if kind == "adam" else 0.05)
    sched = CosineLR(opt, 10)
    steps(3, opt, None, 1)
    sched.step()
    ref.load_state_dict(net.state_dict())
    topt = torch_opt()
    topt.load_state_dict(opt.state_dict())
    assert abs(topt.param_groups[0]["lr"] - opt.lr) < 1e-12
    steps(2, opt, topt, 2)
    same()
    opt2 = flat(123.0)
    opt2.load_state_dict(topt.state_dict())
    sched2 = CosineLR(opt2, 3)
    sched2.load_state_dict(sched.state_dict())
    assert opt2.lr == opt.lr and sched2.epoch == 1 and sched2.T_max == 10
    steps(2, opt2, topt, 3)
    same()
    tsched = torch.optim.lr_scheduler.CosineAnnealingLR(torch_opt(), 10)
    assert set(sched.state_dict()) <= set(tsched.state_dict())


# ----------------------------------------------------------------------------- ops of the SV-PointNet part-segmentation caller

def test_vproject_matches_the_einsum(hip_device):
    """_ops.VProject = einsum('bimj,bijk->bimk') of sv_pointnet_partseg.py:89, flattened: values and both gradients at 1e-4."""
    from svnet_amd import _ops
    B, N, Cc, J = 3, 70, 37, 3
    v, z = C.t("vproj/v", (B, N, 3, Cc)), C.t("vproj/z", (B, N, 3, J))
    r = C.t("vproj/r", (B, N, Cc * J))
    vd, zd = v.to(hip_device).requires_grad_(True), z.to(hip_device).requires_grad_(True)
    out = _ops.VProject.apply(vd, zd)
    (out * r.to(hip_device)).sum().backward()
    vo, zo = v.double().requires_grad_(True), z.double().requires_grad_(True)
    ref = torch.einsum("bimj,bijk->bimk", vo.transpose(-1, -2), zo).reshape(B, N, -1)
    (ref * r.double()).sum().backward()
    compare_case({"out0": out.detach().cpu().numpy(), "dx0": vd.grad.cpu().numpy(), "dx1": zd.grad.cpu().numpy()},
                 {"out0": ref.detach().numpy(), "dx0": vo.grad.numpy(), "dx1": zo.grad.numpy()}, 1e-4, "VProject")


@pytest.mark.parametrize("binary", [False, True], ids=["fp", "bin"])
def test_v2scat_matches_cat_of_vector2scalar(binary, hip_device):
    """_ops.V2SCat = cat[s, Vector2Scalar(v)] (sv_layers.py:187-188) written in place: values, ds, dv and the weight / scale
    gradients against the oracle."""
    from svnet_amd import _ops
    rows, Cs, Cv = (4, 33), 19, 23
    params = H.module_params("Vector2Scalar", (Cv, 3, binary, False), "v2scat")
    s, v = C.sv_pair("v2scat/x", rows, Cs, Cv)
    r = C.t("v2scat/r", rows + (Cs + 3 * Cv,))
    W = params["linear.weight"].to(hip_device).requires_grad_(True)
    sc = params["linear.scale"].to(hip_device).requires_grad_(True) if binary else None
    sd, vd = s.to(hip_device).requires_grad_(True), v.to(hip_device).requires_grad_(True)
    out = _ops.V2SCat.apply(sd, vd, W, sc, True)
    (out * r.to(hip_device)).sum().backward()
    P = {"m." + n: t.clone().requires_grad_(True) for n, t in params.items()}
    so, vo = s.clone().requires_grad_(True), v.clone().requires_grad_(True)
    ref = torch.cat([so, sv_ref.vector2scalar(vo, P, "m", binary=binary, ctx=sv_ref.Ctx(train=True))], dim=-1)
    (ref * r).sum().backward()
    got = {"out0": out.detach().cpu().numpy(), "dx0": sd.grad.cpu().numpy(), "dx1": vd.grad.cpu().numpy(), "d:weight": W.grad.cpu().numpy()}
    want = {"out0": ref.detach().numpy(), "dx0": so.grad.numpy(), "dx1": vo.grad.numpy(), "d:weight": P["m.linear.weight"].grad.numpy()}
    if binary:
        got["d:scale"], want["d:scale"] = sc.grad.cpu().numpy(), P["m.linear.scale"].grad.numpy()
    compare_case(got, want, 1e-4, "V2SCat")


@pytest.mark.parametrize("binary", [False, True], ids=["fp", "bin"])
def test_conv_bn_relu_rows_matches_the_channel_first_oracle(binary, hip_device):
    """[Conv1d, BatchNorm1d, ReLU] of the SV-PointNet part-seg heads on channel-LAST rows (_ConvBNReLU.forward_rows, what the model
    runs) against the reference's channel-first chain (sv_pointnet_partseg.py:38-51) in the oracle: values, input gradient, every
    parameter gradient, train mode."""
    from svnet_amd.models.sv_layers import Conv1d
    from svnet_amd.models.sv_pointnet_partseg import _ConvBNReLU
    B, N, Cin, Cout = 3, 50, 45, 24
    spec = {}
    oparams._conv(spec, "0", Cin, Cout, binary)
    oparams._bn(spec, "1", Cout)
    from svnet_amd import synth
    params = {n: torch.from_numpy(a.copy()) for n, a in synth.synthetic_state(spec, C.SEED + 77).items()}
    with contextlib.redirect_stdout(io.StringIO()):
        blk = _ConvBNReLU(Conv1d(Cin, Cout, binary=binary), torch.nn.BatchNorm1d(Cout), torch.nn.ReLU(inplace=True))
    blk.load_state_dict(params)
    blk = blk.to(hip_device).train()
    rows = C.t("cbr/x", (B, N, Cin))
    r = C.t("cbr/r", (B, N, Cout))
    xd = rows.to(hip_device).requires_grad_(True)
    out = blk.forward_rows(xd)
    (out * r.to(hip_device)).sum().backward()
    P = {n: (t.clone().requires_grad_(True) if t.is_floating_point() and "running" not in n else t.clone()) for n, t in params.items()}
    xo = rows.clone().requires_grad_(True)
    ctx = sv_ref.Ctx(train=True, exact_ste=True)
    y = torch.relu(sv_ref.batch_norm_cf(sv_ref.conv1d(xo.transpose(1, 2), P, "0", binary, ctx), P, "1", ctx)).transpose(1, 2)
    (y * r).sum().backward()
    got = {"out0": out.detach().cpu().numpy(), "dx0": xd.grad.cpu().numpy()}
    want = {"out0": y.detach().numpy(), "dx0": xo.grad.numpy()}
    for n, p in blk.named_parameters():
        got["d:" + n], want["d:" + n] = p.grad.cpu().numpy(), P[n].grad.numpy()
    compare_case(got, want, 1e-4, "ConvBNReLU rows")


def test_sliced_sums_are_complete_every_time(hip_device):
    """The grid-wide reductions add into 16 slices and the NEXT kernel of the stream sums them (csrc/common.h svnet_slices_total; here
    svnet_slices_sum_*): the totals must hold EVERY workgroup's share, every time.  (Round 3 summed them in the reducing kernel's last
    workgroup to arrive; with no-return atomics one launch in a few hundred summed a slice that was still missing a share: a 6 % error.
    The hand-off is a kernel boundary now.)  300 launches per kernel on the conv5-sized tensors of the bench model, each against a
    float64 torch sum."""
    from svnet_amd import _ops
    from svnet_amd._ops import _p, _stream, call, _sliced_len
    torch.manual_seed(3)
    M, C = 32768, 170
    v = torch.randn(M, 3, C, device=hip_device)
    n = v.double().pow(2).sum(1).sqrt() + 1e-6
    ref_v = torch.cat([n.sum(0), n.pow(2).sum(0)])
    x = torch.randn(M, 512, device=hip_device)
    ref_x = torch.cat([x.double().sum(0), x.double().pow(2).sum(0)])
    g = torch.randn(M, 512, device=hip_device)
    mean, invstd = x.mean(0).contiguous(), (1.0 / x.std(0)).contiguous()
    gamma, beta = torch.ones(512, device=hip_device), torch.zeros(512, device=hip_device)
    xh = (x.double() - mean.double()) * invstd.double()
    ref_r = torch.cat([g.double().sum(0), (g.double() * xh).sum(0)])
    worst = [0.0, 0.0, 0.0]
    for it in range(300):
        sv = torch.zeros(_sliced_len(2 * C), dtype=torch.float64, device=hip_device)
        call("svnet_colstats_f64", _p(v), M, C, 1, _p(sv), _stream())
        call("svnet_slices_sum_f64", _p(sv), 2 * C, _stream())
        sx = torch.zeros(_sliced_len(2 * 512), dtype=torch.float64, device=hip_device)
        call("svnet_colstats_f64", _p(x), M, 512, 0, _p(sx), _stream())
        call("svnet_slices_sum_f64", _p(sx), 1024, _stream())
        red = torch.zeros(_sliced_len(2 * 512), dtype=torch.float32, device=hip_device)
        call("svnet_bn_act_bwd_reduce_f32", _p(g), _p(x), _p(mean), _p(invstd), _p(gamma), _p(beta), M, 512, 0, 0.2, _p(red), _stream())
        call("svnet_slices_sum_f32", _p(red), 1024, _stream())
        for i, (got, ref) in enumerate(((sv[:2 * C], ref_v), (sx[:1024], ref_x), (red[:1024].double(), ref_r))):
            worst[i] = max(worst[i], float((got - ref).abs().max() / ref.abs().max()))
    assert worst[0] < 1e-6 and worst[1] < 1e-6 and worst[2] < 1e-4, worst


# ----------------------------------------------------------------------------- linear2 + VectorBN statistics in one product (csrc/vlinear.hip)

@pytest.mark.parametrize("P,K,O,gated", [(1000, 10, 10, False), (4096, 21, 170, True), (2500, 83, 170, True), (777, 96, 256, False),
                                         (64, 20, 21, True), (33, 3, 5, False)])
def test_linear2_with_vectorbn_sums_matches_the_two_pass_chain(P, K, O, gated, hip_device):
    """SVBlock's vector path on rows, sv_layers.py:192-194: v' = bn2(linear2(v)) * gate.  The product that also forms the VectorBN's
    batch sums (n = ||y[p,:,o]|| + 1e-6: sum n, sum n^2) must give the layer-wise chain's numbers - rows GEMM, svnet_colstats_f64
    kind 1, apply - forward and backward, including a last tile of fewer than 32 points and column tiles past O."""
    from svnet_amd import _ops, config
    from svnet_amd.models.sv_layers import Linear, VectorBN
    torch.manual_seed(P + K + O)
    B = 1 if P % 8 else 8
    with contextlib.redirect_stdout(io.StringIO()):
        lin = Linear(K, O, bias=False, bw=True).to(hip_device).train()
        bn = VectorBN(O).to(hip_device).train()
    with torch.no_grad():
        lin.weight.copy_(torch.randn(O, K)); lin.weight[:, :1].zero_()              # a sign(0) = 0 column
        bn.bn.weight.copy_(torch.rand(O) + 0.5); bn.bn.bias.copy_(torch.randn(O) * 0.1)
    v = torch.randn(B, P // B, 3, K, device=hip_device) * 1.7
    gate = torch.rand(B, O, device=hip_device) if gated else None
    g = torch.randn(B, P // B, 3, O, device=hip_device)
    outs, default = {}, config.FUSE_VBN_STATS
    for fused in (False, True):
        config.FUSE_VBN_STATS = fused
        try:
            bn.bn.running_mean.zero_(); bn.bn.running_var.fill_(1.0); bn.bn.num_batches_tracked.zero_()
            vv = v.clone().requires_grad_(True)
            for p_ in list(lin.parameters()) + list(bn.parameters()):
                p_.grad = None
            calls = []
            real = _ops.call
            _ops.call = lambda name, *a: (calls.append(name), real(name, *a))[1]
            try:
                out = bn(lin(vv, vstats=True), gate=gate)
            finally:
                _ops.call = real
            out.backward(g)
            torch.cuda.synchronize()
            outs[fused] = dict(out=out.detach().clone(), dv=vv.grad.clone(), dW=lin.weight.grad.clone(), dsc=lin.scale.grad.clone(),
                               dg=bn.bn.weight.grad.clone(), db=bn.bn.bias.grad.clone(), rm=bn.bn.running_mean.clone(),
                               rv=bn.bn.running_var.clone(), calls=calls)
        finally:
            config.FUSE_VBN_STATS = default
    assert "svnet_vlinear_stats_f32" in outs[True]["calls"] and "svnet_colstats_f64" not in outs[True]["calls"]
    assert "svnet_colstats_f64" in outs[False]["calls"] and "svnet_vlinear_stats_f32" not in outs[False]["calls"]
    for key in ("out", "dv", "dW", "dsc", "dg", "db", "rm", "rv"):
        a, b = outs[True][key], outs[False][key]
        ref = b.abs().max().clamp_min(1e-30)
        if key == "dsc":      # VectorBN normalises the length: the output does not depend on linear2's scale, its gradient is the rounding
            ref = outs[False]["dW"].abs().sum(1).max()      # noise of a cancelling sum - measured against the terms it adds up
        err = float((a - b).abs().max() / ref)
        assert err < 2e-5, (key, err)


# ----------------------------------------------------------------------------- gate MLP with the mean over the rows inside its launch

@pytest.mark.parametrize("B,R,Cin,H,Ov", [(32, 1024, 32, 5, 10), (8, 1000, 64, 10, 21), (4, 256, 200, 85, 170), (3, 1, 96, 16, 32)])
def test_gate_mlp_on_rows_matches_pooling_then_mlp(B, R, Cin, H, Ov, hip_device):
    """sv_layers.py:179-183 on rows: gate = sigmoid(W2 relu(W0 mean_n s[b,n,:])).  _ops.GateMLPRows (the mean formed inside the MLP's launch,
    its gradient handed back as a broadcast view) against the chain it replaces - _ops.Pool (mean) then _ops.GateMLP - forward and
    backward, the gradient of s summed with that of a second consumer as autograd does in an SVBlock."""
    from svnet_amd import _ops
    torch.manual_seed(B * R + Cin)
    s0 = torch.randn(B, R, Cin, device=hip_device)
    W0 = (torch.randn(H, Cin, device=hip_device) * 0.3)
    W2 = (torch.randn(Ov, H, device=hip_device) * 0.3)
    g = torch.randn(B, Ov, device=hip_device)
    w_other = torch.randn(B, R, Cin, device=hip_device)
    res = []
    for fused in (False, True):
        s = s0.clone().requires_grad_(True)
        a, b = W0.clone().requires_grad_(True), W2.clone().requires_grad_(True)
        gate = _ops.GateMLPRows.apply(s, a, b) if fused else _ops.GateMLP.apply(_ops.Pool.apply(s, 1, 1), a, b)
        ((gate * g).sum() + (s * w_other).sum()).backward()
        torch.cuda.synchronize()
        res.append((gate.detach().clone(), s.grad.clone(), a.grad.clone(), b.grad.clone()))
    assert _ops.GateMLPRows.supported(s0)
    for x_, y_, name in zip(res[1], res[0], ("gate", "ds", "dW0", "dW2")):
        err = float((x_ - y_).abs().max() / y_.abs().max().clamp_min(1e-30))
        assert err < 1e-5, (name, err)


@pytest.mark.parametrize("training", [True, False], ids=["train", "eval"])
@pytest.mark.parametrize("B,N,Kc,Kp,O,at", [(4, 300, 200, 96, 64, 0), (2, 2048, 1600, 544, 256, 0), (3, 64, 70, 33, 40, 0), (32, 32, 1600, 544, 256, 0),
                                            (4, 256, 512, 1532, 512, 512), (3, 300, 70, 133, 64, 65)])
def test_binarized_layer_with_per_cloud_columns_equals_the_layer_on_the_concatenation(B, N, Kc, Kp, O, at, training, hip_device):
    """_ops.BinLinearCloud - the binarized layer on cat[expand(x_cloud), x_point] (sv_dgcnn_partseg.py:115-121: conv8 on the repeated
    per-cloud feature + the per-point feature; sv_layers.py:55-78) with the per-cloud columns counted once per cloud - against
    _ops.BinLinear on the materialised concatenation: outputs BIT-identical (the two integer counts add up to the full row's), the
    saved decision planes identical, every gradient to 2e-5 of its largest element (the per-cloud block's are sums in another order)."""
    from svnet_amd import _ops
    from tests.decisions import tapped
    g = torch.Generator().manual_seed(7 + Kc)
    K = Kc + Kp
    xc = torch.round(torch.randn(B, Kc, generator=g) * 3) / 3            # exact zeros / values on the STE clip among them
    xp = torch.randn(B, N, Kp, generator=g)
    xp[0, 1, ::5] = 0.0
    W = torch.randn(O, K, 1, generator=g)
    W[1, ::9] = 0.0
    beta = torch.randn(1, K, 1, generator=g) * 0.3
    beta[0, ::4] = 0.0
    scale = torch.rand(1, O, 1, generator=g) + 0.5
    w = torch.randn(B, N, O, generator=g).to(hip_device)
    res = {}
    for split in (True, False):
        leaves = [t.clone().to(hip_device).requires_grad_(True) for t in (xc, xp, W, beta, scale)]
        c, p, Wd, bd, sd = leaves
        with tapped() as tap:
            if split:
                y = _ops.BinLinearCloud.apply(c, p, Wd, bd, sd, training, at)
            else:           # (at: where the per-cloud block sits in the row - 0 = sv_dgcnn_partseg's conv8, 512 = sv_pointnet_cls's conv_fuse.linear1)
                rows = torch.cat([p[..., :at], c.unsqueeze(1).expand(B, N, Kc), p[..., at:]], dim=-1)
                y = _ops.BinLinear.apply(rows, Wd, bd, sd, None, training)
        (y * w).sum().backward()
        torch.cuda.synchronize()
        planes = [pl.cpu() for pl in tap["signs"][-1][3]]
        res[split] = dict(y=y.detach().cpu(), planes=planes, g=[t.grad.cpu() for t in leaves])
    assert torch.equal(res[True]["y"], res[False]["y"])
    for a, b in zip(res[True]["planes"], res[False]["planes"]):
        assert torch.equal(a, b)
    for name, a, b in zip(("x_cloud", "x_point", "W", "beta", "scale"), res[True]["g"], res[False]["g"]):
        assert a.shape == b.shape and torch.isfinite(a).all(), name
        assert float((a - b).abs().max()) <= 2e-5 * max(float(b.abs().max()), 1e-6), (name, float((a - b).abs().max()), float(b.abs().max()))
