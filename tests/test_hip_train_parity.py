"""GPU tests (-m gpu): a whole train step (fwd + cal_loss + bwd) of every caller of the hot path against the oracle,
ELEMENT-WISE on the logits, the loss and every parameter gradient.

Binary models are compared with the oracle's exact-STE mode (oracle/sv_ref.py Ctx(exact_ste=True)): the reference's train-mode
binarize evaluates (sign + x) - x in fp32, which is 1 +- 1.2e-7 in 10-20 % of the elements, and that noise is the only thing that
orders max-pool ties between equal integer popcounts there.  tests/golden/make_golden.py and tests/test_oracle_golden.py show
that the exact mode and the reference agree to 3e-5 on every gradient once the reference's arg-max selections are replayed, and
that the selections differ ONLY at exact ties.  The HIP path computes exact +-1/0 and uses torch's first-index rule, so it must
match the exact-STE oracle element by element; a gradient that does not is a kernel bug.
"""
import argparse
import contextlib
import io
import json
import os

import numpy as np
import pytest
import torch

from oracle import params as oparams
from oracle import sv_ref
from tests.common import case_errors, compare_case
from tests.decisions import decisions_of, tapped
from tests.golden import cases as C
from tests.golden import harness as H

pytestmark = pytest.mark.gpu
from tests.conftest import DEFAULT_THREADS, cpu_share as conftest_cpu_share    # noqa: E402
OUT = os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "gpurun_out")
GRAD_RTOL = 1e-3        # north_star: 1e-3 relative for activations / gradients


def build_model(model, binary, k, dev, state):
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


def oracle_step(model, binary, k, x, l, y, decisions=None, dtype=torch.float32):
    """One train step of the oracle (exact-STE mode for binary models); `decisions`: the HIP run's discrete choices, replayed and
    certified (tests/decisions.py); dtype float64 = the same function in double precision (the yard-stick of the comparison)."""
    Pg = oparams.synthetic_params(model, binary=binary, seed=C.SEED, requires_grad=True)
    if dtype != torch.float32:
        Pg = {n: (t.detach().to(dtype).requires_grad_(t.requires_grad) if t.is_floating_point() else t) for n, t in Pg.items()}
        x = x.to(dtype)
        l = None if l is None else l.to(dtype)
    ctx = sv_ref.Ctx(train=True, exact_ste=binary)
    ctx.decisions = decisions
    fwd = {"sv_dgcnn_cls": lambda: sv_ref.sv_dgcnn_cls(x, Pg, k, binary, ctx),
           "sv_pointnet_cls": lambda: sv_ref.sv_pointnet_cls(x, Pg, k, binary, ctx),
           "sv_dgcnn_pseg": lambda: sv_ref.sv_dgcnn_pseg(x, l, Pg, k, binary, ctx),
           "sv_pointnet_pseg": lambda: sv_ref.sv_pointnet_pseg(x, l, Pg, k, binary, ctx)}[model]
    lo = fwd()
    ls = sv_ref.cal_loss(lo.permute(0, 2, 1).reshape(-1, lo.shape[1]), y.reshape(-1)) if l is not None else sv_ref.cal_loss(lo, y)
    ls.backward()
    return lo.detach(), float(ls), Pg


def hip_step(m, x, l, y, dev):
    """fwd + loss + bwd on the HIP path with its discrete decisions recorded: (logits, loss, {d:name: grad}, tap)."""
    from svnet_amd.train import cal_loss, seg_loss
    with tapped() as tap:
        if l is not None:
            logits = m(x.to(dev), l.to(dev))
            loss = seg_loss(logits, y.to(dev))
        else:
            logits = m(x.to(dev))
            loss = cal_loss(logits, y.to(dev))
    loss.backward()
    got = {"d:" + n: p.grad.detach().cpu().numpy() for n, p in m.named_parameters()}
    return logits.detach().cpu().numpy(), float(loss), got, tap


# (tag, model, binary, B, N, k): the golden small cases plus, per caller, a size at which the ORACLE's own train step is as well
# conditioned as that caller gets (per-cloud BatchNorms over 16-32 rows instead of 2-4)
# (pseg_fp_small - B = 2 - is not a train-step case: the BatchNorms of sv_dgcnn_partseg's per-cloud blocks conv6 / conv7 then see TWO rows,
#  every gradient is amplified rounding noise, and two correct fp32 implementations differ by 1e-2 on tensors that change from build to
#  build (measured: the same HIP path before and after one GEMM changed its summation order).  Its eval logits are pinned in
#  test_hip_parity.py, its train step by the B = 32 twin pseg_fp_b32, which holds 1e-3.)
TRAIN_CASES = [c for c in C.MODEL_CASES if c[0].endswith("_small") and c[0] != "pseg_fp_small"] + [
    ("dgcnn_bin_b16", "sv_dgcnn_cls", True, 16, 64, 8), ("dgcnn_bin_b16b", "sv_dgcnn_cls", True, 16, 64, 8),
    ("dgcnn_bin_b8", "sv_dgcnn_cls", True, 8, 128, 10), ("dgcnn_fp_b16", "sv_dgcnn_cls", False, 16, 64, 8),
    ("pseg_bin_b32", "sv_dgcnn_pseg", True, 32, 32, 6), ("pseg_fp_b32", "sv_dgcnn_pseg", False, 32, 32, 6),
    ("pointnet_bin_b16", "sv_pointnet_cls", True, 16, 64, 8), ("pointnet_fp_b32", "sv_pointnet_cls", False, 32, 32, 6),
    ("ppseg_fp_b16", "sv_pointnet_pseg", False, 16, 64, 8),
    # the bench's own shape (N = 1024, k = 20; B = 8 so that the XCD-aware cloud order is in play): the N = 1024 kernel instantiations -
    # knn_main<16,4,...>, the conv4 tile kernel <0,8,44>, mfma_tn_aff2, chunked reverse lists, sliced sums over thousands of workgroups -
    # chained against the oracle, where the other cases stop at N = 128 (sv_dgcnn_cls.py:46-82)
    ("dgcnn_bin_n1024", "sv_dgcnn_cls", True, 8, 1024, 20),
    # BASELINE config 5's shape (N = 2048, k = 40; sv_dgcnn_partseg.py:80-128): the k = 40 backward instantiations -
    # edgeblock_bwd_kernel<0,8,48> / <0,4,44> / <0,2,44>, the N = 2048 reverse lists, the vector-form k-NN with its merged selection,
    # the rows head on 2144 columns - chained against the oracle in TRAIN mode (rounds 1-4 stopped at N = 128 there).  B = 2, like the
    # STRICT pseg_bin_small (the binary model's per-cloud blocks are well conditioned at two rows; only the fp twin is not): the float64
    # oracle step on E = 163 840 edges of 272 columns is what bounds the size - B = 4 ran past seven minutes on the GPU box's host
    ("pseg_bin_n2048", "sv_dgcnn_pseg", True, 2, 2048, 40),
]
# cases held to the north-star tolerance itself (1e-3) on every tensor, whatever the yard-sticks say
STRICT = ("dgcnn_bin_small", "dgcnn_fp_small", "pseg_bin_small", "dgcnn_bin_b16", "dgcnn_bin_b16b", "dgcnn_bin_b8", "dgcnn_fp_b16",
          "pseg_bin_b32", "pseg_fp_b32", "dgcnn_bin_n1024", "pseg_bin_n2048")
WIDER_CERTIFICATE = ("ppseg_fp_b16",)          # the one case whose decision certificate allows 30 instead of 20 rms of fp32 noise (see below)
NO_SENSITIVITY_LEG = ("dgcnn_bin_n1024", "pseg_bin_n2048")     # STRICT cases never use the float64 sensitivity (a third oracle step: ~1 min at this size)
YARDSTICK = 3.0         # a tensor may be this many times further from the float64 truth than the fp32 oracle is ...
SENSITIVITY = 10.0      # ... or this many times what the float64 truth itself moves when its input moves by one part in 1e7


def _train_step_case(case, hip_device, corrupt=None):
    """fwd + cal_loss + bwd on the HIP path against the oracle: logits, loss and EVERY parameter gradient, element-wise, for every
    caller of the path, fp and binary.

    1. The HIP run's discrete decisions (neighbour lists, binarized signs + STE masks, max-pool arg-max) are recorded and replayed
       into the oracle, which certifies each one it would have taken differently as a knife edge of its own arithmetic
       (oracle.sv_ref.Decisions.check - the test fails on any other disagreement).  There is no skip and no statistical exit:
       after a certified flip everything downstream is still compared.
    2. Truth T = the same oracle, same decisions, in float64.  Every tensor of the HIP step must lie within
       max(1e-3, YARDSTICK x the fp32 oracle's own distance from T, SENSITIVITY x the distance T itself moves when every input
       coordinate moves by one part in 1e7 - the largest move over three random sign patterns) of T (relative to the tensor's max, tests/common.py case_errors).  With the decisions
       fixed T is a smooth function, so the third term is its condition number times one fp32 ulp - a deterministic yard-stick;
       the second is one DRAW of fp32 rounding noise through the same amplification (it changes 8-fold per tensor between two
       hosts' BLAS / thread counts, measured), which is why it is not used alone.  For the SV-DGCNN callers both are below 1e-4,
       i.e. the bound is the north star's 1e-3 (asserted: STRICT); the PointNet callers' train step, and any caller at B = 2, is
       ill-conditioned in the reference itself (BatchNorms over the B per-cloud rows, vector norms near zero: the fp32 oracle is
       up to 0.17 away from its own float64 evaluation on sv_pointnet_partseg), and there these measured terms are the bound."""
    tag, model, binary, B, N, k = case
    P = oparams.synthetic_params(model, binary=binary, seed=C.SEED)
    x, l, y = C.model_inputs(tag, model, B, N)
    m = build_model(model, binary, k, hip_device, P).train()
    logits, loss, got, tap = hip_step(m, x, l, y, hip_device)
    if corrupt is not None:
        corrupt(got)
    # truth first: the oracle in float64 on the HIP run's decisions, keeping its values at every decision point; then the fp32
    # oracle on the same decisions, whose certificate measures every disagreement against 2e-5 of the value's magnitude plus
    # 10x its own rms distance from the float64 values there
    # (STRICT cases also replay the kink decisions of the BatchNorm + ReLU / LeakyReLU layers the HIP path runs as such - heads, layer-wise
    #  blocks: a head activation 1e-7 from its kink flipped by a one-ulp change upstream moved pseg_fp_b32's gradients by 2e-3)
    # (round 4: EVERY case replays the kink decisions - the fused edge layers' per-edge LeakyReLU and conv5's pooled BatchNorm are tapped
    #  now as well, svnet_amd._ops.  With them replayed, round 3's loosenings for the ill-conditioned callers were tried without: the
    #  doubled yard-stick is needed by NO case any more (retired); the wider certificate by ONE decision of ONE case, named below)
    dmodel = m
    dec64 = decisions_of(tap, model=dmodel)
    dec64.value_record = {"knn": [], "signs": [], "pools": [], "acts": {}}
    lo64, ls64, Pg64 = oracle_step(model, binary, k, x, l, y, dec64, torch.float64)
    def truth_copy():        # (the certificate pops the recorded float64 values as it goes: every fp32 run gets its own lists)
        return {k_: (list(v_) if isinstance(v_, list) else dict(v_)) for k_, v_ in dec64.value_record.items()}
    dec = decisions_of(tap, model=dmodel)
    dec.truth = truth_copy()
    if tag in WIDER_CERTIFICATE:
        # ppseg_fp_b16 (sv_pointnet_partseg: a 1e-7 input change moves its logits by 4e-4) decides its global max over the points among
        # ~20 near-ties in 65 504 values that carry amplified rounding noise, and their gaps spread up to the certificate's threshold:
        # largest margin 0.75 of it with one build of the first-layer kernel, 1.001 with the next (only a summation order changed),
        # 1.0007 this round at the default 20 rms with every other loosening off (gpurun_out/r04_t11_retire.log).  The HIP step and the
        # fp32 oracle are two independent draws of that noise and the largest of 65 504 is several rms: 30 rms for THIS case's certificate.
        dec.noise_factor = 30.0
    lo, ls, Pg = oracle_step(model, binary, k, x, l, y, dec)
    cert_threads = torch.get_num_threads()
    try:
        cert = dec.check()
    except AssertionError:
        # The certificate's noise term is ONE draw of fp32 rounding noise - the fp32 oracle's rms distance from its float64 run at the
        # layer - and for the ill-conditioned callers that draw depends on how torch splits its reductions: at ppseg_bin_small's
        # conv_fuse1 it is 3.9e-4 with 128 threads and 5.8e-5 with 16 (gpurun_out/r05_*: the same 38 replayed signs, margin 0.35 / 2.3).
        # A decision counts as a knife edge when it lies within the noise of SOME correct fp32 evaluation of the oracle: the
        # non-STRICT cases get a second draw at torch's default thread count (tests/conftest.py narrows it to the CPU share).
        if tag in STRICT or DEFAULT_THREADS == cert_threads:
            raise
        torch.set_num_threads(DEFAULT_THREADS)
        try:
            dec = decisions_of(tap, model=dmodel)
            dec.truth = truth_copy()
            if tag in WIDER_CERTIFICATE:
                dec.noise_factor = 30.0
            lo, ls, Pg = oracle_step(model, binary, k, x, l, y, dec)
            cert = dec.check()
            cert_threads = DEFAULT_THREADS
        finally:
            torch.set_num_threads(min(DEFAULT_THREADS, conftest_cpu_share()))
    from svnet_amd import synth
    names = [n for n, _ in m.named_parameters()]
    truth = {"d:" + n: Pg64[n].grad.numpy() for n in names}
    # the condition number is a supremum over perturbation directions: one random direction under-estimates it on single tensors (measured:
    # d:fc1.weight of pointnet_fp_small moved 1.6e-3 under the first direction while the HIP step's error there went from 4e-3 to 1.6e-2
    # when ONE forward GEMM changed its summation order), so the ill-conditioned cases take the largest move over three directions
    e_sens, l_sens_all = {}, []
    for probe in range(0 if tag in NO_SENSITIVITY_LEG else 1 if tag in STRICT else 3):
        wiggle = torch.from_numpy(np.sign(synth.normal(99 + probe, 1, tuple(x.shape)))).double()
        lo64w, _, Pg64w = oracle_step(model, binary, k, x.double() * (1.0 + 1e-7 * wiggle), l, y, decisions_of(tap, model=dmodel), torch.float64)
        for n, e in case_errors({"d:" + n: Pg64w[n].grad.numpy() for n in names}, truth).items():
            e_sens[n] = max(e, e_sens.get(n, 0.0))
        l_sens_all.append(lo64w)
    l_sens = max([H.max_rel_err(w.numpy(), lo64.numpy()) for w in l_sens_all], default=0.0)
    if not e_sens:
        assert tag in STRICT
        e_sens = {"d:" + n: 0.0 for n in names}
    ref = {"d:" + n: Pg[n].grad.numpy() for n in names}
    e_hip, e_orc = case_errors(got, truth), case_errors(ref, truth)
    l_hip, l_orc = H.max_rel_err(logits, lo64.numpy()), H.max_rel_err(lo.numpy(), lo64.numpy())
    strict = tag in STRICT
    # (round 3 doubled the yard-stick where the fp32 oracle itself is > 1e-2 from float64; with the kink decisions of every layer replayed
    #  no case needs that any more - the worst tensor of the family is 1.9e-2 against a bound of 8.5e-2, gpurun_out/r04_t11_retire.log)
    yard = YARDSTICK
    tol = {n: GRAD_RTOL if strict else max(GRAD_RTOL, yard * e_orc[n], SENSITIVITY * e_sens[n]) for n in e_hip}
    tol_logits = 1e-3 if strict else max(1e-3, yard * l_orc, SENSITIVITY * l_sens)
    os.makedirs(OUT, exist_ok=True)
    with open(os.path.join(OUT, "train_step_grad_errors_%s.json" % tag), "w") as f:
        json.dump({"logits_err_vs_f64": l_hip, "oracle_fp32_logits_err_vs_f64": l_orc, "loss": [loss, ls, ls64],
                   "worst_grad_err_vs_f64": max(e_hip.values()), "oracle_fp32_worst_grad_err_vs_f64": max(e_orc.values()),
                   "replayed_decisions": cert, "yardstick_factor": yard, "certificate_noise_factor": dec.noise_factor,
                   "certificate_oracle_threads": cert_threads,
                   "f64_worst_grad_move_under_1e-7_input_change": max(e_sens.values()), "f64_logits_move_under_1e-7_input_change": l_sens,
                   "grads (hip vs f64, oracle fp32 vs f64, f64 sensitivity, bound, name)":
                       sorted(((e, e_orc[n], e_sens[n], tol[n], n) for n, e in e_hip.items()), reverse=True)[:25]},
                  f, indent=0)
    assert np.isfinite(loss) and all(np.isfinite(v).all() for v in got.values())
    assert l_hip <= tol_logits, (l_hip, l_orc, cert)
    assert abs(loss - ls64) <= max(1e-4, tol_logits) * max(1.0, abs(ls64)), (loss, ls, ls64)
    bad = sorted(((e / tol[n], e, e_orc[n], n) for n, e in e_hip.items() if e > tol[n]), reverse=True)
    if corrupt is not None:
        return bad
    assert not bad, "train step grads (%s): %d tensors beyond max(1e-3, %gx the fp32 oracle's error vs float64, %gx float64's sensitivity): %r; replay %r" % (
        tag, len(bad), YARDSTICK, SENSITIVITY, bad[:5], cert)


@pytest.mark.parametrize("case", TRAIN_CASES, ids=[c[0] for c in TRAIN_CASES])
def test_train_step_matches_oracle_elementwise(case, hip_device):
    _train_step_case(case, hip_device)


test_train_step_matches_oracle_elementwise.__doc__ = _train_step_case.__doc__


def test_a_one_percent_gradient_error_is_caught(hip_device):
    """The comparison has teeth on the cases that used to be waved through: dgcnn_bin_b16 (round 2: pytest.skip behind a flip
    certificate) with ONE parameter gradient of a fused edge layer off by 1 % must fail - on exactly that tensor.  (The fused
    layer's gradient with the largest entries: errors are measured against max(the tensor's own max, 1e-2 of the largest gradient
    of the step), tests/common.py, so 1 % of a tensor that small would be below the noise floor by construction.)"""
    case = [c for c in TRAIN_CASES if c[0] == "dgcnn_bin_b16"][0]
    picked = []

    def corrupt(got):
        fused = [n for n in got if n.split(".")[0] in ("d:conv2", "d:conv3", "d:conv4") and not n.endswith(".scale")]
        name = max(fused, key=lambda n: float(np.abs(got[n]).max()))
        assert float(np.abs(got[name]).max()) > 2e-2 * max(float(np.abs(v).max()) for v in got.values()), name     # (above the noise floor)
        got[name] = got[name] * np.float32(1.01)
        picked.append(name)
    bad = _train_step_case(case, hip_device, corrupt)
    assert [b[3] for b in bad] == picked, (bad, picked)


@pytest.mark.parametrize("shape", [((64, 21), (128, 42), 2, 1024, 20), ((32, 10), (32, 10), 2, 1024, 20), ((32, 10), (64, 21), 1, 512, 20),
                                   # sv_dgcnn_partseg's widths at ITS size (sv_dgcnn_partseg.py:52-58, N = 2048, k = 40)
                                   ((32, 16), (32, 16), 2, 2048, 40), ((32, 16), (64, 24), 1, 2048, 40), ((64, 24), (128, 40), 2, 2048, 40)],
                         ids=["conv4", "conv2", "conv3", "pseg_conv2", "pseg_conv3", "pseg_conv4"])
def test_fused_edge_block_backward_matches_exact_oracle(shape, hip_device):
    """get_graph_feature_sv -> SVBlock(binary) -> svpool at the headline widths and N=1024, k=20, and at config 5's widths with
    N=2048, k=40: outputs, input gradients and every
    parameter gradient of the FUSED path (what bench.py runs) against the exact-STE oracle, element-wise."""
    from svnet_amd.models.sv_layers import SVBlock
    from svnet_amd.models.utils.sv_util import get_graph_feature_sv, svpool
    (Cs, Cv), (Os, Ov), B, N, k = shape
    in_dims, out_dims = (2 * Cs, 2 * Cv), (Os, Ov)
    tag = "fused_full_%d" % Os if N <= 1024 else "fused_pseg_%d_n%d" % (Os, N)
    params = H.module_params("SVBlock", (in_dims, out_dims, True), tag)
    params["linear1.beta"][:, ::4] = 0.0
    with contextlib.redirect_stdout(io.StringIO()):
        blk = SVBlock(in_dims, out_dims, binary=True)
    blk.load_state_dict(params)
    blk = blk.to(hip_device).train()
    s, v = C.sv_pair(tag + "/pt", (B, N), Cs, Cv, 1.0)
    s = torch.round(s * 4) / 4                         # discrete scalars like a binary net's: exact zeros / ties
    sd, vd = s.to(hip_device).requires_grad_(True), v.to(hip_device).requires_grad_(True)
    # at config 5's size the layer takes 163 840 x 272 sign decisions and ~10^5 arg-max selections: a handful are knife edges of rounding
    # (|s_v + beta| within an ulp of zero; seen: ONE flipped sign = one popcount off by 2 = 2.7e-2 of the output's range), so there the HIP
    # run's decisions are replayed into the oracle and every one it would have taken differently is certified (tests/decisions.py) -
    # the element-wise bound stays 1e-3.  The headline-size shapes keep running free.
    replay = N > 1024
    with (tapped() if replay else contextlib.nullcontext()) as tap:
        edges = get_graph_feature_sv((sd, vd), k=k)
        idx = edges.idx.cpu()
        os_, ov = svpool(blk(edges))
    rs, rv = C.t(tag + "/rs", tuple(os_.shape)), C.t(tag + "/rv", tuple(ov.shape))
    ((os_ * rs.to(hip_device)).sum() + (ov * rv.to(hip_device)).sum()).backward()
    got = {"out0": os_.detach().cpu().numpy(), "out1": ov.detach().cpu().numpy(), "dx0": sd.grad.cpu().numpy(), "dx1": vd.grad.cpu().numpy()}
    got.update({"d:" + n: p.grad.cpu().numpy() for n, p in blk.named_parameters()})
    # oracle on the same inputs and the SAME graph (the graph itself is checked bit-exactly by the k-NN tests)
    P = {"m." + n: t.clone().requires_grad_(t.is_floating_point()) for n, t in params.items()}
    so, vo = s.clone().requires_grad_(True), v.clone().requires_grad_(True)
    ctx = sv_ref.Ctx(train=True, exact_ste=True)
    if replay:
        dec = decisions_of(tap, model=blk)
        dec.knn = []                                   # (the graph is handed over below, not decided by the oracle)
        dec.acts = {"m." + n: a for n, a in dec.acts.items()}
        ctx.decisions = dec
    glob = (idx + torch.arange(B).view(B, 1, 1) * N).reshape(-1)
    oo, ovv = sv_ref.svpool(sv_ref.svblock(sv_ref.graph_feature_sv((so, vo), k=k, idx=glob, ctx=ctx), P, "m", True, ctx), ctx=ctx)
    cert = ctx.decisions.check() if replay else None
    ((oo * rs).sum() + (ovv * rv).sum()).backward()
    ref = {"out0": oo.detach().numpy(), "out1": ovv.detach().numpy(), "dx0": so.grad.numpy(), "dx1": vo.grad.numpy()}
    ref.update({"d:" + n: P["m." + n].grad.numpy() for n, _ in blk.named_parameters()})
    report = sorted(((H.max_rel_err(got[kn], ref[kn]), kn) for kn in ref), reverse=True)
    os.makedirs(OUT, exist_ok=True)
    with open(os.path.join(OUT, "fused_block_errors_%s.json" % tag), "w") as f:
        json.dump({"errors": [(float(e), n) for e, n in report], "replayed_decisions": cert}, f, indent=0)
    compare_case(got, ref, GRAD_RTOL, "fused edge block vs exact oracle (%s)" % tag)
