#!/usr/bin/env python3
"""Generate tests/golden/*.npz by running the IMPORTED REFERENCE (/root/reference, read-only,
build container only) on the deterministic inputs of cases.py / harness.py.

    python -m tests.golden.make_golden          (from the repo root)

The reference has no tests or golden vectors of its own, so these fixtures are
what pins the oracle (oracle/) to the reference.  Only inputs' recipes and expected OUTPUTS are
stored; no reference source is copied.  The script also prints oracle-vs-reference deviations.
"""
import argparse
import contextlib
import io
import json
import os
import sys
from collections import OrderedDict

import numpy as np
import torch

HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(os.path.dirname(HERE))
sys.path.insert(0, ROOT)
REF = os.environ.get("SVNET_REFERENCE", "/root/reference")

from tests.golden import cases as C                      # noqa: E402
from tests.golden import harness as H                    # noqa: E402
from oracle import knn as oknn                           # noqa: E402
from oracle import params as oparams                     # noqa: E402
from oracle import sv_ref                                # noqa: E402


def import_reference():
    sys.path.insert(0, REF)
    with contextlib.redirect_stdout(io.StringIO()):
        import models as ref_models
        import models.sv_layers as ref_layers
        import models.utils.sv_util as ref_util
    sys.path.pop(0)
    return ref_models, ref_layers, ref_util


def ref_cal_loss():
    """utils.py of the reference imports only torch/numpy at module level."""
    import importlib.util
    spec = importlib.util.spec_from_file_location("ref_utils", os.path.join(REF, "utils.py"))
    mod = importlib.util.module_from_spec(spec)
    spec.loader.exec_module(mod)
    return mod.cal_loss


REF_CLASS = {"sv_dgcnn_cls": "SV_DGCNN_CLS", "sv_pointnet_cls": "SV_PointNet_CLS", "sv_dgcnn_pseg": "SV_DGCNN_PSEG"}


def build_ref_model(ref_models, model, binary, k):
    args = argparse.Namespace(k=k, binary=binary, dropout=0.5)
    with contextlib.redirect_stdout(io.StringIO()):
        if model == "sv_dgcnn_pseg":
            m = getattr(ref_models, REF_CLASS[model])(args, 50)
        else:
            m = getattr(ref_models, REF_CLASS[model])(args, 40)
    for mod in m.modules():
        if isinstance(mod, torch.nn.Dropout):
            mod.p = 0.0                                   # train-mode RNG is not reproducible (Appendix C7)
    return m


def seg_loss(logits, target):
    """main_partseg_dgcnn.py uses cal_loss on [B*N, 50] rows."""
    return logits.permute(0, 2, 1).reshape(-1, logits.shape[1]), target.reshape(-1)


def main():
    torch.manual_seed(0)
    torch.set_num_threads(8)
    ref_models, ref_layers, ref_util = import_reference()
    cal_loss = ref_cal_loss()
    report = []

    # ---- 1. k-NN indices ---------------------------------------------------------------------------
    knn_out = {}
    for (name, B, N, Cc, k, layout) in C.KNN_CASES:
        x = C.knn_input(name, B, N, Cc, k, layout)
        idx_ref = ref_util.knn(x, k)
        idx_orc, pd = oknn.knn_exact(x, k, return_pd=True)
        raw = int((idx_ref != idx_orc).sum())
        bad = oknn.tie_aware_mismatches(idx_ref, idx_orc, pd)
        report.append("knn %-14s B%d N%d C%d k%d %s: oracle mismatches %d (exact ties in other order: %d) / %d" % (
            name, B, N, Cc, k, layout, bad, raw - bad, idx_ref.numel()))
        assert bad == 0, report[-1]
        knn_out[name] = idx_ref.numpy().astype(np.int16)
    np.savez_compressed(os.path.join(HERE, "knn.npz"), **knn_out)

    # ---- 2. op-level cases ---------------------------------------------------------------------------
    ref_api = H.ModuleAPI(ref_layers, ref_util, "cpu")
    orc_api = H.OracleAPI()
    ops = {}
    for name, fn in H.op_cases().items():
        r = H.to_numpy(fn(ref_api))
        o = H.to_numpy(fn(orc_api))
        assert set(r) == set(o), (name, sorted(r), sorted(o))
        worst = max(H.max_rel_err(o[key], r[key]) for key in r) if r else 0.0
        report.append("op  %-28s keys %2d  oracle max rel err %.2e" % (name, len(r), worst))
        for key, val in r.items():
            ops["%s/%s" % (name, key)] = val
    np.savez_compressed(os.path.join(HERE, "ops.npz"), **ops)

    # ---- 3. model-level cases ------------------------------------------------------------------------
    layout = OrderedDict()
    models_out = {}
    for (tag, model, binary, B, N, k) in C.MODEL_CASES:
        P = oparams.synthetic_params(model, binary=binary, seed=C.SEED)
        x, l, y = C.model_inputs(tag, model, B, N)
        res = OrderedDict()
        # eval
        m = build_ref_model(ref_models, model, binary, k)
        layout["%s/%s" % (model, "binary" if binary else "fp")] = [(n, list(t.shape)) for n, t in m.state_dict().items()]
        m.load_state_dict(P, strict=True)
        m.eval()
        with torch.no_grad():
            res["logits_eval"] = (m(x, l) if l is not None else m(x)).numpy()
        # train (fresh module: BN buffers untouched)
        m = build_ref_model(ref_models, model, binary, k)
        m.load_state_dict(P, strict=True)
        m.train()
        logits = m(x, l) if l is not None else m(x)
        loss = cal_loss(*seg_loss(logits, y)) if l is not None else cal_loss(logits, y)
        loss.backward()
        res["logits_train"] = logits.detach().numpy()
        res["loss_train"] = loss.detach().numpy()
        names = [n for n, p in m.named_parameters()]
        res["grad_norms"] = np.array([float(p.grad.norm()) if p.grad is not None else 0.0 for _, p in m.named_parameters()], dtype=np.float32)
        sd = m.state_dict()
        bn_key = "conv2.bn1" if "conv2.bn1.running_mean" in sd else "feat.conv1.bn1"
        res["bn_running_mean"] = sd[bn_key + ".running_mean"].numpy()
        res["bn_running_var"] = sd[bn_key + ".running_var"].numpy()
        small = [n for n in names if n.endswith((".scale", ".beta")) and ("conv4" in n or "conv2" in n or "feat.conv2" in n)]
        for n in small:
            res["grad:" + n] = dict(m.named_parameters())[n].grad.numpy()
        # oracle cross-check
        ctx = sv_ref.Ctx(train=False)
        fwd = {"sv_dgcnn_cls": lambda P_, c: sv_ref.sv_dgcnn_cls(x, P_, k, binary, c),
               "sv_pointnet_cls": lambda P_, c: sv_ref.sv_pointnet_cls(x, P_, k, binary, c),
               "sv_dgcnn_pseg": lambda P_, c: sv_ref.sv_dgcnn_pseg(x, l, P_, k, binary, c)}[model]
        with torch.no_grad():
            e_eval = H.max_rel_err(fwd(P, ctx).numpy(), res["logits_eval"])
        Pg = oparams.synthetic_params(model, binary=binary, seed=C.SEED, requires_grad=True)
        ctx = sv_ref.Ctx(train=True, collect_bn=True)
        lo = fwd(Pg, ctx)
        e_train = H.max_rel_err(lo.detach().numpy(), res["logits_train"])
        lss = sv_ref.cal_loss(*seg_loss(lo, y)) if l is not None else sv_ref.cal_loss(lo, y)
        lss.backward()
        gn = np.array([float(Pg[n].grad.norm()) if Pg[n].grad is not None else 0.0 for n in names], dtype=np.float32)
        e_gn = H.max_rel_err(gn, res["grad_norms"])
        report.append("mdl %-20s eval %.2e train %.2e loss %.2e gradnorms %.2e" % (
            tag, e_eval, e_train, abs(float(lss) - float(res["loss_train"])), e_gn))
        for key, val in res.items():
            models_out["%s/%s" % (tag, key)] = val
        models_out["%s/param_names" % tag] = np.array(names)
    np.savez_compressed(os.path.join(HERE, "models.npz"), **models_out)
    with open(os.path.join(HERE, "state_layout.json"), "w") as f:
        json.dump(layout, f, indent=0)

    print("\n".join(report))
    for fn in ("knn.npz", "ops.npz", "models.npz", "state_layout.json"):
        print("%-20s %8d bytes" % (fn, os.path.getsize(os.path.join(HERE, fn))))


if __name__ == "__main__":
    main()
