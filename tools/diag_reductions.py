"""Diagnostic: every grid-wide reduction of one train step of a parity case, re-computed with torch in float64 from the call's own arguments
(svnet_colstats_f64 kind 0 / 1, svnet_bn_act_bwd_reduce_f32): python tools/diag_reductions.py pseg_fp_b32"""
import os, sys, ctypes
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from svnet_amd import _lib, _ops
from tests import test_hip_train_parity as T
tag = sys.argv[1] if len(sys.argv) > 1 else "pseg_fp_b32"
case = [c for c in T.TRAIN_CASES if c[0] == tag][0]
tag, model, binary, B, N, k = case
dev = torch.device("cuda:0")
real = _lib.call
def t_of(ptr, shape, dtype):
    n = 1
    for s in shape: n *= s
    esz = torch.empty((), dtype=dtype).element_size()
    buf = (ctypes.c_char * (n * esz)).from_address(0)          # placeholder (device memory: read through torch below)
    return None
def dev_tensor(ptr, shape, dtype):
    # wrap a raw device pointer: copy through a ctypes-free route (hipMemcpy via torch.cuda) - use the __cuda_array_interface__ protocol
    class _W:
        pass
    w = _W()
    n = 1
    for s in shape: n *= s
    typestr = {torch.float32: "<f4", torch.float64: "<f8"}[dtype]
    w.__cuda_array_interface__ = {"shape": tuple(shape), "typestr": typestr, "data": (int(ptr.value), False), "version": 2}
    return torch.as_tensor(w, device=dev)
worst = []
def checked(name, *a):
    rc = real(name, *a)
    if name == "svnet_colstats_f64":
        x, M, C, kind, sums = a[0], a[1], a[2], a[3], a[4]
        torch.cuda.synchronize()
        got = dev_tensor(sums, (17, 2 * C), torch.float64)[1:].sum(0)          # the 16 slices (their consumer kernel adds them up)
        if kind == 0:
            xt = dev_tensor(x, (M, C), torch.float32).double()
            ref = torch.cat([xt.sum(0), xt.pow(2).sum(0)])
        else:
            xt = dev_tensor(x, (M, 3, C), torch.float32).double()
            n = xt.pow(2).sum(1).sqrt() + 1e-6
            ref = torch.cat([n.sum(0), n.pow(2).sum(0)])
        worst.append((float((got - ref).abs().max() / ref.abs().max()), name, kind, M, C))
    elif name == "svnet_bn_act_bwd_reduce_f32":
        g, x, mean, invstd, gamma, beta, M, C, act, slope, red = a[:11]
        torch.cuda.synchronize()
        got = dev_tensor(red, (17, 2 * C), torch.float32)[1:].double().sum(0)
        gt, xt = dev_tensor(g, (M, C), torch.float32).double(), dev_tensor(x, (M, C), torch.float32).double()
        mu, isd = dev_tensor(mean, (C,), torch.float32).double(), dev_tensor(invstd, (C,), torch.float32).double()
        ga, be = dev_tensor(gamma, (C,), torch.float32).double(), dev_tensor(beta, (C,), torch.float32).double()
        xh = (xt - mu) * isd
        z = xh * ga + be
        ag = torch.ones_like(z) if act == 0 else (torch.where(z > 0, 1.0, float(slope)) if act == 1 else (z > 0).double())
        gp = gt * ag
        ref = torch.cat([gp.sum(0), (gp * xh).sum(0)])
        worst.append((float((got - ref).abs().max() / ref.abs().max()), name, act, M, C))
    return rc
for mod in list(sys.modules.values()):
    if mod is not None and getattr(mod, "__name__", "").startswith("svnet_amd") and getattr(mod, "call", None) is real:
        mod.call = checked
P = T.oparams.synthetic_params(model, binary=binary, seed=T.C.SEED)
x, l, y = T.C.model_inputs(tag, model, B, N)
m = T.build_model(model, binary, k, dev, P).train()
logits, loss, got, tap = T.hip_step(m, x, l, y, dev)
for w in sorted(worst, reverse=True)[:12]:
    print(w)
print("calls checked:", len(worst))
