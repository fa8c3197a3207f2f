"""Times every C-ABI call of conv5 (the SVBlock on materialised rows of sv_dgcnn_cls: (256, 83) -> (512, 170), binarized) + svfuse +
global max/mean pooling, forward and backward, each ALONE (side stream = main stream, a HIP event pair around every call) at the
headline size B=32, N=1024.  Prints the calls in launch order with their microseconds.  Diagnostic."""
import os, sys, contextlib, io, collections
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
import svnet_amd
from svnet_amd import _lib, _ops
from svnet_amd.models.sv_layers import SVBlock, SVFuse

_ops._side_stream = lambda dev: torch.cuda.current_stream(dev)
records = []
real_call = _lib.call
def timed_call(name, *args):
    a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    a.record()
    real_call(name, *args)
    b.record()
    records.append((name, a, b))
for mod in list(sys.modules.values()):
    if mod is not None and getattr(mod, "__name__", "").startswith("svnet_amd") and getattr(mod, "call", None) is real_call:
        mod.call = timed_call

B, N = 32, 1024
with contextlib.redirect_stdout(io.StringIO()):
    blk = SVBlock((256, 83), (512, 170), binary=True).cuda().train()
    fuse = SVFuse(170, 3, True).cuda().train()
s = torch.randn(B, N, 256, device="cuda", requires_grad=True)
v = torch.randn(B, N, 3, 83, device="cuda", requires_grad=True)
for it in range(3):
    records.clear()
    torch.cuda.synchronize()
    # (the classifier's own tail, sv_dgcnn_cls.py:69-74: bn1 + LeakyReLU of conv5 inside the pooling pass)
    bn = blk.bn1
    if os.environ.get("SVNET_NO_VTAIL"):
        y5, v5 = blk.forward_prebn((s, v))
        sv5 = fuse.v2s(v5)
        pooled = _ops.GlobalMaxMeanPoolBN.apply(y5, sv5, bn.weight, bn.bias, bn.running_mean, bn.running_var, True, 1, 0.2, bn.num_batches_tracked, bn.eps, 0.1)
    else:           # the vector half's tail as one pass each way (csrc/vtail.hip)
        bn2, fz = blk.bn2.bn, fuse.v2s.linear
        y5, v_lin, gate = blk.forward_pretail((s, v))
        pooled = _ops.GlobalMaxMeanPoolBNV.apply(y5, v_lin, gate, bn.weight, bn.bias, bn.running_mean, bn.running_var, bn2.weight, bn2.bias,
                                                 bn2.running_mean, bn2.running_var, fz.weight, fz.scale, True, 1, 0.2, bn.num_batches_tracked,
                                                 bn2.num_batches_tracked, bn.eps, 0.1)
    mark = len(records)
    pooled.sum().backward()
    torch.cuda.synchronize()
tot = [0.0, 0.0]
for i, (name, a, b) in enumerate(records):
    us = a.elapsed_time(b) * 1e3
    tot[i >= mark] += us
    print("%s %-40s %8.1f" % ("bwd" if i >= mark else "fwd", name[6:], us))
print("forward %.1f us, backward %.1f us (sum of the C-ABI calls; torch's own kernels - cat, add - are not in it)" % tuple(tot))
