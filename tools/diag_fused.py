"""Diagnostic: fused train step at growing sizes, eager, synchronising after every C-ABI call; progress to a log."""
import argparse, contextlib, io, os, sys, time
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from svnet_amd import _lib, synth
import svnet_amd.models as M
from svnet_amd.train import cal_loss

LOG = open(os.path.join("gpurun_out", "diag.log"), "a", buffering=1)
orig_call = _lib.call
def traced(name, *args):
    LOG.write("  call %s\n" % name)
    orig_call(name, *args)
    torch.cuda.synchronize()
    LOG.write("  done %s\n" % name)
import svnet_amd._ops as ops
ops.call = traced

def run(B, N, k):
    LOG.write("== B=%d N=%d k=%d\n" % (B, N, k))
    torch.manual_seed(0)
    with contextlib.redirect_stdout(io.StringIO()):
        m = M.SV_DGCNN_CLS(argparse.Namespace(k=k, binary=True), 40).cuda().train()
    x = torch.from_numpy(synth.cloud_batch(1, 0, 0, B, N)).cuda()
    y = torch.from_numpy(synth.class_labels(1, 0, 0, B)).cuda()
    loss = cal_loss(m(x), y)
    LOG.write(" fwd ok loss %f\n" % float(loss))
    loss.backward()
    torch.cuda.synchronize()
    LOG.write(" bwd ok\n")

for cfg in [(2, 256, 20), (2, 1024, 20), (8, 1024, 20), (32, 1024, 20)]:
    run(*cfg)
LOG.write("ALL OK\n")
