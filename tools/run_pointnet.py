"""Full-size sanity run of the PointNet-style caller (sv_pointnet_cls --binary, B=32, N=1024): fwd + loss + bwd, finite
gradients and ms per step (diagnostic)."""
import argparse, contextlib, io, os, sys, time
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from svnet_amd import synth
from svnet_amd.train import cal_loss
from svnet_amd.models.sv_pointnet_cls import SV_PointNet_CLS

dev = torch.device("cuda", 0)
args = argparse.Namespace(k=20, binary=True, emb_dims=1024, dropout=0.5)
with contextlib.redirect_stdout(io.StringIO()):
    net = SV_PointNet_CLS(args, 40).to(dev).train()
B, N = 32, 1024
x = torch.from_numpy(synth.cloud_batch(1234, 0, 0, B, N)).to(dev)
y = torch.from_numpy(synth.class_labels(1234, 0, 0, B)).to(dev)
def step():
    for p in net.parameters():
        p.grad = None
    loss = cal_loss(net(x), y)
    loss.backward()
    return loss
loss = step()
torch.cuda.synchronize()
bad = [n for n, p in net.named_parameters() if p.grad is None or not torch.isfinite(p.grad).all()]
assert torch.isfinite(loss) and not bad, bad
t0 = time.perf_counter()
for _ in range(5):
    step()
torch.cuda.synchronize()
dt = (time.perf_counter() - t0) / 5
print("sv_pointnet_cls B=%d N=%d: loss %.4f, %.1f ms/step eager, %.1f clouds/s" % (B, N, float(loss.detach()), dt * 1e3, B / dt))
