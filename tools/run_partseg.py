"""Full-size sanity run of the part-segmentation caller (sv_dgcnn_partseg --binary, N=2048, k=40): fwd + loss + bwd,
finite outputs/gradients and ms per step (diagnostic; the headline benchmark is bench.py)."""
import argparse, contextlib, io, os, sys, time
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from svnet_amd import synth
from svnet_amd.train import cal_loss
from svnet_amd.models.sv_dgcnn_partseg import SV_DGCNN_PSEG

ap = argparse.ArgumentParser()
ap.add_argument("--batch", type=int, default=8)
ap.add_argument("--steps", type=int, default=5)
a = ap.parse_args()
dev = torch.device("cuda", 0)
args = argparse.Namespace(k=40, binary=True, emb_dims=1024, dropout=0.5)
with contextlib.redirect_stdout(io.StringIO()):
    net = SV_DGCNN_PSEG(args, 50).to(dev).train()
B, N = a.batch, 2048
x = torch.from_numpy(synth.cloud_batch(1234, 0, 0, B, N)).to(dev)
lab = torch.zeros(B, 16, device=dev); lab[torch.arange(B), torch.arange(B) % 16] = 1.0
seg = (torch.arange(B * N, device=dev) % 50).view(B, N)
def step():
    for p in net.parameters():
        p.grad = None
    out = net(x, lab)                                  # [B, 50, N]
    loss = cal_loss(out.permute(0, 2, 1).reshape(-1, 50), seg.view(-1))
    loss.backward()
    return out, loss
out, loss = step()
torch.cuda.synchronize()
assert torch.isfinite(out).all() and torch.isfinite(loss)
bad = [n for n, p in net.named_parameters() if p.grad is None or not torch.isfinite(p.grad).all()]
assert not bad, bad
t0 = time.perf_counter()
for _ in range(a.steps):
    step()
torch.cuda.synchronize()
dt = (time.perf_counter() - t0) / a.steps
print("partseg B=%d N=2048 k=40: loss %.4f, %.1f ms/step eager, %.1f clouds/s" % (B, float(loss), dt * 1e3, B / dt))
