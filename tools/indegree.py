"""In-degree statistics of the kNN graphs the bench workload builds (diagnostic): hubs set the tail of the reverse-list kernels."""
import os, sys, torch
root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, root)
import bench
from svnet_amd import _ops, synth
dev = torch.device("cuda", 0)
model = bench.build_model(dev)
x = torch.from_numpy(synth.cloud_batch(1234, 0, 0, bench.B_PER_GPU, bench.N_POINTS)).to(dev)
orig = _ops.knn
def spy(xx, k):
    idx = orig(xx, k)
    B, N = idx.shape[0], idx.shape[1]
    deg = torch.zeros(B, N, dtype=torch.int64, device=idx.device)
    deg.scatter_add_(1, idx.reshape(B, -1), torch.ones(B, N * k, dtype=torch.int64, device=idx.device))
    d = deg.flatten().float()
    print("kNN on C=%d: in-degree mean %.1f  p50 %d  p99 %d  max %d  zero-degree %.1f%%" % (xx.shape[1], d.mean().item(), int(d.median().item()),
          int(d.kthvalue(int(d.numel() * 0.99)).values.item()), int(d.max().item()), 100.0 * (d == 0).float().mean().item()), flush=True)
    return idx
_ops.knn = spy
import svnet_amd.models.utils.sv_util as U
for name in dir(U):
    if getattr(U, name) is orig: setattr(U, name, spy)
with torch.no_grad():
    model(x)
