"""Runs the fused fwd+bwd of one conv4-shaped edge layer (Cs=64 Cv=21 -> Os=128 Ov=42, B=32 N=1024 k=20) a few times: the profiling
target of tools/tile_pmc.sh (SVNET_BWD_MODE = 0 product, 2 return after phase A, 3 return after phase B)."""
import contextlib, io, os, sys, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from svnet_amd import config
from svnet_amd.models.sv_layers import SVBlock
from svnet_amd.models.utils.sv_util import get_graph_feature_sv, svpool
config.FUSE_EDGE_BLOCKS = True
Cs, Cv, Os, Ov = 64, 21, 128, 42
with contextlib.redirect_stdout(io.StringIO()):
    blk = SVBlock((2 * Cs, 2 * Cv), (Os, Ov), binary=True).cuda().train()
torch.manual_seed(0)
s = torch.randn(32, 1024, Cs, device="cuda", requires_grad=True)
v = torch.randn(32, 1024, 3, Cv, device="cuda", requires_grad=True)
for _ in range(3):
    so, vo = svpool(blk(get_graph_feature_sv((s, v), k=20)))
    (so.sum() + vo.sum()).backward()
torch.cuda.synchronize()
print("done")
