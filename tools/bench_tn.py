"""Times the ternary weight-gradient product GX = x_b^T . dn of the fused edge block alone (diagnostic)."""
import os, sys, torch
root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, root)
from svnet_amd import _ops
E = 32 * 1024 * 20
for (Cs, Cv, Os) in [(64, 21, 128), (32, 10, 64), (32, 10, 32)]:
    dn = torch.randn(E, Os, device="cuda")
    xs = torch.randint(-2**62, 2**62, (E // 64, 320), dtype=torch.int64, device="cuda")
    xz = torch.randint(-2**62, 2**62, (E // 64, 320), dtype=torch.int64, device="cuda")
    GX = torch.zeros(Os, 320, device="cuda")
    used = 0
    for ct in range(10):
        if ((Cs if ct < 4 else 2 * Cv) > 32 * (ct & 1)):
            used |= 1 << ct
    def run():
        _ops.gemm(320, Os, E, a_planes=(xs, xz), B=dn, b_rs=Os, b_cs=1, C=GX, ldc=1, c_cs=320, accumulate=True, tern_tile_mask=used)
    for _ in range(3): run()
    torch.cuda.synchronize()
    a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    a.record()
    for _ in range(20): run()
    b.record(); torch.cuda.synchronize()
    print("Os=%d used=%x  %.1f us   target=%s dbg=%s" % (Os, used, a.elapsed_time(b) / 20 * 1e3, os.environ.get("SVNET_TN_TARGET"), os.environ.get("SVNET_TN_DBG")), flush=True)
