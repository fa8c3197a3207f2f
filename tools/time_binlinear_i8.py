"""svnet_binlinear_i8_fwd_f32 alone at conv5's shape ([32768 x 505] x [505 x 512]) and at the PointNet layer's ([32768 x 2044] x [2044 x 512]):
with / without the saved planes and the column sums (diagnostic)."""
import os, sys, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from svnet_amd import _lib, _ops
from svnet_amd._ops import _p, _stream, call
dev = "cuda"
for (M, K, O) in ((32768, 505, 512), (32768, 2044, 512), (65536, 2144, 256)):
    g = torch.Generator(device=dev).manual_seed(1)
    x = torch.randn(M, K, device=dev, generator=g)
    W = torch.randn(O, K, device=dev, generator=g)
    beta = torch.randn(K, device=dev, generator=g) * 0.1
    sc = torch.rand(O, device=dev, generator=g) + 0.5
    w8 = torch.empty((_lib.lib().svnet_binweight_i8_bytes(O, K),), dtype=torch.int8, device=dev)
    call("svnet_binweight_pack_i8", _p(W), O, K, _p(w8), _stream())
    y = torch.empty(M, O, device=dev)
    pl = [torch.empty(((M + 63) // 64, K), dtype=torch.int64, device=dev) for _ in range(3)]
    sums = torch.zeros(_ops._sliced_len(2 * O), dtype=torch.float64, device=dev)
    for name, planes, cs in (("planes+sums", pl, sums), ("planes", pl, None), ("sums", [None] * 3, sums), ("bare", [None] * 3, None)):
        ts = []
        for it in range(6):
            a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            a.record()
            call("svnet_binlinear_i8_fwd_f32", _p(x), K, _p(beta), _p(w8), _p(sc), None, M, K, O, _p(y), _p(planes[0]), _p(planes[1]), _p(planes[2]), _p(cs), _stream())
            b.record()
            torch.cuda.synchronize()
            ts.append(a.elapsed_time(b) * 1e3)
        t = min(ts[1:])
        print("M=%d K=%d O=%d %-12s %7.1f us  %6.1f TOP/s (%.1f %% of the int8 peak)  %5.2f TB/s of x + y" % (
            M, K, O, name, t, 2.0 * M * K * O / t / 1e6, 2.0 * M * K * O / t / 1e6 / 50.0, (M * K * 4 + M * O * 4) / t / 1e6), flush=True)
