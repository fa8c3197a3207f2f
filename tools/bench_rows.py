"""Diagnostic: one rows-GEMM shape (default: conv5's dx product, [32768 x 512] x [512 x 505] against sign weights, with the per-k scale,
the STE mask and the column sums the step passes), timed alone with HIP events.  SVNET_ROWS_NO_DB=1 selects the unpipelined kernel.
    python tools/bench_rows.py [M K N] [--plain] [--reps R]"""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from svnet_amd import _ops

args = [a for a in sys.argv[1:] if not a.startswith("--")]
M, K, N = (int(a) for a in args[:3]) if len(args) >= 3 else (32768, 512, 505)
plain = "--plain" in sys.argv
reps = int(sys.argv[sys.argv.index("--reps") + 1]) if "--reps" in sys.argv else 20
dev = torch.device("cuda:0")
g = torch.Generator(device=dev).manual_seed(1)
A = torch.randn(M, K, device=dev, generator=g)
Wb = torch.sign(torch.randn(K, N, device=dev, generator=g))
asc = torch.rand(K, device=dev, generator=g) + 0.5
mask = torch.randint(-2**62, 2**62, ((M + 63) // 64, N), device=dev, dtype=torch.int64, generator=g)
C = torch.empty(M, N, device=dev)
cs = torch.zeros(_ops._sliced_len(N), device=dev)        # (sliced accumulator)


def run():
    if plain:
        _ops.gemm(M, N, K, A=A, a_rs=K, a_cs=1, B=Wb, b_rs=N, b_cs=1, b_exact=True, C=C, ldc=N)
    else:
        _ops.gemm(M, N, K, A=A, a_rs=K, a_cs=1, a_scale=asc, B=Wb, b_rs=N, b_cs=1, b_exact=True, C=C, ldc=N, mask=mask, col_sum=cs)


for _ in range(3):
    run()
torch.cuda.synchronize()
ev = [(torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)) for _ in range(reps)]
for a, b in ev:
    a.record()
    run()
    b.record()
torch.cuda.synchronize()
ts = sorted(a.elapsed_time(b) * 1e3 for a, b in ev)
fl = 2.0 * M * N * K
print("rows GEMM [%d x %d] x [%d x %d]%s: median %.1f us, min %.1f us (incl. the B pack launch); %.1f TFLOP/s fp32-equivalent, %.1f bf16 (3 passes)"
      % (M, K, K, N, " plain" if plain else "", ts[len(ts) // 2], ts[0], fl / ts[len(ts) // 2] / 1e6, 3 * fl / ts[len(ts) // 2] / 1e6))
ref = (A[:256].double() * asc.double()) @ Wb.double() if not plain else A[:256].double() @ Wb.double()
got = C[:256].double()
if not plain:
    bits = ((mask[:4].unsqueeze(1) >> torch.arange(64, device=dev).view(1, 64, 1)) & 1).reshape(256, N).double()
    ref = ref * bits
print("max rel err of the first 256 rows: %.2e" % float((got - ref).abs().max() / ref.abs().max()))
