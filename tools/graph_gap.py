"""Per-node cost of a hipGraph of tiny dependent kernels on this box (diagnostic)."""
import torch, time
x = torch.zeros(64, device="cuda")
for n in (50, 200):
    g = torch.cuda.CUDAGraph()
    s = torch.cuda.Stream()
    with torch.cuda.stream(s):
        for _ in range(3): x.add_(1.0)
        torch.cuda.synchronize()
        with torch.cuda.graph(g):
            for _ in range(n): x.add_(1.0)
    torch.cuda.synchronize()
    for _ in range(3): g.replay()
    torch.cuda.synchronize()
    t = time.perf_counter()
    for _ in range(20): g.replay()
    torch.cuda.synchronize()
    print("graph of %d tiny kernels: %.2f us per kernel" % (n, (time.perf_counter() - t) / 20 / n * 1e6), flush=True)
t = time.perf_counter()
for _ in range(2000): x.add_(1.0)
torch.cuda.synchronize()
print("eager: %.2f us per kernel" % ((time.perf_counter() - t) / 2000 * 1e6))
