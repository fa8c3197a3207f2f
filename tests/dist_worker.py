"""world-size-2 gloo worker for tests/test_host.py (data-parallel gradient bucket)."""
import sys

import torch
import torch.distributed as dist

from svnet_amd.dist import GradBucket


def main():
    rank, world = int(sys.argv[1]), int(sys.argv[2])
    dist.init_process_group("gloo", rank=rank, world_size=world)
    torch.manual_seed(0)
    net = torch.nn.Sequential(torch.nn.Linear(5, 7), torch.nn.BatchNorm1d(7), torch.nn.Linear(7, 3))
    bucket = GradBucket(net.parameters())
    x = torch.randn(8, 5, generator=torch.Generator().manual_seed(100 + rank))
    net(x).pow(2).mean().backward()
    for p in net.parameters():
        assert p.grad.data_ptr() >= bucket.flat.data_ptr()            # views into the flat buffer
    local = bucket.flat.clone()
    bucket.all_reduce_mean()
    gathered = [torch.empty_like(local) for _ in range(world)]
    dist.all_gather(gathered, local)
    want = sum(gathered) / world
    assert torch.allclose(bucket.flat, want, atol=1e-7), (bucket.flat - want).abs().max()
    assert torch.allclose(torch.cat([p.grad.reshape(-1) for p in net.parameters()]), want, atol=1e-7)
    bucket.zero()
    assert float(bucket.flat.abs().sum()) == 0.0
    # assign-then-pack mode gives the same flat gradient as zero-then-accumulate
    bucket.begin()
    net(x).pow(2).mean().backward()
    bucket.pack()
    assert torch.allclose(bucket.flat, local, atol=1e-7)
    for p in net.parameters():
        assert p.grad.data_ptr() >= bucket.flat.data_ptr()
    dist.barrier()
    dist.destroy_process_group()
    print("OK rank", rank)


if __name__ == "__main__":
    main()
