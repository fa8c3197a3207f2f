#!/usr/bin/env python3
"""Headline benchmark: point-clouds/sec, forward (train mode) + cal_loss + backward (+ gradient all-reduce for
N > 1) of sv_dgcnn_cls --binary, B=32 per GPU, N=1024, k=20, synthetic clouds resident in HBM.

    python bench.py [--gpus N --steps K --warmup W]
    python -m torch.distributed.run --nnodes=1 --nproc-per-node N ... bench.py --gpus N --steps K --warmup W

Prints ONE JSON line on rank 0 (contract in the task statement).
"""
import argparse
import contextlib
import io
import json
import os
import sys
import time

import torch

ROOT = os.path.dirname(os.path.abspath(__file__))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)

HBM_PEAK_GBS = 8000.0          # MI355X HBM3E spec peak (MI355X_MICROARCH.md)
B_PER_GPU, N_POINTS, K_NN = 32, 1024, 20


def build_model(dev):
    import svnet_amd.models as M
    torch.manual_seed(0)
    with contextlib.redirect_stdout(io.StringIO()):
        model = M.SV_DGCNN_CLS(argparse.Namespace(k=K_NN, binary=True), 40)
    return model.to(dev).train()


def cpu_baseline(sample_b=8, timed=2):
    """The oracle (CPU restatement of the reference, kind "port") timed on this box's host cores on a bounded
    sample of the same workload: fwd + cal_loss + bwd of sv_dgcnn_cls --binary at N=1024, k=20, B=sample_b."""
    from svnet_amd import synth
    from oracle import params as oparams, sv_ref
    cores = torch.get_num_threads()
    P = oparams.synthetic_params("sv_dgcnn_cls", binary=True, seed=1234, requires_grad=True)
    x = torch.from_numpy(synth.cloud_batch(1234, 0, 0, sample_b, N_POINTS))
    y = torch.from_numpy(synth.class_labels(1234, 0, 0, sample_b))
    times = []
    for it in range(1 + timed):
        for p in P.values():
            p.grad = None
        t0 = time.perf_counter()
        ctx = sv_ref.Ctx(train=True, knn="torch")
        loss = sv_ref.cal_loss(sv_ref.sv_dgcnn_cls(x, P, K_NN, True, ctx), y)
        loss.backward()
        dt = time.perf_counter() - t0
        if it > 0:
            times.append(dt)
    best = min(times)
    return {"value": round(sample_b / best, 4), "unit": "point-clouds/sec", "cores": cores, "kind": "port",
            "sample": "oracle fwd+loss+bwd, sv_dgcnn_cls binary, B=%d N=%d k=%d, best of %d after 1 warm-up (%.1f s each)"
                      % (sample_b, N_POINTS, K_NN, timed, best)}


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=20)
    ap.add_argument("--warmup", type=int, default=5)
    ap.add_argument("--no-graph", action="store_true", help="launch eagerly instead of replaying a HIP graph")
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--roofline-kernel", default="svnet_edgeblock_bwd_f32")
    args = ap.parse_args()

    import torch.distributed as dist
    world = int(os.environ.get("WORLD_SIZE", "1"))
    rank = int(os.environ.get("RANK", "0"))
    local = int(os.environ.get("LOCAL_RANK", "0"))
    if world > 1:
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        dist.init_process_group("nccl", rank=rank, world_size=world, device_id=torch.device("cuda", local))
    torch.cuda.set_device(local)
    dev = torch.device("cuda", local)

    from svnet_amd import _lib, synth
    from svnet_amd.dist import GradBucket
    from svnet_amd.train import cal_loss
    _lib.lib()                                                        # fail loudly if the HIP library is missing

    model = build_model(dev)
    bucket = GradBucket(model.parameters())
    x = torch.from_numpy(synth.cloud_batch(1234, 0, rank, B_PER_GPU, N_POINTS)).to(dev)
    y = torch.from_numpy(synth.class_labels(1234, 0, rank, B_PER_GPU)).to(dev)

    def fwd_bwd():
        bucket.begin()
        loss = cal_loss(model(x), y)
        loss.backward()
        bucket.pack()                                                 # one batched copy of all gradients into the flat bucket
        return loss.detach()

    # a few eager steps first (allocator warm-up).  They run on the SAME stream the capture will use, and no autograd
    # graph of theirs is kept alive: AccumulateGrad nodes remembered from another stream make the capture fork into a
    # second stream, and the allocator then re-uses blocks (e.g. the saved neighbour ids) across the fork.
    for _ in range(2):
        loss = fwd_bwd()
        del loss
    torch.cuda.synchronize()

    graph = None
    if not args.no_graph:
        try:
            graph = torch.cuda.CUDAGraph()
            with torch.cuda.graph(graph):
                fwd_bwd()
        except Exception as e:                                        # capture is an optimisation, not a requirement
            if rank == 0:
                print("graph capture failed, running eagerly: %r" % (e,), file=sys.stderr)
            graph = None
            torch.cuda.synchronize()

    def step():
        if graph is not None:
            graph.replay()
        else:
            fwd_bwd()
        bucket.all_reduce_mean()

    for _ in range(args.warmup):
        step()

    # roofline leg: HIP events around one entry point (eager launches on the current stream, outside the timed region
    # when a graph is replayed, inside it otherwise)
    # (the fused backward issues two launches per layer: parts=1 vector path on a side stream, parts=2 the 32-edge tile kernel;
    #  the dominant kernel of the step is the tile kernel of conv4, Os = 128)
    sel = (lambda a: a[0]._obj.parts == 2 and a[0]._obj.Os == 128) if args.roofline_kernel == "svnet_edgeblock_bwd_f32" else None
    timer = _lib.KernelTimer(args.roofline_kernel, sel)

    def barrier():
        torch.cuda.synchronize()
        if world > 1:
            dist.barrier()
        torch.cuda.synchronize()

    barrier()
    if graph is None:
        _lib.TIMER = timer
    t0 = time.perf_counter()
    for _ in range(args.steps):
        step()
    barrier()
    elapsed = time.perf_counter() - t0
    _lib.TIMER = None
    if graph is not None:                                             # same kernels, launched eagerly, after the timed region
        _lib.TIMER = timer
        for _ in range(3):
            fwd_bwd()
        torch.cuda.synchronize()
        _lib.TIMER = None

    t = torch.tensor([elapsed], dtype=torch.float64, device=dev)
    if world > 1:
        dist.all_reduce(t, op=dist.ReduceOp.MAX)
    elapsed = float(t.item())

    if rank == 0:
        clouds = B_PER_GPU * world * args.steps
        # Dominant kernel of the step: edgeblock_bwd_kernel<0,8>, the 32-edge tile kernel of conv4's fused backward (one launch
        # per step).  Algorithmic HBM bytes of that launch: per edge the kept n (2 B x Os) and planes (120 B) and
        # the neighbour id are read, dL/dy (4 B x Os), the row-sliced sign/non-zero planes (80 B) and the message row
        # (Cs + 3 Cv + 9 floats) are written; per point the pooled-edge operands / v / zz are read and the centre sums written.
        ms = timer.elapsed_ms()
        per_launch = None
        if ms:
            E = B_PER_GPU * N_POINTS * K_NN
            P_ = B_PER_GPU * N_POINTS
            Cs, Cv, Os, Ov = 64, 21, 128, 42
            dur = sum(ms) / len(ms) * 1e-3
            reads = E * (2 * Os + 120 + 8) + P_ * (Os * (4 + 2) + 4 * (3 * Cv + 18))
            writes = E * (4 * Os + 80 + 4 * (Cs + 3 * Cv + 9)) + P_ * 4 * (Cs + 3 * Cv + 9)
            alg = reads + writes
            per_launch = {"bound": "hbm", "achieved": round(alg / dur / 1e9, 1), "peak": HBM_PEAK_GBS, "unit": "GB/s",
                          "frac": round(alg / dur / 1e9 / HBM_PEAK_GBS, 4),
                          # not measured live: rocprofv3 --pmc passes of this same command, 2 x FETCH_SIZE (gfx950 counts 64 B per
                          # 128-B request) + WRITE_SIZE, per launch (profiles/r01_v12_pmc_fetch_write_summary.csv)
                          "traffic": 2 * 147664.9e3 + 866907.0e3, "traffic_source": "profiles/r01_v12_pmc_fetch_write_summary.csv",
                          "kernel": "edgeblock_bwd_kernel<0,8> (conv4 backward tile kernel: Cs=64 Cv=21 -> Os=128 Ov=42)",
                          "avg_launch_us": round(dur * 1e6, 1), "algorithmic_bytes": alg,
                          "note": "instruction/latency-bound (64-lane waves carry 21..64 channels), not bandwidth-bound; see DESIGN.md"}
        out = {
            "metric": "point-clouds/sec fwd+bwd, sv_dgcnn_cls B=32 N=1024 k=20",
            "value": round(clouds / elapsed, 2), "unit": "point-clouds/sec", "n_gpus": world, "steps": args.steps,
            "warmup": args.warmup, "ms_per_step": round(elapsed / args.steps * 1e3, 3), "higher_is_better": True,
            "scaling": "weak", "vs_baseline": None, "dtype": "f32 (ternary bit-planes in the binarized layers)",
            "data": "synthetic",
            "config": {"workload": "sv_dgcnn_cls --binary fwd+loss+bwd, B=%d per GPU, N=%d, k=%d" % (B_PER_GPU, N_POINTS, K_NN),
                       "global_batch": B_PER_GPU * world, "parallelism": "dp%d" % world,
                       "launch": "hipGraph replay" if graph is not None else "eager"},
            "roofline": per_launch,
        }
        if world == 1 and not args.no_cpu_baseline:
            out["cpu_baseline"] = cpu_baseline()
        print(json.dumps(out))
    if world > 1:
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
