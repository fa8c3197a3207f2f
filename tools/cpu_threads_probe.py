#!/usr/bin/env python3
"""How many host threads should the CPU oracle use on a GPU box?  Prints the CPU share this process really has (affinity, cgroup quota)
and the oracle's fwd+loss+bwd time of sv_dgcnn_cls --binary B=8 at several torch thread counts.  Diagnostic."""
import os
import sys
import time

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import torch  # noqa: E402
import bench  # noqa: E402

print("cpu_count", os.cpu_count(), "affinity", len(os.sched_getaffinity(0)), "torch threads", torch.get_num_threads())
for f in ("/sys/fs/cgroup/cpu.max", "/sys/fs/cgroup/cpu/cpu.cfs_quota_us", "/sys/fs/cgroup/cpu/cpu.cfs_period_us"):
    try:
        print(f, open(f).read().strip())
    except OSError:
        pass
print("loadavg", open("/proc/loadavg").read().strip())
wl = bench.WORKLOADS["dgcnn_cls"]
for n in [int(a) for a in (sys.argv[1] if len(sys.argv) > 1 else "128,64,32,16,8").split(",")]:
    t0 = time.time()
    r = bench.cpu_baseline(wl, sample_b=8, timed=1, threads=n)
    print("threads %3d: fwd+loss+bwd %.4f clouds/s, forward only %.4f clouds/s (leg %.1f s)" % (n, r["value"], r["forward_only_value"], time.time() - t0), flush=True)
