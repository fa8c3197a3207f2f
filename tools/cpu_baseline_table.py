#!/usr/bin/env python3
"""The CPU baseline of bench.py (the oracle, kind "port") at B in {4, 8, 16, 32} on this box's host cores: is the per-cloud rate flat
in B, and what does BASELINE.md §3's protocol (the same config, B = 32, 1 warm-up + 3 timed) cost?  VERDICT r4 weak #8.

    python tools/cpu_baseline_table.py [--timed 3] > profiles/r05_cpu_baseline_table.txt
"""
import argparse
import os
import sys
import time

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import bench  # noqa: E402


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--timed", type=int, default=3)
    ap.add_argument("--batches", default="4,8,16,32")
    args = ap.parse_args()
    wl = bench.WORKLOADS["dgcnn_cls"]
    print("# oracle (torch CPU ops) of %s, N=%d k=%d; median of %d after 1 warm-up" % (wl["name"], wl["N"], wl["k"], args.timed))
    print("# threads = bench.cpu_share() = %d" % bench.cpu_share())
    print("# B  fwd+loss+bwd clouds/s  forward-only clouds/s  wall s of the leg  threads  cpu")
    for b in [int(t) for t in args.batches.split(",")]:
        t0 = time.time()
        r = bench.cpu_baseline(wl, sample_b=b, timed=args.timed)
        print("%3d  %8.4f  %8.4f  %7.1f  %d  %s (%d physical cores)" % (b, r["value"], r["forward_only_value"], time.time() - t0, r["cores"],
                                                                      r["cpu_model"], r["physical_cores"]))
        sys.stdout.flush()


if __name__ == "__main__":
    main()
