"""Start times of the C-ABI launches of one captured train step WITHOUT a profiler (svnet_amd._lib.StepClock): every launch is
preceded by a one-thread kernel that stores the device's 100 MHz clock, on the launch's own stream, so the table shows when each
stream reached each launch in a real replay (rocprofv3's kernel trace adds ~10 us per launch and stretches the step by 25 %).
usage: python tools/step_clock.py [--workload dgcnn_cls|pointnet_bin|pointnet_fp|partseg] [--filter substr,substr] [--B 32]
Diagnostic; run on the GPU box."""
import argparse, contextlib, io, os, sys, statistics
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import torch
import bench
from svnet_amd import _lib
from svnet_amd.train import TrainStep

ap = argparse.ArgumentParser()
ap.add_argument("--workload", default="dgcnn_cls")
ap.add_argument("--filter", default="")
ap.add_argument("--B", type=int, default=32)
ap.add_argument("--replays", type=int, default=7)
a = ap.parse_args()
dev = torch.device("cuda:0")
wl, model, inputs, target, loss_fn = bench.build_workload(a.workload, dev, 0, a.B)
step = TrainStep(model.train(), inputs, target, loss_fn)
flt = [f for f in a.filter.split(",") if f]
clock = _lib.StepClock(dev, (lambda n: any(f in n for f in flt)) if flt else None)


def arm():
    _lib.CLOCK = clock


step.capture(before_capture=arm)   # (the stamps are armed after the eager warm-up: only the captured step carries them)
_lib.CLOCK = None
names_per_step = len(clock.names)
base = 0
runs = []
for r in range(a.replays):
    step.run(all_reduce=False)
    torch.cuda.synchronize()
    t = clock.buf[base:base + names_per_step].cpu().tolist()
    runs.append(t)
# median over replays of (t_i - t_0)
t0s = [min(t) for t in runs]
rel = [[(x - t0) / 100.0 for x in t] for t, t0 in zip(runs, t0s)]
med = [statistics.median(col) for col in zip(*rel)]
order = sorted(range(names_per_step), key=lambda i: med[i])
streams = {}
ms = []
e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
e0.record()
for _ in range(20):
    step.run(all_reduce=False)
e1.record()
torch.cuda.synchronize()
print("# %s B=%d: %d stamps per step, %.3f ms per replay WITH the stamps" % (a.workload, a.B, names_per_step, e0.elapsed_time(e1) / 20))
print("# start_us  +delta_on_stream  stream  entry point")
last = {}
for i in order:
    n, s = clock.names[i]
    sid = streams.setdefault(s, len(streams))
    d = med[i] - last.get(sid, med[i])
    last[sid] = med[i]
    print("%9.1f %9.1f  s%d  %s" % (med[i], d, sid, n))
