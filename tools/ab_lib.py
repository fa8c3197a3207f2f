"""A/B of whole library builds on the same GPU box: python tools/ab_lib.py [--reps R] NAME=path/to/lib.so ... runs bench.py once per
library (child processes, alternating R times; SVNET_DIAG_LIB selects the library) and prints ms_per_step per run and the medians.
`NAME=default` is the in-tree build.  Variant libraries: make -C svnet_amd/csrc BUILD=_build_x OUT=../../_ab/libx.so EXTRA=-D...
Diagnostic (boxes differ by +-1.5 %: only same-box, alternating runs rank two builds)."""
import json, os, statistics, subprocess, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
args = sys.argv[1:]
reps = 3
if args and args[0] == "--reps":
    reps = int(args[1]); args = args[2:]
extra = os.environ.get("AB_ARGS", "").split()
res = {}
for rep in range(reps):
    for a in args:
        name, path = a.split("=")
        env = dict(os.environ)
        if path != "default":
            env["SVNET_DIAG_LIB"] = os.path.join(ROOT, path)
        out = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--no-cpu-baseline", "--steps", "30"] + extra, capture_output=True, text=True, env=env)
        line = [l for l in out.stdout.splitlines() if l.startswith("{")]
        ms = json.loads(line[-1])["ms_per_step"] if line else None
        res.setdefault(name, []).append(ms)
        print(name, ms if ms is not None else out.stderr[-400:], flush=True)
for name, v in res.items():
    v = [x for x in v if x is not None]
    if v:
        print("median %-12s %.4f  (min %.4f, n=%d)" % (name, statistics.median(v), min(v), len(v)), flush=True)
