#!/bin/bash
# Run ON the GPU box: per-kernel durations (rocprofv3 kernel trace) of tools/knn_lab.py for the given variant names (_ab/libknn_NAME.so)
ROOT=${GRAFT_REPO_ROOT:-$(cd "$(dirname "$0")/.." && pwd)}
cd /tmp && export TMPDIR=/tmp
for v in "$@"; do
  rm -rf $ROOT/gpurun_out/kp_$v
  rocprofv3 --kernel-trace -d $ROOT/gpurun_out/kp_$v -o kp --output-format csv -- python3 $ROOT/tools/knn_lab.py $v=_ab/libknn_$v.so > /dev/null 2>&1
  echo "== $v"
  python3 - "$ROOT/gpurun_out/kp_$v" <<'PY'
import csv, glob, sys, collections
f = glob.glob(sys.argv[1] + "/**/*kernel_trace.csv", recursive=True)[0]
d = collections.OrderedDict()
for r in csv.DictReader(open(f)):
    d.setdefault(r["Kernel_Name"][:70], []).append((int(r["End_Timestamp"]) - int(r["Start_Timestamp"])) / 1e3)
for k, v in d.items():
    print("%-72s n=%3d" % (k, len(v)), " ".join("%.1f" % x for x in v[-30:]))
PY
done
