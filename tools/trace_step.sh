#!/bin/bash
# Run ON the GPU box: rocprofv3 kernel trace of the default bench command -> gpurun_out/$1_step_kernels.txt (per-kernel table of one step)
# and gpurun_out/$1_step_sequence.txt (launch order with durations and start times).  usage: bash tools/trace_step.sh TAG [bench args]
set -e
TAG=${1:-trace}; shift || true
ROOT=${GRAFT_REPO_ROOT:-$(pwd)}
OUT=$ROOT/gpurun_out
cd /tmp && export TMPDIR=/tmp
rocprofv3 --kernel-trace --stats -d $OUT/${TAG}_trace -o run --output-format csv -- python3 $ROOT/bench.py --no-cpu-baseline --no-extras --steps 10 "$@" > $OUT/${TAG}_bench_traced.log 2>&1
cd $ROOT
T=$(find $OUT/${TAG}_trace -name "*kernel_trace.csv" | head -1)
python3 tools/step_trace.py $T > $OUT/${TAG}_step_kernels.txt
python3 tools/step_trace.py $T x > $OUT/${TAG}_step_sequence.txt
cp $(find $OUT/${TAG}_trace -name "*kernel_stats.csv" | head -1) $OUT/${TAG}_kernel_stats.csv
rm -rf $OUT/${TAG}_trace
head -3 $OUT/${TAG}_step_kernels.txt
