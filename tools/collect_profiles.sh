#!/bin/bash
# Run ON the GPU box (via gpurun): rocprofv3 kernel-trace + stats of the default bench command, then the PMC passes
# (FETCH_SIZE and WRITE_SIZE in separate passes, --kernel-trace only, as MI355X_MICROARCH.md prescribes), all into gpurun_out/$1_*.
# usage: bash tools/collect_profiles.sh r02_v1
set -e
TAG=${1:-r02}
ROOT=${GRAFT_REPO_ROOT:-$(pwd)}
OUT=$ROOT/gpurun_out
cd /tmp && export TMPDIR=/tmp
rocprofv3 --kernel-trace --stats -d $OUT/${TAG}_trace -o run --output-format csv -- python3 $ROOT/bench.py --no-cpu-baseline --no-extras --steps 10 > $OUT/${TAG}_bench_traced.log 2>&1
for c in FETCH_SIZE WRITE_SIZE; do
  rocprofv3 --kernel-trace --pmc $c -d $OUT/${TAG}_pmc_$c -o run --output-format csv -- python3 $ROOT/bench.py --steps 2 --warmup 1 --no-graph --no-cpu-baseline > $OUT/${TAG}_pmc_$c.log 2>&1
done
rocprofv3 --kernel-trace --pmc SQ_WAVES SQ_BUSY_CYCLES SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_SCA SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_WAVE_CYCLES -d $OUT/${TAG}_pmc_sq -o run --output-format csv -- python3 $ROOT/bench.py --steps 2 --warmup 1 --no-graph --no-cpu-baseline > $OUT/${TAG}_pmc_sq.log 2>&1 || echo "SQ pass failed"
cd $ROOT
python3 tools/step_trace.py $(find $OUT/${TAG}_trace -name "*kernel_trace.csv" | head -1) > $OUT/${TAG}_step_kernels.txt
cp $(find $OUT/${TAG}_trace -name "*kernel_stats.csv" | head -1) $OUT/${TAG}_kernel_stats.csv
python3 tools/pmc_summary.py $OUT/${TAG}_pmc_FETCH_SIZE $OUT/${TAG}_pmc_WRITE_SIZE > $OUT/${TAG}_pmc_fetch_write_summary.csv
python3 tools/pmc_summary.py $OUT/${TAG}_pmc_sq > $OUT/${TAG}_pmc_sq_summary.csv || true
python3 bench.py > $OUT/${TAG}_bench_default.log 2>&1
tail -1 $OUT/${TAG}_bench_default.log | cut -c1-400
head -12 $OUT/${TAG}_pmc_fetch_write_summary.csv
head -30 $OUT/${TAG}_step_kernels.txt
