#!/bin/bash
# Run ON the GPU box: SQ counters of the conv4 tile kernel (edgeblock_bwd_kernel<.,8,44>) per PHASE, by difference of its timing-only
# ablation builds (SVNET_BWD_MODE=2: return after phase A, 3: return after phase B, 0: the product) - three counter passes per mode,
# each its own run with --kernel-trace + --pmc only.  Output: gpurun_out/$TAG_mode{0,2,3}_{a,b,c}_summary.csv
# usage: TAG=r04_tile bash tools/tile_pmc.sh
TAG=${TAG:-r04_tile}
ROOT=${GRAFT_REPO_ROOT:-$(pwd)}
OUT=$ROOT/gpurun_out
cd /tmp && export TMPDIR=/tmp
for m in 0 2 3; do
  export SVNET_BWD_MODE=$m
  rocprofv3 --kernel-trace --pmc SQ_WAVES SQ_BUSY_CYCLES SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_VALU SQ_WAIT_INST_LDS -d $OUT/${TAG}_mode${m}_a -o run --output-format csv -- python3 $ROOT/tools/one_conv4.py > $OUT/${TAG}_mode${m}_a.log 2>&1
  rocprofv3 --kernel-trace --pmc SQ_VALU_MFMA_BUSY_CYCLES SQ_INSTS_VALU SQ_INSTS_MFMA SQ_INSTS_LDS SQ_INSTS_VMEM_RD SQ_INSTS_SALU SQ_ACTIVE_INST_LDS SQ_ACTIVE_INST_VMEM -d $OUT/${TAG}_mode${m}_b -o run --output-format csv -- python3 $ROOT/tools/one_conv4.py > $OUT/${TAG}_mode${m}_b.log 2>&1
  rocprofv3 --kernel-trace --pmc SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_INST_LEVEL_VMEM SQ_INST_LEVEL_LDS SQ_INST_CYCLES_VMEM_RD SQ_INST_CYCLES_VMEM_WR SQ_INSTS_VMEM_WR SQ_ACTIVE_INST_SCA -d $OUT/${TAG}_mode${m}_c -o run --output-format csv -- python3 $ROOT/tools/one_conv4.py > $OUT/${TAG}_mode${m}_c.log 2>&1
  rocprofv3 --kernel-trace --stats -d $OUT/${TAG}_mode${m}_t -o run --output-format csv -- python3 $ROOT/tools/one_conv4.py > $OUT/${TAG}_mode${m}_t.log 2>&1
done
cd $ROOT
for m in 0 2 3; do
  for p in a b c; do python3 tools/pmc_summary.py $OUT/${TAG}_mode${m}_$p | grep "^kernel\|edgeblock_bwd_kernel" > $OUT/${TAG}_mode${m}_${p}_summary.csv; rm -rf $OUT/${TAG}_mode${m}_$p; done
  grep "edgeblock_bwd_kernel" $(find $OUT/${TAG}_mode${m}_t -name "*kernel_stats.csv" | head -1) > $OUT/${TAG}_mode${m}_time.csv; rm -rf $OUT/${TAG}_mode${m}_t
  echo "mode $m"; cat $OUT/${TAG}_mode${m}_time.csv; cat $OUT/${TAG}_mode${m}_?_summary.csv
done
