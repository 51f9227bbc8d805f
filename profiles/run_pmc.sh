#!/bin/bash
# Hardware counters of the bench command, one rocprofv3 --pmc pass per counter group
# (counters only: no tracing domains alongside).   bash profiles/run_pmc.sh <tag>
tag=${1:-r01}
cd /tmp && export TMPDIR=/tmp
cd "$GRAFT_REPO_ROOT"
out=gpurun_out/pmc_$tag
mkdir -p $out
i=0
for grp in "SQ_VALU_MFMA_BUSY_CYCLES SQ_BUSY_CYCLES SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY GRBM_GUI_ACTIVE" \
           "SQ_INSTS_VALU_MFMA_MOPS_F32 SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_INSTS_VALU SQ_INSTS_LDS SQ_ACTIVE_INST_LDS GRBM_GUI_ACTIVE" \
           "FETCH_SIZE" "WRITE_SIZE" "TCC_HIT_sum TCC_MISS_sum"; do
  i=$((i+1))
  rocprofv3 --pmc $grp --output-format csv -d $out/p$i -o p$i -- python3 bench.py --steps 3 --warmup 1 --cpu-budget 0 > $out/p$i.log 2>&1
  echo "pass $i ($grp): exit $?"
done
find $out -name "*counter_collection.csv" | head
