#!/bin/bash
# PMC passes for the bf16 path (tdnn_pp kernels): bash profiles/run_pmc_pp.sh <tag> [groups...]
tag=${1:-pp}
cd /tmp && export TMPDIR=/tmp
cd "$GRAFT_REPO_ROOT"
out=gpurun_out/pmc_${tag}
mkdir -p $out
i=0
for grp in "SQ_VALU_MFMA_BUSY_CYCLES SQ_BUSY_CYCLES SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY GRBM_GUI_ACTIVE" \
           "SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_INSTS_LDS SQ_ACTIVE_INST_LDS SQ_WAIT_INST_LDS SQ_LDS_UNALIGNED_STALL SQ_LDS_ADDR_CONFLICT GRBM_GUI_ACTIVE" \
           "SQ_INSTS_VMEM SQ_ACTIVE_INST_VMEM SQ_INSTS_VALU SQ_ACTIVE_INST_VALU SQ_INSTS_SALU SQ_ACTIVE_INST_SCA SQ_ACTIVE_INST_MISC GRBM_GUI_ACTIVE" \
           "FETCH_SIZE" "WRITE_SIZE" "TCC_HIT_sum TCC_MISS_sum"; do
  i=$((i+1))
  if [ -n "$PASSES" ] && ! echo " $PASSES " | grep -q " $i "; then continue; fi
  rocprofv3 --pmc $grp --output-format csv -d $out/p$i -o p$i -- python3 bench.py --steps 3 --warmup 1 --cpu-budget 0 --dtype bf16 > $out/p$i.log 2>&1
  echo "pass $i: exit $?"
done
python3 profiles/summarize_pmc.py $out $out/traffic.json > $out/summary.txt 2>&1
