#!/bin/bash
# PMC passes for the bf16 bench (both kernels): bash profiles/run_pmc_bf16.sh <tag>
tag=${1:-bf16}
cd /tmp && export TMPDIR=/tmp
cd "$GRAFT_REPO_ROOT"
for mode in big small; do
  if [ $mode = small ]; then export XVEC_BF16_SMALL=1; else unset XVEC_BF16_SMALL; fi
  out=gpurun_out/pmc_${tag}_$mode
  mkdir -p $out
  i=0
  for grp in "SQ_VALU_MFMA_BUSY_CYCLES SQ_BUSY_CYCLES SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY GRBM_GUI_ACTIVE" \
             "SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_INSTS_LDS SQ_ACTIVE_INST_LDS SQ_WAIT_INST_LDS SQ_LDS_UNALIGNED_STALL GRBM_GUI_ACTIVE" \
             "FETCH_SIZE" "WRITE_SIZE" "TCC_HIT_sum TCC_MISS_sum"; do
    i=$((i+1))
    rocprofv3 --pmc $grp --output-format csv -d $out/p$i -o p$i -- python3 bench.py --steps 3 --warmup 1 --cpu-budget 0 --dtype bf16 > $out/p$i.log 2>&1
    echo "$mode pass $i: exit $?"
  done
done
