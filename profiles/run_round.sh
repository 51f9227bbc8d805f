#!/bin/bash
# Round profile set (run on the GPU box via gpurun):  bash profiles/run_round.sh <tag>
#   kernel-trace stats of the default bench line (fp32 + its secondary legs), of --dtype bf16 and of --dtype bf16x3,
#   FETCH_SIZE / WRITE_SIZE passes for both, utilisation counters for both.  Output: gpurun_out/<tag>/
tag=${1:-r05}
cd /tmp && export TMPDIR=/tmp
cd "$GRAFT_REPO_ROOT"
out=gpurun_out/$tag
mkdir -p $out
if [ -z "$ONLY_PMC" ]; then
python3 bench.py --steps 50 --warmup 10 > $out/bench_fp32.json 2> $out/bench_fp32.err; echo "bench fp32 exit $?"
python3 bench.py --steps 50 --warmup 10 --dtype bf16 --cpu-budget 0 > $out/bench_bf16.json 2> $out/bench_bf16.err; echo "bench bf16 exit $?"
python3 bench.py --steps 20 --warmup 5 --workload ragged --cpu-budget 0 > $out/bench_ragged.json 2> $out/bench_ragged.err; echo "bench ragged exit $?"
python3 bench.py --steps 20 --warmup 5 --workload ragged --dtype bf16 --cpu-budget 0 > $out/bench_ragged_bf16.json 2> $out/bench_ragged_bf16.err; echo "bench ragged bf16 exit $?"
python3 bench.py --steps 50 --warmup 10 --dtype bf16x3 --cpu-budget 0 > $out/bench_bf16x3.json 2> $out/bench_bf16x3.err; echo "bench bf16x3 exit $?"
python3 bench.py --steps 20 --warmup 5 --preroll 0 --cpu-budget 0 --no-secondary > $out/bench_fp32_preroll0.json 2> $out/bench_fp32_preroll0.err; echo "bench fp32 --preroll 0 (round-1 protocol) exit $?"
python3 bench.py --workload job --utterances 20000 --cpu-budget 0 > $out/bench_job20k.json 2> $out/bench_job20k.err; echo "bench job exit $?"
python3 bench.py --workload job --utterances 20000 --dtype bf16 --cpu-budget 0 > $out/bench_job20k_bf16.json 2> $out/bench_job20k_bf16.err; echo "bench job bf16 exit $?"
python3 bench.py --steps 20 --warmup 5 --workload wave --cpu-budget 0 > $out/bench_wave.json 2> $out/bench_wave.err; echo "bench wave exit $?"
python3 bench.py --steps 20 --warmup 5 --workload wave --dtype bf16 --cpu-budget 0 > $out/bench_wave_bf16.json 2> $out/bench_wave_bf16.err; echo "bench wave bf16 exit $?"
python3 bench.py --gpus 1 --force-collective --steps 20 --warmup 5 --cpu-budget 0 > $out/bench_collective1.json 2> $out/bench_collective1.err; echo "bench one-rank RCCL exit $?"
fi
for dt in fp32 bf16 bf16x3; do
  mkdir -p $out/pmc_$dt
  [ -z "$ONLY_PMC" ] && rocprofv3 --kernel-trace --stats --output-format csv -d $out/stats_$dt -o $dt -- python3 bench.py --steps 20 --warmup 5 --cpu-budget 0 --no-secondary --dtype $dt > $out/stats_$dt.log 2>&1
  echo "stats $dt exit $?"
  [ -z "$ONLY_PMC" ] && python3 profiles/summarize_trace.py $out/stats_$dt 20 > $out/timed_region_$dt.txt 2>&1
  i=0
  for grp in "FETCH_SIZE" "WRITE_SIZE" \
             "SQ_VALU_MFMA_BUSY_CYCLES SQ_BUSY_CYCLES SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY GRBM_GUI_ACTIVE" \
             "SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_INSTS_LDS SQ_ACTIVE_INST_LDS SQ_WAIT_INST_LDS SQ_LDS_UNALIGNED_STALL GRBM_GUI_ACTIVE" \
             "TCC_HIT_sum TCC_MISS_sum"; do
    i=$((i+1))
    rocprofv3 --pmc $grp --output-format csv -d $out/pmc_$dt/p$i -o p$i -- python3 bench.py --steps 3 --warmup 1 --cpu-budget 0 --no-secondary --dtype $dt > $out/pmc_$dt/p$i.log 2>&1
    echo "pmc $dt pass $i exit $?"
  done
  python3 profiles/summarize_pmc.py $out/pmc_$dt > $out/pmc_summary_$dt.txt 2>&1
done
# next rows N3 / N4: the MFCC kernel and the fp64 score GEMM (traffic + the MFCC kernel's LDS / issue counters)
mkdir -p $out/pmc_next
i=0
for grp in "FETCH_SIZE" "WRITE_SIZE" \
           "SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_INSTS_LDS SQ_ACTIVE_INST_LDS SQ_INSTS_VALU SQ_ACTIVE_INST_VALU SQ_INSTS_SALU GRBM_GUI_ACTIVE" \
           "SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_WAIT_INST_LDS SQ_VALU_MFMA_BUSY_CYCLES GRBM_GUI_ACTIVE"; do
  i=$((i+1))
  rocprofv3 --pmc $grp --output-format csv -d $out/pmc_next/p$i -o p$i -- python3 profiles/diag/next_rows_pmc.py > $out/pmc_next/p$i.log 2>&1
  echo "pmc next rows pass $i exit $?"
done
python3 profiles/summarize_pmc.py $out/pmc_next > $out/pmc_summary_next.txt 2>&1
python3 profiles/make_traffic.py fp32=$out/pmc_fp32 bf16=$out/pmc_bf16 bf16x3=$out/pmc_bf16x3 next_rows=$out/pmc_next > $out/traffic.json
find $out -name "*kernel_stats.csv" | head
