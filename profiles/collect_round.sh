#!/bin/bash
# Copy the summaries of one `run_round.sh <tag>` run from gpurun_out/<tag>/ (scratch) into profiles/ (committed):
#   bash profiles/collect_round.sh r04
set -e
tag=${1:-r05}
cd "$(dirname "$0")/.."
src=gpurun_out/$tag
cp $src/bench_fp32.json profiles/${tag}_bench.json
for n in bf16 bf16x3 ragged ragged_bf16 job20k job20k_bf16 wave wave_bf16 collective1 fp32_preroll0; do
  cp $src/bench_$n.json profiles/${tag}_bench_$n.json
done
for dt in fp32 bf16 bf16x3; do
  cp $src/stats_$dt/${dt}_kernel_stats.csv profiles/${tag}_kernel_stats_$dt.csv
  python3 profiles/summarize_trace.py $src/stats_$dt 20 > profiles/${tag}_timed_region_$dt.txt
done
cp $src/pmc_summary_fp32.txt profiles/${tag}_pmc_summary.txt
cp $src/pmc_summary_bf16.txt profiles/${tag}_pmc_summary_bf16.txt
cp $src/pmc_summary_bf16x3.txt profiles/${tag}_pmc_summary_bf16x3.txt
cp $src/pmc_summary_next.txt profiles/${tag}_pmc_summary_next.txt
cp $src/traffic.json profiles/traffic.json
ls -la profiles/${tag}_* profiles/traffic.json
