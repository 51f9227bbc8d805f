#!/bin/bash
# FETCH_SIZE / WRITE_SIZE only (two counter passes) of the bench command.   bash profiles/run_fetch.sh <tag> [bench args]
tag=${1:-r01}; shift
cd /tmp && export TMPDIR=/tmp
cd "$GRAFT_REPO_ROOT"
out=gpurun_out/fetch_$tag
mkdir -p $out
i=0
for grp in "FETCH_SIZE" "WRITE_SIZE"; do
  i=$((i+1))
  rocprofv3 --pmc $grp --output-format csv -d $out/p$i -o p$i -- python3 bench.py --steps 3 --warmup 1 --cpu-budget 0 "$@" > $out/p$i.log 2>&1
  echo "pass $i ($grp): exit $?"
done
python3 profiles/summarize_pmc.py $out > $out/summary.txt 2>&1
grep -A3 "tdnn_kernel" $out/summary.txt | grep -E "tdnn_kernel|FETCH|WRITE"
