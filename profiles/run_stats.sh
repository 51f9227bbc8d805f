#!/bin/bash
# Per-kernel time of the bench command (run on the GPU box via gpurun):
#   bash profiles/run_stats.sh <tag>     -> gpurun_out/prof_<tag>/
tag=${1:-r01}
cd /tmp && export TMPDIR=/tmp
cd "$GRAFT_REPO_ROOT"
out=gpurun_out/prof_$tag
mkdir -p $out
rocprofv3 --kernel-trace --stats --output-format csv -d $out -o $tag -- python3 bench.py --steps 20 --warmup 5 --cpu-budget 0 > $out/bench.log 2>&1
echo "rocprofv3 exit $?"
find $out -name "*.csv" | head
