#!/bin/bash
# Sweep XVEC_PAIR_SHARE (first-placed block's share of a CU pair's rows) on one GPU box.
#   bash profiles/sweep_share.sh "<bench args>" v1 v2 ...
args=$1; shift
for rep in 1 2; do
for v in "$@"; do
  XVEC_PAIR_SHARE=$v python bench.py --steps 30 --warmup 5 --cpu-budget 0 $args 2>/dev/null | python -c "
import sys, json
d = json.loads(sys.stdin.readline())
k = d['roofline']['per_kernel_ms']
print('share=$v', d['ms_per_step'], ' '.join(f'{n}={v:.4f}' for n, v in k.items()))"
done
done
