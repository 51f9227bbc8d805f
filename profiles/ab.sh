#!/bin/bash
# A/B two builds of libxvec_hip.so on ONE GPU box (devices differ by several %): alternate runs.
#   bash profiles/ab.sh <libA.so> <libB.so> [extra bench args]      (A/B builds live under build/ab/, never in the package)
A=$1; B=$2; shift 2
for i in 1 2 3; do
  for L in "$A" "$B"; do
    XVEC_LIB=$(realpath $L) python bench.py --steps 30 --warmup 5 --cpu-budget 0 --no-secondary "$@" 2>/dev/null | python -c "
import sys, json
d = json.loads(sys.stdin.readline())
k = d['roofline']['per_kernel_ms']
print('$L'.split('/')[-1], d['value'], d['ms_per_step'], ' '.join(f'{n}={v:.4f}' for n, v in k.items()))"
  done
done
