#!/bin/bash
# A/B one environment knob on ONE GPU box: bash profiles/ab_env.sh VAR=VALUE [bench args]
KV=$1; shift
for i in 1 2 3; do
  for mode in base knob; do
    if [ $mode = knob ]; then export $KV; else unset ${KV%%=*}; fi
    python bench.py --steps 30 --warmup 5 --cpu-budget 0 --no-secondary "$@" 2>/dev/null | python -c "
import sys, json
d = json.loads(sys.stdin.readline())
k = d['roofline']['per_kernel_ms']
print('$mode', d['value'], d['ms_per_step'], ' '.join(f'{n}={v:.4f}' for n, v in k.items()))"
  done
done
