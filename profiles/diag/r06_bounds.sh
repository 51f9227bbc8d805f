#!/bin/bash
# Round 6, VERDICT r05 item 1: the three bounds on ONE box (timing only).
#   (a) profiles/diag/src/onewave_bound.hip (one wave per SIMD, 256 AGPR accumulators) next to the shipped K loop with the
#       same things knocked out (XVEC_KNOCK 5 = no DMA, no epilogue; 4 = no epilogue; 7 = MFMA only; 1 = no DMA)
#   (b) cost side of staggered frame halves on the K = 512 layers: XVEC_KNOCK 16384 (every weight piece requested twice)
#   (c) layers 4 + 5 as one launch: XVEC_KNOCK 32768 (layer 4 without its store epilogue) + 65536 (layer 5 without its
#       activation pieces)
# usage (GPU box): bash profiles/diag/r06_bounds.sh > gpurun_out/r06_bounds.txt
set -e
cd "$(dirname "$0")/../.."
D=build/diag
echo "== (a) one-wave microkernel, real-statistics operands"; $D/onewave_bound rand 27 4
echo "== (a) one-wave microkernel, zero operands"; $D/onewave_bound zero 27 2
for r in 1 2 3; do
  echo "== knock-out round $r"
  python3 profiles/diag/pp_knock.py
  for k in 1 4 5 7 16384 32768 65536; do
    XVEC_LIB=$PWD/$D/libxvec_hip_knock$k.so python3 profiles/diag/pp_knock.py
  done
done
echo "== (a) one-wave microkernel again (clock drift check)"; $D/onewave_bound rand 27 2
