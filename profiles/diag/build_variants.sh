#!/bin/bash
# Diagnostic builds of the library next to the shipping one (not product; build/diag is git-ignored; delete it when done so it does not travel with every push):
#   libxvec_hip_diag.so     -DXVEC_DIAG      in-kernel s_memtime stamps (tdnn_layer.hip, tdnn_pp16.hip)
#   libxvec_hip_knock<m>.so -DXVEC_KNOCK=<m> timing-only knock-outs of tdnn_pp16.hip (bit 0 DMA, 1 LDS reads, 2 epilogue)
set -e
cd "$(dirname "$0")/../../speaker-recognition-x-vectors_amd/csrc"
out=../../build/diag
FLAGS="-O3 -std=c++17 -fPIC --offload-arch=gfx950 -fno-slp-vectorize -Wno-unused-function -Wno-pass-failed -Wno-inline-asm"
rm -rf /tmp/xv_base /tmp/xv_diag; mkdir -p $out /tmp/xv_base /tmp/xv_diag
for f in tdnn_layer tdnn_first pool affine pack mfcc score xvec_api; do
  /opt/rocm/bin/hipcc $FLAGS -c $f.hip -o /tmp/xv_base/$f.o &
done
/opt/rocm/bin/hipcc $FLAGS -DXVEC_DIAG=1 -c tdnn_layer.hip -o /tmp/xv_diag/tdnn_layer.o &
/opt/rocm/bin/hipcc $FLAGS -DXVEC_DIAG=${DIAG:-1} -c tdnn_pp16.hip -o /tmp/xv_diag/tdnn_pp16.o &
for m in ${KNOCKS:-0 1 2 3 4 7}; do
  /opt/rocm/bin/hipcc $FLAGS -DXVEC_KNOCK=$m -c tdnn_pp16.hip -o /tmp/xv_base/knock_pp_$m.o &
done
wait
objs=$(ls /tmp/xv_base/*.o | grep -v knock_pp_ | grep -v tdnn_layer.o)
/opt/rocm/bin/hipcc --offload-arch=gfx950 -shared -fPIC -o $out/libxvec_hip_diag.so $objs /tmp/xv_diag/tdnn_layer.o /tmp/xv_diag/tdnn_pp16.o
for m in ${KNOCKS:-0 1 2 3 4 7}; do
  /opt/rocm/bin/hipcc --offload-arch=gfx950 -shared -fPIC -o $out/libxvec_hip_knock$m.so $objs /tmp/xv_base/tdnn_layer.o /tmp/xv_base/knock_pp_$m.o
done
ls $out
