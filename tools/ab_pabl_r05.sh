#!/bin/bash
# Round 5's timing ablations / block-shape MODELS of fast_bwd_xfp_kernel (CROSSCLR_PABL bits 13-18, csrc/crossclr_kernels_dslp.h; WRONG results).
# In the build container (product objects in /tmp/crossclr_objs, see tools/build_variant_tu.py):
#   for z in 0 1 2 3 131 1920 8192 16384 32768 98304 131072 180224 245760 262144; do
#     python tools/build_variant_tu.py p$z tu_saved_xfp.cpp -DCROSSCLR_DSL_MINIMAL -DCROSSCLR_PABL=$z; done
# On the GPU box:  bash tools/ab_pabl_r05.sh gpurun_out/r05_pabl_raw.txt     (three interleaved rounds; compare within a round)
out=${1:-gpurun_out/r05_pabl_raw.txt}
CROSSCLR_HIP_LIBRARY=variants/libp0.so timeout 120 python tools/time_xfp.py warm > /dev/null 2>&1
for round in 1 2 3; do
  for f in $(ls variants/libp*.so | sort -V); do
    v=$(basename $f .so); v=${v#lib}
    CROSSCLR_HIP_LIBRARY=$f timeout 120 python tools/time_xfp.py "r$round $v" 2>/dev/null | tail -1 >> $out
  done
done
# engine clock / power while the product kernel runs back to back
CROSSCLR_HIP_LIBRARY=variants/libp0.so timeout 120 python tools/time_xfp.py clocks 2>/dev/null >> $out
cat $out
