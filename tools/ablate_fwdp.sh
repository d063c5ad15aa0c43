#!/bin/bash
# Timing ablations of fast_fwd_pair_kernel (CROSSCLR_ZABL, csrc/crossclr_kernels_symp.h; results of the ablated variants are WRONG).
# In the build container:  for z in 0 1 2 4 8 16 32 64 3 47 111; do python tools/build_variant_tu.py z$z tu_fwdp.cpp -DCROSSCLR_ZABL=$z; done
# On the GPU box:          bash tools/ablate_fwdp.sh [B D]   (three interleaved rounds: compare within a round)
B=${1:-8192}; D=${2:-512}
python tools/kbench.py $B $D bf16 > /dev/null 2>&1   # settle the GPU clocks
for round in 1 2 3; do
  for f in $(ls variants/libz*.so | sort -V); do
    echo -n "round $round $(basename $f): "
    CROSSCLR_HIP_LIBRARY=$f python tools/kbench.py $B $D bf16 2>/dev/null | grep -o " forward=[0-9.]*ms\|forward_save=[0-9.]*ms\|backward_saved=[0-9.]*ms" | tr '\n' ' '
    echo
  done
done
