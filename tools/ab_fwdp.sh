#!/bin/bash
# A/B of fast_fwd_pair_kernel tuning variants (variants/lib*.so built by tools/build_variant_tu.py): bit-identity test per variant, then
# three interleaved timing rounds (compare within a round).  usage (GPU box): bash tools/ab_fwdp.sh [B D]
B=${1:-8192}; D=${2:-512}
for f in $(ls variants/lib*.so | sort -V); do
  echo -n "check $(basename $f): "
  CROSSCLR_HIP_LIBRARY=$f python -m pytest tests/test_gpu_fwd_pair.py -x -q 2>&1 | tail -1
done
python tools/kbench.py $B $D bf16 > /dev/null 2>&1   # settle the GPU clocks
for round in 1 2 3; do
  for f in $(ls variants/lib*.so | sort -V); do
    echo -n "round $round $(basename $f): "
    CROSSCLR_HIP_LIBRARY=$f python tools/kbench.py $B $D bf16 2>/dev/null | grep -o " forward=[0-9.]*ms\|forward_save=[0-9.]*ms\|backward_saved=[0-9.]*ms" | tr '\n' ' '
    echo
  done
done
