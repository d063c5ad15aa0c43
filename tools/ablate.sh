#!/bin/bash
# usage: tools/ablate.sh PREFIX [B D]: time every variants/libPREFIX*.so with tools/kbench.py (one process each)
P=$1; B=${2:-8192}; D=${3:-512}
python tools/kbench.py $B $D bf16 > /dev/null 2>&1   # settle the GPU clocks
for f in $(ls variants/lib${P}*.so | sort -V); do
  echo -n "$(basename $f): "
  CROSSCLR_HIP_LIBRARY=$f python tools/kbench.py $B $D bf16 2>/dev/null | grep -o "forward=[0-9.]*ms\|backward=[0-9.]*ms\|forward_save=[0-9.]*ms\|backward_saved=[0-9.]*ms" | tr '\n' ' '
  echo
done
