#!/bin/bash
# dsl vs rows saved backward at several shapes: tools/r03_shapes.sh TAG
TAG=${1:-shapes}
mkdir -p gpurun_out
python tools/kbench.py 8192 512 bf16 > /dev/null 2>&1
for shape in "8192 512" "8192 256" "8192 128" "8192 384" "8192 1024" "8192 768" "4096 512" "2048 512" "16384 512"; do
  for k in dsl rows; do
    echo -n "$shape $k: "; CROSSCLR_SAVED_BWD=$k python tools/kbench.py $shape bf16 2>/dev/null | grep -o "backward_saved=[0-9.]*ms\|forward_save=[0-9.]*ms" | tr '\n' ' '; echo
  done
done 2>&1 | tee gpurun_out/${TAG}_shapes.txt
