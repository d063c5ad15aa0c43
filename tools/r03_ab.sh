#!/bin/bash
# A/B of the saved-backward kernels on the GPU box: tools/r03_ab.sh TAG -> gpurun_out/TAG_*.txt
TAG=${1:-ab}
mkdir -p gpurun_out
python tools/kbench.py 8192 512 bf16 > /dev/null 2>&1   # settle
for i in 1 2 3; do
  echo "dsl : $(python tools/kbench.py 8192 512 bf16 2>/dev/null | tail -1)"
  echo "rows: $(CROSSCLR_SAVED_BWD=rows python tools/kbench.py 8192 512 bf16 2>/dev/null | tail -1)"
done > gpurun_out/${TAG}_kbench.txt 2>&1
cat gpurun_out/${TAG}_kbench.txt
