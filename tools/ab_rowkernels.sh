#!/bin/bash
# Same-box A/B of two builds of the library: whole step (tools/step_time.py) and the stage timings of tools/kbench.py, processes interleaved.
# usage (GPU box): bash tools/ab_rowkernels.sh variants/libOLD.so [B D]        (the second build is the product library)
OLD=$1; B=${2:-8192}; D=${3:-512}
for round in 1 2 3; do
  CROSSCLR_HIP_LIBRARY=$OLD timeout 300 python tools/step_time.py old $B $D 2>/dev/null | tail -1
  timeout 300 python tools/step_time.py new $B $D 2>/dev/null | tail -1
done
for round in 1 2; do
  echo -n "old: "; CROSSCLR_HIP_LIBRARY=$OLD timeout 300 python tools/kbench.py $B $D bf16 2>/dev/null | grep -o " normalize[a-z_]*=[0-9.]*ms\|forward_finish=[0-9.]*ms" | tr '\n' ' '; echo
  echo -n "new: "; timeout 300 python tools/kbench.py $B $D bf16 2>/dev/null | grep -o " normalize[a-z_]*=[0-9.]*ms\|forward_finish=[0-9.]*ms" | tr '\n' ' '; echo
done
