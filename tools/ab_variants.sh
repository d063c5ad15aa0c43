#!/bin/bash
# A/B of library variants on ONE box, interleaved: tools/ab_variants.sh OUTFILE REPS name1 name2 ...   ("main" = the in-tree library)
# prints tools/kbench.py's per-kernel line for every (rep, variant)
out=$1; reps=$2; shift 2
mkdir -p gpurun_out
python tools/kbench.py 8192 512 bf16 > /dev/null 2>&1   # settle the GPU clocks
for rep in $(seq 1 $reps); do
  for v in "$@"; do
    echo -n "$v rep$rep: " >> $out
    if [ "$v" = main ]; then python tools/kbench.py 8192 512 bf16 2>/dev/null | tail -1 >> $out
    else CROSSCLR_HIP_LIBRARY=variants/lib$v.so python tools/kbench.py 8192 512 bf16 2>/dev/null | tail -1 >> $out; fi
  done
done
cat $out
