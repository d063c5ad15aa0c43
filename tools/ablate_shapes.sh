#!/bin/bash
# usage: tools/ablate_shapes.sh "B D" ... -- VARIANT...: backward_saved time of each variant at each shape
shapes=(); while [ "$1" != "--" ]; do shapes+=("$1"); shift; done; shift
python tools/kbench.py 8192 512 bf16 > /dev/null 2>&1
for s in "${shapes[@]}"; do for v in "$@"; do
  echo -n "$s $v: "; CROSSCLR_HIP_LIBRARY=variants/lib$v.so python tools/kbench.py $s bf16 2>/dev/null | grep -o "backward_saved=[0-9.]*ms"; done; done
