#!/bin/bash
# times every variants/lib*.so given as arguments (names without lib/.so) with tools/kbench.py; prints backward_saved (and forward_save)
mkdir -p gpurun_out
python tools/kbench.py 8192 512 bf16 > /dev/null 2>&1   # settle the GPU clocks
for rep in 1 2; do
echo -n "rows(main lib): "; CROSSCLR_SAVED_BWD=rows python tools/kbench.py 8192 512 bf16 2>/dev/null | grep -o "forward_save=[0-9.]*ms\|backward_saved=[0-9.]*ms" | tr '\n' ' '; echo
for v in "$@"; do
  f=variants/lib$v.so
  echo -n "$v: "
  CROSSCLR_HIP_LIBRARY=$f python tools/kbench.py 8192 512 bf16 2>/dev/null | grep -o "forward_save=[0-9.]*ms\|backward_saved=[0-9.]*ms" | tr '\n' ' '
  echo
done
done
