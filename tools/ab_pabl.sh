#!/bin/bash
# times tools/time_xfp.py for every variants/lib<name>.so given (ablation builds of the pair kernel)
export CROSSCLR_AB_OLD_ABI=1
out=$1; shift
for v in "$@"; do CROSSCLR_HIP_LIBRARY=variants/lib$v.so timeout 120 python tools/time_xfp.py $v 2>/dev/null | tail -1 >> $out; done
cat $out
