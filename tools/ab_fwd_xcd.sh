#!/bin/bash
# A/B of the forward's XCD-aware range placement on ONE box (CROSSCLR_FWD_XCD=0: range c on block c): stage times, then the L2 counters of
# the forward kernel under both settings.  usage (gpurun): bash tools/ab_fwd_xcd.sh > gpurun_out/rNN_ab_fwd_xcd.txt
cd "$GRAFT_REPO_ROOT" 2>/dev/null || true
export TMPDIR=/tmp
for rep in 1 2 3; do
  for v in 0 1; do
    echo -n "FWD_XCD=$v rep$rep: "; CROSSCLR_FWD_XCD=$v python tools/kbench.py 8192 512 bf16 2>/dev/null | tail -1 | sed 's/backward=.*forward_save/forward_save/; s/backward_saved_xf1.*//'
  done
done
for v in 0 1; do
  for grp in "TCC_HIT_sum TCC_MISS_sum" "FETCH_SIZE"; do
    name=$(echo $grp | tr ' ' '_')
    rm -rf /tmp/abx_${v}_${name}
    CROSSCLR_FWD_XCD=$v rocprofv3 --kernel-trace --pmc $grp --output-format csv -d /tmp/abx_${v}_${name} -- python tools/kbench.py 8192 512 bf16 > /dev/null 2>&1
    f=$(ls /tmp/abx_${v}_${name}/*/*counter_collection.csv 2>/dev/null | head -1)
    test -n "$f" && python - "$f" "$v" <<'PY'
import csv, sys, collections
acc = collections.defaultdict(lambda: collections.defaultdict(list))
for r in csv.DictReader(open(sys.argv[1])):
    if "fast_fwd_pipe_kernel<32, 1, false, true" in r["Kernel_Name"]:
        acc[r["Counter_Name"]][r["Dispatch_Id"]].append(float(r["Counter_Value"]))
for c, d in acc.items():
    vals = [sum(v) for v in d.values()]
    print(f"FWD_XCD={sys.argv[2]} forward_save {c}: mean per dispatch {sum(vals)/len(vals):.0f} over {len(vals)} dispatches")
PY
  done
done
