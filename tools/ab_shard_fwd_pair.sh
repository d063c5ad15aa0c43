#!/bin/bash
# Per-rank compute of an N-rank step (tools/shard_bench.py: one GPU plays rank N/2, communication excluded) with round 4's forward for the
# remote blocks (CROSSCLR_FWD_PAIR=0) and with fast_fwd_pair_kernel<..., KIND 2 / 3>, alternating.
B=${1:-8192}; D=${2:-512}
for r in 1 2; do
  for e in 0 1; do
    echo "== round $r CROSSCLR_FWD_PAIR=$e"
    CROSSCLR_FWD_PAIR=$e python tools/shard_bench.py $B $D 2>/dev/null | tail -5
  done
done
