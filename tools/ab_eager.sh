#!/bin/bash
# The driver's command with and without the eager gradient product (CROSSCLR_EAGER_BACKWARD), alternating: wall ms/step over the 20 timed steps.
for r in 1 2 3; do
  for e in 1 0; do
    echo -n "round $r eager=$e: "
    CROSSCLR_EAGER_BACKWARD=$e python bench.py --gpus 1 --steps 20 --warmup 5 --no-cpu-baseline --no-secondary --sustained-steps 400 2>/dev/null | python -c "
import json,sys
d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('wall', round(d['ms_per_step'],4), 'event median', round(d['ms_per_step_event_median'],4), 'min/max', [round(x,4) for x in d['ms_per_step_event_min_max']], 'sustained', d['sustained']['ms_per_step'], 'finish', d['kernels']['forward_finish'])"
  done
done
