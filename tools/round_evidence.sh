#!/bin/bash
# One gpurun call = one box: GPU suite, smoke, then tools/profile_round.sh TAG (bench lines, rocprofv3 summaries, PMC passes, micro-benchmarks).
# usage (from the build container):  gpurun --timeout 1500 -- 'bash tools/round_evidence.sh r04m'
TAG=$1
cd "$GRAFT_REPO_ROOT" 2>/dev/null || true
mkdir -p gpurun_out
python -m pytest tests -m gpu -x -q > gpurun_out/${TAG}_gputests_full.txt 2>&1; grep -E "passed|failed|error" gpurun_out/${TAG}_gputests_full.txt | tail -3 > gpurun_out/${TAG}_gputests.txt
python -c "import __graft_entry__ as g; g.smoke()" > gpurun_out/${TAG}_smoke.txt 2>&1
bash tools/profile_round.sh $TAG > gpurun_out/${TAG}_profile.log 2>&1
cat gpurun_out/${TAG}_gputests.txt; tail -2 gpurun_out/${TAG}_smoke.txt
python - <<EOF
import json
d=json.loads(open("gpurun_out/${TAG}_bench.json").read().strip().splitlines()[-1])
print(d["ms_per_step"], d["ms_per_step_event_median"], d["sustained"]["ms_per_step"], d["roofline"]["frac"], d["roofline"]["avg_launch_ms"])
print({k:(v.get("in_step_ms"),v["ms"]) for k,v in d["kernels"].items()})
for k,v in d.get("secondary",{}).items(): print(k, v.get("ms_per_step"), v.get("dominant_kernel_ms"), v.get("frac"))
EOF
head -12 gpurun_out/${TAG}_kernel_stats.csv | cut -c1-60,200-400
