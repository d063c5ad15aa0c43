#!/bin/bash
# Evidence for profiles/: run ON THE GPU BOX (gpurun).  usage: tools/profile_round.sh TAG   -> gpurun_out/TAG_*
#   1. bench.py default line (+ the config 2, D = 1024, influential D = 1024 and fp32 lines) -> TAG_bench*.json
#   2. rocprofv3 --kernel-trace --stats of the same bench command -> TAG_kernel_stats.csv (settled: bench's 40 prewarm steps run first);
#      the same for --mode fp32                                   -> TAG_fp32_kernel_stats.csv
#   3. rocprofv3 --pmc passes over tools/kbench.py, one counter group per pass (FETCH_SIZE and WRITE_SIZE cannot share a pass),
#      bf16 (7 groups) and fp32 (4 groups)
#   4. micro-benchmarks: MFMA chains / operand data, power probe  -> TAG_micro.txt
TAG=$1
cd "$GRAFT_REPO_ROOT" 2>/dev/null || true
export TMPDIR=/tmp
mkdir -p gpurun_out
# the micro-benchmarks of step 4 are built HERE (the binaries are not tracked): a missing compiler or a failed build stops the script
for m in mfma_chains waves_per_simd; do
  /opt/rocm/bin/hipcc --offload-arch=gfx950 -O3 tools/micro/$m.hip -o tools/micro/$m || { echo "building tools/micro/$m failed" >&2; exit 1; }
done
python bench.py --gpus 1 --steps 100 --warmup 10 > gpurun_out/${TAG}_bench.json 2> gpurun_out/${TAG}_bench.err
python bench.py --mode fp32 --fwd-only --rows 4096 --steps 50 > gpurun_out/${TAG}_bench_config2_fp32_fwd.json 2>> gpurun_out/${TAG}_bench.err
python bench.py --mode fp32 --steps 30 > gpurun_out/${TAG}_bench_fp32.json 2>> gpurun_out/${TAG}_bench.err
python bench.py --dim 1024 --steps 50 > gpurun_out/${TAG}_bench_d1024.json 2>> gpurun_out/${TAG}_bench.err
python bench.py --dim 1024 --influential --steps 50 > gpurun_out/${TAG}_bench_influential_d1024.json 2>> gpurun_out/${TAG}_bench.err
python tools/second_order_bench.py > gpurun_out/${TAG}_second_order.txt 2>&1      # create_graph=True: forward + backward + closed-form double backward
# same command as the bench line (40 settle steps, 10 warm-up, 100 timed), profiled; its own JSON line is kept beside the summary so that the
# HIP-event figure and the rocprofv3 figure of ONE process on ONE box can be compared (profiled passes run ~2 % slower: MICROARCH guide)
rocprofv3 --kernel-trace --stats --output-format csv -d gpurun_out/${TAG}_stats -- python bench.py --steps 100 --warmup 10 --no-cpu-baseline --no-secondary --sustained-steps 0 > gpurun_out/${TAG}_bench_under_rocprof.json 2> gpurun_out/${TAG}_stats.log
f=$(ls gpurun_out/${TAG}_stats/*/*_kernel_trace.csv 2>/dev/null | head -1)
test -n "$f" && python tools/kernel_trace_summary.py "$f" > gpurun_out/${TAG}_kernel_stats.csv
f=$(ls gpurun_out/${TAG}_stats/*/*_kernel_stats.csv 2>/dev/null | head -1)
test -n "$f" && cp "$f" gpurun_out/${TAG}_kernel_stats_rocprof.csv
rocprofv3 --kernel-trace --stats --output-format csv -d gpurun_out/${TAG}_stats32 -- python bench.py --mode fp32 --steps 10 --warmup 3 --prewarm 10 --no-cpu-baseline > gpurun_out/${TAG}_stats32.log 2>&1
f=$(ls gpurun_out/${TAG}_stats32/*/*_kernel_stats.csv 2>/dev/null | head -1)
test -n "$f" && cp "$f" gpurun_out/${TAG}_fp32_kernel_stats.csv
for grp in "FETCH_SIZE" "WRITE_SIZE" "TCC_HIT_sum TCC_MISS_sum" "SQ_WAVE_CYCLES SQ_VALU_MFMA_BUSY_CYCLES SQ_BUSY_CYCLES" "SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE" "SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY" "SQ_INSTS_VALU SQ_INSTS_MFMA SQ_INSTS_LDS SQ_INSTS_SALU"; do
  name=$(echo $grp | tr ' ' '_')
  rocprofv3 --kernel-trace --pmc $grp --output-format csv -d gpurun_out/${TAG}pmc_${name} -- python tools/kbench.py 8192 512 bf16 > gpurun_out/${TAG}pmc_${name}.log 2>&1
done
for grp in "FETCH_SIZE" "WRITE_SIZE" "TCC_HIT_sum TCC_MISS_sum" "SQ_WAVE_CYCLES SQ_VALU_MFMA_BUSY_CYCLES SQ_BUSY_CYCLES"; do
  name=$(echo $grp | tr ' ' '_')
  rocprofv3 --kernel-trace --pmc $grp --output-format csv -d gpurun_out/${TAG}pmc32_${name} -- python tools/kbench.py 8192 512 fp32 > gpurun_out/${TAG}pmc32_${name}.log 2>&1
done
python tools/make_pmc_json.py ${TAG}pmc_ gpurun_out/${TAG}_pmc.json 8192 512 bf16 > gpurun_out/${TAG}_pmc_summary.txt 2>&1
python tools/make_pmc_json.py ${TAG}pmc32_ gpurun_out/${TAG}_fp32_pmc.json 8192 512 fp32 >> gpurun_out/${TAG}_pmc_summary.txt 2>&1
test -x tools/micro/mfma_chains || { echo "tools/micro/mfma_chains missing" >&2; exit 1; }
{ echo "# tools/micro/mfma_chains"; tools/micro/mfma_chains > /dev/null; tools/micro/mfma_chains | tail -7; echo "# tools/power_probe.py"; python tools/power_probe.py 2>/dev/null | tail -3;
  echo "# tools/clock_probe.sh"; bash tools/clock_probe.sh 2>/dev/null | tail -6; } > gpurun_out/${TAG}_micro.txt 2>&1
ls gpurun_out | grep ${TAG} | head -40
