#!/bin/bash
# Evidence for profiles/: run ON THE GPU BOX (gpurun).  usage: tools/profile_round.sh TAG   -> gpurun_out/TAG_*
#   1. bench.py default line                                      -> TAG_bench.json
#   2. rocprofv3 --kernel-trace --stats of the same bench command -> TAG_kernel_stats.csv (settled: bench's 40 prewarm steps run first)
#   3. rocprofv3 --pmc passes over tools/kbench.py, one counter group per pass (FETCH_SIZE and WRITE_SIZE cannot share a pass)
TAG=$1
cd "$GRAFT_REPO_ROOT" 2>/dev/null || true
export TMPDIR=/tmp
mkdir -p gpurun_out
python bench.py --gpus 1 --steps 100 --warmup 10 > gpurun_out/${TAG}_bench.json 2> gpurun_out/${TAG}_bench.err
rocprofv3 --kernel-trace --stats --output-format csv -d gpurun_out/${TAG}_stats -- python bench.py --steps 20 --warmup 5 --no-cpu-baseline > gpurun_out/${TAG}_stats.log 2>&1
f=$(ls gpurun_out/${TAG}_stats/*/*_kernel_stats.csv 2>/dev/null | head -1)
test -n "$f" && cp "$f" gpurun_out/${TAG}_kernel_stats.csv
for grp in "FETCH_SIZE" "WRITE_SIZE" "TCC_HIT_sum TCC_MISS_sum" "SQ_WAVE_CYCLES SQ_VALU_MFMA_BUSY_CYCLES SQ_BUSY_CYCLES" "SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE" "SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY" "SQ_INSTS_VALU SQ_INSTS_MFMA SQ_INSTS_LDS SQ_INSTS_SALU"; do
  name=$(echo $grp | tr ' ' '_')
  rocprofv3 --kernel-trace --pmc $grp --output-format csv -d gpurun_out/${TAG}pmc_${name} -- python tools/kbench.py 8192 512 bf16 > gpurun_out/${TAG}pmc_${name}.log 2>&1
done
ls gpurun_out | grep ${TAG} | head -30
