cd $GRAFT_REPO_ROOT
for b in nccl gloo; do
  timeout 120 python -m torch.distributed.run --nnodes=1 --nproc-per-node 2 --master-addr 127.0.0.1 --master-port 29511 tools/probe_one_gpu_ranks.py $b > gpurun_out/r04l_probe_$b.txt 2>&1
  echo "rc $?" >> gpurun_out/r04l_probe_$b.txt
done
grep -h "^\[" gpurun_out/r04l_probe_*.txt | sort | head -40
