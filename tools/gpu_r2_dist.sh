#!/bin/bash
# two ranks of bench.py sharing the one visible GPU: first over RCCL ("nccl": expected to refuse two ranks on one device),
# then with gloo for the rendezvous / collectives; logs kept under profiles/
cd /root/repo; mkdir -p gpurun_out
export SLSLAM_BENCH_SHARE_GPU=1 HSA_ENABLE_IPC_MODE_LEGACY=0
for be in nccl gloo; do
  echo "=== backend $be" 
  SLSLAM_BENCH_BACKEND=$be timeout 600 python -m torch.distributed.run --nnodes=1 --nproc-per-node 2 --master-addr 127.0.0.1 --master-port 29533 \
     bench.py --gpus 2 --steps 3 --warmup 1 --windows 256 --no-cpu-baseline --no-overlap-run --no-extra-configs --gather-results 2>&1 | grep -v "amdgpu.ids" | grep -i "nccl\|rccl\|duplicate\|error\|^{" | cut -c1-400 | head -20
done
echo "=== one rank, same per-rank batch"
timeout 600 python bench.py --gpus 1 --steps 3 --warmup 1 --windows 256 --no-cpu-baseline --no-overlap-run --no-extra-configs --gather-results 2>&1 | grep -v amdgpu.ids | tail -2
