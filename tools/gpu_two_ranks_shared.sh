#!/bin/bash
# two ranks of bench.py sharing the one visible GPU over gloo (RCCL refuses two ranks on one device): exercises the N > 1 code
# of the bench line - shard_range, all-gathers, per-rank clocks and the cross-rank bitwise check - on a 1-GPU box
cd /root/repo; mkdir -p gpurun_out
export SLSLAM_BENCH_SHARE_GPU=1 HSA_ENABLE_IPC_MODE_LEGACY=0 SLSLAM_BENCH_BACKEND=gloo
timeout 600 python -m torch.distributed.run --nnodes=1 --nproc-per-node 2 --master-addr 127.0.0.1 --master-port 29533 \
   bench.py --gpus 2 --steps 3 --warmup 1 --windows 128 --no-cpu-baseline --no-overlap-run --no-extra-configs 2>&1 | grep -v "amdgpu.ids" | grep "^{\|rror" > gpurun_out/two_ranks_shared.json
python - <<'PY'
import json
d=json.loads(open("gpurun_out/two_ranks_shared.json").read().splitlines()[-1])
for k in ("value","n_gpus","rccl_ranks_seen","collective_backend","distinct_devices_seen","per_rank_ms_per_step","results_check"): print(k, d.get(k))
PY
