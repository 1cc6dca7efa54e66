#!/bin/bash
# streamed leg: host-side split of a refill (wait / pack / plan+fill / enqueue) per batch, batch period, PCIe rates
cd /root/repo; mkdir -p gpurun_out; export TMPDIR=/tmp
TAG=${1:-r5_stream4}
SLSLAM_REFILL_TIMING=1 timeout 600 python tools/stream_probe.py --batches 8 > gpurun_out/${TAG}_probe.txt 2>&1
timeout 300 python tools/pcie_probe.py > gpurun_out/${TAG}_pcie.json 2>&1
bash tools/pack_bench.sh > gpurun_out/${TAG}_pack_bench.txt 2>&1
tail -30 gpurun_out/${TAG}_probe.txt; cat gpurun_out/${TAG}_pcie.json; tail -14 gpurun_out/${TAG}_pack_bench.txt
