#!/bin/bash
# differential soak of the streamed path on the final code: every streamed window against its solo solve (bit for bit) and a sample against the oracle
cd /root/repo; mkdir -p gpurun_out
: > gpurun_out/round6_soak_stream.txt
for ARGS in "40 48 21 0 0" "40 48 22 4 0" "40 48 23 0 1" "40 48 24 4 1" "40 48 25 4 2"; do
  echo "== python tests/tools/soak_stream.py $ARGS" >> gpurun_out/round6_soak_stream.txt
  timeout 600 python tests/tools/soak_stream.py $ARGS 2>&1 | grep -v amdgpu | tail -8 >> gpurun_out/round6_soak_stream.txt
done
timeout 900 python tests/tools/soak_device_build.py 2>&1 | tail -2 >> gpurun_out/round6_soak_stream.txt
cat gpurun_out/round6_soak_stream.txt
