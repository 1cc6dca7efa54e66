#!/bin/bash
cd /root/repo; mkdir -p gpurun_out
(timeout 1200 python -m pytest tests/test_gpu_lba.py -m gpu -q -x -k randomised 2>&1 | tail -15) > gpurun_out/fuzz_fuzz_test.log; cat gpurun_out/fuzz_fuzz_test.log
FUZZ_SEED=77 timeout 1500 python tests/tools/fuzz_parity.py 400 2>&1 | grep -v amdgpu.ids | tail -12 > gpurun_out/fuzz_fuzz_400.log; cat gpurun_out/fuzz_fuzz_400.log
FUZZ_SEED=78 FUZZ_OVERSIZE=1 timeout 1500 python tests/tools/fuzz_parity.py 90 2>&1 | grep -v amdgpu.ids | tail -12 > gpurun_out/fuzz_fuzz_oversize_90.log; cat gpurun_out/fuzz_fuzz_oversize_90.log
