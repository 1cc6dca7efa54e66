#!/bin/bash
mkdir -p gpurun_out; cd /root/repo
(timeout 900 python -m pytest tests -m gpu -q 2>&1 | tail -15) > gpurun_out/r2e_tests.log
(timeout 900 python tools/elim_compare.py --modes 1,3,2 --ablate 0) > gpurun_out/r2e_cmp.log 2>&1
(timeout 900 python tools/elim_compare.py --modes 3 --ablate 256,1,2) >> gpurun_out/r2e_cmp.log 2>&1
tail -8 gpurun_out/r2e_tests.log; cut -c1-700 gpurun_out/r2e_cmp.log
