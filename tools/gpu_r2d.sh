#!/bin/bash
mkdir -p gpurun_out; cd /root/repo
(timeout 900 python tools/elim_compare.py --modes 3,2 --ablate 256) > gpurun_out/r2d_cmp.log 2>&1
cat gpurun_out/r2d_cmp.log
