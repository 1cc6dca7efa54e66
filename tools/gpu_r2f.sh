#!/bin/bash
mkdir -p gpurun_out; cd /root/repo
(timeout 900 python tools/elim_compare.py --modes 3,2 --ablate 0) > gpurun_out/r2f_cmp.log 2>&1
(timeout 900 python tools/elim_compare.py --modes 3 --ablate 256) >> gpurun_out/r2f_cmp.log 2>&1
cut -c1-420 gpurun_out/r2f_cmp.log
