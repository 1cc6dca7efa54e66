#!/bin/bash
mkdir -p gpurun_out; cd /root/repo
(timeout 900 python tools/elim_compare.py --modes 3 --ablate 128,129,133,137,145,161,193,255) > gpurun_out/r2c_cmp.log 2>&1
export TMPDIR=/tmp
C1="SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_INSTS_VALU SQ_INSTS_LDS SQ_INSTS_SALU SQ_WAIT_INST_LDS SQ_ACTIVE_INST_VALU SQ_LDS_IDX_ACTIVE"
SLSLAM_DEBUG_ABLATE=129 timeout 600 rocprofv3 --pmc $C1 -d gpurun_out/r2c_pmc1 -o p -- python bench.py --steps 1 --warmup 0 --no-cpu-baseline --no-overlap-run --elim 3 > gpurun_out/r2c_pmc1.log 2>&1
python tools/rocpd_pmc.py $(ls gpurun_out/r2c_pmc1/*.db | head -1) > gpurun_out/r2c_pmc1.txt 2>&1
rm -rf gpurun_out/r2c_pmc1
cat gpurun_out/r2c_cmp.log | cut -c1-330; grep -A9 "k_eliminate" gpurun_out/r2c_pmc1.txt
