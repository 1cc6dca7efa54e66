#!/bin/bash
cd /root/repo; mkdir -p gpurun_out; export TMPDIR=/tmp
C1="SQ_INSTS_VALU SQ_INSTS_LDS SQ_INSTS_SALU SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_ACTIVE_INST_VALU SQ_LDS_IDX_ACTIVE SQ_WAIT_INST_LDS"
timeout 600 rocprofv3 --pmc $C1 -d gpurun_out/pv -o p -- python bench.py --steps 1 --warmup 0 --no-cpu-baseline --no-overlap-run --no-extra-configs > gpurun_out/pv.log 2>&1
python tools/rocpd_pmc.py $(ls gpurun_out/pv/*.db | head -1) > gpurun_out/r2_pmc_valu.txt 2>&1
rm -rf gpurun_out/pv
grep -A9 "k_linearise_schur\|k_backsub\|k_reduced_solve" gpurun_out/r2_pmc_valu.txt
