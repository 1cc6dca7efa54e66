#!/bin/bash
# round-2 GPU session b: tests, ablation of the matrix-core sweep, PMC counters
mkdir -p gpurun_out; cd /root/repo
(timeout 900 python -m pytest tests -m gpu -q 2>&1 | tail -40) > gpurun_out/r2b_tests.log
(timeout 900 python tools/elim_compare.py --modes 3,2 --ablate 0) > gpurun_out/r2b_cmp.log 2>&1
(timeout 900 python tools/elim_compare.py --modes 3 --ablate 1,2,4,5) >> gpurun_out/r2b_cmp.log 2>&1
export TMPDIR=/tmp
C1="SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_INSTS_VALU SQ_INSTS_LDS SQ_INSTS_SALU SQ_WAIT_INST_LDS SQ_ACTIVE_INST_VALU SQ_LDS_IDX_ACTIVE"
C2="SQ_LDS_BANK_CONFLICT SQ_LDS_ADDR_CONFLICT SQ_VALU_MFMA_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_INSTS_MFMA SQ_WAVES"
for i in 1 2; do
  eval CC=\$C$i
  timeout 600 rocprofv3 --pmc $CC -d gpurun_out/r2b_pmc$i -o p -- python bench.py --steps 1 --warmup 0 --no-cpu-baseline --no-overlap-run --elim 3 > gpurun_out/r2b_pmc$i.log 2>&1
  python tools/rocpd_pmc.py $(ls gpurun_out/r2b_pmc$i/*.db | head -1) > gpurun_out/r2b_pmc$i.txt 2>&1
  rm -rf gpurun_out/r2b_pmc$i
done
tail -12 gpurun_out/r2b_tests.log; cat gpurun_out/r2b_cmp.log; grep -A9 "k_eliminate" gpurun_out/r2b_pmc1.txt gpurun_out/r2b_pmc2.txt
