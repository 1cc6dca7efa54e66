#!/bin/bash
# round-3 evidence run: GPU tests, the default bench line, rocprofv3 kernel trace of the same command, HBM traffic counters of the
# two sweeps (separate --pmc passes, per MI355X_MICROARCH.md), instruction / LDS counters; summaries go to profiles/ by hand
cd /root/repo; mkdir -p gpurun_out; export TMPDIR=/tmp
TAG=${1:-round3_v1}
(timeout 1200 python -m pytest tests -m gpu -q 2>&1 | tail -6) > gpurun_out/${TAG}_gputests.log
timeout 1500 python bench.py --steps 10 --warmup 2 > gpurun_out/${TAG}_bench.json 2> gpurun_out/${TAG}_bench.err
BENCH="python bench.py --steps 6 --warmup 0 --no-cpu-baseline --no-overlap-run --no-extra-configs --no-result-check"
timeout 900 rocprofv3 --kernel-trace --stats -d gpurun_out/${TAG}_kt -o t -- $BENCH > gpurun_out/${TAG}_kt.log 2>&1
python tools/rocpd_summary.py $(ls gpurun_out/${TAG}_kt/*.db | head -1) > gpurun_out/${TAG}_kernel_trace.txt 2>&1
rm -rf gpurun_out/${TAG}_kt
: > gpurun_out/${TAG}_pmc.txt
ONE="python bench.py --steps 1 --warmup 0 --no-cpu-baseline --no-overlap-run --no-extra-configs --no-result-check"
for C in FETCH_SIZE WRITE_SIZE; do
  timeout 900 rocprofv3 --pmc $C -d gpurun_out/${TAG}_pmc_$C -o p -- $ONE > gpurun_out/${TAG}_pmc_$C.log 2>&1
  python tools/rocpd_pmc.py $(ls gpurun_out/${TAG}_pmc_$C/*.db | head -1) >> gpurun_out/${TAG}_pmc.txt 2>&1
  rm -rf gpurun_out/${TAG}_pmc_$C
done
C1="SQ_INSTS_VALU SQ_INSTS_LDS SQ_INSTS_SALU SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_ACTIVE_INST_VALU SQ_LDS_IDX_ACTIVE SQ_WAIT_INST_LDS"
timeout 600 rocprofv3 --pmc $C1 -d gpurun_out/${TAG}_pv -o p -- $ONE > gpurun_out/${TAG}_pv.log 2>&1
python tools/rocpd_pmc.py $(ls gpurun_out/${TAG}_pv/*.db | head -1) > gpurun_out/${TAG}_pmc_valu.txt 2>&1
rm -rf gpurun_out/${TAG}_pv
C2="SQ_LDS_ADDR_CONFLICT SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_INSTS_LDS SQ_WAIT_INST_ANY SQ_WAVES"
timeout 600 rocprofv3 --pmc $C2 -d gpurun_out/${TAG}_pl -o p -- $ONE > gpurun_out/${TAG}_pl.log 2>&1
python tools/rocpd_pmc.py $(ls gpurun_out/${TAG}_pl/*.db | head -1) > gpurun_out/${TAG}_pmc_lds.txt 2>&1
rm -rf gpurun_out/${TAG}_pl
bash tools/gpu_two_ranks_shared.sh > gpurun_out/${TAG}_two_ranks_shared.log 2>&1
# windows beyond the tiled sweeps (lba_big.h / lba_big_solve.h) and single-window latency at the reference's study sizes
python tools/big_window_prof.py 3 > gpurun_out/${TAG}_big_window.txt 2>&1
python tools/big_solve_phases.py >> gpurun_out/${TAG}_big_window.txt 2>&1
timeout 600 rocprofv3 --kernel-trace --stats -d gpurun_out/${TAG}_bigkt -o t -- python tools/big_window_prof.py 1 > gpurun_out/${TAG}_bigkt.log 2>&1
python tools/rocpd_summary.py $(ls gpurun_out/${TAG}_bigkt/*.db | head -1) >> gpurun_out/${TAG}_big_window.txt 2>&1
rm -rf gpurun_out/${TAG}_bigkt
python tools/oneshot_stages.py > gpurun_out/${TAG}_latency.txt 2>&1
python tools/latency_families.py >> gpurun_out/${TAG}_latency.txt 2>&1
tail -3 gpurun_out/${TAG}_gputests.log; head -12 gpurun_out/${TAG}_kernel_trace.txt; grep -A1 "k_linearise_schur\|k_backsub" gpurun_out/${TAG}_pmc.txt; cut -c1-300 gpurun_out/${TAG}_bench.json
