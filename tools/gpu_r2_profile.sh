#!/bin/bash
# round-2 evidence run: GPU tests, the default bench line, rocprofv3 kernel trace of the same command, HBM traffic counters
# of the two sweeps (separate --pmc passes, per MI355X_MICROARCH.md), summaries copied to profiles/
cd /root/repo; mkdir -p gpurun_out; export TMPDIR=/tmp
TAG=${1:-round2_v1}
(timeout 1200 python -m pytest tests -m gpu -q 2>&1 | tail -6) > gpurun_out/${TAG}_gputests.log
timeout 1500 python bench.py --steps 10 --warmup 2 > gpurun_out/${TAG}_bench.json 2> gpurun_out/${TAG}_bench.err
BENCH="python bench.py --steps 6 --warmup 0 --no-cpu-baseline --no-overlap-run --no-extra-configs"
timeout 900 rocprofv3 --kernel-trace --stats -d gpurun_out/${TAG}_kt -o t -- $BENCH > gpurun_out/${TAG}_kt.log 2>&1
python tools/rocpd_summary.py $(ls gpurun_out/${TAG}_kt/*.db | head -1) > gpurun_out/${TAG}_kernel_trace.txt 2>&1
rm -rf gpurun_out/${TAG}_kt
: > gpurun_out/${TAG}_pmc.txt
for C in FETCH_SIZE WRITE_SIZE; do
  timeout 900 rocprofv3 --pmc $C -d gpurun_out/${TAG}_pmc_$C -o p -- python bench.py --steps 1 --warmup 0 --no-cpu-baseline --no-overlap-run --no-extra-configs > gpurun_out/${TAG}_pmc_$C.log 2>&1
  python tools/rocpd_pmc.py $(ls gpurun_out/${TAG}_pmc_$C/*.db | head -1) >> gpurun_out/${TAG}_pmc.txt 2>&1
  rm -rf gpurun_out/${TAG}_pmc_$C
done
# pose graph (BASELINE config 5): kernel trace of the structured and the dense factorisation, matrix-core busy cycles of the dense one
for V in structured dense; do
  timeout 600 rocprofv3 --kernel-trace --stats -d gpurun_out/${TAG}_po_$V -o t -- python tools/po_prof.py $V > gpurun_out/${TAG}_po_$V.log 2>&1
  python tools/rocpd_summary.py $(ls gpurun_out/${TAG}_po_$V/*.db | head -1) > gpurun_out/${TAG}_po_kernel_trace_$V.txt 2>&1
  rm -rf gpurun_out/${TAG}_po_$V
done
timeout 600 rocprofv3 --pmc SQ_VALU_MFMA_BUSY_CYCLES SQ_BUSY_CYCLES SQ_INSTS_VALU_MFMA_MOPS_F64 -d gpurun_out/${TAG}_po_pmc -o p -- python tools/po_prof.py dense > gpurun_out/${TAG}_po_pmc.log 2>&1
python tools/rocpd_pmc.py $(ls gpurun_out/${TAG}_po_pmc/*.db | head -1) > gpurun_out/${TAG}_po_pmc.txt 2>&1
rm -rf gpurun_out/${TAG}_po_pmc
timeout 600 rocprofv3 --pmc SQ_VALU_MFMA_BUSY_CYCLES SQ_BUSY_CYCLES SQ_INSTS_VALU_MFMA_MOPS_F64 -d gpurun_out/${TAG}_lba_mfma -o p -- python bench.py --steps 1 --warmup 0 --no-cpu-baseline --no-overlap-run --no-extra-configs > gpurun_out/${TAG}_lba_mfma.log 2>&1
python tools/rocpd_pmc.py $(ls gpurun_out/${TAG}_lba_mfma/*.db | head -1) > gpurun_out/${TAG}_lba_mfma_pmc.txt 2>&1
rm -rf gpurun_out/${TAG}_lba_mfma
# the fp64 issue ceilings the roofline `binding` field quotes (tools/micro/mfma_f64_bench.hip, built by hipcc in-tree)
[ -x tools/micro/lab/mfma_f64_bench ] || /opt/rocm/bin/hipcc --offload-arch=gfx950 -O3 -o tools/micro/lab/mfma_f64_bench tools/micro/mfma_f64_bench.hip
timeout 300 tools/micro/lab/mfma_f64_bench > gpurun_out/${TAG}_micro_fp64.txt 2>&1
bash tools/gpu_r2_dist.sh > gpurun_out/${TAG}_dist_two_ranks.log 2>&1
tail -3 gpurun_out/${TAG}_gputests.log; head -12 gpurun_out/${TAG}_kernel_trace.txt; grep -A1 "k_linearise_schur\|k_backsub" gpurun_out/${TAG}_pmc.txt; cut -c1-400 gpurun_out/${TAG}_bench.json
