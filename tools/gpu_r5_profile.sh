#!/bin/bash
# round-5 evidence run: GPU tests, the default bench line (hipGraph replay + eager profiled pass), rocprofv3 kernel trace of the same
# command, HBM traffic counters of the two sweeps (separate --pmc passes, per MI355X_MICROARCH.md), instruction / LDS / matrix-core
# counters, the LDS-atomic sweep's line for comparison, the two-rank shared-GPU line, the pose-graph dense path; summaries go to
# profiles/ by hand
cd /root/repo; mkdir -p gpurun_out; export TMPDIR=/tmp
TAG=${1:-round5_v1}
(timeout 2700 python -m pytest tests -m gpu -q 2>&1 | tail -6) > gpurun_out/${TAG}_gputests.log
timeout 1500 python bench.py --steps 10 --warmup 2 > gpurun_out/${TAG}_bench.json 2> gpurun_out/${TAG}_bench.err
timeout 900 python bench.py --steps 10 --warmup 2 --elim 1 --no-cpu-baseline --no-overlap-run --no-extra-configs --no-streamed > gpurun_out/${TAG}_bench_lds_atomic_sweep.json 2> gpurun_out/${TAG}_bench_lds_atomic_sweep.err
BENCH="python bench.py --eager --steps 6 --warmup 0 --no-cpu-baseline --no-overlap-run --no-extra-configs --no-streamed --no-result-check"
timeout 900 rocprofv3 --kernel-trace --stats -d gpurun_out/${TAG}_kt -o t -- $BENCH > gpurun_out/${TAG}_kt.log 2>&1
python tools/rocpd_summary.py $(ls gpurun_out/${TAG}_kt/*.db | head -1) > gpurun_out/${TAG}_kernel_trace.txt 2>&1
rm -rf gpurun_out/${TAG}_kt
GRAPH="python bench.py --steps 6 --warmup 0 --profile-steps 0 --no-cpu-baseline --no-overlap-run --no-extra-configs --no-streamed --no-result-check"
timeout 900 rocprofv3 --kernel-trace --stats -d gpurun_out/${TAG}_ktg -o t -- $GRAPH > gpurun_out/${TAG}_ktg.log 2>&1
python tools/rocpd_summary.py $(ls gpurun_out/${TAG}_ktg/*.db | head -1) > gpurun_out/${TAG}_kernel_trace_graph_replay.txt 2>&1
rm -rf gpurun_out/${TAG}_ktg
: > gpurun_out/${TAG}_pmc.txt
ONE="python bench.py --eager --steps 1 --warmup 0 --no-cpu-baseline --no-overlap-run --no-extra-configs --no-streamed --no-result-check"
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
C3="SQ_INSTS_VALU_MFMA_F64 SQ_INSTS_MFMA SQ_VALU_MFMA_BUSY_CYCLES SQ_BUSY_CYCLES SQ_INSTS_VALU_MFMA_MOPS_F64"
timeout 600 rocprofv3 --pmc $C3 -d gpurun_out/${TAG}_pm -o p -- $ONE > gpurun_out/${TAG}_pm.log 2>&1
python tools/rocpd_pmc.py $(ls gpurun_out/${TAG}_pm/*.db | head -1) > gpurun_out/${TAG}_pmc_mfma.txt 2>&1
rm -rf gpurun_out/${TAG}_pm
bash tools/gpu_two_ranks_shared.sh > gpurun_out/${TAG}_two_ranks_shared.log 2>&1
# pose graph: dense path per-kernel times, factorisation / substitution per iteration at three graph sizes
timeout 300 rocprofv3 --kernel-trace --stats -d gpurun_out/${TAG}_pokt -o t -- python tools/po_dense_prof.py > gpurun_out/${TAG}_pokt.log 2>&1
python tools/rocpd_summary.py $(ls gpurun_out/${TAG}_pokt/*.db | head -1) > gpurun_out/${TAG}_po_kernel_trace_dense.txt 2>&1
rm -rf gpurun_out/${TAG}_pokt
timeout 300 rocprofv3 --pmc $C3 -d gpurun_out/${TAG}_popm -o p -- python tools/po_dense_prof.py > gpurun_out/${TAG}_popm.log 2>&1
python tools/rocpd_pmc.py $(ls gpurun_out/${TAG}_popm/*.db | head -1) > gpurun_out/${TAG}_po_pmc_mfma.txt 2>&1
rm -rf gpurun_out/${TAG}_popm
bash tools/gpu_po_check.sh ${TAG}_po > /dev/null 2>&1
python tools/latency_families.py > gpurun_out/${TAG}_latency.txt 2>&1
tail -3 gpurun_out/${TAG}_gputests.log; head -12 gpurun_out/${TAG}_kernel_trace.txt; grep -A2 "k_eliminate_grouped\|k_backsub" gpurun_out/${TAG}_pmc.txt; cut -c1-300 gpurun_out/${TAG}_bench.json
# round 5: the streamed leg's host-side split, the pose-graph structured trace
SLSLAM_REFILL_TIMING=1 timeout 600 python tools/stream_probe.py --batches 16 > gpurun_out/${TAG}_stream_probe.txt 2>&1
timeout 300 rocprofv3 --kernel-trace --stats -d gpurun_out/${TAG}_pos -o t -- python tools/po_structured_prof.py > gpurun_out/${TAG}_pos.log 2>&1
python tools/rocpd_summary.py $(ls gpurun_out/${TAG}_pos/*.db | head -1) > gpurun_out/${TAG}_po_structured_trace.txt 2>&1
rm -rf gpurun_out/${TAG}_pos
