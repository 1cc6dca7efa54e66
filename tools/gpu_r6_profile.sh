#!/bin/bash
# round-6 evidence run: GPU tests, the default bench line, rocprofv3 kernel trace of the same command (eager and graph replay), HBM traffic
# counters of the two sweeps (separate --pmc passes, per MI355X_MICROARCH.md), the streamed leg: its kernels alone (depth-1 stream), its
# timeline with three batches in flight, its host-side split in the three modes, the build phases of one window, the first solve after a
# refill; summaries go to profiles/ by hand
cd /root/repo; mkdir -p gpurun_out; export TMPDIR=/tmp
TAG=${1:-round6_v1}
(timeout 2700 python -m pytest tests -m gpu -q 2>&1 | tail -6) > gpurun_out/${TAG}_gputests.log
timeout 1500 python bench.py > gpurun_out/${TAG}_bench.json 2> gpurun_out/${TAG}_bench.err
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
# the streamed leg
export GPU_MAX_HW_QUEUES=8
timeout 900 rocprofv3 --kernel-trace --stats -d gpurun_out/${TAG}_iso -o t -- python tools/stream_probe.py --batches 6 --depth 1 --mode pinned --host-threads 1 > gpurun_out/${TAG}_iso.log 2>&1
python tools/rocpd_summary.py gpurun_out/${TAG}_iso/t_results.db > gpurun_out/${TAG}_stream_kernels_alone.txt 2>&1
rm -rf gpurun_out/${TAG}_iso
timeout 900 rocprofv3 --kernel-trace --memory-copy-trace -d gpurun_out/${TAG}_tl -o t -- python tools/stream_probe.py --batches 8 --mode pinned --host-threads 1 > gpurun_out/${TAG}_tl.log 2>&1
python tools/rocpd_timeline.py gpurun_out/${TAG}_tl/t_results.db | tail -90 > gpurun_out/${TAG}_stream_timeline.txt 2>&1
rm -rf gpurun_out/${TAG}_tl
unset GPU_MAX_HW_QUEUES
: > gpurun_out/${TAG}_stream_probe.txt
for MODE in pinned packed pageable host; do
  T=1; [ $MODE = pageable ] && T=2; [ $MODE = host ] && T=16
  echo "== mode $MODE, $T host thread(s)" >> gpurun_out/${TAG}_stream_probe.txt
  SLSLAM_REFILL_TIMING=1 timeout 600 python tools/stream_probe.py --batches 16 --mode $MODE --host-threads $T 2>&1 | tail -12 >> gpurun_out/${TAG}_stream_probe.txt
done
python tools/build_phases.py > gpurun_out/${TAG}_build_phases.txt 2>&1
python tools/first_solve_after_refill.py 2>&1 | grep -v amdgpu > gpurun_out/${TAG}_first_solve_after_refill.txt
python tools/solve_phases.py 2>&1 | grep -v amdgpu > gpurun_out/${TAG}_reduced_solve_phases.txt
python tools/po_bench.py 2>&1 | grep -v amdgpu > gpurun_out/${TAG}_po_bench.txt
tools/micro/_build/zcb > gpurun_out/${TAG}_zero_copy_bench.txt 2>&1
tail -3 gpurun_out/${TAG}_gputests.log; head -12 gpurun_out/${TAG}_kernel_trace.txt; grep -A2 "k_eliminate_grouped\|k_backsub" gpurun_out/${TAG}_pmc.txt | head -20; tail -c 1500 gpurun_out/${TAG}_bench.json
