#!/bin/bash
# per-launch durations of the sweeps over the ten LM iterations of a bench step, with and without the kept-Jacobian replay
cd /root/repo; mkdir -p gpurun_out; export TMPDIR=/tmp
TAG=${1:-keeptrace}
: > gpurun_out/${TAG}.txt
for KEEP in 1 0; do
  BENCH="python bench.py --eager --steps 3 --warmup 0 --profile-steps 0 --keep-jacobian $KEEP ${BENCH_ARGS} --no-cpu-baseline --no-overlap-run --no-extra-configs --no-result-check"
  timeout 900 rocprofv3 --kernel-trace --stats -d gpurun_out/${TAG}_kt$KEEP -o t -- $BENCH > gpurun_out/${TAG}_kt$KEEP.log 2>&1
  echo "== lba_keep_jacobian $KEEP" >> gpurun_out/${TAG}.txt
  python tools/rocpd_sequence.py $(ls gpurun_out/${TAG}_kt$KEEP/*.db | head -1) >> gpurun_out/${TAG}.txt 2>&1
  rm -rf gpurun_out/${TAG}_kt$KEEP
done
cat gpurun_out/${TAG}.txt
