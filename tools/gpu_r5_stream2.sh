#!/bin/bash
cd /root/repo; mkdir -p gpurun_out; export TMPDIR=/tmp
TAG=${1:-r5_stream2}
bash tools/pack_bench.sh > gpurun_out/${TAG}_pack_bench.txt 2>&1
(timeout 900 python -m pytest tests/test_gpu_stream.py -m gpu -q -x 2>&1 | tail -20) > gpurun_out/${TAG}_streamtests.log
for T in 8 16 32; do
  SLSLAM_REFILL_TIMING=1 timeout 900 python bench.py --steps 5 --warmup 1 --no-cpu-baseline --no-overlap-run --no-extra-configs --no-result-check --profile-steps 0 --host-threads $T > gpurun_out/${TAG}_bench_T$T.json 2> gpurun_out/${TAG}_bench_T$T.err
done
cat gpurun_out/${TAG}_pack_bench.txt; tail -5 gpurun_out/${TAG}_streamtests.log
for T in 8 16 32; do tail -4 gpurun_out/${TAG}_bench_T$T.err; python -c "
import json,sys
j=json.loads(open('gpurun_out/${TAG}_bench_T$T.json').read().strip().splitlines()[-1]); s=j['streamed']
print('T=$T resident', round(j['value']), 'streamed', round(s['value']), 'frac %.3f' % s['fraction_of_resident'], 'ms/batch %.1f submit %.1f wait %.1f copy %.1f' % (s['ms_per_batch'], s['ms_per_batch_in_submit'], s['ms_per_batch_waiting_in_collect'], s['ms_per_batch_copying_results_out']))
"; done
