#!/bin/bash
cd /root/repo; mkdir -p gpurun_out; export TMPDIR=/tmp
TAG=${1:-r5_f}
timeout 900 python tools/mixed_precision_study.py > gpurun_out/${TAG}_mixed_study.txt 2>&1
(timeout 900 python -m pytest tests/test_gpu_lba.py -m gpu -q -x -s -k "mixed_precision" 2>&1 | tail -12) > gpurun_out/${TAG}_mixed.log
SLSLAM_REFILL_TIMING=1 timeout 900 python bench.py --steps 10 --warmup 2 --no-cpu-baseline --no-overlap-run > gpurun_out/${TAG}_bench.json 2> gpurun_out/${TAG}_bench.err
tail -9 gpurun_out/${TAG}_mixed_study.txt; cat gpurun_out/${TAG}_mixed.log; tail -4 gpurun_out/${TAG}_bench.err
python - <<PY
import json
j=json.loads(open('gpurun_out/${TAG}_bench.json').read().strip().splitlines()[-1])
print('value', j['value'], 'ms', j['ms_per_step'], j['kernel_ms_per_step'], j['results_check'].get('equal_to_stored_1_rank_digest'))
s=j['streamed']; print({k: s[k] for k in ('value','ms_per_batch','fraction_of_resident','ms_per_batch_in_submit','ms_per_batch_waiting_in_collect','bitwise_equal_to_resident_batch')})
m=j['mixed_precision']; print({k: m.get(k) for k in ('value','vs_double_path','sum_final_cost','sum_final_cost_double_path')}, m['roofline']['avg_launch_ms'])
PY
