#!/bin/bash
cd /root/repo; mkdir -p gpurun_out; export TMPDIR=/tmp
TAG=${1:-r5_g}
timeout 900 python bench.py --steps 10 --warmup 2 --no-cpu-baseline --no-overlap-run --no-extra-configs --no-streamed > gpurun_out/${TAG}_bench.json 2> gpurun_out/${TAG}_bench.err
(timeout 2400 python -m pytest tests -m gpu -q 2>&1 | tail -15) > gpurun_out/${TAG}_gputests.log
python - <<PY
import json
j=json.loads(open('gpurun_out/${TAG}_bench.json').read().strip().splitlines()[-1])
print('value', j['value'], 'ms', j['ms_per_step'], j['kernel_ms_per_step'], j['results_check'].get('equal_to_stored_1_rank_digest'), j['results_check']['bitwise_equal_to_rank0_resolve'])
PY
tail -8 gpurun_out/${TAG}_gputests.log
