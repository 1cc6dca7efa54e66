#!/bin/bash
# round 5: the headline test, the stream / refill / reproducible tests, the rest of the GPU suite, then the bench line with the streamed leg
cd /root/repo; mkdir -p gpurun_out; export TMPDIR=/tmp
TAG=${1:-r5_stream}
(timeout 900 python -m pytest tests/test_gpu_stream.py -m gpu -q -x 2>&1 | tail -40) > gpurun_out/${TAG}_streamtests.log
(timeout 2000 python -m pytest tests -m gpu -q --deselect tests/test_gpu_stream.py 2>&1 | tail -40) > gpurun_out/${TAG}_gputests.log
timeout 900 python bench.py --steps 10 --warmup 2 --no-cpu-baseline --no-overlap-run --no-extra-configs > gpurun_out/${TAG}_bench.json 2> gpurun_out/${TAG}_bench.err
cat gpurun_out/${TAG}_streamtests.log; tail -15 gpurun_out/${TAG}_gputests.log; python - <<'PY'
import json,sys
try:
    j=json.loads(open('gpurun_out/%s_bench.json' % sys.argv[1] if len(sys.argv)>1 else 'gpurun_out/r5_stream_bench.json').read().strip().splitlines()[-1])
    print('value', j['value'], 'ms', j['ms_per_step']); print(json.dumps(j.get('streamed'), indent=1))
except Exception as e:
    print('bench parse failed', e); print(open('gpurun_out/r5_stream_bench.err').read()[-2000:])
PY
