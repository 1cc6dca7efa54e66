#!/bin/bash
# round 4: the grouped matrix-core elimination sweep (lba_elimination = 4) against the default sweep - parity tests, the short bench of
# both, and (PHASES=1) the phase stamps of the grouped sweep from a timing build made on the box
cd /root/repo; mkdir -p gpurun_out; export TMPDIR=/tmp
TAG=${1:-r4g}
timeout 900 python -m pytest tests/test_gpu_lba.py -x -q -m gpu -k "matrix_core or grouped" > gpurun_out/${TAG}_tests.log 2>&1; echo "pytest rc $?" >> gpurun_out/${TAG}_tests.log
tail -n 5 gpurun_out/${TAG}_tests.log
for E in ${ELIMS:-0 4}; do
  timeout 600 python bench.py --steps 5 --warmup 2 --elim $E --no-cpu-baseline --no-overlap-run --no-extra-configs > gpurun_out/${TAG}_bench_e$E.json 2> gpurun_out/${TAG}_bench_e$E.err
  python - gpurun_out/${TAG}_bench_e$E.json $E <<'PY'
import json,sys
try:
    d=json.load(open(sys.argv[1])); k=d["kernel_ms_per_step"]
    print("elim %s value %.0f ms/step %.3f K1 %.4f backsub %.4f solve %.4f" % (sys.argv[2], d["value"], d["ms_per_step"], d["roofline"]["avg_launch_ms"], d["roofline_backsub"]["avg_launch_ms"], d["reduced_solve_mfma"]["avg_launch_ms"]))
except Exception as e:
    print("elim %s FAILED %r" % (sys.argv[2], e)); print(open(sys.argv[1].replace('.json','.err')).read()[-1500:])
PY
done
if [ -n "$PHASES" ]; then
  SLSLAM_EXTRA_FLAGS="-DSLSLAM_K1_TIMING=1" python -c "from slslam_amd import build; build.build_lib(force=True)" > gpurun_out/${TAG}_build.log 2>&1
  timeout 600 python tools/grouped_phases.py 1024 4 2>&1 | tail -n 3 | tee gpurun_out/${TAG}_phases.txt
  python -c "from slslam_amd import build; build.build_lib(force=True)" > /dev/null 2>&1
fi
