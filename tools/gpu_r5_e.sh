#!/bin/bash
cd /root/repo; mkdir -p gpurun_out; export TMPDIR=/tmp
TAG=${1:-r5_e}
SLSLAM_EXTRA_FLAGS="-DSLS_MIXED_EXPERIMENT=1" python -c "
from slslam_amd import build as b
print(b.build_lib(force=True, verbose=False))" > gpurun_out/${TAG}_build.log 2>&1
for F in 0 1048576 2097152 4194304 3145728; do
  echo "== SLSLAM_DEBUG_ABLATE=$F" >> gpurun_out/${TAG}_mixed_variants.txt
  SLSLAM_DEBUG_ABLATE=$F timeout 600 python tools/mixed_precision_study.py 2>&1 | tail -5 >> gpurun_out/${TAG}_mixed_variants.txt
done
cat gpurun_out/${TAG}_mixed_variants.txt
