#!/bin/bash
# HBM traffic (FETCH_SIZE / WRITE_SIZE passes) and time of the two sweeps for a build with extra flags: gpu_traffic_variant.sh TAG "flags"
cd /root/repo; mkdir -p gpurun_out; export TMPDIR=/tmp
TAG=$1; FLAGS=$2
SLSLAM_EXTRA_FLAGS="$FLAGS" python -c "from slslam_amd import build; build.build_lib(force=True)" > gpurun_out/${TAG}_build.log 2>&1 || { echo BUILD FAILED; exit 1; }
ONE="python bench.py --eager --steps 1 --warmup 0 --no-cpu-baseline --no-overlap-run --no-extra-configs --no-result-check"
: > gpurun_out/${TAG}_pmc.txt
for C in FETCH_SIZE WRITE_SIZE; do
  timeout 900 rocprofv3 --pmc $C -d gpurun_out/${TAG}_pmc_$C -o p -- $ONE > gpurun_out/${TAG}_pmc_$C.log 2>&1
  python tools/rocpd_pmc.py $(ls gpurun_out/${TAG}_pmc_$C/*.db | head -1) >> gpurun_out/${TAG}_pmc.txt 2>&1
  rm -rf gpurun_out/${TAG}_pmc_$C
done
echo "flags: $FLAGS"; grep -A1 "k_eliminate_grouped<false>\|k_backsub" gpurun_out/${TAG}_pmc.txt | grep -v "^--"
timeout 600 python bench.py --eager --steps 5 --warmup 2 --no-cpu-baseline --no-overlap-run --no-extra-configs --no-result-check 2>/dev/null | python -c "
import json,sys
d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('value %.0f K1 %.4f backsub %.4f' % (d['value'], d['roofline']['avg_launch_ms'], d['roofline_backsub']['avg_launch_ms']))"
python -c "from slslam_amd import build; build.build_lib(force=True)" > /dev/null 2>&1
