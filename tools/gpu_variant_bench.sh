#!/bin/bash
# builds the library with extra compile flags ON the GPU box (hipcc is in the image) and runs the short bench: one line per variant.
# (BENCH_ARGS: extra bench.py arguments, e.g. "--elim 4"; BENCH_STEPS: timed steps, default 10)
# usage: gpu_variant_bench.sh TAG "flags variant 1" "flags variant 2" ...
cd /root/repo; mkdir -p gpurun_out; export TMPDIR=/tmp
TAG=$1; shift
: > gpurun_out/${TAG}_variants.txt
for FLAGS in "$@"; do
  SLSLAM_EXTRA_FLAGS="$FLAGS" python -c "from slslam_amd import build; build.build_lib(force=True)" > gpurun_out/${TAG}_build.log 2>&1 || { echo "BUILD FAILED: $FLAGS" >> gpurun_out/${TAG}_variants.txt; continue; }
  CLK0=$(rocm-smi --showclocks 2>/dev/null | grep -m1 sclk | sed 's/.*(\([0-9]*Mhz\)).*/\1/')
  timeout 600 python bench.py --steps ${BENCH_STEPS:-10} --warmup 2 ${BENCH_ARGS} --no-cpu-baseline --no-overlap-run --no-extra-configs > gpurun_out/${TAG}_b.json 2> gpurun_out/${TAG}_b.err
  CLK1=$(rocm-smi --showclocks 2>/dev/null | grep -m1 sclk | sed 's/.*(\([0-9]*Mhz\)).*/\1/')
  echo -n "[sclk $CLK0 -> $CLK1] " >> gpurun_out/${TAG}_variants.txt
  python - "$FLAGS" gpurun_out/${TAG}_b.json >> gpurun_out/${TAG}_variants.txt <<'PY'
import json,sys
try:
    d=json.load(open(sys.argv[2]))
    k=d["kernel_ms_per_step"]
    print("%-60s value %.0f  K1 %.4f  backsub %.4f  solve %.4f  update %.4f ms/launch" % (sys.argv[1] or "(default)", d["value"], d["roofline"]["avg_launch_ms"], d["roofline_backsub"]["avg_launch_ms"], d["reduced_solve_mfma"]["avg_launch_ms"], k.get("lm_update",0)/10.0))
except Exception as e:
    print("%-60s FAILED %r" % (sys.argv[1], e))
PY
done
python -c "from slslam_amd import build; build.build_lib(force=True)" > /dev/null 2>&1
cat gpurun_out/${TAG}_variants.txt
