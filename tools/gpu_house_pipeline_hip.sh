#!/bin/bash
# the MI355X half of tools/gpu_house_pipeline.sh (the oracle half, host-only and unchanged since, is profiles/round2_house_pipeline_timing.txt)
cd /root/repo; mkdir -p gpurun_out
timeout 1500 python tools/house_study.py --backend hip --sigmas 0.2,1.0 --windows 5,10,20,40 --frames 400 2>&1 | grep -v amdgpu.ids > gpurun_out/house_hip.jsonl
python - <<'PY'
import json
h=[json.loads(l) for l in open("gpurun_out/house_hip.jsonl") if l.startswith("{")]
print("sigma  W   obs/window  its/frame hip|reference   final cost hip|reference     LBA ms/call   motion-only ms/call   ms/keyframe")
for a in h:
    ref=a.get("reference",{})
    print("%.1f  %3d  %8.0f   %5.2f | %5.2f     %.4e | %.4e   %7.2f     %6.2f     %7.2f" % (
        a["sigma_px"],a["W"],a["solver_time"]["avg_observations_per_window"],a["avg_iterations"],ref.get("avg_iterations",float("nan")),
        a["avg_final_cost"],ref.get("avg_final_cost",float("nan")),a["solver_time"]["lba_ms_per_call"],
        a["solver_time"]["motion_only_ms_per_call"],a["solver_time"]["optimisation_ms_per_keyframe"]))
PY
