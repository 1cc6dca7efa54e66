#!/bin/bash
# the reference's own study configuration (house scene, sigma 0.2 / 1.0 px, W = 5 .. 40, 400 keyframes, max 10 iterations) end to end:
# time inside the solver calls per keyframe, MI355X path against the oracle on the box's host
cd /root/repo; mkdir -p gpurun_out
timeout 1500 python tools/house_study.py --backend hip --sigmas 0.2,1.0 --windows 5,10,20,40 --frames 400 2>&1 | grep -v amdgpu.ids > gpurun_out/house_hip.jsonl
timeout 3000 python tools/house_study.py --backend oracle --sigmas 0.2,1.0 --windows 5,10,20,40 --frames 400 > gpurun_out/house_oracle.jsonl 2>&1
python - <<'PY'
import json
h=[json.loads(l) for l in open("gpurun_out/house_hip.jsonl") if l.startswith("{")]
o=[json.loads(l) for l in open("gpurun_out/house_oracle.jsonl") if l.startswith("{")]
print("sigma  W   obs/window  its/frame hip|oracle|reference   final cost hip|oracle|reference     LBA ms/call hip|oracle   motion-only ms hip|oracle   ms/keyframe hip|oracle")
for a,b in zip(h,o):
    ref=a.get("reference",{})
    print("%.1f  %3d  %8.0f   %5.2f | %5.2f | %5.2f     %.4e | %.4e | %.4e   %7.2f | %8.1f     %6.2f | %6.2f     %7.2f | %8.1f" % (
        a["sigma_px"],a["W"],a["solver_time"]["avg_observations_per_window"],a["avg_iterations"],b["avg_iterations"],ref.get("avg_iterations",float("nan")),
        a["avg_final_cost"],b["avg_final_cost"],ref.get("avg_final_cost",float("nan")),a["solver_time"]["lba_ms_per_call"],b["solver_time"]["lba_ms_per_call"],
        a["solver_time"]["motion_only_ms_per_call"],b["solver_time"]["motion_only_ms_per_call"],a["solver_time"]["optimisation_ms_per_keyframe"],b["solver_time"]["optimisation_ms_per_keyframe"]))
PY
