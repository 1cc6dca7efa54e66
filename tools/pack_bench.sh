#!/bin/bash
# host-thread scaling of the packer on this box (no GPU needed)
cd /root/repo; mkdir -p /tmp/pk gpurun_out; cd /tmp/pk
python - <<'PY'
import numpy as np, sys
sys.path.insert(0,'/root/repo')
from slslam_amd import synth
w=synth.make_window(5,num_lines=2000)
np.array([w['num_cameras'],w['num_lines'],len(w['camera_index'])],dtype=np.int32).tofile('hdr.bin')
np.asarray(w['camera_index'],dtype=np.int32).tofile('cam.bin')
np.asarray(w['line_index'],dtype=np.int32).tofile('line.bin')
np.asarray(w['fixed_index'],dtype=np.int32).tofile('fixed.bin')
np.asarray(w['observations'],dtype=np.float64).tofile('obs.bin')
np.asarray(w['parameters'],dtype=np.float64).tofile('par.bin')
PY
g++ -O3 -std=c++17 -pthread -I/root/repo/slslam_amd/csrc /root/repo/tools/pack_bench.cpp /root/repo/slslam_amd/csrc/lba_pack.cpp -o pack_bench
for T in 1 2 4 8 16 32; do ./pack_bench $T; done
/opt/rocm/bin/hipcc -O3 -std=c++17 -I/root/repo/slslam_amd/csrc /root/repo/tools/pack_bench2.cpp /root/repo/slslam_amd/csrc/lba_pack.cpp -o pack_bench2 2>/dev/null
for P in 0 1; do for T in 1 8 16; do ./pack_bench2 $T $P 512; done; done
lscpu | grep -i "numa\|socket\|model name\|thread" | head -8
