cd /root/repo; export TMPDIR=/tmp; mkdir -p gpurun_out
python tools/big_window_prof.py 3 > gpurun_out/big_time.txt 2>&1
timeout 600 rocprofv3 --kernel-trace --stats -d gpurun_out/bigkt -o t -- python tools/big_window_prof.py 1 > gpurun_out/bigkt.log 2>&1
python tools/rocpd_summary.py $(ls gpurun_out/bigkt/*.db | head -1) > gpurun_out/big_kernel_trace.txt 2>&1
rm -rf gpurun_out/bigkt
cat gpurun_out/big_time.txt; head -40 gpurun_out/big_kernel_trace.txt
