cd /root/repo; export TMPDIR=/tmp; mkdir -p gpurun_out
timeout 900 python -m pytest tests/test_gpu_lba.py -x -q -m gpu -k "beyond_the_tiled or house" > gpurun_out/big_check.log 2>&1
timeout 900 python -m pytest tests/test_house_study.py -x -q -m gpu >> gpurun_out/big_check.log 2>&1
python tools/big_window_prof.py 3 > gpurun_out/big_time.txt 2>&1
tail -15 gpurun_out/big_check.log; cat gpurun_out/big_time.txt
