set -e
mkdir -p gpurun_out
timeout -k 10 500 python -m pytest tests/test_gpu_wave.py tests/test_gpu_host.py -m gpu -x -q > gpurun_out/r2_s1_tests.log 2>&1
timeout -k 10 300 python -m pytest tests/test_gpu_fullsize.py -m gpu -x -q -k "step1_geometry" >> gpurun_out/r2_s1_tests.log 2>&1
timeout -k 10 300 python tools/ab.py tools/ab/base.so gams_amd/libgams_gpu.so --step 1 --rounds 5 --reps 10 > gpurun_out/r2_s1_ab.log 2>&1
timeout -k 10 200 python tools/stamps_raw.py 0 384 1 > gpurun_out/r2_s1_stamps.log 2>&1
timeout -k 10 300 bash tools/pmc_quick.sh s1 "GRCh38-step1 --scale 0.1" > gpurun_out/r2_s1_pmc.log 2>&1
tail -3 gpurun_out/r2_s1_tests.log; cat gpurun_out/r2_s1_ab.log; grep -n "median phase" gpurun_out/r2_s1_stamps.log; cat gpurun_out/r2_s1_pmc.log
