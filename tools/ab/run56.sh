set -e
mkdir -p gpurun_out
timeout -k 10 600 python -m pytest tests/test_gpu_wave.py tests/test_gpu_host.py tests/test_gpu_random_params.py -m gpu -x -q > gpurun_out/r2_coop_tests.log 2>&1
timeout -k 10 400 python -m pytest tests/test_gpu_fullsize.py -m gpu -x -q -k "step1_geometry or taper or guard" >> gpurun_out/r2_coop_tests.log 2>&1
timeout -k 10 300 python tools/ab.py tools/ab/base.so tools/ab/coop.so tools/ab/both.so --step 1 --rounds 5 --reps 10 > gpurun_out/r2_coop_ab1.log 2>&1
timeout -k 10 300 python tools/ab.py tools/ab/base.so tools/ab/coop.so --step 10 --rounds 7 --reps 30 > gpurun_out/r2_coop_ab10.log 2>&1
grep -h "passed\|failed" gpurun_out/r2_coop_tests.log; cat gpurun_out/r2_coop_ab1.log gpurun_out/r2_coop_ab10.log
