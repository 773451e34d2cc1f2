set -e
mkdir -p gpurun_out
timeout -k 10 600 python -m pytest tests/test_gpu_wave.py tests/test_gpu_host.py -m gpu -x -q > gpurun_out/r2_w28_tests.log 2>&1
timeout -k 10 400 python -m pytest tests/test_gpu_fullsize.py -m gpu -x -q -k "step1_geometry or taper" >> gpurun_out/r2_w28_tests.log 2>&1
timeout -k 10 300 python tools/ab.py gams_amd/libgams_gpu.so --step 1 --tiles 0,5120 --rounds 5 --reps 10 > gpurun_out/r2_w28_ab.log 2>&1
GAMS_FUZZ_SEEDS=1000000:100000 timeout -k 10 600 python -m pytest tests/test_gpu_random_params.py -m gpu -x -q > gpurun_out/r2_fuzz4.log 2>&1
timeout -k 10 200 python tools/fuzz_many_ctgs.py 40 > gpurun_out/r2_fuzz_many4.log 2>&1
grep -h "passed\|failed" gpurun_out/r2_w28_tests.log; cat gpurun_out/r2_w28_ab.log; tail -1 gpurun_out/r2_fuzz4.log; tail -1 gpurun_out/r2_fuzz_many4.log
