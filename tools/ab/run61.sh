set -e
mkdir -p gpurun_out
timeout -k 10 600 python -m pytest tests/test_gpu_wave.py -m gpu -x -q > gpurun_out/r2_w20_tests.log 2>&1
timeout -k 10 400 python -m pytest tests/test_gpu_fullsize.py -m gpu -x -q -k "config2 or taper or guard" >> gpurun_out/r2_w20_tests.log 2>&1
GAMS_W20_MIN_TILES=1536 timeout -k 10 400 python -m pytest tests/test_gpu_fullsize.py -m gpu -x -q -k "config2_atha or guard" >> gpurun_out/r2_w20_tests.log 2>&1
timeout -k 10 300 python tools/ab.py gams_amd/libgams_gpu.so --step 10 --tiles 0,5120 --rounds 7 --reps 30 > gpurun_out/r2_w20_abA.log 2>&1
GAMS_W20_MIN_TILES=1536 timeout -k 10 300 python tools/ab.py gams_amd/libgams_gpu.so --step 10 --tiles 0,5120 --rounds 7 --reps 30 > gpurun_out/r2_w20_abB.log 2>&1
timeout -k 10 300 python tools/ab.py gams_amd/libgams_gpu.so --step 10 --tiles 0,5120 --rounds 7 --reps 30 > gpurun_out/r2_w20_abA2.log 2>&1
grep -h "passed\|failed" gpurun_out/r2_w20_tests.log; echo A; cat gpurun_out/r2_w20_abA.log; echo B; cat gpurun_out/r2_w20_abB.log; echo A2; cat gpurun_out/r2_w20_abA2.log
