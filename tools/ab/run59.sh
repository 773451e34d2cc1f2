set -e
mkdir -p gpurun_out
timeout -k 10 200 python -m pytest tests/test_gpu_wave.py tests/test_gpu_fullsize.py -m gpu -x -q -k "kernel_name or tapered" > gpurun_out/r2_kname.log 2>&1
GAMS_FUZZ_SEEDS=1000000:120000 timeout -k 10 700 python -m pytest tests/test_gpu_random_params.py -m gpu -x -q > gpurun_out/r2_fuzz4.log 2>&1
timeout -k 10 200 python tools/fuzz_many_ctgs.py 60 > gpurun_out/r2_fuzz_many4.log 2>&1
timeout -k 10 150 python tools/fuzz_intervals.py 20 > gpurun_out/r2_fuzz_iv4.log 2>&1
tail -1 gpurun_out/r2_kname.log; tail -1 gpurun_out/r2_fuzz4.log; tail -1 gpurun_out/r2_fuzz_many4.log; tail -1 gpurun_out/r2_fuzz_iv4.log
