set -e
mkdir -p gpurun_out
timeout -k 10 900 python -m pytest tests -m gpu -x -q > gpurun_out/r2_full_tests3.log 2>&1
GAMS_FUZZ_SEEDS=500000:20000 timeout -k 10 400 python -m pytest tests/test_gpu_random_params.py -m gpu -x -q > gpurun_out/r2_fuzz3.log 2>&1
timeout -k 10 300 python tools/fuzz_wave_rows.py 300 > gpurun_out/r2_fuzz_rows3.log 2>&1
timeout -k 10 600 python bench.py --gpus 1 --steps 20 --warmup 5 > gpurun_out/r2_bench_default3.log 2>&1
tail -2 gpurun_out/r2_full_tests3.log; tail -1 gpurun_out/r2_fuzz3.log; tail -1 gpurun_out/r2_fuzz_rows3.log; tail -1 gpurun_out/r2_bench_default3.log | cut -c1-1500
