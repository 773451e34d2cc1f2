set -e
mkdir -p gpurun_out
timeout -k 10 600 python -m pytest tests/test_gpu_wave.py tests/test_gpu_host.py tests/test_gpu_random_params.py -m gpu -x -q > gpurun_out/r2_coop2_tests.log 2>&1
timeout -k 10 400 python -m pytest tests/test_gpu_fullsize.py -m gpu -x -q -k "step1_geometry or taper" >> gpurun_out/r2_coop2_tests.log 2>&1
timeout -k 10 500 bash tools/prof.sh grch38s1 0 --workload GRCh38-step1 --scale 0.12 --steps 20 --warmup 3 --no-extra > gpurun_out/r2_prof_s1.log 2>&1
timeout -k 10 500 python bench.py --workload GRCh38-step1 --no-cpu --no-extra --no-secondary --steps 20 --warmup 3 > gpurun_out/r2_bench_s1.log 2>&1
grep -h "passed\|failed" gpurun_out/r2_coop2_tests.log; tail -8 gpurun_out/r2_prof_s1.log; tail -2 gpurun_out/r2_bench_s1.log
