set -u
timeout -k 10 900 python -m pytest tests -q -m gpu --durations=25 > gpurun_out/r4_gpu_tests.log 2>&1 || { echo "gpu tests failed"; tail -30 gpurun_out/r4_gpu_tests.log; exit 1; }
tail -3 gpurun_out/r4_gpu_tests.log
timeout -k 10 900 python bench.py > gpurun_out/r4_bench_default.jsonl 2> gpurun_out/r4_bench_default.err || { echo "bench failed"; tail -5 gpurun_out/r4_bench_default.err; exit 1; }
tail -c 500 gpurun_out/r4_bench_default.jsonl; echo
