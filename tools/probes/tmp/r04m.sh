set -o pipefail
mkdir -p gpurun_out/r04m
timeout -k 10 900 python -m pytest tests/test_forward_gpu.py tests/test_ops_gpu.py tests/test_train_gpu.py -x -q -m gpu > gpurun_out/r04m/pytest.log 2>&1; echo "pytest rc=$?" | tee -a gpurun_out/r04m/pytest.log
tail -4 gpurun_out/r04m/pytest.log
timeout -k 10 600 python bench.py --tier fp32 --layers --steps 6 --warmup 2 --other-tier-steps 0 --q8-steps 0 --latency-iters 0 --bf16-steps 0 --int8-steps 0 --large-steps 0 --train-steps 0 --no-cpu-baseline > gpurun_out/r04m/bench_fp32.json 2> gpurun_out/r04m/bench_fp32_layers.txt; echo "bench rc=$?"
tail -30 gpurun_out/r04m/bench_fp32_layers.txt
