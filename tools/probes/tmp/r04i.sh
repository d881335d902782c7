set -o pipefail
mkdir -p gpurun_out/r04i
timeout -k 10 600 python -m pytest tests/test_x3_gpu.py -x -q -m gpu -k "first_conv or modelA or synthetic or batch256" > gpurun_out/r04i/pytest.log 2>&1; echo "pytest rc=$?" | tee -a gpurun_out/r04i/pytest.log
tail -4 gpurun_out/r04i/pytest.log
timeout -k 10 600 python bench.py --layers --steps 10 --warmup 3 --other-tier-steps 0 --q8-steps 0 --latency-iters 0 --bf16-steps 0 --int8-steps 0 --large-steps 0 --train-steps 0 --no-cpu-baseline > gpurun_out/r04i/bench_line.json 2> gpurun_out/r04i/bench_layers.txt; echo "bench rc=$?"
tail -27 gpurun_out/r04i/bench_layers.txt | head -3; tail -1 gpurun_out/r04i/bench_layers.txt
