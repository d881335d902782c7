set -o pipefail
mkdir -p gpurun_out/r04h
timeout -k 10 900 python -m pytest tests/test_x3_gpu.py tests/test_forward_gpu.py tests/test_benchsize_gpu.py -x -q -m gpu > gpurun_out/r04h/pytest.log 2>&1; echo "pytest rc=$?" | tee -a gpurun_out/r04h/pytest.log
tail -6 gpurun_out/r04h/pytest.log
for fs in 1 0; do
UNET_TRAIN_FUSED_STATS=$fs timeout -k 10 300 python tools/probes/gradnorm_dev.py >> gpurun_out/r04h/gradnorm.txt 2>&1
UNET_TRAIN_FUSED_STATS=$fs UNET_TRAIN_X3=0 timeout -k 10 300 python tools/probes/gradnorm_dev.py >> gpurun_out/r04h/gradnorm.txt 2>&1
done
grep -v amdgpu.ids gpurun_out/r04h/gradnorm.txt
timeout -k 10 600 python bench.py --layers --steps 10 --warmup 3 --other-tier-steps 0 --q8-steps 0 --latency-iters 0 --bf16-steps 0 --int8-steps 0 --large-steps 0 --train-steps 0 --no-cpu-baseline > gpurun_out/r04h/bench_line.json 2> gpurun_out/r04h/bench_layers.txt; echo "bench rc=$?"
tail -27 gpurun_out/r04h/bench_layers.txt | head -4; tail -1 gpurun_out/r04h/bench_layers.txt
