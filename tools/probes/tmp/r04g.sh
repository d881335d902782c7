set -o pipefail
mkdir -p gpurun_out/r04g
timeout -k 10 900 python -m pytest tests/test_x3_gpu.py tests/test_benchsize_gpu.py tests/test_train_gpu.py tests/test_q8_gpu.py -x -q -m gpu > gpurun_out/r04g/pytest.log 2>&1; echo "pytest rc=$?" | tee -a gpurun_out/r04g/pytest.log
tail -6 gpurun_out/r04g/pytest.log
timeout -k 10 600 python bench.py --layers --steps 10 --warmup 3 --other-tier-steps 0 --q8-steps 0 --latency-iters 0 --bf16-steps 0 --int8-steps 0 --large-steps 0 --train-steps 3 --no-cpu-baseline > gpurun_out/r04g/bench_line.json 2> gpurun_out/r04g/bench_layers.txt; echo "bench rc=$?"
tail -27 gpurun_out/r04g/bench_layers.txt
UNET_X3_T448_C4=0 timeout -k 10 600 python bench.py --layers --steps 10 --warmup 3 --other-tier-steps 0 --q8-steps 0 --latency-iters 0 --bf16-steps 0 --int8-steps 0 --large-steps 0 --train-steps 0 --no-cpu-baseline > gpurun_out/r04g/bench_line_noc4.json 2> gpurun_out/r04g/bench_layers_noc4.txt; echo "bench(no c4) rc=$?"
tail -2 gpurun_out/r04g/bench_layers_noc4.txt
timeout -k 10 300 python tools/train_layers.py > gpurun_out/r04g/train_layers.txt 2>&1; echo "train_layers rc=$?"
head -3 gpurun_out/r04g/train_layers.txt
python - <<'PY'
import json
l=json.load(open('gpurun_out/r04g/bench_line.json'))
print('fps',l['value'],'ms',l['ms_per_step'],'train',l.get('train',{}).get('ms_per_step'))
PY
