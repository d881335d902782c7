set -o pipefail
mkdir -p gpurun_out/r04j
timeout -k 10 1100 python -m pytest tests -x -q -m gpu > gpurun_out/r04j/gpu_tests.log 2>&1; echo "pytest rc=$?" | tee -a gpurun_out/r04j/gpu_tests.log
tail -6 gpurun_out/r04j/gpu_tests.log
timeout -k 10 300 python tools/train_layers.py > gpurun_out/r04j/train_layers.txt 2>&1; echo "train_layers rc=$?"
head -3 gpurun_out/r04j/train_layers.txt
