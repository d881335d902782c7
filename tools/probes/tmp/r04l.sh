set -o pipefail
mkdir -p gpurun_out/r04l
timeout -k 10 1100 python -m pytest tests -x -q -m gpu > gpurun_out/r04l/gpu_tests.log 2>&1; echo "pytest rc=$?" | tee -a gpurun_out/r04l/gpu_tests.log
tail -4 gpurun_out/r04l/gpu_tests.log
bash tools/batch_sweep.sh gpurun_out/r04l/batch_sweep.txt
