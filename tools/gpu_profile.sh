#!/bin/bash
# Run on the GPU box (through gpurun): kernel-trace stats + PMC passes (each in its own run) of bench.py.
# usage: tools/gpu_profile.sh <tag> [bench args...]
# The single-frame legs (latency, output check) are switched off: their small launches of the same kernels would
# dilute the per-kernel averages that bench.py's roofline block is compared with.
set -o pipefail
TAG=${1:-rXX}; shift
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
OUT=gpurun_out/prof_$TAG
mkdir -p $OUT
rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/stats -o $TAG -- python3 bench.py --steps 3 --warmup 1 --no-cpu-baseline --latency-iters 0 --no-check "$@" > $OUT/bench_stats.log 2>&1 || exit 1
rocprofv3 --kernel-trace --pmc FETCH_SIZE --output-format csv -d $OUT/fetch -o $TAG -- python3 bench.py --steps 1 --warmup 1 --no-cpu-baseline --latency-iters 0 --no-check "$@" > $OUT/bench_fetch.log 2>&1 || exit 1
rocprofv3 --kernel-trace --pmc WRITE_SIZE --output-format csv -d $OUT/write -o $TAG -- python3 bench.py --steps 1 --warmup 1 --no-cpu-baseline --latency-iters 0 --no-check "$@" > $OUT/bench_write.log 2>&1 || exit 1
rocprofv3 --kernel-trace --pmc SQ_VALU_MFMA_BUSY_CYCLES SQ_BUSY_CU_CYCLES SQ_WAVE_CYCLES GRBM_GUI_ACTIVE SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE --output-format csv -d $OUT/sq -o $TAG -- python3 bench.py --steps 1 --warmup 1 --no-cpu-baseline --latency-iters 0 --no-check "$@" > $OUT/bench_sq.log 2>&1 || exit 1
grep '^{' $OUT/bench_stats.log > $OUT/bench_line.json
python3 tools/profile_summary.py --tag $TAG --stats $OUT/stats --fetch $OUT/fetch --write $OUT/write --sq $OUT/sq --out $OUT --bench-line $OUT/bench_line.json
