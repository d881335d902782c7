#!/bin/bash
# GPU box: the headline forward (f16x3, 224 x 224) over batch sizes, with each convolution structure switched off in turn:
# the data behind run_conv_x3's dispatch thresholds (work items >= 128, csrc/unet_x3.inc).
# usage: tools/batch_sweep.sh <out.txt> [batches...]
OUT=${1:-gpurun_out/batch_sweep.txt}; shift
BATCHES=${@:-"2 4 8 12 16 24 32 48 64 96 128 192 256"}
mkdir -p $(dirname $OUT)
echo "batch | all structures | no third (UNET_X3_T448=0 UNET_X3_T448_C4=0) | no third, no second (+ UNET_X3_R512=0)   [ms per step, frames/s]" > $OUT
for B in $BATCHES; do
  line="$B"
  for env in "" "UNET_X3_T448=0 UNET_X3_T448_C4=0" "UNET_X3_T448=0 UNET_X3_T448_C4=0 UNET_X3_R512=0"; do
    r=$(env $env timeout -k 10 300 python bench.py --batch $B --steps 12 --warmup 3 --other-tier-steps 0 --q8-steps 0 --latency-iters 0 \
        --bf16-steps 0 --int8-steps 0 --large-steps 0 --train-steps 0 --no-cpu-baseline --no-check 2>/dev/null | \
        python -c "import sys,json; l=json.loads([x for x in sys.stdin if x.startswith('{')][-1]); print('%.3f ms %.0f fps' % (l['ms_per_step'], l['value']))")
    line="$line | $r"
  done
  echo "$line" | tee -a $OUT
done
