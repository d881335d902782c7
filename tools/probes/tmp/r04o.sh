set -o pipefail
mkdir -p gpurun_out/r04o
python -c "import __graft_entry__ as g; g.smoke()" 2>&1 | grep -v amdgpu.ids | tail -6
for env in "" "UNET_X3_T448=0"; do
  echo "== 640x640 leg, env: $env"
  env $env timeout -k 10 600 python bench.py --steps 2 --warmup 1 --other-tier-steps 0 --q8-steps 0 --latency-iters 0 --bf16-steps 0 --int8-steps 0 --large-steps 4 --train-steps 0 --no-cpu-baseline --no-check 2>/dev/null | python -c "import sys,json; l=json.loads([x for x in sys.stdin if x.startswith('{')][-1]); print(json.dumps(l.get('large_input')))"
done
