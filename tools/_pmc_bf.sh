cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
O=gpurun_out/pmc_bf
rm -rf $O; mkdir -p $O
ARGS="--steps 1 --warmup 1 --no-cpu-baseline --train-steps 0 --latency-iters 0 --bf16-steps 1 --bf16-batch 256"
rocprofv3 --kernel-trace --pmc SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_VALU_MFMA_BUSY_CYCLES SQ_WAVES GRBM_GUI_ACTIVE --output-format csv -d $O/sq -o x -- python3 bench.py $ARGS > $O/log_sq.txt 2>&1 || exit 1
rocprofv3 --kernel-trace --pmc SQ_INSTS_VALU SQ_INSTS_LDS SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR SQ_ACTIVE_INST_LDS SQ_ACTIVE_INST_VMEM SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE --output-format csv -d $O/sq2 -o x -- python3 bench.py $ARGS > $O/log_sq2.txt 2>&1 || exit 1
rocprofv3 --kernel-trace --pmc TCC_HIT_sum TCC_MISS_sum TCC_EA0_RDREQ_sum TCC_REQ_sum --output-format csv -d $O/tcc -o x -- python3 bench.py $ARGS > $O/log_tcc.txt 2>&1 || exit 1
python3 tools/pmc_rows.py $O/sq bf16 > $O/sq.txt
python3 tools/pmc_rows.py $O/sq2 bf16 > $O/sq2.txt
python3 tools/pmc_rows.py $O/tcc bf16 > $O/tcc.txt
find $O -name "*.csv" -size +2M -delete
