set -o pipefail
mkdir -p gpurun_out/r04k
bash tools/probes/run_t448_probe.sh gpurun_out/r04k/t448_probe_overlap.txt > gpurun_out/r04k/probe_tail.txt 2>&1 || { echo PROBE FAILED; tail -40 gpurun_out/r04k/probe_tail.txt; exit 1; }
bash tools/probes/run_t448_c4_probe.sh gpurun_out/r04k/t448_c4_probe_overlap.txt > gpurun_out/r04k/probe_c4_tail.txt 2>&1 || { echo PROBE C4 FAILED; tail -40 gpurun_out/r04k/probe_c4_tail.txt; exit 1; }
grep "third :\|^-- " gpurun_out/r04k/t448_probe_overlap.txt | tail -16
grep "third :\|second:\|^-- " gpurun_out/r04k/t448_c4_probe_overlap.txt | tail -22
bash tools/probes/run_t448_ab.sh gpurun_out/r04k/overlap_ab.txt conv_x3_t448_probe_stamps_o0 conv_x3_t448_probe_stamps_o1 > gpurun_out/r04k/ab_tail.txt 2>&1; echo "ab rc=$?"
grep "third :\|^-- " gpurun_out/r04k/overlap_ab.txt | tail -60
