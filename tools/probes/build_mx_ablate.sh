#!/bin/bash
# container: timing-only builds of the f16+q8 probe (one per ablation mask), run by tools/probes/run_mx_ablate.sh
for m in 0 1 2 4 8 16 7 24 32 39; do
  hipcc --offload-arch=gfx950 -O3 -std=c++17 -I unet_lane_detection_amd/csrc -DUNET_MX_STAMPS=1 -DUNET_MX_ABLATE=$m \
        -o tools/probes/conv_mx_r512_probe_a$m tools/probes/conv_mx_r512_probe.hip 2>/dev/null || echo "build $m failed"
done
