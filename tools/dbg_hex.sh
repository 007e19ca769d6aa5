#!/bin/bash
for m in 4 8 16 20; do
  export GS4D_HEX_DEBUG=$m
  bash tools/prof_kernels.sh hx$m --steps 6 --warmup 2
  echo "mode $m: $(python tools/step_kernels.py "$(find gpurun_out/hx$m -name '*kernel_trace.csv' | head -1)" 4 hexplane_backward_reduce)"
done
