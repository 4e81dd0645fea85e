#!/bin/bash
# In-step A/B of an environment switch on ONE box: alternates `env $1 bench.py` and plain bench.py (headline leg only), prints ms per step and the per-class times.
#   gpurun -- 'bash tools/ab_env.sh PROBAV_GEN1=conv [alternations]'
set -u
cd "$(dirname "$0")/.."
V="$1"; N="${2:-3}"
for i in $(seq 1 $N); do
  for v in "$V" "X_UNUSED=1"; do
    env "$v" python3 bench.py --no-cpu-baseline --no-fp32-mfma-leg --no-other-configs --steps 60 --warmup 20 2>/dev/null | tail -1 | python3 -c "
import json,sys
d=json.loads(sys.stdin.read()); k=d['kernel_classes']
print('%-20s %.4f ms (median %.4f) sustained %.0f | pw_fwd %.4f pw_bwd %.4f fwd %.4f bwdd %.4f wgrad %.4f' % ('$v', d['ms_per_step'], d['step_ms']['median'], d['sustained_mfma']['tflops'], k['conv1x1x1_fwd_x6']['ms_per_step'], k['conv1x1x1_bwd_data_x6']['ms_per_step'], k['conv3x3x3_fwd_x6']['ms_per_step'], k['conv3x3x3_bwd_data_x6']['ms_per_step'], k['conv3x3x3_wgrad_x6']['ms_per_step']))"
  done
done
