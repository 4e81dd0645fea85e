#!/bin/bash
# tools/ab_lib.sh for the T = 13 network (cfg p16t12c85r12's depth): alternates bench.py --frames 13 on prev_libprobav_hip.so and on the current build
set -u
cd "$(dirname "$0")/.."
N="${1:-3}"
C=proba-v_amd/csrc
cp $C/libprobav_hip.so /tmp/ab_new.so
for i in $(seq 1 $N); do
  for v in prev new; do
    if [ $v = prev ]; then cp $C/prev_libprobav_hip.so $C/libprobav_hip.so; else cp /tmp/ab_new.so $C/libprobav_hip.so; fi
    python3 bench.py --frames 13 --no-cpu-baseline --no-fp32-mfma-leg --no-other-configs --steps 40 --warmup 15 2>/dev/null | tail -1 | python3 -c "
import json,sys
d=json.loads(sys.stdin.read()); k=d['kernel_classes']
print('%-6s %.4f ms (median %.4f) %.0f patches/s sustained %.0f | wgrad %.4f' % ('$v', d['ms_per_step'], d['step_ms']['median'], d['value'], d['sustained_mfma']['tflops'], k['conv3x3x3_wgrad_x6']['ms_per_step']))"
  done
done
cp /tmp/ab_new.so $C/libprobav_hip.so
