#!/bin/bash
# In-step A/B of two BUILDS on ONE box: alternates bench.py (headline leg only) on proba-v_amd/csrc/prev_libprobav_hip.so (a copy of an earlier build, kept beside the
# library: *.so travels with gpurun, stays out of git) and on the current build; prints ms per step and the per-class times.
#   cp proba-v_amd/csrc/libprobav_hip.so proba-v_amd/csrc/prev_libprobav_hip.so   (before rebuilding);   gpurun -- 'bash tools/ab_lib.sh [alternations]'
set -u
cd "$(dirname "$0")/.."
N="${1:-3}"
C=proba-v_amd/csrc
cp $C/libprobav_hip.so /tmp/ab_new.so
for i in $(seq 1 $N); do
  for v in prev new; do
    if [ $v = prev ]; then cp $C/prev_libprobav_hip.so $C/libprobav_hip.so; else cp /tmp/ab_new.so $C/libprobav_hip.so; fi
    python3 bench.py --no-cpu-baseline --no-fp32-mfma-leg --no-other-configs --steps 60 --warmup 20 2>/dev/null | tail -1 | python3 -c "
import json,sys
d=json.loads(sys.stdin.read()); k=d['kernel_classes']
print('%-6s %.4f ms (median %.4f) sustained %.0f | pw_fwd %.4f pw_bwd %.4f fwd %.4f bwdd %.4f wgrad %.4f' % ('$v', d['ms_per_step'], d['step_ms']['median'], d['sustained_mfma']['tflops'], k['conv1x1x1_fwd_x6']['ms_per_step'], k['conv1x1x1_bwd_data_x6']['ms_per_step'], k['conv3x3x3_fwd_x6']['ms_per_step'], k['conv3x3x3_bwd_data_x6']['ms_per_step'], k['conv3x3x3_wgrad_x6']['ms_per_step']))"
  done
done
cp /tmp/ab_new.so $C/libprobav_hip.so
