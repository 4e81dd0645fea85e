#!/bin/bash
# In-step A/B of two builds of libprobav_hip.so on ONE box: alternates them under bench.py (headline leg only) and prints ms per step and the
# per-class times.   gpurun -- 'bash tools/ab_step.sh tools/ab/lib_a.so tools/ab/lib_b.so [alternations]'
set -u
cd "$(dirname "$0")/.."
A="$1"; B="$2"; N="${3:-3}"
L=proba-v_amd/csrc/libprobav_hip.so
cp $L /tmp/lib_keep.so
for i in $(seq 1 $N); do
  for v in "$A" "$B"; do
    cp "$v" $L
    python3 bench.py --no-cpu-baseline --no-fp32-mfma-leg --no-other-configs --steps 60 --warmup 20 2>/dev/null | tail -1 | python3 -c "
import json,sys
d=json.loads(sys.stdin.read()); k=d['kernel_classes']
print('%-28s %.4f ms  | pw_fwd %.4f pw_bwd %.4f fwd %.4f bwdd %.4f wgrad %.4f' % ('$v'.split('/')[-1], d['step_ms']['median'], k['conv1x1x1_fwd_x6']['ms_per_step'], k['conv1x1x1_bwd_data_x6']['ms_per_step'], k['conv3x3x3_fwd_x6']['ms_per_step'], k['conv3x3x3_bwd_data_x6']['ms_per_step'], k['conv3x3x3_wgrad_x6']['ms_per_step']))"
  done
done
cp /tmp/lib_keep.so $L
