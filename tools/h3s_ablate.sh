#!/bin/bash
# Ablation builds of the fused pointwise backward (timing only: the results of every build but v0 are wrong), each with the in-kernel clock stamps:
# cycles per wave between the stamps and the clock the kernel ran at.  Build here (no GPU needed), run the binaries on the box:
#   tools/h3s_ablate.sh build && gpurun -- 'tools/h3s_ablate.sh run'
set -u
cd "$(dirname "$0")/.."
names=(v0_full v1_nogate v2_nosums_nostage v3_matrix_only v4_noy v5_noprio v6_skeleton)
flags=("" "-DH3S_NOGATE" "-DH3S_NOSUMS -DH3S_NOSTAGE" "-DH3S_NOGATE -DH3S_NOSUMS -DH3S_NOSTAGE" "-DH3S_NOY" "-DH3S_NOPRIO" "-DH3S_NOY -DH3S_NOGATE -DH3S_NOSUMS -DH3S_NOSTAGE")
if [ "${1:-build}" = build ]; then
    for i in "${!names[@]}"; do
        hipcc -O3 --offload-arch=gfx950 -std=c++17 -w -DPROBAV_STAMP_CLOCK -DKB_ONLY_PW ${flags[$i]} ${EXTRA:-} -I proba-v_amd/csrc -I include tools/kbench.hip -o tools/h3s_${names[$i]}.bin &
    done
    wait
    ls -la tools/h3s_*.bin
else
    for n in "${names[@]}"; do echo "== $n"; tools/h3s_$n.bin 30 | grep -v "^-- pass 1" | tail -3; done
fi
