#!/bin/bash
# Round 4's kernel evidence (profiles/r04_*.txt): the co-issue table in cycles, the instruction cost table, the phase stamps and ablations of the
# fused pointwise backward, and its A/B against round 3's kernel.  Build here (no GPU), run on the box:
#   tools/evidence_round4.sh build && gpurun -- 'tools/evidence_round4.sh run'      then copy gpurun_out/r04_evidence/*.txt to profiles/
set -u
cd "$(dirname "$0")/.."
F="-O3 --offload-arch=gfx950 -std=c++17 -w -I proba-v_amd/csrc -I include"
if [ "${1:-build}" = build ]; then
    hipcc $F tools/coissue_cycles.hip -o tools/coissue_cycles.bin &
    hipcc $F tools/valu_cost.hip -o tools/valu_cost.bin &
    hipcc $F -DPROBAV_STAMP tools/diag_h3t.hip -o tools/diag_h3t.bin &
    hipcc $F -DPROBAV_STAMP -DPROBAV_STAMP_Y tools/diag_h3t.hip -o tools/diag_h3t_y.bin &
    hipcc $F -DPROBAV_STAMP_CLOCK -DKB_ONLY_PW tools/kbench.hip -o tools/kbench_clk.bin &
    wait
    for v in "H3S_NOGATE" "H3S_NOSUMS -DH3S_NOSTAGE" "H3S_NOGATE -DH3S_NOSUMS -DH3S_NOSTAGE" "H3S_NOY" "H3S_NOY -DH3S_NOGATE -DH3S_NOSUMS -DH3S_NOSTAGE" "H3T_NOHSTORE" "H3T_NODHSTORE" "H3T_NOTB" "H3T_PURE" "H3T_LOADS_IN_X"; do
        n=$(echo "$v" | sed 's/ -D/+/g')
        hipcc $F -DPROBAV_STAMP_CLOCK -DKB_ONLY_PW -D$v tools/kbench.hip -o "tools/kbv_$n.bin" &
    done
    wait
    ls tools/kbv_*.bin
else
    O=gpurun_out/r04_evidence; mkdir -p $O
    tools/coissue_cycles.bin > $O/r04_coissue_cycles.txt 2>&1
    tools/valu_cost.bin > $O/r04_instruction_costs.txt 2>&1
    { echo "# tools/diag_h3t.hip: pw_bwd_h3t_kernel (round 4)"; tools/diag_h3t.bin; echo; echo "# the same with five more stamps inside Y"; tools/diag_h3t_y.bin;
      echo; echo "# PROBAV_PW_BWD_H3S=1: pw_bwd_h3s_kernel (round 3)"; PROBAV_PW_BWD_H3S=1 tools/diag_h3t.bin; } > $O/r04_pw_bwd_phases.txt 2>&1
    { echo "# tools/kbench.hip -DPROBAV_STAMP_CLOCK, pw_bwd line only: cycles per wave between the stamps, in-kernel clock; three alternations on one box";
      for i in 1 2 3; do echo -n "round 4 (pw_bwd_h3t_kernel)                : "; tools/kbench_clk.bin 30 | tail -1; echo -n "round 3 (pw_bwd_h3s_kernel, PROBAV_PW_BWD_H3S): "; PROBAV_PW_BWD_H3S=1 tools/kbench_clk.bin 30 | tail -1; done
      echo "# ablation builds of pw_bwd_h3t_kernel (timing only: every build but the first computes wrong results)"
      for b in tools/kbv_*.bin; do echo -n "$(basename $b .bin | sed 's/kbv_//') : "; $b 30 | tail -1; done; } > $O/r04_pw_bwd_ablation.txt 2>&1
    ls -la $O
fi
