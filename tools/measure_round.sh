#!/bin/bash
# The measurement set archived under profiles/: kernel stats, two PMC passes, bench lines.  Run on the GPU box from the repo root.
R=$GRAFT_REPO_ROOT; O=$R/gpurun_out/round
mkdir -p $O
cd /tmp && export TMPDIR=/tmp
rocprofv3 --kernel-trace --stats --output-format csv -d $O/stats -o b -- python3 $R/bench.py --no-cpu-baseline --no-fp32-mfma-leg --no-kernel-events --steps 10 > $O/stats_bench.log 2>&1
rocprofv3 --pmc SQ_WAVE_CYCLES SQ_VALU_MFMA_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_LDS GRBM_GUI_ACTIVE --kernel-trace --output-format csv -d $O/pmc_a -o p -- python3 $R/bench.py --no-cpu-baseline --no-fp32-mfma-leg --no-kernel-events --steps 3 --warmup 1 > /dev/null 2>&1
rocprofv3 --pmc SQ_WAVE_CYCLES SQ_INSTS_VALU SQ_INSTS_LDS SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_WAIT_INST_LDS SQ_INSTS_MFMA SQ_INST_CYCLES_VMEM --kernel-trace --output-format csv -d $O/pmc_b -o p -- python3 $R/bench.py --no-cpu-baseline --no-fp32-mfma-leg --no-kernel-events --steps 3 --warmup 1 > /dev/null 2>&1
cd $R
python3 bench.py 2>/dev/null | tail -1 > $O/bench_line.json
python3 bench.py --frames 13 --no-cpu-baseline 2>/dev/null | tail -1 > $O/bench_line_t13.json
python3 bench.py --full-step --no-cpu-baseline --no-fp32-mfma-leg 2>/dev/null | tail -1 > $O/bench_line_fullstep.json
python3 tools/bench_infer.py 2>/dev/null | tail -2 > $O/bench_infer.log
echo measured
