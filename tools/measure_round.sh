#!/bin/bash
# The measurement set archived under profiles/: kernel stats, two PMC passes, HBM passes, bench lines.  Run on the GPU box from the repo root:
#   gpurun -- 'bash tools/measure_round.sh r02'      then   python tools/collect_round.py r02
R=$GRAFT_REPO_ROOT; TAG=${1:-r02}; O=$R/gpurun_out/round_$TAG
rm -rf $O; mkdir -p $O
cd /tmp && export TMPDIR=/tmp
# per-kernel passes (durations, counters, bytes) with the backward-filter kernels on the launch stream: on the side stream (the default,
# mode 2) they run beside the other kernels and every per-kernel figure would be a figure of two kernels
B="--no-cpu-baseline --no-fp32-mfma-leg --no-other-configs --no-kernel-events --side-stream-mode 1"
rocprofv3 --kernel-trace --stats --output-format csv -d $O/stats -o b -- python3 $R/bench.py $B --steps 10 --warmup 3 > $O/stats_bench.log 2>&1
rocprofv3 --pmc SQ_WAVE_CYCLES SQ_VALU_MFMA_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_LDS GRBM_GUI_ACTIVE --kernel-trace --output-format csv -d $O/pmc_a -o p -- python3 $R/bench.py $B --steps 3 --warmup 1 > /dev/null 2>&1
rocprofv3 --pmc SQ_WAVE_CYCLES SQ_INSTS_VALU SQ_INSTS_LDS SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_WAIT_INST_LDS SQ_INSTS_MFMA SQ_INST_CYCLES_VMEM --kernel-trace --output-format csv -d $O/pmc_b -o p -- python3 $R/bench.py $B --steps 3 --warmup 1 > /dev/null 2>&1
rocprofv3 --pmc FETCH_SIZE --kernel-trace --output-format csv -d $O/hbm_f -o p -- python3 $R/bench.py $B --steps 3 --warmup 1 > /dev/null 2>&1
rocprofv3 --pmc WRITE_SIZE --kernel-trace --output-format csv -d $O/hbm_w -o p -- python3 $R/bench.py $B --steps 3 --warmup 1 > /dev/null 2>&1
cd $R
python3 bench.py 2>/dev/null | tail -1 > $O/bench_line.json
python3 bench.py --full-step --no-cpu-baseline --no-fp32-mfma-leg --no-other-configs 2>/dev/null | tail -1 > $O/bench_line_fullstep.json
python3 tools/pmc_summary.py $O/pmc_a > $O/pmc_a.txt 2>&1
python3 tools/pmc_summary.py $O/pmc_b > $O/pmc_b.txt 2>&1
python3 tools/hbm_summary.py $O/hbm_f $O/hbm_w 4 > $O/hbm_traffic.json 2>$O/hbm_err.txt
# keep what is archived small: the kernel-stats csv and the summaries (the raw counter csvs stay on the box)
find $O/stats -name "*kernel_stats.csv" -exec cp {} $O/kernel_stats.csv \;
rm -rf $O/stats $O/pmc_a $O/pmc_b $O/hbm_f $O/hbm_w
ls -la $O
echo measured
