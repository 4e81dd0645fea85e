#!/bin/bash
# The measurement set archived under profiles/: kernel stats, two PMC passes, HBM passes, bench lines.  Run on the GPU box from the repo root:
#   gpurun -- 'bash tools/measure_round.sh r03'      then   python tools/collect_round.py r03
set -euo pipefail
R="${GRAFT_REPO_ROOT:-$(cd "$(dirname "$0")/.." && pwd)}"
TAG="${1:-r03}"
O="$R/gpurun_out/round_$TAG"
case "$O" in */gpurun_out/round_*) ;; *) echo "refusing to clear '$O'" >&2; exit 2;; esac
rm -rf "$O"; mkdir -p "$O"
cd /tmp && export TMPDIR=/tmp
# per-kernel passes (durations, counters, bytes) with the backward-filter kernels on the launch stream: on the side stream (the default,
# mode 2) they run beside the other kernels and every per-kernel figure would be a figure of two kernels
B="--no-cpu-baseline --no-fp32-mfma-leg --no-other-configs --no-kernel-events --no-power --spin-up 0 --side-stream-mode 1"      # (--spin-up 0: the summaries divide by the steps the command line names)
prof() {            # prof <out dir> <log> <rocprofv3 args...> -- the profiled program goes DIRECTLY after `--` (no env / bash hop)
    local d="$1" log="$2"; shift 2
    if ! rocprofv3 "$@" --kernel-trace --output-format csv -d "$d" -o p -- python3 "$R/bench.py" $B --steps "${STEPS:-3}" --warmup "${WARM:-1}" > "$log" 2>&1; then
        echo "rocprofv3 pass failed: see $log" >&2; tail -5 "$log" >&2; return 1
    fi
}
STEPS=10 WARM=3 prof "$O/stats" "$O/stats_bench.log" --stats
prof "$O/pmc_a" "$O/pmc_a.log" --pmc SQ_WAVE_CYCLES SQ_VALU_MFMA_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_LDS GRBM_GUI_ACTIVE
prof "$O/pmc_b" "$O/pmc_b.log" --pmc SQ_WAVE_CYCLES SQ_INSTS_VALU SQ_INSTS_LDS SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_WAIT_INST_LDS SQ_INSTS_MFMA SQ_INST_CYCLES_VMEM
prof "$O/hbm_f" "$O/hbm_f.log" --pmc FETCH_SIZE
prof "$O/hbm_w" "$O/hbm_w.log" --pmc WRITE_SIZE
cd "$R"
python3 bench.py 2>"$O/bench_line.err" | tail -1 > "$O/bench_line.json"
python3 bench.py --full-step --no-cpu-baseline --no-fp32-mfma-leg --no-other-configs 2>"$O/bench_line_fullstep.err" | tail -1 > "$O/bench_line_fullstep.json"
python3 tools/pmc_summary.py "$O/pmc_a" > "$O/pmc_a.txt"
python3 tools/pmc_summary.py "$O/pmc_b" > "$O/pmc_b.txt"
python3 tools/hbm_summary.py "$O/hbm_f" "$O/hbm_w" 4 > "$O/hbm_traffic.json"
# keep what is archived small: the kernel-stats csv and the summaries (the raw counter csvs stay on the box)
find "$O/stats" -name "*kernel_stats.csv" -exec cp {} "$O/kernel_stats.csv" \;
python3 tools/step_timeline.py "$O/stats" > "$O/step_timeline.txt" || true
test -s "$O/kernel_stats.csv" && test -s "$O/bench_line.json" && test -s "$O/hbm_traffic.json"
rm -rf "$O/stats" "$O/pmc_a" "$O/pmc_b" "$O/hbm_f" "$O/hbm_w"
ls -la "$O"
echo measured
