cd /tmp && export TMPDIR=/tmp
rocprofv3 --pmc FETCH_SIZE --kernel-trace --output-format csv -d $GRAFT_REPO_ROOT/gpurun_out/hbm_f -o p -- python3 $GRAFT_REPO_ROOT/bench.py --no-cpu-baseline --no-fp32-mfma-leg --no-kernel-events --steps 3 --warmup 1 > /dev/null 2>&1
rocprofv3 --pmc WRITE_SIZE --kernel-trace --output-format csv -d $GRAFT_REPO_ROOT/gpurun_out/hbm_w -o p -- python3 $GRAFT_REPO_ROOT/bench.py --no-cpu-baseline --no-fp32-mfma-leg --no-kernel-events --steps 3 --warmup 1 > /dev/null 2>&1
echo done
