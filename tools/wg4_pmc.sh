# rocprofv3 PMC passes over tools/wg4bench.bin (run on the GPU box from the repo root: gpurun -- 'bash tools/wg4_pmc.sh')
cd /tmp && export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
O=$R/gpurun_out/wg4_pmc
rm -rf $O; mkdir -p $O
rocprofv3 --pmc SQ_WAVE_CYCLES SQ_VALU_MFMA_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_LDS GRBM_GUI_ACTIVE --kernel-trace --output-format csv -d $O/a -o p -- $R/tools/wg4bench.bin 3 > $O/a.log 2>&1
rocprofv3 --pmc SQ_WAVE_CYCLES SQ_INSTS_VALU SQ_INSTS_LDS SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_WAIT_INST_LDS SQ_INSTS_MFMA SQ_INSTS_SALU --kernel-trace --output-format csv -d $O/b -o p -- $R/tools/wg4bench.bin 3 > $O/b.log 2>&1
rocprofv3 --pmc SQ_WAVE_CYCLES SQ_INST_CYCLES_VMEM SQ_ACTIVE_INST_VMEM SQ_WAIT_INST_VMEM SQ_LDS_ADDR_CONFLICT SQ_LDS_UNALIGNED_STALL SQ_ACTIVE_INST_MISC SQ_ACTIVE_INST_SCA --kernel-trace --output-format csv -d $O/c -o p -- $R/tools/wg4bench.bin 3 > $O/c.log 2>&1
python3 $R/tools/pmc_summary.py $O/a > $O/summary.txt; python3 $R/tools/pmc_summary.py $O/b >> $O/summary.txt; python3 $R/tools/pmc_summary.py $O/c >> $O/summary.txt
rm -rf $O/a $O/b $O/c
cat $O/summary.txt
