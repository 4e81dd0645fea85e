cd /tmp && export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
O=$R/gpurun_out/pw4_pmc
rm -rf $O; mkdir -p $O
rocprofv3 --pmc SQ_WAVE_CYCLES SQ_VALU_MFMA_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_LDS GRBM_GUI_ACTIVE --kernel-trace --output-format csv -d $O/a -o p -- $R/tools/pw4t_full.bin 5 > $O/a.log 2>&1
rocprofv3 --pmc SQ_WAVE_CYCLES SQ_INSTS_VALU SQ_INSTS_LDS SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_WAIT_INST_LDS SQ_INSTS_MFMA SQ_INSTS_SALU --kernel-trace --output-format csv -d $O/b -o p -- $R/tools/pw4t_full.bin 5 > $O/b.log 2>&1
python3 - <<PY
import csv,glob,collections
for d in ("a","b"):
    for f in glob.glob("$O/%s/**/*counter_collection.csv"%d, recursive=True):
        agg=collections.defaultdict(lambda: collections.defaultdict(float)); cnt=collections.Counter()
        for r in csv.DictReader(open(f)):
            k=r["Kernel_Name"][:40]
            agg[k][r["Counter_Name"]]+=float(r["Counter_Value"]); 
        for k,v in agg.items():
            print(d,k,{c:round(x) for c,x in v.items()})
PY
