cd /tmp && export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT; O=$R/gpurun_out/nfp; rm -rf $O; mkdir -p $O
timeout -k 10 120 rocprofv3 --kernel-trace --pmc SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_INSTS_MFMA SQ_WAIT_INST_LDS SQ_LDS_BANK_CONFLICT --output-format csv -d $O/a -- python3 $R/tools/naf_front_prof.py 64 1024 > /dev/null 2>&1 || exit 1
timeout -k 10 120 rocprofv3 --kernel-trace --pmc SQ_LDS_IDX_ACTIVE SQ_INSTS_LDS SQ_INSTS_VALU SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_LDS SQ_ACTIVE_INST_VMEM --output-format csv -d $O/b -- python3 $R/tools/naf_front_prof.py 64 1024 > /dev/null 2>&1 || exit 1
cd $O && python3 - <<'PY'
import csv,glob,collections
for d in ('a','b'):
    for f in glob.glob(d+'/*/*counter_collection.csv'):
        acc=collections.defaultdict(list)
        for r in csv.DictReader(open(f)):
            if 'naf_front' in r['Kernel_Name']: acc[r['Counter_Name']].append(float(r['Counter_Value']))
        for k,v in sorted(acc.items()): print(f"{k:28s} {sum(v)/len(v):.4e}")
PY
find $O -name "*.csv" -delete
