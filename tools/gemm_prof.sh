cd /tmp && export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT; O=$R/gpurun_out/gp; rm -rf $O; mkdir -p $O
for shp in "4096 1024 2048" "4096 1024 1024" "16384 512 1024"; do
  tag=$(echo $shp | tr ' ' '_')
  timeout -k 10 120 rocprofv3 --kernel-trace --stats --output-format csv -d $O/st_$tag -- python3 $R/tools/gemm_prof.py $shp > /dev/null 2>&1 || exit 1
  timeout -k 10 120 rocprofv3 --kernel-trace --pmc SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_WAVE_CYCLES SQ_VALU_MFMA_BUSY_CYCLES SQ_BUSY_CYCLES SQ_INSTS_MFMA SQ_WAIT_INST_LDS --output-format csv -d $O/sq_$tag -- python3 $R/tools/gemm_prof.py $shp > /dev/null 2>&1 || exit 1
  # (a TCC_HIT_sum / TCC_MISS_sum / TCC_EA0_RDREQ_sum pass hung the profiler on this pool: do not add it back)
  echo "[gp] $tag done"
done
cd $O && python3 - <<'PY'
import csv,glob,collections
for d in sorted(glob.glob('*_*')):
    for f in glob.glob(d+'/*/*counter_collection.csv'):
        acc=collections.defaultdict(list)
        for r in csv.DictReader(open(f)):
            if 'conv_igemm' in r['Kernel_Name']: acc[r['Counter_Name']].append(float(r['Counter_Value']))
        print(d, {k: sum(v)/len(v) for k,v in acc.items()})
    for f in glob.glob(d+'/*/*kernel_stats.csv'):
        for r in csv.DictReader(open(f)):
            if 'conv_igemm' in r['Name']: print(d, r['Name'][:60], r['AverageNs'])
PY
find $O -name "*kernel_trace.csv" -delete; find $O -name "*counter_collection.csv" -delete
