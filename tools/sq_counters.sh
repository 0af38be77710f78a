#!/bin/bash
# SQ counter set of ONE kernel family (two rocprofv3 --pmc passes of eight counters each, means over the launches of the run):
#   bash tools/sq_counters.sh <kernel-name substring> <tag> <python driver + args ...>      -> gpurun_out/sq_<tag>.txt
# e.g.  bash tools/sq_counters.sh win_attn_fused wf_bf16 tools/wf_prof.py 0 bf16
cd /tmp && export TMPDIR=/tmp
R=${GRAFT_REPO_ROOT:-/root/repo}
K=$1; TAG=$2; shift 2
O=$R/gpurun_out/sq_$TAG; rm -rf $O; mkdir -p $O
cd $R
timeout -k 10 180 rocprofv3 --kernel-trace --pmc SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_INSTS_MFMA SQ_VALU_MFMA_BUSY_CYCLES SQ_LDS_BANK_CONFLICT --output-format csv -d $O/a -- python3 "$@" > /dev/null 2>&1 || exit 1
timeout -k 10 180 rocprofv3 --kernel-trace --pmc SQ_LDS_IDX_ACTIVE SQ_INSTS_LDS SQ_INSTS_VALU SQ_INSTS_SALU SQ_WAIT_INST_LDS SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_LDS SQ_WAVES --output-format csv -d $O/b -- python3 "$@" > /dev/null 2>&1 || exit 1
timeout -k 10 180 rocprofv3 --kernel-trace --stats --output-format csv -d $O/c -- python3 "$@" > /dev/null 2>&1 || exit 1
K=$K python3 - > $R/gpurun_out/sq_$TAG.txt <<PY
import csv, glob, collections, os
k = os.environ["K"]
print("rocprofv3 --kernel-trace --pmc <8 SQ counters> (two passes) -- python3 $*   kernels matching '%s'; per launch, whole chip" % k)
for d in ("a", "b"):
    for f in glob.glob("$O/" + d + "/*/*counter_collection.csv"):
        acc = collections.defaultdict(list)
        for r in csv.DictReader(open(f)):
            if k in r["Kernel_Name"]:
                acc[r["Counter_Name"]].append(float(r["Counter_Value"]))
        for c, v in sorted(acc.items()):
            print(f"{c:28s} {sum(v) / len(v):.4e}   (mean of {len(v)} launches)")
for f in glob.glob("$O/c/*/*kernel_stats.csv"):
    for r in csv.DictReader(open(f)):
        if k in r["Name"]:
            print("kernel_stats:", r["Name"][:90], "calls", r["Calls"], "avg ns", r["AverageNs"])
PY
find $O -name "*.csv" -delete
cat $R/gpurun_out/sq_$TAG.txt
