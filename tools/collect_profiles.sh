#!/bin/bash
# Round-end measurement set (run on the GPU box through gpurun): default bench line, rocprofv3 kernel stats of the same
# command (multi-stream and single-stream) and the two PMC traffic passes.  Outputs under gpurun_out/final/.
set -o pipefail
cd /tmp && export TMPDIR=/tmp
R=${GRAFT_REPO_ROOT:-/root/repo}
O=$R/gpurun_out/final
rm -rf $O && mkdir -p $O && cd $R
python3 bench.py > $O/bench_default.json 2> $O/bench_default.err || exit 1
echo "[collect] bench done"
rocprofv3 --kernel-trace --stats --output-format csv -d $O/stats_ms -- python3 bench.py --steps 5 --warmup 2 --no-cpu-baseline --no-extra > $O/bench_under_rocprof.json 2> $O/stats_ms.err || exit 1
echo "[collect] stats (multi-stream) done"
FF_STREAMS=0 rocprofv3 --kernel-trace --stats --output-format csv -d $O/stats_ss -- python3 bench.py --steps 5 --warmup 2 --no-cpu-baseline --no-extra > $O/bench_under_rocprof_ss.json 2> $O/stats_ss.err || exit 1
echo "[collect] stats (single stream) done"
FF_STREAMS=0 rocprofv3 --kernel-trace --pmc FETCH_SIZE --output-format csv -d $O/pmc_f -- python3 bench.py --steps 1 --warmup 0 --no-graph --no-cpu-baseline --no-extra > $O/pmc_f.json 2> $O/pmc_f.err || exit 1
echo "[collect] pmc fetch done"
FF_STREAMS=0 rocprofv3 --kernel-trace --pmc WRITE_SIZE --output-format csv -d $O/pmc_w -- python3 bench.py --steps 1 --warmup 0 --no-graph --no-cpu-baseline --no-extra > $O/pmc_w.json 2> $O/pmc_w.err || exit 1
echo "[collect] pmc write done"
FF_GIT_HASH=${FF_GIT_HASH:-n/a} python3 tools/pmc_traffic.py $O/pmc_f $O/pmc_w $O/pmc_hbm_traffic.json
# keep the merged-back payload small (gpurun copies back at most 64 MiB): traces and raw counter dumps are large, the stats
# and the traffic summary are what is committed
find $O -name "*kernel_trace.csv" -delete
find $O -name "*counter_collection.csv" -delete
du -sh $O
