#!/bin/bash
# Round-end measurement set (run on the GPU box through gpurun): the two PMC traffic passes first (so that the bench line that
# follows can quote HBM traffic of exactly the kernels it times), then the default bench line, then rocprofv3 kernel stats of the
# same command (multi-stream and single-stream).  Outputs under gpurun_out/final/.   usage: collect_profiles.sh [vN]
set -o pipefail
V=${1:-v3}
cd /tmp && export TMPDIR=/tmp
R=${GRAFT_REPO_ROOT:-/root/repo}
O=$R/gpurun_out/final
rm -rf $O && mkdir -p $O && cd $R
FF_STREAMS=0 timeout -k 10 300 rocprofv3 --kernel-trace --pmc FETCH_SIZE --output-format csv -d $O/pmc_f -- python3 bench.py --steps 1 --warmup 0 --no-graph --no-cpu-baseline --no-extra > $O/pmc_f.json 2> $O/pmc_f.err || exit 1
echo "[collect] pmc fetch done"
FF_STREAMS=0 timeout -k 10 300 rocprofv3 --kernel-trace --pmc WRITE_SIZE --output-format csv -d $O/pmc_w -- python3 bench.py --steps 1 --warmup 0 --no-graph --no-cpu-baseline --no-extra > $O/pmc_w.json 2> $O/pmc_w.err || exit 1
echo "[collect] pmc write done"
FF_GIT_HASH=${FF_GIT_HASH:-n/a} python3 tools/pmc_traffic.py $O/pmc_f $O/pmc_w $O/pmc_hbm_traffic.json || exit 1
cp $O/pmc_hbm_traffic.json $R/profiles/r03_pmc_hbm_traffic_$V.json       # (in this box's copy: bench.py reads the newest matching file)
python3 bench.py > $O/bench_default.json 2> $O/bench_default.err || exit 1
echo "[collect] bench done"
timeout -k 10 300 rocprofv3 --kernel-trace --stats --output-format csv -d $O/stats_ms -- python3 bench.py --steps 5 --warmup 2 --no-cpu-baseline --no-extra > $O/bench_under_rocprof.json 2> $O/stats_ms.err || exit 1
echo "[collect] stats (multi-stream) done"
FF_STREAMS=0 timeout -k 10 300 rocprofv3 --kernel-trace --stats --output-format csv -d $O/stats_ss -- python3 bench.py --steps 5 --warmup 2 --no-cpu-baseline --no-extra > $O/bench_under_rocprof_ss.json 2> $O/stats_ss.err || exit 1
echo "[collect] stats (single stream) done"
# the plugin's fp32-grade mode (bf16x3), single stream: the like-for-like successor of profiles/r02_*_singlestream.csv
FF_STREAMS=0 timeout -k 10 300 rocprofv3 --kernel-trace --stats --output-format csv -d $O/stats_ss_x3 -- python3 bench.py --dtype bf16x3 --steps 5 --warmup 2 --no-cpu-baseline --no-extra > $O/bench_under_rocprof_ss_bf16x3.json 2> $O/stats_ss_x3.err || exit 1
python3 tools/kernel_stats_per_forward.py $O/stats_ss_x3 $O/kernels_per_forward_singlestream_bf16x3.csv || exit 1
echo "[collect] stats (bf16x3, single stream) done"
# per-forward tables of the library's own kernels, setup (ATen weight preparation, runtime copies) listed apart
python3 tools/kernel_stats_per_forward.py $O/stats_ms $O/kernels_per_forward_multistream.csv || exit 1
python3 tools/kernel_stats_per_forward.py $O/stats_ss $O/kernels_per_forward_singlestream.csv || exit 1
# the training step (BASELINE config 5) under the same profiler
timeout -k 10 300 rocprofv3 --kernel-trace --stats --output-format csv -d $O/stats_train -- python3 tools/train_bench.py --steps 5 --warmup 2 > $O/train_bench.txt 2> $O/stats_train.err || exit 1
python3 tools/kernel_stats_per_forward.py $O/stats_train $O/kernels_per_train_step.csv 7 || exit 1
echo "[collect] training-step stats done"
# keep the merged-back payload small (gpurun copies back at most 64 MiB): traces and raw counter dumps are large, the stats
# and the traffic summary are what is committed
find $O -name "*kernel_trace.csv" -delete
find $O -name "*counter_collection.csv" -delete
du -sh $O
