#!/bin/bash
# tools/profile_bench.sh TAG -- kernel trace + stats of the default bench.py command only (the first step of
# tools/profile_round.sh, without the PMC passes): profiles/<TAG>_bench_line.json, _bench_kernel_stats.csv,
# _dominant_kernel_dispatches.csv.  Run on the GPU box through gpurun.
TAG=${1:-r02_g}
R=${GRAFT_REPO_ROOT:-/root/repo}
O=$R/gpurun_out
mkdir -p $O
cd /tmp && export TMPDIR=/tmp
rocprofv3 --kernel-trace --stats --output-format csv -d $O/${TAG}_kt -o kt -- python $R/bench.py --steps 10 --warmup 2 > $O/${TAG}_bench.log 2>&1
grep '^{' $O/${TAG}_bench.log | tail -1 > $O/${TAG}_bench_line.json
find $O/${TAG}_kt -name "*kernel_stats.csv" -exec cp {} $O/${TAG}_bench_kernel_stats.csv \;
python $R/tools/dispatch_summary.py $O/${TAG}_kt $O/${TAG}_dominant_kernel_dispatches.csv flat_filter_kernel flat_rerank_kernel ivfpq_fused_kernel ivfflat_fused_kernel ivfsq_fused_kernel ivf_finish_kernel
head -16 $O/${TAG}_bench_kernel_stats.csv | cut -c1-180
tail -3 $O/${TAG}_bench.log | cut -c1-600
rm -rf $O/${TAG}_kt
