#!/bin/bash
# tools/profile_bench.sh TAG -- rocprofv3 kernel trace + stats of the DEFAULT bench command (all legs, the driver's
# --steps 20 --warmup 5), run on the GPU box through gpurun; outputs under gpurun_out/<TAG>_*.
TAG=${1:-r03_j}
R=${GRAFT_REPO_ROOT:-/root/repo}
O=$R/gpurun_out
cd /tmp && export TMPDIR=/tmp
rocprofv3 --kernel-trace --stats --output-format csv -d $O/${TAG}_kt -o kt -- python $R/bench.py --gpus 1 --steps 20 --warmup 5 > $O/${TAG}_profiled_bench.log 2>&1
grep '^{' $O/${TAG}_profiled_bench.log | tail -1 > $O/${TAG}_profiled_bench_line.json
find $O/${TAG}_kt -name "*kernel_stats.csv" -exec cp {} $O/${TAG}_bench_kernel_stats.csv \;
python $R/tools/dispatch_summary.py $O/${TAG}_kt $O/${TAG}_dominant_kernel_dispatches.csv flat_filter_kernel ivfpq_fused_kernel ivf_lm_flat_reg_kernel ivf_lm_scan_kernel ivf_lm_pq_kernel select_k_kernel wave_select_kernel
head -14 $O/${TAG}_bench_kernel_stats.csv | cut -c1-180
rm -rf $O/${TAG}_kt
