#!/bin/bash
# tools/profile_round.sh TAG -- the rocprofv3 passes behind profiles/<TAG>_* (run on the GPU box through gpurun):
#   kernel trace + stats of the default bench.py command, and one --pmc pass per counter group on the
#   flat search loop (never combined with runtime/sys tracing).
TAG=${1:-r01_d}
R=${GRAFT_REPO_ROOT:-/root/repo}
O=$R/gpurun_out
cd /tmp && export TMPDIR=/tmp
rocprofv3 --kernel-trace --stats --output-format csv -d $O/${TAG}_kt -o kt -- python $R/bench.py --steps 10 --warmup 3 > $O/${TAG}_bench.log 2>&1
tail -1 $O/${TAG}_bench.log | grep '^{' > $O/${TAG}_bench_line.json
i=0
for grp in "FETCH_SIZE" "WRITE_SIZE" "SQ_VALU_MFMA_BUSY_CYCLES SQ_BUSY_CYCLES SQ_WAVE_CYCLES SQ_INSTS_VALU SQ_INSTS_LDS SQ_LDS_BANK_CONFLICT SQ_WAIT_INST_ANY SQ_WAIT_ANY"; do
    i=$((i + 1))
    rocprofv3 --pmc $grp --kernel-trace --output-format csv -d $O/${TAG}_pmc$i -o p -- python $R/tools/flat_only.py 3 > $O/${TAG}_pmc$i.log 2>&1
done
# IVFPQ scan: HBM traffic of the fused kernel
rocprofv3 --pmc FETCH_SIZE --kernel-trace --output-format csv -d $O/${TAG}_pmc4 -o p -- python $R/tools/ivfpq_only.py 3 > $O/${TAG}_pmc4.log 2>&1
python $R/tools/pmc_summary.py $O/${TAG}_pmc_counters.txt $O/${TAG}_pmc_counters.json $O/${TAG}_pmc1 $O/${TAG}_pmc2 $O/${TAG}_pmc3 $O/${TAG}_pmc4:ivfpq_fused | grep -i "flat_filter\|rerank\|ivfpq_fused" | cut -c1-230
find $O/${TAG}_kt -name "*kernel_stats.csv" -exec cp {} $O/${TAG}_bench_kernel_stats.csv \;
head -12 $O/${TAG}_bench_kernel_stats.csv | cut -c1-200
