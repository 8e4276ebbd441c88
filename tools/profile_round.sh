#!/bin/bash
# tools/profile_round.sh TAG -- the rocprofv3 passes behind profiles/<TAG>_* (run on the GPU box through gpurun):
#   kernel trace + stats of the default bench.py command, and one --pmc pass per counter group on the flat / IVFPQ /
#   IVFFlat / IVFSQ search loops (counter passes carry --kernel-trace only, never runtime / sys tracing).
TAG=${1:-r02_d}
R=${GRAFT_REPO_ROOT:-/root/repo}
O=$R/gpurun_out
cd /tmp && export TMPDIR=/tmp
rocprofv3 --kernel-trace --stats --output-format csv -d $O/${TAG}_kt -o kt -- python $R/bench.py --steps 10 --warmup 2 > $O/${TAG}_bench.log 2>&1
grep '^{' $O/${TAG}_bench.log | tail -1 > $O/${TAG}_bench_line.json
find $O/${TAG}_kt -name "*kernel_stats.csv" -exec cp {} $O/${TAG}_bench_kernel_stats.csv \;
python $R/tools/dispatch_summary.py $O/${TAG}_kt $O/${TAG}_dominant_kernel_dispatches.csv flat_filter_kernel ivfpq_fused_kernel ivfflat_fused_kernel ivfsq_fused_kernel ivf_finish_kernel
dirs=""
for w in flat ivfpq ivfflat ivfsq; do
  i=0
  for grp in "FETCH_SIZE" "WRITE_SIZE" "SQ_VALU_MFMA_BUSY_CYCLES SQ_BUSY_CYCLES SQ_WAVE_CYCLES SQ_INSTS_VALU SQ_INSTS_LDS SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_ACTIVE_INST_LDS GRBM_GUI_ACTIVE" "SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_WAIT_INST_LDS SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_VALU SQ_INSTS_VMEM_RD SQ_INSTS_SALU SQ_WAVES"; do
    i=$((i + 1))
    rocprofv3 --pmc $grp --kernel-trace --output-format csv -d $O/${TAG}_${w}_pmc$i -o p -- python $R/tools/${w}_only.py 3 > $O/${TAG}_${w}_pmc$i.log 2>&1
    sub=$w; [ $w = flat ] && sub=flat_; [ $w != flat ] && sub=${w}_fused
    dirs="$dirs $O/${TAG}_${w}_pmc$i:$sub"
  done
done
python $R/tools/pmc_summary.py $O/${TAG}_pmc_counters.txt $O/${TAG}_pmc_counters.json $dirs | grep -i "filter_kernel\|rerank\|fused" | cut -c1-220
head -14 $O/${TAG}_bench_kernel_stats.csv | cut -c1-200
# keep the merged-back directory small: the raw traces stay on the box
rm -rf $O/${TAG}_kt $O/${TAG}_*_pmc[0-9]
