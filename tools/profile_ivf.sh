#!/bin/bash
# tools/profile_ivf.sh TAG -- rocprofv3 passes on the fused IVF kernels (IVF4096, nprobe 32, nb = 1M, 10k queries):
# kernel trace + stats, then one --pmc pass per counter group (never combined with runtime/sys tracing).
TAG=${1:-r02_a}
R=${GRAFT_REPO_ROOT:-/root/repo}
O=$R/gpurun_out
cd /tmp && export TMPDIR=/tmp
for w in ivfpq ivfflat; do
  rocprofv3 --kernel-trace --stats --output-format csv -d $O/${TAG}_${w}_kt -o kt -- python $R/tools/${w}_only.py 3 > $O/${TAG}_${w}_kt.log 2>&1
  find $O/${TAG}_${w}_kt -name "*kernel_stats.csv" -exec cp {} $O/${TAG}_${w}_kernel_stats.csv \;
  i=0
  for grp in "FETCH_SIZE" "WRITE_SIZE" "SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_INSTS_LDS SQ_ACTIVE_INST_LDS SQ_WAIT_INST_LDS SQ_INSTS_VALU SQ_WAVE_CYCLES SQ_BUSY_CYCLES" "SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_VALU SQ_INSTS_VMEM_RD SQ_INSTS_SALU SQ_WAVES GRBM_GUI_ACTIVE"; do
    i=$((i + 1))
    rocprofv3 --pmc $grp --kernel-trace --output-format csv -d $O/${TAG}_${w}_pmc$i -o p -- python $R/tools/${w}_only.py 2 > $O/${TAG}_${w}_pmc$i.log 2>&1
  done
  python $R/tools/pmc_summary.py $O/${TAG}_${w}_pmc_counters.txt $O/${TAG}_${w}_pmc_counters.json $O/${TAG}_${w}_pmc1:${w}_fused $O/${TAG}_${w}_pmc2:${w}_fused $O/${TAG}_${w}_pmc3:${w}_fused $O/${TAG}_${w}_pmc4:${w}_fused | cut -c1-200
  tail -3 $O/${TAG}_${w}_kt.log
done
