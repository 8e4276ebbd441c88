#!/bin/bash
# tools/profile_round.sh TAG -- the rocprofv3 passes behind profiles/<TAG>_* (run on the GPU box through gpurun):
#   kernel trace + stats of the default bench.py command (scale leg ivfflat_10m included), and one --pmc pass per counter
#   group on the search loops of the bench legs (counter passes carry --kernel-trace only, never runtime / sys tracing):
#   flat, IVFPQ nb=1M (query-major), IVFFlat nb=1M and IVF-SQ8 nb=1M (list-major), IVFFlat nb=10M and IVFPQ nb=10M (list-major).
TAG=${1:-r03_f}
R=${GRAFT_REPO_ROOT:-/root/repo}
O=$R/gpurun_out
cd /tmp && export TMPDIR=/tmp
rocprofv3 --kernel-trace --stats --output-format csv -d $O/${TAG}_kt -o kt -- python $R/bench.py --steps 10 --warmup 2 --scale-legs ivfflat_10m > $O/${TAG}_bench.log 2>&1
grep '^{' $O/${TAG}_bench.log | tail -1 > $O/${TAG}_bench_line.json
find $O/${TAG}_kt -name "*kernel_stats.csv" -exec cp {} $O/${TAG}_bench_kernel_stats.csv \;
python $R/tools/dispatch_summary.py $O/${TAG}_kt $O/${TAG}_dominant_kernel_dispatches.csv flat_filter_kernel ivfpq_fused_kernel ivf_lm_flat_reg_kernel ivf_lm_scan_kernel ivf_lm_pq_kernel ivfsq_fused_kernel select_k_kernel
G_FETCH="FETCH_SIZE"
G_WRITE="WRITE_SIZE"
G_SQ="SQ_VALU_MFMA_BUSY_CYCLES SQ_BUSY_CYCLES SQ_WAVE_CYCLES SQ_INSTS_VALU SQ_INSTS_LDS SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_ACTIVE_INST_LDS GRBM_GUI_ACTIVE"
G_WAIT="SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_WAIT_INST_LDS SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_VALU SQ_INSTS_VMEM_RD SQ_INSTS_SALU SQ_WAVES"
pass() { # name script nb filter groups...
  local name=$1 script=$2 nb=$3 sub=$4; shift 4
  local i=0 dirs=""
  for grp in "$@"; do
    i=$((i + 1))
    timeout 400 rocprofv3 --pmc $grp --kernel-trace --output-format csv -d $O/${TAG}_${name}_pmc$i -o p -- python $R/tools/$script 3 $nb > $O/${TAG}_${name}_pmc$i.log 2>&1
    dirs="$dirs $O/${TAG}_${name}_pmc$i:$sub"
  done
  python $R/tools/pmc_summary.py $O/${TAG}_pmc_${name}.txt $O/${TAG}_pmc_${name}.json $dirs | cut -c1-230
  rm -rf $O/${TAG}_${name}_pmc[0-9]
}
pass flat flat_only.py 1000000 flat_ "$G_FETCH" "$G_WRITE" "$G_SQ" "$G_WAIT"
pass ivfpq_1m ivfpq_only.py 1000000 ivf "$G_FETCH" "$G_SQ"
pass ivfflat_1m ivfflat_only.py 1000000 ivf_lm "$G_FETCH" "$G_SQ"
pass ivfsq_1m ivfsq_only.py 1000000 ivf_lm "$G_FETCH" "$G_SQ"
pass ivfflat_10m ivfflat_only.py 10000000 ivf_lm "$G_FETCH" "$G_SQ"
pass ivfpq_10m ivfpq_only.py 10000000 ivf_lm "$G_FETCH" "$G_SQ"
head -16 $O/${TAG}_bench_kernel_stats.csv | cut -c1-200
# keep the merged-back directory small: the raw traces stay on the box
rm -rf $O/${TAG}_kt
