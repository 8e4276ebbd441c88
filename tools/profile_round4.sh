#!/bin/bash
# tools/profile_round4.sh TAG -- the rocprofv3 passes behind profiles/<TAG>_* (run on the GPU box through gpurun):
#   1. kernel trace + stats of the DEFAULT bench command (the driver's --steps 20 --warmup 5, all legs) + per-dispatch
#      times of the dominant kernels;
#   2. one --pmc pass per counter group (never combined with runtime / sys tracing) on the search loop of every bench leg:
#      flat, IVFPQ nb=1M (query-major), IVFFlat nb=1M / 10M and IVFPQ nb=10M / 100M (list-major behind the f16 filter),
#      IVF-SQ8 nb=1M (list-major, f32 pipe).  bench.py reads its roofline.traffic blocks from the JSON summaries.
TAG=${1:-r05}
R=${GRAFT_REPO_ROOT:-/root/repo}
O=$R/gpurun_out
cd /tmp && export TMPDIR=/tmp
rocprofv3 --kernel-trace --stats --output-format csv -d $O/${TAG}_kt -o kt -- python $R/bench.py --gpus 1 --steps 20 --warmup 5 > $O/${TAG}_profiled_bench.log 2>&1
grep '^{' $O/${TAG}_profiled_bench.log | tail -1 > $O/${TAG}_profiled_bench_line.json
find $O/${TAG}_kt -name "*kernel_stats.csv" -exec cp {} $O/${TAG}_bench_kernel_stats.csv \;
python $R/tools/dispatch_summary.py $O/${TAG}_kt $O/${TAG}_dominant_kernel_dispatches.csv flat_filter_kernel ivfpq_fused_kernel ivf_lmf_flat_kernel ivf_lmf_pq_kernel lmf_rerank ivf_lm_flat_reg_kernel select_k_kernel wave_select_kernel
rm -rf $O/${TAG}_kt
G_FETCH="FETCH_SIZE"
G_SQ="SQ_VALU_MFMA_BUSY_CYCLES SQ_BUSY_CYCLES SQ_WAVE_CYCLES SQ_INSTS_VALU SQ_INSTS_LDS SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_ACTIVE_INST_LDS GRBM_GUI_ACTIVE"
pass() { # name script nb filter groups...
  local name=$1 script=$2 nb=$3 sub=$4; shift 4
  local i=0 dirs=""
  for grp in "$@"; do
    i=$((i + 1))
    timeout 600 rocprofv3 --pmc $grp --kernel-trace --output-format csv -d $O/${TAG}_${name}_pmc$i -o p -- python $R/tools/$script 3 $nb > $O/${TAG}_${name}_pmc$i.log 2>&1
    dirs="$dirs $O/${TAG}_${name}_pmc$i:$sub"
  done
  python $R/tools/pmc_summary.py $O/${TAG}_pmc_${name}.txt $O/${TAG}_pmc_${name}.json $dirs | cut -c1-200 | tail -6
  rm -rf $O/${TAG}_${name}_pmc[0-9]
}
pass flat flat_only.py 1000000 flat_ "$G_FETCH" "$G_SQ"
pass ivfpq_1m ivfpq_only.py 1000000 ivf "$G_FETCH" "$G_SQ"
pass ivfflat_1m ivfflat_only.py 1000000 ivf_lm "$G_FETCH" "$G_SQ"
pass ivfsq_1m ivfsq_only.py 1000000 ivf_lm "$G_FETCH" "$G_SQ"
pass ivfflat_10m ivfflat_only.py 10000000 ivf_lm "$G_FETCH" "$G_SQ"
pass ivfpq_10m ivfpq_only.py 10000000 ivf_lm "$G_FETCH" "$G_SQ"
pass ivfpq_100m ivfpq_only.py 100000000 ivf_lm "$G_FETCH" "$G_SQ"
head -16 $O/${TAG}_bench_kernel_stats.csv | cut -c1-200
