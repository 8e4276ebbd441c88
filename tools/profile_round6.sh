#!/bin/bash
# tools/profile_round6.sh TAG -- the rocprofv3 passes behind profiles/<TAG>_* (run on the GPU box through gpurun):
#   1. kernel trace + stats of the DEFAULT bench command (the driver's --steps 20 --warmup 5, all legs);
#   2. PER LEG: kernel trace + stats of the search loop of that leg alone -> <TAG>_<leg>_kernel_stats.csv,
#      one row per kernel (calls, average ns): a `frac` of the bench line can be recomputed from ONE row;
#   3. one --pmc pass per counter group (never combined with runtime / sys tracing) on the same search loops:
#      <TAG>_pmc_<leg>.{txt,json} (bench.py reads its roofline.traffic numbers from the JSON summaries).
TAG=${1:-r6}
MODE=${2:-all} # "stats": the per-leg kernel statistics only
R=${GRAFT_REPO_ROOT:-/root/repo}
O=$R/gpurun_out
cd /tmp && export TMPDIR=/tmp
if [ "$MODE" != "stats" ]; then
rocprofv3 --kernel-trace --stats --output-format csv -d $O/${TAG}_kt -o kt -- python $R/bench.py --gpus 1 --steps 20 --warmup 5 > $O/${TAG}_profiled_bench.log 2>&1
grep '^{' $O/${TAG}_profiled_bench.log | tail -1 > $O/${TAG}_profiled_bench_line.json
find $O/${TAG}_kt -name "*kernel_stats.csv" -exec cp {} $O/${TAG}_bench_kernel_stats.csv \;
rm -rf $O/${TAG}_kt
fi
G_FETCH="FETCH_SIZE"
G_SQ="SQ_VALU_MFMA_BUSY_CYCLES SQ_BUSY_CYCLES SQ_WAVE_CYCLES SQ_INSTS_VALU SQ_INSTS_LDS SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_ACTIVE_INST_LDS GRBM_GUI_ACTIVE"
leg() { # name script nb filter
  local name=$1 script=$2 nb=$3 sub=$4
  timeout 600 rocprofv3 --kernel-trace --stats --output-format csv -d $O/${TAG}_${name}_kt -o kt -- python $R/tools/$script 5 $nb > $O/${TAG}_${name}_kt.log 2>&1
  python - <<PY
import csv, glob
rows = []
for f in glob.glob("$O/${TAG}_${name}_kt/**/*kernel_trace.csv", recursive=True):
    rows += [(r["Kernel_Name"], int(r["Start_Timestamp"]), int(r["End_Timestamp"])) for r in csv.DictReader(open(f))]
rows.sort(key=lambda r: r[1])
# the SEARCH loop only: from 1 ms before the first plan launch of an IVF leg (its coarse quantizer runs ahead of the plan; the
# build ended long before: host-side prints, query upload); the flat leg has no build kernels worth the name
plan = [r[1] for r in rows if "lm_plan_kernel" in r[0]]
t0 = plan[0] - 1000000 if plan else 0
agg = {}
for name, a, b in rows:
    if a >= t0:
        e = agg.setdefault(name, [0, 0, 1 << 62, 0])
        e[0] += 1; e[1] += b - a; e[2] = min(e[2], b - a); e[3] = max(e[3], b - a)
nsearch = max(1, len(plan)) if plan else 6
with open("$O/${TAG}_${name}_kernel_stats.csv", "w") as fo:
    fo.write("# rocprofv3 --kernel-trace of tools/$script 5 $nb: the %d searches of 10 000 queries behind the build (dispatches from the first search on)\n" % nsearch)
    fo.write("kernel,calls,calls_per_search,total_ns,average_ns,min_ns,max_ns\n")
    for k, e in sorted(agg.items(), key=lambda kv: -kv[1][1]):
        fo.write('"%s",%d,%.2f,%d,%.1f,%d,%d\n' % (k[:150], e[0], e[0] / nsearch, e[1], e[1] / e[0], e[2], e[3]))
PY
  rm -rf $O/${TAG}_${name}_kt
  if [ "$MODE" = "stats" ]; then return; fi
  local i=0 dirs=""
  for grp in "$G_FETCH" "$G_SQ"; do
    i=$((i + 1))
    timeout 600 rocprofv3 --pmc $grp --kernel-trace --output-format csv -d $O/${TAG}_${name}_pmc$i -o p -- python $R/tools/$script 3 $nb > $O/${TAG}_${name}_pmc$i.log 2>&1
    dirs="$dirs $O/${TAG}_${name}_pmc$i:$sub"
  done
  python $R/tools/pmc_summary.py $O/${TAG}_pmc_${name}.txt $O/${TAG}_pmc_${name}.json $dirs | cut -c1-200 | tail -4
  rm -rf $O/${TAG}_${name}_pmc[0-9]
}
leg flat flat_only.py 1000000 flat_
leg ivfpq_1m ivfpq_only.py 1000000 ivf
leg ivfflat_1m ivfflat_only.py 1000000 ivf_lm
leg ivfsq_1m ivfsq_only.py 1000000 ivf_lm
leg ivfflat_10m ivfflat_only.py 10000000 ivf_lm
leg ivfpq_10m ivfpq_only.py 10000000 ivf_lm
leg ivfpq_100m ivfpq_only.py 100000000 ivf_lm
ls $O | grep "^${TAG}_" | head -60
