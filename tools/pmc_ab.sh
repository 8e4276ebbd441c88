#!/bin/bash
# tools/pmc_ab.sh TAG NB -- SQ counters of the IVFPQ filter sweeps with the two-copy codebook on / off (same box, same data)
TAG=${1:-r5}; NB=${2:-10000000}
R=${GRAFT_REPO_ROOT:-/root/repo}; O=$R/gpurun_out; cd /tmp && export TMPDIR=/tmp
G_SQ="SQ_VALU_MFMA_BUSY_CYCLES SQ_BUSY_CYCLES SQ_WAVE_CYCLES SQ_INSTS_VALU SQ_INSTS_LDS SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_ACTIVE_INST_LDS GRBM_GUI_ACTIVE"
G_WAIT="SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_WAIT_INST_LDS SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_VALU SQ_INSTS_VMEM_RD SQ_INSTS_SALU SQ_WAVES"
for two in 1 0; do
  i=0; dirs=""
  for grp in "$G_SQ" "$G_WAIT"; do
    i=$((i + 1))
    TWO_COPIES=$two timeout 600 rocprofv3 --pmc $grp --kernel-trace --output-format csv -d $O/${TAG}_ab${two}_pmc$i -o p -- python $R/tools/ivfpq_only.py 3 $NB > $O/${TAG}_ab${two}_pmc$i.log 2>&1
    dirs="$dirs $O/${TAG}_ab${two}_pmc$i:ivf_lmf_pq"
  done
  python $R/tools/pmc_summary.py $O/${TAG}_pmc_two_copies_${two}.txt $O/${TAG}_pmc_two_copies_${two}.json $dirs > /dev/null
  rm -rf $O/${TAG}_ab${two}_pmc[0-9]
done
python - <<PY
import json
for two in (1, 0):
    js = json.load(open("$O/${TAG}_pmc_two_copies_%d.json" % two))
    for k, e in js.items():
        if "SQ_BUSY_CYCLES" not in e: continue
        print("two_copies=%d %s: %.3f ms  LDS active/busy %.2f  conflict/idx_active %.2f  insts LDS %.3g VALU %.3g  mfma busy/busy*4 %.2f  wait_inst_lds/wave_cycles %.2f" % (
            two, k.split("(")[0][-48:], e["avg_duration_ns"] / 1e6, e["SQ_ACTIVE_INST_LDS"] / e["SQ_BUSY_CYCLES"],
            e["SQ_LDS_BANK_CONFLICT"] / max(e["SQ_LDS_IDX_ACTIVE"], 1), e["SQ_INSTS_LDS"], e["SQ_INSTS_VALU"],
            e["SQ_VALU_MFMA_BUSY_CYCLES"] / e["SQ_BUSY_CYCLES"] / 4, e.get("SQ_WAIT_INST_LDS", 0) / max(e.get("SQ_WAVE_CYCLES", 1), 1)))
PY
