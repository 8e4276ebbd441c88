#!/bin/bash
# tools/pmc_lmf.sh TAG SCRIPT NB -- rocprofv3 --pmc passes (one counter group per run, --kernel-trace only) on the search
# loop of tools/SCRIPT at nb = NB, summarised for the kernels of the list-major filter path (ivf_lmf_*):
# profiles/<TAG>_pmc_<name>.{txt,json}.  Run on the GPU box through gpurun.
TAG=${1:-r04_e}; SCRIPT=${2:-ivfpq_only.py}; NB=${3:-10000000}; NAME=${4:-ivfpq_10m}
R=${GRAFT_REPO_ROOT:-/root/repo}; O=$R/gpurun_out; cd /tmp && export TMPDIR=/tmp
G_FETCH="FETCH_SIZE"
G_SQ="SQ_VALU_MFMA_BUSY_CYCLES SQ_BUSY_CYCLES SQ_WAVE_CYCLES SQ_INSTS_VALU SQ_INSTS_LDS SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_ACTIVE_INST_LDS GRBM_GUI_ACTIVE"
G_WAIT="SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_WAIT_INST_LDS SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_VALU SQ_INSTS_VMEM_RD SQ_INSTS_SALU SQ_WAVES"
i=0; dirs=""
for grp in "$G_FETCH" "$G_SQ" "$G_WAIT"; do
  i=$((i + 1))
  timeout 600 rocprofv3 --pmc $grp --kernel-trace --output-format csv -d $O/${TAG}_${NAME}_pmc$i -o p -- python $R/tools/$SCRIPT 3 $NB > $O/${TAG}_${NAME}_pmc$i.log 2>&1
  dirs="$dirs $O/${TAG}_${NAME}_pmc$i:ivf_lm"
done
python $R/tools/pmc_summary.py $O/${TAG}_pmc_${NAME}.txt $O/${TAG}_pmc_${NAME}.json $dirs | cut -c1-260
rm -rf $O/${TAG}_${NAME}_pmc[0-9]
