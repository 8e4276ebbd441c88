R=${GRAFT_REPO_ROOT:-/root/repo}; O=$R/gpurun_out; cd /tmp && export TMPDIR=/tmp
G1="SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_WAIT_INST_LDS SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_VALU SQ_INSTS_VMEM_RD SQ_INSTS_SALU SQ_WAVES"
G2="SQ_ACTIVE_INST_SCA SQ_ACTIVE_INST_VMEM SQ_ACTIVE_INST_MISC SQ_INSTS_SMEM SQ_INSTS_VALU SQ_INSTS_MFMA SQ_WAVE_CYCLES SQ_BUSY_CYCLES"
G4="FETCH_SIZE"
G3="SQ_INST_CYCLES_VMEM_RD SQ_INST_CYCLES_SMEM SQ_IFETCH SQ_INSTS_BRANCH SQ_WAIT_INST_ANY GRBM_GUI_ACTIVE SQ_VALU_MFMA_BUSY_CYCLES SQ_ACTIVE_INST_FLAT"
i=0; dirs=""
for grp in "$G1" "$G2" "$G3" "$G4"; do i=$((i+1)); timeout 300 rocprofv3 --pmc $grp --kernel-trace --output-format csv -d $O/x_pmc$i -o p -- python $R/tools/${SCRIPT:-ivfflat_only.py} 3 1000000 > $O/x_pmc$i.log 2>&1; dirs="$dirs $O/x_pmc$i:ivf_lm"; done
python $R/tools/pmc_summary.py $O/${OUTN:-s9_pmc_ivfflat_1m_extra}.txt $O/${OUTN:-s9_pmc_ivfflat_1m_extra}.json $dirs | cut -c1-200
rm -rf $O/x_pmc[0-9]
