#!/bin/bash
# tools/pmc_pair_ab.sh -- FETCH_SIZE (rocprofv3 --pmc, kernel trace only) of the IVFFlat filter sweeps at nb = 10M with the lock-step
# pair sweeps (ivf_lm_filter.hip PAIR, round 6) off and on: profiles/r6_ab_pmc_ivfflat_10m_pair{0,1}.{txt,json}
R=${GRAFT_REPO_ROOT:-/root/repo}; O=$R/gpurun_out; cd /tmp && export TMPDIR=/tmp
for on in 0 1; do
  LMF_PAIR=$on timeout 600 rocprofv3 --pmc FETCH_SIZE --kernel-trace --output-format csv -d $O/pair${on}_pmc -o p -- python $R/tools/ivfflat_only.py 3 10000000 > $O/pair${on}_pmc.log 2>&1
  python $R/tools/pmc_summary.py $O/r6_ab_pmc_ivfflat_10m_pair$on.txt $O/r6_ab_pmc_ivfflat_10m_pair$on.json $O/pair${on}_pmc:ivf_lmf | cut -c1-230
  rm -rf $O/pair${on}_pmc
done
