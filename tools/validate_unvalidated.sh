#!/bin/bash
# tools/validate_unvalidated.sh -- ONE GPU call that runs everything written after round 2's GPU budget was spent:
#   * the gated tests (extra metrics of the flat index, staggered filter schedule, one-launch small-database kernel),
#   * A/B timings: flat search with FAISS_AMD_FILTER_STAGGER = 0 / 1 / 2, IVF4096,PQ64 search with FAISS_AMD_FLAT_SMALL = 0 / 1.
# usage (through gpurun): bash tools/validate_unvalidated.sh   -> gpurun_out/unvalidated_*.log
R=${GRAFT_REPO_ROOT:-/root/repo}
O=$R/gpurun_out
mkdir -p $O
cd $R
FAISS_AMD_RUN_UNVALIDATED=1 timeout 600 python -m pytest tests/test_gpu_unvalidated.py -q --tb=short --maxfail=10 -p no:cacheprovider > $O/unvalidated_tests.log 2>&1
tail -25 $O/unvalidated_tests.log
DBG_LIST=0,0/1,0/2,0/3,0/5,0/9,0/11,0,0/1,0/3 timeout 300 python tools/flat_only.py 20 > $O/unvalidated_flat_stagger.log 2>&1
grep -v amdgpu.ids $O/unvalidated_flat_stagger.log | tail -18
for s in 0 1; do
  FAISS_AMD_FLAT_SMALL=$s timeout 200 python tools/ivfpq_only.py 20 2>&1 | grep -v amdgpu.ids | sed "s/^/[FLAT_SMALL=$s] /" >> $O/unvalidated_ivfpq_small.log
done
grep -i "ms/step\|rerank\|flat_small\|QPS" $O/unvalidated_ivfpq_small.log | tail -12
for s in 0 1; do
  for t in ivfpq ivfflat ivfsq; do
    FAISS_AMD_IVF_SORT=$s timeout 200 python tools/${t}_only.py 10 2>&1 | grep -i "ms/step\|fused_kernel\|QPS" | sed "s/^/[IVF_SORT=$s $t] /" >> $O/unvalidated_ivf_sort.log
  done
done
cat $O/unvalidated_ivf_sort.log | tail -24

