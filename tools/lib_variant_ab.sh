#!/bin/bash
# tools/lib_variant_ab.sh VARIANT -- the IVF search loops with lib/variants/libfaiss_amd_VARIANT.so against the shipped library,
# alternating processes on one box (per-leg search time + sweep spans)
V=${1:-noslp}
export FAISS_AMD_EXPERIMENTS=1
for rep in 1 2; do
  for lib in "" $V; do
    echo "=== library: ${lib:-shipped} (run $rep)"
    FAISS_AMD_LIB_VARIANT=$lib python tools/pq_fg_ab.py 1 10 100 2>&1 | grep "nb =\|run 2 fast_gather 1"
    for nb in 1000000 10000000; do
      FAISS_AMD_LIB_VARIANT=$lib python tools/ivfflat_only.py 10 $nb 2>&1 | grep "ms/step\|sweep"
    done
    FAISS_AMD_LIB_VARIANT=$lib python tools/ivfsq_only.py 10 1000000 2>&1 | grep "ms/step\|sweep"
  done
done
