"""GPU tests of round 6: ADVICE r5's findings (stale decoded copies after a codebook change, transactional patching of the
sweeps' copy), and the boundary additions of VERDICT r5 (caller-owned / fp16 coarse quantizer, indices options, memory
report, scalar-quantizer range statistics).  Every comparison is bit-exact unless a tolerance is written next to it.
Reference behaviour: faiss/gpu/GpuIndexIVF.cu:41-110, faiss/gpu/GpuIndexIVFPQ.cu:170-217, faiss/gpu/impl/IVFBase.cu:509-593."""
import numpy as np
import pytest

import faiss_amd
from oracle.pyoracle import METRIC_INNER_PRODUCT, METRIC_L2, synthetic_dataset

pytestmark = pytest.mark.gpu


@pytest.mark.parametrize("d,M", [(96, 32), (192, 48)])
def test_new_pq_codebook_invalidates_the_decoded_copy_of_the_lists(res, d, M):
    """the decoded-residual mode of the IVFPQ filter sweeps keeps an fp16 copy of the residuals DECODED WITH
    the codebook.  copy_pq_centroids() on an index that holds rows must drop that copy: otherwise the sweeps estimate with
    the old codebook's values while the band and the rerank use the new one, and the superset guarantee is gone."""
    nlist, k = 16, 30
    xt, xb, xq = synthetic_dataset(d, 6000, 30000, 700, seed=d + 3)
    idx = faiss_amd.GpuIndexIVFPQ(res, d, nlist, M, 8, METRIC_L2)
    idx.train(xt)
    idx.add(xb)
    idx.nprobe = 6
    idx.set_scan_mode(idx.SCAN_LIST_MAJOR)
    D0, I0 = idx.search(xq, k)  # builds the decoded copy
    assert idx.scan_info()[1] == 2 and idx.last_scan_arith() == 0
    pq = idx.get_pq_centroids()
    rs = np.random.RandomState(7)
    # a codebook that differs a lot: entries permuted inside every sub-quantizer and scaled
    pq2 = np.stack([pq[m][rs.permutation(256)] * (0.6 + 0.05 * (m % 5)) for m in range(M)]).astype(np.float32)
    idx.copy_pq_centroids(pq2)
    D1, I1 = idx.search(xq, k)
    assert idx.scan_info()[1] == 2
    idx.set_scan_mode(idx.SCAN_QUERY_MAJOR)
    Dq, Iq = idx.search(xq, k)
    assert np.array_equal(I1, Iq) and np.array_equal(D1, Dq)
    assert not np.array_equal(I0, I1)  # (the codebook really changed the answer)
