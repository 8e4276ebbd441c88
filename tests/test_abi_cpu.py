"""CPU checks of the drop-in boundary: the library loads without a GPU, exports every symbol
declared in include/faiss_amd_c.h, follows the reference's error convention, and its host-side
logic (shard merge) agrees with the oracle."""
import ctypes
import os
import re

import numpy as np
import pytest

import faiss_amd
from compare import check_knn
from oracle.pyoracle import METRIC_INNER_PRODUCT, METRIC_L2, Oracle, integer_dataset

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _declared(path):
    hdr = re.sub(r"/\*.*?\*/", "", open(path).read(), flags=re.S)
    return set(re.findall(r"\b(faiss_amd_\w+)\s*\(", hdr))


def test_library_exports_every_declared_symbol():
    public = _declared(os.path.join(ROOT, "include", "faiss_amd_c.h"))
    # test / tuning hooks without a reference counterpart live in a private header, not in the drop-in boundary
    internal = _declared(os.path.join(ROOT, "faiss_amd", "csrc", "faiss_amd_internal.h"))
    assert len(public) >= 45
    assert internal == {"faiss_amd_test_select", "faiss_amd_GpuIndexFlat_filter_scores",
                        "faiss_amd_GpuIndexIVF_set_lmf_tuning", "faiss_amd_GpuIndexIVF_test_filter_dump",
                        "faiss_amd_GpuIndexIVF_set_lmf_sampling", "faiss_amd_GpuIndexIVFPQ_set_lmf_two_copies",
                        "faiss_amd_GpuIndexIVFPQ_set_lmf_fast_gather", "faiss_amd_sq_train_rangestat",
                        "faiss_amd_Index_set_small_fused", "faiss_amd_GpuIndexIVF_set_lmf_pair"}
    assert not (public & internal)
    lib = ctypes.CDLL(faiss_amd.LIB_PATH)
    missing = [s for s in sorted(public | internal) if not hasattr(lib, s)]
    assert not missing, missing
    # the python mirror declares prototypes for all of them
    assert public | internal == set(faiss_amd.exported_symbols())


def test_tuning_setter_validates_its_ranges_without_a_device():
    """bad tuning values must fail in the setter (a null handle is reported first: -2 either way, no crash)"""
    lib = faiss_amd.load_library()
    assert lib.faiss_amd_GpuIndexIVF_set_lmf_tuning(None, 0, 3, 0, 0) == -2


def test_no_gpu_fails_loudly_not_silently():
    n = faiss_amd.get_num_gpus()
    if n > 0:
        pytest.skip("a GPU is present")
    with pytest.raises(faiss_amd.FaissAmdError) as e:
        faiss_amd.StandardGpuResources(0)
    assert "no CPU fallback" in str(e.value)
    # error convention: code -2 + message, as c_api/macros_impl.h:22-56
    lib = faiss_amd.load_library()
    h = ctypes.c_void_p()
    assert lib.faiss_amd_StandardGpuResources_new(ctypes.byref(h), 0) == -2
    assert b"no HIP device" in lib.faiss_amd_get_last_error()


def test_null_handles_are_errors_not_crashes():
    lib = faiss_amd.load_library()
    assert lib.faiss_amd_Index_reset(None) == -2
    assert lib.faiss_amd_Index_d(None) == -1
    assert lib.faiss_amd_Index_ntotal(None) == -1


@pytest.mark.parametrize("metric", [METRIC_L2, METRIC_INNER_PRODUCT])
def test_host_merge_equals_oracle_and_unsharded(metric):
    xb, xq = integer_dataset(12, 2000, 31, seed=3, hi=4)
    parts = [(0, 700), (700, 701), (701, 2000)]
    k = 40
    aD = np.stack([Oracle.flat_search(metric, xb[a:b], xq, k)[0] for a, b in parts])
    aI = np.stack([Oracle.flat_search(metric, xb[a:b], xq, k)[1] for a, b in parts])  # shard 1 has 1 row: padding
    base = [a for a, _ in parts]
    D, I = faiss_amd.merge_knn_results(metric, aD, aI, base)
    check_knn(D, I, *Oracle.merge_shards(metric, aD, aI, base), exact=True, name="merge vs oracle")
    check_knn(D, I, *Oracle.flat_search(metric, xb, xq, k), exact=True, name="merge vs unsharded")


def test_product_does_not_import_the_oracle():
    """Only tests/, smoke() and bench.py's cpu_baseline may touch oracle/."""
    for dirpath, _, files in os.walk(os.path.join(ROOT, "faiss_amd")):
        for f in files:
            if f.endswith((".py", ".cpp", ".hip", ".h")) or f == "Makefile":
                src = open(os.path.join(dirpath, f), errors="replace").read()
                code = "\n".join(l for l in src.splitlines()
                                 if not l.strip().startswith(("//", "#", "*", "/*")))
                assert "pyoracle" not in code and "libfaiss_oracle" not in code and "libfaiss_ref" not in code, f


def test_meta_indexes_without_members_raise():
    """IndexShards / IndexReplicas need no GPU themselves; their error behaviour mirrors the reference
    (faiss/IndexReplicas.cpp:132 "no replicas in index", faiss/IndexShards.cpp FAISS_THROW_IF_NOT(count() > 0))."""
    xq = np.zeros((3, 8), dtype=np.float32)
    rep = faiss_amd.IndexReplicas(8)
    assert rep.ntotal == 0 and rep.d == 8
    with pytest.raises(faiss_amd.FaissAmdError) as e:
        rep.search(xq, 2)
    assert "no replicas" in str(e.value)
    with pytest.raises(faiss_amd.FaissAmdError):
        rep.reconstruct(0)
    sh = faiss_amd.IndexShards(8)
    with pytest.raises(faiss_amd.FaissAmdError) as e:
        sh.search(xq, 2)
    assert "no shards" in str(e.value)
    with pytest.raises(faiss_amd.FaissAmdError):
        sh.add(xq)


def test_bfknn_without_resources_is_an_error():
    lib = faiss_amd.load_library()
    x = np.zeros((4, 8), dtype=np.float32)
    D = np.zeros((4, 2), dtype=np.float32)
    I = np.zeros((4, 2), dtype=np.int64)
    p = lambda a: a.ctypes.data_as(ctypes.c_void_p)  # noqa: E731
    assert lib.faiss_amd_bfKnn(None, 1, p(x), 4, p(x), 4, 8, 2, p(D), p(I)) == -2
    assert b"null resources" in lib.faiss_amd_get_last_error()


def test_metric_predicate_matches_metrictype_h():
    """faiss_amd_metric_supported (the predicate the constructors apply) over every value of faiss/MetricType.h:31-52."""
    flat_ok = {0, 1, 2, 3, 4, 20, 21, 22, 23}
    for m in list(range(-2, 30)) + [99, 1 << 20]:
        assert faiss_amd.metric_supported(0, m) == (m in flat_ok), m
        assert faiss_amd.metric_supported(1, m) == (m in (0, 1)), m


def test_gpu_tests_expect_refusal_only_of_metrics_the_library_refuses():
    """Round 2's driver run stopped on `pytest.raises(...): GpuIndexFlat(res, 8, 23)` after 23 (Jaccard) had become a
    valid metric.  Walk every `with pytest.raises` block of the GPU tests: a constructor call in it whose metric argument
    is a literal (or a faiss_amd.METRIC_* name) and whose other arguments are plainly valid must name a metric the
    library's own predicate refuses -- checked here without a device."""
    import ast
    import glob
    metric_pos = {"GpuIndexFlat": (2, 0), "GpuIndexIVFFlat": (3, 1), "GpuIndexIVFPQ": (5, 1),
                  "GpuIndexIVFScalarQuantizer": (4, 1)}

    def value(node):
        if isinstance(node, ast.Constant) and isinstance(node.value, int):
            return node.value
        if isinstance(node, ast.Attribute) and node.attr.startswith("METRIC_"):
            return getattr(faiss_amd, node.attr)
        return None

    checked = 0
    for path in sorted(glob.glob(os.path.join(ROOT, "tests", "test_gpu_*.py"))):
        tree = ast.parse(open(path).read())
        for w in ast.walk(tree):
            if not isinstance(w, ast.With):
                continue
            ctx = w.items[0].context_expr
            if not (isinstance(ctx, ast.Call) and getattr(ctx.func, "attr", "") == "raises"):
                continue
            match = next((kw.value.value for kw in ctx.keywords if kw.arg == "match" and isinstance(kw.value, ast.Constant)),
                         None)
            for c in ast.walk(w):
                if isinstance(c, ast.Call) and getattr(c.func, "attr", "") in metric_pos:
                    pos, kind = metric_pos[c.func.attr]
                    if len(c.args) != pos + 1:
                        continue
                    m = value(c.args[pos])
                    if m is None:
                        continue
                    # the block expects a refusal; when it is about the metric (the only non-trivial argument, or the
                    # message says so) the library must indeed refuse that value
                    about_metric = match is None and c.func.attr == "GpuIndexFlat" or (match and "metric" in match)
                    if about_metric:
                        assert not faiss_amd.metric_supported(kind, m), (os.path.basename(path), c.lineno, m)
                        checked += 1
    assert checked >= 2
