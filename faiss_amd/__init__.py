"""faiss_amd -- thin ctypes mirror of the reference's Python surface for the hot path.

The product is the C-ABI shared library ``faiss_amd/lib/libfaiss_amd.so`` (hand-written
gfx950 kernels + C++ host code, see ``include/faiss_amd_c.h``).  This module only marshals
numpy arrays / raw pointers into that ABI, using the reference's class and method names
(faiss/python/class_wrappers.py: ``index.add(x)``, ``D, I = index.search(x, k)``,
``index.train(x)``, ``index.nprobe``), so the reference's tests read the same here.

There is NO CPU fallback: importing works anywhere, but constructing
``StandardGpuResources`` raises if the library is missing or no HIP device is visible.
"""
import ctypes
import os

import numpy as np

METRIC_INNER_PRODUCT = 0
METRIC_L2 = 1
# the extra metrics of the flat index / knn_gpu (faiss.METRIC_*, faiss/MetricType.h:31-52)
METRIC_L1 = 2
METRIC_Linf = 3
METRIC_Lp = 4
METRIC_Canberra = 20
METRIC_BrayCurtis = 21
METRIC_JensenShannon = 22
METRIC_Jaccard = 23

_HERE = os.path.dirname(os.path.abspath(__file__))
LIB_PATH = os.path.join(_HERE, "lib", "libfaiss_amd.so")

_c_idx_p = ctypes.c_void_p
_lib = None


class FaissAmdError(RuntimeError):
    """Raised for non-zero C-ABI return codes (reference: faiss.FaissException via SWIG)."""


def load_library():
    """Load libfaiss_amd.so (built in-tree by ``__graft_entry__.build()``) and declare prototypes."""
    global _lib
    if _lib is not None:
        return _lib
    variant = os.environ.get("FAISS_AMD_LIB_VARIANT") if os.environ.get("FAISS_AMD_EXPERIMENTS") == "1" else None
    if variant:  # tools/ only: an A/B build of the library under lib/variants/ (`make -C faiss_amd/csrc variants`)
        globals()["LIB_PATH"] = os.path.join(_HERE, "lib", "variants", "libfaiss_amd_%s.so" % variant)
    if not os.path.exists(LIB_PATH):
        raise FaissAmdError(
            "%s not found: build it with `python -c 'import __graft_entry__ as g; g.build()'` "
            "(hipcc --offload-arch=gfx950). There is no CPU fallback." % LIB_PATH)
    lib = ctypes.CDLL(LIB_PATH)
    i64, i32, vp, sz = ctypes.c_int64, ctypes.c_int, ctypes.c_void_p, ctypes.c_size_t
    P = ctypes.POINTER
    protos = {
        "faiss_amd_get_last_error": (ctypes.c_char_p, []),
        "faiss_amd_get_num_gpus": (i32, [P(i32)]),
        "faiss_amd_metric_supported": (i32, [i32, i32, P(i32)]),
        "faiss_amd_StandardGpuResources_new": (i32, [P(vp), i32]),
        "faiss_amd_StandardGpuResources_free": (None, [vp]),
        "faiss_amd_StandardGpuResources_sync": (i32, [vp]),
        "faiss_amd_StandardGpuResources_getDefaultStream": (i32, [vp, P(vp)]),
        "faiss_amd_StandardGpuResources_setTempMemory": (i32, [vp, sz]),
        "faiss_amd_StandardGpuResources_setPagedSearch": (i32, [vp, sz, i64]),
        "faiss_amd_StandardGpuResources_getPagedSearchCount": (i32, [vp, P(i64)]),
        "faiss_amd_StandardGpuResources_setDefaultStream": (i32, [vp, vp]),
        "faiss_amd_StandardGpuResources_getMemoryInfo": (i32, [vp, P(sz), P(sz), P(sz), P(sz), P(sz), P(sz)]),
        "faiss_amd_StandardGpuResources_setLogMemoryAllocations": (i32, [vp, i32]),
        "faiss_amd_GpuIndexIVFFlat_new_with_quantizer": (i32, [P(vp), vp, vp, i32, i32, i32, vp]),
        "faiss_amd_GpuIndexIVFPQ_new_with_quantizer": (i32, [P(vp), vp, vp, i32, i32, i32, i32, i32, vp]),
        "faiss_amd_GpuIndexIVFScalarQuantizer_new_with_quantizer": (i32, [P(vp), vp, vp, i32, i32, i32, i32, i32, vp]),
        "faiss_amd_GpuIndexIVF_quantizer_info": (i32, [vp, P(i32), P(i32), P(i32)]),
        "faiss_amd_sq_train_rangestat": (i32, [i32, i32, ctypes.c_float, i64, i32, vp, vp]),
        "faiss_amd_GpuIndexFlat_new": (i32, [P(vp), vp, i32, i32]),
        "faiss_amd_GpuIndexIVFFlat_new": (i32, [P(vp), vp, i32, i32, i32]),
        "faiss_amd_GpuIndexIVFPQ_new": (i32, [P(vp), vp, i32, i32, i32, i32, i32]),
        "faiss_amd_GpuIndexIVFScalarQuantizer_new": (i32, [P(vp), vp, i32, i32, i32, i32, i32]),
        "faiss_amd_IndexIVFSQ_info": (i32, [vp, P(i32), P(i32), P(sz), P(sz)]),
        "faiss_amd_IndexIVFSQ_get_trained": (i32, [vp, vp]),
        "faiss_amd_IndexIVFSQ_copy_trained": (i32, [vp, vp, sz]),
        "faiss_amd_IndexIVFSQ_set_rangestat": (i32, [vp, i32, ctypes.c_float]),
        "faiss_amd_IndexShards_new": (i32, [P(vp), i32, i32, i32]),
        "faiss_amd_IndexShards_add_shard": (i32, [vp, vp]),
        "faiss_amd_IndexReplicas_new": (i32, [P(vp), i32, i32]),
        "faiss_amd_IndexReplicas_add_replica": (i32, [vp, vp]),
        "faiss_amd_Index_free": (None, [vp]),
        "faiss_amd_Index_d": (i32, [vp]),
        "faiss_amd_Index_is_trained": (i32, [vp]),
        "faiss_amd_Index_ntotal": (i64, [vp]),
        "faiss_amd_Index_metric_type": (i32, [vp]),
        "faiss_amd_Index_metric_arg": (ctypes.c_float, [vp]),
        "faiss_amd_Index_set_metric_arg": (i32, [vp, ctypes.c_float]),
        "faiss_amd_Index_train": (i32, [vp, i64, vp]),
        "faiss_amd_Index_add": (i32, [vp, i64, vp]),
        "faiss_amd_Index_add_with_ids": (i32, [vp, i64, vp, vp]),
        "faiss_amd_Index_search": (i32, [vp, i64, vp, i64, vp, vp]),
        "faiss_amd_Index_assign": (i32, [vp, i64, vp, vp, i64]),
        "faiss_amd_Index_reset": (i32, [vp]),
        "faiss_amd_Index_reconstruct": (i32, [vp, i64, vp]),
        "faiss_amd_Index_reconstruct_n": (i32, [vp, i64, i64, vp]),
        "faiss_amd_IndexIVF_nlist": (i32, [vp, P(i32)]),
        "faiss_amd_IndexIVF_nprobe": (i32, [vp, P(i32)]),
        "faiss_amd_IndexIVF_set_nprobe": (i32, [vp, i32]),
        "faiss_amd_IndexIVF_get_list_size": (i32, [vp, i64, P(sz)]),
        "faiss_amd_IndexIVF_get_list_ids": (i32, [vp, i64, vp]),
        "faiss_amd_IndexIVF_get_list_codes": (i32, [vp, i64, vp]),
        "faiss_amd_IndexIVF_code_size": (i32, [vp, P(sz)]),
        "faiss_amd_IndexIVF_get_centroids": (i32, [vp, vp]),
        "faiss_amd_IndexIVF_set_clustering": (i32, [vp, i32, i32]),
        "faiss_amd_IndexIVF_copy_centroids": (i32, [vp, vp]),
        "faiss_amd_IndexIVFPQ_copy_pq_centroids": (i32, [vp, vp]),
        "faiss_amd_IndexIVFPQ_get_pq_centroids": (i32, [vp, vp]),
        "faiss_amd_IndexIVF_copy_lists": (i32, [vp, vp, vp, vp]),
        "faiss_amd_kmeans_clustering": (i32, [vp, i32, i64, i32, vp, i32, i32, vp, vp]),
        "faiss_amd_Clustering_train": (i32, [vp, i64, vp, i32, i32, i32, vp, vp, vp]),
        "faiss_amd_ClusteringParameters_init": (None, [vp]),
        "faiss_amd_Clustering_train_ex": (i32, [vp, i64, vp, i32, vp, vp, i32, vp, vp, vp]),
        "faiss_amd_IndexIVF_set_clustering_params": (i32, [vp, vp]),
        "faiss_amd_merge_knn_results": (i32, [i32, i64, i64, i32, vp, vp, vp, vp, vp]),
        "faiss_amd_merge_knn_results_device": (i32, [vp, i32, i64, i64, i32, vp, vp, vp, vp, vp]),
        "faiss_amd_profile_enable": (i32, [vp, i32]),
        "faiss_amd_profile_reset": (i32, [vp]),
        "faiss_amd_profile_get": (i32, [vp, ctypes.c_char_p, P(ctypes.c_double), P(ctypes.c_long)]),
        "faiss_amd_GpuIndexFlat_pairwise_distances": (i32, [vp, i64, vp, vp]),
        "faiss_amd_GpuIndexFlat_set_use_simple_kernel": (i32, [vp, i32]),
        "faiss_amd_GpuIndexIVF_set_use_fused_scan": (i32, [vp, i32]),
        "faiss_amd_GpuIndexIVF_set_scan_mode": (i32, [vp, i32]),
        "faiss_amd_GpuIndexIVF_scan_info": (i32, [vp, P(i32), P(i32), P(i64)]),
        "faiss_amd_GpuIndexIVF_set_use_filter_shadow": (i32, [vp, i32]),
        "faiss_amd_GpuIndexIVF_resident_bytes": (i32, [vp, P(ctypes.c_size_t), P(ctypes.c_size_t)]),
        "faiss_amd_GpuIndexIVF_last_scan_arith": (i32, [vp, P(i32)]),
        "faiss_amd_GpuIndexIVF_test_filter_dump": (i32, [vp, i64, vp, i32, i64, i64, vp, vp]),
        "faiss_amd_GpuIndexIVF_set_lmf_tuning": (i32, [vp, i32, i32, i32, i32]),
        "faiss_amd_GpuIndexIVF_set_lmf_sampling": (i32, [vp, i32]),
        "faiss_amd_GpuIndexIVFPQ_set_lmf_two_copies": (i32, [vp, i32]),
        "faiss_amd_GpuIndexIVFPQ_set_lmf_fast_gather": (i32, [vp, i32]),
        "faiss_amd_Index_set_small_fused": (i32, [vp, i32]),
        "faiss_amd_GpuIndexIVF_set_lmf_pair": (i32, [vp, i32]),
        "faiss_amd_GpuIndexIVF_list_major_rule": (i32, [vp, i64, i32, i64, P(i32)]),
        "faiss_amd_GpuIndexFlat_set_use_filter_kernel": (i32, [vp, i32, i64]),
        "faiss_amd_GpuIndexFlat_filter_stats": (i32, [vp, P(i32), P(i32)]),
        "faiss_amd_GpuIndexFlat_filter_scores": (i32, [vp, i64, vp, vp, vp]),
        "faiss_amd_Index_reconstruct_batch": (i32, [vp, i64, vp, vp]),
        "faiss_amd_Index_compute_residual": (i32, [vp, vp, vp, i64]),
        "faiss_amd_Index_compute_residual_n": (i32, [vp, i64, vp, vp, vp]),
        "faiss_amd_GpuIndexIVF_search_preassigned": (i32, [vp, i64, vp, i64, vp, vp, vp, vp]),
        "faiss_amd_GpuIndexIVF_add_core": (i32, [vp, i64, vp, vp, vp]),
        "faiss_amd_GpuIndexIVF_reserveMemory": (i32, [vp, sz]),
        "faiss_amd_GpuIndexIVF_reclaimMemory": (i32, [vp, P(sz)]),
        "faiss_amd_GpuIndexIVF_updateQuantizer": (i32, [vp]),
        "faiss_amd_GpuIndexIVFPQ_setPrecomputedCodes": (i32, [vp, i32]),
        "faiss_amd_GpuIndexIVFPQ_getTableInfo": (i32, [vp, P(i32), P(i32), P(i32)]),
        "faiss_amd_GpuIndexIVFPQ_getInfo": (i32, [vp, P(i32), P(i32), P(i32), P(i32)]),
        "faiss_amd_IndexIVF_quantizer_search": (i32, [vp, i64, vp, i64, vp, vp]),
        "faiss_amd_bfKnn": (i32, [vp, i32, vp, i64, vp, i64, i32, i64, vp, vp]),
        "faiss_amd_GpuIndexFlat_new_with_config": (i32, [P(vp), vp, i32, i32, vp]),
        "faiss_amd_GpuIndexIVFFlat_new_with_config": (i32, [P(vp), vp, i32, i32, i32, vp]),
        "faiss_amd_GpuIndexIVFPQ_new_with_config": (i32, [P(vp), vp, i32, i32, i32, i32, i32, vp]),
        "faiss_amd_GpuIndexFlat_resident_bytes": (i32, [vp, P(sz)]),
        "faiss_amd_set_interrupt_callback": (i32, [vp, vp]),
        "faiss_amd_GpuParameterSpace_set_index_parameter": (i32, [vp, ctypes.c_char_p, ctypes.c_double]),
        "faiss_amd_bfKnn_params": (i32, [vp, vp]),
        "faiss_amd_bfKnn_tiling": (i32, [vp, vp, sz, sz]),
        "faiss_amd_test_select": (i32, [vp, i32, i32, i32, i32, i32, vp, vp, vp]),
        "faiss_amd_GpuIndexIVF_search_with_params": (i32, [vp, i64, vp, i64, vp, vp, vp]),
        "faiss_amd_IDSelectorAll_new": (i32, [P(vp)]),
        "faiss_amd_IDSelectorRange_new": (i32, [P(vp), i64, i64]),
        "faiss_amd_IDSelectorBatch_new": (i32, [P(vp), sz, vp]),
        "faiss_amd_IDSelectorArray_new": (i32, [P(vp), sz, vp]),
        "faiss_amd_IDSelectorBitmap_new": (i32, [P(vp), sz, vp]),
        "faiss_amd_IDSelectorNot_new": (i32, [P(vp), vp]),
        "faiss_amd_IDSelectorAnd_new": (i32, [P(vp), vp, vp]),
        "faiss_amd_IDSelectorOr_new": (i32, [P(vp), vp, vp]),
        "faiss_amd_IDSelectorXOr_new": (i32, [P(vp), vp, vp]),
        "faiss_amd_IDSelector_is_member": (i32, [vp, i64]),
        "faiss_amd_IDSelector_free": (None, [vp]),
        "faiss_amd_SearchParameters_new": (i32, [P(vp), vp]),
        "faiss_amd_SearchParametersIVF_new_with": (i32, [P(vp), vp, sz, sz]),
        "faiss_amd_SearchParameters_free": (None, [vp]),
        "faiss_amd_Index_search_with_params": (i32, [vp, i64, vp, i64, vp, vp, vp]),
        "faiss_amd_GpuIndexIVF_search_preassigned_with_params": (i32, [vp, i64, vp, i64, vp, vp, vp, vp, vp]),
        "faiss_amd_GpuIndexIVF_stored_vectors": (i32, [vp, P(i64)]),
        "faiss_amd_GpuIndexIVF_arena_stats": (i32, [vp, P(i64), P(i64), P(i64)]),
    }
    for name, (res, args) in protos.items():
        fn = getattr(lib, name)  # AttributeError => the library does not export the symbol
        fn.restype = res
        fn.argtypes = args
    lib._protos = protos
    _lib = lib
    return lib


def exported_symbols():
    """Names declared in include/faiss_amd_c.h that the loaded library must export."""
    return sorted(load_library()._protos.keys())


def _check(rc):
    if rc != 0:
        msg = load_library().faiss_amd_get_last_error()
        raise FaissAmdError("faiss_amd error %d: %s" % (rc, msg.decode("utf-8", "replace") if msg else "?"))


def metric_supported(index_kind, metric):
    """does an index of the kind (0 = GpuIndexFlat / bfKnn, 1 = IVF) accept the metric?  (no device needed)"""
    out = ctypes.c_int(0)
    _check(load_library().faiss_amd_metric_supported(int(index_kind), int(metric), ctypes.byref(out)))
    return bool(out.value)


def get_num_gpus():
    n = ctypes.c_int(0)
    _check(load_library().faiss_amd_get_num_gpus(ctypes.byref(n)))
    return n.value


def _f32(x, d=None):
    x = np.ascontiguousarray(x, dtype=np.float32)
    if x.ndim != 2:
        raise ValueError("expected a 2-D float32 array")
    if d is not None and x.shape[1] != d:
        raise ValueError("expected %d columns, got %d" % (d, x.shape[1]))
    return x


def _ptr(a):
    return ctypes.c_void_p(a.ctypes.data) if a is not None else None


class StandardGpuResources:
    """faiss.StandardGpuResources (faiss/gpu/StandardGpuResources.h): one stream + scratch per device."""

    def __init__(self, device=0):
        self._lib = load_library()
        h = ctypes.c_void_p()
        _check(self._lib.faiss_amd_StandardGpuResources_new(ctypes.byref(h), int(device)))
        self._h = h
        self.device = int(device)

    def __del__(self):
        h, self._h = getattr(self, "_h", None), None
        if h:
            self._lib.faiss_amd_StandardGpuResources_free(h)

    def syncDefaultStreamCurrentDevice(self):
        _check(self._lib.faiss_amd_StandardGpuResources_sync(self._h))

    def getDefaultStream(self):
        s = ctypes.c_void_p()
        _check(self._lib.faiss_amd_StandardGpuResources_getDefaultStream(self._h, ctypes.byref(s)))
        return s.value

    def setTempMemory(self, nbytes):
        _check(self._lib.faiss_amd_StandardGpuResources_setTempMemory(self._h, int(nbytes)))

    def getMemoryInfo(self):
        """StandardGpuResources.getMemoryInfo: the library's live device allocations on this device"""
        v = [ctypes.c_size_t(0) for _ in range(6)]
        _check(self._lib.faiss_amd_StandardGpuResources_getMemoryInfo(self._h, *[ctypes.byref(x) for x in v]))
        keys = ("allocations", "bytes", "peak_bytes", "temp_memory", "device_free", "device_total")
        return {k: x.value for k, x in zip(keys, v)}

    def setLogMemoryAllocations(self, enable):
        _check(self._lib.faiss_amd_StandardGpuResources_setLogMemoryAllocations(self._h, int(bool(enable))))

    def setDefaultStream(self, stream):
        """order all work of this resources object on `stream` (a hipStream_t value, e.g.
        torch.cuda.current_stream().cuda_stream); None / 0 = back to the private stream"""
        _check(self._lib.faiss_amd_StandardGpuResources_setDefaultStream(self._h, ctypes.c_void_p(stream or 0)))

    def setPagedSearch(self, min_bytes=64 << 20, page_queries=0):
        """host-resident query batches of at least min_bytes take the pinned double-buffered path"""
        _check(self._lib.faiss_amd_StandardGpuResources_setPagedSearch(self._h, int(min_bytes), int(page_queries)))

    @property
    def paged_search_count(self):
        v = ctypes.c_int64(0)
        _check(self._lib.faiss_amd_StandardGpuResources_getPagedSearchCount(self._h, ctypes.byref(v)))
        return v.value

    # measurement hooks ---------------------------------------------------------------
    def profile_enable(self, on=True):
        _check(self._lib.faiss_amd_profile_enable(self._h, 1 if on else 0))

    def profile_reset(self):
        _check(self._lib.faiss_amd_profile_reset(self._h))

    def profile_get(self, kernel_name):
        ms, n = ctypes.c_double(0), ctypes.c_long(0)
        _check(self._lib.faiss_amd_profile_get(self._h, kernel_name.encode(), ctypes.byref(ms), ctypes.byref(n)))
        return ms.value, n.value


class Index:
    """faiss.Index surface (faiss/Index.h:101-431) over an opaque C-ABI handle."""

    def __init__(self):
        self._lib = load_library()
        self._h = ctypes.c_void_p()
        self._keep = []

    def __del__(self):
        h, self._h = getattr(self, "_h", None), None
        if h:
            self._lib.faiss_amd_Index_free(h)

    d = property(lambda self: self._lib.faiss_amd_Index_d(self._h))
    ntotal = property(lambda self: self._lib.faiss_amd_Index_ntotal(self._h))
    is_trained = property(lambda self: bool(self._lib.faiss_amd_Index_is_trained(self._h)))
    metric_type = property(lambda self: self._lib.faiss_amd_Index_metric_type(self._h))

    @property
    def metric_arg(self):
        """faiss.Index.metric_arg: the p of METRIC_Lp"""
        return float(self._lib.faiss_amd_Index_metric_arg(self._h))

    @metric_arg.setter
    def metric_arg(self, v):
        _check(self._lib.faiss_amd_Index_set_metric_arg(self._h, float(v)))

    def train(self, x):
        x = _f32(x, self.d)
        _check(self._lib.faiss_amd_Index_train(self._h, x.shape[0], _ptr(x)))

    def add(self, x):
        x = _f32(x, self.d)
        _check(self._lib.faiss_amd_Index_add(self._h, x.shape[0], _ptr(x)))

    def add_with_ids(self, x, ids):
        x = _f32(x, self.d)
        ids = np.ascontiguousarray(ids, dtype=np.int64)
        if ids.shape != (x.shape[0],):
            raise ValueError("ids must have one entry per vector")
        _check(self._lib.faiss_amd_Index_add_with_ids(self._h, x.shape[0], _ptr(x), _ptr(ids)))

    def search(self, x, k, params=None):
        """D, I = index.search(x, k, params=None) (faiss/python/class_wrappers.py replacement_search); params:
        SearchParameters(sel=...) / SearchParametersIVF(sel=..., nprobe=...)"""
        x = _f32(x, self.d)
        n = x.shape[0]
        D = np.empty((n, k), dtype=np.float32)
        I = np.empty((n, k), dtype=np.int64)
        if params is None:
            _check(self._lib.faiss_amd_Index_search(self._h, n, _ptr(x), int(k), _ptr(D), _ptr(I)))
        else:
            with _params_handle(self._lib, params) as h:
                _check(self._lib.faiss_amd_Index_search_with_params(self._h, n, _ptr(x), int(k), h, _ptr(D), _ptr(I)))
        return D, I

    def set_small_fused(self, on):
        """A/B knob: flat searches over <= 4096 rows (this index, or the coarse quantizer of an IVF index) in one launch"""
        _check(self._lib.faiss_amd_Index_set_small_fused(self._h, int(bool(on))))

    def search_ptr(self, n, x_ptr, k, d_ptr, i_ptr, params=None):
        """Raw-pointer search (host or device addresses), e.g. torch tensors' ``data_ptr()``."""
        if params is None:
            _check(self._lib.faiss_amd_Index_search(self._h, int(n), ctypes.c_void_p(x_ptr), int(k),
                                                    ctypes.c_void_p(d_ptr), ctypes.c_void_p(i_ptr)))
        else:
            with _params_handle(self._lib, params) as h:
                _check(self._lib.faiss_amd_Index_search_with_params(self._h, int(n), ctypes.c_void_p(x_ptr), int(k), h,
                                                                    ctypes.c_void_p(d_ptr), ctypes.c_void_p(i_ptr)))

    def add_ptr(self, n, x_ptr):
        _check(self._lib.faiss_amd_Index_add(self._h, int(n), ctypes.c_void_p(x_ptr)))

    def add_with_ids_ptr(self, n, x_ptr, ids):
        """add_with_ids with the vectors behind a raw (host or device) address; `ids`: numpy int64 [n] or an address"""
        if isinstance(ids, np.ndarray):
            ids = np.ascontiguousarray(ids, dtype=np.int64)
            if ids.shape != (int(n),):
                raise ValueError("ids must have one entry per vector")
            ip = _ptr(ids)
        else:
            ip = ctypes.c_void_p(ids)
        _check(self._lib.faiss_amd_Index_add_with_ids(self._h, int(n), ctypes.c_void_p(x_ptr), ip))

    def train_ptr(self, x_ptr, n):
        _check(self._lib.faiss_amd_Index_train(self._h, int(n), ctypes.c_void_p(x_ptr)))

    def assign(self, x, k=1):
        x = _f32(x, self.d)
        I = np.empty((x.shape[0], k), dtype=np.int64)
        _check(self._lib.faiss_amd_Index_assign(self._h, x.shape[0], _ptr(x), _ptr(I), int(k)))
        return I

    def reset(self):
        _check(self._lib.faiss_amd_Index_reset(self._h))

    def reconstruct(self, key):
        out = np.empty(self.d, dtype=np.float32)
        _check(self._lib.faiss_amd_Index_reconstruct(self._h, int(key), _ptr(out)))
        return out

    def reconstruct_batch(self, keys):
        keys = np.ascontiguousarray(keys, dtype=np.int64)
        out = np.empty((keys.shape[0], self.d), dtype=np.float32)
        _check(self._lib.faiss_amd_Index_reconstruct_batch(self._h, keys.shape[0], _ptr(keys), _ptr(out)))
        return out

    def compute_residual(self, x, key):
        x = _f32(np.asarray(x).reshape(1, -1), self.d)
        out = np.empty(self.d, dtype=np.float32)
        _check(self._lib.faiss_amd_Index_compute_residual(self._h, _ptr(x), _ptr(out), int(key)))
        return out

    def compute_residual_n(self, x, keys):
        x = _f32(x, self.d)
        keys = np.ascontiguousarray(keys, dtype=np.int64)
        out = np.empty((x.shape[0], self.d), dtype=np.float32)
        _check(self._lib.faiss_amd_Index_compute_residual_n(self._h, x.shape[0], _ptr(x), _ptr(out), _ptr(keys)))
        return out

    def reconstruct_n(self, i0, ni):
        out = np.empty((ni, self.d), dtype=np.float32)
        _check(self._lib.faiss_amd_Index_reconstruct_n(self._h, int(i0), int(ni), _ptr(out)))
        return out


class GpuIndexFlatConfig(ctypes.Structure):
    """faiss.GpuIndexFlatConfig (faiss/gpu/GpuIndexFlat.h:24-40 + GpuIndexConfig): device, memorySpace, useFloat16,
    storeTransposed -- see include/faiss_amd_c.h for what each means on this backend"""
    _fields_ = [("device", ctypes.c_int), ("memorySpace", ctypes.c_int), ("useFloat16", ctypes.c_int),
                ("storeTransposed", ctypes.c_int)]

    def __init__(self, device=-1, memorySpace=0, useFloat16=False, storeTransposed=False):
        super().__init__(int(device), int(memorySpace), int(useFloat16), int(storeTransposed))


class GpuIndexIVFConfig(ctypes.Structure):
    """faiss.GpuIndexIVFConfig / GpuIndexIVFFlatConfig (faiss/gpu/GpuIndexIVF.h:24-38)"""
    _fields_ = [("device", ctypes.c_int), ("memorySpace", ctypes.c_int), ("indicesOptions", ctypes.c_int),
                ("flat_useFloat16", ctypes.c_int), ("allowCpuCoarseQuantizer", ctypes.c_int)]

    def __init__(self, device=-1, memorySpace=0, indicesOptions=3, flat_useFloat16=False, allowCpuCoarseQuantizer=False):
        super().__init__(int(device), int(memorySpace), int(indicesOptions), int(flat_useFloat16),
                         int(allowCpuCoarseQuantizer))


class GpuIndexIVFPQConfig(ctypes.Structure):
    """faiss.GpuIndexIVFPQConfig (faiss/gpu/GpuIndexIVFPQ.h:25-49)"""
    _fields_ = [("ivf", GpuIndexIVFConfig), ("useFloat16LookupTables", ctypes.c_int), ("usePrecomputedTables", ctypes.c_int),
                ("interleavedLayout", ctypes.c_int), ("useMMCodeDistance", ctypes.c_int)]

    def __init__(self, useFloat16LookupTables=False, usePrecomputedTables=False, interleavedLayout=False,
                 useMMCodeDistance=False, **ivf):
        super().__init__(GpuIndexIVFConfig(**ivf), int(useFloat16LookupTables), int(usePrecomputedTables),
                         int(interleavedLayout), int(useMMCodeDistance))


class GpuIndexFlat(Index):
    """faiss.GpuIndexFlat (faiss/gpu/GpuIndexFlat.h:41-153)."""

    def __init__(self, res, d, metric=METRIC_L2, config=None):
        super().__init__()
        self._keep.append(res)
        if config is None:
            _check(self._lib.faiss_amd_GpuIndexFlat_new(ctypes.byref(self._h), res._h, int(d), int(metric)))
        else:
            _check(self._lib.faiss_amd_GpuIndexFlat_new_with_config(ctypes.byref(self._h), res._h, int(d), int(metric),
                                                                   ctypes.byref(config)))

    @property
    def resident_bytes(self):
        v = ctypes.c_size_t(0)
        _check(self._lib.faiss_amd_GpuIndexFlat_resident_bytes(self._h, ctypes.byref(v)))
        return v.value

    def search_and_reconstruct(self, x, k):
        """D, I, R = index.search_and_reconstruct(x, k) (faiss::Index::search_and_reconstruct, faiss/Index.h:322-335; reference
        test TestGpuIndexFlat.cpp SearchAndReconstruct): R[i, j] = the stored vector I[i, j], NaN rows for missing results"""
        D, I = self.search(x, k)
        R = self.reconstruct_batch(I.reshape(-1)).reshape(I.shape[0], int(k), self.d)
        return D, I, R

    def pairwise_distances(self, x):
        x = _f32(x, self.d)
        out = np.empty((x.shape[0], self.ntotal), dtype=np.float32)
        _check(self._lib.faiss_amd_GpuIndexFlat_pairwise_distances(self._h, x.shape[0], _ptr(x), _ptr(out)))
        return out

    def set_use_simple_kernel(self, on):
        _check(self._lib.faiss_amd_GpuIndexFlat_set_use_simple_kernel(self._h, 1 if on else 0))

    def set_use_filter_kernel(self, on, min_rows=-1):
        """fp16 MFMA filter + exact fp32 re-rank (bit-identical results); min_rows: smallest database
        that takes this path"""
        _check(self._lib.faiss_amd_GpuIndexFlat_set_use_filter_kernel(self._h, 1 if on else 0, int(min_rows)))

    def filter_stats(self):
        """(used_filter, overflow_queries) of the last search tile"""
        a, b = ctypes.c_int(0), ctypes.c_int(0)
        _check(self._lib.faiss_amd_GpuIndexFlat_filter_stats(self._h, ctypes.byref(a), ctypes.byref(b)))
        return bool(a.value), b.value

    def filter_scores(self, x):
        """test hook: (approximate scores [n, ntotal], error bound [n]) of the filter kernel"""
        x = _f32(x, self.d)
        sc = np.empty((x.shape[0], self.ntotal), dtype=np.float32)
        eb = np.empty(x.shape[0], dtype=np.float32)
        _check(self._lib.faiss_amd_GpuIndexFlat_filter_scores(self._h, x.shape[0], _ptr(x), _ptr(sc), _ptr(eb)))
        return sc, eb


class GpuIndexFlatL2(GpuIndexFlat):
    def __init__(self, res, d, config=None):
        super().__init__(res, d, METRIC_L2, config)


class GpuIndexFlatIP(GpuIndexFlat):
    def __init__(self, res, d, config=None):
        super().__init__(res, d, METRIC_INNER_PRODUCT, config)


INDICES_CPU, INDICES_IVF, INDICES_32_BIT, INDICES_64_BIT = range(4)  # faiss/gpu/GpuIndicesOptions.h


class _GpuIndexIVF(Index):
    """faiss.GpuIndexIVF (faiss/gpu/GpuIndexIVF.h:37-153)."""

    def _adopt(self, quantizer):
        # GpuIndexIVF*(resources, coarseQuantizer, ...): the caller's flat index, not owned -- kept alive with this object
        if quantizer is not None:
            if not isinstance(quantizer, GpuIndexFlat):
                raise TypeError("the coarse quantizer must be a GpuIndexFlat of this library (any other index: run its search on "
                                "the host and call search_preassigned / add_core)")
            self._keep.append(quantizer)
            self.quantizer = quantizer
        return quantizer._h if quantizer is not None else None

    def set_lmf_pair(self, on):
        _check(self._lib.faiss_amd_GpuIndexIVF_set_lmf_pair(self._h, int(on)))  # 0 off, 1 sweep 1 only (default), 2 both sweeps

    def quantizer_info(self):
        """(own_fields, the quantizer stores fp16, IndicesOptions)"""
        a, b, c = ctypes.c_int(0), ctypes.c_int(0), ctypes.c_int(0)
        _check(self._lib.faiss_amd_GpuIndexIVF_quantizer_info(self._h, ctypes.byref(a), ctypes.byref(b), ctypes.byref(c)))
        return bool(a.value), bool(b.value), c.value

    def quantizer_search(self, x, k):
        """index.quantizer.search(x, k): the coarse centroids nearest to x (distances, list ids)"""
        x = _f32(x, self.d)
        D = np.empty((x.shape[0], k), dtype=np.float32)
        I = np.empty((x.shape[0], k), dtype=np.int64)
        _check(self._lib.faiss_amd_IndexIVF_quantizer_search(self._h, x.shape[0], _ptr(x), int(k), _ptr(D), _ptr(I)))
        return D, I

    def reserveMemory(self, num_vecs):
        """GpuIndexIVF*.reserveMemory: room for num_vecs vectors up front"""
        _check(self._lib.faiss_amd_GpuIndexIVF_reserveMemory(self._h, int(num_vecs)))

    def reclaimMemory(self):
        """GpuIndexIVF*.reclaimMemory: release slack, holes and add scratch; returns the device bytes given back"""
        b = ctypes.c_size_t(0)
        _check(self._lib.faiss_amd_GpuIndexIVF_reclaimMemory(self._h, ctypes.byref(b)))
        return b.value

    def updateQuantizer(self):
        _check(self._lib.faiss_amd_GpuIndexIVF_updateQuantizer(self._h))

    def add_core(self, x, assign, ids=None):
        """index.add_core(n, x, ids, assign) of the reference (GpuIndexIVF::add_core; contrib/ivf_tools.py
        add_preassigned): add with the inverted list of every vector given by the caller"""
        x = _f32(x, self.d)
        assign = np.ascontiguousarray(assign, dtype=np.int64)
        if assign.shape != (x.shape[0],):
            raise ValueError("one list number per vector")
        if ids is not None:
            ids = np.ascontiguousarray(ids, dtype=np.int64)
            if ids.shape != (x.shape[0],):
                raise ValueError("ids must have one entry per vector")
        _check(self._lib.faiss_amd_GpuIndexIVF_add_core(self._h, x.shape[0], _ptr(x), _ptr(ids) if ids is not None else None,
                                                        _ptr(assign)))

    def search_preassigned(self, x, k, Iq, Dq, params=None):
        """faiss python `index.search_preassigned(x, k, Iq, Dq, params=None)` (class_wrappers.py
        replacement_search_preassigned): Iq / Dq are the [n, nprobe] list ids / coarse distances of the queries."""
        x = _f32(x, self.d)
        n = x.shape[0]
        Iq = np.ascontiguousarray(Iq, dtype=np.int64)
        Dq = np.ascontiguousarray(Dq, dtype=np.float32)
        if Iq.shape != (n, self.nprobe) or Dq.shape != (n, self.nprobe):
            raise ValueError("Iq and Dq must be [n, nprobe]")
        D = np.empty((n, k), dtype=np.float32)
        I = np.empty((n, k), dtype=np.int64)
        if params is None:
            _check(self._lib.faiss_amd_GpuIndexIVF_search_preassigned(self._h, n, _ptr(x), int(k), _ptr(Iq), _ptr(Dq),
                                                                      _ptr(D), _ptr(I)))
        else:
            with _params_handle(self._lib, params) as h:
                _check(self._lib.faiss_amd_GpuIndexIVF_search_preassigned_with_params(
                    self._h, n, _ptr(x), int(k), _ptr(Iq), _ptr(Dq), h, _ptr(D), _ptr(I)))
        return D, I

    def search(self, x, k, params=None):
        """index.search(x, k, params=SearchParametersIVF(nprobe=..., sel=...)): per-call nprobe (faiss/IndexIVF.h:70-80)
        and / or an IDSelector on the stored ids; a bare integer is taken as nprobe"""
        if params is not None and not isinstance(params, SearchParameters):
            params = SearchParametersIVF(nprobe=int(params))
        return Index.search(self, x, k, params)

    @property
    def stored_vectors(self):
        """vectors held by the lists (ntotal also counts NaN rows that add() skipped, like the reference)"""
        v = ctypes.c_int64(0)
        _check(self._lib.faiss_amd_GpuIndexIVF_stored_vectors(self._h, ctypes.byref(v)))
        return v.value

    def arena_stats(self):
        """(used_rows, hole_rows, allocated_rows) of the list arena"""
        a, b, c = ctypes.c_int64(0), ctypes.c_int64(0), ctypes.c_int64(0)
        _check(self._lib.faiss_amd_GpuIndexIVF_arena_stats(self._h, ctypes.byref(a), ctypes.byref(b), ctypes.byref(c)))
        return a.value, b.value, c.value

    SCAN_AUTO, SCAN_QUERY_MAJOR, SCAN_LIST_MAJOR, SCAN_LIST_MAJOR_F32 = 0, 1, 2, 3

    def set_scan_mode(self, mode):
        """0 = automatic (large batches list-major), 1 = query-major always, 2 = list-major always (IVFFlat / IVFPQ: behind
        the f16 filter, results bit-identical to the query-major scan), 3 = list-major on the f32 matrix pipe (round 3)"""
        _check(self._lib.faiss_amd_GpuIndexIVF_set_scan_mode(self._h, int(mode)))

    def set_use_filter_shadow(self, on):
        """False: the automatic mode never builds the filter sweeps' copy of the lists (query-major serves, same bits)"""
        _check(self._lib.faiss_amd_GpuIndexIVF_set_use_filter_shadow(self._h, 1 if on else 0))

    def resident_bytes(self):
        """(device bytes of the lists, device bytes of the filter sweeps' copies of them)"""
        a, b = ctypes.c_size_t(0), ctypes.c_size_t(0)
        _check(self._lib.faiss_amd_GpuIndexIVF_resident_bytes(self._h, ctypes.byref(a), ctypes.byref(b)))
        return a.value, b.value

    def scan_info(self):
        """(mode set, mode of the last search: 1 query-major / 2 list-major, queries redone after a segment overflow)"""
        a, b, c = ctypes.c_int(0), ctypes.c_int(0), ctypes.c_int64(0)
        _check(self._lib.faiss_amd_GpuIndexIVF_scan_info(self._h, ctypes.byref(a), ctypes.byref(b), ctypes.byref(c)))
        return a.value, b.value, c.value

    def last_scan_arith(self):
        """the `arith` argument under which Oracle.ivf_search restates the last search: 0 = the query-major arithmetic
        (also what the list-major scan behind the f16 filter returns), 1 = the f32 list-major arithmetic"""
        v = ctypes.c_int(0)
        _check(self._lib.faiss_amd_GpuIndexIVF_last_scan_arith(self._h, ctypes.byref(v)))
        return v.value

    def set_lmf_sampling(self, sample_shift=0):
        """sweep 1 of the filter path on the first rows_per_item >> shift rows of every item (0 rule, -1 all rows)"""
        _check(self._lib.faiss_amd_GpuIndexIVF_set_lmf_sampling(self._h, int(sample_shift)))

    def set_lmf_fast_gather(self, on):
        _check(self._lib.faiss_amd_GpuIndexIVFPQ_set_lmf_fast_gather(self._h, int(bool(on))))

    def set_lmf_two_copies(self, on):
        """IVFPQ, PQ64 over d = 128: two-copy codebook of the filter sweeps on / off (A/B; results never change)"""
        _check(self._lib.faiss_amd_GpuIndexIVFPQ_set_lmf_two_copies(self._h, 1 if on else 0))

    def set_lmf_tuning(self, rows_per_item=0, gran_blocks=0, cand_cap=0, min_stride=0):
        """tuning experiments of the list-major scan behind the f16 filter (0 = the built-in rule); results never change"""
        _check(self._lib.faiss_amd_GpuIndexIVF_set_lmf_tuning(self._h, int(rows_per_item), int(gran_blocks), int(cand_cap),
                                                              int(min_stride)))

    def filter_dump(self, x, k, stride, nprobe=None):
        """test hook: (estimates [n][stride] float32 with NaN where no row sits, band [n]) of the f16 filter sweeps -- the
        estimate of the row at scan position p of query q is estimates[q, p] (include/faiss_amd_c.h
        faiss_amd_GpuIndexIVF_test_filter_dump); band[q] = NaN for queries without a bound"""
        x = _f32(x, self.d)
        n = x.shape[0]
        keys = np.empty((n, stride), dtype=np.uint64)
        band = np.full(n, np.nan, dtype=np.float32)
        _check(self._lib.faiss_amd_GpuIndexIVF_test_filter_dump(self._h, n, _ptr(x), int(nprobe or self.nprobe), int(k),
                                                                int(stride), _ptr(keys), _ptr(band)))
        u = (keys >> np.uint64(32)).astype(np.uint32)
        if self.metric_type != METRIC_L2:
            u = ~u
        bits = np.where(u & np.uint32(0x80000000), u & np.uint32(0x7fffffff), ~u).astype(np.uint32)
        est = bits.view(np.float32).copy()
        est[keys == np.uint64(0xffffffffffffffff)] = np.nan
        return est, band

    def list_major_rule(self, n, nprobe=None, k=1):
        v = ctypes.c_int(0)
        _check(self._lib.faiss_amd_GpuIndexIVF_list_major_rule(self._h, int(n), int(nprobe or self.nprobe), int(k), ctypes.byref(v)))
        return bool(v.value)

    def set_use_fused_scan(self, on):
        """test hook: False routes search() through the unfused scan + select kernels"""
        _check(self._lib.faiss_amd_GpuIndexIVF_set_use_fused_scan(self._h, 1 if on else 0))

    @property
    def nlist(self):
        v = ctypes.c_int(0)
        _check(self._lib.faiss_amd_IndexIVF_nlist(self._h, ctypes.byref(v)))
        return v.value

    @property
    def nprobe(self):
        v = ctypes.c_int(0)
        _check(self._lib.faiss_amd_IndexIVF_nprobe(self._h, ctypes.byref(v)))
        return v.value

    @nprobe.setter
    def nprobe(self, v):
        _check(self._lib.faiss_amd_IndexIVF_set_nprobe(self._h, int(v)))

    @property
    def code_size(self):
        v = ctypes.c_size_t(0)
        _check(self._lib.faiss_amd_IndexIVF_code_size(self._h, ctypes.byref(v)))
        return v.value

    def set_clustering(self, niter=10, seed=1234):
        _check(self._lib.faiss_amd_IndexIVF_set_clustering(self._h, int(niter), int(seed)))

    def set_clustering_params(self, **params):
        """GpuIndexIVF.cp: ClusteringParameters of the coarse quantizer's training (niter, nredo, spherical, seed, ...)"""
        cp = ClusteringParameters(**params)
        _check(self._lib.faiss_amd_IndexIVF_set_clustering_params(self._h, ctypes.byref(cp)))

    def get_list_size(self, l):
        v = ctypes.c_size_t(0)
        _check(self._lib.faiss_amd_IndexIVF_get_list_size(self._h, int(l), ctypes.byref(v)))
        return v.value

    def get_list_ids(self, l):
        out = np.empty(self.get_list_size(l), dtype=np.int64)
        _check(self._lib.faiss_amd_IndexIVF_get_list_ids(self._h, int(l), _ptr(out)))
        return out

    def get_list_codes(self, l):
        out = np.empty((self.get_list_size(l), self.code_size), dtype=np.uint8)
        _check(self._lib.faiss_amd_IndexIVF_get_list_codes(self._h, int(l), _ptr(out)))
        return out

    def get_centroids(self):
        out = np.empty((self.nlist, self.d), dtype=np.float32)
        _check(self._lib.faiss_amd_IndexIVF_get_centroids(self._h, _ptr(out)))
        return out

    # copyFrom decomposed into arrays (see include/faiss_amd_c.h)
    def copy_centroids(self, centroids):
        c = _f32(centroids, self.d)
        assert c.shape[0] == self.nlist
        _check(self._lib.faiss_amd_IndexIVF_copy_centroids(self._h, _ptr(c)))

    def copy_lists(self, list_sizes, codes, ids):
        ls = np.ascontiguousarray(list_sizes, dtype=np.uint32)
        codes = np.ascontiguousarray(codes).view(np.uint8).reshape(-1)
        ids = np.ascontiguousarray(ids, dtype=np.int64)
        assert ls.shape == (self.nlist,) and ids.shape == (int(ls.sum()),)
        assert codes.size == ids.size * self.code_size
        _check(self._lib.faiss_amd_IndexIVF_copy_lists(self._h, _ptr(ls), _ptr(codes), _ptr(ids)))


class IDSelector:
    """faiss.IDSelector (faiss/impl/IDSelector.h:21-24): a predicate on the labels a search may return (row numbers for
    GpuIndexFlat, stored ids for the IVF indexes).  Evaluated on the device by the search kernels."""

    def __init__(self):
        self._lib = load_library()
        self._h = ctypes.c_void_p()
        self._keep = []  # operands / arrays that must outlive this selector

    def __del__(self):
        h, self._h = getattr(self, "_h", None), None
        if h:
            self._lib.faiss_amd_IDSelector_free(h)

    def is_member(self, i):
        rc = self._lib.faiss_amd_IDSelector_is_member(self._h, int(i))
        if rc < 0:  # error codes (null / freed handle ...) are negative and must not read as True
            _check(rc)
        return rc == 1

    def __invert__(self):
        return IDSelectorNot(self)

    def __and__(self, other):
        return IDSelectorAnd(self, other)

    def __or__(self, other):
        return IDSelectorOr(self, other)

    def __xor__(self, other):
        return IDSelectorXOr(self, other)


class IDSelectorAll(IDSelector):
    def __init__(self):
        super().__init__()
        _check(self._lib.faiss_amd_IDSelectorAll_new(ctypes.byref(self._h)))


class IDSelectorRange(IDSelector):
    """imin <= id < imax (faiss/impl/IDSelector.h:71-92)"""

    def __init__(self, imin, imax, assume_sorted=False):
        super().__init__()
        self.imin, self.imax = int(imin), int(imax)
        _check(self._lib.faiss_amd_IDSelectorRange_new(ctypes.byref(self._h), self.imin, self.imax))


class IDSelectorBatch(IDSelector):
    """the listed ids (faiss/impl/IDSelector.h:98-141: IDSelectorArray / IDSelectorBatch); the ids are copied"""

    def __init__(self, ids):
        super().__init__()
        ids = np.ascontiguousarray(ids, dtype=np.int64).reshape(-1)
        _check(self._lib.faiss_amd_IDSelectorBatch_new(ctypes.byref(self._h), ids.size, _ptr(ids)))


IDSelectorArray = IDSelectorBatch


class IDSelectorBitmap(IDSelector):
    """id selected iff id // 8 < len(bitmap) and bit id % 8 of bitmap[id // 8] is set (IDSelector.h:145-158); copied"""

    def __init__(self, bitmap):
        super().__init__()
        bitmap = np.ascontiguousarray(bitmap, dtype=np.uint8).reshape(-1)
        _check(self._lib.faiss_amd_IDSelectorBitmap_new(ctypes.byref(self._h), bitmap.size, _ptr(bitmap)))


class IDSelectorNot(IDSelector):
    def __init__(self, sel):
        super().__init__()
        self._keep.append(sel)
        _check(self._lib.faiss_amd_IDSelectorNot_new(ctypes.byref(self._h), sel._h))


class _IDSelectorBinary(IDSelector):
    _ctor = None

    def __init__(self, lhs, rhs):
        super().__init__()
        self._keep += [lhs, rhs]
        _check(getattr(self._lib, self._ctor)(ctypes.byref(self._h), lhs._h, rhs._h))


class IDSelectorAnd(_IDSelectorBinary):
    _ctor = "faiss_amd_IDSelectorAnd_new"


class IDSelectorOr(_IDSelectorBinary):
    _ctor = "faiss_amd_IDSelectorOr_new"


class IDSelectorXOr(_IDSelectorBinary):
    _ctor = "faiss_amd_IDSelectorXOr_new"


class SearchParameters:
    """faiss.SearchParameters (faiss/Index.h:86-93): sel = IDSelector or None"""

    def __init__(self, sel=None):
        self.sel = sel


class SearchParametersIVF(SearchParameters):
    """faiss.SearchParametersIVF (faiss/IndexIVF.h:70-80): nprobe override of one search call (0 = the index's own),
    sel = IDSelector on the stored ids"""

    def __init__(self, nprobe=0, sel=None, max_codes=0):
        super().__init__(sel)
        self.nprobe = int(nprobe)
        self.max_codes = int(max_codes)


class _params_handle:
    """context manager: the C handle of a SearchParameters object for the duration of one call"""

    def __init__(self, lib, params):
        self._lib, self._h = lib, ctypes.c_void_p()
        sel = params.sel._h if params.sel is not None else None
        if isinstance(params, SearchParametersIVF):
            _check(lib.faiss_amd_SearchParametersIVF_new_with(ctypes.byref(self._h), sel, max(params.nprobe, 0),
                                                              params.max_codes))
        else:
            _check(lib.faiss_amd_SearchParameters_new(ctypes.byref(self._h), sel))

    def __enter__(self):
        return self._h

    def __exit__(self, *exc):
        self._lib.faiss_amd_SearchParameters_free(self._h)
        return False


class GpuIndexIVFFlat(_GpuIndexIVF):
    """faiss.GpuIndexIVFFlat (faiss/gpu/GpuIndexIVFFlat.h:33-126)."""

    def __init__(self, res, d, nlist, metric=METRIC_L2, config=None, quantizer=None):
        super().__init__()
        self._keep.append(res)
        _check(self._lib.faiss_amd_GpuIndexIVFFlat_new_with_quantizer(ctypes.byref(self._h), res._h, self._adopt(quantizer), int(d),
                                                                      int(nlist), int(metric),
                                                                      ctypes.byref(config) if config else None))


class GpuIndexIVFPQ(_GpuIndexIVF):
    """faiss.GpuIndexIVFPQ (faiss/gpu/GpuIndexIVFPQ.h:53-176)."""

    def __init__(self, res, d, nlist, M, nbits=8, metric=METRIC_L2, config=None, quantizer=None):
        super().__init__()
        self._keep.append(res)
        self.M = int(M)
        _check(self._lib.faiss_amd_GpuIndexIVFPQ_new_with_quantizer(ctypes.byref(self._h), res._h, self._adopt(quantizer), int(d),
                                                                    int(nlist), int(M), int(nbits), int(metric),
                                                                    ctypes.byref(config) if config else None))

    def _info(self):
        v = [ctypes.c_int(0) for _ in range(4)]
        _check(self._lib.faiss_amd_GpuIndexIVFPQ_getInfo(self._h, *[ctypes.byref(x) for x in v]))
        return [x.value for x in v]

    def setPrecomputedCodes(self, enable):
        """GpuIndexIVFPQ.setPrecomputedCodes: kept and reported; the per-vector term it stands for is always on here"""
        _check(self._lib.faiss_amd_GpuIndexIVFPQ_setPrecomputedCodes(self._h, int(bool(enable))))

    def getPrecomputedCodes(self):
        """what is IN FORCE: True for L2 (the per-vector term is always used), False for inner product"""
        return bool(self._info()[0])

    def getTableInfo(self):
        """(precomputed codes in force, precomputed codes requested, fp16 lookup tables in force -- always False)"""
        a, b, c = ctypes.c_int(0), ctypes.c_int(0), ctypes.c_int(0)
        _check(self._lib.faiss_amd_GpuIndexIVFPQ_getTableInfo(self._h, ctypes.byref(a), ctypes.byref(b), ctypes.byref(c)))
        return bool(a.value), bool(b.value), bool(c.value)

    def getNumSubQuantizers(self):
        return self._info()[1]

    def getBitsPerCode(self):
        return self._info()[2]

    def getCentroidsPerSubQuantizer(self):
        return self._info()[3]

    def copy_pq_centroids(self, pq):
        pq = np.ascontiguousarray(pq, dtype=np.float32).reshape(-1)
        assert pq.size == self.M * 256 * (self.d // self.M)
        _check(self._lib.faiss_amd_IndexIVFPQ_copy_pq_centroids(self._h, _ptr(pq)))

    def get_pq_centroids(self):
        out = np.empty((self.M, 256, self.d // self.M), dtype=np.float32)
        _check(self._lib.faiss_amd_IndexIVFPQ_get_pq_centroids(self._h, _ptr(out)))
        return out


class ScalarQuantizer:
    """faiss.ScalarQuantizer.QuantizerType values (faiss/impl/ScalarQuantizer.h:27-34) of the types the GPU index stores"""
    QT_8bit, QT_4bit, QT_8bit_uniform, QT_4bit_uniform, QT_fp16, QT_8bit_direct, QT_6bit = range(7)
    RS_minmax, RS_meanstd, RS_quantiles, RS_optim = range(4)


class GpuIndexIVFScalarQuantizer(_GpuIndexIVF):
    """faiss.GpuIndexIVFScalarQuantizer (faiss/gpu/GpuIndexIVFScalarQuantizer.h:27-131)."""

    def __init__(self, res, d, nlist, qtype, metric=METRIC_L2, encodeResidual=True, config=None, quantizer=None):
        super().__init__()
        self._keep.append(res)
        _check(self._lib.faiss_amd_GpuIndexIVFScalarQuantizer_new_with_quantizer(
            ctypes.byref(self._h), res._h, self._adopt(quantizer), int(d), int(nlist), int(qtype), int(metric),
            int(bool(encodeResidual)), ctypes.byref(config) if config else None))

    def _info(self):
        qt, br = ctypes.c_int(0), ctypes.c_int(0)
        cs, ts = ctypes.c_size_t(0), ctypes.c_size_t(0)
        _check(self._lib.faiss_amd_IndexIVFSQ_info(self._h, ctypes.byref(qt), ctypes.byref(br), ctypes.byref(cs),
                                                   ctypes.byref(ts)))
        return qt.value, bool(br.value), cs.value, ts.value

    qtype = property(lambda self: self._info()[0])
    by_residual = property(lambda self: self._info()[1])

    def get_trained(self):
        """index.sq.trained: {vmin, vdiff} (uniform types) or vmin[d] then vdiff[d]"""
        out = np.empty(self._info()[3], dtype=np.float32)
        if out.size:
            _check(self._lib.faiss_amd_IndexIVFSQ_get_trained(self._h, _ptr(out)))
        return out

    def copy_trained(self, trained):
        t = np.ascontiguousarray(trained, dtype=np.float32).reshape(-1)
        _check(self._lib.faiss_amd_IndexIVFSQ_copy_trained(self._h, _ptr(t), t.size))

    def set_rangestat(self, rangestat, rangestat_arg=0.0):
        _check(self._lib.faiss_amd_IndexIVFSQ_set_rangestat(self._h, int(rangestat), float(rangestat_arg)))


class IndexShards(Index):
    """faiss.IndexShards (faiss/IndexShards.h:19-108): host threads + host merge."""

    def __init__(self, d, threaded=True, successive_ids=True):
        super().__init__()
        _check(self._lib.faiss_amd_IndexShards_new(ctypes.byref(self._h), int(d), int(threaded),
                                                   int(successive_ids)))

    def add_shard(self, index):
        self._keep.append(index)
        _check(self._lib.faiss_amd_IndexShards_add_shard(self._h, index._h))


class IndexReplicas(Index):
    """faiss.IndexReplicas (faiss/IndexReplicas.h:20-82): every replica holds the database, queries are split."""

    def __init__(self, d, threaded=True):
        super().__init__()
        _check(self._lib.faiss_amd_IndexReplicas_new(ctypes.byref(self._h), int(d), int(threaded)))

    def add_replica(self, index):
        self._keep.append(index)
        _check(self._lib.faiss_amd_IndexReplicas_add_replica(self._h, index._h))

    addIndex = add_replica


def add_preassigned(index_ivf, x, a, ids=None):
    """faiss.contrib.ivf_tools.add_preassigned (contrib/ivf_tools.py:12-24)"""
    index_ivf.add_core(x, a, ids)


def sq_train_rangestat(qtype, rangestat, rangestat_arg, x):
    """test hook: `trained` of faiss.ScalarQuantizer(d, qtype) trained on x with RS_meanstd / RS_quantiles / RS_optim (host code)"""
    x = _f32(x)
    uniform = qtype in (ScalarQuantizer.QT_8bit_uniform, ScalarQuantizer.QT_4bit_uniform)
    out = np.empty(2 if uniform else 2 * x.shape[1], dtype=np.float32)
    _check(load_library().faiss_amd_sq_train_rangestat(int(qtype), int(rangestat), float(rangestat_arg), x.shape[0], x.shape[1],
                                                       _ptr(x), _ptr(out)))
    return out


def kmeans(res, x, k, niter=25, seed=1234):
    """faiss.Kmeans-style helper: returns (centroids [k, d], objective per iteration)."""
    lib = load_library()
    x = _f32(x)
    n, d = x.shape
    cent = np.empty((k, d), dtype=np.float32)
    obj = np.empty(niter, dtype=np.float32)
    _check(lib.faiss_amd_kmeans_clustering(res._h, d, n, int(k), _ptr(x), int(niter), int(seed), _ptr(cent),
                                           _ptr(obj)))
    return cent, obj


class ClusteringParameters(ctypes.Structure):
    """faiss.ClusteringParameters (faiss/Clustering.h:27-60) as the C ABI takes it"""
    _fields_ = [(n, ctypes.c_int) for n in ("niter", "nredo", "verbose", "spherical", "int_centroids", "update_index",
                                            "frozen_centroids", "min_points_per_centroid", "max_points_per_centroid", "seed")]

    def __init__(self, **kw):
        super().__init__()
        load_library().faiss_amd_ClusteringParameters_init(ctypes.byref(self))
        for k_, v in kw.items():
            if k_ not in dict(self._fields_):
                raise TypeError("unknown clustering parameter " + k_)
            setattr(self, k_, int(v))


class Clustering:
    """faiss.Clustering (faiss/Clustering.h:88-196): k-means with an index as the assignment engine.

    c = Clustering(d, k, niter=..., seed=..., nredo=..., spherical=..., int_centroids=..., frozen_centroids=...);
    c.centroids = initial centroids (optional); c.train(x, index); c.centroids, c.obj (objective per iteration of the
    winning run).  With a GpuIndexFlat the loop runs on the device (c.on_device is True)."""

    def __init__(self, d, k, niter=25, seed=1234, **params):
        self.d, self.k, self.niter, self.seed = int(d), int(k), int(niter), int(seed)
        self.params = params
        self.centroids = None
        self.obj = None
        self.on_device = None

    def train(self, x, index):
        x = _f32(x)
        n, d = x.shape
        assert d == self.d
        cent = np.empty((self.k, d), dtype=np.float32)
        obj = np.empty(self.niter, dtype=np.float32)
        dev = ctypes.c_int(0)
        cp = ClusteringParameters(niter=self.niter, seed=self.seed, **self.params)
        init = None if self.centroids is None else _f32(self.centroids, self.d)
        _check(load_library().faiss_amd_Clustering_train_ex(index._h, n, _ptr(x), self.k, ctypes.byref(cp),
                                                            _ptr(init) if init is not None and len(init) else None,
                                                            0 if init is None else len(init), _ptr(cent), _ptr(obj),
                                                            ctypes.byref(dev)))
        self.centroids, self.obj, self.on_device = cent, obj, bool(dev.value)


class GpuParameterSpace:
    """faiss.GpuParameterSpace (faiss/gpu/GpuAutoTune.h / .cpp:40-114): the tunable parameters of a backend index"""

    def initialize(self, index):
        """GpuParameterSpace::initialize (faiss/gpu/GpuAutoTune.cpp:33-77): nprobe = 1, 2, 4, ... below nlist and up to the k-selection
        limit of 2048 (at most 12 values)"""
        self.parameter_ranges = {}
        nl = getattr(index, "nlist", None)
        if nl:
            vals = []
            for i in range(12):
                v = 1 << i
                if v >= nl or v > 2048:
                    break
                vals.append(v)
            self.parameter_ranges["nprobe"] = vals
        return self.parameter_ranges

    def explore(self, index, xq, k, gt, criterion="1-recall@1"):
        """ParameterSpace::explore (faiss/AutoTune.cpp:632-737) over the ranges of initialize(): every combination is applied and timed
        on `index` (one warm-up search, then the best of three), perf = the share of queries whose true nearest neighbour gt[q] is the
        first result.  Returns the OPTIMAL operating points [(perf, seconds, 'nprobe=..')], sorted by perf, like OperatingPoints."""
        import time as _time
        ranges = self.initialize(index)
        names = sorted(ranges)
        combos = [()]
        for nm in names:
            combos = [c + ((nm, v),) for c in combos for v in ranges[nm]]
        gt = np.asarray(gt).reshape(-1)
        pts = []
        for combo in combos:
            for nm, v in combo:
                self.set_index_parameter(index, nm, v)
            index.search(xq, k)
            best = float("inf")
            for _ in range(3):
                t0 = _time.perf_counter()
                _, I = index.search(xq, k)
                best = min(best, _time.perf_counter() - t0)
            pts.append((float((I[:, 0] == gt).mean()), best, ",".join("%s=%d" % (nm, v) for nm, v in combo)))
        pts.sort(key=lambda p: (p[1], -p[0]))
        optimal, top = [], -1.0
        for p in pts:  # a point is optimal when nothing faster reaches its perf
            if p[0] > top:
                optimal.append(p)
                top = p[0]
        return sorted(optimal)

    def set_index_parameter(self, index, name, val):
        _check(load_library().faiss_amd_GpuParameterSpace_set_index_parameter(index._h, name.encode(), float(val)))

    def set_index_parameters(self, index, description):
        """'nprobe=32,use_precomputed_table=1'"""
        for item in description.split(","):
            if item.strip():
                name, val = item.split("=")
                self.set_index_parameter(index, name.strip(), float(val))


_INTERRUPT_KEEP = []


def set_interrupt_callback(fn):
    """faiss.InterruptCallback: fn() -> truthy aborts the running call with FaissAmdError; None removes the hook"""
    lib = load_library()
    if fn is None:
        _check(lib.faiss_amd_set_interrupt_callback(None, None))
        _INTERRUPT_KEEP.clear()
        return
    cb = ctypes.CFUNCTYPE(ctypes.c_int, ctypes.c_void_p)(lambda _u: 1 if fn() else 0)
    old = list(_INTERRUPT_KEEP)
    _INTERRUPT_KEEP.append(cb)  # alive before the C side can call it ...
    _check(lib.faiss_amd_set_interrupt_callback(ctypes.cast(cb, ctypes.c_void_p), None))
    for o in old:  # ... and the previous thunk is dropped only after the C side has swapped the pointer
        _INTERRUPT_KEEP.remove(o)


class GpuDistanceParams(ctypes.Structure):
    """faiss.GpuDistanceParams (faiss/gpu/GpuDistance.h:32-152) as the C ABI takes it"""
    _fields_ = [("metric", ctypes.c_int), ("metricArg", ctypes.c_float), ("k", ctypes.c_int), ("dims", ctypes.c_int),
                ("vectors", ctypes.c_void_p), ("vectorType", ctypes.c_int), ("vectorsRowMajor", ctypes.c_int),
                ("numVectors", ctypes.c_int64), ("vectorNorms", ctypes.c_void_p), ("queries", ctypes.c_void_p),
                ("queryType", ctypes.c_int), ("queriesRowMajor", ctypes.c_int), ("numQueries", ctypes.c_int64),
                ("outDistances", ctypes.c_void_p), ("ignoreOutDistances", ctypes.c_int), ("outIndicesType", ctypes.c_int),
                ("outIndices", ctypes.c_void_p), ("device", ctypes.c_int)]


def _matrix_arg(x, name):
    """(pointer, DistanceDataType, row_major, n, d) of a float32 / float16 matrix in C or Fortran order"""
    x = np.asarray(x)
    if x.dtype not in (np.float32, np.float16) or x.ndim != 2:
        raise TypeError("%s must be a 2-D float32 or float16 array" % name)
    if x.flags.c_contiguous:
        row_major = True
    elif x.flags.f_contiguous:
        row_major = False
    else:
        x, row_major = np.ascontiguousarray(x), True
    return x, (1 if x.dtype == np.float32 else 2), row_major


def knn_gpu(res, xq, xb, k, D=None, I=None, metric=METRIC_L2, vectorsMemoryLimit=0, queriesMemoryLimit=0, metric_arg=0.0):
    """faiss.knn_gpu (faiss/python/gpu_wrappers.py:56-206): brute-force k-NN of xq in xb -> (D, I).  float32 or float16
    inputs in row- or column-major order, int64 or int32 labels (dtype of a supplied I); k = -1 returns the full
    distance matrix as D (I is None); the memory limits select bfKnn_tiling."""
    lib = load_library()
    xb, vt, vrm = _matrix_arg(xb, "xb")
    xq, qt, qrm = _matrix_arg(xq, "xq")
    if xb.shape[1] != xq.shape[1]:
        raise ValueError("dimension mismatch")
    nq, d = xq.shape
    if k == -1:
        D = np.empty((nq, xb.shape[0]), dtype=np.float32)
        I = None
    else:
        if D is None:
            D = np.empty((nq, k), dtype=np.float32)
        if I is None:
            I = np.empty((nq, k), dtype=np.int64)
        if D.shape != (nq, k) or I.shape != (nq, k) or D.dtype != np.float32 or I.dtype not in (np.int64, np.int32):
            raise ValueError("D / I have the wrong shape or dtype")
    a = GpuDistanceParams(int(metric), float(metric_arg), int(k), int(d), xb.ctypes.data, vt, int(vrm), xb.shape[0], None,
                          xq.ctypes.data, qt, int(qrm), nq, D.ctypes.data, 0,
                          2 if (I is not None and I.dtype == np.int32) else 1, I.ctypes.data if I is not None else None, -1)
    if vectorsMemoryLimit or queriesMemoryLimit:
        _check(lib.faiss_amd_bfKnn_tiling(res._h, ctypes.byref(a), int(vectorsMemoryLimit), int(queriesMemoryLimit)))
    else:
        _check(lib.faiss_amd_bfKnn_params(res._h, ctypes.byref(a)))
    return D, I


def test_select(res, which, vals, k, metric=METRIC_L2):
    """test hook: k best per row of `vals` through one selection primitive (include/faiss_amd_c.h faiss_amd_test_select)"""
    lib = load_library()
    vals = _f32(vals)
    rows, cols = vals.shape
    D = np.empty((rows, k), dtype=np.float32)
    I = np.empty((rows, k), dtype=np.int64)
    _check(lib.faiss_amd_test_select(res._h, int(which), int(metric), rows, cols, int(k), _ptr(vals), _ptr(D), _ptr(I)))
    return D, I


test_select.__test__ = False  # (not a pytest test)


def merge_knn_results(metric, all_D, all_I, base=None):
    """faiss.merge_knn_results: all_D/all_I are [nshard, n, k]; returns merged (D, I)."""
    lib = load_library()
    all_D = np.ascontiguousarray(all_D, dtype=np.float32)
    all_I = np.ascontiguousarray(all_I, dtype=np.int64)
    ns, n, k = all_D.shape
    D = np.empty((n, k), dtype=np.float32)
    I = np.empty((n, k), dtype=np.int64)
    b = None if base is None else np.ascontiguousarray(base, dtype=np.int64)
    _check(lib.faiss_amd_merge_knn_results(int(metric), n, k, ns, _ptr(all_D), _ptr(all_I), _ptr(b), _ptr(D),
                                           _ptr(I)))
    return D, I


def merge_knn_results_device(res, metric, n, k, nshard, all_d_ptr, all_i_ptr, base, d_ptr, i_ptr):
    """Device-side shard merge; *_ptr are device addresses on res's device, base a host list/array."""
    lib = load_library()
    b = None if base is None else np.ascontiguousarray(base, dtype=np.int64)
    _check(lib.faiss_amd_merge_knn_results_device(res._h, int(metric), int(n), int(k), int(nshard),
                                                  ctypes.c_void_p(all_d_ptr), ctypes.c_void_p(all_i_ptr), _ptr(b),
                                                  ctypes.c_void_p(d_ptr), ctypes.c_void_p(i_ptr)))
