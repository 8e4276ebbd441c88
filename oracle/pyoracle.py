"""oracle/pyoracle.py -- ctypes access to the oracles.  TEST INFRASTRUCTURE ONLY.

Only tests/, __graft_entry__.smoke() and bench.py's cpu_baseline leg may import this module;
nothing under faiss_amd/ does.  Two oracles:

* ``Oracle``  : oracle/libfaiss_oracle.so, the plain-C restatement (faiss_oracle.c) whose
                summation order equals the gfx950 kernels' -> bit-exact comparisons.
* ``Ref``     : oracle/_ref/libfaiss_ref.so, the UNMODIFIED reference (faiss v1.15.0 CPU
                path) compiled from /root/reference by oracle/Makefile.ref.  It is built in
                the dev container and travels to the GPU box as a prebuilt .so; when it is
                absent, ``Ref.available()`` is False and the tests that need it skip.
"""
import ctypes
import os

import sys

import numpy as np

_HERE = os.path.dirname(os.path.abspath(__file__))
if os.path.dirname(_HERE) not in sys.path:
    sys.path.insert(0, os.path.dirname(_HERE))
ORACLE_SO = os.path.join(_HERE, "libfaiss_oracle.so")
REF_SO = os.path.join(_HERE, "_ref", "libfaiss_ref.so")

METRIC_INNER_PRODUCT = 0
METRIC_L2 = 1


def _p(a):
    return ctypes.c_void_p(a.ctypes.data) if a is not None else None


def effective_cores():
    """CPUs this process may really use: scheduler affinity capped by the cgroup CPU quota"""
    try:
        n = len(os.sched_getaffinity(0))
    except AttributeError:
        n = os.cpu_count() or 1
    try:
        quota, period = open("/sys/fs/cgroup/cpu.max").read().split()[:2]
        if quota != "max":
            n = max(1, min(n, int(int(quota) // int(period))))
    except (OSError, ValueError):
        pass
    return n


def _f32(x):
    return np.ascontiguousarray(x, dtype=np.float32)


# both libraries size their OpenMP pools when they are loaded: keep them inside the CPU quota (see effective_cores)
os.environ.setdefault("OMP_NUM_THREADS", str(effective_cores()))


class Oracle:
    _lib = None

    @classmethod
    def lib(cls):
        if cls._lib is None:
            if not os.path.exists(ORACLE_SO):
                raise RuntimeError("%s missing: run `make -C oracle`" % ORACLE_SO)
            lib = ctypes.CDLL(ORACLE_SO)
            lib.orc_ip_chain.restype = ctypes.c_float
            lib.orc_kmeans_objective.restype = ctypes.c_double
            cls._lib = lib
        return cls._lib

    @classmethod
    def set_pair_order(cls, swapped):
        cls.lib().orc_set_pair_order(ctypes.c_int(int(swapped)))

    @classmethod
    def pairwise(cls, metric, xb, xq):
        xb, xq = _f32(xb), _f32(xq)
        out = np.empty((xq.shape[0], xb.shape[0]), dtype=np.float32)
        cls.lib().orc_pairwise(ctypes.c_int(metric), ctypes.c_int(xb.shape[1]), ctypes.c_int64(xb.shape[0]),
                               _p(xb), ctypes.c_int64(xq.shape[0]), _p(xq), _p(out))
        return out

    @classmethod
    def flat_search(cls, metric, xb, xq, k):
        xb, xq = _f32(xb), _f32(xq)
        d = xq.shape[1]
        D = np.empty((xq.shape[0], k), dtype=np.float32)
        I = np.empty((xq.shape[0], k), dtype=np.int64)
        rc = cls.lib().orc_flat_search(ctypes.c_int(metric), ctypes.c_int(d), ctypes.c_int64(xb.shape[0]),
                                       _p(xb) if xb.size else None, ctypes.c_int64(xq.shape[0]), _p(xq),
                                       ctypes.c_int(k), _p(D), _p(I))
        assert rc == 0
        return D, I

    @classmethod
    def flat_search_general(cls, metric, xb, xq, k, metric_arg=0.0):
        """IndexFlat(d, metric).search for the extra metrics (faiss.METRIC_L1 = 2, Linf 3, Lp 4, Canberra 20,
        BrayCurtis 21, JensenShannon 22, Jaccard 23)"""
        xb, xq = _f32(xb), _f32(xq)
        D = np.empty((xq.shape[0], k), dtype=np.float32)
        I = np.empty((xq.shape[0], k), dtype=np.int64)
        rc = cls.lib().orc_flat_search_general(ctypes.c_int(metric), ctypes.c_float(metric_arg), ctypes.c_int(xq.shape[1]),
                                               ctypes.c_int64(xb.shape[0]), _p(xb) if xb.size else None,
                                               ctypes.c_int64(xq.shape[0]), _p(xq), ctypes.c_int(k), _p(D), _p(I))
        assert rc == 0
        return D, I

    @classmethod
    def ivf_search(cls, kind, metric, centroids, list_sizes, codes, ids, xq, nprobe, k, M=0, pq=None, arith=0):
        """arith: 0 = the query-major scans, 1 = the list-major scan of large batches (faiss_oracle.c orc_ivf_search_ex;
        GpuIndexIVF.last_scan_mode() - 1 says which one served a search)"""
        centroids, xq = _f32(centroids), _f32(xq)
        nlist, d = centroids.shape
        ls = np.ascontiguousarray(list_sizes, dtype=np.uint32)
        codes = np.ascontiguousarray(codes).view(np.uint8).reshape(-1)
        ids = np.ascontiguousarray(ids, dtype=np.int64)
        pqc = _f32(pq).reshape(-1) if pq is not None else None
        nq = xq.shape[0]
        D = np.empty((nq, k), dtype=np.float32)
        I = np.empty((nq, k), dtype=np.int64)
        np_eff = min(nprobe, nlist)
        cD = np.empty((nq, np_eff), dtype=np.float32)
        cI = np.empty((nq, np_eff), dtype=np.int64)
        rc = cls.lib().orc_ivf_search_ex(ctypes.c_int(kind), ctypes.c_int(metric), ctypes.c_int(d),
                                         ctypes.c_int(nlist), _p(centroids), _p(ls), _p(codes), _p(ids),
                                         ctypes.c_int(M), _p(pqc), ctypes.c_int64(nq), _p(xq),
                                         ctypes.c_int(nprobe), ctypes.c_int(k), _p(D), _p(I), _p(cD), _p(cI),
                                         ctypes.c_int(arith))
        assert rc == 0
        return D, I, cD, cI

    @classmethod
    def ivf_assign(cls, metric, centroids, x):
        centroids, x = _f32(centroids), _f32(x)
        lab = np.empty(x.shape[0], dtype=np.int64)
        rc = cls.lib().orc_ivf_assign(ctypes.c_int(metric), ctypes.c_int(x.shape[1]),
                                      ctypes.c_int(centroids.shape[0]), _p(centroids),
                                      ctypes.c_int64(x.shape[0]), _p(x), _p(lab))
        assert rc == 0
        return lab

    @classmethod
    def pq_encode(cls, pq, centroids, x, labels):
        pq, centroids, x = _f32(pq), _f32(centroids), _f32(x)
        M = pq.shape[0]
        labels = np.ascontiguousarray(labels, dtype=np.int64)
        codes = np.empty((x.shape[0], M), dtype=np.uint8)
        rc = cls.lib().orc_pq_encode(ctypes.c_int(x.shape[1]), ctypes.c_int(M), _p(pq.reshape(-1)), _p(centroids),
                                     ctypes.c_int64(x.shape[0]), _p(x), _p(labels), _p(codes))
        assert rc == 0
        return codes

    # ---- IVF scalar quantizer (faiss/impl/ScalarQuantizer.h qtype values; vmin / vdiff are [d], see sq_unpack)
    @staticmethod
    def sq_code_size(qtype, d):
        return {1: (d + 1) // 2, 3: (d + 1) // 2, 4: 2 * d, 6: (6 * d + 7) // 8}.get(qtype, d)

    @staticmethod
    def sq_unpack(qtype, d, trained):
        """ScalarQuantizer::trained -> (vmin[d], vdiff[d]); the types without a trained range give zeros"""
        t = np.asarray(trained, dtype=np.float32).reshape(-1)
        if qtype in (2, 3):
            return np.full(d, t[0], np.float32), np.full(d, t[1], np.float32)
        if qtype in (0, 1, 6):
            return t[:d].copy(), t[d:2 * d].copy()
        return np.zeros(d, np.float32), np.zeros(d, np.float32)

    @classmethod
    def sq_encode(cls, qtype, x, vmin, vdiff, labels=None, centroids=None):
        x = _f32(x)
        n, d = x.shape
        by_res = centroids is not None
        lab = np.ascontiguousarray(labels, dtype=np.int64) if by_res else None
        cen = _f32(centroids) if by_res else None
        codes = np.empty((n, cls.sq_code_size(qtype, d)), dtype=np.uint8)
        rc = cls.lib().orc_sq_encode(ctypes.c_int(qtype), ctypes.c_int(d), ctypes.c_int64(n), _p(x), _p(lab), _p(cen),
                                     ctypes.c_int(int(by_res)), _p(_f32(vmin)), _p(_f32(vdiff)), _p(codes))
        assert rc == 0
        return codes

    @classmethod
    def sq_decode(cls, qtype, d, codes, vmin, vdiff):
        codes = np.ascontiguousarray(codes, dtype=np.uint8).reshape(-1, cls.sq_code_size(qtype, d))
        out = np.empty((codes.shape[0], d), dtype=np.float32)
        rc = cls.lib().orc_sq_decode(ctypes.c_int(qtype), ctypes.c_int(d), ctypes.c_int64(codes.shape[0]), _p(codes),
                                     _p(_f32(vmin)), _p(_f32(vdiff)), _p(out))
        assert rc == 0
        return out

    @classmethod
    def ivfsq_search(cls, qtype, by_residual, metric, centroids, list_sizes, codes, ids, vmin, vdiff, xq, nprobe, k, arith=0):
        """arith: 0 = the query-major scan (ivfsq_fused_kernel), 1 = the list-major scan of large batches
        (faiss_oracle.c orc_ivfsq_search_ex)"""
        centroids, xq = _f32(centroids), _f32(xq)
        nlist, d = centroids.shape
        ls = np.ascontiguousarray(list_sizes, dtype=np.uint32)
        codes = np.ascontiguousarray(codes).view(np.uint8).reshape(-1)
        ids = np.ascontiguousarray(ids, dtype=np.int64)
        nq = xq.shape[0]
        D = np.empty((nq, k), dtype=np.float32)
        I = np.empty((nq, k), dtype=np.int64)
        rc = cls.lib().orc_ivfsq_search_ex(ctypes.c_int(qtype), ctypes.c_int(int(by_residual)), ctypes.c_int(metric),
                                           ctypes.c_int(d), ctypes.c_int(nlist), _p(centroids), _p(ls), _p(codes), _p(ids),
                                           _p(_f32(vmin)), _p(_f32(vdiff)), ctypes.c_int64(nq), _p(xq), ctypes.c_int(nprobe),
                                           ctypes.c_int(k), _p(D), _p(I), ctypes.c_int(arith))
        assert rc == 0
        return D, I

    @classmethod
    def merge_shards(cls, metric, all_D, all_I, base=None):
        all_D = np.ascontiguousarray(all_D, dtype=np.float32)
        all_I = np.ascontiguousarray(all_I, dtype=np.int64)
        ns, nq, k = all_D.shape
        D = np.empty((nq, k), dtype=np.float32)
        I = np.empty((nq, k), dtype=np.int64)
        b = None if base is None else np.ascontiguousarray(base, dtype=np.int64)
        rc = cls.lib().orc_merge_shards(ctypes.c_int(metric), ctypes.c_int64(nq), ctypes.c_int(k),
                                        ctypes.c_int(ns), _p(all_D), _p(all_I), _p(b), _p(D), _p(I))
        assert rc == 0
        return D, I

    @classmethod
    def kmeans_objective(cls, x, centroids):
        x, centroids = _f32(x), _f32(centroids)
        return cls.lib().orc_kmeans_objective(ctypes.c_int(x.shape[1]), ctypes.c_int64(x.shape[0]), _p(x),
                                              ctypes.c_int(centroids.shape[0]), _p(centroids))

    @classmethod
    def build_ivf_lists(cls, kind, metric, centroids, x, ids=None, pq=None):
        """IndexIVF::add on the restatement: returns (list_sizes, codes, ids) in list order,
        entries in insertion order inside each list (faiss/IndexIVF.cpp:194-260)."""
        x = _f32(x)
        n = x.shape[0]
        ids = np.arange(n, dtype=np.int64) if ids is None else np.asarray(ids, dtype=np.int64)
        lab = cls.ivf_assign(metric, centroids, x)
        nlist = centroids.shape[0]
        order = np.argsort(lab, kind="stable")
        order = order[lab[order] >= 0]
        sizes = np.bincount(lab[lab >= 0], minlength=nlist).astype(np.uint32)
        if kind == 0:
            codes = x[order].view(np.uint8).reshape(len(order), -1)
        else:
            codes = cls.pq_encode(pq, centroids, x, np.maximum(lab, 0))[order]
        return sizes, np.ascontiguousarray(codes), ids[order], lab


class RefIndex:
    def __init__(self, lib, h, d):
        self.lib, self.h, self.d = lib, h, d

    def __del__(self):
        if getattr(self, "h", None):
            self.lib.ref_index_free(ctypes.c_void_p(self.h))
            self.h = None

    def _ck(self, rc):
        if rc != 0:
            raise RuntimeError("reference error: " + self.lib.ref_last_error().decode())

    def train(self, x):
        x = _f32(x)
        self._ck(self.lib.ref_index_train(ctypes.c_void_p(self.h), ctypes.c_int64(x.shape[0]), _p(x)))

    def add(self, x):
        x = _f32(x)
        self._ck(self.lib.ref_index_add(ctypes.c_void_p(self.h), ctypes.c_int64(x.shape[0]), _p(x)))

    def add_with_ids(self, x, ids):
        x = _f32(x)
        ids = np.ascontiguousarray(ids, dtype=np.int64)
        self._ck(self.lib.ref_index_add_with_ids(ctypes.c_void_p(self.h), ctypes.c_int64(x.shape[0]), _p(x), _p(ids)))

    def search(self, x, k):
        x = _f32(x)
        D = np.empty((x.shape[0], k), dtype=np.float32)
        I = np.empty((x.shape[0], k), dtype=np.int64)
        self._ck(self.lib.ref_index_search(ctypes.c_void_p(self.h), ctypes.c_int64(x.shape[0]), _p(x),
                                           ctypes.c_int64(k), _p(D), _p(I)))
        return D, I

    def reset(self):
        self._ck(self.lib.ref_index_reset(ctypes.c_void_p(self.h)))

    def set_metric_arg(self, arg):
        """index.metric_arg (the p of METRIC_Lp)"""
        self._ck(self.lib.ref_index_set_metric_arg(ctypes.c_void_p(self.h), ctypes.c_float(arg)))

    def search_nprobe(self, x, k, nprobe):
        """index.search(x, k, params=SearchParametersIVF(nprobe=nprobe))"""
        x = _f32(x)
        D = np.empty((x.shape[0], k), dtype=np.float32)
        I = np.empty((x.shape[0], k), dtype=np.int64)
        self._ck(self.lib.ref_index_search_nprobe(ctypes.c_void_p(self.h), ctypes.c_int64(x.shape[0]), _p(x),
                                                  ctypes.c_int64(k), ctypes.c_int(nprobe), _p(D), _p(I)))
        return D, I

    def add_core(self, x, assign, ids=None):
        """contrib.ivf_tools.add_preassigned: IndexIVF::add_core (or the bridge's) with the list of every vector given"""
        x = _f32(x)
        assign = np.ascontiguousarray(assign, dtype=np.int64)
        ids = None if ids is None else np.ascontiguousarray(ids, dtype=np.int64)
        self._ck(self.lib.ref_ivf_add_core(ctypes.c_void_p(self.h), ctypes.c_int64(x.shape[0]), _p(x), _p(ids), _p(assign)))

    def search_sel(self, x, k, kind, a=0, b=0, data=None, negate=False, nprobe=0):
        """index.search(x, k, params=SearchParameters[IVF](sel=...)) with a faiss::IDSelector described by scalars
        (oracle/ref_shim.cpp ref_index_search_sel): kind 0 Range [a, b), 1 Batch(ids), 2 Array(ids), 3 Bitmap(bytes),
        4 And(Range [a, b), Not(Batch(ids))), 5 a custom selector id % a == b, 6 XOr(Range [a, b), Or(Bitmap, All))"""
        x = _f32(x)
        D = np.empty((x.shape[0], k), dtype=np.float32)
        I = np.empty((x.shape[0], k), dtype=np.int64)
        if data is None:
            data = np.zeros(0, dtype=np.int64)
        data = np.ascontiguousarray(data, dtype=np.uint8 if kind in (3, 6) else np.int64).reshape(-1)
        self._ck(self.lib.ref_index_search_sel(ctypes.c_void_p(self.h), ctypes.c_int64(x.shape[0]), _p(x), ctypes.c_int64(k),
                                               ctypes.c_int(nprobe), ctypes.c_int(kind), ctypes.c_int64(a), ctypes.c_int64(b),
                                               ctypes.c_size_t(data.size), _p(data), ctypes.c_int(int(negate)), _p(D), _p(I)))
        return D, I

    def assign(self, x, k=1):
        x = _f32(x)
        I = np.empty((x.shape[0], k), dtype=np.int64)
        self._ck(self.lib.ref_index_assign(ctypes.c_void_p(self.h), ctypes.c_int64(x.shape[0]), _p(x), _p(I),
                                           ctypes.c_int64(k)))
        return I

    def reconstruct_n(self, i0, ni):
        out = np.empty((ni, self.d), dtype=np.float32)
        self._ck(self.lib.ref_index_reconstruct_n(ctypes.c_void_p(self.h), ctypes.c_int64(i0), ctypes.c_int64(ni), _p(out)))
        return out

    def compute_residual_n(self, x, keys):
        x = _f32(x)
        keys = np.ascontiguousarray(keys, dtype=np.int64)
        out = np.empty_like(x)
        self._ck(self.lib.ref_index_compute_residual_n(ctypes.c_void_p(self.h), ctypes.c_int64(x.shape[0]), _p(x), _p(out),
                                                       _p(keys)))
        return out

    def type_name(self):
        buf = ctypes.create_string_buffer(256)
        self._ck(self.lib.ref_index_type(ctypes.c_void_p(self.h), buf, ctypes.c_int(256)))
        return buf.value.decode()

    @property
    def is_trained(self):
        return bool(self.lib.ref_index_is_trained(ctypes.c_void_p(self.h)))

    @property
    def ntotal(self):
        return self.lib.ref_index_ntotal(ctypes.c_void_p(self.h))

    # IVF -----------------------------------------------------------------------------
    def set_nprobe(self, nprobe):
        self._ck(self.lib.ref_ivf_set_nprobe(ctypes.c_void_p(self.h), ctypes.c_int(nprobe)))

    def set_trained(self, centroids, pq_centroids):
        """install coarse centroids [nlist, d] and a PQ codebook [M, 256, dsub] into an untrained IVFPQ index"""
        c, pq = _f32(centroids), np.ascontiguousarray(pq_centroids, dtype=np.float32)
        self._ck(self.lib.ref_ivfpq_set_trained(ctypes.c_void_p(self.h), _p(c), _p(pq)))

    def set_centroids(self, centroids):
        """install coarse centroids [nlist, d] into an untrained IVFFlat index"""
        c = _f32(centroids)
        self._ck(self.lib.ref_ivf_set_centroids(ctypes.c_void_p(self.h), _p(c)))

    def set_train_niter(self, niter_coarse, niter_pq=0):
        self._ck(self.lib.ref_ivf_set_train_niter(ctypes.c_void_p(self.h), ctypes.c_int(niter_coarse),
                                                  ctypes.c_int(niter_pq)))

    @property
    def nlist(self):
        return self.lib.ref_ivf_nlist(ctypes.c_void_p(self.h))

    @property
    def code_size(self):
        return self.lib.ref_ivf_code_size(ctypes.c_void_p(self.h))

    def centroids(self):
        out = np.empty((self.nlist, self.d), dtype=np.float32)
        self._ck(self.lib.ref_ivf_get_centroids(ctypes.c_void_p(self.h), _p(out)))
        return out

    def lists(self):
        sizes = np.empty(self.nlist, dtype=np.uint32)
        self._ck(self.lib.ref_ivf_list_sizes(ctypes.c_void_p(self.h), _p(sizes)))
        n = int(sizes.sum())
        codes = np.empty((n, self.code_size), dtype=np.uint8)
        ids = np.empty(n, dtype=np.int64)
        self._ck(self.lib.ref_ivf_get_lists(ctypes.c_void_p(self.h), _p(codes), _p(ids)))
        return sizes, codes, ids

    def add_list_entries(self, list_no, ids, codes):
        """append entries (ids int64 [n], codes uint8 [n][code_size] / float32 rows for IVFFlat) to one inverted list"""
        ids = np.ascontiguousarray(ids, dtype=np.int64)
        codes = np.ascontiguousarray(codes)
        assert codes.nbytes == len(ids) * self.code_size, (codes.nbytes, len(ids), self.code_size)
        self._ck(self.lib.ref_ivf_add_list_entries(ctypes.c_void_p(self.h), ctypes.c_int64(int(list_no)),
                                                   ctypes.c_int64(len(ids)), _p(ids), _p(codes)))

    def pq_info(self):
        v = [ctypes.c_int(0) for _ in range(4)]
        self._ck(self.lib.ref_ivfpq_info(ctypes.c_void_p(self.h), *[ctypes.byref(x) for x in v]))
        return dict(M=v[0].value, dsub=v[1].value, nbits=v[2].value, use_precomputed_table=v[3].value)

    def pq_centroids(self):
        info = self.pq_info()
        out = np.empty((info["M"], 1 << info["nbits"], info["dsub"]), dtype=np.float32)
        self._ck(self.lib.ref_ivfpq_get_pq_centroids(ctypes.c_void_p(self.h), _p(out)))
        return out

    def sq_info(self):
        qt, br = ctypes.c_int(0), ctypes.c_int(0)
        cs, ts = ctypes.c_size_t(0), ctypes.c_size_t(0)
        self._ck(self.lib.ref_ivfsq_info(ctypes.c_void_p(self.h), ctypes.byref(qt), ctypes.byref(br), ctypes.byref(cs),
                                         ctypes.byref(ts)))
        return dict(qtype=qt.value, by_residual=bool(br.value), code_size=cs.value, trained_size=ts.value)

    def sq_trained(self):
        out = np.empty(self.sq_info()["trained_size"], dtype=np.float32)
        if out.size:
            self._ck(self.lib.ref_ivfsq_get_trained(ctypes.c_void_p(self.h), _p(out)))
        return out

    def set_sq_trained(self, centroids, trained):
        """install coarse centroids [nlist, d] and ScalarQuantizer::trained into an untrained IndexIVFScalarQuantizer"""
        c = _f32(centroids)
        t = np.ascontiguousarray(trained, dtype=np.float32).reshape(-1)
        self._ck(self.lib.ref_ivfsq_set_trained(ctypes.c_void_p(self.h), _p(c), _p(t), ctypes.c_size_t(t.size)))

    def sq_decode(self, codes):
        codes = np.ascontiguousarray(codes, dtype=np.uint8)
        out = np.empty((codes.shape[0], self.d), dtype=np.float32)
        self._ck(self.lib.ref_ivfsq_decode(ctypes.c_void_p(self.h), ctypes.c_int64(codes.shape[0]), _p(codes), _p(out)))
        return out

    def set_precomputed_table(self, use):
        self._ck(self.lib.ref_ivfpq_set_precomputed_table(ctypes.c_void_p(self.h), ctypes.c_int(use)))


class Ref:
    _lib = None

    @classmethod
    def available(cls):
        return os.path.exists(REF_SO)

    @classmethod
    def lib(cls):
        if cls._lib is None:
            # MKL must use the GNU OpenMP runtime the reference objects were compiled against
            os.environ.setdefault("MKL_THREADING_LAYER", "GNU")
            lib = ctypes.CDLL(REF_SO)
            lib.ref_index_factory.restype = ctypes.c_void_p
            lib.ref_index_factory.argtypes = [ctypes.c_int, ctypes.c_char_p, ctypes.c_int]
            lib.ref_last_error.restype = ctypes.c_char_p
            lib.ref_index_ntotal.restype = ctypes.c_int64
            lib.ref_shards_new.restype = ctypes.c_void_p
            cls._lib = lib
            # OpenMP sizes its pool by the visible CPUs; under a cgroup quota (GPU boxes: 256 visible, 16 allowed) that
            # many threads are throttled to a crawl -- use what the process may really run on
            lib.ref_set_omp_threads(ctypes.c_int(effective_cores()))
        return cls._lib

    @classmethod
    def version(cls):
        a, b, c = ctypes.c_int(0), ctypes.c_int(0), ctypes.c_int(0)
        cls.lib().ref_version(ctypes.byref(a), ctypes.byref(b), ctypes.byref(c))
        return (a.value, b.value, c.value)

    @classmethod
    def set_threads(cls, n):
        cls.lib().ref_set_omp_threads(ctypes.c_int(n))

    @classmethod
    def max_threads(cls):
        return cls.lib().ref_get_max_threads()

    @classmethod
    def index_factory(cls, d, desc, metric=METRIC_L2):
        h = cls.lib().ref_index_factory(d, desc.encode(), metric)
        if not h:
            raise RuntimeError("reference error: " + cls.lib().ref_last_error().decode())
        return RefIndex(cls.lib(), h, d)

    @classmethod
    def ivfsq(cls, d, nlist, qtype, metric=METRIC_L2, by_residual=True):
        """faiss::IndexIVFScalarQuantizer(new IndexFlat(d, metric), d, nlist, qtype, metric, by_residual)"""
        cls.lib().ref_ivfsq_new.restype = ctypes.c_void_p
        h = cls.lib().ref_ivfsq_new(ctypes.c_int(d), ctypes.c_int(nlist), ctypes.c_int(qtype), ctypes.c_int(metric),
                                    ctypes.c_int(int(by_residual)))
        if not h:
            raise RuntimeError("reference error: " + cls.lib().ref_last_error().decode())
        return RefIndex(cls.lib(), h, d)

    @classmethod
    def kmeans(cls, x, k, niter=25, seed=1234):
        x = _f32(x)
        cent = np.empty((k, x.shape[1]), dtype=np.float32)
        obj = ctypes.c_float(0)
        rc = cls.lib().ref_kmeans(ctypes.c_int(x.shape[1]), ctypes.c_int64(x.shape[0]), ctypes.c_int(k), _p(x),
                                  ctypes.c_int(niter), ctypes.c_int(seed), _p(cent), ctypes.byref(obj))
        if rc != 0:
            raise RuntimeError("reference error: " + cls.lib().ref_last_error().decode())
        return cent, obj.value

    # ---- drop-in proof: the reference's own callers running on a faiss_amd handle
    @classmethod
    def _wrap_ptr(cls, h, d, keep=None):
        if not h:
            raise RuntimeError("reference error: " + cls.lib().ref_last_error().decode())
        r = RefIndex(cls.lib(), h, d)
        r._keep = keep
        return r

    @classmethod
    def adapter(cls, amd_index):
        """Wrap a faiss_amd.Index handle into the bridge's faiss::Index subclass living in the reference library
        (integration/faiss_amd_bridge.h AmdIndex / AmdIndexIVF; the handle stays owned by `amd_index`)."""
        cls.lib().ref_amd_wrap.restype = ctypes.c_void_p
        return cls._wrap_ptr(cls.lib().ref_amd_wrap(amd_index._h), amd_index.d, amd_index)

    @classmethod
    def amd_resources(cls, device=0):
        """faiss::amd::AmdGpuResources of the bridge (an opaque pointer; freed with amd_resources_free)"""
        cls.lib().ref_amd_resources_new.restype = ctypes.c_void_p
        r = cls.lib().ref_amd_resources_new(ctypes.c_int(device))
        if not r:
            raise RuntimeError("reference error: " + cls.lib().ref_last_error().decode())
        return r

    @classmethod
    def amd_resources_free(cls, r):
        cls.lib().ref_amd_resources_free(ctypes.c_void_p(r))

    @classmethod
    def index_cpu_to_gpu(cls, res, cpu_index):
        """faiss::amd::index_cpu_to_gpu(res, index) of the bridge -> RefIndex over the new backend index"""
        cls.lib().ref_amd_index_cpu_to_gpu.restype = ctypes.c_void_p
        return cls._wrap_ptr(cls.lib().ref_amd_index_cpu_to_gpu(ctypes.c_void_p(res), ctypes.c_void_p(cpu_index.h)),
                             cpu_index.d, [cpu_index])

    @classmethod
    def index_cpu_to_gpu_multiple(cls, res_list, cpu_index, shard=False, shard_type=1, common_ivf_quantizer=False):
        cls.lib().ref_amd_index_cpu_to_gpu_multiple.restype = ctypes.c_void_p
        arr = (ctypes.c_void_p * len(res_list))(*res_list)
        h = cls.lib().ref_amd_index_cpu_to_gpu_multiple(arr, ctypes.c_int(len(res_list)), ctypes.c_void_p(cpu_index.h),
                                                        ctypes.c_int(int(shard)), ctypes.c_int(shard_type),
                                                        ctypes.c_int(int(common_ivf_quantizer)))
        return cls._wrap_ptr(h, cpu_index.d, [cpu_index])

    @classmethod
    def index_gpu_to_cpu(cls, gpu_index):
        cls.lib().ref_amd_index_gpu_to_cpu.restype = ctypes.c_void_p
        return cls._wrap_ptr(cls.lib().ref_amd_index_gpu_to_cpu(ctypes.c_void_p(gpu_index.h)), gpu_index.d, [gpu_index])

    @classmethod
    def sq_train(cls, qtype, rangestat, rangestat_arg, x):
        """faiss::ScalarQuantizer(d, qtype).train(x) with the given RangeStat: the `trained` vector"""
        x = _f32(x)
        out = np.empty(2 * x.shape[1], dtype=np.float32)
        n = ctypes.c_size_t(0)
        rc = cls.lib().ref_sq_train(ctypes.c_int(x.shape[1]), ctypes.c_int(qtype), ctypes.c_int(rangestat), ctypes.c_float(rangestat_arg),
                                    ctypes.c_int64(x.shape[0]), _p(x), _p(out), ctypes.byref(n))
        if rc != 0:
            raise RuntimeError("reference error: " + cls.lib().ref_last_error().decode())
        return out[:n.value].copy()

    @classmethod
    def amd_flat(cls, res, d, metric=METRIC_L2):
        """a flat index of the backend as a faiss::Index of the bridge (AmdIndexFlat)"""
        cls.lib().ref_amd_flat_new.restype = ctypes.c_void_p
        return cls._wrap_ptr(cls.lib().ref_amd_flat_new(ctypes.c_void_p(res), ctypes.c_int(d), ctypes.c_int(metric)), d)

    @classmethod
    def amd_ivf_with_quantizer(cls, res, quantizer, kind, d, nlist, arg=0, metric=METRIC_L2, coarse_f16=False, indices_options=3):
        """the bridge's GpuIndexIVFFlat (kind 0) / IVFPQ (1, arg = M) / IVFScalarQuantizer (2, arg = qtype) over the CALLER's coarse
        quantizer `quantizer` (any RefIndex: Ref.amd_flat(...) runs on the device, a CPU index on the host)"""
        cls.lib().ref_amd_ivf_new_with_quantizer.restype = ctypes.c_void_p
        h = cls.lib().ref_amd_ivf_new_with_quantizer(ctypes.c_void_p(res), ctypes.c_void_p(quantizer.h), ctypes.c_int(kind),
                                                     ctypes.c_int(d), ctypes.c_int(nlist), ctypes.c_int(arg), ctypes.c_int(metric),
                                                     ctypes.c_int(int(coarse_f16)), ctypes.c_int(indices_options))
        return cls._wrap_ptr(h, d, [quantizer])

    @classmethod
    def amd_autotune(cls, index, xq, k, gt):
        """the bridge's GpuParameterSpace::initialize + the reference's ParameterSpace::explore on `index` (1-recall@1 against gt [nq]):
        (number of parameter ranges, [(perf, seconds, key)] of the optimal operating points)"""
        xq = _f32(xq)
        gt = np.ascontiguousarray(gt, dtype=np.int64).reshape(-1)
        cap = 64
        perf, t = np.zeros(cap), np.zeros(cap)
        nr, npts = ctypes.c_int(0), ctypes.c_int(0)
        buf = ctypes.create_string_buffer(4096)
        rc = cls.lib().ref_amd_autotune(ctypes.c_void_p(index.h), ctypes.c_int64(xq.shape[0]), _p(xq), ctypes.c_int64(k), _p(gt),
                                        ctypes.c_int(cap), ctypes.byref(nr), ctypes.byref(npts), _p(perf), _p(t), buf, ctypes.c_int(4096))
        if rc != 0:
            raise RuntimeError("reference error: " + cls.lib().ref_last_error().decode())
        keys = [x for x in buf.value.decode().split(";")][:npts.value]
        return nr.value, [(float(perf[i]), float(t[i]), keys[i]) for i in range(npts.value)]

    @classmethod
    def write_index(cls, index, path):
        if cls.lib().ref_write_index(ctypes.c_void_p(index.h), path.encode()) != 0:
            raise RuntimeError("reference error: " + cls.lib().ref_last_error().decode())

    @classmethod
    def read_index(cls, path, d):
        cls.lib().ref_read_index.restype = ctypes.c_void_p
        return cls._wrap_ptr(cls.lib().ref_read_index(path.encode()), d)

    @classmethod
    def ivfflat_with_quantizer(cls, quantizer, d, nlist, metric=METRIC_L2):
        """a reference faiss::IndexIVFFlat whose coarse quantizer is `quantizer` (any RefIndex, e.g. Ref.adapter(...))"""
        cls.lib().ref_ivfflat_with_quantizer.restype = ctypes.c_void_p
        h = cls.lib().ref_ivfflat_with_quantizer(ctypes.c_void_p(quantizer.h), ctypes.c_int(d), ctypes.c_int(nlist),
                                                 ctypes.c_int(metric))
        return cls._wrap_ptr(h, d, [quantizer])

    @classmethod
    def kmeans_with_index(cls, x, k, ref_index, niter=25, seed=1234):
        x = _f32(x)
        cent = np.empty((k, x.shape[1]), dtype=np.float32)
        obj = ctypes.c_float(0)
        rc = cls.lib().ref_kmeans_with_index(ctypes.c_int(x.shape[1]), ctypes.c_int64(x.shape[0]), ctypes.c_int(k),
                                             _p(x), ctypes.c_int(niter), ctypes.c_int(seed),
                                             ctypes.c_void_p(ref_index.h), _p(cent), ctypes.byref(obj))
        if rc != 0:
            raise RuntimeError("reference error: " + cls.lib().ref_last_error().decode())
        return cent, obj.value

    @classmethod
    def shards(cls, d, subs, threaded=True, successive_ids=True):
        h = cls.lib().ref_shards_new(ctypes.c_int(d), ctypes.c_int(int(threaded)), ctypes.c_int(int(successive_ids)))
        r = RefIndex(cls.lib(), h, d)
        r._keep = list(subs)
        for s in subs:
            rc = cls.lib().ref_shards_add(ctypes.c_void_p(h), ctypes.c_void_p(s.h))
            if rc != 0:
                raise RuntimeError("reference error: " + cls.lib().ref_last_error().decode())
        return r


# synthetic inputs live in faiss_amd/datasets.py (no oracle code there); re-exported for the tests
from faiss_amd.datasets import integer_dataset, synthetic_dataset  # noqa: E402,F401
