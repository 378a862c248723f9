"""TEST INFRASTRUCTURE ONLY (oracle/): ctypes wrapper around oracle/libivf_oracle.so, the CPU
restatement of the reference hot path.  Importable from tests/, __graft_entry__.smoke() and
bench.py's cpu_baseline leg only -- never from the product package."""
import ctypes as C
import os
import subprocess

import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))
_LIB = None

METRIC_IP, METRIC_L2 = 0, 1

c_f32p = C.POINTER(C.c_float)
c_i64p = C.POINTER(C.c_int64)
c_szp = C.POINTER(C.c_size_t)


class OrcIndex(C.Structure):
    _fields_ = [("metric", C.c_int), ("d", C.c_size_t), ("nlist", C.c_size_t), ("list_off", c_szp),
                ("codes", c_f32p), ("ids", c_i64p)]


class OrcTuner(C.Structure):
    _fields_ = [("max_topk", C.c_size_t), ("query_topk", C.c_size_t), ("ntraces", C.c_size_t),
                ("multipler", C.c_float), ("std_m", C.c_float), ("interdis_cem", c_f32p), ("arcos_list", c_f32p),
                ("trace_off", c_szp), ("trace_x", c_f32p), ("trace_y", c_f32p), ("trace_std", c_f32p),
                ("require_acc", c_f32p), ("gt_D", c_f32p), ("my_nprobe", c_szp), ("t_recalls", c_f32p),
                ("profile", C.c_int), ("overhead_profile", C.c_int)]


def build():
    subprocess.run(["make", "-C", HERE, "oracle"], check=True, stdout=subprocess.DEVNULL)


def lib():
    global _LIB
    if _LIB is None:
        path = os.path.join(HERE, "libivf_oracle.so")
        if not os.path.exists(path):
            build()
        L = C.CDLL(path)
        L.orc_fvec_L2sqr.restype = C.c_float
        L.orc_fvec_inner_product.restype = C.c_float
        L.orc_scan_codes.restype = C.c_size_t
        L.orc_trace_sb.restype = C.c_size_t
        L.orc_last_error.restype = C.c_char_p
        _LIB = L
    return _LIB


def _f(a):
    return a.ctypes.data_as(c_f32p)


def _i(a):
    return a.ctypes.data_as(c_i64p)


def _s(a):
    return a.ctypes.data_as(c_szp)


def f32(a):
    return np.ascontiguousarray(a, dtype=np.float32)


def i64(a):
    return np.ascontiguousarray(a, dtype=np.int64)


class Lists:
    """CSR-packed inverted lists built the way IndexIVFFlat::add_core appends
    (IndexIVFFlat.cpp:41-80): database order within each list."""

    def __init__(self, metric, centroids, xb, assign, ids=None):
        self.metric = metric
        self.centroids = f32(centroids)
        self.nlist, self.d = self.centroids.shape
        assign = np.asarray(assign, dtype=np.int64)
        keep = np.nonzero(assign >= 0)[0]
        order = keep[np.argsort(assign[keep], kind="stable")]
        sizes = np.bincount(assign[keep], minlength=self.nlist)
        self.off = np.zeros(self.nlist + 1, dtype=np.uintp)
        self.off[1:] = np.cumsum(sizes)
        self.codes = f32(xb[order])
        gid = np.arange(xb.shape[0], dtype=np.int64) if ids is None else np.asarray(ids, dtype=np.int64)
        self.ids = i64(gid[order])
        self.sizes = sizes.astype(np.int64)
        self.struct = OrcIndex(metric, self.d, self.nlist, _s(self.off), _f(self.codes), _i(self.ids))


class Tuner:
    def __init__(self, interdis, traces, max_topk, nq_alloc, arcos=None):
        """traces: list of (x, y, std) arrays"""
        self.interdis = f32(interdis)
        self.arcos = f32(arcos) if arcos is not None else arcos_table()
        self.off = np.zeros(len(traces) + 1, dtype=np.uintp)
        self.off[1:] = np.cumsum([len(t[0]) for t in traces])
        self.tx = f32(np.concatenate([t[0] for t in traces]))
        self.ty = f32(np.concatenate([t[1] for t in traces]))
        self.ts = f32(np.concatenate([t[2] for t in traces]))
        self.max_topk = max_topk
        self.ntraces = len(traces)
        self.my_nprobe = np.zeros(nq_alloc, dtype=np.uintp)
        self.t_recalls = np.zeros(nq_alloc, dtype=np.float32)

    def struct(self, query_topk, require_acc, multipler, std_m, gt_D=None, profile=False, overhead_profile=False):
        self.req = f32(require_acc)
        self.gt = f32(gt_D) if gt_D is not None else None
        return OrcTuner(self.max_topk, query_topk, self.ntraces, multipler, std_m, _f(self.interdis),
                        _f(self.arcos), _s(self.off), _f(self.tx), _f(self.ty), _f(self.ts), _f(self.req),
                        _f(self.gt) if self.gt is not None else None, _s(self.my_nprobe), _f(self.t_recalls),
                        int(profile), int(overhead_profile))


def arcos_table():
    out = np.zeros(500, dtype=np.float32)
    lib().orc_arcos_table(_f(out))
    return out


def knn(metric, x, y, k, gemm=False, nthreads=1):
    x, y = f32(x), f32(y)
    nx, d = x.shape
    D = np.empty((nx, k), dtype=np.float32)
    I = np.empty((nx, k), dtype=np.int64)
    lib().orc_knn(metric, _f(x), _f(y), C.c_size_t(d), C.c_size_t(nx), C.c_size_t(y.shape[0]), C.c_size_t(k), _f(D),
                  _i(I), int(gemm), nthreads)
    return D, I


def interdis(metric, centroids):
    c = f32(centroids).copy()
    nlist, d = c.shape
    out = np.empty(nlist * (nlist - 1) // 2, dtype=np.float32)
    lib().orc_interdis(metric, _f(c), C.c_size_t(nlist), C.c_size_t(d), _f(out))
    return out


def set_online(metric, nlist, cd, ci, interdis_cem, arcos):
    m = nlist // 8 + 20
    dtb = np.zeros(m, dtype=np.float32)
    c2c = np.zeros(m, dtype=np.float32)
    cd, ci, interdis_cem, arcos = f32(cd), i64(ci), f32(interdis_cem), f32(arcos)
    rc = lib().orc_set_online(metric, C.c_size_t(nlist), _f(cd), _i(ci), _f(interdis_cem), _f(arcos), _f(dtb), _f(c2c))
    if rc:
        raise RuntimeError(lib().orc_last_error().decode())
    return dtb, c2c


def scan_codes(metric, query, codes, ids, list_no, store_pairs, simi, idxi):
    query, codes, ids = f32(query), f32(codes), i64(ids)
    k = simi.shape[0]
    return lib().orc_scan_codes(metric, C.c_size_t(codes.shape[1]), _f(query), C.c_size_t(codes.shape[0]), _f(codes),
                                _i(ids), C.c_int64(list_no), int(store_pairs), C.c_size_t(k), _f(simi), _i(idxi))


def search_preassigned(lists, x, k, keys, coarse_dis, store_pairs=False, max_codes=0, tuner=None, offset=0,
                       nthreads=1):
    x, keys, coarse_dis = f32(x), i64(keys), f32(coarse_dis)
    n, nprobe = keys.shape
    D = np.empty((n, k), dtype=np.float32)
    I = np.empty((n, k), dtype=np.int64)
    stats = np.zeros(3, dtype=np.uintp)
    rc = lib().orc_search_preassigned(C.byref(lists.struct), C.c_size_t(n), _f(x), C.c_size_t(k), C.c_size_t(nprobe),
                                      _i(keys), _f(coarse_dis), _f(D), _i(I), int(store_pairs), C.c_size_t(max_codes),
                                      C.byref(tuner) if tuner is not None else None, C.c_size_t(offset), _s(stats),
                                      nthreads)
    if rc:
        raise RuntimeError(lib().orc_last_error().decode())
    return D, I, stats.astype(np.int64)


def kmeans(metric, x, k, niter=25, seed=1234, max_points_per_centroid=256, spherical=False, int_centroids=False, gemm=False,
           nthreads=8):
    """Clustering::train restated -> centroids (k, d), objective per iteration"""
    x = f32(x)
    n, d = x.shape
    cen = np.zeros((k, d), dtype=np.float32)
    obj = np.zeros(niter, dtype=np.float32)
    L = lib()
    L.orc_kmeans.restype = None
    L.orc_kmeans(int(metric), C.c_size_t(d), C.c_size_t(n), _f(x), C.c_size_t(k), int(niter), C.c_long(seed),
                 C.c_size_t(max_points_per_centroid), int(spherical), int(int_centroids), int(gemm), _f(cen), _f(obj), int(nthreads))
    return cen, obj


def range_search_preassigned(lists, x, radius, keys):
    """-> lims (n + 1), labels, distances, stats {nlist, ndis}"""
    x, keys = f32(x), i64(keys)
    n, nprobe = keys.shape
    lims = np.zeros(n + 1, dtype=np.uintp)
    stats = np.zeros(2, dtype=np.uintp)
    L = lib()
    L.orc_range_search_preassigned.restype = C.c_int
    args = (C.byref(lists.struct), C.c_size_t(n), _f(x), C.c_float(radius), C.c_size_t(nprobe), _i(keys), _s(lims))
    if L.orc_range_search_preassigned(*args, None, None, _s(stats)):
        raise RuntimeError(L.orc_last_error().decode())
    tot = int(lims[n])
    labels = np.empty(max(tot, 1), dtype=np.int64)
    dist = np.empty(max(tot, 1), dtype=np.float32)
    if L.orc_range_search_preassigned(*args, _i(labels), _f(dist), _s(stats)):
        raise RuntimeError(L.orc_last_error().decode())
    return lims.astype(np.int64), labels[:tot], dist[:tot], stats.astype(np.int64)


def train_samples(lists, x, max_topk, keys, coarse_dis, interdis_cem, arcos, gt_D, offset, train_num, raw):
    """raw: list of (train_num*(max_topk//4), 2) float32 arrays pre-filled with -1, updated in place"""
    x, keys, coarse_dis = f32(x), i64(keys), f32(coarse_dis)
    interdis_cem, arcos, gt_D = f32(interdis_cem), f32(arcos), f32(gt_D)
    n, nprobe = keys.shape
    D = np.empty((n, max_topk), dtype=np.float32)
    I = np.empty((n, max_topk), dtype=np.int64)
    ptrs = (c_f32p * len(raw))(*[_f(r) for r in raw])
    rc = lib().orc_train_samples(C.byref(lists.struct), C.c_size_t(n), _f(x), C.c_size_t(max_topk), C.c_size_t(nprobe),
                                 _i(keys), _f(coarse_dis), _f(interdis_cem), _f(arcos), _f(gt_D), C.c_size_t(offset),
                                 C.c_size_t(train_num), ptrs, _f(D), _i(I))
    if rc:
        raise RuntimeError(lib().orc_last_error().decode())
    return D, I


def trace_sb(raw_xy, bs=250):
    raw = f32(raw_xy).copy()
    n = raw.shape[0]
    cap = n // bs + 2
    ox, oy, os_ = (np.zeros(cap, dtype=np.float32) for _ in range(3))
    sz = lib().orc_trace_sb(_f(raw), C.c_size_t(n), C.c_size_t(bs), _f(ox), _f(oy), _f(os_))
    return ox[:sz].copy(), oy[:sz].copy(), os_[:sz].copy()


def merge_tables(metric, all_D, all_I):
    all_D, all_I = f32(all_D), i64(all_I)
    nshard, n, k = all_D.shape
    D = np.empty((n, k), dtype=np.float32)
    I = np.empty((n, k), dtype=np.int64)
    lib().orc_merge_tables(metric, C.c_size_t(n), C.c_size_t(k), C.c_size_t(nshard), _f(all_D), _i(all_I), _f(D), _i(I))
    return D, I
