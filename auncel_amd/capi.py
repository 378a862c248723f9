"""ctypes binding of include/auncel_amd.h (the C-ABI drop-in boundary).

Fails loudly when libauncel_amd.so is missing: there is no CPU path in this package."""
import ctypes as C
import os

import numpy as np

from . import build as _build

METRIC_IP, METRIC_L2 = 0, 1

_f32p = C.POINTER(C.c_float)
_i64p = C.POINTER(C.c_int64)
_u64p = C.POINTER(C.c_uint64)
_szp = C.POINTER(C.c_size_t)
_LIB = None

SYMBOLS = [
    "amd_ivf_last_error", "amd_ivf_device_count", "amd_ivf_create", "amd_ivf_clone", "amd_ivf_destroy", "amd_ivf_set_centroids",
    "amd_ivf_set_lists", "amd_ivf_add", "amd_ivf_ntotal", "amd_ivf_list_size", "amd_ivf_get_list", "amd_ivf_coarse",
    "amd_ivf_search_preassigned", "amd_ivf_search", "amd_ivf_scan_codes", "amd_ivf_scan_codes_at", "amd_ivf_scan_codes_range",
    "amd_ivf_scan_codes_range_results", "amd_ivf_distance_to_code", "amd_ivf_stats",
    "amd_ivf_set_queries", "amd_ivf_search_resident", "amd_ivf_search_resident_preassigned", "amd_ivf_coarse_resident", "amd_ivf_set_interdis", "amd_ivf_get_interdis",
    "amd_ivf_set_tuner", "amd_ivf_search_adaptive", "amd_ivf_search_adaptive_x", "amd_ivf_search_adaptive_pre", "amd_ivf_search_timed", "amd_ivf_search_timed_x",
    "amd_ivf_train_samples",
    "amd_ivf_train_samples_x", "amd_ivf_train_samples_pre", "amd_ivf_trace_sb", "amd_ivf_arcos_table", "amd_ivf_merge_tables",
    "amd_ivf_last_timing", "amd_ivf_last_timing_detail", "amd_ivf_last_scan_min_bytes", "amd_ivf_coarse_tie_rows", "amd_ivf_last_tie_fixed", "amd_ivf_last_filter", "amd_ivf_last_direct_out", "amd_ivf_last_coarse_pick", "amd_ivf_set_async_depth", "amd_ivf_submit_adaptive", "amd_ivf_submit_search_resident", "amd_ivf_submit_coarse_resident",
    "amd_ivf_submit_search_resident_preassigned", "amd_ivf_wait", "amd_ivf_async_counts", "amd_ivf_last_tie_redone", "amd_ivf_last_tie_patched", "amd_ivf_self_check", "amd_ivf_last_round_hints", "amd_ivf_set_byte_codes", "amd_ivf_set_option", "amd_ivf_get_option",
    "amd_ivf_kmeans",
    "amd_ivf_range_search_preassigned", "amd_ivf_range_search", "amd_ivf_range_results",
    "amd_ivf_scan_arith",
    "amd_ivf_read_fvecs", "amd_ivf_read_ivecs", "amd_ivf_read_fbin", "amd_ivf_read_ibin", "amd_ivf_free",
]


class EngineError(RuntimeError):
    def __init__(self, code, msg):
        super().__init__(f"auncel_amd error {code}: {msg}")
        self.code = code


def _share_torch_hip_runtime():
    """PyTorch-ROCm wheels bundle their own libamdhip64 / libhsa-runtime64.  A process that loads torch first runs this library on
    that runtime too (its hip* symbols bind to the copy already in the global scope); a process that creates an index first gets
    ROCm's copy for the library and torch's for torch -- and the second runtime to start finds no device ("No HIP GPUs are
    available").  So where torch is installed but not yet imported, its runtime is loaded first, globally, and both orders behave
    like the first.  (C and C++ callers link one runtime and have no such choice to make.)"""
    import importlib.util
    import sys
    if "torch" in sys.modules or os.environ.get("AUNCEL_AMD_OWN_HIP_RUNTIME"):
        return
    try:
        spec = importlib.util.find_spec("torch")
    except (ImportError, ValueError):
        return
    if spec is None or not spec.origin:
        return
    d = os.path.join(os.path.dirname(spec.origin), "lib")
    names = ("libhsa-runtime64.so", "libamdhip64.so")
    if not all(os.path.exists(os.path.join(d, n)) for n in names):
        return  # (a torch build without a bundled runtime: it uses ROCm's, like the engine)
    for name in names:
        try:
            _SHARED[name] = C.CDLL(os.path.join(d, name), mode=C.RTLD_GLOBAL)
        except OSError as e:
            # half a runtime (torch's HSA under ROCm's HIP) is worse than either: say so instead of going on
            raise ImportError(f"auncel_amd: torch's bundled {name} could not be loaded next to {sorted(_SHARED)} ({e}); set "
                              "AUNCEL_AMD_OWN_HIP_RUNTIME=1 to run the engine on ROCm's own runtime (and do not use torch's GPU "
                              "side in the same process)") from e
    if os.environ.get("AUNCEL_AMD_VERBOSE"):
        print(f"[auncel_amd] running on torch's bundled HIP runtime ({d}); AUNCEL_AMD_OWN_HIP_RUNTIME=1 keeps ROCm's", file=sys.stderr)


_SHARED = {}


def hip_runtime():
    """a ctypes handle on the HIP runtime this process uses (hipHostMalloc for page-locked result buffers, ...): torch's bundled
    copy where torch is loaded or was pre-loaded by lib(), else ROCm's -- never a second runtime beside the first"""
    import sys
    lib()
    if "libamdhip64.so" in _SHARED:
        return _SHARED["libamdhip64.so"]
    if "torch" in sys.modules:
        p = os.path.join(os.path.dirname(sys.modules["torch"].__file__), "lib", "libamdhip64.so")
        if os.path.exists(p):
            return C.CDLL(p)
    return C.CDLL("libamdhip64.so")


def lib():
    global _LIB
    if _LIB is None:
        if not os.path.exists(_build.LIB):
            raise ImportError(f"{_build.LIB} is missing: run `python -c 'import __graft_entry__ as g; g.build()'` "
                              "(hipcc --offload-arch=gfx950).  auncel_amd has no CPU fallback.")
        _share_torch_hip_runtime()
        L = C.CDLL(os.environ.get("AUNCEL_AMD_LIB", _build.LIB))  # AUNCEL_AMD_LIB: a differently built engine (kernel experiments)
        L.amd_ivf_last_error.restype = C.c_char_p
        for s in SYMBOLS[1:]:
            getattr(L, s).restype = C.c_int
        L.amd_ivf_free.restype = None
        _LIB = L
    return _LIB


def _chk(rc):
    if rc != 0:
        raise EngineError(rc, lib().amd_ivf_last_error().decode())


def _f(a):
    return a.ctypes.data_as(_f32p) if a is not None else None


def _i(a):
    return a.ctypes.data_as(_i64p) if a is not None else None


def f32(a):
    return np.ascontiguousarray(a, dtype=np.float32)


def i64(a):
    return np.ascontiguousarray(a, dtype=np.int64)


def device_count():
    n = C.c_int(0)
    _chk(lib().amd_ivf_device_count(C.byref(n)))
    return n.value


def self_check(device=0):
    """include/auncel_amd.h: amd_ivf_self_check -> dict(adds, counted, wrong, xcd_mask)"""
    v = (C.c_uint64 * 4)()
    _chk(lib().amd_ivf_self_check(int(device), v))
    return {"adds": int(v[0]), "counted": int(v[1]), "wrong": int(v[2]), "xcd_mask": int(v[3])}


def kmeans(metric, x, k, niter=25, seed=1234, max_points_per_centroid=256, spherical=False, int_centroids=False, coarse_mode=0,
           device=0):
    """Clustering::train on the GPU (include/auncel_amd.h: amd_ivf_kmeans) -> centroids (k, d), objective per iteration"""
    x = f32(x)
    n, d = x.shape
    cen = np.zeros((k, d), np.float32)
    obj = np.zeros(niter, np.float32)
    _chk(lib().amd_ivf_kmeans(int(d), C.c_size_t(n), _f(x), C.c_size_t(k), int(metric), int(niter), C.c_long(seed),
                              C.c_size_t(max_points_per_centroid), int(spherical), int(int_centroids), int(coarse_mode), int(device),
                              _f(cen), _f(obj)))
    return cen, obj


def merge_tables(metric, all_D, all_I):
    all_D, all_I = f32(all_D), i64(all_I)
    nshard, n, k = all_D.shape
    D = np.empty((n, k), np.float32)
    I = np.empty((n, k), np.int64)
    _chk(lib().amd_ivf_merge_tables(metric, C.c_size_t(n), C.c_size_t(k), C.c_size_t(nshard), _f(all_D), _i(all_I), _f(D), _i(I)))
    return D, I


def trace_sb(raw_xy, bs=250):
    raw = f32(raw_xy)
    n = raw.shape[0]
    cap = n // bs + 2
    ox, oy, os_ = (np.zeros(cap, np.float32) for _ in range(3))
    nb = C.c_size_t(0)
    _chk(lib().amd_ivf_trace_sb(_f(raw), C.c_size_t(n), C.c_size_t(bs), _f(ox), _f(oy), _f(os_), C.byref(nb)))
    return ox[:nb.value].copy(), oy[:nb.value].copy(), os_[:nb.value].copy()


def _take(ptr, n, d, ctype, dtype):
    """copy a malloc'ed n x d matrix out of the library and release it"""
    try:
        out = np.ctypeslib.as_array(C.cast(ptr, C.POINTER(ctype)), shape=(n * d,)).astype(dtype, copy=True).reshape(n, d) if n * d else np.zeros((n, d), dtype)
    finally:
        lib().amd_ivf_free(ptr)
    return out


def read_fvecs(path):
    """fvecs_read (Auncel/eval/bound.cpp:29-58) -> float32 (n, d)"""
    d, n, p = C.c_size_t(0), C.c_size_t(0), _f32p()
    _chk(lib().amd_ivf_read_fvecs(os.fsencode(path), C.byref(d), C.byref(n), C.byref(p)))
    return _take(p, n.value, d.value, C.c_float, np.float32)


def read_ivecs(path):
    """ivecs_read (Auncel/eval/bound.cpp:61-63) -> int32 (n, d)"""
    d, n, p = C.c_size_t(0), C.c_size_t(0), C.POINTER(C.c_int32)()
    _chk(lib().amd_ivf_read_ivecs(os.fsencode(path), C.byref(d), C.byref(n), C.byref(p)))
    return _take(p, n.value, d.value, C.c_int32, np.int32)


def read_fbin(path, num=0, nbytes=4):
    """fbin_read (Auncel/eval/bound.cpp:65-109): `num` rows (0: the header's count); nbytes 1 = signed chars widened, as the
    harness reads its u8 SIFT files -> float32 (rows, d), header row count"""
    d, n, p = C.c_size_t(0), C.c_size_t(0), _f32p()
    _chk(lib().amd_ivf_read_fbin(os.fsencode(path), C.c_size_t(num), int(nbytes), C.byref(d), C.byref(n), C.byref(p)))
    return _take(p, num if num else n.value, d.value, C.c_float, np.float32), n.value


def read_ibin(path, num=0):
    """ibin_read (Auncel/eval/bound.cpp:111-113) -> int32 (rows, d), header row count"""
    d, n, p = C.c_size_t(0), C.c_size_t(0), C.POINTER(C.c_int32)()
    _chk(lib().amd_ivf_read_ibin(os.fsencode(path), C.c_size_t(num), C.byref(d), C.byref(n), C.byref(p)))
    return _take(p, num if num else n.value, d.value, C.c_int32, np.int32), n.value


def arcos_table():
    """error_pro::construct_arcos (IVF_pro.cpp:151-160): 500-entry acos LUT, fp32"""
    out = np.zeros(500, np.float32)
    _chk(lib().amd_ivf_arcos_table(_f(out)))
    return out


class Handle:
    """thin owner of an amd_ivf_t*"""

    def __init__(self, d, nlist, metric=METRIC_L2, device=0):
        self.d, self.nlist, self.metric, self.device = int(d), int(nlist), int(metric), int(device)
        self._h = C.c_void_p()
        _chk(lib().amd_ivf_create(self.d, C.c_size_t(self.nlist), self.metric, self.device, C.byref(self._h)))

    def clone(self):
        """a second search context over the same device-resident index (include/auncel_amd.h: amd_ivf_clone); the
        owner is kept alive by the clone"""
        c = Handle.__new__(Handle)
        c.d, c.nlist, c.metric, c.device = self.d, self.nlist, self.metric, self.device
        c._h = C.c_void_p()
        c._owner = self
        for attr in ("max_topk",):  # tuner geometry set on the owner by set_tuner
            if hasattr(self, attr):
                setattr(c, attr, getattr(self, attr))
        _chk(lib().amd_ivf_clone(self._h, C.byref(c._h)))
        return c

    def close(self):
        if self._h:
            lib().amd_ivf_destroy(self._h)
            self._h = C.c_void_p()

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass

    # ---- contents
    def set_centroids(self, c):
        c = f32(c)
        assert c.shape == (self.nlist, self.d)
        _chk(lib().amd_ivf_set_centroids(self._h, _f(c)))

    def set_lists(self, sizes, codes, ids):
        sizes = np.ascontiguousarray(sizes, dtype=np.uintp)
        codes = [f32(c) for c in codes]
        ids = [i64(i) for i in ids]
        cp = (_f32p * self.nlist)(*[_f(c) for c in codes])
        ip = (_i64p * self.nlist)(*[_i(i) for i in ids])
        _chk(lib().amd_ivf_set_lists(self._h, sizes.ctypes.data_as(_szp), cp, ip))

    def set_lists_from_assign(self, xb, assign, ids=None):
        """lists as IndexIVFFlat::add_core builds them: database order inside every list"""
        assign = np.asarray(assign, dtype=np.int64)
        gid = np.arange(len(assign), dtype=np.int64) if ids is None else np.asarray(ids, dtype=np.int64)
        keep = np.nonzero(assign >= 0)[0]
        order = keep[np.argsort(assign[keep], kind="stable")]
        sizes = np.bincount(assign[keep], minlength=self.nlist)
        off = np.concatenate([[0], np.cumsum(sizes)])
        xs, gs = f32(xb)[order], gid[order]
        self.set_lists(sizes, [xs[off[l]:off[l + 1]] for l in range(self.nlist)],
                       [gs[off[l]:off[l + 1]] for l in range(self.nlist)])

    def add(self, x, xids=None, precomputed_idx=None):
        x = f32(x)
        xi = i64(xids) if xids is not None else None
        pi = i64(precomputed_idx) if precomputed_idx is not None else None
        _chk(lib().amd_ivf_add(self._h, C.c_size_t(x.shape[0]), _f(x), _i(xi), _i(pi)))

    @property
    def ntotal(self):
        n = C.c_size_t(0)
        _chk(lib().amd_ivf_ntotal(self._h, C.byref(n)))
        return n.value

    def list_size(self, l):
        n = C.c_size_t(0)
        _chk(lib().amd_ivf_list_size(self._h, C.c_size_t(l), C.byref(n)))
        return n.value

    def get_list(self, l):
        n = self.list_size(l)
        codes = np.empty((n, self.d), np.float32)
        ids = np.empty(n, np.int64)
        _chk(lib().amd_ivf_get_list(self._h, C.c_size_t(l), _f(codes), _i(ids)))
        return codes, ids

    # ---- search
    def coarse(self, x, nprobe, mode=0):
        x = f32(x)
        n = x.shape[0]
        D = np.empty((n, nprobe), np.float32)
        I = np.empty((n, nprobe), np.int64)
        _chk(lib().amd_ivf_coarse(self._h, C.c_size_t(n), _f(x), C.c_size_t(nprobe), _f(D), _i(I), mode))
        return D, I

    def search_preassigned(self, x, k, keys, coarse_dis=None, store_pairs=False, max_codes=0):
        x, keys = f32(x), i64(keys)
        n, nprobe = keys.shape
        cd = f32(coarse_dis) if coarse_dis is not None else None
        D = np.empty((n, k), np.float32)
        I = np.empty((n, k), np.int64)
        _chk(lib().amd_ivf_search_preassigned(self._h, C.c_size_t(n), _f(x), C.c_size_t(k), C.c_size_t(nprobe), _i(keys),
                                              _f(cd), _f(D), _i(I), int(store_pairs), C.c_size_t(max_codes)))
        return D, I

    def search(self, x, k, nprobe, coarse_mode=0):
        x = f32(x)
        n = x.shape[0]
        D = np.empty((n, k), np.float32)
        I = np.empty((n, k), np.int64)
        _chk(lib().amd_ivf_search(self._h, C.c_size_t(n), _f(x), C.c_size_t(k), C.c_size_t(nprobe), coarse_mode, _f(D), _i(I)))
        return D, I

    def range_search(self, x, radius, nprobe, keys=None, coarse_mode=0):
        """IndexIVF::range_search(_preassigned) -> lims (n + 1), labels, distances"""
        x = f32(x)
        n = x.shape[0]
        lims = np.zeros(n + 1, dtype=np.uintp)
        lp = lims.ctypes.data_as(_szp)
        if keys is None:
            _chk(lib().amd_ivf_range_search(self._h, C.c_size_t(n), _f(x), C.c_float(radius), C.c_size_t(nprobe), coarse_mode, lp))
        else:
            keys = i64(keys)
            assert keys.shape == (n, nprobe)
            _chk(lib().amd_ivf_range_search_preassigned(self._h, C.c_size_t(n), _f(x), C.c_float(radius), C.c_size_t(nprobe), _i(keys), lp))
        tot = int(lims[n])
        labels = np.empty(max(tot, 1), np.int64)
        dist = np.empty(max(tot, 1), np.float32)
        _chk(lib().amd_ivf_range_results(self._h, _i(labels), _f(dist)))
        return lims.astype(np.int64), labels[:tot], dist[:tot]

    def scan_codes(self, query, list_no, simi, idxi, store_pairs=False):
        query = f32(query)
        nup = C.c_size_t(0)
        _chk(lib().amd_ivf_scan_codes(self._h, _f(query), C.c_size_t(list_no), int(store_pairs), C.c_size_t(simi.shape[0]),
                                      _f(simi), _i(idxi), C.byref(nup)))
        return nup.value

    def scan_codes_at(self, query, list_no, offset, n, simi, idxi, store_pairs=False):
        """InvertedListScanner::scan_codes over vectors [offset, offset + n) of the list"""
        query = f32(query)
        nup = C.c_size_t(0)
        _chk(lib().amd_ivf_scan_codes_at(self._h, _f(query), C.c_size_t(list_no), C.c_size_t(offset), C.c_size_t(n), int(store_pairs),
                                         C.c_size_t(simi.shape[0]), _f(simi), _i(idxi), C.byref(nup)))
        return nup.value

    def scan_codes_range(self, query, list_no, offset, n, radius):
        """InvertedListScanner::scan_codes_range over the same kind of run: (positions counted from offset, distances), in order"""
        query = f32(query)
        cnt = C.c_size_t(0)
        _chk(lib().amd_ivf_scan_codes_range(self._h, _f(query), C.c_size_t(list_no), C.c_size_t(offset), C.c_size_t(n), C.c_float(radius),
                                            C.byref(cnt)))
        pos = np.empty(cnt.value, np.uint32)
        dis = np.empty(cnt.value, np.float32)
        if cnt.value:
            _chk(lib().amd_ivf_scan_codes_range_results(self._h, pos.ctypes.data_as(C.POINTER(C.c_uint32)), _f(dis)))
        return pos, dis

    def distance_to_code(self, query, list_no, offset):
        query = f32(query)
        out = C.c_float(0)
        _chk(lib().amd_ivf_distance_to_code(self._h, _f(query), C.c_size_t(list_no), C.c_size_t(offset), C.byref(out)))
        return np.float32(out.value)

    def stats(self, reset=False):
        st = (C.c_size_t * 4)()
        _chk(lib().amd_ivf_stats(self._h, st, int(reset)))
        return dict(nq=st[0], nlist=st[1], ndis=st[2], nheap_updates=st[3])

    def set_queries(self, x):
        x = f32(x)
        _chk(lib().amd_ivf_set_queries(self._h, C.c_size_t(x.shape[0]), _f(x)))

    def search_resident(self, start, n, k, nprobe, coarse_mode=0):
        D = np.empty((n, k), np.float32)
        I = np.empty((n, k), np.int64)
        _chk(lib().amd_ivf_search_resident(self._h, C.c_size_t(start), C.c_size_t(n), C.c_size_t(k), C.c_size_t(nprobe),
                                           coarse_mode, _f(D), _i(I)))
        return D, I

    # ---- Auncel
    def set_interdis(self, table=None):
        t = f32(table) if table is not None else None
        _chk(lib().amd_ivf_set_interdis(self._h, _f(t)))

    def get_interdis(self):
        t = np.empty(self.nlist * (self.nlist - 1) // 2, np.float32)
        _chk(lib().amd_ivf_get_interdis(self._h, _f(t)))
        return t

    def set_tuner(self, max_topk, traces, arcos):
        """traces: list of (x, y, std)"""
        n = len(traces)
        lens = np.array([len(t[0]) for t in traces], dtype=np.uintp)
        xs, ys, ss = [f32(t[0]) for t in traces], [f32(t[1]) for t in traces], [f32(t[2]) for t in traces]
        arcos = f32(arcos)
        _chk(lib().amd_ivf_set_tuner(self._h, C.c_size_t(max_topk), C.c_size_t(n), lens.ctypes.data_as(_szp),
                                     (_f32p * n)(*[_f(a) for a in xs]), (_f32p * n)(*[_f(a) for a in ys]),
                                     (_f32p * n)(*[_f(a) for a in ss]), _f(arcos)))
        self.max_topk = max_topk

    def search_adaptive(self, start, n, query_topk, multipler, std_m, require_acc, my_nprobe, t_recalls, gt_D=None,
                        profile=False, coarse_mode=0, out=None):
        """out: optional (D, I) arrays to fill -- a caller that hands over page-locked memory gets its results by direct DMA"""
        req = f32(require_acc)
        gt = f32(gt_D) if gt_D is not None else None
        assert my_nprobe.dtype == np.uint64 and t_recalls.dtype == np.float32
        K = self.max_topk
        if out is not None:
            D, I = out
            assert D.shape == (n, K) and D.dtype == np.float32 and D.flags.c_contiguous
            assert I.shape == (n, K) and I.dtype == np.int64 and I.flags.c_contiguous
        else:
            D = np.empty((n, K), np.float32)
            I = np.empty((n, K), np.int64)
        _chk(lib().amd_ivf_search_adaptive(self._h, C.c_size_t(start), C.c_size_t(n), C.c_size_t(query_topk),
                                           C.c_float(multipler), C.c_float(std_m), _f(req), _f(gt), int(profile), coarse_mode,
                                           my_nprobe.ctypes.data_as(_u64p), _f(t_recalls), _f(D), _i(I)))
        return D, I

    def set_async_depth(self, depth):
        """searches amd_ivf_submit_adaptive keeps running at a time (internal contexts; fixed by the first submit)"""
        _chk(lib().amd_ivf_set_async_depth(self._h, int(depth)))

    def submit_adaptive(self, start, n, query_topk, multipler, std_m, require_acc, my_nprobe, t_recalls, gt_D=None, profile=False,
                        coarse_mode=0, out=None):
        """asynchronous search_adaptive: returns a ticket for wait(); every array passed stays referenced until then"""
        req = f32(require_acc)
        gt = f32(gt_D) if gt_D is not None else None
        assert my_nprobe.dtype == np.uint64 and t_recalls.dtype == np.float32
        K = self.max_topk
        if out is not None:
            D, I = out
            assert D.shape == (n, K) and D.dtype == np.float32 and D.flags.c_contiguous
            assert I.shape == (n, K) and I.dtype == np.int64 and I.flags.c_contiguous
        else:
            D = np.empty((n, K), np.float32)
            I = np.empty((n, K), np.int64)
        t = C.c_uint64(0)
        _chk(lib().amd_ivf_submit_adaptive(self._h, C.c_size_t(start), C.c_size_t(n), C.c_size_t(query_topk), C.c_float(multipler),
                                           C.c_float(std_m), _f(req), _f(gt), int(profile), coarse_mode,
                                           my_nprobe.ctypes.data_as(_u64p), _f(t_recalls), _f(D), _i(I), C.byref(t)))
        if not hasattr(self, "_tickets"):
            self._tickets = {}
        self._tickets[int(t.value)] = (req, gt, my_nprobe, t_recalls, D, I)
        return int(t.value)

    def submit_search_resident(self, start, n, k, nprobe, coarse_mode=0, out=None):
        """asynchronous search_resident: returns a ticket for wait()"""
        if out is not None:
            D, I = out
        else:
            D = np.empty((n, k), np.float32)
            I = np.empty((n, k), np.int64)
        t = C.c_uint64(0)
        _chk(lib().amd_ivf_submit_search_resident(self._h, C.c_size_t(start), C.c_size_t(n), C.c_size_t(k), C.c_size_t(nprobe), coarse_mode,
                                                  _f(D), _i(I), C.byref(t)))
        if not hasattr(self, "_tickets"):
            self._tickets = {}
        self._tickets[int(t.value)] = (None, None, None, None, D, I)
        return int(t.value)

    def submit_coarse_resident(self, start, n, nprobe, mode=0, want_dis=False):
        """asynchronous coarse_resident: wait() returns (coarse_dis or None, keys) in place of (D, I)"""
        keys = np.empty((n, nprobe), np.int64)
        dis = np.empty((n, nprobe), np.float32) if want_dis else None
        t = C.c_uint64(0)
        _chk(lib().amd_ivf_submit_coarse_resident(self._h, C.c_size_t(start), C.c_size_t(n), C.c_size_t(nprobe), _f(dis), _i(keys), mode,
                                                  C.byref(t)))
        if not hasattr(self, "_tickets"):
            self._tickets = {}
        self._tickets[int(t.value)] = (None, None, None, None, dis, keys)
        return int(t.value)

    def submit_search_resident_preassigned(self, start, n, k, keys, out=None):
        """asynchronous search_resident_preassigned (keys: n x nprobe, kept alive until wait())"""
        keys = i64(keys)
        assert keys.shape[0] == n
        if out is not None:
            D, I = out
        else:
            D = np.empty((n, k), np.float32)
            I = np.empty((n, k), np.int64)
        t = C.c_uint64(0)
        _chk(lib().amd_ivf_submit_search_resident_preassigned(self._h, C.c_size_t(start), C.c_size_t(n), C.c_size_t(k), C.c_size_t(keys.shape[1]),
                                                              _i(keys), _f(D), _i(I), C.byref(t)))
        if not hasattr(self, "_tickets"):
            self._tickets = {}
        self._tickets[int(t.value)] = (keys, None, None, None, D, I)
        return int(t.value)

    def async_counts(self):
        """(tickets served by the asynchronous entry points, passes that served them)"""
        out = (C.c_uint64 * 2)()
        _chk(lib().amd_ivf_async_counts(self._h, out))
        return int(out[0]), int(out[1])

    def wait(self, ticket):
        """blocks until the search of `ticket` has ended; returns (D, I, timing dict, diag dict); raises as the synchronous call"""
        tm = (C.c_double * 9)()
        dg = (C.c_uint64 * 4)()
        rc = lib().amd_ivf_wait(self._h, C.c_uint64(ticket), tm, dg)
        held = self._tickets.pop(ticket, None) if hasattr(self, "_tickets") else None
        _chk(rc)
        keys = ("coarse_ms", "scan_ms", "select_ms", "total_ms", "scan_launches", "scan_bytes", "slot_efficiency", "rounds", "scan_min_bytes")
        timing = {k: float(tm[i]) for i, k in enumerate(keys)}
        diag = {"hinted_launches": int(dg[0]), "short_hints": int(dg[1]), "tie_redone": int(dg[2]), "direct_out": bool(dg[3])}
        D, I = (held[4], held[5]) if held else (None, None)
        return D, I, timing, diag

    def search_adaptive_pre(self, x, id_offset, keys, coarse_dis, query_topk, multipler, std_m, require_acc, my_nprobe, t_recalls,
                            gt_D=None, profile=False):
        """search_preassigned with tune on over the caller's coarse ranking (keys / coarse_dis rows, nprobe entries each)"""
        x, keys, cd = f32(x), i64(keys), f32(coarse_dis)
        n, nprobe = keys.shape
        req = f32(require_acc)
        gt = f32(gt_D) if gt_D is not None else None
        assert my_nprobe.dtype == np.uint64 and t_recalls.dtype == np.float32
        K = self.max_topk
        D = np.empty((n, K), np.float32)
        I = np.empty((n, K), np.int64)
        _chk(lib().amd_ivf_search_adaptive_pre(self._h, C.c_size_t(n), _f(x), C.c_size_t(id_offset), C.c_size_t(nprobe), _i(keys), _f(cd),
                                               C.c_size_t(query_topk), C.c_float(multipler), C.c_float(std_m), _f(req), _f(gt),
                                               int(profile), my_nprobe.ctypes.data_as(_u64p), _f(t_recalls), _f(D), _i(I)))
        return D, I

    def train_samples_pre(self, x, id_offset, keys, coarse_dis, max_topk, gt_D, train_num, raw):
        """the training branch over the caller's coarse ranking"""
        x, keys, cd, gt = f32(x), i64(keys), f32(coarse_dis), f32(gt_D)
        n, nprobe = keys.shape
        D = np.empty((n, max_topk), np.float32)
        I = np.empty((n, max_topk), np.int64)
        ptrs = (_f32p * len(raw))(*[_f(r) for r in raw])
        _chk(lib().amd_ivf_train_samples_pre(self._h, C.c_size_t(n), _f(x), C.c_size_t(id_offset), C.c_size_t(nprobe), _i(keys), _f(cd),
                                             C.c_size_t(max_topk), _f(gt), C.c_size_t(train_num), ptrs, _f(D), _i(I)))
        return D, I

    def search_timed(self, start, n, k, nprobe, budget_ms, coarse_mode=0):
        """Error_sys::time_search over resident queries [start, start+n); budget_ms is indexed by absolute query id.
        Returns (D, I, nprobe_used)."""
        b = f32(budget_ms)
        assert b.size >= start + n
        D = np.empty((n, k), np.float32)
        I = np.empty((n, k), np.int64)
        used = np.zeros(n, np.uint64)
        _chk(lib().amd_ivf_search_timed(self._h, C.c_size_t(start), C.c_size_t(n), C.c_size_t(k), C.c_size_t(nprobe), _f(b), coarse_mode,
                                        used.ctypes.data_as(_u64p), _f(D), _i(I)))
        return D, I, used

    def search_timed_x(self, x, id_offset, k, nprobe, budget_ms, coarse_mode=0):
        x = f32(x)
        n = x.shape[0]
        b = f32(budget_ms)
        assert b.size >= id_offset + n
        D = np.empty((n, k), np.float32)
        I = np.empty((n, k), np.int64)
        used = np.zeros(n, np.uint64)
        _chk(lib().amd_ivf_search_timed_x(self._h, C.c_size_t(n), _f(x), C.c_size_t(id_offset), C.c_size_t(k), C.c_size_t(nprobe), _f(b),
                                          coarse_mode, used.ctypes.data_as(_u64p), _f(D), _i(I)))
        return D, I, used

    def train_samples(self, start, n, max_topk, gt_D, train_num, raw, coarse_mode=0):
        gt = f32(gt_D)
        D = np.empty((n, max_topk), np.float32)
        I = np.empty((n, max_topk), np.int64)
        ptrs = (_f32p * len(raw))(*[_f(r) for r in raw])
        _chk(lib().amd_ivf_train_samples(self._h, C.c_size_t(start), C.c_size_t(n), C.c_size_t(max_topk), _f(gt),
                                         C.c_size_t(train_num), coarse_mode, ptrs, _f(D), _i(I)))
        return D, I

    def scan_arith(self):
        """0 fp32 reference order, 1 fp32 fused, 2 byte codes (include/auncel_amd.h)"""
        return int(lib().amd_ivf_scan_arith(self._h))

    def last_timing(self):
        t = (C.c_double * 8)()
        lib().amd_ivf_last_timing(self._h, t)
        mb = C.c_double(0)
        lib().amd_ivf_last_scan_min_bytes(self._h, C.byref(mb))
        return dict(coarse_ms=t[0], scan_ms=t[1], select_ms=t[2], total_ms=t[3], scan_launches=t[4], scan_bytes=t[5],
                    slot_efficiency=t[6], rounds=t[7], scan_min_bytes=mb.value)

    PHASES = ("coarse", "scan_dense", "select_dense", "scan_thr", "select_thr", "tie_fix", "plan")

    def last_timing_detail(self):
        """{phase: (ms, launches)} of the last search on this handle (include/auncel_amd.h: amd_ivf_last_timing_detail)"""
        t = (C.c_double * 16)()
        lib().amd_ivf_last_timing_detail(self._h, t)
        d = {p: (t[2 * i], t[2 * i + 1]) for i, p in enumerate(self.PHASES)}
        d["min_bytes_dense"], d["min_bytes_thr"] = t[14], t[15]
        return d

    def coarse_tie_rows(self):
        """coarse rankings re-run through the reference's heap so far (runs of equal distances, include/auncel_amd.h)"""
        v = C.c_uint64(0)
        _chk(lib().amd_ivf_coarse_tie_rows(self._h, C.byref(v)))
        return int(v.value)

    def coarse_resident(self, start, n, nprobe, mode=0, want_dis=True):
        """the coarse ranking of resident queries [start, start + n) -> (coarse_dis or None, keys)"""
        keys = np.empty((n, nprobe), np.int64)
        dis = np.empty((n, nprobe), np.float32) if want_dis else None
        _chk(lib().amd_ivf_coarse_resident(self._h, C.c_size_t(start), C.c_size_t(n), C.c_size_t(nprobe), _f(dis), _i(keys), mode))
        return dis, keys

    def search_resident_preassigned(self, start, n, k, keys, out=None):
        """search_preassigned over resident queries [start, start + n) with the caller's keys (n x nprobe)"""
        keys = i64(keys)
        assert keys.shape[0] == n
        if out is not None:
            D, I = out
        else:
            D = np.empty((n, k), np.float32)
            I = np.empty((n, k), np.int64)
        _chk(lib().amd_ivf_search_resident_preassigned(self._h, C.c_size_t(start), C.c_size_t(n), C.c_size_t(k), C.c_size_t(keys.shape[1]),
                                                       _i(keys), _f(D), _i(I)))
        return D, I

    def last_round_hints(self):
        """(scan launches of the last search sized from the previous search's counts, those whose hint was too small)"""
        v = (C.c_uint64 * 2)()
        _chk(lib().amd_ivf_last_round_hints(self._h, v))
        return int(v[0]), int(v[1])

    def last_tie_redone(self):
        """queries the last adaptive call searched again with the reference's coarse tie order (AUNCEL_AMD_COARSE_TIES=redo)"""
        v = C.c_uint64(0)
        _chk(lib().amd_ivf_last_tie_redone(self._h, C.byref(v)))
        return int(v.value)

    def last_tie_patched(self):
        """rankings of the last adaptive call whose order the reference's heap changed while its one pass was under way"""
        v = C.c_uint64(0)
        _chk(lib().amd_ivf_last_tie_patched(self._h, C.byref(v)))
        return int(v.value)

    def last_tie_fixed(self):
        """queries of the last search whose result came from the heap replayed over their admission log (include/auncel_amd.h)"""
        v = C.c_uint64(0)
        _chk(lib().amd_ivf_last_tie_fixed(self._h, C.byref(v)))
        return int(v.value)

    def last_filter(self):
        """(threshold rounds of the last search that ran as matrix-core filter + exact recomputation, candidates the last one kept)"""
        v = (C.c_uint64 * 2)()
        _chk(lib().amd_ivf_last_filter(self._h, v))
        return int(v[0]), int(v[1])

    def last_direct_out(self):
        """True if the last search wrote (D, I) straight into the caller's (page-locked) buffers (include/auncel_amd.h)"""
        return bool(lib().amd_ivf_last_direct_out(self._h))

    def last_coarse_pick(self):
        """rankings of the last coarse call that came from matrix-core distances + exact recomputation (include/auncel_amd.h)"""
        v = C.c_uint64(0)
        _chk(lib().amd_ivf_last_coarse_pick(self._h, C.byref(v)))
        return int(v.value)

    def set_byte_codes(self, enable):
        """False: scan the fp32 lists even where the byte codes qualify (same results)"""
        lib().amd_ivf_set_byte_codes(self._h, int(bool(enable)))

    def set_option(self, key, value):
        """include/auncel_amd.h: amd_ivf_set_option (None returns the key to "unset")"""
        _chk(lib().amd_ivf_set_option(self._h, key.encode(), C.c_double(float("nan") if value is None else float(value))))

    def get_option(self, key):
        v = C.c_double(0)
        _chk(lib().amd_ivf_get_option(self._h, key.encode(), C.byref(v)))
        return v.value
