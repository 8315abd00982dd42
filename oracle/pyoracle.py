"""ctypes binding of oracle/liboracle.so (the CPU restatement) and oracle/_ref/libref_hnsw.so.

TEST INFRASTRUCTURE ONLY: importable from tests/, __graft_entry__.smoke() and bench.py's
cpu_baseline leg.  The product package (regard3d_amd/) must never import this module.
"""
from __future__ import annotations

import ctypes as C
import os
import subprocess

import numpy as np

_HERE = os.path.dirname(os.path.abspath(__file__))
_LIB = None
_REF = None


class Match(C.Structure):
    _fields_ = [("i", C.c_uint32), ("j", C.c_uint32)]


class FResult(C.Structure):
    _fields_ = [("F", C.c_double * 9), ("threshold", C.c_double), ("nfa", C.c_double),
                ("n_inliers", C.c_uint32), ("n_iter", C.c_uint32), ("n_models", C.c_uint32),
                ("accepted", C.c_int)]


def build(force: bool = False) -> None:
    """Compile liboracle.so (and oracle/_ref when /root/reference exists)."""
    so = os.path.join(_HERE, "liboracle.so")
    srcs = [os.path.join(_HERE, f) for f in ("matching.c", "acransac.c", "io.c", "liop.c", "kgraph.c", "essential.c", "akaze.c", "hnsw.c", "mrpt.c", "r3d_oracle.h", "Makefile")]
    stale = force or not os.path.exists(so) or any(os.path.getmtime(s) > os.path.getmtime(so) for s in srcs)
    if stale:
        subprocess.check_call(["make", "-C", _HERE, "liboracle.so"], stdout=subprocess.DEVNULL)
    ref = os.path.join(_HERE, "_ref", "libref_hnsw.so")
    ref2 = os.path.join(_HERE, "_ref", "libref_liop.so")
    if os.path.isdir("/root/reference/src/thirdparty/hnswlib") and (force or not os.path.exists(ref) or not os.path.exists(ref2)):
        subprocess.check_call(["make", "-C", _HERE, "ref"], stdout=subprocess.DEVNULL)


def lib():
    global _LIB
    if _LIB is None:
        so = os.path.join(_HERE, "liboracle.so")
        if not os.path.exists(so):
            build()
        L = C.CDLL(so)
        L.orc_l2sq_f32.restype = C.c_float
        L.orc_l2sq_u8.restype = C.c_float
        L.orc_hamming.restype = C.c_uint32
        L.orc_match_collection.restype = C.c_int64
        L.orc_filter_F_collection.restype = C.c_int64
        L.orc_rng_u64.restype = C.c_uint64
        L.orc_rng_u64.argtypes = [C.c_uint64, C.c_uint32, C.c_uint32, C.c_uint32, C.c_uint32]
        L.orc_sym_epipolar_err.restype = C.c_double
        L.orc_filter_E_collection.restype = C.c_int64
        L.orc_filter_H_collection.restype = C.c_int64
        L.orc_epipolar_dist_err.restype = C.c_double
        L.orc_epipolar_dist_err.argtypes = [C.c_void_p] + [C.c_double] * 4
        L.orc_akaze_kcontrast.restype = C.c_float
        L.orc_kgraph_build_exact.restype = C.c_void_p
        L.orc_kgraph_build_nndescent.restype = C.c_void_p
        L.orc_kgraph_free.argtypes = [C.c_void_p]
        L.orc_kgraph_size.argtypes = [C.c_void_p]; L.orc_kgraph_size.restype = C.c_uint32
        L.orc_kgraph_edges.argtypes = [C.c_void_p]; L.orc_kgraph_edges.restype = C.c_uint64
        L.orc_kgraph_export.argtypes = [C.c_void_p] * 4
        L.orc_match_collection_kgraph.restype = C.c_int64
        L.orc_sym_epipolar_err.argtypes = [C.c_void_p] + [C.c_double] * 4
        _LIB = L
    return _LIB


def ref_lib():
    """hnswlib::BruteforceSearch compiled from the reference tree (None if not built)."""
    global _REF
    if _REF is None:
        so = os.path.join(_HERE, "_ref", "libref_hnsw.so")
        if not os.path.exists(so):
            return None
        _REF = C.CDLL(so)
    return _REF


_REF_LIOP = None


def ref_liop_lib():
    """the reference's own vl_liop.c compiled stand-alone (None if not built)"""
    global _REF_LIOP
    if _REF_LIOP is None:
        so = os.path.join(_HERE, "_ref", "libref_liop.so")
        if not os.path.exists(so):
            return None
        L = C.CDLL(so)
        L.r3d_vl_liopdesc_new_basic.restype = C.c_void_p
        L.r3d_vl_liopdesc_new_basic.argtypes = [C.c_size_t]
        L.r3d_vl_liopdesc_process.argtypes = [C.c_void_p, C.c_void_p, C.c_void_p]
        L.r3d_vl_liopdesc_delete.argtypes = [C.c_void_p]
        _REF_LIOP = L
    return _REF_LIOP


def ref_liop(patches: np.ndarray) -> np.ndarray:
    L = ref_liop_lib()
    if L is None:
        raise RuntimeError("oracle/_ref/libref_liop.so not built")
    patches = np.ascontiguousarray(patches, np.float32)
    n, side = patches.shape[0], patches.shape[1]
    out = np.zeros((n, 144), np.float32)
    h = L.r3d_vl_liopdesc_new_basic(side)
    for k in range(n):
        L.r3d_vl_liopdesc_process(h, out[k].ctypes.data_as(C.c_void_p), patches[k].ctypes.data_as(C.c_void_p))
    L.r3d_vl_liopdesc_delete(h)
    return out


def liop_extract_patches(image: np.ndarray, kps: np.ndarray, kp_size_factor: float = 8.0) -> np.ndarray:
    image = np.ascontiguousarray(image, np.float32); kps = np.ascontiguousarray(kps, np.float32)
    n = kps.shape[0]
    out = np.zeros((n, 41, 41), np.float32)
    lib().orc_liop_extract_patches(_p(image), image.shape[1], image.shape[0], _p(kps), n, C.c_float(kp_size_factor), _p(out))
    return out


def liop_describe(patches: np.ndarray) -> np.ndarray:
    patches = np.ascontiguousarray(patches, np.float32)
    n, side = patches.shape[0], patches.shape[1]
    out = np.zeros((n, 144), np.float32)
    lib().orc_liop_describe(_p(patches), n, side, _p(out))
    return out


def _p(a):
    return a.ctypes.data_as(C.c_void_p)


def _dtype_code(desc: np.ndarray, binary: bool) -> int:
    if desc.dtype == np.float32:
        return 0
    if desc.dtype == np.uint8:
        return 2 if binary else 1
    raise TypeError(desc.dtype)


def l2sq(a: np.ndarray, b: np.ndarray) -> float:
    a = np.ascontiguousarray(a); b = np.ascontiguousarray(b)
    if a.dtype == np.uint8:
        return float(lib().orc_l2sq_u8(_p(a), _p(b), C.c_size_t(a.size)))
    return float(lib().orc_l2sq_f32(_p(a.astype(np.float32)), _p(b.astype(np.float32)), C.c_size_t(a.size)))


def hamming(a: np.ndarray, b: np.ndarray) -> int:
    a = np.ascontiguousarray(a, dtype=np.uint8); b = np.ascontiguousarray(b, dtype=np.uint8)
    return int(lib().orc_hamming(_p(a), _p(b), C.c_size_t(a.size)))


def knn2(dataset: np.ndarray, query: np.ndarray, binary: bool = False):
    """Brute-force 2-NN of every query row among the dataset rows -> (idx [nJ,2], dist [nJ,2])."""
    dataset = np.ascontiguousarray(dataset); query = np.ascontiguousarray(query)
    nI, dim = dataset.shape; nJ = query.shape[0]
    idx = np.full((nJ, 2), -1, np.int32)
    code = _dtype_code(dataset, binary)
    if code == 2:
        dist = np.zeros((nJ, 2), np.uint32)
        rc = lib().orc_knn2_hamming(_p(dataset), nI, _p(query), nJ, dim, _p(idx), _p(dist))
    elif code == 1:
        dist = np.zeros((nJ, 2), np.float32)
        rc = lib().orc_knn2_l2_u8(_p(dataset), nI, _p(query), nJ, dim, _p(idx), _p(dist))
    else:
        dist = np.zeros((nJ, 2), np.float32)
        rc = lib().orc_knn2_l2_f32(_p(dataset), nI, _p(query), nJ, dim, _p(idx), _p(dist))
    if rc != 0:
        raise ValueError("knn2 failed (nJ < 1 or nI < 2)")
    return idx, dist


def ref_knn(dataset: np.ndarray, query: np.ndarray, k: int = 2):
    L = ref_lib()
    if L is None:
        raise RuntimeError("oracle/_ref/libref_hnsw.so not built")
    dataset = np.ascontiguousarray(dataset, np.float32); query = np.ascontiguousarray(query, np.float32)
    nI, dim = dataset.shape; nJ = query.shape[0]
    idx = np.zeros((nJ, k), np.int32); dist = np.zeros((nJ, k), np.float32)
    rc = L.ref_hnsw_knn_l2(_p(dataset), nI, _p(query), nJ, dim, k, _p(idx), _p(dist))
    if rc != 0:
        raise RuntimeError("ref_hnsw_knn_l2 failed")
    return idx, dist


def match_distance_ratio(descI, descJ, ratio: float, squared: bool = True, xyI=None, xyJ=None,
                         binary: bool = False) -> np.ndarray:
    descI = np.ascontiguousarray(descI); descJ = np.ascontiguousarray(descJ)
    nI, dim = descI.shape; nJ = descJ.shape[0]
    out = np.zeros((max(nJ, 1), 2), np.uint32)
    xi = np.ascontiguousarray(xyI, np.float32) if xyI is not None else None
    xj = np.ascontiguousarray(xyJ, np.float32) if xyJ is not None else None
    m = lib().orc_match_distance_ratio(_dtype_code(descI, binary), _p(descI), nI,
                                       _p(xi) if xi is not None else None,
                                       _p(descJ), nJ, _p(xj) if xj is not None else None, dim,
                                       C.c_float(ratio), int(squared), _p(out))
    return out[:m].copy()


def match_collection(descs, xys, pairs: np.ndarray, ratio: float, squared: bool = True, binary: bool = False):
    """descs: list of [n_i, dim] arrays.  pairs: [P,2] uint32.  -> (counts [P], matches [M,2])."""
    n = len(descs)
    descs = [np.ascontiguousarray(d) for d in descs]
    dim = descs[0].shape[1]
    code = _dtype_code(descs[0], binary)
    dptr = (C.c_void_p * n)(*[d.ctypes.data for d in descs])
    nrows = np.array([d.shape[0] for d in descs], np.int32)
    if xys is not None:
        xys = [np.ascontiguousarray(x, np.float32) for x in xys]
        xptr = (C.c_void_p * n)(*[x.ctypes.data for x in xys])
    else:
        xptr = None
    pairs = np.ascontiguousarray(pairs, np.uint32)
    P = pairs.shape[0]
    counts = np.zeros(P, np.uint32)
    cap = int(sum(int(nrows[j]) for j in pairs[:, 1])) + 1
    out = np.zeros((cap, 2), np.uint32)
    tot = lib().orc_match_collection(code, n, dptr, _p(nrows), xptr, dim, _p(pairs), C.c_int64(P),
                                     C.c_float(ratio), int(squared), _p(counts), _p(out), C.c_int64(cap))
    if tot < 0:
        raise RuntimeError("capacity")
    return counts, out[:tot].copy()


def acransac_F(xI, xJ, wI, hI, wJ, hJ, precision_px=4.0, max_iter=2048, seed=5489, I=0, J=1):
    xI = np.ascontiguousarray(xI, np.float64); xJ = np.ascontiguousarray(xJ, np.float64)
    m = xI.shape[0]
    inl = np.zeros(max(m, 1), np.uint32)
    fr = FResult()
    n = lib().orc_acransac_F(_p(xI), _p(xJ), m, wI, hI, wJ, hJ, C.c_double(precision_px), C.c_uint32(max_iter),
                             C.c_uint64(seed), C.c_uint32(I), C.c_uint32(J), _p(inl), C.byref(fr))
    return inl[:n].copy(), fr


def acransac_H(xI, xJ, wI, hI, wJ, hJ, precision_px=4.0, max_iter=2048, seed=5489, I=0, J=1):
    xI = np.ascontiguousarray(xI, np.float64); xJ = np.ascontiguousarray(xJ, np.float64)
    m = xI.shape[0]
    inl = np.zeros(max(m, 1), np.uint32)
    fr = FResult()
    n = lib().orc_acransac_H(_p(xI), _p(xJ), m, wI, hI, wJ, hJ, C.c_double(precision_px), C.c_uint32(max_iter),
                             C.c_uint64(seed), C.c_uint32(I), C.c_uint32(J), _p(inl), C.byref(fr))
    return inl[:n].copy(), fr


def four_point_h(x1, x2):
    x1 = np.ascontiguousarray(x1, np.float64); x2 = np.ascontiguousarray(x2, np.float64)
    H = np.zeros(9, np.float64)
    lib().orc_four_point_h(_p(x1), _p(x2), _p(H))
    return H.reshape(3, 3)


def acransac_F_traced(xI, xJ, wI, hI, wJ, hJ, precision_px=4.0, max_iter=2048, seed=5489, I=0, J=1, cap=16384):
    buf = np.zeros((cap, 5), np.float64)
    lib().orc_set_trace(_p(buf), cap)
    inl, fr = acransac_F(xI, xJ, wI, hI, wJ, hJ, precision_px, max_iter, seed, I, J)
    n = lib().orc_trace_rows()
    lib().orc_set_trace(None, 0)
    return inl, fr, buf[:n].copy()


def filter_H_collection(xys, widths, heights, pairs, counts, matches, precision_px=4.0, max_iter=2048,
                        seed=5489, want_F=False):
    return filter_F_collection(xys, widths, heights, pairs, counts, matches, precision_px, max_iter, seed, want_F,
                               _fn="orc_filter_H_collection")


def filter_F_collection(xys, widths, heights, pairs, counts, matches, precision_px=4.0, max_iter=2048,
                        seed=5489, want_F=False, _fn="orc_filter_F_collection"):
    n = len(xys)
    xys = [np.ascontiguousarray(x, np.float32) for x in xys]
    xptr = (C.c_void_p * n)(*[x.ctypes.data for x in xys])
    nrows = np.array([x.shape[0] for x in xys], np.int32)
    widths = np.ascontiguousarray(widths, np.uint32); heights = np.ascontiguousarray(heights, np.uint32)
    pairs = np.ascontiguousarray(pairs, np.uint32); counts = np.ascontiguousarray(counts, np.uint32)
    matches = np.ascontiguousarray(matches, np.uint32).reshape(-1, 2)
    P = pairs.shape[0]
    oc = np.zeros(P, np.uint32)
    out = np.zeros((max(matches.shape[0], 1), 2), np.uint32)
    Fo = np.zeros((P, 9), np.float64) if want_F else None
    fn = getattr(lib(), _fn); fn.restype = C.c_int64
    tot = fn(n, _p(nrows), xptr, _p(widths), _p(heights), _p(pairs), C.c_int64(P),
                                        _p(counts), _p(matches), C.c_double(precision_px), C.c_uint32(max_iter),
                                        C.c_uint64(seed), _p(oc), _p(out), _p(Fo) if want_F else None)
    return (oc, out[:tot].copy(), Fo) if want_F else (oc, out[:tot].copy())


def seven_point(x1, x2):
    x1 = np.ascontiguousarray(x1, np.float64); x2 = np.ascontiguousarray(x2, np.float64)
    Fs = np.zeros((3, 9), np.float64)
    n = lib().orc_seven_point(_p(x1), _p(x2), _p(Fs))
    return Fs[:n].reshape(n, 3, 3).copy()


def solve_cubic(coeffs):
    c = np.ascontiguousarray(coeffs, np.float64); r = np.zeros(3, np.float64)
    n = lib().orc_solve_cubic(_p(c), _p(r))
    return r[:n].copy()


def sample7(seed, I, J, it, pool):
    pool = np.ascontiguousarray(pool, np.uint32)
    s = np.zeros(7, np.uint32)
    lib().orc_sample7(C.c_uint64(seed), C.c_uint32(I), C.c_uint32(J), C.c_uint32(it), _p(pool),
                      C.c_uint32(pool.size), _p(s))
    return s


def logcombi_tables(n, k=7):
    a = np.zeros(n + 1, np.float32); b = np.zeros(n + 1, np.float32)
    lib().orc_logcombi_tables(C.c_uint32(n), C.c_uint32(k), _p(a), _p(b))
    return a, b


def save_matches(path, pairs, counts, matches):
    pairs = np.ascontiguousarray(pairs, np.uint32); counts = np.ascontiguousarray(counts, np.uint32)
    matches = np.ascontiguousarray(matches, np.uint32).reshape(-1, 2)
    rc = lib().orc_save_matches(path.encode(), C.c_int64(pairs.shape[0]), _p(pairs), _p(counts), _p(matches))
    if rc != 0:
        raise IOError(rc)


def load_matches(path):
    np_ = C.c_int64(0); nm = C.c_int64(0)
    rc = lib().orc_load_matches(path.encode(), C.byref(np_), C.byref(nm), None, None, None)
    if rc != 0:
        raise IOError(rc)
    pairs = np.zeros((np_.value, 2), np.uint32); counts = np.zeros(np_.value, np.uint32)
    matches = np.zeros((max(nm.value, 1), 2), np.uint32)
    rc = lib().orc_load_matches(path.encode(), C.byref(np_), C.byref(nm), _p(pairs), _p(counts), _p(matches))
    if rc != 0:
        raise IOError(rc)
    return pairs, counts, matches[:nm.value].copy()


# ---- KGraph plugin path (oracle/kgraph.c) -------------------------------------------------------
KGRAPH_PRESETS = {          # src/R3DComputeMatches.cpp:844-873: (K, L, recall, P)
    "fast": (2, 20, 0.6, 2), "medium": (16, 24, 0.2, 6), "precise": (16, 24, 0.8, 12), "default": (16, 24, 0.99, 10),
}


class KGraphIndex:
    """owning handle of an orc_kgraph (CSR: off[n+1], ids, dist)"""

    def __init__(self, handle, data):
        if not handle:
            raise RuntimeError("index build failed (too few rows?)")
        self.h = C.c_void_p(handle)
        self.data = data

    def __del__(self):
        try:
            lib().orc_kgraph_free(self.h)
        except Exception:
            pass

    def csr(self):
        n = lib().orc_kgraph_size(self.h); e = lib().orc_kgraph_edges(self.h)
        off = np.zeros(n + 1, np.uint64); ids = np.zeros(max(e, 1), np.uint32); dist = np.zeros(max(e, 1), np.float32)
        lib().orc_kgraph_export(self.h, _p(off), _p(ids), _p(dist))
        return off, ids[:e], dist[:e]

    def knn2(self, query, P=10, S=10, seed=1998, I=0, J=1, min_rows=0):
        query = np.ascontiguousarray(query, np.float32)
        nq = query.shape[0]
        idx = np.zeros((nq, 2), np.int32); dist = np.zeros((nq, 2), np.float32); comps = C.c_uint64(0)
        rc = lib().orc_kgraph_knn2(self.h, _p(self.data), self.data.shape[1], _p(query), nq, P, S, C.c_uint64(seed), I, J,
                                   min_rows, _p(idx), _p(dist), C.byref(comps))
        if rc != 0:
            raise RuntimeError("orc_kgraph_knn2 failed")
        return idx, dist, comps.value


def kgraph_build_exact(data, K=16, cap=64) -> KGraphIndex:
    data = np.ascontiguousarray(data, np.float32)
    return KGraphIndex(lib().orc_kgraph_build_exact(_p(data), data.shape[0], data.shape[1], K, cap), data)


def kgraph_build_nndescent(data, K=16, L=24, recall=0.99, S=10, R=100, iterations=30, delta=0.002, controls=100, seed=1998):
    data = np.ascontiguousarray(data, np.float32)
    info = np.zeros(4, np.float32)
    h = lib().orc_kgraph_build_nndescent(_p(data), data.shape[0], data.shape[1], K, L, S, R, iterations,
                                         C.c_float(recall), C.c_float(delta), controls, seed, _p(info))
    g = KGraphIndex(h, data)
    g.info = dict(iterations=int(info[0]), recall=float(info[1]), delta=float(info[2]), cost=float(info[3]))
    return g


def kgraph_seeds(seed, I, J, q, n, P):
    out = np.zeros(P, np.uint32)
    lib().orc_kgraph_seeds(C.c_uint64(seed), I, J, q, n, P, _p(out))
    return out


def match_collection_kgraph(descs, xys, pairs, ratio, builder="exact", K=16, L=24, recall=0.99, cap=64, P=10, S=10,
                            seed=1998, min_rows=0):
    pairs = np.ascontiguousarray(pairs, np.uint32).reshape(-1, 2)
    n_img = len(descs)
    descs = [np.ascontiguousarray(d, np.float32) for d in descs]
    dim = descs[0].shape[1]
    desc_p = (C.c_void_p * n_img)(*[d.ctypes.data for d in descs])
    n_rows = np.array([d.shape[0] for d in descs], np.int32)
    if xys is not None:
        xys = [np.ascontiguousarray(x, np.float32) for x in xys]
        xy_p = (C.c_void_p * n_img)(*[x.ctypes.data for x in xys])
    else:
        xy_p = None
    counts = np.zeros(len(pairs), np.uint32)
    cap_out = int(sum(int(n_rows[j]) for j in pairs[:, 1])) + 1
    out = np.zeros((cap_out, 2), np.uint32)
    comps = C.c_uint64(0)
    tot = lib().orc_match_collection_kgraph(n_img, desc_p, _p(n_rows), xy_p, dim, _p(pairs), C.c_int64(len(pairs)),
                                            C.c_float(ratio), 0 if builder == "exact" else 1, K, L, C.c_float(recall), cap,
                                            P, S, C.c_uint64(seed), min_rows, _p(counts), _p(out), C.c_int64(cap_out),
                                            C.byref(comps))
    if tot < 0:
        raise RuntimeError("orc_match_collection_kgraph: output capacity")
    return counts, out[:tot].copy(), comps.value


# ---- essential matrix (oracle/essential.c) ----------------------------------------------------------
def five_point(x1, x2):
    x1 = np.ascontiguousarray(x1, np.float64); x2 = np.ascontiguousarray(x2, np.float64)
    Es = np.zeros((10, 3, 3), np.float64)
    n = lib().orc_five_point(_p(x1), _p(x2), _p(Es))
    return Es[:n].copy()


def real_roots(coeffs_ascending):
    c = np.ascontiguousarray(coeffs_ascending, np.float64)
    r = np.zeros(10, np.float64)
    n = lib().orc_real_roots10(_p(c), len(c) - 1, _p(r))
    return r[:n].copy()


def acransac_E(xI, xJ, wI, hI, wJ, hJ, K1, K2, precision_px=4.0, max_iter=2048, seed=5489, I=0, J=1):
    xI = np.ascontiguousarray(xI, np.float64); xJ = np.ascontiguousarray(xJ, np.float64)
    K1 = np.ascontiguousarray(K1, np.float64); K2 = np.ascontiguousarray(K2, np.float64)
    m = xI.shape[0]
    inl = np.zeros(max(m, 1), np.uint32)
    res = FResult()
    n = lib().orc_acransac_E(_p(xI), _p(xJ), m, wI, hI, wJ, hJ, _p(K1), _p(K2), C.c_double(precision_px), max_iter,
                             C.c_uint64(seed), I, J, _p(inl), C.byref(res))
    return inl[:n].copy(), res


def filter_E_collection(xys, widths, heights, Ks, pairs, counts, matches, precision_px=4.0, max_iter=2048, seed=5489,
                        prune_min_count=50, prune_min_ratio=0.3, want_E=False):
    """Ks: [n_images, 3, 3] float64; an all-zero K marks a view without intrinsics"""
    n = len(xys)
    xys = [np.ascontiguousarray(x, np.float32) for x in xys]
    xptr = (C.c_void_p * n)(*[x.ctypes.data for x in xys])
    nrows = np.array([x.shape[0] for x in xys], np.int32)
    widths = np.ascontiguousarray(widths, np.uint32); heights = np.ascontiguousarray(heights, np.uint32)
    Ks = np.ascontiguousarray(Ks, np.float64)
    pairs = np.ascontiguousarray(pairs, np.uint32); counts = np.ascontiguousarray(counts, np.uint32)
    matches = np.ascontiguousarray(matches, np.uint32)
    oc = np.zeros(len(pairs), np.uint32); out = np.zeros((max(len(matches), 1), 2), np.uint32)
    Eo = np.zeros((len(pairs), 9), np.float64)
    tot = lib().orc_filter_E_collection(n, _p(nrows), xptr, _p(widths), _p(heights), _p(Ks), _p(pairs), C.c_int64(len(pairs)),
                                        _p(counts), _p(matches), C.c_double(precision_px), max_iter, C.c_uint64(seed),
                                        prune_min_count, C.c_float(prune_min_ratio), _p(oc), _p(out), _p(Eo))
    return (oc, out[:tot].copy(), Eo) if want_E else (oc, out[:tot].copy())


# ---- Fast-A-KAZE detector (oracle/akaze.c) ------------------------------------------------------------
def akaze_gaussian(img, sigma):
    img = np.ascontiguousarray(img, np.float32); out = np.empty_like(img)
    lib().orc_akaze_gaussian(_p(img), img.shape[1], img.shape[0], C.c_float(sigma), _p(out))
    return out


def akaze_scharr(img):
    img = np.ascontiguousarray(img, np.float32); lx = np.empty_like(img); ly = np.empty_like(img)
    lib().orc_akaze_scharr(_p(img), img.shape[1], img.shape[0], _p(lx), _p(ly))
    return lx, ly


def akaze_scaled_deriv(img, s, dx):
    img = np.ascontiguousarray(img, np.float32); out = np.empty_like(img)
    lib().orc_akaze_scaled_deriv(_p(img), img.shape[1], img.shape[0], s, int(dx), _p(out))
    return out


def akaze_kcontrast(lx, ly, perc=0.7, nbins=300):
    lx = np.ascontiguousarray(lx, np.float32); ly = np.ascontiguousarray(ly, np.float32)
    return float(lib().orc_akaze_kcontrast(_p(lx), _p(ly), lx.shape[1], lx.shape[0], C.c_float(perc), nbins))


def akaze_halfsample(img):
    img = np.ascontiguousarray(img, np.float32)
    out = np.empty((img.shape[0] // 2, img.shape[1] // 2), np.float32)
    lib().orc_akaze_halfsample(_p(img), img.shape[1], img.shape[0], _p(out))
    return out


def akaze_fed_tau(T):
    tau = np.zeros(256, np.float32)
    n = lib().orc_akaze_fed_tau(C.c_float(T), _p(tau))
    return tau[:n].copy()


def akaze_detect(img, threshold=0.001, cap=200000, dbg_level=-1):
    """-> dict(kps [n,4] (x, y, size, angle_deg), responses, levels, info[, ldet, lt of dbg_level])"""
    img = np.ascontiguousarray(img, np.float32)
    h, w = img.shape
    kps = np.zeros((cap, 4), np.float32); resp = np.zeros(cap, np.float32); lev = np.zeros(cap, np.int32)
    info = np.zeros(8, np.float32)
    ldet = np.zeros((h, w), np.float32) if dbg_level >= 0 else None
    lt = np.zeros((h, w), np.float32) if dbg_level >= 0 else None
    n = lib().orc_akaze_detect(_p(img), w, h, C.c_float(threshold), _p(kps), cap, _p(resp), _p(lev), dbg_level,
                               _p(ldet) if ldet is not None else None, _p(lt) if lt is not None else None, _p(info))
    n = min(n, cap)
    out = dict(kps=kps[:n].copy(), responses=resp[:n].copy(), levels=lev[:n].copy(), info=info)
    if dbg_level >= 0:
        lw, lh = int(info[2]), int(info[3])
        out["ldet"] = ldet.ravel()[:lw * lh].reshape(lh, lw).copy()
        out["lt"] = lt.ravel()[:lw * lh].reshape(lh, lw).copy()
    return out


def akaze_detect_mldb(img, threshold=0.001, cap=200000):
    """AKAZE2::detectAndCompute(DESCRIPTOR_MLDB): -> (kps [n,4], desc [n,61] uint8, responses)"""
    img = np.ascontiguousarray(img, np.float32)
    h, w = img.shape
    kps = np.zeros((cap, 4), np.float32); resp = np.zeros(cap, np.float32); desc = np.zeros((cap, 61), np.uint8)
    n = min(lib().orc_akaze_detect_mldb(_p(img), w, h, C.c_float(threshold), _p(kps), _p(desc), cap, _p(resp)), cap)
    return kps[:n].copy(), desc[:n].copy(), resp[:n].copy()


# ---- HNSW plugin path: the restatement (hnsw.c) and the reference-built library (oracle/_ref) ----
class HnswIndex:
    """orc_hnsw: built by the restatement (hnswlib's single-thread insertion) or wrapped around exported arrays"""

    def __init__(self, handle, data, M):
        self._h, self._data, self.M = handle, data, M

    def __del__(self):
        try:
            lib().orc_hnsw_free(C.c_void_p(self._h))
        except Exception:
            pass

    def export(self):
        n = self._data.shape[0]; M = self.M
        L = lib(); L.orc_hnsw_up_rows.restype = C.c_uint32
        rows = L.orc_hnsw_up_rows(C.c_void_p(self._h))
        levels = np.zeros(n, np.int32); links0 = np.zeros((n, 1 + 2 * M), np.int32); up_off = np.zeros(n + 1, np.int32)
        up = np.zeros((max(rows, 1), 1 + M), np.int32); ep = C.c_int32(0); ml = C.c_int32(0)
        L.orc_hnsw_export(C.c_void_p(self._h), _p(levels), _p(links0), _p(up_off), _p(up), C.byref(ep), C.byref(ml))
        return dict(levels=levels, links0=links0, up_off=up_off, up_links=up[:rows], enterpoint=ep.value, maxlevel=ml.value)

    def knn2(self, query, ef):
        query = np.ascontiguousarray(query, np.float32)
        idx = np.zeros((len(query), 2), np.int32); dist = np.zeros((len(query), 2), np.float32)
        rc = lib().orc_hnsw_knn2(C.c_void_p(self._h), _p(query), len(query), ef, _p(idx), _p(dist), None)
        if rc != 0:
            raise ValueError("orc_hnsw_knn2 failed")
        return idx, dist


HNSW_PRESETS = {"fast": (5, 112, 5), "medium": (15, 112, 10), "precise": (19, 100, 15)}      # M, efConstruction, ef (src/R3DComputeMatches.cpp:533-565)


def hnsw_levels(n, M, seed=100):
    out = np.zeros(n, np.int32)
    lib().orc_hnsw_levels(n, M, seed, _p(out))
    return out


def hnsw_build(data, M, ef_construction, seed=100) -> HnswIndex:
    data = np.ascontiguousarray(data, np.float32)
    L = lib(); L.orc_hnsw_build.restype = C.c_void_p
    h = L.orc_hnsw_build(_p(data), data.shape[0], data.shape[1], M, ef_construction, seed)
    if not h:
        raise ValueError("orc_hnsw_build: dim % 16 != 0 or M out of range")
    return HnswIndex(h, data, M)


def hnsw_build_batch(data, M, seed=100) -> HnswIndex:
    """the batch construction the HIP path uses (hnsw.c: orc_hnsw_build_batch)"""
    data = np.ascontiguousarray(data, np.float32)
    L = lib(); L.orc_hnsw_build_batch.restype = C.c_void_p
    h = L.orc_hnsw_build_batch(_p(data), data.shape[0], data.shape[1], M, seed)
    if not h:
        raise ValueError("orc_hnsw_build_batch: dim % 16 != 0 or M out of range")
    return HnswIndex(h, data, M)


def hnsw_from_arrays(data, M, ix) -> HnswIndex:
    data = np.ascontiguousarray(data, np.float32)
    L = lib(); L.orc_hnsw_from_arrays.restype = C.c_void_p
    lv = np.ascontiguousarray(ix["levels"], np.int32); l0 = np.ascontiguousarray(ix["links0"], np.int32)
    uo = np.ascontiguousarray(ix["up_off"], np.int32); ul = np.ascontiguousarray(ix["up_links"], np.int32).reshape(-1, 1 + M)
    if ul.shape[0] == 0:
        ul = np.zeros((1, 1 + M), np.int32)
    h = L.orc_hnsw_from_arrays(_p(data), data.shape[0], data.shape[1], M, _p(lv), _p(l0), _p(uo), _p(ul), int(ix["enterpoint"]), int(ix["maxlevel"]))
    return HnswIndex(h, data, M)


def match_collection_hnsw(descs, xys, pairs, ratio, preset="precise", builder="batch", min_rows=128, seed=100):
    """hnsw_match over a collection (hnsw.c: orc_match_collection_hnsw); builder "batch" = the HIP path's index, "hnswlib" = the reference's"""
    pairs = np.ascontiguousarray(pairs, np.uint32).reshape(-1, 2)
    n_img = len(descs)
    descs = [np.ascontiguousarray(d, np.float32) for d in descs]
    dim = descs[0].shape[1]
    desc_p = (C.c_void_p * n_img)(*[d.ctypes.data for d in descs])
    n_rows = np.array([d.shape[0] for d in descs], np.int32)
    if xys is not None:
        xys = [np.ascontiguousarray(x, np.float32) for x in xys]
        xy_p = (C.c_void_p * n_img)(*[x.ctypes.data for x in xys])
    else:
        xy_p = None
    M, efc, ef = HNSW_PRESETS[preset] if isinstance(preset, str) else preset
    counts = np.zeros(len(pairs), np.uint32)
    cap_out = int(sum(int(n_rows[j]) for j in pairs[:, 1])) + 1
    out = np.zeros((cap_out, 2), np.uint32)
    L = lib(); L.orc_match_collection_hnsw.restype = C.c_int64
    tot = L.orc_match_collection_hnsw(n_img, desc_p, _p(n_rows), xy_p, dim, _p(pairs), C.c_int64(len(pairs)), C.c_float(ratio),
                                      0 if builder == "batch" else 1, M, efc, ef, seed, min_rows, _p(counts), _p(out), C.c_int64(cap_out))
    if tot < 0:
        raise RuntimeError("orc_match_collection_hnsw: output capacity")
    return counts, out[:tot].copy()


def ref_hnsw_export(dataset, query, M, ef_construction, ef):
    """the reference-built HierarchicalNSW (single-thread insertion) as arrays + its searchKnn(ef, 2) of `query`"""
    R = ref_lib()
    if R is None:
        raise RuntimeError("oracle/_ref/libref_hnsw.so not built")
    dataset = np.ascontiguousarray(dataset, np.float32); query = np.ascontiguousarray(query, np.float32)
    n, dim = dataset.shape; nq = query.shape[0]
    levels = np.zeros(n, np.int32); links0 = np.zeros((n, 1 + 2 * M), np.int32); up_off = np.zeros(n + 1, np.int32)
    cap = n + 64
    up = np.zeros((cap, 1 + M), np.int32); ep = C.c_int32(0); ml = C.c_int32(0)
    idx = np.zeros((nq, 2), np.int32); dist = np.zeros((nq, 2), np.float32)
    rc = R.ref_hnsw_export(_p(dataset), n, _p(query), nq, dim, M, ef_construction, ef, _p(levels), _p(links0), _p(up_off), _p(up), cap,
                           C.byref(ep), C.byref(ml), _p(idx), _p(dist))
    if rc != 0:
        raise RuntimeError(f"ref_hnsw_export failed ({rc})")
    return dict(levels=levels, links0=links0, up_off=up_off, up_links=up[:up_off[n]], enterpoint=ep.value, maxlevel=ml.value), idx, dist


# ---- MRPT plugin path (matchingAlgorithm 5): the CPU model (mrpt.c) ----
MRPT_PRESET = dict(n_trees=26, depth=6, votes=5, density=0.088)      # src/R3DComputeMatches.cpp:453-456


class MrptIndex:
    def __init__(self, handle, data, n_trees, depth):
        self._h, self._data, self.n_trees, self.depth = handle, data, n_trees, depth

    def __del__(self):
        try:
            lib().orc_mrpt_free(C.c_void_p(self._h))
        except Exception:
            pass

    def export(self):
        n, dim = self._data.shape
        nl = 1 << self.depth
        R = np.zeros((self.n_trees * self.depth, dim), np.float32); sp = np.zeros((self.n_trees, nl - 1), np.float32)
        lv = np.zeros((self.n_trees, n), np.int32); lf = np.zeros(nl + 1, np.int32)
        lib().orc_mrpt_export(C.c_void_p(self._h), _p(R), _p(sp), _p(lv), _p(lf))
        return dict(R=R, splits=sp, leaves=lv, leaf_first=lf)

    def knn2(self, query, votes):
        query = np.ascontiguousarray(query, np.float32)
        idx = np.zeros((len(query), 2), np.int32); dist = np.zeros((len(query), 2), np.float32); ne = np.zeros(len(query), np.uint32)
        rc = lib().orc_mrpt_knn2(C.c_void_p(self._h), _p(query), len(query), votes, _p(idx), _p(dist), _p(ne))
        if rc != 0:
            raise ValueError("orc_mrpt_knn2 failed")
        return idx, dist, ne


def mrpt_depth_for(n, depth):
    L = lib(); L.orc_mrpt_depth_for.restype = C.c_uint32
    return int(L.orc_mrpt_depth_for(n, depth))


def mrpt_build(data, n_trees=26, depth=6, density=0.088, seed=0) -> MrptIndex:
    """Mrpt::grow(n_trees, depth clamped as ArrayMatcher_mrpt::Build clamps it, density)"""
    data = np.ascontiguousarray(data, np.float32)
    L = lib(); L.orc_mrpt_build.restype = C.c_void_p
    L.orc_mrpt_build.argtypes = [C.c_void_p, C.c_uint32, C.c_uint32, C.c_uint32, C.c_uint32, C.c_float, C.c_uint64]
    d = mrpt_depth_for(data.shape[0], depth)
    h = L.orc_mrpt_build(_p(data), data.shape[0], data.shape[1], n_trees, d, density, seed)
    return MrptIndex(h, data, n_trees, d)


def mrpt_random_matrix(n_pool, dim, density, seed=0):
    R = np.zeros((n_pool, dim), np.float32)
    L = lib(); L.orc_mrpt_random_matrix.argtypes = [C.c_uint32, C.c_uint32, C.c_float, C.c_uint64, C.c_void_p]
    L.orc_mrpt_random_matrix(n_pool, dim, density, seed, _p(R))
    return R


def ratio_dedup(idx, dist, xyI, xyJ, ratio, squared_metric=True):
    """NNdistanceRatio + both de-duplications on a 2-NN table (matching.c: orc_ratio_dedup_f32) -> matches [m, 2] (i_, j_)"""
    idx = np.ascontiguousarray(idx, np.int32); dist = np.ascontiguousarray(dist, np.float32)
    nJ = idx.shape[0]
    out = np.zeros((max(nJ, 1), 2), np.uint32)
    xi = np.ascontiguousarray(xyI, np.float32) if xyI is not None else None
    xj = np.ascontiguousarray(xyJ, np.float32) if xyJ is not None else None
    m = lib().orc_ratio_dedup_f32(_p(idx), _p(dist), nJ, _p(xi) if xi is not None else None, _p(xj) if xj is not None else None,
                                  C.c_float(ratio), 1 if squared_metric else 0, _p(out))
    return out[:m].copy()


def match_collection_mrpt(descs, xys, pairs, ratio, n_trees=26, depth=6, votes=5, density=None, seed=0, min_rows=128):
    """mrpt_match over a collection with the CPU model: per pair (counts, matches) as the GPU path must produce them.  Views below
    min_rows are scanned exactly with the squared ratio (the exhaustive arm), the others answer through their forest with the
    un-squared ratio on square-root distances; dropped queries (-1) match nothing."""
    pairs = np.ascontiguousarray(pairs, np.uint32).reshape(-1, 2)
    descs = [np.ascontiguousarray(d, np.float32) for d in descs]
    index = {}
    counts = np.zeros(len(pairs), np.uint32); out = []
    for k, (I, J) in enumerate(pairs):
        dI, dJ = descs[I], descs[J]
        if len(dI) == 0 or len(dJ) == 0:
            continue
        if len(dI) < min_rows:
            if len(dI) < 2:
                continue
            idx, dist = knn2(dI, dJ)
            m = ratio_dedup(idx, dist, xys[I], xys[J], ratio, True)
        else:
            if I not in index:
                dens = density if density and density > 0 else float(np.float32(1.0 / np.sqrt(np.float64(dI.shape[1]))))
                index[I] = mrpt_build(dI, n_trees, depth, dens, seed)
            idx, dist, _ = index[I].knn2(dJ, votes)
            dropped = idx[:, 0] < 0
            idx = idx.copy(); dist = dist.copy()
            idx[dropped] = 0; dist[dropped] = (1.0, 0.0)
            m = ratio_dedup(idx, dist, xys[I], xys[J], ratio, False)
        counts[k] = len(m); out.append(m)
    return counts, (np.concatenate(out) if out else np.zeros((0, 2), np.uint32))
