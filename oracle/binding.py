"""ctypes binding of oracle/liboracle.so (built by oracle/Makefile from mmseq_oracle.c).

TEST INFRASTRUCTURE ONLY -- see oracle/__init__.py.
"""
import ctypes as C
import os
import subprocess

import numpy as np

_HERE = os.path.dirname(os.path.abspath(__file__))
_LIB = None
_SOKAL_REF = None

u32p = np.ctypeslib.ndpointer(np.uint32, flags="C_CONTIGUOUS")
u64p = np.ctypeslib.ndpointer(np.uint64, flags="C_CONTIGUOUS")
i32p = np.ctypeslib.ndpointer(np.int32, flags="C_CONTIGUOUS")
f64p = np.ctypeslib.ndpointer(np.float64, flags="C_CONTIGUOUS")
u8p = np.ctypeslib.ndpointer(np.uint8, flags="C_CONTIGUOUS")


def build():
    """(Re)build liboracle.so (and oracle/_ref when the reference tree is present)."""
    subprocess.check_call(["make", "-C", _HERE, "-s"])


def lib():
    global _LIB
    if _LIB is not None:
        return _LIB
    path = os.path.join(_HERE, "liboracle.so")
    if not os.path.exists(path):
        build()
    if "OMP_NUM_THREADS" not in os.environ:
        # a container's CPU quota is invisible to OpenMP: one thread per host core, throttled to the quota's worth of time
        try:
            q, per = open("/sys/fs/cgroup/cpu.max").read().split()[:2]
            if q != "max":
                os.environ["OMP_NUM_THREADS"] = str(max(1, -(-int(q) // int(per))))
        except (OSError, ValueError):
            pass
    L = C.CDLL(path)
    L.orc_philox.argtypes = [u32p, u32p, u32p]
    L.orc_philox2x32.argtypes = [u32p, C.c_uint32, u32p]
    L.orc_log_v.argtypes = [C.c_int64, f64p, f64p]
    L.orc_exp_v.argtypes = [C.c_int64, f64p, f64p]
    L.orc_gamma_draw.argtypes = [C.c_uint64, C.c_uint32, C.c_uint32, C.c_uint64, C.c_double, C.c_double]
    L.orc_gamma_draw.restype = C.c_double
    L.orc_binomial_keyed.argtypes = [C.c_uint64, C.c_uint64, C.c_uint32, C.c_double]
    L.orc_binomial_keyed.restype = C.c_uint32
    L.orc_simu_gamma_trace.argtypes = [C.c_uint64, C.c_uint64, C.c_double, C.c_double, C.c_int, f64p]
    L.orc_sample_counts.argtypes = [C.c_uint64, C.c_uint32, u64p, u32p, C.c_void_p, f64p, C.c_uint64,
                                    C.c_uint32, C.c_uint32, C.c_uint64, i32p]
    L.orc_gamma_update.argtypes = [C.c_uint32, i32p, f64p, C.c_double, C.c_double, C.c_uint64, C.c_uint32,
                                   C.c_uint32, f64p]
    L.orc_gibbs_keyed.argtypes = [C.c_uint64, C.c_uint32, u64p, u32p, C.c_void_p, f64p, f64p, C.c_double,
                                  C.c_double, C.c_uint64, C.c_uint32, C.c_int, C.c_int, C.c_uint64,
                                  C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p]
    L.orc_gibbs_keyed.restype = C.c_int
    L.orc_gibbs_ref.argtypes = [C.c_uint64, C.c_uint32, u64p, u32p, C.c_void_p, f64p, f64p, C.c_double,
                                C.c_double, C.c_uint32, C.c_int, C.c_int, C.c_int, C.c_void_p, C.c_void_p,
                                C.c_void_p, C.POINTER(C.c_double)]
    L.orc_gibbs_ref.restype = C.c_int
    L.orc_mt19937.argtypes = [C.c_uint32, C.c_int64, u32p]
    L.orc_mt_gamma_v.argtypes = [C.c_uint32, C.c_double, C.c_double, C.c_int64, f64p]
    L.orc_mt_binomial_v.argtypes = [C.c_uint32, C.c_uint32, C.c_double, C.c_int64, u32p]
    L.orc_keyed_gamma_v.argtypes = [C.c_uint64, C.c_double, C.c_double, C.c_int64, f64p]
    L.orc_keyed_binomial_v.argtypes = [C.c_uint64, C.c_uint32, C.c_double, C.c_int64, u32p]
    L.orc_keyed_normal_v.argtypes = [C.c_uint64, C.c_int64, f64p]
    L.orc_start_values.argtypes = [C.c_uint64, C.c_uint32, u64p, u32p, C.c_void_p, f64p, f64p, i32p]
    L.orc_start_values_exact.argtypes = [C.c_uint64, C.c_uint32, u64p, u32p, C.c_void_p, f64p, f64p]
    L.orc_em.argtypes = [C.c_uint64, C.c_uint32, u64p, u32p, C.c_void_p, f64p, f64p, C.c_int, C.c_double,
                         C.POINTER(C.c_double)]
    L.orc_em.restype = C.c_int
    L.orc_em_seq.argtypes = L.orc_em.argtypes
    L.orc_em_x.argtypes = L.orc_em.argtypes + [C.POINTER(C.c_int)]
    L.orc_em_x.restype = C.c_int
    L.orc_em_seq.restype = C.c_int
    L.orc_uh.argtypes = [C.c_uint64, C.c_uint32, u64p, u32p, C.c_void_p, C.c_uint32, u8p, i32p]
    L.orc_sokal.argtypes = [C.c_int, f64p, C.POINTER(C.c_double), C.POINTER(C.c_double), C.POINTER(C.c_int)]
    L.orc_sokal.restype = C.c_int
    L.orc_synth_transcripts.argtypes = [C.c_uint64, C.c_uint32, f64p, f64p, f64p]
    L.orc_synth_len_cdf.argtypes = [C.c_double, f64p]
    L.orc_synth_row_len.argtypes = [C.c_uint64, C.c_uint64, f64p]
    L.orc_synth_row_len.restype = C.c_uint32
    L.orc_synth_csr.argtypes = [C.c_uint64, C.c_uint64, C.c_uint64, C.c_uint32, f64p, f64p, C.c_int, C.c_double, u64p,
                                C.c_void_p]
    L.orc_synth_csr_genes.argtypes = [C.c_uint64, C.c_uint64, C.c_uint64, C.c_uint32, f64p, f64p, C.c_double, C.c_uint32, C.c_uint32, u64p,
                                      C.c_void_p]
    _LIB = L
    return L


def _kptr(k):
    return None if k is None else k.ctypes.data_as(C.c_void_p)


def _optr(a):
    return None if a is None else a.ctypes.data_as(C.c_void_p)


# ----------------------------------------------------------------------------- primitives
def philox(ctr, key):
    out = np.zeros(4, np.uint32)
    lib().orc_philox(np.asarray(ctr, np.uint32), np.asarray(key, np.uint32), out)
    return out


def philox2x32(ctr, key):
    out = np.zeros(2, np.uint32)
    lib().orc_philox2x32(np.asarray(ctr, np.uint32), int(key), out)
    return out


def log_v(x):
    x = np.ascontiguousarray(x, np.float64)
    out = np.empty_like(x)
    lib().orc_log_v(x.size, x, out)
    return out


def exp_v(x):
    x = np.ascontiguousarray(x, np.float64)
    out = np.empty_like(x)
    lib().orc_exp_v(x.size, x, out)
    return out


def mt19937(seed, n):
    out = np.empty(n, np.uint32)
    lib().orc_mt19937(seed, n, out)
    return out


def simu_gamma_trace(seed, sid, shape, scale, n):
    out = np.empty(n, np.float64)
    lib().orc_simu_gamma_trace(seed, sid, shape, scale, n, out)
    return out


# ----------------------------------------------------------------------------- Gibbs
class Problem:
    """CSR hit-set problem: rows = hit sets (or reads, k=1), columns = observed transcripts."""

    def __init__(self, row_ptr, col_idx, l, k=None, mu0=None, n=None):
        self.row_ptr = np.ascontiguousarray(row_ptr, np.uint64)
        self.col_idx = np.ascontiguousarray(col_idx, np.uint32)
        self.l = np.ascontiguousarray(l, np.float64)
        self.k = None if k is None else np.ascontiguousarray(k, np.uint32)
        self.m = self.row_ptr.size - 1
        self.n = int(self.l.size if n is None else n)
        self.mu0 = None if mu0 is None else np.ascontiguousarray(mu0, np.float64)
        assert int(self.row_ptr[-1]) == self.col_idx.size

    @property
    def nnz(self):
        return int(self.col_idx.size)

    def total_k(self):
        return int(self.m if self.k is None else self.k.sum(dtype=np.int64))


def sample_counts(p, mu, seed, chain, it, row_id_base=0):
    cnt = np.zeros(p.n, np.int32)
    lib().orc_sample_counts(p.m, p.n, p.row_ptr, p.col_idx, _kptr(p.k), np.ascontiguousarray(mu, np.float64),
                            seed, chain, it, row_id_base, cnt)
    return cnt


def gamma_update(cnt, l, alpha, beta, seed, chain, it):
    mu = np.empty(len(l), np.float64)
    lib().orc_gamma_update(len(l), np.ascontiguousarray(cnt, np.int32), np.ascontiguousarray(l, np.float64),
                           alpha, beta, seed, chain, it, mu)
    return mu


def gibbs_keyed(p, mu0, alpha=0.1, beta=0.1, seed=1234, chain=0, n_iter=1024, trace_len=1024,
                row_id_base=0, want_trace=True):
    """Full keyed chain; returns dict(trace[n,trace_len], cnt, sum_log, sum_log2, mu)."""
    trace = np.empty((p.n, trace_len), np.float64) if want_trace else None
    cnt = np.empty(p.n, np.int32)
    sl = np.empty(p.n, np.float64)
    sl2 = np.empty(p.n, np.float64)
    mu = np.empty(p.n, np.float64)
    rc = lib().orc_gibbs_keyed(p.m, p.n, p.row_ptr, p.col_idx, _kptr(p.k), p.l,
                               np.ascontiguousarray(mu0, np.float64), alpha, beta, seed, chain, n_iter,
                               trace_len, row_id_base, _optr(trace), _optr(cnt), _optr(sl), _optr(sl2),
                               _optr(mu))
    if rc != 0:
        raise ValueError("orc_gibbs_keyed rc=%d" % rc)
    return dict(trace=trace, cnt=cnt, sum_log=sl, sum_log2=sl2, mu=mu)


def gibbs_ref(p, mu0, alpha=0.1, beta=0.1, seed=1234, n_iter=1024, trace_len=1024, threads=1,
              want_trace=True):
    """Reference-structured chain (MT19937 per thread). Returns dict incl. loop seconds."""
    trace = np.empty((p.n, trace_len), np.float64) if want_trace else None
    cnt = np.empty(p.n, np.int32)
    mu = np.empty(p.n, np.float64)
    secs = C.c_double(0.0)
    rc = lib().orc_gibbs_ref(p.m, p.n, p.row_ptr, p.col_idx, _kptr(p.k), p.l,
                             np.ascontiguousarray(mu0, np.float64), alpha, beta, seed, n_iter, trace_len,
                             threads, _optr(trace), _optr(cnt), _optr(mu), C.byref(secs))
    if rc != 0:
        raise ValueError("orc_gibbs_ref rc=%d" % rc)
    return dict(trace=trace, cnt=cnt, mu=mu, seconds=secs.value)


def start_values(p):
    mu0 = np.empty(p.n, np.float64)
    uh = np.empty(p.n, np.int32)
    lib().orc_start_values(p.m, p.n, p.row_ptr, p.col_idx, _kptr(p.k), p.l, mu0, uh)
    return mu0, uh


def start_values_exact(p):
    """mu0 as the device computes it (mmg_problem_start_values): exact fixed-point sum of the shares, order-independent."""
    mu0 = np.empty(p.n, np.float64)
    lib().orc_start_values_exact(p.m, p.n, p.row_ptr, p.col_idx, _kptr(p.k), p.l, mu0)
    return mu0


def em(p, mu, max_iter=1000, epsilon=0.1):
    mu = np.array(mu, np.float64, copy=True)
    ll = C.c_double(0.0)
    it = lib().orc_em(p.m, p.n, p.row_ptr, p.col_idx, _kptr(p.k), p.l, mu, max_iter, epsilon, C.byref(ll))
    return mu, it, ll.value


def em_x(p, mu, max_iter=1000, epsilon=0.1):
    """em() that also reports how many rows passes were repeated on the safe scale: (mu, iterations, loglik, redo)."""
    mu = np.array(mu, np.float64, copy=True)
    ll = C.c_double(0.0)
    redo = C.c_int(0)
    it = lib().orc_em_x(p.m, p.n, p.row_ptr, p.col_idx, _kptr(p.k), p.l, mu, max_iter, epsilon, C.byref(ll), C.byref(redo))
    return mu, it, ll.value, redo.value


def em_seq(p, mu, max_iter=1000, epsilon=0.1):
    """EM in the reference's own summation order (src/mmseq.cpp:761-811); pins em() to the reference arithmetic."""
    mu = np.array(mu, np.float64, copy=True)
    ll = C.c_double(0.0)
    it = lib().orc_em_seq(p.m, p.n, p.row_ptr, p.col_idx, _kptr(p.k), p.l, mu, max_iter, epsilon, C.byref(ll))
    return mu, it, ll.value


def uh(p, member):
    member = np.ascontiguousarray(member, np.uint8)
    G = member.shape[1]
    res = np.empty(G, np.int32)
    lib().orc_uh(p.m, p.n, p.row_ptr, p.col_idx, _kptr(p.k), G, member, res)
    return res


def sokal(x):
    """Oracle restatement of src/sokal.cc:33-87. Returns (rc, var, tau, m)."""
    x = np.array(x, np.float64, copy=True)
    var, tau, m = C.c_double(0), C.c_double(0), C.c_int(0)
    rc = lib().orc_sokal(x.size, x, C.byref(var), C.byref(tau), C.byref(m))
    return rc, var.value, tau.value, m.value


def sokal_ref(x):
    """The compiled REFERENCE sokal (oracle/_ref/libsokal_ref.so); None if absent."""
    global _SOKAL_REF
    path = os.path.join(_HERE, "_ref", "libsokal_ref.so")
    if not os.path.exists(path):
        return None
    if _SOKAL_REF is None:
        _SOKAL_REF = C.CDLL(path)
        _SOKAL_REF.sokal.argtypes = [C.POINTER(C.c_int), f64p, C.POINTER(C.c_double), C.POINTER(C.c_double),
                                     C.POINTER(C.c_int)]
        _SOKAL_REF.sokal.restype = C.c_int
    x = np.array(x, np.float64, copy=True)
    n = C.c_int(x.size)
    var, tau, m = C.c_double(0), C.c_double(0), C.c_int(0)
    rc = _SOKAL_REF.sokal(C.byref(n), x, C.byref(var), C.byref(tau), C.byref(m))
    return rc, var.value, tau.value, m.value


# ----------------------------------------------------------------------------- canonical layout
# Restates mmseq_amd/csrc/mmg_types.h + layout.hip:k_row_keys: the order in which libmmgibbs stores the rows of a problem.
LAYOUT_BAND_SHIFT = 6
LAYOUT_NEAR_SPAN = 240
K_SMALL = 64
LAYOUT_EXPAND_MAX_RATIO = 8
K_DRAWS_PER_HIT = 16


SELL_WIN = 255  # transcripts per LDS window (mmg_types.h)


def draws_categoricals(k, L):
    """mmg_types.h: draws_categoricals (spec version 8): k categorical draws while k <= min(K_SMALL, K_DRAWS_PER_HIT * (hits - 1)), or k <= 1;
    above: the conditional-binomial chain.  k, L: integer arrays (L >= 1)."""
    k = np.asarray(k).astype(np.uint64)
    L = np.asarray(L).astype(np.uint64)
    cap = np.minimum(np.uint64(K_DRAWS_PER_HIT) * (L - np.uint64(1)), np.uint64(K_SMALL))
    return (k <= 1) | (k <= cap)


def row_keys(row_ptr, col_idx, k=None):
    """(key, tie) per row: key = !near << 63 | band << 18 | kclass << 16 | (kclass == 1 ? k : k_bucket(k)) << 9 | min(len, 0x1ff) (0 for an empty
    row); kclass 0 (k <= 1), 1 (k categorical draws: draws_categoricals), 3 (conditional-binomial chain); band = the band
    of the smallest hit for a near row, the home band (one below the band of hit[(len - 1) // 2]) for a far row;
    tie = csum << 48 | hash >> 16, hash = fold of (len, k, hits in stored order), csum = sum of (hit - 64 * band) over the hits inside
    [64 * band, 64 * band + SELL_WIN).  Spec: mmseq_amd/csrc/mmg_types.h."""
    rp = np.asarray(row_ptr).astype(np.int64)
    col = np.asarray(col_idx, np.uint32)
    L = np.diff(rp)
    m = L.size
    kk = np.ones(m, np.uint64) if k is None else np.asarray(k).astype(np.uint64)
    key = np.zeros(m, np.uint64)
    ne = L > 0
    if ne.any():
        starts = rp[:-1][ne]
        lo = np.minimum.reduceat(col, starts).astype(np.uint64)
        hi = np.maximum.reduceat(col, starts).astype(np.uint64)
        band = lo >> np.uint64(LAYOUT_BAND_SHIFT)
        Ln = L[ne].astype(np.uint64)
        near = (Ln <= 255) & (hi - (band << np.uint64(LAYOUT_BAND_SHIFT)) < LAYOUT_NEAR_SPAN)
        mid = col[starts + (L[ne] - 1) // 2].astype(np.uint64) >> np.uint64(LAYOUT_BAND_SHIFT)
        band = np.where(near, band, np.maximum(mid, np.uint64(1)) - np.uint64(1))
        kn = kk[ne]
        kclass = np.where(kn <= 1, 0, np.where(draws_categoricals(kn, Ln), 1, 3)).astype(np.uint64)   # spec version 8: no class 2
        # class 2 rows sort by a logarithmic bucket of k (spec version 5; mmg_types.h: k_bucket)
        kb = np.maximum(kn, 65).astype(np.uint64)
        e = np.floor(np.log2(kb.astype(np.float64))).astype(np.uint64)      # exact for k < 2^32: log2 of an integer below 2^53
        e = np.where((np.uint64(1) << e) > kb, e - np.uint64(1), e)
        e = np.where((np.uint64(2) << e) <= kb, e + np.uint64(1), e)
        bucket = np.minimum(np.uint64(8) * (e - np.uint64(6)) + ((kb >> (e - np.uint64(3))) & np.uint64(7)), np.uint64(127))
        ksmall = np.where(kclass == 1, kn, np.where(kclass >= 2, bucket, 0)).astype(np.uint64)
        key[ne] = ((~near).astype(np.uint64) << np.uint64(63)) | (band << np.uint64(18)) | (kclass << np.uint64(16)) | \
            (ksmall << np.uint64(9)) | np.minimum(Ln, 0x1ff)
    with np.errstate(over="ignore"):
        h = np.uint64(0x9E3779B97F4A7C15) + L.astype(np.uint64) + (kk << np.uint64(32))
        M = np.uint64(0xFF51AFD7ED558CCD)
        for j in range(int(L.max()) if m else 0):
            sel = np.nonzero(L > j)[0]
            c = col[rp[sel] + j].astype(np.uint64)
            hj = (h[sel] ^ c) * M
            h[sel] = hj ^ (hj >> np.uint64(32))
    # tie order inside a key (ABI 4): the CENTRE of the row first -- the sum of its hits' offsets in the window its key names
    # (hits outside that window count 0) -- then the content hash: tie = csum << 48 | hash >> 16.  Rows of one tile then gather
    # neighbouring window slots at every step of the walk (fewer LDS bank conflicts, mmg_types.h).
    cs = np.zeros(m, np.uint64)
    if col.size:
        wbase = (((key >> np.uint64(18)) & np.uint64((1 << 45) - 1)) << np.uint64(LAYOUT_BAND_SHIFT)).astype(np.uint32)
        rid = np.repeat(np.arange(m, dtype=np.int64), L)
        with np.errstate(over="ignore"):
            d = (col - wbase[rid]).astype(np.uint32)        # u32 wrap-around: hits below the window are outside
        cs = np.bincount(rid, weights=np.where(d < SELL_WIN, d, 0).astype(np.float64), minlength=m).astype(np.uint64)
    tie = (cs << np.uint64(48)) | (h >> np.uint64(16))
    return key, tie


def permute_rows(row_ptr, col_idx, k, perm):
    rp = np.asarray(row_ptr).astype(np.int64)
    lens = np.diff(rp)[perm]
    new_rp = np.zeros(perm.size + 1, np.uint64)
    new_rp[1:] = np.cumsum(lens)
    idx = np.repeat(rp[:-1][perm] - new_rp[:-1].astype(np.int64), lens) + np.arange(int(lens.sum()), dtype=np.int64)
    return new_rp, np.ascontiguousarray(np.asarray(col_idx)[idx]), (None if k is None else np.ascontiguousarray(np.asarray(k)[perm]))



def sort_hits(row_ptr, col_idx):
    """Every row's hits in ascending order (a row is a set; src/mmseq.cpp:871 walks it in ascending order)."""
    rp = np.asarray(row_ptr).astype(np.int64)
    col = np.asarray(col_idx, np.uint32)
    if col.size == 0:
        return col.copy()
    rid = np.repeat(np.arange(rp.size - 1, dtype=np.int64), np.diff(rp))
    return np.ascontiguousarray(col[np.lexsort((col, rid))])


def canonical_layout(row_ptr, col_idx, k=None):
    """Rows in the library's stored order (spec: mmseq_amd/csrc/mmg_types.h): hits ascending within every row, rows sorted by
    (key, hash), ties in the caller's order; a far row then keeps the hits inside its home window in front of the others.
    A caller row that draws k >= 2 categoricals (draws_categoricals) is stored as k rows with k = 1 (k comes back as None when no other
    multiplicity is left).
    Returns (row_ptr, col_idx, k, perm) with perm[stored row] = caller row."""
    col_sorted = sort_hits(row_ptr, col_idx)
    src = None
    if k is not None:
        # step 0 (ABI 4; spec version 8: the rows that draw categoricals, draws_categoricals): a row with k >= 2 categorical draws is stored
        # as k rows with k = 1; an array of ones is no array
        kk0 = np.asarray(k).astype(np.int64)
        L0 = np.diff(np.asarray(row_ptr).astype(np.int64))
        reps = np.where((kk0 >= 2) & (L0 >= 1) & draws_categoricals(kk0, np.maximum(L0, 1)), kk0, 1)
        if int(reps.sum()) > LAYOUT_EXPAND_MAX_RATIO * kk0.size or int(reps.sum()) >= 0xffffffff:
            reps = np.ones_like(reps)          # a heavily collapsed file is not un-collapsed (mmg_types.h: LAYOUT_EXPAND_MAX_RATIO): the rows keep their k
        if (reps > 1).any():
            src = np.repeat(np.arange(kk0.size, dtype=np.int64), reps)
            row_ptr, col_sorted, k = permute_rows(row_ptr, col_sorted, np.where(reps > 1, 1, kk0).astype(np.uint32), src)
        if (np.asarray(k) == 1).all():
            k = None
    key, h = row_keys(row_ptr, col_sorted, k)
    perm = np.lexsort((h, key))
    rp, ci, kk = permute_rows(row_ptr, col_sorted, k, perm)
    skey = key[perm]
    if src is not None:
        perm = src[perm]
    far = (skey >> np.uint64(63)).astype(bool)
    if far.any() and ci.size:
        lens = np.diff(rp.astype(np.int64))
        rid = np.repeat(np.arange(lens.size, dtype=np.int64), lens)
        wbase = (((skey >> np.uint64(18)) & np.uint64((1 << 45) - 1)) << np.uint64(LAYOUT_BAND_SHIFT)).astype(np.uint32)
        outside = far[rid] & ((ci - wbase[rid]).astype(np.uint32) >= SELL_WIN)  # u32 wrap-around: hits below the window are outside
        ci = np.ascontiguousarray(ci[np.lexsort((ci, outside, rid))])
    return rp, ci, kk, perm


# ----------------------------------------------------------------------------- synthetic input
def synth_problem(R, T, avg_hits, seed=1234, uniform=False, row0=0, mapped_reads=None, sort=True, far_fraction=0.0, gene_size=0, far_family=0):
    """Synthetic problem of SURVEY.md App. D: rows [row0,row0+R), k=1. Returns (Problem, tables).  sort: rows in the library's
    canonical order (what mmg_problem_create_synthetic stores with sorted = 1); otherwise generator order."""
    L = lib()
    efflen = np.empty(T, np.float64)
    theta = np.empty(T, np.float64)
    cdf = np.empty(T, np.float64)
    L.orc_synth_transcripts(seed, T, efflen, theta, cdf)
    len_cdf = np.empty(99, np.float64)
    L.orc_synth_len_cdf(float(avg_hits) - 1.0, len_cdf)
    row_ptr = np.empty(R + 1, np.uint64)
    if gene_size:                                       # gene-block mode (mmg_synth_desc.gene_size / far_family)
        assert not uniform
        L.orc_synth_csr_genes(seed, row0, R, T, cdf, len_cdf, float(far_fraction), int(gene_size), int(far_family), row_ptr, None)
        col = np.empty(int(row_ptr[-1]), np.uint32)
        L.orc_synth_csr_genes(seed, row0, R, T, cdf, len_cdf, float(far_fraction), int(gene_size), int(far_family), row_ptr, col.ctypes.data_as(C.c_void_p))
    else:
        L.orc_synth_csr(seed, row0, R, T, cdf, len_cdf, int(uniform), float(far_fraction), row_ptr, None)
        col = np.empty(int(row_ptr[-1]), np.uint32)
        L.orc_synth_csr(seed, row0, R, T, cdf, len_cdf, int(uniform), float(far_fraction), row_ptr, col.ctypes.data_as(C.c_void_p))
    if sort and R > 0:
        row_ptr, col, _, _ = canonical_layout(row_ptr, col)
    nreads = R if mapped_reads is None else mapped_reads
    l = efflen * float(nreads) / 1e9  # src/mmseq.cpp:603
    return Problem(row_ptr, col, l), dict(efflen=efflen, theta=theta, cdf=cdf, len_cdf=len_cdf)
