"""Host-side view of the Gibbs hot path (thin mirror of include/mmgibbs.h).

`Problem` is the reference's (M, k, l) triple (src/mmseq.cpp:117, :385, :593-608) resident in
HBM; `Sampler` is the state of the loop at src/mmseq.cpp:851-918.  All compute happens in
libmmgibbs.so's HIP kernels; this module only marshals numpy arrays across the C ABI.
"""
import ctypes as C

import numpy as np

from . import _lib
from ._lib import Config, ProblemDesc, ProblemInfo, SynthDesc, Timing, check


def device_count():
    n = C.c_int(0)
    check(_lib.load().mmg_device_count(C.byref(n)))
    return n.value


def _ptr(a):
    return None if a is None else a.ctypes.data_as(C.c_void_p)


class Problem:
    """Device-resident CSR hit-set matrix + multiplicities + l."""

    def __init__(self, handle):
        self._h = handle
        self._lib = _lib.load()

    @classmethod
    def from_csr(cls, row_ptr, col_idx, l, k=None, row_id_base=0, device=0, keep_rows=False, tx_order=None):
        """keep_rows: MMG_LAYOUT_KEEP_ROWS (default: the library's canonical row order); tx_order: n uint64 keys, the
        device numbers transcripts by ascending (key, index) -- every array crossing this class stays in the caller's."""
        lib = _lib.load()
        row_ptr = np.ascontiguousarray(row_ptr, np.uint64)
        col_idx = np.ascontiguousarray(col_idx, np.uint32)
        l = np.ascontiguousarray(l, np.float64)
        k = None if k is None else np.ascontiguousarray(k, np.uint32)
        tx_order = None if tx_order is None else np.ascontiguousarray(tx_order, np.uint64)
        if row_ptr.size < 1:
            raise ValueError("row_ptr needs m+1 >= 1 entries")
        if k is not None and k.size != row_ptr.size - 1:
            raise ValueError("k must have one entry per row")
        if tx_order is not None and tx_order.size != l.size:
            raise ValueError("tx_order must have one key per transcript")
        d = ProblemDesc(row_ptr.size - 1, l.size, _ptr(row_ptr), _ptr(col_idx), _ptr(k), _ptr(l), row_id_base,
                        _lib.LAYOUT_KEEP_ROWS if keep_rows else _lib.LAYOUT_CANONICAL, _ptr(tx_order))
        h = C.c_void_p()
        check(lib.mmg_problem_create(C.byref(d), device, C.byref(h)))
        return cls(h)

    @classmethod
    def synthetic(cls, rows, n, avg_hits, seed=1234, row0=0, uniform=False, mapped_reads=0, device=0,
                  sort=True, far_fraction=0.0, gene_size=0, far_family=0):
        """gene_size > 0: gene-block mode (a read's hits are isoforms of its gene); far_family F >= 2: a far hit goes to another gene of
        the read's paralogue family of F genes (include/mmgibbs.h: mmg_synth_desc)."""
        lib = _lib.load()
        d = SynthDesc(seed, rows, row0, n, float(avg_hits), int(uniform), int(sort), mapped_reads, float(far_fraction), int(gene_size), int(far_family))
        h = C.c_void_p()
        check(lib.mmg_problem_create_synthetic(C.byref(d), device, C.byref(h)))
        return cls(h)

    @property
    def info(self):
        inf = ProblemInfo()
        check(self._lib.mmg_problem_info_get(self._h, C.byref(inf)))
        return inf

    def download(self, with_k=False):
        """The CSR in STORED order (caller's transcript numbering): what a checker replays row by row."""
        inf = self.info
        rp = np.empty(inf.m + 1, np.uint64)
        ci = np.empty(inf.nnz, np.uint32)
        k = np.empty(inf.m, np.uint32) if with_k else None
        check(self._lib.mmg_problem_download(self._h, _ptr(rp), _ptr(ci), _ptr(k)))
        return (rp, ci, k) if with_k else (rp, ci)

    def shard(self, lo, hi, device=0):
        """Rows [lo, hi) of the stored problem as a problem of its own (mmg_problem_shard: cut and copied on the device side)."""
        h = C.c_void_p()
        check(self._lib.mmg_problem_shard(self._h, int(lo), int(hi), int(device), C.byref(h)))
        return Problem(h)

    def shard_bounds(self, parts):
        b = np.empty(parts + 1, np.uint64)
        check(self._lib.mmg_problem_shard_bounds(self._h, int(parts), _ptr(b)))
        return b

    def shard_bounds_timed(self, mu, parts):
        """mmg_problem_shard_bounds_timed: the cut by measured cost (profiling sweeps with the weights mu)."""
        mu = np.ascontiguousarray(mu, np.float64)
        assert mu.size == self.info.n
        b = np.empty(parts + 1, np.uint64)
        check(self._lib.mmg_problem_shard_bounds_timed(self._h, _ptr(mu), int(parts), _ptr(b)))
        return b

    def tx_perm(self):
        out = np.empty(self.info.n, np.uint32)
        check(self._lib.mmg_problem_tx_perm(self._h, _ptr(out)))
        return out

    def l(self):
        out = np.empty(self.info.n, np.float64)
        check(self._lib.mmg_problem_get_l(self._h, _ptr(out)))
        return out

    def start_values(self):
        """(mu0, unique_hits) of src/mmseq.cpp:617-638."""
        n = self.info.n
        mu0 = np.empty(n, np.float64)
        uh = np.empty(n, np.int32)
        check(self._lib.mmg_problem_start_values(self._h, _ptr(mu0), _ptr(uh)))
        return mu0, uh

    def em(self, mu, max_iter=1000, epsilon=0.1):
        """EM of src/mmseq.cpp:741-811. Returns (mu, iterations, loglik)."""
        mu = np.array(mu, np.float64, copy=True)
        it = C.c_int(0)
        ll = C.c_double(0.0)
        check(self._lib.mmg_problem_em(self._h, _ptr(mu), max_iter, epsilon, C.byref(it), C.byref(ll)))
        return mu, it.value, ll.value

    def em_stepper(self, mu0):
        """Sweep-by-sweep EM (mmg_em_*): the caller owns the loop, as at src/mmseq.cpp:761."""
        return Em(self, mu0)

    def close(self):
        if self._h:
            self._lib.mmg_problem_destroy(self._h)
            self._h = None

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass


class Em:
    """EM sweeps on the device (src/mmseq.cpp:741-811); `loglik` is the log-likelihood of the current mu."""

    def __init__(self, prob, mu0):
        self._lib = prob._lib
        self._prob = prob
        self._n = prob.info.n
        mu0 = np.ascontiguousarray(mu0, np.float64)
        assert mu0.size == self._n
        h = C.c_void_p()
        ll = C.c_double(0.0)
        check(self._lib.mmg_em_create(prob._h, _ptr(mu0), C.byref(h), C.byref(ll)))
        self._h = h
        self.loglik = ll.value

    def step(self):
        ll = C.c_double(0.0)
        check(self._lib.mmg_em_step(self._h, C.byref(ll)))
        self.loglik = ll.value
        return ll.value

    def mu(self):
        out = np.empty(self._n, np.float64)
        check(self._lib.mmg_em_get_mu(self._h, _ptr(out)))
        return out

    def stats_raw(self):
        """sweeps, repeated passes, rows-pass kernel id (2 sliced-ELL stream, 1 16-bit tile stream, 0 row per thread)."""
        a, b, c = C.c_int(0), C.c_int(0), C.c_int(0)
        check(self._lib.mmg_em_stats(self._h, C.byref(a), C.byref(b), C.byref(c)))
        return {"sweeps": a.value, "repeated_passes": b.value, "stream_kernel": c.value}

    def stats(self):
        d = self.stats_raw()
        d["stream_kernel"] = bool(d["stream_kernel"])
        return d

    def close(self):
        if self._h:
            self._lib.mmg_em_destroy(self._h)
            self._h = None

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass


class Sampler:
    """Chains of the Gibbs loop (src/mmseq.cpp:851-918) on one device."""

    def __init__(self, problem, mu0, alpha=0.1, beta=0.1, seed=1234, n_chains=1, chain_base=0,
                 gibbs_iter=1024, trace_len=1024, keep_trace=True, timing=False):
        self._lib = _lib.load()
        self.problem = problem
        self.n = problem.info.n
        self.n_chains = n_chains
        self.trace_len = trace_len
        mu0 = np.ascontiguousarray(mu0, np.float64)
        if mu0.size != self.n:
            raise ValueError("mu0 must have n entries")
        cfg = Config(alpha, beta, seed, n_chains, chain_base, gibbs_iter, trace_len, int(keep_trace), int(timing))
        h = C.c_void_p()
        check(self._lib.mmg_sampler_create(problem._h, C.byref(cfg), _ptr(mu0), C.byref(h)))
        self._h = h

    def set_stream(self, hip_stream):
        check(self._lib.mmg_sampler_set_stream(self._h, C.c_void_p(hip_stream) if hip_stream else None))

    def run(self, n_iter):
        check(self._lib.mmg_sampler_run(self._h, n_iter))

    def sample(self):
        check(self._lib.mmg_sampler_sample(self._h))

    def update(self):
        check(self._lib.mmg_sampler_update(self._h))

    def sync(self):
        check(self._lib.mmg_sampler_sync(self._h))

    def wait_iterations(self, n_done):
        """Block until the first n_done iterations are complete on the device (later ones may still be running)."""
        check(self._lib.mmg_sampler_wait_iterations(self._h, int(n_done)))

    @property
    def iteration(self):
        it = C.c_int(0)
        check(self._lib.mmg_sampler_iteration(self._h, C.byref(it)))
        return it.value

    def counts_devptr(self):
        p = C.c_void_p()
        cnt = C.c_uint64()
        check(self._lib.mmg_sampler_counts_devptr(self._h, C.byref(p), C.byref(cnt)))
        return p.value, cnt.value

    def moments_devptr(self):
        p = C.c_void_p()
        cnt = C.c_uint64()
        check(self._lib.mmg_sampler_moments_devptr(self._h, C.byref(p), C.byref(cnt)))
        return p.value, cnt.value

    def trace(self, chain=0):
        """Transcript-major trace [n, trace_len], the reference's mu_trace (src/mmseq.cpp:914)."""
        out = np.empty((self.n, self.trace_len), np.float64)
        check(self._lib.mmg_sampler_get_trace(self._h, chain, _ptr(out)))
        return out

    def trace_rows(self, chain=0, first=0, count=None):
        count = self.trace_len - first if count is None else count
        out = np.empty((count, self.n), np.float64)
        check(self._lib.mmg_sampler_get_trace_rows(self._h, chain, first, count, _ptr(out)))
        return out

    def trace_rows_done(self, chain=0, first=0, count=None):
        """rows of samples the device has finished (mmg_sampler_get_trace_rows_done: no wait for iterations enqueued behind them)"""
        count = self.trace_len - first if count is None else count
        out = np.empty((count, self.n), np.float64)
        check(self._lib.mmg_sampler_get_trace_rows_done(self._h, chain, first, count, _ptr(out)))
        return out

    def mu(self, chain=0):
        out = np.empty(self.n, np.float64)
        check(self._lib.mmg_sampler_get_mu(self._h, chain, _ptr(out)))
        return out

    def counts(self, chain=0):
        out = np.empty(self.n, np.int32)
        check(self._lib.mmg_sampler_get_counts(self._h, chain, _ptr(out)))
        return out

    def moments(self, chain=0):
        sl = np.empty(self.n, np.float64)
        sl2 = np.empty(self.n, np.float64)
        ns = C.c_int64(0)
        check(self._lib.mmg_sampler_get_moments(self._h, chain, _ptr(sl), _ptr(sl2), C.byref(ns)))
        return sl, sl2, ns.value

    def timing(self):
        t = Timing()
        check(self._lib.mmg_sampler_get_timing(self._h, C.byref(t)))
        return dict(sample_ms=t.sample_ms, update_ms=t.update_ms, sample_launches=t.sample_launches,
                    update_launches=t.update_launches)

    def reset_timing(self):
        check(self._lib.mmg_sampler_reset_timing(self._h))

    def close(self):
        if getattr(self, "_h", None):
            self._lib.mmg_sampler_destroy(self._h)
            self._h = None

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass


def shard_bounds(row_ptr, parts):
    """mmg_shard_bounds: first rows of `parts` contiguous shards of (nearly) equal hit counts, even boundaries."""
    row_ptr = np.ascontiguousarray(row_ptr, np.uint64)
    out = np.empty(parts + 1, np.uint64)
    check(_lib.load().mmg_shard_bounds(_ptr(row_ptr), row_ptr.size - 1, parts, _ptr(out)))
    return out


class Group:
    """Several GPUs of this node driven from one process over RCCL (mmg_group_*); samplers[i] lives on devices[i]."""

    def __init__(self, devices):
        self._lib = _lib.load()
        devs = (C.c_int * len(devices))(*devices)
        h = C.c_void_p()
        check(self._lib.mmg_group_create(devs, len(devices), C.byref(h)))
        self._h = h
        self.size = len(devices)

    def _arr(self, samplers):
        assert len(samplers) == self.size
        return (C.c_void_p * self.size)(*[s._h for s in samplers])

    def run_sharded(self, samplers, n_iter):
        check(self._lib.mmg_group_run_sharded(self._h, self._arr(samplers), n_iter))

    def run_chains(self, samplers, n_iter):
        check(self._lib.mmg_group_run_chains(self._h, self._arr(samplers), n_iter))

    def pool_moments(self, samplers):
        n = samplers[0].n
        sl, sl2, ns = np.empty(n), np.empty(n), C.c_int64(0)
        check(self._lib.mmg_group_pool_moments(self._h, self._arr(samplers), _ptr(sl), _ptr(sl2), C.byref(ns)))
        return sl, sl2, ns.value

    def em(self, shards, mu0, sweeps):
        """mmg_group_em_create + `sweeps` x mmg_em_step on the leader: EM over the read shards shards[i] (on device i), exchanged
        with RCCL.  Returns (mu, loglik)."""
        assert len(shards) == self.size
        arr = (C.c_void_p * self.size)(*[p._h for p in shards])
        ems = (C.c_void_p * self.size)()
        mu0 = np.ascontiguousarray(mu0, np.float64)
        ll = C.c_double(0.0)
        check(self._lib.mmg_group_em_create(self._h, arr, _ptr(mu0), ems, C.byref(ll)))
        try:
            for _ in range(sweeps):
                check(self._lib.mmg_em_step(ems[0], C.byref(ll)))
            mu = np.empty_like(mu0)
            check(self._lib.mmg_em_get_mu(ems[0], _ptr(mu)))
        finally:
            for e in ems:
                self._lib.mmg_em_destroy(e)
        return mu, ll.value

    def enqueue_us(self):
        us = C.c_double(0.0)
        check(self._lib.mmg_group_enqueue_us(self._h, C.byref(us)))
        return us.value

    def close(self):
        if getattr(self, "_h", None):
            self._lib.mmg_group_destroy(self._h)
            self._h = None

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass


SERIES_TRANSCRIPT, SERIES_VIRTUAL, SERIES_IDENTICAL, SERIES_GENE = range(4)


def _csr(groups):
    ptr = np.zeros(len(groups) + 1, np.uint64)
    ptr[1:] = np.cumsum([len(g) for g in groups])
    mem = np.array([m for g in groups for m in g], np.uint32)
    return ptr, mem


class Summary:
    """Posterior summary of a sampler's resident trace on the device (mmg_summary_*; src/mmseq.cpp:927-1363).
    identical / genes: lists of member lists (a member < n: transcript, n + v: virtual transcript v)."""

    def __init__(self, sampler, chain=0, virtual_id=(), virtual_scale=(), identical=(), genes=(), percentile_index=(), staged=False):
        """staged: mmg_summary_begin only -- the caller then feeds finished samples with advance() while the chain runs and calls
        finish() after the last one (rows() serves the advanced samples at any time)."""
        from ._lib import SummaryDesc
        self._lib = _lib.load()
        self.n = sampler.n
        self.S = sampler.trace_len
        vid = np.ascontiguousarray(virtual_id, np.uint64)
        vsc = np.ascontiguousarray(virtual_scale, np.float64)
        iptr, imem = _csr(identical)
        gptr, gmem = _csr(genes)
        pidx = np.ascontiguousarray(percentile_index, np.int32)
        self.counts = [self.n, vid.size, len(identical), len(genes)]
        self.np_ = pidx.size
        d = SummaryDesc(chain, vid.size, _ptr(vid), _ptr(vsc), len(identical), _ptr(iptr), _ptr(imem), len(genes), _ptr(gptr), _ptr(gmem),
                        pidx.size, _ptr(pidx))
        h = C.c_void_p()
        check((self._lib.mmg_summary_begin if staged else self._lib.mmg_summary_create)(sampler._h, C.byref(d), C.byref(h)))
        self._h = h

    def advance(self, samples_done):
        check(self._lib.mmg_summary_advance(self._h, int(samples_done)))

    def finish(self):
        check(self._lib.mmg_summary_finish(self._h))

    def series(self, kind):
        c = self.counts[kind]
        lm, var, tau = np.empty(c), np.empty(c), np.empty(c)
        rc = np.empty(c, np.int32)
        pct = np.empty((c, self.np_))
        check(self._lib.mmg_summary_get(self._h, kind, _ptr(lm), _ptr(var), _ptr(tau), _ptr(rc), _ptr(pct)))
        return dict(log_mean=lm, var=var, tau=tau, rc=rc, percentiles=pct)

    def proportions(self, kind):
        c = self.counts[kind]
        mp, pm, ps = np.empty(c), np.empty(c), np.empty(c)
        pct = np.empty((c, self.np_))
        check(self._lib.mmg_summary_get_proportions(self._h, kind, _ptr(mp), _ptr(pm), _ptr(ps), _ptr(pct)))
        return dict(mean=mp, probit_mean=pm, probit_sd=ps, percentiles=pct)

    def rows(self, kind, first=0, count=None):
        count = self.S - first if count is None else count
        out = np.empty((count, self.counts[kind]))
        check(self._lib.mmg_summary_get_rows(self._h, kind, first, count, _ptr(out)))
        return out

    def close(self):
        if getattr(self, "_h", None):
            self._lib.mmg_summary_destroy(self._h)
            self._h = None

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass


def em_shards_selftest(shards, mu0, sweeps):
    """mmg_selftest_em_shards: the sharded EM of mmg_group_em_create with all shards on one device (exchange by kernels).
    Returns (mu, loglik, repeated_passes)."""
    lib = _lib.load()
    arr = (C.c_void_p * len(shards))(*[p._h for p in shards])
    mu0 = np.ascontiguousarray(mu0, np.float64)
    mu = np.empty_like(mu0)
    ll = C.c_double(0.0)
    rep = C.c_int(0)
    check(lib.mmg_selftest_em_shards(arr, len(shards), _ptr(mu0), int(sweeps), _ptr(mu), C.byref(ll), C.byref(rep)))
    return mu, ll.value, rep.value


def gibbs_shards_selftest(samplers, n_iter):
    """mmg_selftest_gibbs_shards: the sharded chain of mmg_group_run_sharded with all shards on one device (count exchange by a
    kernel); every sampler ends up holding the chain of the unsharded problem."""
    arr = (C.c_void_p * len(samplers))(*[s._h for s in samplers])
    check(_lib.load().mmg_selftest_gibbs_shards(arr, len(samplers), int(n_iter)))


# ---- self-test hooks -----------------------------------------------------------------------
def selftest_option(option, value):
    """Process-wide override of a choice the library normally makes itself (tests only); value < 0 restores the default."""
    check(_lib.load().mmg_selftest_option(int(option), int(value)))


OPT = dict(sample_kernel=_lib.OPT_SAMPLE_KERNEL, force_idx64=_lib.OPT_FORCE_IDX64, sell_waves_per_cu=_lib.OPT_SELL_WAVES_PER_CU,
           em_kernel=_lib.OPT_EM_KERNEL, em_grid=_lib.OPT_EM_GRID, fuse_chains=_lib.OPT_FUSE_CHAINS, cnt_replicas=_lib.OPT_CNT_REPLICAS, group_fail=_lib.OPT_GROUP_FAIL, derive_order=_lib.OPT_DERIVE_ORDER,
           wire_check=_lib.OPT_WIRE_CHECK, bigk_per_wave=_lib.OPT_BIGK_PER_WAVE, bigk_side_stream=_lib.OPT_BIGK_SIDE_STREAM)


class options:
    """with gibbs.options(sample_kernel=0, em_grid=7): ...   -- restores the defaults on exit."""

    def __init__(self, **kw):
        self.kw = kw

    def __enter__(self):
        for k, v in self.kw.items():
            selftest_option(OPT[k], v)
        return self

    def __exit__(self, *a):
        for k in self.kw:
            selftest_option(OPT[k], -1)
        return False



def selftest_math(x, device):
    x = np.ascontiguousarray(x, np.float64)
    outs = [np.empty_like(x) for _ in range(4)]
    check(_lib.load().mmg_selftest_math(device, x.size, _ptr(x), *[_ptr(o) for o in outs]))
    return dict(log=outs[0], exp=outs[1], sqrt=outs[2], rcp=outs[3])


def selftest_philox(ctr, key, device):
    ctr = np.ascontiguousarray(ctr, np.uint32)
    key = np.ascontiguousarray(key, np.uint32)
    out = np.zeros(6, np.uint32)
    check(_lib.load().mmg_selftest_philox(device, _ptr(ctr), _ptr(key), _ptr(out)))
    return out


def selftest_gamma(seed, shape, scale, n, device):
    out = np.empty(n, np.float64)
    check(_lib.load().mmg_selftest_gamma(device, seed, shape, scale, n, _ptr(out)))
    return out


def selftest_binomial(seed, nn, p, n, device):
    out = np.empty(n, np.uint32)
    check(_lib.load().mmg_selftest_binomial(device, seed, nn, p, n, _ptr(out)))
    return out


def selftest_btrs_pretest(seed, n_cases, n_lo, n_hi, device=0):
    """(reached the exact test, decided by the fp32 estimate, decided and WRONG, accepted, largest error / bound in millionths) over n_cases BTRS
    attempts (mmg_math.h: btrs_pretest)"""
    out = np.zeros(5, np.uint64)
    check(_lib.load().mmg_selftest_btrs_pretest(device, seed, n_cases, float(n_lo), float(n_hi), _ptr(out)))
    return tuple(int(v) for v in out)


def selftest_binv_pretest(seed, n_cases, n_lo, n_hi, slack=1.0, device=0):
    """(cases, decided by the fp32 search, decided and WRONG, fp64 searches that ran off the end, sum of the outcomes) over n_cases inversions
    (mmg_math.h: binv_pretest); slack scales the error bound (1: the sampler's)"""
    out = np.zeros(5, np.uint64)
    check(_lib.load().mmg_selftest_binv_pretest(device, seed, n_cases, float(n_lo), float(n_hi), float(slack), _ptr(out)))
    return tuple(int(v) for v in out)
