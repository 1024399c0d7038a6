"""Multi-GPU from the C++ side (mmg_group_*, RCCL behind the C ABI).  CPU: the shard arithmetic (mmg_shard_bounds) at world
sizes 2..8 and the loud failure without a device.  GPU: the native test binary (tests/native/test_group.cpp, no Python or torch
in the process) -- read-shard chain over a group bit-identical to the unsharded chain, pooled moments -- plus the same entry
points through ctypes."""
import os
import subprocess

import numpy as np
import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
BIN = os.path.join(ROOT, "mmseq_amd", "csrc", "test_group")


@pytest.mark.parametrize("world", [1, 2, 3, 8])
def test_shard_bounds_cover_rows_once_and_balance_hits(orc, world):
    from mmseq_amd import gibbs
    from mmseq_amd import dist as mdist
    p, _ = orc.synth_problem(R=50000, T=3000, avg_hits=8, seed=2)
    b = gibbs.shard_bounds(p.row_ptr, world)
    assert b[0] == 0 and b[-1] == p.m and (np.diff(b.astype(np.int64)) >= 0).all()
    assert (b[:-1] % 2 == 0).all()                       # shards start on even rows: a Philox block (rows 2q, 2q+1) stays on one device
    hits = np.diff(p.row_ptr[b.astype(np.int64)].astype(np.int64))
    assert abs(hits / p.nnz - 1.0 / world).max() < 0.01
    # the sharded count vectors add up to the unsharded one (the all-reduce), with the oracle standing in for the devices
    mu0, _ = orc.start_values(p)
    ref = orc.sample_counts(p, mu0, 7, 0, 3)
    tot = np.zeros_like(ref)
    for i in range(world):
        lo, hi = int(b[i]), int(b[i + 1])
        nz0, nz1 = int(p.row_ptr[lo]), int(p.row_ptr[hi])
        q = orc.Problem(p.row_ptr[lo:hi + 1] - p.row_ptr[lo], p.col_idx[nz0:nz1], p.l)
        tot += orc.sample_counts(q, mu0, 7, 0, 3, row_id_base=lo)
    assert np.array_equal(tot, ref)
    # the Python mirror's plain row split covers the rows once as well
    cover = [mdist.row_shard(p.m, r, world) for r in range(world)]
    assert cover[0][0] == 0 and cover[-1][1] == p.m and all(a[1] == c[0] for a, c in zip(cover, cover[1:]))


def test_group_needs_a_device():
    from mmseq_amd import gibbs
    from mmseq_amd._lib import MMGError
    if gibbs.device_count() > 0:
        pytest.skip("a HIP device is present")
    with pytest.raises(MMGError) as e:
        gibbs.Group([0])
    assert e.value.code == 2


@pytest.mark.gpu
def test_native_group_binary(gpu):
    assert os.path.exists(BIN), "test_group not built (make -C mmseq_amd/csrc)"
    r = subprocess.run([BIN], stdout=subprocess.PIPE, stderr=subprocess.PIPE, timeout=600)
    assert r.returncode == 0, r.stdout.decode() + r.stderr.decode()
    assert b"read-shard chain bit-identical to the unsharded chain" in r.stdout


@pytest.mark.gpu
def test_group_through_ctypes(gpu, orc):
    p, _ = orc.synth_problem(R=30000, T=1200, avg_hits=6, seed=4)
    mu0, _ = orc.start_values(p)
    prob = gpu.Problem.from_csr(p.row_ptr, p.col_idx, p.l)
    grp = gpu.Group([0])
    s = gpu.Sampler(prob, mu0, seed=5, n_chains=2, gibbs_iter=16, trace_len=16)
    grp.run_sharded([s], 16)
    for c in range(2):
        ref = orc.gibbs_keyed(p, mu0, seed=5, chain=c, n_iter=16, trace_len=16)
        assert np.array_equal(s.trace(c), ref["trace"])
    sl, sl2, ns = grp.pool_moments([s])
    a0, b0, _ = s.moments(0)
    a1, b1, _ = s.moments(1)
    assert ns == 32 and np.array_equal(sl, a0 + a1) and np.array_equal(sl2, b0 + b1)
    sl_again, _, ns_again = grp.pool_moments([s])                 # into scratch buffers: pooling twice counts nothing twice
    assert ns_again == 32 and np.array_equal(sl_again, sl)
    assert grp.enqueue_us() > 0.0                                 # host time per device-iteration of the last run call
    # EM over the group's (one) read shard = the problem's own EM, bit for bit (N shards: tests/test_gpu_parity.py, same-device exchange)
    mu_g, ll_g = grp.em([prob], mu0, 5)
    mu_o, _, ll_o = orc.em(p, mu0, max_iter=5, epsilon=-1e308)
    assert np.array_equal(mu_g, mu_o) and ll_g == ll_o
    grp.close()


@pytest.mark.gpu
def test_group_failure_aborts_the_communicators_and_long_chains_recycle_their_timing_events(gpu, orc):
    """A member that fails inside a run call (injected: MMG_OPT_GROUP_FAIL) stops the others, the communicators are aborted
    (ncclCommAbort: a peer already inside the iteration's all-reduce must not wait for ever -- src/mmseq.cpp:278-296 fails early and
    loudly too), the caller gets the member's message, and the group refuses further work instead of hanging.  Also: a sampler timed
    on every iteration harvests its HIP events as they complete -- thousands of timed iterations, a bounded pool, the same sums."""
    from mmseq_amd._lib import MMGError
    p, _ = orc.synth_problem(R=20000, T=900, avg_hits=5, seed=9)
    mu0, _ = orc.start_values(p)
    prob = gpu.Problem.from_csr(p.row_ptr, p.col_idx, p.l)
    grp = gpu.Group([0])
    s = gpu.Sampler(prob, mu0, seed=5, gibbs_iter=4096, trace_len=1, keep_trace=False, timing=1)
    grp.run_sharded([s], 3000)
    s.sync()
    t = s.timing()
    assert t["sample_launches"] == 3000 and t["update_launches"] == 3000 and t["sample_ms"] > 0.0
    with gpu.options(group_fail=0):
        with pytest.raises(MMGError) as e:
            grp.run_sharded([s], 8)
    assert "injected failure" in str(e.value) and "aborted" in str(e.value)
    s.sync()                                                     # nothing is left waiting on the device
    with pytest.raises(MMGError) as e2:
        grp.run_chains([s], 1)
    assert e2.value.code == 4 and "aborted" in str(e2.value)
    grp.close(); s.close(); prob.close()


@pytest.mark.gpu
def test_a_groups_first_exchanges_are_verified_against_the_hosts_own_reduction(gpu, orc):
    """The wire check of mmg_group_* (group.hip): the first count all-reduce of a sharded chain and the first EM exchanges of every
    kind are compared, on every device, with the reduction of the members' buffers computed on the host from downloads.  No N > 1
    RCCL run of this code exists (one-GPU boxes): the check is what turns a misbehaving transport on the first multi-GPU node into an
    error instead of a wrong table.  Here it is forced in a group of ONE device (MMG_OPT_WIRE_CHECK = 1): same results as without,
    and with a word damaged behind the exchange (= 2) both paths fail loudly, abort the group and name the device."""
    from mmseq_amd._lib import MMGError
    p, _ = orc.synth_problem(R=30000, T=1200, avg_hits=6, seed=4)
    mu0, _ = orc.start_values(p)
    prob = gpu.Problem.from_csr(p.row_ptr, p.col_idx, p.l)
    ref = orc.gibbs_keyed(p, mu0, seed=5, chain=0, n_iter=8, trace_len=8)
    mu_o, _, ll_o = orc.em(p, mu0, max_iter=5, epsilon=-1e308)
    with gpu.options(wire_check=1):
        grp = gpu.Group([0])
        s = gpu.Sampler(prob, mu0, seed=5, gibbs_iter=8, trace_len=8)
        grp.run_sharded([s], 3)                                   # first iteration step by step with the check, two by the driver threads
        grp.run_sharded([s], 5)
        assert np.array_equal(s.trace(0), ref["trace"])
        mu_g, ll_g = grp.em([prob], mu0, 5)                       # every exchange of the EM (column counts, exponents, accumulators) checked once
        assert np.array_equal(mu_g, mu_o) and ll_g == ll_o
        s.close(); grp.close()
        # empty rows (allowed: row_ptr only non-decreasing) carry reads that no sweep allocates -- the check's "every read counted once"
        # clause counts the reads of the rows WITH hits -- next to multiplicities on both sides of the chain's boundary
        rng = np.random.default_rng(3)
        cut = np.sort(rng.choice(np.arange(1, p.m), size=40, replace=False))
        rp_e = np.insert(p.row_ptr, cut, p.row_ptr[cut])
        k_e = rng.choice([1, 1, 2, 7, 70, 3000], size=rp_e.size - 1).astype(np.uint32)
        prob_e = gpu.Problem.from_csr(rp_e, p.col_idx, p.l, k=k_e)
        assert prob_e.info.total_k == int(k_e.astype(np.int64).sum())
        d_rp, d_ci, d_k = prob_e.download(with_k=True)
        pe = orc.Problem(d_rp, d_ci, p.l, k=d_k)
        ref_e = orc.gibbs_keyed(pe, mu0, seed=6, chain=0, n_iter=4, trace_len=4)
        grp = gpu.Group([0])
        se = gpu.Sampler(prob_e, mu0, seed=6, gibbs_iter=4, trace_len=4)
        grp.run_sharded([se], 4)
        assert np.array_equal(se.trace(0), ref_e["trace"]) and int(se.counts(0).astype(np.int64).sum()) < prob_e.info.total_k
        se.close(); grp.close(); prob_e.close()
    with gpu.options(wire_check=2):
        grp = gpu.Group([0])
        s = gpu.Sampler(prob, mu0, seed=5, gibbs_iter=8, trace_len=8)
        with pytest.raises(MMGError) as e:
            grp.run_sharded([s], 3)
        assert "wire check" in str(e.value) and "device 0" in str(e.value) and e.value.code == 4
        with pytest.raises(MMGError) as e2:
            grp.run_sharded([s], 1)
        assert "aborted" in str(e2.value)
        s.close(); grp.close()
        grp = gpu.Group([0])
        with pytest.raises(MMGError) as e3:
            grp.em([prob], mu0, 2)
        assert "wire check" in str(e3.value) and "accumulators" in str(e3.value)
        grp.close()
    prob.close()


@pytest.mark.gpu
def test_em_over_two_devices_equals_the_em_of_one(gpu, orc):
    """The EM twin of tests/test_cli.py::test_two_devices_give_the_output_of_one: mmg_group_em_create over read shards on TWO devices
    (peer copies of the shards, ncclMax / ncclSum(uint64) between the phases) against mmg_problem_em on one, bit for bit.  Skipped on
    the one-GPU boxes of the pool; the first box with two devices turns it into evidence."""
    if gpu.device_count() < 2:
        pytest.skip("needs two HIP devices")
    p, _ = orc.synth_problem(R=200_000, T=4000, avg_hits=6, seed=14, far_fraction=0.1)
    mu0, _ = orc.start_values(p)
    prob = gpu.Problem.from_csr(p.row_ptr, p.col_idx, p.l)
    mu_1, it_1, ll_1 = prob.em(mu0, max_iter=40, epsilon=-1e308)
    b = prob.shard_bounds_timed(mu0, 2)
    parts = [prob.shard(int(b[i]), int(b[i + 1]), device=i) for i in range(2)]
    grp = gpu.Group([0, 1])
    mu_g, ll_g = grp.em(parts, mu0, 40)
    assert np.array_equal(mu_g, mu_1) and ll_g == ll_1
    s = [gpu.Sampler(parts[i], mu_1, seed=3, gibbs_iter=32, trace_len=32) for i in range(2)]
    one = gpu.Sampler(prob, mu_1, seed=3, gibbs_iter=32, trace_len=32)
    grp.run_sharded(s, 32)
    one.run(32)
    assert np.array_equal(s[0].trace(0), one.trace(0)) and np.array_equal(s[1].trace(0), one.trace(0))
    for x in s + [one]:
        x.close()
    grp.close()
    for q in parts:
        q.close()
    prob.close()
