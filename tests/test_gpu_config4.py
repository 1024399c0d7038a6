"""BASELINE.json configs[4] at full size on ONE MI355X: the read-sharded single chain over 400 M reads x 200 k transcripts (8.0 G hits:
32 GB of column ids, 64-bit row offsets, an 8.7 GB tile stream), built on one device, cut into the 8 shards `mmseq -gpus 8` would
place on 8 devices (mmg_problem_shard_bounds + mmg_problem_shard) and run as the sharded chain with the int32 count exchange done by a
kernel where the group runs ncclAllReduce (mmg_selftest_gibbs_shards).  No 8-GPU node exists on the pool; this is the arithmetic of
src/mmseq.cpp:864 (the static split of the rows) and :893-899 (the reduction) at the real size, everything but the wire.

Checks: (a) every shard's first sweep against the CPU oracle on the shard's downloaded rows (50 M rows at a time: host memory stays
below 6 GB), bit for bit; (b) the shards' counts add up to the unsharded sweep's; (c) two sweeps of the sharded chain leave trace,
counts and mu bit-identical to the unsharded chain's on the 400 M-row problem; (d) every read is assigned exactly once per sweep;
(e) the cut is balanced in modelled cost and starts every shard on an even row."""
import numpy as np
import pytest

pytestmark = pytest.mark.gpu

R4, T4, AVG4, PARTS = 400_000_000, 200_000, 20.0, 8


def test_config4_read_sharded_chain_at_full_size_on_one_device(gpu, orc):
    prob = gpu.Problem.synthetic(R4, T4, AVG4, seed=1234, sort=True, mapped_reads=R4)
    inf = prob.info
    assert inf.m == R4 and inf.nnz > 7_900_000_000 and inf.index_bits == 64 and inf.sample_kernel == 2 and inf.layout == 0
    assert inf.fast_tiles == inf.n_tiles
    mu0, _ = prob.start_values()
    b = prob.shard_bounds(PARTS)
    assert b[0] == 0 and b[-1] == R4 and np.all(b[:-1] % 2 == 0)
    rows = np.diff(b.astype(np.int64))
    assert rows.min() > 0 and rows.max() <= 1.02 * rows.min(), rows   # near rows only, one length distribution in every band: equal cost = equal rows
    shards = [prob.shard(int(b[i]), int(b[i + 1])) for i in range(PARTS)]
    for i, sh in enumerate(shards):
        si = sh.info
        assert si.row_id_base == int(b[i]) and si.m == rows[i] and si.sample_kernel == 2 and si.index_bits == 32 and si.fast_tiles == si.n_tiles
    assert sum(sh.info.nnz for sh in shards) == inf.nnz

    # (a) + (b): the first sweep of every shard against the oracle, and the sum against the unsharded sweep
    whole = gpu.Sampler(prob, mu0, seed=1234, gibbs_iter=2, trace_len=2)
    whole.sample()
    cnt_whole = whole.counts(0).astype(np.int64)
    assert int(cnt_whole.sum()) == R4
    total = np.zeros(T4, np.int64)
    for i, sh in enumerate(shards):
        s = gpu.Sampler(sh, mu0, seed=1234, gibbs_iter=2, trace_len=2, keep_trace=False)
        s.sample()
        got = s.counts(0)
        s.close()
        rp, ci = sh.download()
        want = orc.gibbs_keyed(orc.Problem(rp, ci, prob.l()), mu0, seed=1234, n_iter=1, trace_len=1, row_id_base=int(b[i]), want_trace=False)["cnt"]
        del rp, ci
        assert np.array_equal(got, want), "shard %d: first sweep differs from the oracle" % i
        total += got
    assert np.array_equal(total, cnt_whole)

    # (c) + (d): two sweeps sharded against two sweeps unsharded
    whole.update()
    whole.run(1)
    smps = [gpu.Sampler(sh, mu0, seed=1234, gibbs_iter=2, trace_len=2, keep_trace=(i == PARTS - 1)) for i, sh in enumerate(shards)]
    gpu.gibbs_shards_selftest(smps, 2)
    tr = whole.trace(0)
    assert np.array_equal(smps[-1].trace(0), tr)
    cw, mw = whole.counts(0), whole.mu(0)
    assert int(cw.astype(np.int64).sum()) == R4
    for sm in smps:
        assert np.array_equal(sm.counts(0), cw) and np.array_equal(sm.mu(0), mw)
        sm.close()
    whole.close()
    for sh in shards:
        sh.close()
    prob.close()
