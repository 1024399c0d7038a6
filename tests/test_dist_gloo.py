"""world_size-2 gloo tests of the two multi-GPU protocols of mmseq_amd/dist.py, with the CPU oracle
standing in for the kernels (the collectives, the sharding and the keyed-stream argument are what is
under test here; the kernels themselves are covered by the -m gpu parity tests).

  shard mode : rank r owns rows [lo_r, hi_r); per iteration  local counts -> all_reduce(SUM, int32)
               -> identical Gamma update on every rank.  Must equal the 1-process chain BIT FOR BIT.
  chains mode: rank r runs chain r; one all_reduce of (sum log mu, sum log^2 mu) at the end.
"""
import os
import sys

import numpy as np
import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _worker(rank, world, store, q):
    sys.path.insert(0, ROOT)
    os.environ["GLOO_SOCKET_IFNAME"] = "lo"          # (the container's hostname may not resolve)
    os.environ["OMP_NUM_THREADS"] = "2"
    import torch
    import torch.distributed as dist
    from mmseq_amd import dist as mdist
    from oracle import binding as B
    import datetime
    # (rendezvous through a file: no port to pick and lose to another job on the host; a stalled rendezvous fails instead of hanging)
    dist.init_process_group("gloo", init_method="file://" + store, rank=rank, world_size=world, timeout=datetime.timedelta(seconds=120))
    p, _ = B.synth_problem(R=6000, T=250, avg_hits=5, seed=21)
    mu0, _ = B.start_values(p)
    n_iter = 6
    # ---- shard mode
    lo, hi = mdist.row_shard(p.m, rank, world)
    nz0, nz1 = int(p.row_ptr[lo]), int(p.row_ptr[hi])
    shard = B.Problem(p.row_ptr[lo:hi + 1] - p.row_ptr[lo], p.col_idx[nz0:nz1], p.l)
    mu = mu0.copy()
    for it in range(n_iter):
        cnt = B.sample_counts(shard, mu, 77, 0, it, row_id_base=lo)        # K1 on the local shard
        t = torch.from_numpy(cnt)
        dist.all_reduce(t, op=dist.ReduceOp.SUM)                            # the per-iteration collective
        mu = B.gamma_update(t.numpy(), p.l, 0.1, 0.1, 77, 0, it)            # K2, identical on every rank
    # ---- chains mode
    r = B.gibbs_keyed(p, mu0, seed=77, chain=rank, n_iter=16, trace_len=16, want_trace=False)
    mom = torch.from_numpy(np.concatenate([r["sum_log"], r["sum_log2"]]))
    mdist.pool_moments(mom)
    q.put((rank, mu.tobytes(), t.numpy().tobytes(), mom.numpy().tobytes()))
    dist.destroy_process_group()


def test_two_rank_protocols_reproduce_single_process():
    import torch.multiprocessing as mp
    from oracle import binding as B
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    import tempfile
    store = os.path.join(tempfile.mkdtemp(prefix="mmseq_gloo_"), "store")
    procs = [ctx.Process(target=_worker, args=(r, 2, store, q)) for r in range(2)]
    for pr in procs:
        pr.start()
    try:
        res = sorted(q.get(timeout=300) for _ in range(2))
        for pr in procs:
            pr.join(60)
            assert pr.exitcode == 0
    finally:
        for pr in procs:
            if pr.is_alive():
                pr.terminate()
                pr.join(10)
    p, _ = B.synth_problem(R=6000, T=250, avg_hits=5, seed=21)
    mu0, _ = B.start_values(p)
    ref = B.gibbs_keyed(p, mu0, seed=77, chain=0, n_iter=6, trace_len=6)
    for rank, mu_b, cnt_b, mom_b in res:
        assert np.array_equal(np.frombuffer(mu_b), ref["mu"])               # sharded == unsharded, bit for bit
        assert np.array_equal(np.frombuffer(cnt_b, np.int32), ref["cnt"])
    c0 = B.gibbs_keyed(p, mu0, seed=77, chain=0, n_iter=16, trace_len=16, want_trace=False)
    c1 = B.gibbs_keyed(p, mu0, seed=77, chain=1, n_iter=16, trace_len=16, want_trace=False)
    pooled = np.concatenate([c0["sum_log"] + c1["sum_log"], c0["sum_log2"] + c1["sum_log2"]])
    for rank, _, _, mom_b in res:
        assert np.allclose(np.frombuffer(mom_b), pooled, rtol=1e-14)
    assert not np.array_equal(c0["sum_log"], c1["sum_log"])                 # chains really are independent


def test_row_shard_partitions_exactly():
    from mmseq_amd.dist import row_shard
    for total, world in ((10, 3), (400_000_000, 8), (7, 8), (0, 2)):
        spans = [row_shard(total, r, world) for r in range(world)]
        assert spans[0][0] == 0 and spans[-1][1] == total
        assert all(spans[i][1] == spans[i + 1][0] for i in range(world - 1))
        sizes = [b - a for a, b in spans]
        assert max(sizes) - min(sizes) <= 1


def test_pooled_summary():
    from mmseq_amd.dist import pooled_summary
    x = np.random.default_rng(0).normal(2.0, 0.5, (3, 4000))
    mean, sd = pooled_summary(x.sum(axis=1), (x * x).sum(axis=1), 4000)
    assert np.allclose(mean, x.mean(axis=1)) and np.allclose(sd, x.std(axis=1, ddof=1))
