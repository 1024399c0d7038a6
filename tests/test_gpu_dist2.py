"""mmseq_amd/dist.py at world size 2 against the REAL library (libmmgibbs kernels, zero-copy torch views of its device buffers,
kernels and collective ordered on torch's stream).  The GPU box has one GPU, which RCCL refuses to give to two ranks, so the two
ranks share cuda:0 and the process group is gloo (it all-reduces CUDA tensors through the host): the protocol -- shard_step,
pool_moments, counts_tensor / moments_tensor views -- is exactly what runs over RCCL on 8 GPUs, only the transport differs.
Children are fresh processes spawned before any GPU call."""
import os
import sys

import numpy as np
import pytest

pytestmark = pytest.mark.gpu
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _child(rank, world, store, q):
    sys.path.insert(0, ROOT)
    os.environ["GLOO_SOCKET_IFNAME"] = "lo"          # (the container's hostname may not resolve)
    import torch
    import torch.distributed as dist
    from mmseq_amd import gibbs as G
    from mmseq_amd import dist as mdist
    from oracle import binding as B
    import datetime
    # rendezvous through a FILE (no port to pick: a "free" port chosen by the parent can be taken, on a host that runs other jobs, before
    # the ranks bind it -- two sessions of this suite hung here while the pool was full); a stalled rendezvous fails, it does not hang
    dist.init_process_group("gloo", init_method="file://" + store, rank=rank, world_size=world, timeout=datetime.timedelta(seconds=180))
    torch.cuda.set_device(0)
    p, _ = B.synth_problem(R=60000, T=1500, avg_hits=6, seed=21)            # canonical order: shards are cut from it and kept
    mu0, _ = B.start_values(p)
    n_iter = 12
    # ---- shard mode: rows [lo, hi) of the stored order on this rank, row_id_base = lo
    b = G.shard_bounds(p.row_ptr, world)
    lo, hi = int(b[rank]), int(b[rank + 1])
    nz0, nz1 = int(p.row_ptr[lo]), int(p.row_ptr[hi])
    prob = G.Problem.from_csr(p.row_ptr[lo:hi + 1] - p.row_ptr[lo], p.col_idx[nz0:nz1], p.l, row_id_base=lo, keep_rows=True)
    s = G.Sampler(prob, mu0, seed=77, gibbs_iter=n_iter, trace_len=n_iter)
    mdist.use_current_stream(s)
    counts = mdist.counts_tensor(s)
    for _ in range(n_iter):
        mdist.shard_step(s, counts)
    torch.cuda.synchronize()
    mu_sh, tr_sh = s.mu(0), s.trace(0)
    cnt_sh = s.counts(0)
    # ---- chains mode: the full problem on every rank, chain = rank; one all-reduce of the moments
    full = G.Problem.from_csr(p.row_ptr, p.col_idx, p.l)
    c = G.Sampler(full, mu0, seed=77, chain_base=rank, gibbs_iter=16, trace_len=16, keep_trace=False)
    mdist.use_current_stream(c)
    c.run(16)
    own = np.concatenate(c.moments(0)[:2])
    mom = mdist.moments_tensor(c)
    mdist.pool_moments(mom)
    torch.cuda.synchronize()
    q.put((rank, mu_sh.tobytes(), tr_sh.tobytes(), cnt_sh.tobytes(), own.tobytes(), mom.cpu().numpy().tobytes()))
    dist.destroy_process_group()


def test_two_ranks_drive_the_library_through_dist_py(orc):
    import torch.multiprocessing as mp
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    import tempfile
    store = os.path.join(tempfile.mkdtemp(prefix="mmseq_dist2_"), "store")
    procs = [ctx.Process(target=_child, args=(r, 2, store, q)) for r in range(2)]
    for pr in procs:
        pr.start()
    try:
        res = sorted(q.get(timeout=400) for _ in range(2))   # (seconds on a quiet box; the children import torch on a host shared with three other jobs)
        for pr in procs:
            pr.join(120)
            assert pr.exitcode == 0
    finally:
        for pr in procs:                                     # a child that is still there must not keep the test session from ending
            if pr.is_alive():
                pr.terminate()
                pr.join(10)
    p, _ = orc.synth_problem(R=60000, T=1500, avg_hits=6, seed=21)
    mu0, _ = orc.start_values(p)
    ref = orc.gibbs_keyed(p, mu0, seed=77, chain=0, n_iter=12, trace_len=12)
    for rank, mu_b, tr_b, cnt_b, own_b, mom_b in res:
        assert np.array_equal(np.frombuffer(mu_b), ref["mu"])                          # every rank holds the unsharded chain, bit for bit
        assert np.array_equal(np.frombuffer(tr_b).reshape(ref["trace"].shape), ref["trace"])
        assert np.array_equal(np.frombuffer(cnt_b, np.int32), ref["cnt"])
    pooled = np.frombuffer(res[0][4]) + np.frombuffer(res[1][4])
    for rank, *_, own_b, mom_b in res:
        c = orc.gibbs_keyed(p, mu0, seed=77, chain=rank, n_iter=16, trace_len=16, want_trace=False)
        assert np.array_equal(np.frombuffer(own_b), np.concatenate([c["sum_log"], c["sum_log2"]]))
        assert np.array_equal(np.frombuffer(mom_b), pooled)                            # two terms: the sum is exact in either order
