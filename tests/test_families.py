"""tools/families.py (host side, numpy): the benchmark's hit graph with a real tail -- paralogue families of power-law size whose
reads also hit a NEIGHBOUR in the family, hub transcripts (src/bam2hits.cpp:271-300: a read keeps up to 100 alignments).  Properties
the GPU tests and bench.py's `families_pl` rely on."""
import numpy as np

import os
import sys

sys.path.insert(0, os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "tools"))
import families as fam  # noqa: E402


def _base(m, T, G, seed):
    rng = np.random.default_rng(seed)
    g = rng.integers(0, T // G, size=m)
    L = rng.integers(1, 7, size=m)
    rp = np.concatenate(([0], np.cumsum(L))).astype(np.uint64)
    col = np.concatenate([np.sort(rng.choice(G, size=l, replace=False)) + gg * G for gg, l in zip(g, L)]).astype(np.uint32)
    return rp, col


def test_family_sizes_follow_a_power_law_and_partition_the_genes():
    order, start, size, fam_of, pos_of = fam.family_tables(200_000, 32, seed=7)
    assert np.array_equal(np.sort(order), np.arange(200_000 // 32)) and int(size.sum()) == order.size
    assert size.min() == 1 and size.max() == 5000 // 32 and (size == 1).mean() > 0.4      # most genes stand alone, a few families are huge
    assert int(size[size >= 16].sum()) > 0.2 * order.size                                # ... and hold a good share of the genes
    g = order[start[5] + np.arange(size[5])]
    assert (fam_of[g] == 5).all() and np.array_equal(pos_of[g], np.arange(size[5]))


def test_rows_keep_their_hits_and_gain_neighbours_and_hubs():
    T, G = 40_000, 16
    rp, col = _base(30_000, T, G, seed=3)
    rp2, col2, tx, info = fam.power_law_families(rp, col, T, G, seed=11, n_hubs=5, chunk_rows=7_000)
    _, col3, _, _ = fam.power_law_families(rp, col, T, G, seed=11, n_hubs=5, chunk_rows=7_000)
    assert np.array_equal(col2, col3)                                                    # deterministic
    order, start, size, fam_of, pos_of = fam.family_tables(T, G, seed=11)
    L, L2 = np.diff(rp.astype(np.int64)), np.diff(rp2.astype(np.int64))
    assert ((L2 - L) >= 0).all() and ((L2 - L) <= 3).all()
    assert 0.1 < info["paralogue_reads"] < 0.2 and 0.005 < info["hub_reads"] < 0.015
    hub_set = set(np.random.default_rng(11 + 1).choice((T // G) * G, size=5, replace=False).tolist())   # (the generator's own draw)
    n_par = n_hub = 0
    for r in range(0, 30_000, 7):
        a = col[int(rp[r]):int(rp[r + 1])]
        b = col2[int(rp2[r]):int(rp2[r + 1])]
        assert np.array_equal(b[:a.size], a)                                             # the row's own hits, in place
        g0 = int(a[0]) // G
        extra = [int(c) for c in b[a.size:]]
        if extra and extra[-1] in hub_set and (len(extra) == 3 or fam_of[extra[-1] // G] != fam_of[g0] or abs(int(pos_of[extra[-1] // G]) - int(pos_of[g0])) > 24):
            n_hub += 1                                                                   # the hub hit comes last
            extra = extra[:-1]
        for c in extra:                                                                  # the others: ONE other gene of the family, a neighbour in its chain
            g1 = c // G
            assert fam_of[g1] == fam_of[g0] and g1 != g0 and abs(int(pos_of[g1]) - int(pos_of[g0])) <= 24
            n_par += 1
        assert len({c // G for c in extra}) <= 1
    assert n_par > 300 and n_hub > 10
    assert np.array_equal(tx >> np.uint64(32), np.arange(T, dtype=np.uint64) // np.uint64(G))   # the CLI's keys: gene << 32 | transcript
