"""A hit graph with the tail of a real transcriptome, for benchmarks and tests (host side, numpy; nothing here is on the product path).

The device generator's gene-block mode (mmg_synth_desc.gene_size / far_family) has paralogue families of ONE fixed size whose members a
read picks uniformly.  Real families are not like that (src/bam2hits.cpp:271-300 keeps up to `-m 100` alignments of a read; the hit sets
of README.md:77-82): their sizes follow a power law -- most genes have no paralogue, a few families (olfactory receptors, zinc fingers,
histones) have hundreds of members -- a read that also hits a paralogue hits a SIMILAR one, i.e. a neighbour in sequence space, and a few
transcripts (repeat-bearing UTRs) share reads with hundreds of unrelated genes.  `power_law_families` adds exactly that to rows whose hits
lie inside one gene:

  * the genes (gene_size consecutive transcripts) are dealt into families in a random order; family sizes are Pareto(alpha) in genes,
    from 1 to max_family_transcripts / gene_size -- 32 ... 5 000 transcripts at the defaults; a family's members are scattered over the
    caller's gene numbering, and have an ORDER inside the family (sequence similarity: a chain);
  * a fraction `paralogue` (0.2) of the reads of genes with a family get one or two more hits in ANOTHER gene of the family at distance
    1 + Geometric(1/2) in the family's order;
  * a fraction `hub` (0.01) of all reads get a hit on one of `n_hubs` (50) hub transcripts scattered over the transcriptome.

The rows come back as a CSR in the given row order with unsorted hits (the library's canonical layout sorts them); tx_order is what the
CLI passes for a hits file: gene << 32 | transcript, the genes in the caller's (name) order.
"""
import numpy as np


def family_tables(T, gene_size, seed, alpha=1.2, max_family_transcripts=5000):
    """(order, fam_start, fam_size, fam_of_gene, pos_of_gene): `order` = the genes in family order (a permutation), family f = order[fam_start[f] :
    fam_start[f] + fam_size[f]]."""
    rng = np.random.default_rng(seed)
    n_genes = T // gene_size
    order = rng.permutation(n_genes).astype(np.int64)
    cap = max(1, max_family_transcripts // gene_size)
    sizes = []
    left = n_genes
    while left > 0:
        draw = np.minimum(cap, np.floor((1.0 - rng.random(4096)) ** (-1.0 / alpha))).astype(np.int64)
        for s in draw:
            s = int(min(s, left))
            sizes.append(s)
            left -= s
            if left == 0:
                break
    fam_size = np.asarray(sizes, np.int64)
    fam_start = np.concatenate(([0], np.cumsum(fam_size)[:-1]))
    fam_of_pos = np.repeat(np.arange(fam_size.size), fam_size)
    fam_of_gene = np.empty(n_genes, np.int64)
    pos_of_gene = np.empty(n_genes, np.int64)
    fam_of_gene[order] = fam_of_pos
    pos_of_gene[order] = np.arange(n_genes) - fam_start[fam_of_pos]
    return order, fam_start, fam_size, fam_of_gene, pos_of_gene


def power_law_families(row_ptr, col_idx, T, gene_size, seed=1234, paralogue=0.2, hub=0.01, n_hubs=50, alpha=1.2, max_family_transcripts=5000,
                       chunk_rows=4_000_000, max_distance=None):
    """Adds paralogue and hub hits to rows whose hits lie inside one gene each (see the module text).  Returns (row_ptr u64, col_idx u32,
    tx_order u64, info): info = family sizes in genes, the share of reads that got a paralogue hit and a hub hit."""
    rp = np.asarray(row_ptr).astype(np.int64)
    col = np.asarray(col_idx)
    m = rp.size - 1
    order, fam_start, fam_size, fam_of_gene, pos_of_gene = family_tables(T, gene_size, seed, alpha, max_family_transcripts)
    n_genes = order.size
    rng = np.random.default_rng(seed + 1)
    hubs = rng.choice(n_genes * gene_size, size=n_hubs, replace=False).astype(np.uint32)
    L = np.diff(rp)
    out_cols, out_len = [], np.empty(m, np.int64)
    n_par = n_hub = 0
    for r0 in range(0, m, chunk_rows):
        r1 = min(m, r0 + chunk_rows)
        n = r1 - r0
        Lc = L[r0:r1]
        crng = np.random.default_rng([seed, r0])                        # a chunk's extras depend on (seed, first row), not on the chunking of other rows
        lead = col[np.minimum(rp[r0:r1], col.size - 1)].astype(np.int64)
        g0 = np.minimum(lead // gene_size, n_genes - 1)
        f = fam_of_gene[g0]
        fs = fam_size[f]
        u = crng.random((4, n))
        par = (Lc > 0) & (fs > 1) & (u[0] < paralogue)
        # the other gene: distance 1 + Geometric(1/2) along the family's chain, towards the side that has room (reflected at the ends)
        d = crng.geometric(0.5, size=n)
        if max_distance:
            d = np.minimum(d, max_distance)
        d = np.minimum(d, fs - 1)
        p0 = pos_of_gene[g0]
        up = u[1] < 0.5
        p1 = np.where(up, p0 + d, p0 - d)
        p1 = np.where(p1 >= fs, p0 - d, p1)
        p1 = np.where(p1 < 0, p0 + d, p1)
        p1 = np.clip(p1, 0, np.maximum(fs - 1, 0))
        g1 = order[fam_start[f] + p1]
        par &= g1 != g0
        n_extra_par = np.where(par, 1 + (u[2] < 0.5), 0)
        isoform = crng.integers(0, gene_size, size=(2, n))
        isoform[1] = np.where(isoform[1] == isoform[0], (isoform[0] + 1) % gene_size, isoform[1])
        hb = (Lc > 0) & (u[3] < hub)
        hub_t = hubs[crng.integers(0, n_hubs, size=n)]
        extra = n_extra_par + hb
        Ln = Lc + extra
        out_len[r0:r1] = Ln
        rpn = np.concatenate(([0], np.cumsum(Ln)))
        cn = np.empty(int(rpn[-1]), np.uint32)
        base = col[rp[r0]:rp[r1]]
        shift = np.repeat(rpn[:-1] - (rp[r0:r1] - rp[r0]), Lc)
        cn[np.arange(base.size) + shift] = base
        tail = rpn[:-1] + Lc
        sel = np.nonzero(par)[0]
        cn[tail[sel]] = (g1[sel] * gene_size + isoform[0][sel]).astype(np.uint32)
        sel2 = np.nonzero(n_extra_par == 2)[0]
        cn[tail[sel2] + 1] = (g1[sel2] * gene_size + isoform[1][sel2]).astype(np.uint32)
        selh = np.nonzero(hb)[0]
        cn[tail[selh] + n_extra_par[selh]] = hub_t[selh]
        out_cols.append(cn)
        n_par += int(par.sum())
        n_hub += int(hb.sum())
    new_rp = np.zeros(m + 1, np.uint64)
    new_rp[1:] = np.cumsum(out_len)
    t_ids = np.arange(T, dtype=np.uint64)
    tx_order = ((t_ids // np.uint64(gene_size)) << np.uint64(32)) | t_ids
    info = dict(families=int(fam_size.size), largest_family_transcripts=int(fam_size.max()) * gene_size, genes_in_families_of_2_or_more=int(fam_size[fam_size > 1].sum()),
                paralogue_reads=n_par / max(m, 1), hub_reads=n_hub / max(m, 1))
    return new_rp, (np.concatenate(out_cols) if out_cols else np.zeros(0, np.uint32)), tx_order, info
