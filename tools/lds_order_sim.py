"""Model of the LDS bank rule (MI355X_MICROARCH.md: ds_read_b64 = 2 passes of 32 lanes over 32 bank pairs, one cycle per distinct address on the
busiest bank) applied to the gathers of K1 under candidate tie orders of the canonical layout: cycles per gather instruction.
Run in the build container (needs oracle/liboracle.so); the result picked the csum tie-break of ABI 4 (mmg_types.h)."""
import sys, numpy as np
sys.path.insert(0, '/root/repo')
from oracle import binding as B
R, T = 2_000_000, 8000
p, _ = B.synth_problem(R=R, T=T, avg_hits=20, seed=1234, sort=True)
rp = p.row_ptr.astype(np.int64); ci = p.col_idx
L = np.diff(rp)
key, h = B.row_keys(rp, ci)
def lds_cycles(order_rows):
    """order_rows: array of row indices in stored order; tiles = consecutive 64 rows inside equal (band) runs (approx: ignore run cuts)."""
    tot = 0; base = 0; ninst = 0
    n = len(order_rows) // 64 * 64
    rows = order_rows[:n].reshape(-1, 64)
    # sample tiles
    sel = np.random.default_rng(0).choice(rows.shape[0], 3000, replace=False)
    for t in sel:
        r = rows[t]
        lens = L[r]; ng = (lens.max() + 3) // 4
        band = (key[r[0]] >> np.uint64(18)) & np.uint64((1<<45)-1)
        wbase = int(band) * 64
        for j in range(int(ng) * 4):
            slot = np.where(j < lens, ci[np.minimum(rp[r] + j, rp[r+1]-1)].astype(np.int64) - wbase, 255)
            if (slot < 0).any() or (slot > 255).any(): continue
            for half in (slot[:32], slot[32:]):
                u = np.unique(half)
                cyc = np.bincount(u % 32, minlength=32).max()
                tot += cyc
            ninst += 1
    return tot / ninst
order_hash = np.arange(len(L))  # already canonical (key, hash)
print("hash order: LDS cycles per ds_read_b64:", lds_cycles(order_hash))
# lexicographic within key: sort by key then by hits tuple
maxl = int(L.max())
mat = np.full((len(L), maxl), 0xffffffff, np.uint32)
rid = np.repeat(np.arange(len(L)), L); pos = np.arange(len(ci)) - rp[rid]
mat[rid, pos] = ci
cols = [mat[:, j] for j in range(maxl-1, -1, -1)] + [key]
order_lex = np.lexsort(cols)
print("lexicographic order: ", lds_cycles(order_lex))
# variants
def lex_prefix(n):
    cols = [h] + [mat[:, j] for j in range(n-1, -1, -1)] + [key]
    return np.lexsort(cols)
for n in (2, 4, 8):
    print("lex prefix", n, "then hash:", lds_cycles(lex_prefix(n)))
# descending lexicographic (last hit first)
last = np.full((len(L), maxl), 0, np.uint32)
last[rid, (L[rid]-1-pos)] = ci
cols = [last[:, j] for j in range(maxl-1, -1, -1)] + [key]
print("lex from the last hit:", lds_cycles(np.lexsort(cols)))
first = mat[:, 0].astype(np.int64); lastv = last[:, 0].astype(np.int64)
print("(first, last, hash):", lds_cycles(np.lexsort((h, lastv, first, key))))
print("(last, first, hash):", lds_cycles(np.lexsort((h, first, lastv, key))))
print("(last, first, second):", lds_cycles(np.lexsort((h, mat[:,1], first, lastv, key))))
print("(last, 2nd last, first):", lds_cycles(np.lexsort((h, first, last[:,1], lastv, key))))
mid = ci[rp[:-1] + (L-1)//2].astype(np.int64)
print("(first, mid, last):", lds_cycles(np.lexsort((h, lastv, mid, first, key))))
print("(last, mid, first):", lds_cycles(np.lexsort((h, first, mid, lastv, key))))
sums = np.add.reduceat(ci.astype(np.int64), rp[:-1][L>0]) if (L>0).all() else None
if sums is not None:
    print("(sum of hits):", lds_cycles(np.lexsort((h, sums, key))))
    print("(last, sum):", lds_cycles(np.lexsort((h, sums, lastv, key))))
    print("(last+first, last):", lds_cycles(np.lexsort((h, lastv, lastv+first, key))))
cs = np.concatenate(([0], np.cumsum(ci.astype(np.int64))))
halfpos = rp[:-1] + L//2
s1 = cs[halfpos] - cs[rp[:-1]]; s2 = cs[rp[1:]] - cs[halfpos]
def morton(a, b, bits=14):
    a = a.astype(np.uint64); b = b.astype(np.uint64); r = np.zeros_like(a)
    for i in range(bits):
        r |= ((a >> np.uint64(i)) & np.uint64(1)) << np.uint64(2*i+1)
        r |= ((b >> np.uint64(i)) & np.uint64(1)) << np.uint64(2*i)
    return r
wb = ((key >> np.uint64(18)) & np.uint64((1<<45)-1)).astype(np.int64) * 64
r1 = s1 - wb*(L//2); r2 = s2 - wb*(L - L//2)
print("morton(s1,s2):", lds_cycles(np.lexsort((h, morton(r1, r2), key))))
print("(sum//4, last):", lds_cycles(np.lexsort((h, lastv, sums//4, key))))
print("(sum//16, last):", lds_cycles(np.lexsort((h, lastv, sums//16, key))))
print("(sum, last):", lds_cycles(np.lexsort((h, lastv, sums, key))))
