#!/usr/bin/env python3
"""bench.py -- Gibbs iterations/sec of the mmseq hot path on MI355X (contract: see the task prompt).

A "step" is one Gibbs sweep: K1 (per-row multinomial allocation + count scatter, the stream of the hit matrix; src/mmseq.cpp:857-891)
+ K2 (Gamma redraw + trace capture, :896-917) over the whole synthetic hit matrix, for every chain on the GPU.
Headline workload = BASELINE.json's 50M-read / 200k-transcript shape (configs[2]/[3]), 1 chain per GPU; the other shapes the
round-1 review asked for (config 2, 8 chains, uniform hits, rows kept in generator order, far-hit mixes) are measured in the same
run and reported in the `extra` block of the same JSON line.

HIP events bracket K1 and K2 on every --time-every-th step INSIDE the timed region, on the stream the kernels are launched on
(default 4: an event pair costs about 9 us of stream time); roofline.avg_launch_ms is the mean over those launches.

roofline (K1 = k_sample_sell, the dominant kernel; the same block for the chain-pair kernel and the EM kernel in `roofline_other`):
  bound        "hbm".  K1 is bound by the stream it reads: with the same stream served from the caches it runs 21 % faster
               (DESIGN.md section 4), and the instruction side (VALU issue) sits right behind it
  traffic      HBM bytes per launch from the FETCH_SIZE / WRITE_SIZE passes of this exact kernel build (profiles/pmc_counters.json,
               written by tools/pmc_summary.py, stamped with a hash of the kernel sources and build flags; null if they changed since)
  achieved     traffic / avg_launch_ms in GB/s;  peak 8000 GB/s (MI355X_MICROARCH.md);  frac = hbm_counter_frac = achieved / peak
  pattern_read_peak_gbs   what a pure read with K1's access pattern reaches on this part (tools/stream_bench.hip: 6.5-6.7 TB/s)
  algorithmic_x_peak      SURVEY 8(d)'s figure: bytes of the u32 CSR / time / 8 TB/s.  The kernel streams a 1.1 byte-per-hit encoding of
               that CSR, so this exceeds 1 -- it says how much faster than a CSR-streaming kernel at the HBM roofline this is
  valu         counted VALU instructions x 4.3 clocks (tools/issue_bench.hip: what a wave64 VALU instruction of K1's mix occupies its
               SIMD for) / (1024 SIMDs x effective clock x time); effective_clock_ghz = GRBM_GUI_ACTIVE / 8 XCDs / time
  lds          SQ_LDS_IDX_ACTIVE / 256 CUs / kernel cycles

  python bench.py --gpus 1 --steps 256 --warmup 16
  python -m torch.distributed.run --nnodes=1 --nproc-per-node N --master-addr 127.0.0.1 --master-port P \
         bench.py --gpus N --steps K --warmup W            (one rank per GPU, RCCL)
"""
import argparse
import hashlib
import json
import os
import sys
import time

ROOT = os.path.dirname(os.path.abspath(__file__))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)

HBM_PEAK_GBS = 8000.0        # MI355X HBM3E spec peak (/opt/skills/guides/MI355X_MICROARCH.md)
PATTERN_READ_PEAK_GBS = 6600.0   # tools/stream_bench.hip on MI355X: single-wave workgroups reading contiguous ranges of 1.5 KB blocks
VALU_CLOCKS_PER_INST = 4.3   # tools/issue_bench.hip (profiles/r03_issue_bench.txt): VOP3 / fp64 / compare / SDWA / 32-bit multiply
                             # instructions occupy their SIMD for 4.2-4.5 clocks per wave64 instruction, and in a mixed stream the
                             # cheap VOP1 / VOP2 class (2.3-2.9 clocks alone) costs the same; v_mad_u64_u32 5.1
# everything that decides what the kernels do and how they are launched: sources of the device code, the host code that lays the
# problem out and picks grids and ranges, and the build flags
KERNEL_SOURCES = ["mmg_math.h", "mmg_types.h", "gibbs_kernels.h", "sell_kernels.h", "sell_multi_kernels.h", "em_kernels.h", "k1.hip", "em.hip",
                  "mmgibbs.hip", "sampler.hip", "em_host.hip", "layout.hip", "Makefile"]


def kernel_hash():
    """Hash of the kernel and launch-geometry sources with comments and white space removed (what the compiler sees) + the Makefile."""
    import re
    h = hashlib.sha256()
    for f in KERNEL_SOURCES:
        src = open(os.path.join(ROOT, "mmseq_amd", "csrc", f)).read()
        if f != "Makefile":
            src = re.sub(r"/\*.*?\*/", "", src, flags=re.S)
            src = re.sub(r"//[^\n]*", "", src)
        h.update(re.sub(r"\s+", "", src).encode())
    return h.hexdigest()[:16]


def pmc_entry(entry):
    """Counters per launch of one kernel from the committed rocprofv3 PMC passes of this same kernel build (profiles/pmc_counters.json,
    written by tools/pmc_summary.py: FETCH_SIZE / WRITE_SIZE / SQ passes collected separately, FETCH doubled per the gfx950
    correction).  PMC counters cannot be collected from inside this process; a stale stamp gives None."""
    try:
        d = json.load(open(os.path.join(ROOT, "profiles", "pmc_counters.json")))
        if d["kernel_sources_sha16"] == kernel_hash():
            return d["entries"].get(entry)
    except Exception:
        pass
    return None


def roofline_block(kernel_name, entry, t_s, launches, n_tiles, stream_bytes=None, algorithmic_bytes=None):
    """The roofline object of one kernel: t_s = average launch duration in seconds (HIP events in this run), PMC figures from `entry`."""
    pmc = pmc_entry(entry)
    c = (pmc or {}).get("counters_per_launch", {})
    traffic = ((pmc["hbm_read_bytes_per_launch"] or 0) + (pmc["hbm_write_bytes_per_launch"] or 0)) if pmc and pmc.get("hbm_read_bytes_per_launch") else None
    ach = traffic / t_s / 1e9 if traffic else None
    clk = c.get("GRBM_GUI_ACTIVE")
    # effective clock of the profiled pass: its GRBM_GUI_ACTIVE (summed over the 8 XCDs) / its own mean kernel duration
    dur = (pmc or {}).get("duration_ns_in_the_grbm_pass")
    clock_ghz = clk / 8.0 / dur if clk and dur else None
    valu = c.get("SQ_INSTS_VALU")
    out = {"bound": "hbm", "kernel": kernel_name, "achieved": ach, "peak": HBM_PEAK_GBS, "unit": "GB/s",
           "frac": (ach / HBM_PEAK_GBS) if ach else None, "traffic": traffic, "hbm_counter_frac": (ach / HBM_PEAK_GBS) if ach else None,
           "pattern_read_peak_gbs": PATTERN_READ_PEAK_GBS, "frac_of_pattern_peak": (ach / PATTERN_READ_PEAK_GBS) if ach else None,
           "avg_launch_ms": t_s * 1e3, "timed_launches": launches,
           "effective_clock_ghz": clock_ghz,
           "valu": ({"insts_per_launch": valu, "clocks_per_inst": VALU_CLOCKS_PER_INST,
                     "busy_frac": valu * VALU_CLOCKS_PER_INST / (1024 * (clk / 8.0))} if valu and clk else None),
           "lds": ({"busy_frac": c["SQ_LDS_IDX_ACTIVE"] / 256.0 / (clk / 8.0),
                    "bank_conflict_frac_of_lds_cycles": c.get("SQ_LDS_BANK_CONFLICT", 0) / max(c["SQ_LDS_IDX_ACTIVE"], 1)}
                   if c.get("SQ_LDS_IDX_ACTIVE") and clk else None),
           "wave_cycles": ({"at_s_waitcnt": c["SQ_WAIT_ANY"] / c["SQ_WAVE_CYCLES"], "waiting_to_issue": c.get("SQ_WAIT_INST_ANY", 0) / c["SQ_WAVE_CYCLES"]}
                           if c.get("SQ_WAIT_ANY") and c.get("SQ_WAVE_CYCLES") else None),
           "instructions_per_64_row_tile": ({k[9:].lower(): round(v / max(n_tiles, 1), 1) for k, v in c.items()
                                             if k in ("SQ_INSTS_VALU", "SQ_INSTS_SALU", "SQ_INSTS_BRANCH", "SQ_INSTS_LDS", "SQ_INSTS_VMEM", "SQ_INSTS_SMEM")}
                                            if c and n_tiles else None),
           "pmc_source": (pmc or {}).get("source")}
    if stream_bytes is not None:
        out["stream_bytes_per_launch"] = stream_bytes
        out["stream_frac_of_peak"] = stream_bytes / t_s / 1e9 / HBM_PEAK_GBS
    if algorithmic_bytes is not None:
        out["algorithmic_bytes_per_launch"] = algorithmic_bytes
        out["algorithmic_x_peak"] = algorithmic_bytes / t_s / 1e9 / HBM_PEAK_GBS
    return out


def cpu_quota():
    """CPUs this process may use under a cgroup CFS quota (v2 cpu.max / v1 cpu.cfs_quota_us), or None."""
    try:
        q, per = open("/sys/fs/cgroup/cpu.max").read().split()[:2]
        return None if q == "max" else max(1, -(-int(q) // int(per)))
    except (OSError, ValueError):
        pass
    try:
        q = int(open("/sys/fs/cgroup/cpu/cpu.cfs_quota_us").read())
        per = int(open("/sys/fs/cgroup/cpu/cpu.cfs_period_us").read())
        return None if q <= 0 else max(1, -(-q // per))
    except (OSError, ValueError):
        return None


def cpu_baseline(args, total_reads):
    """Oracle ("port" of src/mmseq.cpp:851-918, reference-structured: per-thread MT19937, count slabs,
    conditional-binomial multinomial) timed on this host's cores on a bounded sample of the same workload."""
    from oracle import binding as B
    Rs = min(args.cpu_sample_rows, args.rows)
    p, _ = B.synth_problem(R=Rs, T=args.transcripts, avg_hits=args.avg_hits, seed=args.seed, mapped_reads=total_reads, sort=False)
    mu0, _ = B.start_values(p)
    ncpu = os.cpu_count() or 1
    quota = cpu_quota()
    # the reference's per-thread count slabs (src/mmseq.cpp:850-855, :896-899) stop scaling at high thread
    # counts: probe a few and time the best one, so the baseline is the strongest this host offers.  A container's CPU quota
    # (16 CPUs' worth on the 256-core GPU boxes here) is a candidate of its own: more threads than that are throttled.
    # (thread counts far above a quota are throttled erratically: a 2-iteration probe once picked 128 threads on a 16-CPU quota and the
    # timed run then crawled; with a quota the candidates stop at twice its size and the probe runs 4 iterations)
    cands = sorted({t for t in (ncpu, ncpu // 2, 64, 32, 16, 8, 1, quota or 1, 2 * (quota or 1))
                    if 1 <= t <= ncpu and (quota is None or t <= 2 * quota)}, reverse=True)
    best_t, best_rate = 1, 0.0
    for t in cands:
        r = B.gibbs_ref(p, mu0, seed=args.seed, n_iter=4, trace_len=4, threads=t, want_trace=False)
        rate = Rs * 4 / r["seconds"]
        if rate > best_rate:
            best_t, best_rate = t, rate
    threads = best_t
    iters = args.cpu_iters
    r = B.gibbs_ref(p, mu0, seed=args.seed, n_iter=iters, trace_len=iters, threads=threads, want_trace=False)
    reads_it_s = Rs * iters / r["seconds"]
    r1 = B.gibbs_ref(p, mu0, seed=args.seed, n_iter=max(1, iters // 8), trace_len=max(1, iters // 8), threads=1,
                     want_trace=False)
    reads_it_s_1 = Rs * max(1, iters // 8) / r1["seconds"]
    return {"value": reads_it_s / args.rows, "unit": "iterations/s", "cores": threads, "kind": "port",
            "sample": "first %d generator rows of the same workload (T=%d, avg %.0f hits), %d iterations on %d host threads "
                      "(best of %s on this %d-CPU host, CPU quota %s); reads*iter/s scaled to the %d-read problem"
                      % (Rs, args.transcripts, args.avg_hits, iters, threads, cands, ncpu, quota, args.rows),
            "reads_iters_per_sec": reads_it_s, "single_thread_iterations_per_sec": reads_it_s_1 / args.rows}


def side_measurement(name, rows, transcripts, avg_hits, chains=1, uniform=False, sort=True, far_fraction=0.0, steps=48, warmup=8,
                     seed=1234, device=0, note="", multiplicities=False):
    """One more workload, same protocol (inputs resident, HIP events on the launch stream), shorter: ms per sweep and which kernel ran.
    multiplicities: the rows get a k array with the distribution a collapsed 50 M-read file of this generator has (93.6 % k = 1,
    5.3 % k = 2, ... 0.12 % k in 9..36: tools/collapse_probe.py) -- what every real hits file produces (src/mmseq.cpp:409-418)."""
    import numpy as np
    import torch
    from mmseq_amd import Problem, Sampler
    from mmseq_amd import dist as mdist
    t0 = time.perf_counter()
    prob = Problem.synthetic(rows, transcripts, avg_hits, seed=seed, uniform=uniform, sort=sort, far_fraction=far_fraction,
                             mapped_reads=rows, device=device)
    if multiplicities:
        rp, ci = prob.download()
        l = prob.l()
        prob.close()
        rng = np.random.default_rng(seed)
        u = rng.random(rows)
        k = np.ones(rows, np.uint32)
        if multiplicities == "heavy":
            # a heavily collapsed file: most READS sit in hit sets shared by tens to thousands (half the rows k = 1, 30 % 2..8,
            # 15 % 9..64, 4 % 65..300, 1 % 300..20000 log-uniform): all three row paths of spec version 5
            for lo_u, hi_u, lo_k, hi_k in ((0.5, 0.8, 2, 8), (0.8, 0.95, 9, 64), (0.95, 0.99, 65, 300), (0.99, 1.0, 300, 20000)):
                sel = (u >= lo_u) & (u < hi_u)
                k[sel] = np.exp(rng.uniform(np.log(lo_k), np.log(hi_k + 1), size=int(sel.sum()))).astype(np.uint32).clip(lo_k, hi_k)
        else:
            for thr, val in ((0.064, 2), (0.011, 3), (0.0035, 4), (0.002, 6)):
                k[u < thr] = val
            big = u < 0.0012
            k[big] = rng.integers(9, 37, size=int(big.sum())).astype(np.uint32)
        prob = Problem.from_csr(rp, ci, l, k=k, device=device)
        del rp, ci, k, u
    build_s = time.perf_counter() - t0
    inf = prob.info
    mu0, _ = prob.start_values()
    # clocks first (as in the headline: the GPU raises them over the first ~100 ms of load, and the problem build before this point is
    # mostly idle time for it): a scratch sampler of the same shape runs until 0.2 s of sweeps have passed; nothing of it is kept
    scratch = Sampler(prob, mu0, seed=seed + 1, n_chains=chains, chain_base=1 << 20, gibbs_iter=1 << 20, trace_len=1, keep_trace=False, timing=0)
    mdist.use_current_stream(scratch)
    ts = time.perf_counter()
    while time.perf_counter() - ts < 0.2:
        scratch.run(8)
        torch.cuda.synchronize()
    scratch.close()
    smp = Sampler(prob, mu0, seed=seed, n_chains=chains, gibbs_iter=1024, trace_len=1024, keep_trace=True, timing=4)
    mdist.use_current_stream(smp)
    smp.run(warmup)
    torch.cuda.synchronize()
    smp.reset_timing()
    t0 = time.perf_counter()
    smp.run(steps)
    torch.cuda.synchronize()
    el = time.perf_counter() - t0
    tm = smp.timing()
    for c in range(chains):   # every chain assigned every read exactly once in the last sweep
        assert int(smp.counts(c).astype(np.int64).sum()) == inf.total_k, "count conservation, chain %d" % c
    out = {"name": name, "reads": inf.m, "mapped_reads_total_k": inf.total_k, "transcripts": inf.n, "hits": inf.nnz, "chains": chains, "uniform": bool(uniform),
           "canonical_layout": bool(sort), "far_fraction": far_fraction, "steps": steps, "ms_per_step": el / steps * 1e3,
           "chain_iterations_per_sec": chains * steps / el,
           "k1_avg_launch_ms_all_chains": tm["sample_ms"] / max(tm["sample_launches"], 1),
           "k2_avg_launch_ms": tm["update_ms"] / max(tm["update_launches"], 1),
           "sample_kernel": {0: "k_sample (CSR tiles)", 2: "k_sample_sell"}[inf.sample_kernel],
           "fast_tile_fraction": (inf.fast_tiles / inf.n_tiles) if inf.sample_kernel == 2 and inf.n_tiles else 0.0,
           "far_tile_fraction": (inf.far_tiles / inf.n_tiles) if inf.sample_kernel == 2 and inf.n_tiles else 0.0,
           "stream_bytes": inf.stream_bytes, "problem_build_s": build_s}
    if note:
        out["note"] = note
    smp.close()
    prob.close()
    return out


def em_measurement(rows, transcripts, avg_hits, seed=1234, device=0, sweeps=20):
    """EM sweeps (src/mmseq.cpp:761-806) on the config-3 problem: wall time per sweep of mmg_em_step (the rows pass k_em_sell plus the
    per-transcript kernels and one read-back of the log-likelihood)."""
    import torch
    from mmseq_amd import Problem
    prob = Problem.synthetic(rows, transcripts, avg_hits, seed=seed, mapped_reads=rows, device=device)
    mu0, _ = prob.start_values()
    em = prob.em_stepper(mu0)
    for _ in range(3):
        em.step()
    t0 = time.perf_counter()
    for _ in range(sweeps):
        em.step()
    ms = (time.perf_counter() - t0) / sweeps * 1e3
    out = {"name": "EM sweep, 50M x 200k", "ms_per_sweep": ms, "sweeps": sweeps, "stream_kernel": em.stats_raw()["stream_kernel"],
           "n_tiles": prob.info.n_tiles, "stream_bytes": prob.info.stream_bytes}
    em.close()
    prob.close()
    return out


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=256)
    ap.add_argument("--warmup", type=int, default=16)
    ap.add_argument("--rows", type=int, default=50_000_000, help="reads per GPU")
    ap.add_argument("--transcripts", type=int, default=200_000)
    ap.add_argument("--avg-hits", type=float, default=20.0)
    ap.add_argument("--chains", type=int, default=1, help="chains per GPU")
    ap.add_argument("--mode", choices=["chains", "shard"], default="chains")
    ap.add_argument("--seed", type=int, default=1234)
    ap.add_argument("--time-every", type=int, default=4, help="HIP-event pairs around K1/K2 on every N-th step of the timed region "
                    "(a pair costs about 9 us of stream time; 1 = every step)")
    ap.add_argument("--settle-iters", type=int, default=256, help="iterations of a scratch chain before the warm-up steps (GPU clock ramp)")
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--no-extra", action="store_true", help="skip the side measurements of the extra block")
    ap.add_argument("--cpu-sample-rows", type=int, default=2_000_000)
    ap.add_argument("--cpu-iters", type=int, default=24)
    args = ap.parse_args()

    import numpy as np
    import torch
    import torch.distributed as dist
    from mmseq_amd import Problem, Sampler
    from mmseq_amd import dist as mdist

    world = int(os.environ.get("WORLD_SIZE", "1"))
    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    if world != args.gpus:
        raise SystemExit("--gpus %d but WORLD_SIZE=%d: launch with torch.distributed.run --nproc-per-node %d"
                         % (args.gpus, world, args.gpus))
    if not torch.cuda.is_available():
        raise SystemExit("bench.py needs a HIP device (no CPU fallback)")
    torch.cuda.set_device(local_rank)
    if world > 1:
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        dist.init_process_group("nccl", device_id=torch.device("cuda", local_rank))

    # ---- workload (synthetic, generated straight into device CSR and laid out by the library; not timed)
    if args.mode == "shard":
        total_reads = args.rows * world
        row0 = args.rows * rank
    else:
        total_reads = args.rows
        row0 = 0
    prob = Problem.synthetic(args.rows, args.transcripts, args.avg_hits, seed=args.seed, row0=row0,
                             mapped_reads=total_reads, device=local_rank)
    inf = prob.info
    mu0, _ = prob.start_values()
    if args.mode == "shard" and world > 1:
        t = torch.from_numpy(mu0 * prob.l()).cuda()       # shares k/|row| summed over ranks, then / l
        dist.all_reduce(t)
        mu0 = t.cpu().numpy() / prob.l()
    # every iteration is a kept sample (BASELINE.md B formula): the reference's 1024-iteration, 1024-sample run; a longer timed region
    # (--warmup + --steps > 1024) keeps every iteration as well, in a longer resident trace
    need = args.warmup + args.steps
    trace_len = gibbs_iter = 1024 if need <= 1024 else -(-need // 64) * 64
    if gibbs_iter * args.transcripts * args.chains * 8 > 200e9:
        raise SystemExit("warmup + steps = %d: the resident trace would exceed 200 GB" % need)
    chain_base = 0 if args.mode == "shard" else rank * args.chains
    smp = Sampler(prob, mu0, seed=args.seed, n_chains=args.chains, chain_base=chain_base, gibbs_iter=gibbs_iter,
                  trace_len=trace_len, keep_trace=True,
                  timing=args.time_every if args.steps >= 4 * args.time_every else 1)   # short runs: every step
    mdist.use_current_stream(smp)
    counts = mdist.counts_tensor(smp) if args.mode == "shard" else None
    moments = mdist.moments_tensor(smp)

    def step():
        if args.mode == "shard":
            mdist.shard_step(smp, counts)
        else:
            smp.run(1)

    cold_ms = None
    if args.settle_iters > 0:
        # The GPU raises its clocks over the first ~100 ms of load: steps right after start-up are ~10 % slower than steps 200+.
        # A scratch chain (different key, nothing kept) brings the clocks up before the W warm-up steps, so that short timed
        # regions measure the steady state a 1024-iteration run lives in.  Not part of W or K; both regimes are reported.
        scratch = Sampler(prob, mu0, seed=args.seed + 1, n_chains=1, chain_base=1 << 20, gibbs_iter=1 << 20, trace_len=1,
                          keep_trace=False, timing=0)
        mdist.use_current_stream(scratch)
        torch.cuda.synchronize()
        tc = time.perf_counter()
        scratch.run(min(64, args.settle_iters))
        torch.cuda.synchronize()
        cold_ms = (time.perf_counter() - tc) / min(64, args.settle_iters) * 1e3
        scratch.run(max(0, args.settle_iters - 64))
        torch.cuda.synchronize()
        scratch.close()
    for _ in range(args.warmup):
        step()
    torch.cuda.synchronize()
    smp.reset_timing()
    if world > 1:
        dist.barrier()
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(args.steps):
        step()
    if args.mode == "chains":
        mdist.pool_moments(moments)                        # the one collective of chains mode
    torch.cuda.synchronize()
    if world > 1:
        dist.barrier()
    torch.cuda.synchronize()
    elapsed = time.perf_counter() - t0
    if world > 1:
        te = torch.tensor([elapsed], dtype=torch.float64, device="cuda")
        dist.all_reduce(te, op=dist.ReduceOp.MAX)
        elapsed = float(te.item())
    tm = smp.timing()

    # sanity inside the bench: every read was allocated exactly once in the last sweep
    if args.mode == "chains":
        assert int(smp.counts(0).astype(np.int64).sum()) == inf.total_k
    smp.close()

    if rank == 0:
        C = args.chains
        chains_total = C * (world if args.mode == "chains" else 1)
        iters_per_s = chains_total * args.steps / elapsed
        reads_per_chain = total_reads
        k1_ms = tm["sample_ms"] / max(tm["sample_launches"], 1) / C   # per chain: the sample() call of C chains / C
        k2_ms = tm["update_ms"] / max(tm["update_launches"], 1)
        # algorithmic bytes of one K1 launch (one GPU, one chain): u32 row_ptr + u32 col_idx streamed once, fp64 mu read + int32
        # count write (SURVEY 8d / DESIGN.md section 4)
        b_k1 = 4 * (inf.m + 1) + 4 * inf.nnz + 12 * inf.n
        b_sweep = 4 * (inf.m + 1) + 4 * inf.nnz + 28 * C * inf.n
        kname = {0: "k_sample", 2: "k_sample_sell"}[inf.sample_kernel]
        t_k1 = k1_ms * 1e-3
        # chains > 1: one event pair brackets the whole sample() call -- pair launches plus the launches for the other tile lists;
        # per-kernel PMC figures belong to the 1-chain run only (roofline_other carries the pair kernel's own block)
        roof = roofline_block(kname + " (K1)", "k1_1chain" if C == 1 else "none", t_k1, tm["sample_launches"], inf.n_tiles,
                              stream_bytes=inf.stream_bytes, algorithmic_bytes=b_k1)
        roof["padded_slots_per_hit"] = (inf.padded_slots / inf.nnz) if inf.nnz else None
        roof["k_update_avg_launch_ms"] = k2_ms
        roof["sweep_bytes"] = b_sweep
        roof["note"] = ("HBM: traffic / time against the 8 TB/s peak (and against the 6.6 TB/s a pure read with this access pattern reaches); "
                        "valu.busy_frac = counted VALU instructions x 4.3 clocks / SIMD cycles; algorithmic_x_peak = SURVEY 8(d) u32-CSR bytes / "
                        "time / 8 TB/s (> 1: the kernel streams a 1.1 B/hit encoding, not the CSR).  PMC figures are null when "
                        "profiles/pmc_counters.json was collected on other kernel sources, and at chains > 1 (avg_launch_ms is then the "
                        "sample() call of all chains divided by the chains)")
        out = {
            "metric": "gibbs_iterations_per_sec", "value": iters_per_s, "unit": "iterations/s",
            "n_gpus": world, "steps": args.steps, "warmup": args.warmup,
            "ms_per_step": elapsed / args.steps * 1e3, "higher_is_better": True, "scaling": "weak",
            "vs_baseline": None, "dtype": "f64", "data": "synthetic",
            "config": {"workload": "50M-read x 200k-transcript synthetic CSR hits (BASELINE.json configs[2]/[3] shape)"
                       if (args.rows, args.transcripts) == (50_000_000, 200_000) else "custom synthetic CSR hits",
                       "reads_per_gpu": inf.m, "transcripts": inf.n, "hits_per_gpu": inf.nnz,
                       "avg_hits_per_read": args.avg_hits, "chains_per_gpu": C, "mode": args.mode,
                       "parallelism": ("%d independent chains (1 all-reduce of posterior moments)" % chains_total)
                       if args.mode == "chains" else ("read-sharded single chain over %d GPUs "
                                                      "(int32 count all-reduce per iteration)" % world),
                       "trace": "every iteration kept (fp64 mu trace resident in HBM)", "generator_seed": args.seed,
                       "layout": "rows generated in generator order, stored in the library's canonical order (device radix sort)",
                       "clock_settle_iters_before_warmup": args.settle_iters,
                       "cold_ms_per_step_first_64_iterations_after_start": cold_ms},
            "reads_iters_per_sec": iters_per_s * reads_per_chain,
            "roofline": roof,
        }
        if world == 1 and not args.no_extra:
            prob.close()
            prob = None
            torch.cuda.empty_cache()
            extra = []
            R3, T3, H3 = 50_000_000, 200_000, 20.0
            for kw in (dict(name="config 2: 5M reads x 50k transcripts, avg 8 hits, 1 chain", rows=5_000_000, transcripts=50_000, avg_hits=8.0, steps=256, warmup=64),
                       dict(name="config 3: 50M x 200k, 8 chains in one GPU", rows=R3, transcripts=T3, avg_hits=H3, chains=8, steps=16, warmup=4),
                       dict(name="50M x 200k with multiplicities (k > 1 on 6.4 % of the rows, up to 36; such a row is stored k times)", rows=R3, transcripts=T3, avg_hits=H3,
                            multiplicities=True, steps=32),
                       dict(name="50M x 200k like a real hits file: multiplicities (k > 1 on 6.4 % of the rows) AND 2 % of the rows with a hit anywhere in the "
                                 "transcriptome, 8 chains in one GPU (pairs over the k = 1 register-path tiles, one launch each for the far and the multiplicity tiles)",
                            rows=R3, transcripts=T3, avg_hits=H3, multiplicities=True, far_fraction=0.02, chains=8, steps=16, warmup=4),
                       dict(name="a heavily collapsed file: 5M hit sets x 200k transcripts, avg 20 hits, multiplicities from 1 to 20000 (mapped_reads_total_k reads; "
                                 "k draws up to 16 per step of the binomial chain, the chain above)", rows=5_000_000, transcripts=T3, avg_hits=H3,
                            multiplicities="heavy", steps=32),
                       dict(name="50M x 200k, 2 % of the rows with a hit anywhere in the transcriptome", rows=R3, transcripts=T3, avg_hits=H3, far_fraction=0.02, steps=32),
                       dict(name="50M x 200k, 20 % of the rows with a hit anywhere in the transcriptome", rows=R3, transcripts=T3, avg_hits=H3, far_fraction=0.2, steps=24),
                       dict(name="50M x 200k, hits uniform over all transcripts (SURVEY App. D worst case)", rows=R3, transcripts=T3, avg_hits=H3, uniform=True, steps=8, warmup=2),
                       dict(name="50M x 200k, rows kept in generator order (MMG_LAYOUT_KEEP_ROWS: what round 1 ran when the caller did not sort)",
                            rows=R3, transcripts=T3, avg_hits=H3, sort=False, steps=8, warmup=2)):
                try:
                    extra.append(side_measurement(seed=args.seed, device=local_rank, **kw))
                except Exception as e:                      # a side measurement must not take the headline down
                    extra.append({"name": kw["name"], "error": repr(e)})
                torch.cuda.empty_cache()
            out["extra"] = extra
            # the same roofline object for the two other hot kernels: the chain-pair kernel (BASELINE configs[2]) and the EM rows pass
            other = []
            c3 = next((e for e in extra if e.get("name", "").startswith("config 3") and "error" not in e), None)
            if c3:
                tiles = inf.n_tiles
                pairs = c3["chains"] // 2
                # ONE launch advances all the chains, pair by pair (grid.y = pair): its duration, its counters, `pairs` passes over the
                # stream and `pairs` x tiles tile-visits (instructions_per_64_row_tile is per visit, i.e. per tile and PAIR of chains)
                rb = roofline_block("k_sample_sell_multi<2> (K1m: one launch advances the %d chains in %d pairs)" % (c3["chains"], pairs), "k1m_8chains",
                                    c3["k1_avg_launch_ms_all_chains"] * 1e-3, c3["steps"] // 4 + 1, tiles * pairs,
                                    stream_bytes=c3["stream_bytes"] * pairs,
                                    algorithmic_bytes=pairs * (4 * (c3["reads"] + 1) + 4 * c3["hits"] + 24 * c3["transcripts"]))
                rb["chain_iterations_per_sec"] = c3["chain_iterations_per_sec"]
                rb["bound"] = "lds+valu"
                rb["note"] = ("the stream is read once per PAIR of chains: about 3 TB/s of HBM traffic, not the bound; lds.busy_frac and valu.busy_frac "
                              "(counted VALU instructions x 4.3 clocks / SIMD cycles) are -- see DESIGN.md section 4 for the instruction budget per tile")
                other.append(rb)
            try:
                em = em_measurement(R3, T3, H3, seed=args.seed, device=local_rank)
                rb = roofline_block("k_em_sell (K3: rows pass of one EM sweep)", "em_sweep", em["ms_per_sweep"] * 1e-3, em["sweeps"], em["n_tiles"],
                                    stream_bytes=em["stream_bytes"])
                rb["bound"] = "lds"
                rb["note"] = ("avg_launch_ms is the wall time of one mmg_em_step (rows pass + per-transcript kernels + one read-back); bound by the LDS: "
                              "per hit one 8-byte gather, one scale word and two 64-bit atomic adds on random window slots (lds.busy_frac, most of it "
                              "bank conflicts); the limbs of a term are three 64-bit shifts (valu.busy_frac)")
                other.append(rb)
            except Exception as e:
                other.append({"kernel": "k_em_sell", "error": repr(e)})
            torch.cuda.empty_cache()
            out["roofline_other"] = other
        if world == 1 and not args.no_cpu_baseline:
            out["cpu_baseline"] = cpu_baseline(args, total_reads)
            out["speedup_vs_cpu_baseline"] = iters_per_s / out["cpu_baseline"]["value"]
        print(json.dumps(out), flush=True)
    if prob is not None:
        prob.close()
    if world > 1:
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
