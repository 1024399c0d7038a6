#!/usr/bin/env python3
"""bench.py -- Gibbs iterations/sec of the mmseq hot path on MI355X (contract: see the task prompt).

A "step" is one Gibbs sweep: K1 (per-row multinomial allocation + count scatter, the stream of the hit matrix; src/mmseq.cpp:857-891)
+ K2 (Gamma redraw + trace capture, :896-917) over the whole synthetic hit matrix, for every chain on the GPU.
Headline workload = BASELINE.json's 50M-read / 200k-transcript shape (configs[2]/[3]), 1 chain per GPU; the other shapes the
round-1 review asked for (config 2, 8 chains, uniform hits, rows kept in generator order, far-hit mixes) are measured in the same
run and reported in the `extra` block of the same JSON line.

HIP events bracket K1 and K2 on every --time-every-th step INSIDE the timed region, on the stream the kernels are launched on
(default 4: an event pair costs about 9 us of stream time); roofline.avg_launch_ms is the mean over those launches.

roofline (K1 = k_sample_sell, the dominant kernel):
  bound        "valu": the kernel is bound by VALU issue, not by HBM (profiles/*_sq_counters.md)
  achieved     wave-level VALU issue passes per second = counted passes per launch / avg_launch_ms, passes = SQ_INSTS_VALU + 3 per
               quarter-rate instruction (the v_mad_u64_u32 of Philox), from the committed PMC pass of this exact kernel build
               (profiles/pmc_counters.json, stamped with a hash of the kernel sources; null if the sources changed since)
  peak         1024 SIMDs x 2.4 GHz / 4 clocks per wave64 instruction = 614.4 G passes/s
  frac         achieved / peak  (<= 1 by construction)
  traffic      HBM bytes per launch from the FETCH_SIZE / WRITE_SIZE passes (same file); hbm_frac = traffic / time / 8 TB/s
  algorithmic_x_peak   SURVEY 8(d)'s figure: bytes of the u32 CSR / time / 8 TB/s.  The kernel streams a 1.1 byte-per-hit encoding of
               that CSR, so this exceeds 1 -- it says how much faster than a CSR-streaming kernel at the HBM roofline this is.

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
VALU_PEAK_GPASS = 1024 * 2.4 / 4   # 256 CUs x 4 SIMDs, 2.4 GHz, a wave64 VALU instruction occupies its SIMD for 4 clocks
KERNEL_SOURCES = ["mmg_math.h", "mmg_types.h", "gibbs_kernels.h", "sell_kernels.h", "k1.hip"]


def kernel_hash():
    """Hash of the K1 kernel sources with comments and white space removed (what the compiler sees)."""
    import re
    h = hashlib.sha256()
    for f in KERNEL_SOURCES:
        src = open(os.path.join(ROOT, "mmseq_amd", "csrc", f)).read()
        src = re.sub(r"/\*.*?\*/", "", src, flags=re.S)
        src = re.sub(r"//[^\n]*", "", src)
        h.update(re.sub(r"\s+", "", src).encode())
    return h.hexdigest()[:16]


def pmc_counters(rows, transcripts, avg_hits, chains, kernel):
    """Counters per K1 launch from the committed rocprofv3 PMC passes of this same workload AND this same kernel build
    (profiles/pmc_counters.json, written by tools/pmc_summary.py: FETCH_SIZE / WRITE_SIZE / SQ passes collected separately, FETCH
    doubled per the gfx950 correction).  PMC counters cannot be collected from inside this process; any mismatch gives None."""
    try:
        d = json.load(open(os.path.join(ROOT, "profiles", "pmc_counters.json")))
        w = d["workload"]
        if (w["rows"], w["transcripts"], w["avg_hits"], w["chains"]) == (rows, transcripts, avg_hits, chains) \
                and d["kernel"] == kernel and d["kernel_sources_sha16"] == kernel_hash():
            return d
    except Exception:
        pass
    return None


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
    cands = sorted({t for t in (ncpu, ncpu // 2, 64, 32, 16, 8, 1, quota or 1, 2 * (quota or 1)) if 1 <= t <= ncpu}, reverse=True)
    best_t, best_rate = 1, 0.0
    for t in cands:
        r = B.gibbs_ref(p, mu0, seed=args.seed, n_iter=2, trace_len=2, threads=t, want_trace=False)
        rate = Rs * 2 / r["seconds"]
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
        for thr, val in ((0.064, 2), (0.011, 3), (0.0035, 4), (0.002, 6)):
            k[u < thr] = val
        big = u < 0.0012
        k[big] = rng.integers(9, 37, size=int(big.sum())).astype(np.uint32)
        prob = Problem.from_csr(rp, ci, l, k=k, device=device)
        del rp, ci, k, u
    build_s = time.perf_counter() - t0
    inf = prob.info
    mu0, _ = prob.start_values()
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
    out = {"name": name, "reads": inf.m, "transcripts": inf.n, "hits": inf.nnz, "chains": chains, "uniform": bool(uniform),
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
        k1_ms = tm["sample_ms"] / max(tm["sample_launches"], 1) / C   # per launch (chains are advanced by one launch each)
        k2_ms = tm["update_ms"] / max(tm["update_launches"], 1)
        # algorithmic bytes of one K1 launch (one GPU, one chain): u32 row_ptr + u32 col_idx streamed once, fp64 mu read + int32
        # count write (SURVEY 8d / DESIGN.md section 4)
        b_k1 = 4 * (inf.m + 1) + 4 * inf.nnz + 12 * inf.n
        b_sweep = 4 * (inf.m + 1) + 4 * inf.nnz + 28 * C * inf.n
        kname = {0: "k_sample", 2: "k_sample_sell"}[inf.sample_kernel]
        pmc = pmc_counters(args.rows, args.transcripts, args.avg_hits, 1, kname)
        t_k1 = k1_ms * 1e-3
        traffic = (pmc["hbm_read_bytes_per_launch"] + pmc["hbm_write_bytes_per_launch"]) if pmc else None
        # VALU issue passes: every instruction one pass, the quarter-rate Philox multiplies three more (5 per register-path tile)
        passes = (pmc["valu_insts_per_launch"] + 3 * pmc["quarter_rate_valu_per_fast_tile"] * inf.fast_tiles) if pmc and "valu_insts_per_launch" in pmc else None
        ach = passes / t_k1 / 1e9 if passes else None
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
            "roofline": {"bound": "valu", "kernel": kname + " (K1)", "achieved": ach, "peak": VALU_PEAK_GPASS,
                         "unit": "G wave-level VALU issue passes/s", "frac": (ach / VALU_PEAK_GPASS) if ach else None,
                         "traffic": traffic, "hbm_frac": (traffic / t_k1 / 1e9 / HBM_PEAK_GBS) if traffic else None,
                         "algorithmic_bytes_per_launch": b_k1, "algorithmic_x_peak": b_k1 / t_k1 / 1e9 / HBM_PEAK_GBS,
                         "stream_bytes_per_launch": inf.stream_bytes, "stream_frac_of_peak": inf.stream_bytes / t_k1 / 1e9 / HBM_PEAK_GBS,
                         "padded_slots_per_hit": (inf.padded_slots / inf.nnz) if inf.nnz else None,
                         "avg_launch_ms": k1_ms, "timed_launches": tm["sample_launches"] * C,
                         "valu_issue_passes_per_launch": passes,
                         "instructions_per_64_row_tile": ({k[9:].lower(): round(v / max(inf.n_tiles, 1), 1) for k, v in pmc["counters_per_launch"].items()
                                                           if k in ("SQ_INSTS_VALU", "SQ_INSTS_SALU", "SQ_INSTS_BRANCH", "SQ_INSTS_LDS", "SQ_INSTS_VMEM", "SQ_INSTS_SMEM")}
                                                          if pmc else None),
                         "pmc_source": (pmc or {}).get("source"),
                         "note": "bound by instruction issue, not by HBM: a wave issues one instruction per 4-clock slot and a 64-row tile "
                                 "costs ~140 VALU + ~150 scalar/branch + ~30 LDS instructions; frac = counted VALU issue passes / "
                                 "(1024 SIMDs x 2.4 GHz / 4) / time; hbm_frac = PMC HBM bytes / time / 8 TB/s; algorithmic_x_peak = SURVEY "
                                 "8(d) u32-CSR bytes / time / 8 TB/s (> 1: the kernel streams a 1.1 B/hit encoding, not the CSR).  PMC "
                                 "figures are null when profiles/pmc_counters.json was collected on other kernel sources",
                         "k_update_avg_launch_ms": k2_ms, "sweep_bytes": b_sweep},
        }
        if world == 1 and not args.no_extra:
            prob.close()
            prob = None
            torch.cuda.empty_cache()
            extra = []
            R3, T3, H3 = 50_000_000, 200_000, 20.0
            for kw in (dict(name="config 2: 5M reads x 50k transcripts, avg 8 hits, 1 chain", rows=5_000_000, transcripts=50_000, avg_hits=8.0, steps=256, warmup=64),
                       dict(name="config 3: 50M x 200k, 8 chains in one GPU", rows=R3, transcripts=T3, avg_hits=H3, chains=8, steps=16, warmup=4),
                       dict(name="50M x 200k with multiplicities (k > 1 on 6.4 % of the rows, up to 36): two launches per sweep", rows=R3, transcripts=T3, avg_hits=H3,
                            multiplicities=True, steps=32),
                       dict(name="50M x 200k like a real hits file: multiplicities (k > 1 on 6.4 % of the rows) AND 2 % of the rows with a hit anywhere in the "
                                 "transcriptome, 8 chains in one GPU (pairs over the k = 1 register-path tiles, one launch each for the far and the multiplicity tiles)",
                            rows=R3, transcripts=T3, avg_hits=H3, multiplicities=True, far_fraction=0.02, chains=8, steps=16, warmup=4),
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
