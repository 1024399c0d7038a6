#!/usr/bin/env python3
"""bench.py -- Gibbs iterations/sec of the mmseq hot path on MI355X (contract: see the task prompt).

A "step" is one Gibbs sweep: K1 (per-row multinomial allocation + count scatter, the stream of the hit matrix; src/mmseq.cpp:857-891)
+ K2 (Gamma redraw + trace capture, :896-917) over the whole synthetic hit matrix, for every chain on the GPU.
Headline workload = BASELINE.json's 50M-read / 200k-transcript shape (configs[2]/[3]), 1 chain per GPU; the other shapes the
round-1 review asked for (config 2, 8 chains, uniform hits, rows kept in generator order, far-hit mixes) are measured in the same
run and reported in the `extra` block of the same JSON line.

HIP events bracket K1 and K2 on every --time-every-th step INSIDE the timed region, on the stream the kernels are launched on
(default 4; the library attaches them to the launches -- hipExtLaunchKernel, the dispatch's own time stamps -- so that a timed step costs
the stream next to nothing); roofline.avg_launch_ms is the mean over those launches.

roofline (K1 = k_sample_sell, the dominant kernel; the same block for the chain-pair kernel and the EM kernel in `roofline_other`):
  bound        "hbm".  K1 is bound by the stream it reads: with the same stream served from the caches it runs 21 % faster
               (DESIGN.md section 4), and the instruction side (VALU issue) sits right behind it
  traffic      HBM bytes per launch of K1, measured IN THIS RUN at --gpus 1: two short child runs of this script under rocprofv3, one
               per counter (FETCH_SIZE, WRITE_SIZE; --kernel-trace --pmc only), while this process idles behind its timed region
               (roofline.traffic_source says so; --no-live-pmc, no rocprofv3 on PATH, or a failed pass: the committed passes of this
               exact kernel build instead -- profiles/pmc_counters.json, written by tools/pmc_summary.py, stamped with a hash of the
               kernel sources and build flags; null if they changed since).  The other counters (VALU, LDS, wave cycles) and the
               chain-pair / EM blocks always come from the committed passes
  achieved     traffic / avg_launch_ms in GB/s;  peak 8000 GB/s (MI355X_MICROARCH.md);  frac = hbm_counter_frac = achieved / peak
  pattern_read_peak_gbs   what a pure read with K1's access pattern reaches on this part (tools/stream_bench.hip: 6.5-6.7 TB/s)
  algorithmic_x_peak      SURVEY 8(d)'s figure: bytes of the u32 CSR / time / 8 TB/s.  The kernel streams a 1.1 byte-per-hit encoding of
               that CSR, so this exceeds 1 -- it says how much faster than a CSR-streaming kernel at the HBM roofline this is
  valu         counted VALU instructions x 4.0 clocks (tools/issue_bench.hip: what a wave64 VALU instruction of K1's mix occupies its
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
VALU_CLOCKS_PER_INST = 4.0   # tools/issue_bench.hip (profiles/r04_issue_bench.txt; waves grouped by the SIMD they ran on, s_memtime = shader
                             # cycles): fp64 / VOP3 / SDWA / compares / conversions / 32-bit multiplies occupy their SIMD for 4.1-4.2 cycles per
                             # wave64 instruction, plain 32-bit VOP1 / VOP2 2.1-2.4 alone and about 3.6 in a mix with the other class.  Three
                             # quarters of K1's VALU instructions are of the first class: 4.0 on average, which is also what the hardware's own
                             # SQ_ACTIVE_INST_VALU charges (one quad-cycle per instruction)
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


def live_traffic(args, kernel_substr="k_sample_sell<"):
    """HBM bytes per launch of the dominant kernel measured IN THIS RUN: two child runs of this very script (same workload, 8 steps) under
    rocprofv3, one per counter -- FETCH_SIZE and WRITE_SIZE in separate --kernel-trace --pmc passes, as MI355X_MICROARCH.md prescribes
    (TCC slots: both do not fit one pass), FETCH doubled per its gfx950 correction, both in KiB.  PMC counters cannot be read from inside
    a process; children can (started as child processes, program directly behind `--`).  None when rocprofv3 is missing or a pass fails:
    the committed passes of the same kernel build (profiles/pmc_counters.json) then stand in, as before."""
    import csv
    import glob
    import shutil
    import subprocess
    import tempfile
    exe = shutil.which("rocprofv3")
    if not exe or any(k.startswith(("ROCPROF", "ROCP_")) for k in os.environ) or "rocprof" in os.environ.get("LD_PRELOAD", ""):
        return None                                       # no profiler here, or this process itself runs under one
    got = {}
    t0 = time.perf_counter()
    for ctr in ("FETCH_SIZE", "WRITE_SIZE"):
        d = tempfile.mkdtemp(prefix="mmseq_pmc_", dir="/tmp")
        cmd = [exe, "--kernel-trace", "--pmc", ctr, "--output-format", "csv", "-d", d, "--", sys.executable, os.path.abspath(__file__),
               "--no-extra", "--no-cpu-baseline", "--no-live-pmc", "--steps", "8", "--warmup", "2", "--settle-iters", "0", "--rows", str(args.rows),
               "--transcripts", str(args.transcripts), "--avg-hits", str(args.avg_hits), "--seed", str(args.seed)]
        try:
            r = subprocess.run(cmd, cwd="/tmp", env=dict(os.environ, TMPDIR="/tmp"), stdout=subprocess.PIPE, stderr=subprocess.PIPE, timeout=150)
            vals = []
            for f in glob.glob(os.path.join(d, "**", "*counter_collection.csv"), recursive=True):
                for row in csv.DictReader(open(f)):
                    if kernel_substr in row["Kernel_Name"] and row["Counter_Name"] == ctr:
                        vals.append(float(row["Counter_Value"]))
            if r.returncode != 0 or len(vals) < 4:
                return None
            got[ctr] = (sum(vals) / len(vals), len(vals))
        except Exception:
            return None
        finally:
            shutil.rmtree(d, ignore_errors=True)
    return {"hbm_read_bytes_per_launch": 2 * 1024 * got["FETCH_SIZE"][0], "hbm_write_bytes_per_launch": 1024 * got["WRITE_SIZE"][0],
            "launches": got["FETCH_SIZE"][1], "seconds": time.perf_counter() - t0,
            "source": "this run: rocprofv3 --kernel-trace --pmc FETCH_SIZE | WRITE_SIZE, two child passes of bench.py over the same workload "
                      "(KiB; FETCH doubled per MI355X_MICROARCH.md)"}


def roofline_block(kernel_name, entry, t_s, launches, n_tiles, stream_bytes=None, algorithmic_bytes=None, live=None):
    """The roofline object of one kernel: t_s = average launch duration in seconds (HIP events in this run), PMC figures from `entry`;
    live: HBM bytes per launch measured in this run (live_traffic), which then take the place of the committed passes' bytes."""
    pmc = pmc_entry(entry)
    c = (pmc or {}).get("counters_per_launch", {})
    traffic = ((pmc["hbm_read_bytes_per_launch"] or 0) + (pmc["hbm_write_bytes_per_launch"] or 0)) if pmc and pmc.get("hbm_read_bytes_per_launch") else None
    committed_traffic = traffic
    if live:
        traffic = live["hbm_read_bytes_per_launch"] + live["hbm_write_bytes_per_launch"]
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
    if live:
        out["traffic_source"] = live["source"]
        out["traffic_live_passes_s"] = live["seconds"]
        out["traffic_committed_passes"] = committed_traffic
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


def cpu_baseline(args, prob, mu0, device=0):
    """Oracle ("port" of src/mmseq.cpp:851-918, reference-structured: per-thread MT19937, count slabs of n x threads ints cleared and
    reduced every iteration, conditional-binomial multinomial) timed on this host's cores on the FULL problem: the device CSR is
    downloaded (stored order: sorted rows, which only helps the CPU's caches) and `--cpu-iters` iterations run at the best thread count;
    the thread count is picked on a 2 M-row sample, where the single-thread figure is measured too."""
    from oracle import binding as B
    inf = prob.info
    rp, ci = prob.download()
    l = prob.l()
    p = B.Problem(rp, ci, l)
    Rs = min(args.cpu_sample_rows, inf.m)
    ps = B.Problem(rp[:Rs + 1].copy(), ci[:int(rp[Rs])].copy(), l)
    ncpu = os.cpu_count() or 1
    quota = cpu_quota()
    # the reference's per-thread count slabs (src/mmseq.cpp:850-855, :896-899) stop scaling at high thread counts, and thread counts
    # far above a container's CPU quota are throttled erratically: candidates stop at twice the quota
    cands = sorted({t for t in (ncpu, ncpu // 2, 64, 32, 16, 8, quota or 1, 2 * (quota or 1))
                    if 1 <= t <= ncpu and (quota is None or t <= 2 * quota)}, reverse=True)
    best_t, best_rate = 1, 0.0
    for t in cands:
        r = B.gibbs_ref(ps, mu0, seed=args.seed, n_iter=4, trace_len=4, threads=t, want_trace=False)
        rate = Rs * 4 / r["seconds"]
        if rate > best_rate:
            best_t, best_rate = t, rate
    iters = args.cpu_iters
    r = B.gibbs_ref(p, mu0, seed=args.seed, n_iter=iters, trace_len=iters, threads=best_t, want_trace=False)
    it_s = iters / r["seconds"]
    r1 = B.gibbs_ref(ps, mu0, seed=args.seed, n_iter=2, trace_len=2, threads=1, want_trace=False)
    # SURVEY 8(d) also asks for the config-2 shape (5 M reads x 50 k transcripts, 1 + Poisson(7) hits) for >= 64 iterations at all cores
    # and at one thread (there: 8 iterations, 64 would take half a minute)
    from mmseq_amd import Problem
    p2 = Problem.synthetic(5_000_000, 50_000, 8.0, seed=args.seed, mapped_reads=5_000_000, device=device)
    mu2, _ = p2.start_values()
    rp2, ci2 = p2.download()
    q2 = B.Problem(rp2, ci2, p2.l())
    p2.close()
    c2 = B.gibbs_ref(q2, mu2, seed=args.seed, n_iter=64, trace_len=64, threads=best_t, want_trace=False)
    c2_1 = B.gibbs_ref(q2, mu2, seed=args.seed, n_iter=8, trace_len=8, threads=1, want_trace=False)
    cfg2 = {"iterations_per_sec": 64 / c2["seconds"], "iterations": 64, "cores": best_t,
            "single_thread_iterations_per_sec": 8 / c2_1["seconds"], "single_thread_iterations": 8}
    return {"value": it_s, "unit": "iterations/s", "cores": best_t, "kind": "port", "cfg2": cfg2,
            "sample": "%d rows (the full problem, %d hits), %d iterations, %d threads (best of %s; %d CPUs, quota %s)"
                      % (inf.m, inf.nnz, iters, best_t, cands, ncpu, quota),
            "reads_iters_per_sec": it_s * inf.m, "seconds": r["seconds"],
            "single_thread_iterations_per_sec": Rs * 2 / r1["seconds"] / inf.m,
            "single_thread_sample": "first %d stored rows, 2 iterations, scaled by rows" % Rs}


def side_measurement(name, rows, transcripts, avg_hits, chains=1, uniform=False, sort=True, far_fraction=0.0, steps=48, warmup=8,
                     seed=1234, device=0, note="", multiplicities=False, scatter=False, genes=None, families=False):
    """One more workload, same protocol (inputs resident, HIP events on the launch stream), shorter: ms per sweep and which kernel ran.
    multiplicities: the rows get a k array with the distribution a collapsed 50 M-read file of this generator has (93.6 % k = 1,
    5.3 % k = 2, ... 0.12 % k in 9..36: tools/collapse_probe.py) -- what every real hits file produces (src/mmseq.cpp:409-418)."""
    import numpy as np
    import torch
    from mmseq_amd import Problem, Sampler
    from mmseq_amd import dist as mdist
    t0 = time.perf_counter()
    prob = Problem.synthetic(rows, transcripts, avg_hits, seed=seed, uniform=uniform, sort=sort and not genes, far_fraction=far_fraction,
                             mapped_reads=rows, device=device, gene_size=genes[0] if genes else 0, far_family=genes[1] if genes else 0)
    if genes:
        # an aligner's output (generator gene-block mode): a read's hits are isoforms of one gene, far hits go to a gene of the read's
        # paralogue family -- uploaded as the CLI uploads a hits file: rows in generator order, tx_order = gene << 32 | transcript, the
        # genes in the caller's (name) order, which puts a family's members anywhere.  Spec version 7: the library reorders the genes.
        rp, ci = prob.download()
        l = prob.l()
        prob.close()
        t_ids = np.arange(transcripts, dtype=np.uint64)
        tx_order = ((t_ids // np.uint64(genes[0])) << np.uint64(32)) | t_ids
        fam_info = None
        if families:
            # paralogue families with a power-law size distribution (32 ... 5 000 transcripts, members scattered over the caller's gene order,
            # a read's second gene a NEIGHBOUR in its family) and 1 % of the reads on 50 hub transcripts: tools/families.py
            sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), "tools"))
            import families as fam
            rp, ci, tx_order, fam_info = fam.power_law_families(rp, ci, transcripts, genes[0], seed=seed)
        t1 = time.perf_counter()
        prob = Problem.from_csr(rp, ci, l, device=device, tx_order=tx_order)
        scatter_create_s = time.perf_counter() - t1
        del rp, ci
    if scatter:
        # the reference's first-seen numbering (src/mmseq.cpp:399-408): the transcripts get random ids, the rows come in generator order,
        # and NO tx_order is passed -- the library has to find the locality itself (spec version 6: an order derived from the hit graph)
        rp, ci = prob.download()
        l = prob.l()
        prob.close()
        perm = np.random.default_rng(seed).permutation(transcripts).astype(np.uint32)
        l_ext = np.empty_like(l)
        l_ext[perm] = l
        t1 = time.perf_counter()
        prob = Problem.from_csr(rp, perm[ci], l_ext, device=device)
        scatter_create_s = time.perf_counter() - t1
        del rp, ci
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
        elif multiplicities == "k1000":
            k = np.full(rows, 1000, np.uint32)     # every row on the conditional-binomial chain: k_sample_bigk alone, device full
        elif multiplicities == "zipf":
            # what a 50 M-read file collapses to (src/mmseq.cpp:409-440): few million hit sets, k Pareto/Zipf with exponent 1.92 -- 47 % of
            # the hit sets k = 1, 2.3 % above 64, 0.55 % above 304, the tail capped at 10^6 -- about 50 M reads in all
            k = np.minimum(1e6, np.floor((1.0 - u) ** (-1.0 / 0.92))).astype(np.uint32)
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
    # a stream of torch's that is not the legacy NULL stream: beside it the library runs the launch of the rows on the
    # conditional-binomial chain on a side stream of the sampler (sampler.hip), as it does beside a sampler's own stream (the CLI)
    with torch.cuda.stream(torch.cuda.Stream()):
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
    ms = el / steps * 1e3
    # the u32-CSR bytes of one sweep over the STORED problem for C chains, against 8 TB/s
    b_sweep = 4 * (inf.m + 1) + 4 * inf.nnz + 28 * chains * inf.n
    out = {"id": name, "reads": inf.m, "total_k": inf.total_k, "transcripts": inf.n, "hits": inf.nnz, "chains": chains,
           "steps": steps, "ms_per_step": ms, "chain_it_s": chains * steps / el,
           "k1_ms_all_chains": tm["sample_ms"] / max(tm["sample_launches"], 1),
           "k2_ms": tm["update_ms"] / max(tm["update_launches"], 1),
           "kernel": {0: "k_sample", 2: "k_sample_sell"}[inf.sample_kernel],
           "fast_tiles": (inf.fast_tiles / inf.n_tiles) if inf.sample_kernel == 2 and inf.n_tiles else 0.0,
           "far_tiles": (inf.far_tiles / inf.n_tiles) if inf.sample_kernel == 2 and inf.n_tiles else 0.0,
           "stream_bytes": inf.stream_bytes, "n_tiles": inf.n_tiles, "alg_bytes": b_sweep, "alg_frac": b_sweep / (ms * 1e-3) / 1e9 / HBM_PEAK_GBS,
           "build_s": round(build_s, 2)}
    if scatter or genes:
        out["tx_renumbered"] = inf.tx_renumbered
        out["create_s"] = round(scatter_create_s, 2)
    if genes and families:
        out["families"] = fam_info
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
    out = {"id": "em", "ms_per_sweep": ms, "sweeps": sweeps, "stream_kernel": em.stats_raw()["stream_kernel"],
           "n_tiles": prob.info.n_tiles, "stream_bytes": prob.info.stream_bytes}
    em.close()
    prob.close()
    return out


# the side measurements of the default run: id -> (what it is, arguments).  BASELINE.json configs[1] = "cfg2", configs[2] = "cfg3x8".
R3, T3, H3 = 50_000_000, 200_000, 20.0
SIDE = [
    ("cfg2", "BASELINE configs[1]: 5M reads x 50k transcripts, avg 8 hits, 1 chain", dict(rows=5_000_000, transcripts=50_000, avg_hits=8.0, steps=256, warmup=64)),
    ("cfg3x8", "BASELINE configs[2]: 50M x 200k, 8 chains in one GPU", dict(rows=R3, transcripts=T3, avg_hits=H3, chains=8, steps=16, warmup=4)),
    ("mult", "50M x 200k, multiplicities of a collapsed file (k > 1 on 6.4 % of the rows, up to 36; stored k times)", dict(rows=R3, transcripts=T3, avg_hits=H3, multiplicities=True, steps=32)),
    ("real8", "50M x 200k like a real hits file: those multiplicities + 2 % far rows, 8 chains", dict(rows=R3, transcripts=T3, avg_hits=H3, multiplicities=True, far_fraction=0.02, chains=8, steps=16, warmup=4)),
    ("heavy", "a heavily collapsed file: 5M hit sets, multiplicities 1..20000 (total_k reads)", dict(rows=5_000_000, transcripts=T3, avg_hits=H3, multiplicities="heavy", steps=32)),
    ("collapsed", "what a 50M-read file collapses to: 2M hit sets, 1 + Poisson(9) hits, k Zipf (exponent 1.92) up to 10^6, about 50M reads in all", dict(rows=2_000_000, transcripts=T3, avg_hits=10.0, multiplicities="zipf", steps=32)),
    ("bigk", "the conditional-binomial chain alone: 2M hit sets x 20 hits, k = 1000 on every one (38 M binomials per sweep)", dict(rows=2_000_000, transcripts=T3, avg_hits=H3, multiplicities="k1000", steps=32)),
    ("far2", "50M x 200k, 2 % of the rows with a hit anywhere in the transcriptome", dict(rows=R3, transcripts=T3, avg_hits=H3, far_fraction=0.02, steps=32)),
    ("far20", "50M x 200k, 20 % of the rows with a hit anywhere", dict(rows=R3, transcripts=T3, avg_hits=H3, far_fraction=0.2, steps=24)),
    ("gene0", "50M x 200k like an aligner's output: a read's hits are isoforms of ONE gene (32 isoforms per gene), tx_order = the caller's genes",
     dict(rows=R3, transcripts=T3, avg_hits=H3, genes=(32, 3), steps=32)),
    ("far20p", "the same with 20 % of the reads also hitting a gene of their paralogue family (3 genes, anywhere in the caller's gene order): the library reorders the genes",
     dict(rows=R3, transcripts=T3, avg_hits=H3, genes=(32, 3), far_fraction=0.2, steps=32)),
    ("families_pl", "the same gene blocks with paralogue families of power-law size (32 ... 5 000 transcripts, scattered over the caller's gene order; 20 % of their reads also hit a neighbouring member) and 1 % of the reads on 50 hub transcripts",
     dict(rows=R3, transcripts=T3, avg_hits=H3, genes=(32, 0), families=True, steps=32)),
    ("uniform", "50M x 200k, hits uniform over all transcripts (SURVEY App. D worst case)", dict(rows=R3, transcripts=T3, avg_hits=H3, uniform=True, steps=8, warmup=2)),
    ("keeprows", "50M x 200k, rows kept in generator order (MMG_LAYOUT_KEEP_ROWS)", dict(rows=R3, transcripts=T3, avg_hits=H3, sort=False, steps=8, warmup=2)),
    ("scatter", "50M x 200k, transcripts numbered at random (first-seen numbering), rows in generator order, NO tx_order: order derived from the hit graph",
     dict(rows=R3, transcripts=T3, avg_hits=H3, sort=False, scatter=True, steps=32)),
]


def compact(d, drop=()):
    """JSON-line diet: floats to 5 significant digits, the given keys dropped (the driver keeps an 8 KB tail of stdout)."""
    out = {}
    for k, v in d.items():
        if k in drop or v is None:
            continue
        if isinstance(v, float):
            v = float("%.5g" % v)
        elif isinstance(v, dict):
            v = compact(v)
        out[k] = v
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
                    "(1 = every step)")
    ap.add_argument("--settle-iters", type=int, default=256, help="iterations of a scratch chain before the warm-up steps (GPU clock ramp)")
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--no-extra", action="store_true", help="skip the side measurements")
    ap.add_argument("--only", default="", help="comma-separated ids of the side measurements to run (default: all)")
    ap.add_argument("--cpu-sample-rows", type=int, default=2_000_000, help="rows of the sample the CPU thread count is picked on")
    ap.add_argument("--cpu-iters", type=int, default=8, help="iterations of the CPU baseline on the full problem")
    ap.add_argument("--full-json", default="", help="also write the unabridged record (every field of every measurement) to this file")
    ap.add_argument("--no-live-pmc", action="store_true", help="do not collect FETCH_SIZE / WRITE_SIZE of K1 in child runs under rocprofv3 "
                    "(roofline.traffic then comes from the committed passes of the same kernel build)")
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
    # one rank per GPU; MMSEQ_BENCH_BACKEND=gloo lets several ranks share the devices present (a functional check of the N > 1 paths on a
    # one-GPU box: the collectives then go through the host, the numbers mean nothing)
    backend = os.environ.get("MMSEQ_BENCH_BACKEND", "nccl")
    if backend == "nccl" and local_rank >= torch.cuda.device_count():
        raise SystemExit("rank %d has no GPU of its own (%d visible)" % (local_rank, torch.cuda.device_count()))
    local_rank = local_rank % torch.cuda.device_count()
    torch.cuda.set_device(local_rank)
    if world > 1:
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        if backend == "nccl":
            dist.init_process_group("nccl", device_id=torch.device("cuda", local_rank))
        else:
            dist.init_process_group(backend)

    # ---- workload (synthetic, generated straight into device CSR and laid out by the library; not timed)
    shard_rows = None
    if args.mode == "shard" and world > 1:
        # What `mmseq -gpus N` does (host/mmseq_main.cpp): ONE canonical problem of rows x N reads, cut where the library cuts it
        # (mmg_problem_shard_bounds: contiguous ranges of the stored rows of equal modelled cost), shard `rank` kept.  Every rank builds
        # the whole problem on its own device (the generator is keyed: the same rows everywhere) and keeps its range; no rank gets rows
        # of "its own" generator stream, which would be balanced by construction.
        total_reads = args.rows * world
        full = Problem.synthetic(total_reads, args.transcripts, args.avg_hits, seed=args.seed, mapped_reads=total_reads, device=local_rank)
        mu0, _ = full.start_values()
        # the cut by MEASURED cost (mmg_problem_shard_bounds_timed), taken on rank 0's device and broadcast: the ranks must agree on it
        bt = torch.from_numpy(full.shard_bounds_timed(mu0, world).astype(np.int64) if rank == 0 else np.zeros(world + 1, np.int64)).cuda()
        dist.broadcast(bt, src=0)
        b = bt.cpu().numpy().astype(np.uint64)
        shard_rows = [int(b[i + 1] - b[i]) for i in range(world)]
        prob = full.shard(int(b[rank]), int(b[rank + 1]), device=local_rank)
        full.close()
        del full
        torch.cuda.empty_cache()
    else:
        total_reads = args.rows
        prob = Problem.synthetic(args.rows, args.transcripts, args.avg_hits, seed=args.seed, mapped_reads=total_reads, device=local_rank)
        mu0, _ = prob.start_values()
    inf = prob.info
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
    tm = smp.timing()
    k1_own_ms = tm["sample_ms"] / max(tm["sample_launches"], 1)
    k1_ranks = None
    if world > 1:
        te = torch.tensor([elapsed], dtype=torch.float64, device="cuda")
        dist.all_reduce(te, op=dist.ReduceOp.MAX)
        elapsed = float(te.item())
        tk = torch.zeros(world, dtype=torch.float64, device="cuda")
        tk[rank] = k1_own_ms
        dist.all_reduce(tk)
        k1_ranks = [float(x) for x in tk.cpu()]

    # sanity inside the bench: every read was allocated exactly once in the last sweep
    if args.mode == "chains":
        assert int(smp.counts(0).astype(np.int64).sum()) == inf.total_k
    else:
        assert int(smp.counts(0).astype(np.int64).sum()) == total_reads   # the all-reduced counts: every read of every shard
    smp.close()

    if rank == 0:
        C = args.chains
        chains_total = C * (world if args.mode == "chains" else 1)
        iters_per_s = chains_total * args.steps / elapsed
        k1_ms = k1_own_ms / C                                  # per chain: the sample() call of C chains / C
        k2_ms = tm["update_ms"] / max(tm["update_launches"], 1)
        # algorithmic bytes of one K1 launch (one GPU, one chain): u32 row_ptr + u32 col_idx streamed once, fp64 mu read + int32
        # count write (SURVEY 8d / DESIGN.md section 4)
        b_k1 = 4 * (inf.m + 1) + 4 * inf.nnz + 12 * inf.n
        kname = {0: "k_sample", 2: "k_sample_sell"}[inf.sample_kernel]
        # chains > 1: one event pair brackets the whole sample() call -- pair launches plus the launches for the other tile lists;
        # per-kernel PMC figures belong to the 1-chain run only.  Chains mode at ANY world size runs the kernel and the problem of the
        # 1-GPU run on every GPU (k1_ms is rank 0's own launch time): the same counter entry applies.  A read shard is another problem
        # (fewer tiles per launch): no counter entry, the fraction then comes from the shard's own stream bytes (stream_frac_of_peak).
        entry = "k1_1chain" if C == 1 and args.mode == "chains" and (args.rows, args.transcripts, args.avg_hits) == (50_000_000, 200_000, 20.0) else "none"
        # roofline.traffic measured in THIS run (1 GPU, 1 chain): two short child passes under rocprofv3 while this process idles
        live = None
        if world == 1 and C == 1 and args.mode == "chains" and not args.no_live_pmc and inf.sample_kernel == 2:
            live = live_traffic(args)
        full_roof = roofline_block(kname + " (K1)", entry, k1_ms * 1e-3, tm["sample_launches"], inf.n_tiles,
                                   stream_bytes=inf.stream_bytes, algorithmic_bytes=b_k1, live=live)
        if full_roof["frac"] is None:
            # no counter pass of this exact launch: what the kernel must read (its stream) + the algorithmic mu / count bytes, over its time
            full_roof["traffic_is"] = "stream bytes + 12 n (no PMC pass of this launch shape)"
            full_roof["achieved"] = (inf.stream_bytes + 12 * inf.n) / (k1_ms * 1e-3) / 1e9
            full_roof["frac"] = full_roof["achieved"] / HBM_PEAK_GBS
        if k1_ranks:
            full_roof["k1_ms_per_rank_max"] = max(k1_ranks)
            full_roof["k1_ms_per_rank_mean"] = sum(k1_ranks) / len(k1_ranks)
            full_roof["shard_balance" if args.mode == "shard" else "rank_balance"] = max(k1_ranks) / (sum(k1_ranks) / len(k1_ranks))
        full_roof["padded_slots_per_hit"] = (inf.padded_slots / inf.nnz) if inf.nnz else None
        full_roof["k_update_avg_launch_ms"] = k2_ms
        # the contract's six first, then one scalar per BASELINE config and side measurement (filled in below), then the detail
        roof = {k: full_roof[k] for k in ("bound", "achieved", "peak", "unit", "frac", "traffic", "kernel", "avg_launch_ms")}
        for k in ("shard_balance", "rank_balance", "k1_ms_per_rank_max", "k1_ms_per_rank_mean", "traffic_is", "traffic_source", "traffic_live_passes_s", "traffic_committed_passes"):
            if k in full_roof:
                roof[k] = full_roof[k]
        out = {
            "metric": "gibbs_iterations_per_sec", "value": iters_per_s, "unit": "iterations/s",
            "n_gpus": world, "steps": args.steps, "warmup": args.warmup,
            "ms_per_step": elapsed / args.steps * 1e3, "higher_is_better": True, "scaling": "weak",
            "vs_baseline": None, "dtype": "f64", "data": "synthetic",
            "config": {"workload": ("50M-read x 200k-transcript synthetic CSR hits (BASELINE.json configs[2]/[3] shape)"
                                    if (args.rows, args.transcripts) == (50_000_000, 200_000) else "custom synthetic CSR hits")
                                   + ("" if shard_rows is None else ", x %d GPUs = configs[4]" % world),
                       "reads_per_gpu": inf.m, "transcripts": inf.n, "hits_per_gpu": inf.nnz,
                       "avg_hits_per_read": args.avg_hits, "chains_per_gpu": C, "mode": args.mode,
                       "parallelism": ("%d independent chains (1 all-reduce of posterior moments)" % chains_total)
                       if args.mode == "chains" else ("read-sharded single chain over %d GPUs (int32 count all-reduce per iteration), "
                                                      "one canonical problem cut by measured cost" % world),
                       "trace": "every iteration kept (fp64 mu trace resident in HBM)", "generator_seed": args.seed,
                       "layout": "generator order in, the library's canonical order stored (device radix sort)",
                       "settle_iters": args.settle_iters, "cold_ms_per_step": cold_ms},
            "reads_iters_per_sec": iters_per_s * total_reads,
            "roofline": roof,
        }
        if shard_rows is not None:
            out["config"]["rows_per_shard"] = shard_rows
            out["config"]["k1_ms_per_rank"] = k1_ranks
        record = {"headline": dict(out), "roofline_k1": full_roof}
        if world == 1 and not args.no_cpu_baseline:
            out["cpu_baseline"] = cpu_baseline(args, prob, mu0, device=local_rank)
            out["speedup_vs_cpu_baseline"] = iters_per_s / out["cpu_baseline"]["value"]
        if world == 1 and not args.no_extra:
            prob.close()
            prob = None
            torch.cuda.empty_cache()
            only = [x for x in args.only.split(",") if x]
            side = {}
            for sid, what, kw in SIDE:
                if only and sid not in only:
                    continue
                try:
                    side[sid] = side_measurement(sid, seed=args.seed, device=local_rank, **kw)
                    side[sid]["what"] = what
                except Exception as e:                      # a side measurement must not take the headline down
                    side[sid] = {"id": sid, "error": repr(e)[:120]}
                torch.cuda.empty_cache()
            # the same roofline object for the two other hot kernels: the chain-pair kernel (BASELINE configs[2]) and the EM rows pass
            other = {}
            c3 = side.get("cfg3x8")
            if c3 and "error" not in c3:
                pairs = c3["chains"] // 2
                # ONE launch advances all the chains, pair by pair (grid.y = pair): its duration, its counters, `pairs` passes over the
                # stream and `pairs` x tiles tile-visits (instructions_per_64_row_tile is per visit, i.e. per tile and PAIR of chains)
                rb = roofline_block("k_sample_sell_multi<2> (K1m)", "k1m_8chains", c3["k1_ms_all_chains"] * 1e-3, c3["steps"] // 4 + 1, c3["n_tiles"] * pairs,
                                    stream_bytes=c3["stream_bytes"] * pairs,
                                    algorithmic_bytes=pairs * (4 * (c3["reads"] + 1) + 4 * c3["hits"] + 24 * c3["transcripts"]))
                rb["bound"] = "lds+valu"
                other["k1m"] = rb
            if not only or "em" in only:
                try:
                    em = em_measurement(R3, T3, H3, seed=args.seed, device=local_rank)
                    rb = roofline_block("k_em_sell (K3, one mmg_em_step)", "em_sweep", em["ms_per_sweep"] * 1e-3, em["sweeps"], em["n_tiles"], stream_bytes=em["stream_bytes"])
                    rb["bound"] = "lds"
                    other["em"] = rb
                    side["em"] = em
                except Exception as e:
                    other["em"] = {"kernel": "k_em_sell", "error": repr(e)[:120]}
                torch.cuda.empty_cache()
            # one scalar per configuration, directly under `roofline` (records that keep only an object's leading scalars still carry
            # every BASELINE config): iterations/s, the SURVEY 8(d) fraction (u32-CSR bytes / time / 8 TB/s) and the counter fraction
            g = lambda sid, key: (side.get(sid) or {}).get(key)
            roof["cfg2_it_s"] = g("cfg2", "chain_it_s")
            roof["cfg2_alg_frac"] = g("cfg2", "alg_frac")
            roof["cfg3x8_chain_it_s"] = g("cfg3x8", "chain_it_s")
            roof["cfg3x8_alg_frac"] = g("cfg3x8", "alg_frac")
            roof["cfg3x8_hbm_frac"] = (other.get("k1m") or {}).get("frac")
            roof["em_sweep_ms"] = g("em", "ms_per_sweep")
            roof["em_hbm_frac"] = (other.get("em") or {}).get("frac")
            roof["real8_chain_it_s"] = g("real8", "chain_it_s")
            # the regime every real (collapsed) hits file runs in: hit sets with multiplicities (src/mmseq.cpp:409-440, the draw :880)
            roof["heavy_ms"] = g("heavy", "ms_per_step")
            roof["collapsed_ms"] = g("collapsed", "ms_per_step")
            roof["collapsed_alg_frac"] = g("collapsed", "alg_frac")
            roof["bigk_ms"] = g("bigk", "ms_per_step")
            roof["far20_ms"] = g("far20", "ms_per_step")
            roof["gene0_ms"] = g("gene0", "ms_per_step")
            roof["far20p_ms"] = g("far20p", "ms_per_step")
            roof["far20p_far_tiles"] = g("far20p", "far_tiles")
            roof["families_pl_ms"] = g("families_pl", "ms_per_step")
            roof["families_pl_far_tiles"] = g("families_pl", "far_tiles")
            roof["uniform_ms"] = g("uniform", "ms_per_step")
            roof["keeprows_ms"] = g("keeprows", "ms_per_step")
            roof["scatter_no_tx_order_ms"] = g("scatter", "ms_per_step")
            roof["algorithmic_x_peak"] = full_roof["algorithmic_x_peak"]
            # (the driver keeps an 8 KB tail of stdout: the line stays under 7.5 KB -- which kernel ran and the reads are in --full-json's record)
            keep = ("id", "chains", "ms_per_step", "chain_it_s", "k1_ms_all_chains", "k2_ms", "alg_frac", "far_tiles", "error", "ms_per_sweep",
                    "tx_renumbered", "create_s")
            roof["configs"] = [compact({k: v for k, v in r.items() if k in keep}) for r in side.values()]
            roof["other"] = {k: compact(v, drop=("pmc_source", "pattern_read_peak_gbs", "peak", "unit", "timed_launches", "hbm_counter_frac")) for k, v in other.items()}
            record["side"] = side
            record["roofline_other"] = other
        for k in ("hbm_counter_frac", "frac_of_pattern_peak", "effective_clock_ghz", "stream_bytes_per_launch", "algorithmic_bytes_per_launch",
                  "padded_slots_per_hit", "k_update_avg_launch_ms", "valu", "lds", "wave_cycles", "instructions_per_64_row_tile", "pmc_source"):
            roof.setdefault(k, full_roof.get(k))
        roof.setdefault("algorithmic_x_peak", full_roof["algorithmic_x_peak"])
        line = json.dumps(compact(out))
        if args.full_json:
            record["line"] = out
            with open(args.full_json, "w") as f:
                json.dump(record, f, indent=1)
        print(line, flush=True)
    if prob is not None:
        prob.close()
    if world > 1:
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
