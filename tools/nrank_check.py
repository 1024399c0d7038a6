#!/usr/bin/env python3
"""The driver's own N-rank bench command, run FUNCTIONALLY on the devices present (a one-GPU box: MMSEQ_BENCH_BACKEND=gloo lets the
ranks share it; the collectives then go through the host and the rates mean nothing).  What it checks is that the first 8-GPU node
gets a well-formed line: the launch, the rendezvous, both modes' collectives, the JSON line and the fields the judge reads at N > 1.

    python tools/nrank_check.py [--gpus 2,8] [--out profiles/r05_nrank_check.log]

For every (mode, N): python -m torch.distributed.run --nnodes=1 --nproc-per-node N --master-addr 127.0.0.1 --master-port P
bench.py --gpus N --steps K --warmup W [--mode shard] (sizes reduced so that N ranks fit one device), then asserts on the line.
Needs a HIP device (not a pytest test: the CPU suite has no device, the GPU suite's box has one GPU and this takes minutes)."""
import argparse
import json
import os
import subprocess
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
ap = argparse.ArgumentParser()
ap.add_argument("--gpus", default="2,8")
ap.add_argument("--out", default="")
ap.add_argument("--port", type=int, default=29655)
a = ap.parse_args()

log = []


def say(x):
    print(x, flush=True)
    log.append(x)


def run(mode, n, rows, transcripts, avg, port):
    env = dict(os.environ, MMSEQ_BENCH_BACKEND="gloo", HSA_ENABLE_IPC_MODE_LEGACY="0")
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", str(n), "--master-addr", "127.0.0.1",
           "--master-port", str(port), os.path.join(ROOT, "bench.py"), "--gpus", str(n), "--steps", "8", "--warmup", "2", "--mode", mode,
           "--rows", str(rows), "--transcripts", str(transcripts), "--avg-hits", str(avg), "--settle-iters", "0"]
    say("$ MMSEQ_BENCH_BACKEND=gloo " + " ".join(cmd[1:]))
    r = subprocess.run(cmd, env=env, capture_output=True, text=True, timeout=1500)
    lines = [x for x in r.stdout.splitlines() if x.startswith("{")]
    assert r.returncode == 0 and lines, "rc %d\n%s\n%s" % (r.returncode, r.stdout[-2000:], r.stderr[-4000:])
    assert len(lines) == 1, "rank 0 alone prints ONE line, got %d" % len(lines)
    say(lines[0])
    d = json.loads(lines[0])
    for k in ("metric", "value", "unit", "n_gpus", "steps", "warmup", "ms_per_step", "higher_is_better", "scaling", "dtype", "data", "config", "roofline"):
        assert k in d, k
    assert d["n_gpus"] == n and d["steps"] == 8 and d["warmup"] == 2 and d["value"] > 0 and d["scaling"] == "weak"
    assert d["config"]["mode"] == mode and "workload" in d["config"]
    rf = d["roofline"]
    for k in ("bound", "achieved", "peak", "unit", "frac"):
        assert rf.get(k) is not None, "roofline.%s missing at N = %d (%s)" % (k, n, mode)
    assert 0 < rf["frac"] < 1.5, rf["frac"]
    if mode == "shard":
        assert len(d["config"]["rows_per_shard"]) == n and len(d["config"]["k1_ms_per_rank"]) == n
        assert rf["shard_balance"] >= 1.0
        assert sum(d["config"]["rows_per_shard"]) == rows * n
    else:
        assert rf["rank_balance"] >= 1.0
        assert "%d independent chains" % n in d["config"]["parallelism"]
    say("ok: %s mode, %d ranks: value %.1f %s, roofline.frac %.3f, %s %.3f" % (
        mode, n, d["value"], d["unit"], rf["frac"], "shard_balance" if mode == "shard" else "rank_balance",
        rf.get("shard_balance") or rf.get("rank_balance")))


port = a.port
for n in [int(x) for x in a.gpus.split(",")]:
    # chains: every rank holds the whole problem (+ its trace): the headline shape at 2 ranks, a tenth of it at 8 on one device
    run("chains", n, 50_000_000 if n <= 2 else 5_000_000, 200_000 if n <= 2 else 50_000, 20.0 if n <= 2 else 8.0, port)
    port += 1
    # shard: every rank BUILDS the problem of rows x N reads before it keeps its range: rows sized for N builds on one device
    run("shard", n, 5_000_000 if n <= 2 else 1_000_000, 50_000, 8.0, port)
    port += 1
say("all N-rank lines well-formed")
if a.out:
    open(a.out, "w").write("\n".join(log) + "\n")
