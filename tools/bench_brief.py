#!/usr/bin/env python3
"""Print the headline and the extra block of a bench.py JSON line in a few lines (for gpurun tails)."""
import json, sys
lines = [l for l in open(sys.argv[1]) if l.startswith("{")]
if not lines:
    print("no JSON line in", sys.argv[1]); sys.exit(1)
d = json.loads(lines[-1])
r = d.get("roofline", {})
print("value %.1f it/s  ms_per_step %.4f  K1 %.4f ms  K2 %.4f ms" % (d["value"], d["ms_per_step"], r.get("avg_launch_ms") or 0, r.get("k_update_avg_launch_ms") or 0))
for e in d.get("extra", []):
    if "error" in e:
        print("  %-70s ERROR %s" % (e["name"][:70], e["error"][:80]))
    else:
        print("  %-70s %.4f ms/step  %8.1f chain-it/s  K1 %.4f" % (e["name"][:70], e["ms_per_step"], e["chain_iterations_per_sec"], e["k1_avg_launch_ms_all_chains"]))
