#!/usr/bin/env python3
"""Static facts of the sample kernels from a hipcc -S listing: registers, scratch, load and wait mix.

    hipcc --offload-arch=gfx950 -O3 -std=c++17 -ffp-contract=off -munsafe-fp-atomics -S --cuda-device-only -o k1.s mmseq_amd/csrc/k1.hip
    python3 tools/isa_stats.py k1.s [name-substring]
"""
import re
import sys
from collections import Counter

s = open(sys.argv[1]).read()
want = sys.argv[2] if len(sys.argv) > 2 else "k_sample_sell"
for m in re.finditer(r"\n(_Z\w+):[^\n]*\n(.*?)\n\.Lfunc_end\d+:", s, re.S):
    name, body = m.group(1), m.group(2)
    if want not in name:
        continue
    def sym(k):
        r = re.search(re.escape(name) + r"\." + k + r", (\d+)", s)
        return r.group(1) if r else "?"
    w = Counter(re.findall(r"s_waitcnt vmcnt\((\d+)\)", body))
    cnt = lambda pat: len(re.findall(pat, body))
    print(name[:48], "vgpr", sym("num_vgpr"), "sgpr", sym("numbered_sgpr"), "scratch", sym("private_seg_size"),
          "| buffer_load", cnt(r"\n\tbuffer_load"), "global_load", cnt(r"\n\tglobal_load"), "s_load", cnt(r"\n\ts_load"),
          "| static s_", cnt(r"\n\ts_"), "v_", cnt(r"\n\tv_"), "ds_", cnt(r"\n\tds_"),
          "| vmcnt waits", sorted(w.items(), key=lambda x: -x[1])[:8])
