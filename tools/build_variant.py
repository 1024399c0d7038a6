"""Builds an experimental variant of libmmgibbs.so without touching the tree: copies mmseq_amd/csrc + include to build_ab/<name>/,
applies the given textual substitutions (file::old::new, old must occur), builds there and leaves build_ab/lib_<name>.so.
Used with MMSEQ_AMD_LIB=build_ab/lib_<name>.so (mmseq_amd/_lib.py) for A/B timing on one box: tools/k1_ab.py, tools/shard_probe.py.
usage: build_variant.py name [file::old::new ...] [@patches.py]      (patches.py defines P = [(file, old, new), ...])"""
import os, shutil, subprocess, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
name, subs = sys.argv[1], sys.argv[2:]
dst = os.path.join(ROOT, "build_ab", name)
shutil.rmtree(dst, ignore_errors=True)
os.makedirs(os.path.join(dst, "mmseq_amd"))
shutil.copytree(os.path.join(ROOT, "mmseq_amd", "csrc"), os.path.join(dst, "mmseq_amd", "csrc"), ignore=shutil.ignore_patterns("*.o", "*.o.*", "*.so", "asan", "huffenc_test", "mmseq", "hitstools", "t2g_hits", "synth_hits", "test_group"))
shutil.copytree(os.path.join(ROOT, "include"), os.path.join(dst, "include"))
triples = []
for sub in subs:
    if sub.startswith("@"):
        ns = {}
        exec(open(sub[1:]).read(), ns)
        triples += list(ns["P"])
    else:
        triples.append(tuple(sub.split("::", 2)))
for f, old, new in triples:
    p = os.path.join(dst, "mmseq_amd", "csrc", f)
    s = open(p).read()
    assert old in s, "%s: %r not found" % (f, old)
    open(p, "w").write(s.replace(old, new))
subprocess.check_call(["make", "-j8", "-C", os.path.join(dst, "mmseq_amd", "csrc"), "libmmgibbs.so"], stdout=subprocess.DEVNULL)
out = os.path.join(ROOT, "build_ab", "lib_%s.so" % name)
shutil.copy(os.path.join(dst, "mmseq_amd", "csrc", "libmmgibbs.so"), out)
shutil.rmtree(dst)
print(out)
