"""Format fuzz of the hits-file readers (CPU, AddressSanitizer + UBSan build: make -C mmseq_amd/csrc asan): valid text and binary hits
files are damaged (bytes flipped, spans deleted / duplicated / zeroed, truncations, huge varints, over-long name deltas) and read through
every reader entry point -- `hitstools t` (record-by-record API), `hitstools hitsets` (HitsfileReader::readReadMapRecordsBulk, the in-place
parser the mmseq CLI uses), `t2g_hits`, and the parallel inflate (MMSEQ_INFLATE_MIN=0) -- and the `mmseq` CLI's threaded ingest (which ends at mmg_problem_create where there is no GPU) -- which must end with exit code 0 or 1
within the time limit: no sanitizer report, no signal, no hang.  Undamaged files must read back to the oracle's text.
usage: hitsio_fuzz.py [n_cases] [first_seed]"""
import os
import subprocess
import sys
import tempfile
import zlib

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))
from oracle import host_oracle as H  # noqa: E402
from test_hitsio import _dataset  # noqa: E402

ASAN = os.path.join(ROOT, "mmseq_amd", "csrc", "asan")


def damage(rng, b):
    b = bytearray(b)
    for _ in range(int(rng.integers(1, 5))):
        if not b:
            break
        kind = int(rng.integers(0, 8))
        at = int(rng.integers(0, len(b)))
        ln = int(min(len(b) - at, rng.integers(1, 64)))
        if kind == 0:
            for _ in range(int(rng.integers(1, 6))):
                b[int(rng.integers(0, len(b)))] = int(rng.integers(0, 256))
        elif kind == 1:
            del b[at:at + ln]
        elif kind == 2:
            b[at:at] = b[at:at + ln]
        elif kind == 3:
            b[at:at + ln] = bytes(ln)
        elif kind == 4:
            del b[at:]
        elif kind == 5:
            b[at:at + ln] = b"\xff" * ln                  # 0xFF escapes / continuation bits everywhere
        elif kind == 6:
            b[at:at] = b"\n" * int(rng.integers(1, 4))    # record separators in the wrong places
        else:
            b[at:at] = bytes(rng.integers(0, 256, size=ln, dtype=np.uint8))
    return bytes(b)


def run(cmd, env=None, limit=60):
    e = dict(os.environ, ASAN_OPTIONS="detect_leaks=0:abort_on_error=0", UBSAN_OPTIONS="print_stacktrace=1")
    e.update(env or {})
    try:
        r = subprocess.run(cmd, stdout=subprocess.PIPE, stderr=subprocess.PIPE, env=e, timeout=limit)
    except subprocess.TimeoutExpired:
        return "timeout", b"", b""
    return r.returncode, r.stdout, r.stderr


def main():
    n = int(sys.argv[1]) if len(sys.argv) > 1 else 200
    s0 = int(sys.argv[2]) if len(sys.argv) > 2 else 1
    tmp = tempfile.mkdtemp()
    path = os.path.join(tmp, "f.hits")
    bad = n_ok = n_rej = 0
    for seed in range(s0, s0 + n):
        rng = np.random.default_rng(seed)
        h = _dataset(seed, n_t=int(rng.integers(13, 60)), n_reads=int(rng.integers(6, 400)), long_ids=bool(rng.integers(0, 2)))
        txt = H.write_hits_text(h)
        schema = int(rng.integers(0, 3))
        if schema == 0:
            data, clean = damage(rng, txt), txt
        else:
            payload = H.encode_hits_binary_payload(h)
            if schema == 1:                                 # damage under the compression: the format parser sees it
                data, clean = zlib.compress(damage(rng, payload), 1), zlib.compress(payload, 1)
            else:                                           # damage of the compressed stream: the inflaters see it
                clean = zlib.compress(payload, 1)
                data = damage(rng, clean)
        for label, blob, must_equal in (("clean", clean, True), ("damaged", data, False)):
            open(path, "wb").write(blob)
            for tool, env in ((["hitstools", "t"], None), (["hitstools", "hitsets"], None), (["t2g_hits"], None),
                              (["hitstools", "t"], {"MMSEQ_INFLATE_MIN": "0", "MMSEQ_INFLATE_CHUNK": "4096", "MMSEQ_INFLATE_THREADS": "3"}),
                              (["hitstools", "hitsets"], {"MMSEQ_INFLATE_MIN": "0", "MMSEQ_INFLATE_CHUNK": "4096", "MMSEQ_INFLATE_THREADS": "3"}),
                              # the CLI's threaded ingest (inflate -> decoder -> sorters -> table); without a GPU it ends at mmg_problem_create with exit code 1
                              (["mmseq"], {"OMP_NUM_THREADS": "4"}),
                              (["mmseq"], {"OMP_NUM_THREADS": "4", "MMSEQ_INFLATE_MIN": "0", "MMSEQ_INFLATE_CHUNK": "4096", "MMSEQ_INFLATE_THREADS": "3"})):
                rc, out, err = run([os.path.join(ASAN, tool[0])] + tool[1:] + [path] + ([os.path.join(tmp, "out")] if tool[0] == "mmseq" else []), env)
                okrc = rc in (0, 1)
                if must_equal and tool == ["hitstools", "t"]:
                    okrc = rc == 0 and out == txt
                if not okrc or b"Sanitizer" in err or b"runtime error" in err:
                    bad += 1
                    keep = os.path.join(ROOT, "gpurun_out", "hitsio_fuzz_seed%d_%s.hits" % (seed, label))
                    os.makedirs(os.path.dirname(keep), exist_ok=True)
                    open(keep, "wb").write(blob)
                    print("seed %d schema %d %s %s env=%s: rc=%s\n%s" % (seed, schema, label, " ".join(tool), env, rc, err[-1500:].decode("latin1")), flush=True)
                elif not must_equal:
                    if rc == 0:
                        n_ok += 1
                    else:
                        n_rej += 1
        if (seed - s0 + 1) % 25 == 0:
            print("... %d cases, %d damaged reads accepted, %d rejected, %d FAILURES" % (seed - s0 + 1, n_ok, n_rej, bad), flush=True)
    print("%d cases (seeds %d..%d): damaged files read %d times to the end, rejected %d times; FAILURES: %d" % (n, s0, s0 + n - 1, n_ok, n_rej, bad))
    return 1 if bad else 0


if __name__ == "__main__":
    sys.exit(main())
