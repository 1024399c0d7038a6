"""CPU checks of the drop-in boundary: libmmgibbs.so loads, exports every symbol include/mmgibbs.h
declares, refuses to compute without a device (no fallback), and the host instantiation of its
inline math/RNG equals the oracle bit for bit."""
import ctypes as C
import os
import re

import numpy as np
import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _declared_symbols():
    txt = open(os.path.join(ROOT, "include", "mmgibbs.h")).read()
    txt = re.sub(r"/\*.*?\*/", "", txt, flags=re.S)
    return sorted(set(re.findall(r"\b(mmg_[a-z0-9_]+)\s*\(", txt)))


def test_library_exports_every_declared_symbol():
    from mmseq_amd import _lib
    lib = _lib.load()
    declared = _declared_symbols()
    assert len(declared) >= 30
    for name in declared:
        assert hasattr(lib, name), "libmmgibbs.so does not export %s" % name
    assert sorted(_lib.SYMBOLS) == declared, "python binding table out of sync with include/mmgibbs.h"
    assert lib.mmg_abi_version() == 8


def test_struct_layouts_match_header():
    from mmseq_amd import _lib
    assert C.sizeof(_lib.ProblemDesc) == 72
    assert C.sizeof(_lib.SynthDesc) == 72
    assert C.sizeof(_lib.Config) == 48
    assert C.sizeof(_lib.Timing) == 32
    assert C.sizeof(_lib.ProblemInfo) == 112


def test_no_cpu_fallback_without_device():
    from mmseq_amd import gibbs
    from mmseq_amd._lib import MMGError
    if gibbs.device_count() > 0:
        pytest.skip("a HIP device is present")
    with pytest.raises(MMGError) as e:
        gibbs.Problem.from_csr(np.array([0, 1], np.uint64), np.array([0], np.uint32), np.ones(1))
    assert e.value.code == 2 and "no CPU fallback" in str(e.value)
    with pytest.raises(MMGError):
        gibbs.Problem.synthetic(100, 10, 4)
    with pytest.raises(MMGError):
        gibbs.selftest_math(np.ones(4), 0)


def test_argument_validation_happens_before_device_use():
    from mmseq_amd import gibbs
    from mmseq_amd._lib import MMGError
    with pytest.raises(MMGError) as e:
        gibbs.Problem.from_csr(np.array([0, 2], np.uint64), np.array([0, 7], np.uint32), np.ones(3))
    assert e.value.code == 1          # col index out of range
    with pytest.raises(MMGError) as e:
        gibbs.Problem.from_csr(np.array([0, 1], np.uint64), np.array([0], np.uint32), np.array([0.0]))
    assert e.value.code == 1          # l must be > 0 (src/mmseq.cpp:604)
    with pytest.raises(MMGError) as e:
        gibbs.Problem.from_csr(np.array([1, 1], np.uint64), np.array([0], np.uint32), np.ones(1))
    assert e.value.code == 1


def test_host_instantiation_of_library_math_equals_oracle(orc):
    from mmseq_amd import gibbs
    rng = np.random.default_rng(11)
    x = np.concatenate([np.exp(rng.uniform(-700, 700, 100000)), rng.uniform(-745, 709, 100000),
                        [5e-324, 1e-310, 0.0, 1.0, np.inf]])
    r = gibbs.selftest_math(x, -1)
    assert np.array_equal(r["log"], orc.log_v(x), equal_nan=True)
    assert np.array_equal(r["exp"], orc.exp_v(x), equal_nan=True)
    got = gibbs.selftest_philox([1, 2, 3, 4], [5, 6], -1)
    assert np.array_equal(got[:4], orc.philox([1, 2, 3, 4], [5, 6])) and np.array_equal(got[4:], orc.philox2x32([1, 2], 5))
    for shape in (0.1, 1.0, 9.5):
        ref = np.empty(20000)
        orc.lib().orc_keyed_gamma_v(21, shape, 1.5, 20000, ref)
        assert np.array_equal(gibbs.selftest_gamma(21, shape, 1.5, 20000, -1), ref)
    for nn, p in ((1, 0.4), (25, 0.3), (5000, 0.6)):
        ref = np.empty(20000, np.uint32)
        orc.lib().orc_keyed_binomial_v(8, nn, p, 20000, ref)
        assert np.array_equal(gibbs.selftest_binomial(8, nn, p, 20000, -1), ref)
