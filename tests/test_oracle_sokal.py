"""The oracle's restatement of src/sokal.cc:33-87 against golden vectors produced by the REFERENCE's
own sokal() (compiled unmodified into oracle/_ref, tools/gen_golden.py) -- and, in the build
container where oracle/_ref exists, against the compiled reference live."""
import json
import os

import numpy as np

GOLD = os.path.join(os.path.dirname(__file__), "golden", "sokal_reference.json")


def _cases():
    return json.load(open(GOLD))["cases"]


def _close(a, b, rel):
    if np.isnan(b):
        return np.isnan(a)
    if b == 0:
        return abs(a) < 1e-300
    return abs(a - b) <= rel * abs(b)


def test_sokal_oracle_matches_reference_golden(orc):
    for c in _cases():
        x = np.array([float.fromhex(h) for h in c["x"]])
        rc, var, tau, m = orc.sokal(x)
        assert rc == c["rc"], c["name"]
        if rc != 0:
            continue
        assert m == c["m"], c["name"]
        assert _close(var, float.fromhex(c["var"]), 1e-12), c["name"]   # SURVEY App. E.1 tolerances
        assert _close(tau, float.fromhex(c["tau"]), 1e-9), c["name"]


def test_sokal_oracle_matches_compiled_reference_live(orc):
    rng = np.random.default_rng(5)
    if orc.sokal_ref(np.zeros(4)) is None:
        import pytest
        pytest.skip("oracle/_ref/libsokal_ref.so not built (reference tree absent)")
    for n in (4, 8, 64, 1024, 4096):
        for rho in (0.0, 0.5, 0.95):
            x = np.empty(n)
            x[0] = rng.normal()
            for i in range(1, n):
                x[i] = rho * x[i - 1] + rng.normal()
            a = orc.sokal(x)
            b = orc.sokal_ref(x)
            assert a[0] == b[0] == 0 and a[3] == b[3]
            assert _close(a[1], b[1], 1e-12) and _close(a[2], b[2], 1e-9)
    assert orc.sokal(np.ones(1000))[0] == orc.sokal_ref(np.ones(1000))[0] == 201
    assert orc.sokal(np.ones(2))[0] == orc.sokal_ref(np.ones(2))[0] == 200


def test_sokal_known_probe_value(orc):
    """SURVEY.md section 8c quotes the reference's behaviour on a constant trace: var 0, tau NaN, m = n+1."""
    rc, var, tau, m = orc.sokal(np.full(1024, 2.0))
    assert rc == 0 and var == 0.0 and np.isnan(tau) and m == 1025
