"""Differential fuzz on the device: random problems (shape, order, multiplicities, empty rows, dead / wildly scaled start
values) through randomly chosen kernel paths (sample kernels 2/1/0, EM kernels 2/1/0, 64-bit offsets, short tile ranges),
every count, trace entry, EM mu and log-likelihood compared bit for bit with the oracle.  tools/fuzz_parity.py runs more."""
import os
import sys

import pytest

pytestmark = pytest.mark.gpu
sys.path.insert(0, os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "tools"))


@pytest.mark.parametrize("block", range(4))
def test_fuzz_block(gpu, orc, block):
    import fuzz_parity
    for seed in range(1000 + 12 * block, 1000 + 12 * (block + 1)):
        fuzz_parity.one_case(seed, gpu, orc, verbose=False)
