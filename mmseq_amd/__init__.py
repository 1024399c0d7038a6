"""mmseq_amd -- MI355X-native Gibbs hot path of mmseq (eturro/mmseq), behind a C ABI.

csrc/      HIP kernels (gfx950), the C ABI (include/mmgibbs.h) and the C++ host code
gibbs.py   numpy-facing mirror of the C ABI (Problem, Sampler)
"""
from .gibbs import Problem, Sampler, device_count  # noqa: F401
