"""ctypes loader for libmmgibbs.so (the C ABI declared in include/mmgibbs.h).

The library is built in-tree (mmseq_amd/csrc/Makefile, via __graft_entry__.build()).
There is no fallback: if the shared object is missing, loading raises.
"""
import ctypes as C
import os

_HERE = os.path.dirname(os.path.abspath(__file__))
# MMSEQ_AMD_LIB: another build of the same library (A/B timing of kernel changes on one box, tools/k1_ab.py)
LIB_PATH = os.environ.get("MMSEQ_AMD_LIB") or os.path.join(_HERE, "csrc", "libmmgibbs.so")
_lib = None


def _point_rccl_at_pytorchs_copy():
    """One RCCL per process: the library loads RCCL on first use of a device group (dlopen); a Python process that ALSO imports torch
    (mmseq_amd/dist.py does) must not get /opt/rocm's build from here and PyTorch's bundled one from torch -- two builds in one process
    crash in the exit handlers.  So when a PyTorch installation carries its own librccl, MMG_RCCL_LIBRARY (read by csrc/group.hip)
    names that file; whichever of the two loads it first, the other gets the same handle.  torch itself is not imported here."""
    if os.environ.get("MMG_RCCL_LIBRARY"):
        return
    try:
        import importlib.util
        spec = importlib.util.find_spec("torch")
    except (ImportError, ValueError):
        return
    for base in (spec.submodule_search_locations or []) if spec else []:
        cand = os.path.join(base, "lib", "librccl.so")
        if os.path.exists(cand):
            os.environ["MMG_RCCL_LIBRARY"] = cand
            return


class MMGError(RuntimeError):
    def __init__(self, code, msg):
        super().__init__("libmmgibbs error %d: %s" % (code, msg))
        self.code = code


LAYOUT_CANONICAL, LAYOUT_KEEP_ROWS = 0, 1
# mmg_selftest_option ids
OPT_SAMPLE_KERNEL, OPT_FORCE_IDX64, OPT_SELL_WAVES_PER_CU, OPT_EM_KERNEL, OPT_EM_GRID, OPT_FUSE_CHAINS, OPT_CNT_REPLICAS, OPT_GROUP_FAIL, OPT_DERIVE_ORDER, OPT_WIRE_CHECK, OPT_BIGK_PER_WAVE, OPT_BIGK_SIDE_STREAM = range(12)


class ProblemDesc(C.Structure):
    _fields_ = [("m", C.c_uint64), ("n", C.c_uint32), ("row_ptr", C.c_void_p), ("col_idx", C.c_void_p),
                ("k", C.c_void_p), ("l", C.c_void_p), ("row_id_base", C.c_uint64), ("layout", C.c_uint32),
                ("tx_order", C.c_void_p)]


class SynthDesc(C.Structure):
    _fields_ = [("seed", C.c_uint64), ("rows", C.c_uint64), ("row0", C.c_uint64), ("n", C.c_uint32),
                ("avg_hits", C.c_double), ("uniform", C.c_int32), ("sorted", C.c_int32),
                ("mapped_reads", C.c_uint64), ("far_fraction", C.c_double), ("gene_size", C.c_uint32), ("far_family", C.c_uint32)]


class ProblemInfo(C.Structure):
    _fields_ = [("m", C.c_uint64), ("nnz", C.c_uint64), ("total_k", C.c_uint64), ("row_id_base", C.c_uint64),
                ("n", C.c_uint32), ("max_row_len", C.c_uint32), ("n_tiles", C.c_uint64),
                ("device_bytes", C.c_uint64), ("index_bits", C.c_int32), ("sample_kernel", C.c_int32),
                ("stream_bytes", C.c_uint64), ("fast_tiles", C.c_uint64), ("far_tiles", C.c_uint64), ("padded_slots", C.c_uint64),
                ("layout", C.c_int32), ("tx_renumbered", C.c_int32), ("sample_grid", C.c_int32), ("cu_count", C.c_int32)]


class SummaryDesc(C.Structure):
    _fields_ = [("chain", C.c_int32), ("n_virtual", C.c_uint32), ("virtual_id", C.c_void_p), ("virtual_scale", C.c_void_p),
                ("n_identical", C.c_uint32), ("identical_ptr", C.c_void_p), ("identical_member", C.c_void_p),
                ("n_genes", C.c_uint32), ("gene_ptr", C.c_void_p), ("gene_member", C.c_void_p),
                ("n_percentiles", C.c_uint32), ("percentile_index", C.c_void_p)]


class Config(C.Structure):
    _fields_ = [("alpha", C.c_double), ("beta", C.c_double), ("seed", C.c_uint64), ("n_chains", C.c_int32),
                ("chain_base", C.c_int32), ("gibbs_iter", C.c_int32), ("trace_len", C.c_int32),
                ("keep_trace", C.c_int32), ("timing", C.c_int32)]


class Timing(C.Structure):
    _fields_ = [("sample_ms", C.c_double), ("update_ms", C.c_double), ("sample_launches", C.c_uint64),
                ("update_launches", C.c_uint64)]


# every symbol include/mmgibbs.h declares: name -> (restype, argtypes)
SYMBOLS = {
    "mmg_last_error": (C.c_char_p, []),
    "mmg_abi_version": (C.c_int, []),
    "mmg_device_count": (C.c_int, [C.POINTER(C.c_int)]),
    "mmg_problem_create": (C.c_int, [C.POINTER(ProblemDesc), C.c_int, C.POINTER(C.c_void_p)]),
    "mmg_problem_create_synthetic": (C.c_int, [C.POINTER(SynthDesc), C.c_int, C.POINTER(C.c_void_p)]),
    "mmg_problem_info_get": (C.c_int, [C.c_void_p, C.POINTER(ProblemInfo)]),
    "mmg_problem_download": (C.c_int, [C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p]),
    "mmg_problem_tx_perm": (C.c_int, [C.c_void_p, C.c_void_p]),
    "mmg_selftest_option": (C.c_int, [C.c_int, C.c_int]),
    "mmg_selftest_kernel_info": (C.c_int, [C.c_int, C.POINTER(C.c_int), C.POINTER(C.c_int), C.POINTER(C.c_int), C.POINTER(C.c_int)]),
    "mmg_problem_get_l": (C.c_int, [C.c_void_p, C.c_void_p]),
    "mmg_problem_start_values": (C.c_int, [C.c_void_p, C.c_void_p, C.c_void_p]),
    "mmg_problem_em": (C.c_int, [C.c_void_p, C.c_void_p, C.c_int, C.c_double, C.POINTER(C.c_int),
                                 C.POINTER(C.c_double)]),
    "mmg_problem_destroy": (None, [C.c_void_p]),
    "mmg_em_create": (C.c_int, [C.c_void_p, C.c_void_p, C.POINTER(C.c_void_p), C.POINTER(C.c_double)]),
    "mmg_em_step": (C.c_int, [C.c_void_p, C.POINTER(C.c_double)]),
    "mmg_em_get_mu": (C.c_int, [C.c_void_p, C.c_void_p]),
    "mmg_em_stats": (C.c_int, [C.c_void_p, C.POINTER(C.c_int), C.POINTER(C.c_int), C.POINTER(C.c_int)]),
    "mmg_em_destroy": (None, [C.c_void_p]),
    "mmg_sampler_create": (C.c_int, [C.c_void_p, C.POINTER(Config), C.c_void_p, C.POINTER(C.c_void_p)]),
    "mmg_sampler_set_stream": (C.c_int, [C.c_void_p, C.c_void_p]),
    "mmg_sampler_run": (C.c_int, [C.c_void_p, C.c_int]),
    "mmg_sampler_sample": (C.c_int, [C.c_void_p]),
    "mmg_sampler_update": (C.c_int, [C.c_void_p]),
    "mmg_sampler_counts_devptr": (C.c_int, [C.c_void_p, C.POINTER(C.c_void_p), C.POINTER(C.c_uint64)]),
    "mmg_sampler_moments_devptr": (C.c_int, [C.c_void_p, C.POINTER(C.c_void_p), C.POINTER(C.c_uint64)]),
    "mmg_sampler_sync": (C.c_int, [C.c_void_p]),
    "mmg_sampler_wait_iterations": (C.c_int, [C.c_void_p, C.c_int]),
    "mmg_sampler_iteration": (C.c_int, [C.c_void_p, C.POINTER(C.c_int)]),
    "mmg_sampler_get_trace": (C.c_int, [C.c_void_p, C.c_int, C.c_void_p]),
    "mmg_sampler_get_trace_rows": (C.c_int, [C.c_void_p, C.c_int, C.c_int, C.c_int, C.c_void_p]),
    "mmg_sampler_get_trace_rows_done": (C.c_int, [C.c_void_p, C.c_int, C.c_int, C.c_int, C.c_void_p]),
    "mmg_sampler_get_mu": (C.c_int, [C.c_void_p, C.c_int, C.c_void_p]),
    "mmg_sampler_get_counts": (C.c_int, [C.c_void_p, C.c_int, C.c_void_p]),
    "mmg_sampler_get_moments": (C.c_int, [C.c_void_p, C.c_int, C.c_void_p, C.c_void_p, C.POINTER(C.c_int64)]),
    "mmg_sampler_get_timing": (C.c_int, [C.c_void_p, C.POINTER(Timing)]),
    "mmg_sampler_reset_timing": (C.c_int, [C.c_void_p]),
    "mmg_sampler_destroy": (None, [C.c_void_p]),
    "mmg_summary_create": (C.c_int, [C.c_void_p, C.POINTER(SummaryDesc), C.POINTER(C.c_void_p)]),
    "mmg_summary_begin": (C.c_int, [C.c_void_p, C.POINTER(SummaryDesc), C.POINTER(C.c_void_p)]),
    "mmg_summary_advance": (C.c_int, [C.c_void_p, C.c_int]),
    "mmg_summary_finish": (C.c_int, [C.c_void_p]),
    "mmg_summary_get": (C.c_int, [C.c_void_p, C.c_int, C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p]),
    "mmg_summary_get_proportions": (C.c_int, [C.c_void_p, C.c_int, C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p]),
    "mmg_summary_get_rows": (C.c_int, [C.c_void_p, C.c_int, C.c_int, C.c_int, C.c_void_p]),
    "mmg_summary_destroy": (None, [C.c_void_p]),
    "mmg_group_create": (C.c_int, [C.c_void_p, C.c_int, C.POINTER(C.c_void_p)]),
    "mmg_group_size": (C.c_int, [C.c_void_p, C.POINTER(C.c_int)]),
    "mmg_group_destroy": (None, [C.c_void_p]),
    "mmg_group_run_sharded": (C.c_int, [C.c_void_p, C.c_void_p, C.c_int]),
    "mmg_group_run_chains": (C.c_int, [C.c_void_p, C.c_void_p, C.c_int]),
    "mmg_group_enqueue_us": (C.c_int, [C.c_void_p, C.POINTER(C.c_double)]),
    "mmg_problem_shard": (C.c_int, [C.c_void_p, C.c_uint64, C.c_uint64, C.c_int, C.POINTER(C.c_void_p)]),
    "mmg_problem_shard_bounds": (C.c_int, [C.c_void_p, C.c_int, C.c_void_p]),
    "mmg_problem_shard_bounds_timed": (C.c_int, [C.c_void_p, C.c_void_p, C.c_int, C.c_void_p]),
    "mmg_group_em_create": (C.c_int, [C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p, C.POINTER(C.c_double)]),
    "mmg_selftest_em_shards": (C.c_int, [C.c_void_p, C.c_int, C.c_void_p, C.c_int, C.c_void_p, C.POINTER(C.c_double), C.POINTER(C.c_int)]),
    "mmg_selftest_gibbs_shards": (C.c_int, [C.c_void_p, C.c_int, C.c_int]),
    "mmg_group_pool_moments": (C.c_int, [C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p, C.POINTER(C.c_int64)]),
    "mmg_shard_bounds": (C.c_int, [C.c_void_p, C.c_uint64, C.c_int, C.c_void_p]),
    "mmg_host_gamma_trace": (C.c_int, [C.c_uint64, C.c_uint64, C.c_double, C.c_double, C.c_int, C.c_void_p]),
    "mmg_selftest_math": (C.c_int, [C.c_int, C.c_int64, C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p]),
    "mmg_selftest_philox": (C.c_int, [C.c_int, C.c_void_p, C.c_void_p, C.c_void_p]),
    "mmg_selftest_gamma": (C.c_int, [C.c_int, C.c_uint64, C.c_double, C.c_double, C.c_int64, C.c_void_p]),
    "mmg_selftest_binomial": (C.c_int, [C.c_int, C.c_uint64, C.c_uint32, C.c_double, C.c_int64, C.c_void_p]),
    "mmg_selftest_btrs_pretest": (C.c_int, [C.c_int, C.c_uint64, C.c_int64, C.c_double, C.c_double, C.c_void_p]),
    "mmg_selftest_binv_pretest": (C.c_int, [C.c_int, C.c_uint64, C.c_int64, C.c_double, C.c_double, C.c_double, C.c_void_p]),
}


def _share_hip_runtime_with_torch():
    """PyTorch wheels bundle their own libamdhip64.so (same SONAME as /opt/rocm's).  Two HIP runtimes in
    one process cannot both own the GPU, so when torch is installed its copy is loaded first and
    libmmgibbs.so binds to it by SONAME; a later `import torch` then reuses the same object."""
    try:
        import importlib.util
        spec = importlib.util.find_spec("torch")
        if spec is None or not spec.origin:
            return
        cand = os.path.join(os.path.dirname(spec.origin), "lib", "libamdhip64.so")
        if os.path.exists(cand):
            C.CDLL(cand, mode=C.RTLD_GLOBAL)
    except Exception:
        pass  # no torch (or an unusual layout): the system HIP runtime is used


def load():
    """Load libmmgibbs.so and bind every declared symbol; raises if anything is missing."""
    global _lib
    if _lib is not None:
        return _lib
    _share_hip_runtime_with_torch()
    _point_rccl_at_pytorchs_copy()
    if not os.path.exists(LIB_PATH):
        raise OSError("libmmgibbs.so not built at %s -- run `python -c 'import __graft_entry__ as g; g.build()'` "
                      "(there is no CPU fallback)" % LIB_PATH)
    lib = C.CDLL(LIB_PATH)
    for name, (res, args) in SYMBOLS.items():
        fn = getattr(lib, name)  # AttributeError if the symbol is not exported
        fn.restype = res
        fn.argtypes = args
    _lib = lib
    return lib


def check(rc):
    if rc != 0:
        raise MMGError(rc, load().mmg_last_error().decode("utf-8", "replace"))
