"""ctypes binding of libwisecondor_hip.so (the C ABI in include/wisecondor_hip.h).

There is deliberately no CPU fallback: if the HIP library is missing or no GPU
is visible, every numeric entry point of this package raises.
"""
import ctypes
import os

import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))
LIB_PATH = os.path.join(HERE, "libwisecondor_hip.so")

_c = ctypes
_i64 = _c.c_int64
_i32 = _c.c_int
_dbl = _c.c_double
_vp = _c.c_void_p

#: every symbol include/wisecondor_hip.h declares: name -> (restype, argtypes)
SIGNATURES = {
    "wc_create": (_vp, [_i32]),
    "wc_destroy": (None, [_vp]),
    "wc_last_error": (_c.c_char_p, []),
    "wc_version": (_c.c_char_p, []),
    "wc_newref_stats": (_i32, [_vp, _vp]),
    "wc_get_part": (None, [_i64, _i64, _i64, _vp, _vp]),
    "wc_get_reference": (_i32, [_vp, _vp, _i64, _i64, _vp, _i32, _i32, _i32, _i64, _i64, _vp, _vp]),
    "wc_get_reference_dev": (_i32, [_vp, _vp, _vp, _i64, _i64, _vp, _i32, _i32, _i32, _i64, _i64, _vp, _vp]),
    "wc_newref_prepare_dev": (_i32, [_vp, _vp, _vp, _i64, _i64, _vp, _i32, _i32, _i32]),
    "wc_newref_thresholds_dev": (_i32, [_vp, _vp, _i64, _i64]),
    "wc_newref_get_thresholds_dev": (_i32, [_vp, _vp, _i64, _i64, _vp]),
    "wc_newref_set_thresholds_dev": (_i32, [_vp, _vp, _i64, _i64, _vp]),
    "wc_newref_get_bounds_dev": (_i32, [_vp, _vp, _i64, _i64, _vp, _vp]),
    "wc_newref_collect_dev": (_i32, [_vp, _vp, _i64, _i64, _i32, _i32]),
    "wc_newref_list_capacity": (_i64, [_vp]),
    "wc_newref_export_lists_dev": (_i32, [_vp, _vp, _i64, _i64, _i64, _vp, _vp]),
    "wc_newref_import_lists_dev": (_i32, [_vp, _vp, _i64, _i64, _i64, _vp, _vp]),
    "wc_newref_finish_dev": (_i32, [_vp, _vp, _i64, _i64, _vp, _vp]),
    "wc_newref_rescore_dev": (_i32, [_vp, _vp, _i64, _i64, _vp, _vp]),
    "wc_newref_fallback_dev": (_i32, [_vp, _vp, _i64, _i64, _vp, _vp]),
    "wc_newref_pick_dev": (_i32, [_vp, _vp, _i64, _i64, _vp, _vp]),
    "wc_newref_rescore_pairs_dev": (_i32, [_vp, _vp, _i64, _i64, _vp, _vp]),
    "wc_newref_exact_dev": (_i32, [_vp, _vp, _i64, _i64, _vp, _vp]),
    "wc_launch_floor_us": (_i32, [_vp, _vp, _i32, _i32, _vp]),
    "wc_read_samples": (_i32, [_vp, _i32, _i32, _vp, _i32, _dbl, _vp, _i64, _vp, _vp]),
    "wc_read_sample_lengths": (_i32, [_vp, _i32, _i32, _i32, _dbl, _vp, _vp, _vp]),
    "wc_write_test_results": (_i32, [_i32, _i32, _vp, _vp, _vp, _vp, _i64, _dbl, _i32, _dbl, _vp, _i32, _vp, _vp, _i64, _vp,
                                     _i32, _vp, _vp, _i32, _vp, _i32, _vp]),
    "wc_newref_prep_gram": (_i32, [_vp, _vp, _i64, _i64, _vp, _i32, _vp, _vp, _vp, _vp]),
    "wc_newref_prep_eig": (_i32, [_vp, _i32, _vp, _vp]),
    "wc_sym_eigh_leading_dev": (_i32, [_vp, _vp, _i64, _i32, _vp, _vp]),
    "wc_newref_prep_finish": (_i32, [_vp, _i32, _vp, _vp, _vp, _vp, _vp, _vp]),
    "wc_newref_prep_finish_dev": (_i32, [_vp, _i32, _vp, _vp, _vp, _vp, _vp, _vp]),
    "wc_newref_prep": (_i32, [_vp, _vp, _i64, _i64, _vp, _i32, _i32, _vp, _vp, _vp, _vp, _vp, _vp, _vp]),
    "wc_reference_create": (_vp, [_vp, _vp, _vp, _i64, _i32, _vp, _vp, _i32, _vp, _vp, _vp, _i32, _i32, _vp]),
    "wc_apply_pca": (_i32, [_vp, _vp, _i64, _i64, _vp, _vp, _i32, _vp]),
    "wc_reference_destroy": (None, [_vp]),
    "wc_reference_cutoff": (_dbl, [_vp]),
    "wc_optimal_cutoff": (_i32, [_vp, _vp, _i64, _i32, _vp]),
    "wc_optimal_cutoff_mask": (_i32, [_vp, _vp, _i64, _i32, _vp, _vp]),
    "wc_prepare_samples": (_i32, [_vp, _vp, _vp, _i64, _vp, _vp]),
    "wc_repeat_test": (_i32, [_vp, _vp, _vp, _i64, _dbl, _i32, _vp, _vp, _vp, _vp]),
    "wc_std_dev_avg": (_i32, [_vp, _vp, _i64, _i64, _vp, _vp]),
    "wc_stouffer_segments": (_i32, [_vp, _vp, _vp, _dbl, _vp, _i64, _dbl, _i32, _i32, _vp, _vp, _vp, _vp, _vp]),
    "wc_test_batch": (_i32, [_vp, _vp, _vp, _i64, _dbl, _i32, _i32, _dbl, _vp, _i32, _i32,
                             _vp, _vp, _vp, _vp, _vp, _vp]),
    "wc_test_profile": (_i32, [_vp, _i32]),
    "wc_debug_times": (_i32, [_vp, _i32, _vp]),
    "wc_test_profile_read": (_i32, [_vp, _vp]),
    "wc_test_batch_dev": (_i32, [_vp, _vp, _vp, _vp, _i64, _dbl, _i32, _i32, _dbl, _vp, _i32, _i32,
                                 _vp, _vp, _vp, _vp, _vp, _vp]),
}

SUM_PAIRWISE = 0
SUM_SEQUENTIAL = 1
E_ARG, E_HIP, E_LIMIT, E_INTERNAL = -1, -2, -3, -4     # WC_E_* of include/wisecondor_hip.h

_lib = None


class WisecondorHipError(RuntimeError):
    pass


def load():
    """dlopen the library and attach prototypes; raises if it has not been built."""
    global _lib
    if _lib is not None:
        return _lib
    if not os.path.exists(LIB_PATH):
        # a fresh checkout: compile the HIP sources (still the GPU product, not a fallback)
        try:
            from . import build
            build.build_library(verbose=False)
        except Exception as exc:
            raise WisecondorHipError(
                "%s is missing and could not be built (%s): run `python -m wisecondor_amd.build` "
                "(this package has no CPU fallback)" % (LIB_PATH, exc))
    _share_torch_hip_runtime()
    # WC_LIB_PATH: a differently built copy of the library (kernel A/B experiments); same C ABI
    lib = ctypes.CDLL(os.environ.get("WC_LIB_PATH") or LIB_PATH)
    _warn_if_two_hip_runtimes()
    for name, (res, args) in SIGNATURES.items():
        fn = getattr(lib, name)
        fn.restype = res
        fn.argtypes = args
    _lib = lib
    return lib


def _share_torch_hip_runtime():
    """One HIP runtime per process.  The PyTorch wheel bundles its own libamdhip64.so; if this
    library pulled the system ROCm's copy in first, a later `import torch` would initialise a second
    runtime and see no GPU.  So when torch is installed (not necessarily imported), its copy is
    loaded first, globally; libwisecondor_hip.so's libamdhip64.so.N then resolves to it."""
    import importlib.util
    try:
        spec = importlib.util.find_spec("torch")
    except (ImportError, ValueError):
        spec = None
    if spec is None or not spec.origin:
        return
    path = os.path.join(os.path.dirname(spec.origin), "lib", "libamdhip64.so")
    if os.path.exists(path):
        try:
            ctypes.CDLL(path, mode=ctypes.RTLD_GLOBAL)
        except OSError as exc:
            import warnings
            warnings.warn("wisecondor_amd: could not preload torch's HIP runtime %s (%s); importing torch "
                          "after this library may initialise a second runtime that sees no GPU" % (path, exc))


def _warn_if_two_hip_runtimes():
    """The preload above only helps if libwisecondor_hip.so's DT_NEEDED soname resolves to torch's
    copy.  With a different soname both runtimes get mapped and the second one sees no GPU: say so
    instead of failing later with an unrelated error."""
    try:
        with open("/proc/self/maps") as f:
            libs = {line.split()[-1] for line in f if "libamdhip64.so" in line}
    except OSError:
        return
    real = {os.path.realpath(p) for p in libs}
    if len(real) > 1:
        import warnings
        warnings.warn("wisecondor_amd: two HIP runtimes are mapped in this process (%s); the library was "
                      "built against a different libamdhip64 than the one torch bundles -- rebuild it with "
                      "the matching ROCm or expect 'no HIP device' errors" % ", ".join(sorted(real)))


def check(rc):
    if rc != 0:
        err = WisecondorHipError("wisecondor_hip error %d: %s" % (rc, load().wc_last_error().decode()))
        err.code = rc          # one of E_ARG / E_HIP / E_LIMIT / E_INTERNAL
        raise err


def ptr(arr):
    """Host pointer of a C-contiguous numpy array (or None)."""
    if arr is None:
        return None
    assert isinstance(arr, np.ndarray) and arr.flags["C_CONTIGUOUS"]
    return arr.ctypes.data_as(_vp)


_contexts = {}


def new_context(device=0):
    """A further wc_ctx on `device` with its own scratch, streams and error state (distributed.TestPipeline keeps
    several batches in flight with one each); the caller destroys it with destroy_context."""
    lib = load()
    h = lib.wc_create(device)
    if not h:
        raise WisecondorHipError("wc_create(%d) failed: %s" % (device, lib.wc_last_error().decode()))
    return h


def destroy_context(handle):
    if handle:
        load().wc_destroy(handle)


def context(device=0):
    """One wc_ctx per device, created lazily; fails loudly without a GPU."""
    if device not in _contexts:
        lib = load()
        h = lib.wc_create(device)
        if not h:
            raise WisecondorHipError("wc_create(%d) failed: %s" % (device, lib.wc_last_error().decode()))
        _contexts[device] = h
    return _contexts[device]
