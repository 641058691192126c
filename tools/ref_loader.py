"""Load the upstream WISECONDOR reference as a live Python-3 module (dev container only).

The reference under /root/reference is Python 2.  This helper copies its three
source files to a *temporary directory outside the repo*, runs the stdlib
``lib2to3`` fixer over the copies, installs the shims listed in SURVEY.md App. C
and imports the result.  Nothing the reference contains is written under
/root/repo: only arrays produced by calling it are ever committed (as golden
fixtures, see tools/make_goldens.py).

The GPU box has no /root/reference, so nothing in tests marked ``gpu``,
``bench.py`` or ``__graft_entry__.smoke()`` may import this module.
"""
import atexit
import importlib
import os
import shutil
import subprocess
import sys
import tempfile
import types

import numpy as np

REFERENCE_DIR = os.environ.get("WISECONDOR_REFERENCE", "/root/reference")
_cached = None


def available():
    return os.path.isfile(os.path.join(REFERENCE_DIR, "wisetools.py"))


def load(full_svd=True):
    """Return (wisetools_module, wisecondor_module, triarray_module)."""
    global _cached
    if _cached is not None:
        return _cached
    if not available():
        raise RuntimeError("reference sources not present at %s" % REFERENCE_DIR)
    tmp = tempfile.mkdtemp(prefix="wc_ref_")
    atexit.register(shutil.rmtree, tmp, ignore_errors=True)
    for name in ("triarray.py", "wisetools.py", "wisecondor.py"):
        dst = os.path.join(tmp, name)
        shutil.copyfile(os.path.join(REFERENCE_DIR, name), dst)
        os.chmod(dst, 0o644)
    subprocess.run(
        [sys.executable, "-W", "ignore", "-m", "lib2to3", "-w", "-n",
         "triarray.py", "wisetools.py", "wisecondor.py"],
        cwd=tmp, check=True, stdout=subprocess.DEVNULL, stderr=subprocess.DEVNULL)

    # shim (i): pysam is not installed and not needed for newref/test
    sys.modules.setdefault("pysam", types.ModuleType("pysam"))
    # shim (ii): fast_dot vanished from sklearn; it was np.dot
    import sklearn.utils.extmath as extmath
    if not hasattr(extmath, "fast_dot"):
        extmath.fast_dot = np.dot
    # shim (iii)+(vi): un-fitted PCA.transform, deterministic full SVD
    import sklearn.decomposition as skd
    _PCA = skd.PCA

    class RefPCA(_PCA):
        def __init__(self, n_components=None, copy=True, whiten=False, **kw):
            if full_svd:
                kw.setdefault("svd_solver", "full")
            super().__init__(n_components=n_components, copy=copy, whiten=whiten, **kw)

        def transform(self, X):
            if not hasattr(self, "explained_variance_"):
                return np.dot(np.asarray(X) - self.mean_, self.components_.T)
            return super().transform(X)

    skd.PCA = RefPCA
    # shim (iv): ragged lists -> explicit object arrays on save
    _savez = np.savez_compressed

    def savez_compat(file, *args, **kw):
        fixed = {}
        for k, v in kw.items():
            if isinstance(v, list) and v and isinstance(v[0], np.ndarray) \
                    and len({a.shape for a in v}) > 1:
                o = np.empty(len(v), dtype=object)
                for i, a in enumerate(v):
                    o[i] = a
                v = o
            fixed[k] = v
        return _savez(file, *args, **fixed)

    np.savez_compressed = savez_compat
    # shim (v): object members need allow_pickle
    _load = np.load

    def load_compat(file, *a, **kw):
        kw.setdefault("allow_pickle", True)
        return _load(file, *a, **kw)

    np.load = load_compat

    sys.path.insert(0, tmp)
    try:
        import matplotlib
        matplotlib.use("agg")
    except Exception:
        pass
    cwd = os.getcwd()
    os.chdir(tmp)  # `git describe` in getVersion() must not see our repo
    try:
        triarray = importlib.import_module("triarray")
        wisetools = importlib.import_module("wisetools")
        wisecondor = importlib.import_module("wisecondor")
    finally:
        os.chdir(cwd)
    _cached = (wisetools, wisecondor, triarray)
    return _cached
