"""Host-side mirror of the reference's wisetools.py for the newref / test hot path.

Same function names, argument meaning and return shapes as the upstream
functions (cited per function as wisetools.py:line), but every numeric step
runs in the gfx950 HIP library through the C ABI (include/wisecondor_hip.h).
No numpy fallback exists: without the library or a GPU these functions raise.
"""
import ctypes

import numpy as np

from . import _lib

SENTINEL_INDEX = -1
SENTINEL_DISTANCE = 1e10


def getPart(partnum, outof, bincount):
    """Rows [start, end) of zero-based part `partnum` (wisetools.py:358-361)."""
    a, b = ctypes.c_int64(), ctypes.c_int64()
    _lib.load().wc_get_part(int(partnum), int(outof), int(bincount), ctypes.byref(a), ctypes.byref(b))
    return a.value, b.value


def sum_order_of(correctedData):
    """Rounding order numpy would use for np.sum(..., 1) on this array (wisetools.py:302).

    numpy reduces along the smaller-stride axis in its inner loop: a C-ordered
    [bins, samples] array gets a pairwise sum per row, the Fortran-ordered array
    np.load returns for a prep file (trainPCA's corrected.T, wisetools.py:101) a
    plain sample-by-sample sum.  The HIP library reproduces either.
    """
    a = np.asarray(correctedData)
    if a.ndim != 2 or a.shape[1] <= 1 or a.flags["C_CONTIGUOUS"]:
        return _lib.SUM_PAIRWISE
    return _lib.SUM_PAIRWISE if abs(a.strides[1]) <= abs(a.strides[0]) else _lib.SUM_SEQUENTIAL


def getReference(correctedData, chromosomeBins, chromosomeBinSums, selectRefAmount=100, part=1,
                 splitParts=1, device=0):
    """Reference bins for the rows of part `part` of `splitParts` (wisetools.py:364-398).

    Returns (int32 [rows, k] positions in the other-chromosomes concatenation,
    float64 [rows, k] ascending squared distances), -1 / 1e10 padded.
    """
    lib = _lib.load()
    ctx = _lib.context(device)
    order = sum_order_of(correctedData)
    data = np.ascontiguousarray(correctedData, dtype=np.float64)
    bins = np.ascontiguousarray(chromosomeBins, dtype=np.int64)
    n_bins = int(np.asarray(chromosomeBinSums)[-1])
    if data.ndim != 2 or data.shape[0] != n_bins:
        raise ValueError("correctedData must be [bins, samples] with %d bins" % n_bins)
    start, end = getPart(part - 1, splitParts, n_bins)
    print('Working on part', part, 'of', splitParts, 'meaning bins', start, 'up to', end)
    k = int(selectRefAmount)
    rows = max(end - start, 0)
    idx = np.empty((rows, k), dtype=np.int32)
    dst = np.empty((rows, k), dtype=np.float64)
    _lib.check(lib.wc_get_reference(ctx, _lib.ptr(data), n_bins, data.shape[1], _lib.ptr(bins),
                                    bins.shape[0], k, order, start, end, _lib.ptr(idx), _lib.ptr(dst)))
    return idx, dst


def newref_stats(device=0):
    """Counters of the last getReference call on this device (see wc_newref_stats)."""
    out = np.zeros(8, dtype=np.int64)
    _lib.check(_lib.load().wc_newref_stats(_lib.context(device), _lib.ptr(out)))
    return dict(fast_rows=int(out[0]), fallback_rows=int(out[1]), tiles=int(out[2]),
                sample_cols=int(out[3]), rescored=int(out[4]))
