"""Host-side mirror of the reference's wisetools.py for the newref / test hot path.

Same function names, argument meaning and return shapes as the upstream
functions (cited per function as wisetools.py:line), but every numeric step
runs in the gfx950 HIP library through the C ABI (include/wisecondor_hip.h).
No numpy fallback exists: without the library or a GPU these functions raise.
"""
import ctypes
import os

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


# ---------------------------------------------------------------------------
# test path
# ---------------------------------------------------------------------------
MAX_CALLS = 256  # per sample (and per region) capacity handed to the C ABI


def scaleSample(sample, fromSize, toSize):
    """Merge bins to a coarser size (wisetools.py:220-237); host data marshalling."""
    if fromSize == toSize or toSize is None:
        return sample
    if toSize == 0 or fromSize == 0 or toSize < fromSize or toSize % fromSize > 0:
        print('ERROR: Impossible binsize scaling requested:', fromSize, 'to', toSize)
        raise SystemExit(1)
    scale = int(toSize / fromSize)
    out = dict()
    for chrom in sample:
        data = np.asarray(sample[chrom])
        new_len = int(np.ceil(len(data) / float(scale)))
        padded = np.zeros(new_len * scale, dtype=np.int64)
        padded[:len(data)] = data
        out[chrom] = padded.reshape(new_len, scale).sum(axis=1).astype(np.int32)
    return out


def samples_to_counts(samples, chromosome_sizes):
    """Dense int32 [n_samples, sum(chromosome_sizes)] image of sample dicts.

    The pad/truncate-to-reference-length part of toNumpyRefFormat
    (wisetools.py:268-274); the arithmetic happens on the GPU.
    """
    sizes = [int(v) for v in chromosome_sizes]
    out = np.zeros((len(samples), int(sum(sizes))), dtype=np.int32)
    for row, sample in enumerate(samples):
        at = 0
        for chrom, want in enumerate(sizes, start=1):
            data = np.asarray(sample[str(chrom)])
            have = min(want, len(data))
            out[row, at:at + have] = data[:have]
            at += want
    return out


class Reference(object):
    """Device-resident reference (`newref` output) shared by every sample of a batch.

    Holds what toolTest derives from the reference file alone
    (wisecondor.py:177-201): the arrays themselves, getOptimalCutoff and the
    per-bin reference lists.
    """

    def __init__(self, indexes, distances, chromosome_sizes, masked_sizes, mask, pca_mean,
                 pca_components, binsize=None, cutoff=None, device=0, ctx=None):
        lib = _lib.load()
        self.device = device
        self.ctx = ctx if ctx is not None else _lib.context(device)     # ctx: a context of _lib.new_context
        self.binsize = binsize
        self.indexes = np.ascontiguousarray(indexes, dtype=np.int32)
        self.distances = np.ascontiguousarray(distances, dtype=np.float64)
        self.chromosome_sizes = np.ascontiguousarray(chromosome_sizes, dtype=np.int64)
        self.masked_sizes = np.ascontiguousarray(masked_sizes, dtype=np.int64)
        self.mask = np.ascontiguousarray(np.asarray(mask).astype(np.uint8))
        self.pca_mean = np.ascontiguousarray(pca_mean, dtype=np.float64)
        comps = np.ascontiguousarray(pca_components, dtype=np.float64)
        self.pca_components = comps.reshape(-1, self.pca_mean.shape[0]) if comps.size else comps.reshape(0, self.pca_mean.shape[0])
        self.n_bins, self.k = self.indexes.shape
        self.n_total = int(self.chromosome_sizes.sum())
        override = None
        if cutoff is not None:
            override = ctypes.byref(ctypes.c_double(float(cutoff)))
        self.handle = lib.wc_reference_create(
            self.ctx, _lib.ptr(self.indexes), _lib.ptr(self.distances), self.n_bins, self.k,
            _lib.ptr(self.chromosome_sizes), _lib.ptr(self.masked_sizes), len(self.chromosome_sizes),
            _lib.ptr(self.mask), _lib.ptr(self.pca_mean), _lib.ptr(self.pca_components),
            self.pca_components.shape[0], 3, override)
        if not self.handle:
            raise _lib.WisecondorHipError("wc_reference_create: " + lib.wc_last_error().decode())
        self.cutoff = lib.wc_reference_cutoff(self.handle)

    @classmethod
    def from_npz(cls, npz, device=0):
        binsize = npz['binsize'].item() if hasattr(npz['binsize'], 'item') else npz['binsize']
        return cls(npz['indexes'], npz['distances'], npz['chromosome_sizes'], npz['masked_sizes'],
                   npz['mask'], npz['pca_mean'], npz['pca_components'], binsize=binsize, device=device)

    def clone(self, ctx):
        """The same reference in another context of the same device (its own device copy and user lists)."""
        return Reference(self.indexes, self.distances, self.chromosome_sizes, self.masked_sizes, self.mask,
                         self.pca_mean, self.pca_components, binsize=self.binsize, cutoff=self.cutoff,
                         device=self.device, ctx=ctx)

    def close(self):
        if getattr(self, "handle", None):
            _lib.load().wc_reference_destroy(self.handle)
            self.handle = None

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass


def toNumpyRefFormat(sample, chromBins, mask, device=0):
    """Pad/truncate, normalise to unit sum, apply the mask (wisetools.py:267-278)."""
    counts = samples_to_counts([sample], chromBins)
    mask = np.asarray(mask).astype(bool)
    n_bins = int(mask.sum())
    # a reference with only the layout (no neighbours, no PCA components) drives the kernels
    sizes = np.asarray(chromBins, dtype=np.int64)
    offs = np.concatenate([[0], np.cumsum(sizes)])
    msz = np.array([int(mask[offs[i]:offs[i + 1]].sum()) for i in range(len(sizes))], dtype=np.int64)
    ref = Reference(np.zeros((n_bins, 1), np.int32), np.full((n_bins, 1), 1e10), sizes, msz, mask,
                    np.ones(n_bins), np.zeros((0, n_bins)), cutoff=0.0, device=device)
    out = np.empty((1, n_bins))
    raw = np.empty((1, n_bins))
    _lib.check(_lib.load().wc_prepare_samples(ref.ctx, ref.handle, _lib.ptr(counts), 1, _lib.ptr(out), _lib.ptr(raw)))
    ref.close()
    return raw[0]


def applyPCA(sampleData, mean, components, device=0):
    """x / reconstruction from the stored components (wisetools.py:104-113)."""
    x = np.ascontiguousarray(np.atleast_2d(sampleData), dtype=np.float64)
    mean = np.ascontiguousarray(mean, dtype=np.float64)
    comps = np.ascontiguousarray(components, dtype=np.float64).reshape(-1, mean.shape[0])
    out = np.empty_like(x)
    _lib.check(_lib.load().wc_apply_pca(_lib.context(device), _lib.ptr(x), x.shape[0], x.shape[1],
                                        _lib.ptr(mean), _lib.ptr(comps), comps.shape[0], _lib.ptr(out)))
    return out[0] if np.ndim(sampleData) == 1 else out


def getOptimalCutoff(reference, repeats, device=0):
    """Iterated mean + 3 sd clip of the reference distances (wisetools.py:328-336)."""
    d = np.ascontiguousarray(reference, dtype=np.float64)
    if int(repeats) <= 0:       # the loop body never runs: +inf and the float zeros of wisetools.py:330
        return float("inf"), np.zeros(np.shape(reference))
    cutoff = ctypes.c_double()
    # the mask is the LAST iteration's, i.e. against the cutoff of the iteration before (wisetools.py:332)
    mask = np.empty(d.shape, dtype=np.uint8)
    _lib.check(_lib.load().wc_optimal_cutoff_mask(_lib.context(device), _lib.ptr(d), d.size, int(repeats),
                                                  ctypes.byref(cutoff), _lib.ptr(mask)))
    return cutoff.value, mask.view(np.bool_).reshape(np.shape(reference))


def repeatTest(testData, indexes, distances, chromosomeBins, chromosomeBinSums, cutoff, threshold,
               repeats, device=0, reference=None):
    """`repeats` z-score passes with flagging (wisetools.py:438-448).

    testData may be one vector [bins] or a batch [samples, bins]; returns
    (Z, R, refSizes, stdDevAvg) shaped like the input.
    """
    data = np.ascontiguousarray(np.atleast_2d(testData), dtype=np.float64)
    own = reference is None
    if own:
        sizes = np.asarray(chromosomeBins, dtype=np.int64)
        n_bins = int(sizes.sum())
        reference = Reference(indexes, distances, sizes, sizes, np.ones(n_bins, np.uint8), np.zeros(n_bins),
                              np.zeros((0, n_bins)), cutoff=cutoff, device=device)
    z = np.empty_like(data)
    r = np.empty_like(data)
    n = np.empty_like(data)
    sd = np.empty(data.shape[0])
    _lib.check(_lib.load().wc_repeat_test(reference.ctx, reference.handle, _lib.ptr(data), data.shape[0],
                                          float(threshold), int(repeats), _lib.ptr(z), _lib.ptr(r),
                                          _lib.ptr(n), _lib.ptr(sd)))
    if own:
        reference.close()
    if np.ndim(testData) == 1:
        return z[0], r[0], n[0], sd[0]
    return z, r, n, sd


def stdDevAvg(stdDevs, device=0, return_serial_count=False):
    """Mean of the non-NaN standard deviations, added bin by bin like trySample's Python loop
    (wisetools.py:428-435).  stdDevs: [bins] or [samples, bins]."""
    sd = np.ascontiguousarray(np.atleast_2d(stdDevs), dtype=np.float64)
    out = np.empty(sd.shape[0])
    serial = ctypes.c_int32(0)
    _lib.check(_lib.load().wc_std_dev_avg(_lib.context(device), _lib.ptr(sd), sd.shape[0], sd.shape[1],
                                          _lib.ptr(out), ctypes.byref(serial)))
    res = out[0] if np.ndim(stdDevs) == 1 else out
    return (res, serial.value) if return_serial_count else res


def stouffer_segments(regions, threshold, min_search=3, device=0, ratios=None, mineffectsize=0):
    """fillTri / fillTriMin + segmentTri for a list of 1-D z arrays
    (wisetools.py:466-487, triarray.py:59-84).

    With mineffectsize != 0, `ratios` (same shapes as `regions`) drive fillTriMin's
    median filter.  Returns (whole_region_z [n], [[(value, (x, y)), ...] per region]).
    """
    lib = _lib.load()
    regions = [np.ascontiguousarray(r, dtype=np.float64) for r in regions]
    offs = np.zeros(len(regions) + 1, dtype=np.int64)
    offs[1:] = np.cumsum([r.shape[0] for r in regions])
    z = np.ascontiguousarray(np.concatenate(regions) if regions else np.zeros(0))
    if z.size == 0:
        z = np.zeros(1)
    rat = None
    if mineffectsize != 0:
        if ratios is None or [len(r) for r in ratios] != [len(r) for r in regions]:
            raise ValueError("mineffectsize needs one ratio array per region")
        rat = np.ascontiguousarray(np.concatenate([np.asarray(r, dtype=np.float64) for r in ratios])
                                   if regions else np.zeros(1))
        if rat.size == 0:
            rat = np.zeros(1)
    nreg = len(regions)
    whole = np.empty(nreg)
    ncalls = np.zeros(nreg, dtype=np.int32)
    val = np.zeros((nreg, MAX_CALLS))
    cx = np.zeros((nreg, MAX_CALLS), dtype=np.int32)
    cy = np.zeros((nreg, MAX_CALLS), dtype=np.int32)
    _lib.check(lib.wc_stouffer_segments(_lib.context(device), _lib.ptr(z), _lib.ptr(rat), float(mineffectsize),
                                        _lib.ptr(offs), nreg, float(threshold), int(min_search), MAX_CALLS,
                                        _lib.ptr(whole), _lib.ptr(ncalls), _lib.ptr(val), _lib.ptr(cx),
                                        _lib.ptr(cy)))
    segs = [[(val[r, c], (int(cx[r, c]), int(cy[r, c]))) for c in range(ncalls[r])] for r in range(nreg)]
    return whole, segs


def fillTri(region, device=0):
    """Window triangle of a region (wisetools.py:466-472), never materialised on the GPU."""
    from .triarray import TriArr
    return TriArr.from_region(region, device=device)


def fillTriMin(regionZ, regionR, threshold, device=0):
    """fillTri, or its median-effect filtered variant when threshold != 0 (wisetools.py:475-487)."""
    from .triarray import TriArr
    if threshold == 0:
        return fillTri(regionZ, device=device)
    return TriArr.from_region(regionZ, device=device, ratio=regionR, mineffectsize=threshold)


def inflateArray(array, mask):
    """Scatter into the True positions of mask (wisetools.py:281-288); host shaping helper."""
    mask = np.asarray(mask)
    temp = np.zeros(mask.shape[0])
    temp[np.flatnonzero(mask)] = array
    return temp


def inflateArrayMulti(array, mask_list):
    """wisetools.py:291-295."""
    temp = array
    for mask in reversed(mask_list):
        temp = inflateArray(temp, mask)
    return temp


def test_batch(reference, samples, threshold, minrefbins=25, repeats=5, chromosomes=None, mineffectsize=0):
    """Numeric content of toolTest (wisecondor.py:199-268) for a list of sample dicts.

    Returns a list of dicts with results_z / results_r (per-chromosome lists),
    results_cwz, results_calls, asdef.  Samples must already be at the
    reference's bin size (see scaleSample).
    """
    lib = _lib.load()
    if chromosomes is None:
        chromosomes = list(range(1, 23))
    sel = np.ascontiguousarray(chromosomes, dtype=np.int32)
    out = []
    max_batch = max(1, 60000 // max(1, len(sel)))
    sizes = [int(v) for v in reference.chromosome_sizes]
    for lo in range(0, len(samples), max_batch):
        chunk = samples[lo:lo + max_batch]
        counts = samples_to_counts(chunk, sizes)
        ns = counts.shape[0]
        rz = np.empty((ns, reference.n_total))
        rr = np.empty((ns, reference.n_total))
        cwz = np.empty((ns, max(len(sel), 1)))
        ncalls = np.zeros(ns, dtype=np.int32)
        asdef = np.empty(ns)
        # The reference has no limit on the number of calls; the library's output arrays have one
        # (max_calls per sample and chromosome).  A sample that exceeds it -- e.g. one with almost
        # no reads -- is simply run again with more room.
        max_calls = MAX_CALLS
        while True:
            calls = np.zeros((ns, max_calls, 5))
            rc = lib.wc_test_batch(reference.ctx, reference.handle, _lib.ptr(counts), ns, float(threshold),
                                   int(minrefbins), int(repeats), float(mineffectsize), _lib.ptr(sel), len(sel),
                                   max_calls, _lib.ptr(rz), _lib.ptr(rr), _lib.ptr(cwz), _lib.ptr(calls),
                                   _lib.ptr(ncalls), _lib.ptr(asdef))
            if rc == _lib.E_LIMIT and b"max_calls" in lib.wc_last_error() and max_calls < reference.n_total:
                max_calls *= 4
                continue
            _lib.check(rc)
            break
        offs = np.concatenate([[0], np.cumsum(sizes)])
        for i in range(ns):
            out.append(dict(
                results_z=[rz[i, offs[c]:offs[c + 1]].copy() for c in range(len(sizes))],
                results_r=[rr[i, offs[c]:offs[c + 1]].copy() for c in range(len(sizes))],
                results_cwz=cwz[i, :len(sel)].copy(),
                results_calls=calls[i, :ncalls[i]].copy(),
                asdef=float(asdef[i])))
    return out


# ---------------------------------------------------------------------------
# newref prep (upstream of the hot path)
# ---------------------------------------------------------------------------
def _leading_eigenpairs(gram, n):
    """The n largest eigenvalues (descending) and unit eigenvectors (rows) of the symmetric `gram`
    on the host (LAPACK): the route of WC_PREP_EIG=host and of matrices eigh.hip does not take."""
    n_s = gram.shape[0]
    try:
        from scipy.linalg import eigh as _eigh          # LAPACK dsyevr on the wanted pairs only
        vals, vecs = _eigh(gram, subset_by_index=(max(0, n_s - n), n_s - 1), driver='evr')
    except Exception:                                    # no scipy: numpy's full decomposition
        vals, vecs = np.linalg.eigh(gram)
    order = np.argsort(vals)[::-1][:n]
    return np.ascontiguousarray(vals[order]), np.ascontiguousarray(vecs[:, order].T)


EIG_ON_GPU_FROM = 3         # samples: every size the solver takes stays on the GPU (up to 128 samples the whole
                            # tridiagonalisation is ONE workgroup with the matrix in LDS; LAPACK on the fetched Gram
                            # matrix is ~0.25 ms quicker at 100 samples: WC_PREP_EIG=host)


def _eig_on_gpu(n_s, pcacomp):
    """Where trainPCA's [samples, samples] eigenproblem is solved: WC_PREP_EIG=gpu|host, else by size."""
    mode = os.environ.get('WC_PREP_EIG', 'auto')
    if mode not in ('auto', 'gpu', 'host'):
        raise ValueError("WC_PREP_EIG must be gpu, host or auto, not %r" % mode)
    possible = 3 <= n_s <= 4096 and 1 <= pcacomp <= 8
    if mode == 'gpu' and not possible:
        raise ValueError("WC_PREP_EIG=gpu: the GPU solver takes 3..4096 samples and up to 8 components")
    return possible and (mode == 'gpu' or (mode == 'auto' and n_s >= EIG_ON_GPU_FROM))


def sym_eigh_leading(matrix, n_pairs, device=0):
    """The n_pairs largest eigenvalues (descending) and unit eigenvectors (rows) of a symmetric
    float64 matrix by the GPU solver of csrc/eigh.hip.  `matrix`: numpy array or CUDA tensor."""
    import torch
    lib = _lib.load()
    ctx = _lib.context(device)
    m = matrix if isinstance(matrix, torch.Tensor) else torch.from_numpy(np.ascontiguousarray(matrix, dtype=np.float64))
    m = m.to(device=torch.device('cuda', device), dtype=torch.float64).contiguous()
    n = m.shape[0]
    vals = np.empty(n_pairs)
    vecs = np.empty((n_pairs, n))
    torch.cuda.current_stream(device).synchronize()
    _lib.check(lib.wc_sym_eigh_leading_dev(ctx, ctypes.c_void_p(m.data_ptr()), n, int(n_pairs), _lib.ptr(vals),
                                           _lib.ptr(vecs)))
    return vals, vecs


def _pinned(shape):
    """float64 host array in page-locked memory (a device-to-host copy into it runs at the link's
    rate, several times the pageable rate); plain numpy when torch cannot pin."""
    try:
        import torch
        return torch.empty(shape, dtype=torch.float64, pin_memory=True).numpy()   # the array keeps the tensor alive
    except Exception:
        return np.empty(shape)


def prepReference(samples, pcacomp=3, device=0, device_out=False, counts=None, chrom_bins=None):
    """toNumpyArray + trainPCA (wisetools.py:240-264, 89-101) with the bins-sized work on the GPU.

    The Gram matrix of the centred [samples, bins] data comes from the GPU (float64 matrix
    cores); its small [samples, samples] eigenproblem is solved for the leading pairs by the direct
    solver of csrc/eigh.hip (from EIG_ON_GPU_FROM samples; below that, and under WC_PREP_EIG=host,
    by LAPACK on the fetched matrix), and the GPU finishes (components, projection,
    reconstruction, division).  Returns
    (maskedData [B,S], chromosomeBins, mask, correctedData [B,S] Fortran-ordered like the
    reference's, pca_components [n,B], pca_mean [B], maskedChromBins).

    device_out=True keeps the two bins x samples matrices in HBM: maskedData and correctedData
    come back as torch tensors on the device, correctedData as the C-ordered [B,S] tensor that
    getReference / NewrefJob take directly -- its values are those of the reference's
    Fortran-ordered array, so pass sum_order=_lib.SUM_SEQUENTIAL.

    counts / chrom_bins: the samples already as the dense int32 [samples, bins] matrix of
    samples_to_counts (a caller that ingests many files keeps them that way); `samples` is ignored.
    """
    lib = _lib.load()
    ctx = _lib.context(device)
    if counts is None:
        chromBins = [max(len(s[str(c)]) for s in samples) for c in range(1, 23)]
        counts = samples_to_counts(samples, chromBins)
    else:
        chromBins = [int(v) for v in chrom_bins]
        counts = np.ascontiguousarray(counts, dtype=np.int32)
    n_s, n_total = counts.shape
    sizes = np.ascontiguousarray(chromBins, dtype=np.int64)
    mask = np.empty(n_total, dtype=np.uint8)
    mbins = np.empty(len(sizes), dtype=np.int64)
    n_b = ctypes.c_int64()
    on_gpu = _eig_on_gpu(n_s, pcacomp)
    gram = None if on_gpu else np.empty((n_s, n_s))
    _lib.check(lib.wc_newref_prep_gram(ctx, _lib.ptr(counts), n_s, n_total, _lib.ptr(sizes), len(sizes),
                                       _lib.ptr(mask), _lib.ptr(mbins), ctypes.byref(n_b),
                                       None if on_gpu else _lib.ptr(gram)))
    if on_gpu:          # the Gram matrix never leaves HBM
        evals, evecs = np.empty(pcacomp), np.empty((pcacomp, n_s))
        _lib.check(lib.wc_newref_prep_eig(ctx, int(pcacomp), _lib.ptr(evals), _lib.ptr(evecs)))
    else:
        evals, evecs = _leading_eigenpairs(gram, pcacomp)
    B = n_b.value
    comps = np.empty((pcacomp, B))
    mean = np.empty(B)
    if device_out:
        import torch
        dev = torch.device('cuda', device)
        masked = torch.empty((B, n_s), dtype=torch.float64, device=dev)
        corrected = torch.empty((B, n_s), dtype=torch.float64, device=dev)
        _lib.check(lib.wc_newref_prep_finish_dev(ctx, int(pcacomp), _lib.ptr(evecs), _lib.ptr(evals),
                                                 ctypes.c_void_p(masked.data_ptr()),
                                                 ctypes.c_void_p(corrected.data_ptr()), _lib.ptr(comps), _lib.ptr(mean)))
        return masked, chromBins, mask.astype(bool), corrected, comps, mean, [int(v) for v in mbins]
    masked = _pinned((B, n_s))
    corrected_t = _pinned((n_s, B))
    _lib.check(lib.wc_newref_prep_finish(ctx, int(pcacomp), _lib.ptr(evecs), _lib.ptr(evals), _lib.ptr(masked),
                                         _lib.ptr(corrected_t), _lib.ptr(comps), _lib.ptr(mean)))
    return masked, chromBins, mask.astype(bool), corrected_t.T, comps, mean, [int(v) for v in mbins]
