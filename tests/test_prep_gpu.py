"""newref prep on the GPU against the golden prep arrays of the
reference (sklearn PCA forced to its exact full-SVD solver)."""
import numpy as np
import pytest

pytestmark = pytest.mark.gpu
KEYS = [str(c) for c in range(1, 23)] + ["X", "Y"]


def test_prep_matches_reference(golden):
    from wisecondor_amd import wisetools as wt
    g = golden("cfg1_pipeline.npz")
    offs = np.concatenate([[0], np.cumsum(g["sample_chrom_lengths"])])
    samples = [{k: row[offs[i]:offs[i + 1]] for i, k in enumerate(KEYS)} for row in g["ref_samples"]]
    masked, bins, mask, corrected, comps, mean, mbins = wt.prepReference(samples)
    assert list(bins) == list(g["prep_chromosomeBins"])
    assert np.array_equal(mask, g["prep_mask"])
    assert list(mbins) == list(g["prep_maskedChromBins"])
    assert np.array_equal(masked, g["prep_maskedData"])                 # one IEEE division per element
    assert np.array_equal(mean, g["prep_pca_mean"])                     # numpy's sample-by-sample mean
    assert corrected.flags["F_CONTIGUOUS"] and corrected.shape == g["prep_correctedData"].shape
    # exact PCA through the Gram matrix vs LAPACK SVD: 1e-9 on unit-norm components, 1e-10 relative on the ratios
    assert np.allclose(comps, g["prep_pca_components"], rtol=0, atol=1e-9)
    assert np.allclose(corrected, g["prep_correctedData"], rtol=1e-10, atol=0)


def test_prep_one_call_c_api(golden):
    """The single-call C entry point (host Jacobi instead of LAPACK) on a few samples."""
    import ctypes
    from wisecondor_amd import wisetools as wt, _lib
    g = golden("cfg1_pipeline.npz")
    offs = np.concatenate([[0], np.cumsum(g["sample_chrom_lengths"])])
    samples = [{k: row[offs[i]:offs[i + 1]] for i, k in enumerate(KEYS)} for row in g["ref_samples"][:9]]
    want = wt.prepReference(samples)
    lib = _lib.load()
    ctx = _lib.context(0)
    sizes = np.ascontiguousarray(want[1], dtype=np.int64)
    counts = wt.samples_to_counts(samples, want[1])
    mask = np.empty(counts.shape[1], dtype=np.uint8)
    mbins = np.empty(22, dtype=np.int64)
    nb = ctypes.c_int64()
    _lib.check(lib.wc_newref_prep(ctx, _lib.ptr(counts), 9, counts.shape[1], _lib.ptr(sizes), 22, 3, _lib.ptr(mask),
                                  _lib.ptr(mbins), ctypes.byref(nb), None, None, None, None))
    B = nb.value
    assert B == want[0].shape[0]
    masked, ct, comps, mean = np.empty((B, 9)), np.empty((9, B)), np.empty((3, B)), np.empty(B)
    _lib.check(lib.wc_newref_prep(ctx, _lib.ptr(counts), 9, counts.shape[1], _lib.ptr(sizes), 22, 3, _lib.ptr(mask),
                                  _lib.ptr(mbins), ctypes.byref(nb), _lib.ptr(masked), _lib.ptr(ct), _lib.ptr(comps),
                                  _lib.ptr(mean)))
    assert np.array_equal(masked, want[0]) and np.array_equal(mean, want[5])
    assert np.allclose(comps, want[4], rtol=0, atol=1e-9)
    assert np.allclose(ct.T, want[3], rtol=1e-10, atol=0)
