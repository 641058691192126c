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


def test_prep_device_resident_equals_host_form(golden):
    """device_out=True hands newref the same values without the trip through the host."""
    import torch
    from wisecondor_amd import wisetools as wt
    g = golden("cfg1_pipeline.npz")
    offs = np.concatenate([[0], np.cumsum(g["sample_chrom_lengths"])])
    samples = [{k: row[offs[i]:offs[i + 1]] for i, k in enumerate(KEYS)} for row in g["ref_samples"]]
    host = wt.prepReference(samples)
    dev = wt.prepReference(samples, device_out=True)
    assert isinstance(dev[3], torch.Tensor) and dev[3].is_cuda and dev[3].is_contiguous()
    assert np.array_equal(dev[0].cpu().numpy(), host[0])
    assert np.array_equal(dev[3].cpu().numpy(), np.ascontiguousarray(host[3]))   # same bits, row-major
    assert np.array_equal(dev[4], host[4]) and np.array_equal(dev[5], host[5])
    assert list(dev[6]) == list(host[6]) and np.array_equal(dev[2], host[2])


@pytest.mark.parametrize("n_s,n_total", [(7, 300), (64, 4097), (100, 2500), (130, 999)])
def test_prep_gram_on_the_matrix_cores(n_s, n_total):
    """The split-K float64 MFMA SYRK against numpy on ragged shapes (tile and panel tails, odd bin counts)."""
    import ctypes
    from wisecondor_amd import _lib
    rng = np.random.RandomState(n_s)
    counts = rng.poisson(rng.gamma(5.0, 8.0, n_total)[None, :] * rng.uniform(0.5, 2.0, n_s)[:, None]).astype(np.int32)
    counts[:, rng.choice(n_total, n_total // 10, replace=False)] = 0          # masked-out bins
    sizes = np.ascontiguousarray([n_total - 21 * 3] + [3] * 21, dtype=np.int64)
    lib, ctx = _lib.load(), _lib.context(0)
    mask, mbins = np.empty(n_total, dtype=np.uint8), np.empty(22, dtype=np.int64)
    nb, gram = ctypes.c_int64(), np.empty((n_s, n_s))
    _lib.check(lib.wc_newref_prep_gram(ctx, _lib.ptr(counts), n_s, n_total, _lib.ptr(sizes), 22, _lib.ptr(mask),
                                       _lib.ptr(mbins), ctypes.byref(nb), _lib.ptr(gram)))
    norm = counts / counts.sum(axis=1, keepdims=True).astype(np.float64)
    keep = norm.sum(axis=0) > 0
    assert np.array_equal(mask.astype(bool), keep) and nb.value == int(keep.sum())
    x = norm[:, keep]
    xc = x - x.mean(axis=0)
    want = xc @ xc.T
    assert np.array_equal(gram, gram.T)
    assert np.allclose(gram, want, rtol=1e-11, atol=1e-12 * np.abs(want).max())


def test_prep_between_two_test_calls_leaves_the_test_path_intact(golden):
    """prep borrows workspaces of the context's test path (among them the buffer that caches the
    chromosome selection on the device): a `test` call after a prep must not see stale contents."""
    from wisecondor_amd import wisetools as wt
    g = golden("cfg1_pipeline.npz")
    offs = np.concatenate([[0], np.cumsum(g["sample_chrom_lengths"])])
    samples = [{k: row[offs[i]:offs[i + 1]] for i, k in enumerate(KEYS)} for row in g["ref_samples"]]
    reference = wt.Reference(g["ref_indexes"], g["ref_distances"], g["ref_chromosome_sizes"], g["ref_masked_sizes"],
                             g["ref_mask"], g["ref_pca_mean"], g["ref_pca_components"], binsize=float(g["ref_binsize"]))
    try:
        before = wt.test_batch(reference, samples[:3], 5.0)
        wt.prepReference(samples)
        after = wt.test_batch(reference, samples[:3], 5.0)
    finally:
        reference.close()
    for a, b in zip(before, after):
        assert np.array_equal(np.asarray(a["results_calls"]), np.asarray(b["results_calls"]))
        assert np.array_equal(np.concatenate(a["results_z"]), np.concatenate(b["results_z"]), equal_nan=True)
        assert np.array_equal(a["results_cwz"], b["results_cwz"], equal_nan=True)
