"""The GPU eigen-solver of the prep step (csrc/eigh.hip; the reference hands trainPCA's problem to
scikit-learn, wisetools.py:89-101) against LAPACK: eigenvalues, eigenvectors up to sign (where they
are determined), residuals and orthogonality (always)."""
import os

import numpy as np
import pytest

pytestmark = pytest.mark.gpu


def _check(matrix, k, vec_tol=None, val_tol=1e-13):
    from wisecondor_amd import wisetools as wt
    vals, vecs = wt.sym_eigh_leading(matrix, k)
    w, v = np.linalg.eigh(matrix)
    w, v = w[::-1], v[:, ::-1].T
    scale = max(abs(w[0]), abs(w[-1]), 1e-300)
    assert np.abs(vals - w[:k]).max() <= val_tol * scale
    # residual and orthonormality hold whatever the multiplicities
    res = np.abs(matrix @ vecs.T - vecs.T * vals).max()
    assert res <= 1e-12 * scale * np.sqrt(matrix.shape[0])
    assert np.abs(vecs @ vecs.T - np.eye(k)).max() <= 1e-12
    if vec_tol is not None:
        for j in range(k):
            s = np.sign(np.dot(vecs[j], v[j]))
            assert np.abs(s * vecs[j] - v[j]).max() <= vec_tol
    return vals, vecs


@pytest.mark.parametrize("n", [3, 4, 5, 17, 64, 65, 100, 257, 600, 1025, 1500])
def test_random_symmetric(n):
    rng = np.random.default_rng(n)
    a = rng.standard_normal((n, n))
    _check(a + a.T, min(3, n), vec_tol=1e-10)


def test_largest_order():
    """4 096 is the limit of the prep step (and of this solver): the 96 KB LDS configuration of the
    column kernel, the 1 024-thread back-transformation."""
    rng = np.random.default_rng(4096)
    a = rng.standard_normal((4096, 4096))
    _check(a + a.T, 3, vec_tol=1e-9, val_tol=1e-12)


def test_random_sizes_and_pair_counts():
    """Sizes on both sides of every switch (one-workgroup tail at 128, register form up to 1 024,
    LDS or global work arrays) with 1..8 wanted pairs; WC_SWEEP scales the number of cases."""
    rng = np.random.default_rng(99)
    cases = 12 * max(1, int(os.environ.get("WC_SWEEP", "1")))
    for _ in range(cases):
        n = int(rng.choice([rng.integers(3, 12), rng.integers(120, 140), rng.integers(3, 700), rng.integers(1000, 1100)]))
        k = int(rng.integers(1, min(8, n) + 1))
        a = rng.standard_normal((n, n)) * rng.choice([1e-8, 1.0, 1e6])
        _check(a + a.T, k, vec_tol=1e-9 if n > 8 else None, val_tol=1e-12)


@pytest.mark.parametrize("n,bins", [(100, 4000), (300, 12000), (600, 30000)])
def test_gram_like_spectrum(n, bins):
    """One systematic component on top of Poisson-like noise: the wanted second and third
    eigenvalues sit inside the bulk, a few 1e-4 of the norm from their neighbours."""
    rng = np.random.default_rng(bins)
    profile = rng.uniform(0.5, 1.5, bins)
    x = rng.poisson(profile * 200.0 * rng.uniform(0.8, 1.2, (n, 1))).astype(np.float64)
    x /= x.sum(axis=1, keepdims=True)
    x -= x.mean(axis=0)
    g = x @ x.T
    _check(g, 3, vec_tol=1e-9)
    _check(g, 8, vec_tol=1e-8)


def test_structured_matrices():
    n = 200
    _check(np.eye(n), 4)                                         # one eigenvalue, any orthonormal vectors
    _check(np.diag(np.arange(1.0, n + 1)), 3, vec_tol=1e-12)     # already tridiagonal, zero reflectors
    z = np.zeros((n, n))
    z[5, 5] = 2.0
    z[7, 9] = z[9, 7] = 1.0
    _check(z, 3)                                                 # zero columns
    rng = np.random.default_rng(0)
    q, _ = np.linalg.qr(rng.standard_normal((n, n)))
    lam = np.concatenate([[10.0, 5.0, 5.0, 5.0], rng.uniform(0, 1, n - 4)])
    _check((q * lam) @ q.T, 4, val_tol=1e-12)                    # an exact triple eigenvalue among the wanted ones
    tiny = (q * lam) @ q.T * 1e-200
    _check(tiny, 3)
    huge = (q * lam) @ q.T * 1e+150
    _check(huge, 3)


def test_input_is_left_untouched_and_runs_repeat():
    import torch
    from wisecondor_amd import wisetools as wt
    rng = np.random.default_rng(3)
    a = rng.standard_normal((300, 300))
    a = a + a.T
    dev = torch.from_numpy(a).cuda()
    first = wt.sym_eigh_leading(dev, 3)
    again = wt.sym_eigh_leading(dev, 3)
    assert np.array_equal(dev.cpu().numpy(), a)
    assert np.array_equal(first[0], again[0]) and np.array_equal(first[1], again[1])     # deterministic


def test_limits_and_bad_input():
    from wisecondor_amd import _lib
    from wisecondor_amd import wisetools as wt
    with pytest.raises(_lib.WisecondorHipError):
        wt.sym_eigh_leading(np.eye(2), 1)
    with pytest.raises(_lib.WisecondorHipError):
        wt.sym_eigh_leading(np.eye(10), 9)
    bad = np.eye(50)
    bad[3, 4] = bad[4, 3] = np.nan
    with pytest.raises(_lib.WisecondorHipError):
        wt.sym_eigh_leading(bad, 2)


@pytest.mark.parametrize("mode", ["gpu", "host"])
def test_prep_with_either_solver_matches_the_reference(mode, monkeypatch, golden):
    """prepReference end to end with the eigenproblem on the GPU and on the host against the golden
    prep arrays of the reference (tests/test_prep_gpu.py holds the rest of that comparison)."""
    from wisecondor_amd import wisetools as wt
    monkeypatch.setenv("WC_PREP_EIG", mode)
    keys = [str(c) for c in range(1, 23)] + ["X", "Y"]
    g = golden("cfg1_pipeline.npz")
    offs = np.concatenate([[0], np.cumsum(g["sample_chrom_lengths"])])
    samples = [{k: row[offs[i]:offs[i + 1]] for i, k in enumerate(keys)} for row in g["ref_samples"]]
    _, _, mask, corrected, comps, mean, _ = wt.prepReference(samples)
    assert np.array_equal(mask, g["prep_mask"])
    assert np.allclose(comps, g["prep_pca_components"], rtol=0, atol=1e-9)
    assert np.allclose(corrected, g["prep_correctedData"], rtol=1e-10, atol=0)


def test_bad_solver_choice(monkeypatch):
    from wisecondor_amd import wisetools as wt
    monkeypatch.setenv("WC_PREP_EIG", "sometimes")
    with pytest.raises(ValueError):
        wt._eig_on_gpu(100, 3)
    monkeypatch.setenv("WC_PREP_EIG", "gpu")
    with pytest.raises(ValueError):
        wt._eig_on_gpu(2, 1)
