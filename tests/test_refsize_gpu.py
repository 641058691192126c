"""`-refsize` beyond the sizes the fast paths are built for -- the reference has no limit
(wisecondor.py:379-381): newref above 256 (every row takes the exact scan) and test above 128
(numpy's pairwise tree then depends on the number of kept references).  Goldens: the real
reference with -refsize 300 on the 1 Mb prep seam (tests/golden/refsize300.npz)."""
import hashlib

import numpy as np
import pytest

from oracle import wc_oracle as wo

pytestmark = pytest.mark.gpu
KEYS = [str(c) for c in range(1, 23)] + ["X", "Y"]


def same_bits(a, b):
    a = np.asarray(a, dtype=np.float64)
    b = np.asarray(b, dtype=np.float64)
    if a.shape != b.shape:
        return False
    nan = np.isnan(a) & np.isnan(b)
    return bool(np.all(nan | (a.view(np.int64) == b.view(np.int64))))


@pytest.fixture(scope="module")
def wt():
    from wisecondor_amd import wisetools
    return wisetools


def test_newref_refsize_300(wt, golden):
    g1, g = golden("cfg1_pipeline.npz"), golden("refsize300.npz")
    corrected = np.asfortranarray(g1["prep_correctedData"])
    bins = g1["prep_maskedChromBins"]
    idx, dst = wt.getReference(corrected, bins, np.cumsum(bins), 300, 1, 1)
    assert np.array_equal(idx, g["ref_indexes"])
    assert hashlib.sha256(np.ascontiguousarray(dst).tobytes()).hexdigest() == str(g["ref_distances_sha256"])
    assert same_bits(dst[g["ref_distance_rows"]], g["ref_distances_sampled"])
    # parts, and the C-ordered (pairwise) variant against the oracle
    pi, pd = wt.getReference(corrected, bins, np.cumsum(bins), 300, 2, 3)
    lo, hi = wt.getPart(1, 3, corrected.shape[0])
    assert np.array_equal(pi, idx[lo:hi]) and same_bits(pd, dst[lo:hi])
    c_data = np.ascontiguousarray(corrected)
    ci, cd = wt.getReference(c_data, bins, np.cumsum(bins), 300, 1, 1)
    want_i, want_d = wo.get_reference(c_data, bins, np.cumsum(bins), 300, 1, 1, fast=True)
    assert np.array_equal(ci, want_i) and same_bits(cd, want_d)


@pytest.mark.parametrize("k", [129, 257, 300, 1024])
def test_newref_refsize_sweep_against_oracle(wt, k):
    from wisecondor_amd import synth
    data, bins, sums = synth.corrected_matrix(0, 20, seed=k, sizes=[50 + 7 * c for c in range(22)])
    data = np.asfortranarray(data)
    idx, dst = wt.getReference(data, bins, sums, k, 1, 1)
    want_i, want_d = wo.get_reference(data, bins, sums, k, 1, 1, fast=True)
    assert np.array_equal(idx, want_i) and same_bits(dst, want_d)


def test_test_path_refsize_300(wt, golden):
    g1, g = golden("cfg1_pipeline.npz"), golden("refsize300.npz")
    corrected = np.asfortranarray(g1["prep_correctedData"])
    bins = g1["prep_maskedChromBins"]
    idx, dst = wt.getReference(corrected, bins, np.cumsum(bins), 300, 1, 1)
    ref = wt.Reference(idx, dst, g1["ref_chromosome_sizes"], g1["ref_masked_sizes"], g1["ref_mask"],
                       g1["ref_pca_mean"], g1["ref_pca_components"], binsize=float(g1["ref_binsize"]))
    assert ref.cutoff == float(g["cutoff"])
    thr = float(g1["t_loss2_threshold_z"])
    names = ["gain5_gap", "loss2"]
    data = np.stack([g1["t_%s_xpca" % n] for n in names])
    for batch in (data, np.tile(data, (20, 1))):                     # latency-mode and batch-mode dispatch
        z, r, n, sd = wt.repeatTest(batch, None, None, None, None, None, thr, 5, reference=ref)
        for row in range(batch.shape[0]):
            name = names[row % 2]
            assert np.array_equal(n[row], g["t_%s_rep5_n" % name].astype(np.float64)), name
            assert same_bits(z[row], g["t_%s_rep5_z" % name]) and same_bits(r[row], g["t_%s_rep5_r" % name]), name
            assert sd[row] == float(g["t_%s_rep5_sd" % name]), name
    assert int(np.max(g["t_loss2_rep5_n"])) > 128                     # the lists really are longer than one block
    lengths = g1["sample_chrom_lengths"]
    offs = np.concatenate([[0], np.cumsum(lengths)])
    for name in names:
        sample = {k: g1["t_%s_sample" % name][offs[i]:offs[i + 1]] for i, k in enumerate(KEYS)}
        out = wt.test_batch(ref, [sample], thr)[0]
        want = g["t_%s_results_calls" % name]
        got = np.asarray(out["results_calls"], dtype=np.float64).reshape(-1, 5)
        assert np.array_equal(got[:, :3], want[:, :3]), name
        assert np.allclose(got[:, 3:], want[:, 3:], rtol=1e-9, atol=0), name
        assert np.allclose(np.concatenate(out["results_z"]), g["t_%s_results_z" % name], rtol=1e-9, atol=1e-11), name
        assert np.allclose(out["results_cwz"], g["t_%s_results_cwz" % name], rtol=1e-9, atol=1e-11), name
    ref.close()
