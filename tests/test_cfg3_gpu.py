"""BASELINE configs 2 and 3 at full size against the REAL reference (tests/golden/cfg3_250kb.npz,
written by tools/make_goldens.py --only cfg3 from the reference's own newrefprep / newrefpart /
newrefpost / test drivers on 100 samples x 250 kb bins):

* `newref` from the reference's prep seam: every index equal, the 1.1 M float64 distances
  bit-equal (SHA-256 of the bytes + every 89th row compared value by value);
* `test` of four samples (mild x1.05 gain, strong gain + loss whose flags change other bins'
  reference sets, small loss, normal): cutoff the reference's double; reference counts exact and
  z / ratio / stdDevAvg bit-equal from the golden PCA output; call coordinates exact; stored
  results within 1e-9 relative (the PCA projection's BLAS order is not reproducible) -- in
  latency mode (one sample per call, the < 32-sample kernels) and in a 32-sample batch (the
  wave-per-bin kernels).  The 250 kb chromosomes exceed 2048 bins' worth of end blocks only at
  chr1-2 (< 1000 bins each): this is the whole toolTest path at the size cfg3 names.
"""
import hashlib

import numpy as np
import pytest

pytestmark = pytest.mark.gpu
KEYS = [str(c) for c in range(1, 23)] + ["X", "Y"]
NAMES = ["mild18", "strong5", "loss2", "normal"]


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


@pytest.fixture(scope="module")
def g(golden):
    return golden("cfg3_250kb.npz")


@pytest.fixture(scope="module")
def built(wt, g):
    """indexes / distances computed here from the reference's correctedData (Fortran ordered)."""
    corrected = np.asfortranarray(g["prep_correctedData"])
    bins = g["prep_maskedChromBins"]
    idx, dst = wt.getReference(corrected, bins, np.cumsum(bins), 100, 1, 1)
    return idx, dst


def test_newref_equals_reference(wt, g, built):
    idx, dst = built
    assert idx.dtype == np.int32 and idx.shape == g["ref_indexes"].shape
    assert np.array_equal(idx, g["ref_indexes"])
    assert hashlib.sha256(np.ascontiguousarray(dst).tobytes()).hexdigest() == str(g["ref_distances_sha256"])
    assert same_bits(dst[g["ref_distance_rows"]], g["ref_distances_sampled"])
    stats = wt.newref_stats()
    assert stats["fast_rows"] + stats["fallback_rows"] == idx.shape[0]


def test_newref_parts_equal_whole(wt, g, built):
    """Row parts (the reference's `newrefpart m n`) of the same job: bit-identical rows."""
    corrected = np.asfortranarray(g["prep_correctedData"])
    bins = g["prep_maskedChromBins"]
    idx, dst = built
    at = 0
    for part in (1, 2, 3):
        pi, pd = wt.getReference(corrected, bins, np.cumsum(bins), 100, part, 3)
        assert np.array_equal(pi, idx[at:at + pi.shape[0]]) and same_bits(pd, dst[at:at + pi.shape[0]])
        at += pi.shape[0]
    assert at == idx.shape[0]


@pytest.fixture(scope="module")
def reference(wt, g, built):
    idx, dst = built
    ref = wt.Reference(idx, dst, g["ref_chromosome_sizes"], g["ref_masked_sizes"], g["ref_mask"],
                       g["ref_pca_mean"], g["ref_pca_components"], binsize=float(g["ref_binsize"]))
    yield ref
    ref.close()


def test_cutoff_is_the_references_double(g, reference):
    assert reference.cutoff == float(g["cutoff"])


@pytest.mark.parametrize("reps", [1, 5])
def test_repeat_test_bits_250kb(wt, g, reference, reps):
    data = np.stack([g["t_%s_xpca" % n] for n in NAMES])
    thr = float(g["t_mild18_threshold_z"])
    for batch in (data, np.tile(data, (8, 1))):            # 4 samples: pair kernels; 32: wave-per-bin kernels
        z, r, n, sd = wt.repeatTest(batch, None, None, None, None, None, thr, reps, reference=reference)
        for row in range(batch.shape[0]):
            name = NAMES[row % 4]
            assert np.array_equal(n[row], g["t_%s_rep%d_n" % (name, reps)].astype(np.float64)), (name, row)
            assert same_bits(z[row], g["t_%s_rep%d_z" % (name, reps)]), (name, row)
            if reps == 5:
                assert same_bits(r[row], g["t_%s_rep5_r" % name]), (name, row)
                assert sd[row] == float(g["t_%s_rep5_sd" % name]), (name, row)


def test_strong_sample_needs_the_repeats(g):
    """The golden would not notice a build that skips the repeat loop unless flags change
    reference sets: they do for the strong sample (SURVEY.md section 4 warning)."""
    assert not np.array_equal(g["t_strong5_rep1_n"], g["t_strong5_rep5_n"])
    assert not same_bits(g["t_strong5_rep1_z"], g["t_strong5_rep5_z"])


def _samples(g):
    lengths = g["sample_chrom_lengths"]
    offs = np.concatenate([[0], np.cumsum(lengths)])
    return [{k: g["t_%s_sample" % n][offs[i]:offs[i + 1]] for i, k in enumerate(KEYS)} for n in NAMES]


def _check_output(g, name, out):
    want_calls = g["t_%s_results_calls" % name]
    got_calls = np.asarray(out["results_calls"], dtype=np.float64).reshape(-1, 5)
    assert np.array_equal(got_calls[:, :3], want_calls[:, :3]), (name, got_calls, want_calls)
    assert np.allclose(got_calls[:, 3:], want_calls[:, 3:], rtol=1e-9, atol=0), name
    z = np.concatenate(out["results_z"])
    r = np.concatenate(out["results_r"])
    wz, wr = g["t_%s_results_z" % name], g["t_%s_results_r" % name]
    assert np.array_equal(z == 0, wz == 0) and np.array_equal(r == 0, wr == 0), name       # same removed bins
    assert np.allclose(z, wz, rtol=1e-9, atol=1e-11), name
    assert np.allclose(r, wr, rtol=1e-9, atol=1e-13), name
    assert np.allclose(out["results_cwz"], g["t_%s_results_cwz" % name], rtol=1e-9, atol=1e-11), name
    assert np.isclose(out["asdef"], float(g["t_%s_asdef" % name]), rtol=1e-11), name


def test_whole_test_single_sample_250kb(wt, g, reference):
    """BASELINE config 3: one sample per call (latency mode)."""
    thr = float(g["t_mild18_threshold_z"])
    from wisecondor_amd.wisecondor import zThreshold
    assert np.isclose(zThreshold([int(v) for v in g["ref_masked_sizes"]], 1000, None), thr, rtol=1e-14)
    for name, sample in zip(NAMES, _samples(g)):
        out = wt.test_batch(reference, [sample], thr)[0]
        _check_output(g, name, out)
    # the survey's landmark: the x1.05 gain on chr18 bins 100-219 is called as [18, 98, 217]
    assert [18.0, 98.0, 217.0] in g["t_mild18_results_calls"][:, :3].tolist()


def test_whole_test_batch_250kb(wt, g, reference):
    """The same four samples eight times over in one 32-sample batch (batch kernels)."""
    thr = float(g["t_mild18_threshold_z"])
    outs = wt.test_batch(reference, _samples(g) * 8, thr)
    for i, out in enumerate(outs):
        _check_output(g, NAMES[i % 4], out)
