"""Synthetic read-depth data with the shapes BASELINE.json names.

Recreates the generators described in SURVEY.md section 8(d): hg19 chromosome
lengths binned as the reference's `convert` does (wisetools.py:151-152), a
Gamma per-bin profile with an all-sample-zero stretch per chromosome, Poisson
sample counts, and the kernel-level `1 + 0.02 N(0,1)` corrected matrix.
"""
import numpy as np

HG19_LENGTHS = [
    249250621, 243199373, 198022430, 191154276, 180915260, 171115067,
    159138663, 146364022, 141213431, 135534747, 135006516, 133851895,
    115169878, 107349540, 102531392, 90354753, 81195210, 78077248,
    59128983, 63025520, 48129895, 51304566, 155270560, 59373566]
CHROM_KEYS = [str(c) for c in range(1, 23)] + ["X", "Y"]


def chrom_bins(binsize, autosomes_only=True):
    """Bins per chromosome: int(length / binsize + 1) (wisetools.py:151-152)."""
    n = 22 if autosomes_only else 24
    return [int(length / float(binsize) + 1) for length in HG19_LENGTHS[:n]]


def corrected_matrix(binsize, n_samples, seed=0, sizes=None):
    """Kernel-level newref input: (correctedData[B,S] f64, bins[22], cumulative[22])."""
    sizes = list(chrom_bins(binsize) if sizes is None else sizes)
    total = int(np.sum(sizes))
    data = 1.0 + 0.02 * np.random.RandomState(seed).standard_normal((total, n_samples))
    return data, np.array(sizes, dtype=np.int64), np.cumsum(sizes).astype(np.int64)


def bin_profile(binsize, sizes=None):
    """Per-bin expected depth for all 24 chromosomes, with a zeroed stretch each."""
    sizes = list(chrom_bins(binsize, autosomes_only=False) if sizes is None else sizes)
    rng = np.random.RandomState(1234)
    prof = []
    for n in sizes:
        p = rng.gamma(20.0, 1.0 / 20.0, size=n)
        hole = n // 25
        p[n // 3:n // 3 + hole] = 0.0
        prof.append(p)
    return prof


def make_sample(profile, seed, reads=1e7, events=()):
    """One converted sample: dict chrom -> int32[bins].

    `events` is a list of (chrom_key, first_bin, last_bin_exclusive, factor).
    """
    rng = np.random.RandomState(seed)
    g = 0.1 * rng.standard_normal()
    total = float(sum(p.sum() for p in profile))
    out = {}
    for key, p in zip(CHROM_KEYS, profile):
        lam = p / total * reads * (1.0 + g * (p - 1.0))
        lam = np.clip(lam, 0.0, None)
        for (ck, lo, hi, factor) in events:
            if ck == key:
                lam = lam.copy()
                lam[lo:hi] *= factor
        out[key] = rng.poisson(lam).astype(np.int32)
    return out
