"""Randomised GPU-vs-oracle parity over shapes that do not line up with any tile size:
ragged chromosome layouts (including empty and 1-bin chromosomes), k above and below the
candidate count, duplicated rows, outlier rows, both numpy summation orders; and the
whole `test` path on references produced by the GPU newref."""
import os

import numpy as np
import pytest

from oracle import wc_oracle as wo

pytestmark = pytest.mark.gpu
SWEEP = int(os.environ.get("WC_SWEEP", "1"))      # WC_SWEEP=16: sixteen times the seeds (soak run)


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


def random_layout(rng, n_chrom, lo, hi):
    bins = rng.randint(lo, hi, size=n_chrom)
    if n_chrom > 3 and rng.rand() < 0.5:
        bins[rng.randint(0, n_chrom)] = 0          # a chromosome that lost every bin to the mask
    if n_chrom > 3 and rng.rand() < 0.5:
        bins[rng.randint(0, n_chrom)] = 1
    return bins.astype(np.int64)


@pytest.mark.parametrize("seed", range(16 * SWEEP))
def test_newref_random(wt, seed):
    rng = np.random.RandomState(1000 + seed)
    n_chrom = int(rng.choice([2, 3, 5, 22, 22, 24]))
    bins = random_layout(rng, n_chrom, 1, int(rng.choice([6, 30, 90, 140])))
    B = int(bins.sum())
    if B < 3:
        bins[0] += 3
        B = int(bins.sum())
    S = int(rng.choice([1, 2, 7, 8, 9, 31, 33, 64, 100, 129, 200, 513]))
    k = int(rng.choice([1, 5, 17, 100, 128]))
    data = 1.0 + 0.03 * rng.standard_normal((B, S))
    if rng.rand() < 0.5:      # exact duplicates across the genome
        for _ in range(max(1, B // 10)):
            data[rng.randint(0, B)] = data[rng.randint(0, B)]
    if rng.rand() < 0.3:
        data[rng.randint(0, B)] *= rng.choice([0.0, 30.0, 1e4])
    if rng.rand() < 0.2:
        data[rng.randint(0, B), rng.randint(0, S)] = rng.choice([np.nan, np.inf])
    if rng.rand() < 0.5:
        data = np.asfortranarray(data)
    sums = np.cumsum(bins)
    parts = int(rng.choice([1, 1, 2, 5]))
    for part in range(1, parts + 1):
        idx, dst = wt.getReference(data, bins, sums, k, part, parts)
        with np.errstate(all="ignore"):
            want_i, want_d = wo.get_reference(data, bins, sums, k, part, parts, fast=True)
        want_i = np.asarray(want_i).reshape(-1, k)
        want_d = np.asarray(want_d, dtype=np.float64).reshape(-1, k)
        assert np.array_equal(idx, want_i), (seed, B, S, k, part, parts)
        assert same_bits(dst, want_d), (seed, B, S, k, part, parts)


@pytest.mark.parametrize("seed", range(6 * SWEEP))
def test_whole_test_path_random(wt, seed):
    """Reference from the GPU newref on random data, samples with planted events, oracle toolTest."""
    rng = np.random.RandomState(77 + seed)
    sizes = rng.randint(25, 70, size=22).astype(np.int64)          # genomic bins per chromosome
    total = int(sizes.sum())
    mask = rng.rand(total) > 0.06
    offs = np.concatenate([[0], np.cumsum(sizes)])
    msizes = np.array([int(mask[offs[i]:offs[i + 1]].sum()) for i in range(22)], dtype=np.int64)
    B = int(msizes.sum())
    S = 20
    corrected = 1.0 + 0.02 * rng.standard_normal((B, S))
    k = int(rng.choice([30, 60]))
    idx, dst = wt.getReference(np.asfortranarray(corrected), msizes, np.cumsum(msizes), k, 1, 1)
    comps = np.linalg.qr(rng.standard_normal((B, 3)))[0].T
    mean = np.full(B, 1.0 / B) * (1 + 0.01 * rng.standard_normal(B))
    ref = dict(binsize=np.float64(1e6), indexes=idx, distances=dst, chromosome_sizes=sizes, mask=mask,
               masked_sizes=msizes, pca_mean=mean, pca_components=comps)
    reference = wt.Reference(idx, dst, sizes, msizes, mask, mean, comps, binsize=1e6)
    samples = []
    for _ in range(5):
        lam = np.full(total, 3000.0) * (1 + 0.02 * rng.standard_normal(total)).clip(0.5)
        c = rng.randint(0, 22)
        a = offs[c] + rng.randint(0, max(1, sizes[c] - 12))
        lam[a:a + rng.randint(4, 12)] *= rng.choice([0.6, 1.4, 1.08])
        counts = rng.poisson(lam).astype(np.int32)
        if rng.rand() < 0.5:
            counts[offs[3]:offs[3] + 2] = 0
        sample = {str(c + 1): counts[offs[c]:offs[c + 1]] for c in range(22)}
        # ragged inputs: one chromosome longer, one shorter than the reference layout
        sample["7"] = np.concatenate([sample["7"], np.array([5, 6, 7], dtype=np.int32)])
        sample["9"] = sample["9"][:-2]
        samples.append(sample)
    minref = int(rng.choice([5, 25]))
    thr = 4.2
    outs = wt.test_batch(reference, samples, thr, minrefbins=minref, repeats=4)
    for sample, out in zip(samples, outs):
        with np.errstate(all="ignore"):
            want = wo.test_sample(sample, 1e6, ref, minzscore=thr, minrefbins=minref, repeats=4)
        wc_ = np.asarray(want["results_calls"], dtype=np.float64).reshape(-1, 5)
        gc_ = out["results_calls"].reshape(-1, 5)
        assert np.array_equal(gc_[:, :3], wc_[:, :3]), (seed, gc_, wc_)
        assert np.allclose(gc_[:, 3:], wc_[:, 3:], rtol=1e-8, equal_nan=True)
        assert np.allclose(np.concatenate(out["results_z"]), np.concatenate(want["results_z"]),
                           rtol=1e-8, atol=1e-10, equal_nan=True)
        assert np.allclose(out["results_cwz"], want["results_cwz"], rtol=1e-8, atol=1e-9, equal_nan=True)
    reference.close()


@pytest.mark.parametrize("seed", range(3 * SWEEP))
def test_segments_mineffectsize_random(wt, seed):
    """-mineffectsize branch (fillTriMin, wisetools.py:479-487) on random regions: windows whose
    median ratio is too close to 1 count as zero; rounded ratios (median ties), NaN ratios."""
    rng = np.random.RandomState(12000 + seed)
    zs, rs = [], []
    for _ in range(8):
        n = int(rng.choice([1, 2, 4, 5, 31, 33, 64, 77, 130, 200]))
        z = rng.standard_normal(n)
        r = 1.0 + 0.05 * rng.standard_normal(n)
        if rng.rand() < 0.5:
            r = np.round(r, 2)
        for _ in range(int(rng.choice([0, 1, 2]))):
            if n > 10:
                a = rng.randint(0, n - 6)
                w = rng.randint(2, 6)
                sgn = rng.choice([-1, 1])
                z[a:a + w] += sgn * rng.choice([3.0, 5.0])
                r[a:a + w] += sgn * rng.choice([0.02, 0.1])
        if rng.rand() < 0.15 and n > 3:
            r[rng.randint(0, n)] = np.nan
        zs.append(z)
        rs.append(r)
    thr = float(rng.choice([2.0, 3.0, 4.5]))
    eff = float(rng.choice([0.01, 0.05, 0.15]))
    whole, segs = wt.stouffer_segments(zs, thr, 3, ratios=rs, mineffectsize=eff)
    for z, r, w, s in zip(zs, rs, whole, segs):
        with np.errstate(all="ignore"):
            tri = wo.fill_tri_min(z, r, eff)
        want = wo.segment_tri(tri, len(z), thr, 3)
        assert [(x, y) for _, (x, y) in s] == [(x, y) for _, (x, y) in want], (seed, len(z), thr, eff)
        assert same_bits([v for v, _ in s], [v for v, _ in want]), (seed, len(z))
        assert same_bits([w], [tri[len(z) - 1]]), (seed, len(z))


@pytest.mark.parametrize("seed", range(2 * SWEEP))
def test_mineffectsize_boundary_values(wt, seed):
    """The counting form of the median filter decides |median - 1| >= t from counts against two
    boundary doubles; ratios sitting exactly on, one ulp inside and one ulp outside those boundaries,
    heavy ties, infinities and NaN must come out as np.median + the reference's comparison say
    (even window lengths average the two middle values first)."""
    rng = np.random.RandomState(15000 + seed)
    eff = float(rng.choice([0.01, 0.05, 0.1, 0.5, 1.0, 2.5]))
    hi, lo = 1.0 + eff, 1.0 - eff
    pool = [hi, np.nextafter(hi, 0.0), np.nextafter(hi, 9.0), lo, np.nextafter(lo, 9.0), np.nextafter(lo, -9.0), 1.0,
            1.0 + eff / 2, 1.0 - eff / 2, 1.0 + 2 * eff, 1.0 - 2 * eff]
    zs, rs = [], []
    for _ in range(10):
        n = int(rng.choice([2, 3, 4, 7, 16, 40, 65, 90]))
        z = rng.standard_normal(n) + rng.choice([0.0, 1.5])
        r = rng.choice(pool, size=n)
        if rng.rand() < 0.3:
            r = np.where(rng.rand(n) < 0.5, r, 1.0 + 3 * eff * rng.standard_normal(n))
        if rng.rand() < 0.2:
            r[rng.randint(0, n)] = rng.choice([np.inf, -np.inf, np.nan])
        zs.append(z)
        rs.append(r)
    thr = float(rng.choice([1.5, 2.5]))
    whole, segs = wt.stouffer_segments(zs, thr, 3, ratios=rs, mineffectsize=eff)
    for z, r, w, s in zip(zs, rs, whole, segs):
        with np.errstate(all="ignore"):
            tri = wo.fill_tri_min(z, r, eff)
        want = wo.segment_tri(tri, len(z), thr, 3)
        assert [(x, y) for _, (x, y) in s] == [(x, y) for _, (x, y) in want], (seed, len(z), thr, eff, r)
        assert same_bits([v for v, _ in s], [v for v, _ in want]), (seed, len(z))
        assert same_bits([w], [tri[len(z) - 1]]), (seed, len(z))


def test_mineffectsize_counting_equals_sorted_insert_on_long_regions(wt, monkeypatch):
    """Regions far too long for the oracle's O(n^3) fillTriMin (3000 and 5000 bins, the 50 kb size): the
    O(1)-per-window counting kernel against the sorted-insert kernel it replaces (WC_MINEFFECT=sorted,
    itself pinned on the reference's goldens)."""
    rng = np.random.RandomState(77)
    zs, rs = [], []
    for n in (3000, 5000, 1237, 6500):        # (6 500: beyond the LDS-staged prefix slice of the masked value search)
        z = rng.standard_normal(n)
        r = np.round(1.0 + 0.03 * rng.standard_normal(n), 3)
        a = n // 3
        z[a:a + n // 10] += 0.8
        r[a:a + n // 10] += 0.04
        r[rng.randint(0, n, size=3)] = np.nan
        zs.append(z)
        rs.append(r)
    monkeypatch.delenv("WC_MINEFFECT", raising=False)
    whole_a, segs_a = wt.stouffer_segments(zs, 4.0, 3, ratios=rs, mineffectsize=0.02)
    monkeypatch.setenv("WC_MINEFFECT", "sorted")
    whole_b, segs_b = wt.stouffer_segments(zs, 4.0, 3, ratios=rs, mineffectsize=0.02)
    assert sum(len(s) for s in segs_a) >= 3
    for sa, sb in zip(segs_a, segs_b):
        assert [(x, y) for _, (x, y) in sa] == [(x, y) for _, (x, y) in sb]
        assert same_bits([v for v, _ in sa], [v for v, _ in sb])
    assert same_bits(whole_a, whole_b)


@pytest.mark.parametrize("longest", [1500, 6500])
def test_mineffectsize_on_many_regions(wt, monkeypatch, longest):
    """A round with many jobs (the four-wave forms of the masked value search, with and without the prefix slice in
    LDS): 180 regions of ~1 500 bins and one of 6 500 with the median filter on, counting kernel against the sorted-insert kernel, and the short regions among them against
    the oracle's fillTriMin."""
    rng = np.random.RandomState(78)
    zs, rs = [], []
    for i in range(180):
        n = longest if i == 1 else (1500 if i % 30 else 90)    # 6 500: one region beyond the LDS-staged prefix slice
        z = rng.standard_normal(n)
        r = np.round(1.0 + 0.03 * rng.standard_normal(n), 3)
        if i % 3 == 0:
            a = n // 4
            z[a:a + n // 8] += 1.0
            r[a:a + n // 8] += 0.05
        zs.append(z)
        rs.append(r)
    monkeypatch.delenv("WC_MINEFFECT", raising=False)
    whole_a, segs_a = wt.stouffer_segments(zs, 4.5, 3, ratios=rs, mineffectsize=0.02)
    monkeypatch.setenv("WC_MINEFFECT", "sorted")
    whole_b, segs_b = wt.stouffer_segments(zs, 4.5, 3, ratios=rs, mineffectsize=0.02)
    assert sum(len(s) for s in segs_a) >= 50
    for sa, sb in zip(segs_a, segs_b):
        assert [(x, y) for _, (x, y) in sa] == [(x, y) for _, (x, y) in sb]
        assert same_bits([v for v, _ in sa], [v for v, _ in sb])
    assert same_bits(whole_a, whole_b)
    for i in range(0, 180, 30):
        tri = wo.fill_tri_min(zs[i], rs[i], 0.02)
        want = wo.segment_tri(tri, len(zs[i]), 4.5, 3)
        assert [(x, y) for _, (x, y) in segs_a[i]] == [(x, y) for _, (x, y) in want], i
        assert same_bits([v for v, _ in segs_a[i]], [v for v, _ in want]), i


def test_degenerate_samples(wt):
    """Samples the reference still processes: a single read (hundreds of calls -- more than the
    library's default room per sample, the wrapper runs it again with more), a few spikes, half
    of the genome without reads, constant depth.  Call coordinates exact, values to 1e-8."""
    rng = np.random.RandomState(5)
    sizes = rng.randint(25, 60, size=22).astype(np.int64)
    total = int(sizes.sum())
    mask = rng.rand(total) > 0.05
    offs = np.concatenate([[0], np.cumsum(sizes)])
    msizes = np.array([int(mask[offs[i]:offs[i + 1]].sum()) for i in range(22)], dtype=np.int64)
    B = int(msizes.sum())
    corrected = 1.0 + 0.02 * rng.standard_normal((B, 20))
    idx, dst = wt.getReference(np.asfortranarray(corrected), msizes, np.cumsum(msizes), 40, 1, 1)
    comps = np.linalg.qr(rng.standard_normal((B, 3)))[0].T
    mean = np.full(B, 1.0 / B) * (1 + 0.01 * rng.standard_normal(B))
    ref = dict(binsize=np.float64(1e6), indexes=idx, distances=dst, chromosome_sizes=sizes, mask=mask,
               masked_sizes=msizes, pca_mean=mean, pca_components=comps)
    reference = wt.Reference(idx, dst, sizes, msizes, mask, mean, comps, binsize=1e6)

    def mk(counts):
        return {str(c + 1): np.asarray(counts[offs[c]:offs[c + 1]]).astype(np.int32) for c in range(22)}
    lam = np.full(total, 3000.0)
    samples = [mk(np.eye(1, total, 17).ravel()),
               mk(np.full(total, 2000000.0)),
               mk(rng.poisson(lam) * (1 + 50 * (rng.rand(total) < 0.01))),
               mk(np.where(np.arange(total) < total // 2, rng.poisson(lam), 0))]
    outs = wt.test_batch(reference, samples, 4.0, minrefbins=5, repeats=5)
    assert len(outs[0]["results_calls"]) > wt.MAX_CALLS
    for sample, out in zip(samples, outs):
        with np.errstate(all="ignore"):
            want = wo.test_sample(sample, 1e6, ref, minzscore=4.0, minrefbins=5, repeats=5)
        wc_ = np.asarray(want["results_calls"], dtype=np.float64).reshape(-1, 5)
        gc_ = out["results_calls"].reshape(-1, 5)
        assert np.array_equal(gc_[:, :3], wc_[:, :3])
        assert np.allclose(gc_[:, 3:], wc_[:, 3:], rtol=1e-8, equal_nan=True)
        assert np.allclose(np.concatenate(out["results_z"]), np.concatenate(want["results_z"]),
                           rtol=1e-8, atol=1e-10, equal_nan=True)
        assert np.allclose(out["results_cwz"], want["results_cwz"], rtol=1e-8, atol=1e-9, equal_nan=True)
    reference.close()


# (128 samples and more: the first repeat by k_zscore_tiled -- eight whole tiles; nine, the ninth dealt to all XCDs;
#  a list stride that is not a multiple of four; lists longer than the 103 references its registers hold)
_FLAG_SHAPES = [(3, 40), (40, 40), (40, 128), (70, 24), (40, 10), (5, 9), (128, 100), (144, 40), (128, 103), (160, 128)]


@pytest.mark.parametrize("n_samples,k,seed", [_FLAG_SHAPES[i % len(_FLAG_SHAPES)] + (i,)
                                              for i in range(len(_FLAG_SHAPES) * SWEEP)])
def test_repeats_heavy_flagging(wt, n_samples, k, seed):
    """A low threshold on noisy samples: every repeat adds flags, so later repeats recompute many
    (bin, sample) pairs with dropped references (cooperative pair kernel; full 128-entry lists
    at k = 128, fewer than eight kept values -- numpy's plain left-to-right sum -- at k = 9 / 10; negative and NaN values dropped from the first repeat on).  numpy's bits."""
    rng = np.random.RandomState(500 + seed)
    bins = np.array([120, 90, 1, 140, 75, 110], dtype=np.int64)
    sums = np.cumsum(bins)
    B = int(bins.sum())
    idx = np.empty((B, k), dtype=np.int32)
    b = 0
    for c in range(len(bins)):
        others = B - int(bins[c])
        for _ in range(int(bins[c])):
            idx[b] = rng.choice(others, size=k, replace=False)
            b += 1
    dst = np.sort(rng.rand(B, k), axis=1)
    dst[rng.randint(0, B, size=10)] = 2.0          # bins without any reference below the cutoff
    cutoff = 0.8
    data = 1.0 + 0.05 * rng.standard_normal((n_samples, B))
    hot = rng.rand(n_samples, B) < 0.08
    data[hot] *= 1.5
    data[rng.rand(n_samples, B) < 0.003] = -0.5
    data[rng.rand(n_samples, B) < 0.002] = np.nan
    z, r, n, sd = wt.repeatTest(data, idx, dst, bins, sums, cutoff, 1.5, 6)
    for s_ in range(n_samples):
        with np.errstate(all="ignore"):
            wz, wr, wn, wsd = wo.repeat_test(data[s_], idx, dst, bins, sums, cutoff, 1.5, 6)
        assert np.array_equal(n[s_], wn), (s_,)
        assert same_bits(z[s_], wz), (s_,)
        assert same_bits(r[s_], wr), (s_,)
        assert same_bits([sd[s_]], [wsd]), (s_,)


def test_calls_longer_than_the_staging_buffer(wt, monkeypatch):
    """A chromosome of 7 000 bins with gains over 5 600 and 6 900 of them: call rows whose ratios do not fit k_walk_rows'
    LDS staging (5 120 values: the radix selection then reads global memory) and ones that do, next to short ones --
    the walker's rows against the host-driven levels + k_call_post, bit for bit, and the effect sizes against
    np.median of the same bins."""
    rng = np.random.RandomState(4)
    sizes = np.array([7000] + [45] * 21, dtype=np.int64)
    total = int(sizes.sum())
    offs = np.concatenate([[0], np.cumsum(sizes)])
    mask = np.ones(total, dtype=bool)
    B, S = total, 24
    corrected = 1.0 + 0.02 * rng.standard_normal((B, S))
    idx, dst = wt.getReference(np.asfortranarray(corrected), sizes, np.cumsum(sizes), 60, 1, 1)
    comps = np.linalg.qr(rng.standard_normal((B, 3)))[0].T
    mean = np.full(B, 1.0 / B) * (1 + 0.01 * rng.standard_normal(B))
    reference = wt.Reference(idx, dst, sizes, sizes, mask, mean, comps, binsize=1e6)
    samples = []
    for span in ((100, 5700), (50, 6950), (3000, 3040), (10, 2100)):
        lam = np.full(total, 3000.0) * (1 + 0.02 * rng.standard_normal(total)).clip(0.5)
        lam[span[0]:span[1]] *= 1.05
        counts = rng.poisson(lam).astype(np.int32)
        samples.append({str(c + 1): counts[offs[c]:offs[c + 1]] for c in range(22)})
    samples = samples * 10                                  # 40 samples: the batch path
    thr = 4.5
    got = wt.test_batch(reference, samples, thr)
    monkeypatch.setenv("WC_TEST_WALK", "0")
    want = wt.test_batch(reference, samples, thr)
    longest = 0
    for a, b in zip(got, want):
        ca = np.asarray(a["results_calls"], dtype=np.float64).reshape(-1, 5)
        cb = np.asarray(b["results_calls"], dtype=np.float64).reshape(-1, 5)
        assert ca.shape == cb.shape and same_bits(ca, cb)
        r1 = np.asarray(a["results_r"][0])                  # chromosome 1: ratio - 1 per genomic bin (nothing masked)
        for row in ca[ca[:, 0] == 1]:
            x, y = int(row[1]), int(row[2])
            longest = max(longest, y - x)
            # (every bin is kept here, so positions are bin numbers and the call covers [start, end] -- the reference's end
            #  is the position of its last-but-one survivor + 1; bins dropped for their reference count would read 0)
            if y > x and not np.any(r1[x:y + 1] == 0.0):
                assert np.isclose(row[4], np.median(r1[x:y + 1] + 1.0) - 1.0, rtol=0, atol=1e-12)
    assert longest > 5200
    reference.close()


def test_tiled_and_untiled_first_repeat_agree(wt, monkeypatch):
    """k_zscore_tiled (sample tiles dealt to XCDs, four bins per wave) against k_zscore (WC_ZSCORE_TILED=0: a bin per
    wave) on 176 samples -- eleven tiles, three of them dealt to all XCDs -- with negative and NaN values: same bits."""
    rng = np.random.RandomState(9)
    bins = np.array([300, 250, 1, 410, 77], dtype=np.int64)
    B, k = int(bins.sum()), 100
    idx = np.empty((B, k), dtype=np.int32)
    b = 0
    for c in range(len(bins)):
        others = np.setdiff1d(np.arange(B), np.arange(int(bins[:c].sum()), int(bins[:c + 1].sum())))
        for _ in range(int(bins[c])):
            idx[b] = rng.choice(others, size=k, replace=False)
            b += 1
    dst = np.sort(rng.rand(B, k), axis=1)
    data = 1.0 + 0.05 * rng.standard_normal((176, B))
    data[rng.rand(176, B) < 0.002] = -0.5
    data[rng.rand(176, B) < 0.001] = np.nan
    got = wt.repeatTest(data, idx, dst, bins, np.cumsum(bins), 0.9, 2.0, 3)
    monkeypatch.setenv("WC_ZSCORE_TILED", "0")
    want = wt.repeatTest(data, idx, dst, bins, np.cumsum(bins), 0.9, 2.0, 3)
    for a, b_ in zip(got, want):
        assert same_bits(a, b_)


@pytest.mark.parametrize("seed", range(4 * SWEEP))
def test_segments_random(wt, seed):
    """Stouffer segmentation of random regions against the oracle's triangle walk: lengths around
    the block sizes of the search (64 rows, 32-entry end blocks), planted events of both signs,
    rounded values (exact ties between windows), NaN / inf entries, thresholds from 'everything
    is a call' to 'nothing is'; coordinates and window values bit for bit."""
    rng = np.random.RandomState(9000 + seed)
    regions = []
    for _ in range(12):
        n = int(rng.choice([1, 2, 3, 5, 31, 32, 33, 63, 64, 65, 127, 128, 129, 191, 200, 256, 257, 400]))
        z = rng.standard_normal(n) * rng.choice([0.3, 1.0, 1.0, 2.0])
        for _ in range(int(rng.choice([0, 1, 1, 2, 4]))):
            if n > 12:
                a = rng.randint(0, n - 10)
                z[a:a + rng.randint(2, 10)] += rng.choice([-1, 1]) * rng.choice([1.5, 3.0, 6.0])
        if rng.rand() < 0.2:
            z = np.round(z)
        if rng.rand() < 0.1:
            z[:] = rng.choice([0.0, 0.5, -2.0])
        if rng.rand() < 0.15:          # non-finite values: numpy's NaN-first argmax decides the calls
            z[rng.randint(0, n)] = rng.choice([np.nan, np.inf, -np.inf])
        regions.append(z)
    thr = float(rng.choice([0.5, 2.0, 3.5, 5.0, 8.0]))
    min_search = 3
    whole, segs = wt.stouffer_segments(regions, thr, min_search)
    for z, w, s in zip(regions, whole, segs):
        with np.errstate(all="ignore"):
            tri = wo.fill_tri(z)
            want = wo.segment_tri(tri, z.shape[0], thr, min_search)
        assert [(x, y) for _, (x, y) in s] == [(x, y) for _, (x, y) in want], (seed, z.shape, thr)
        assert same_bits([v for v, _ in s], [v for v, _ in want]), (seed, z.shape, thr)
        assert same_bits([w], [tri[z.shape[0] - 1]]), (seed, z.shape)


@pytest.mark.parametrize("order", ["C", "F"])
def test_newref_rare_paths(wt, order):
    """Clusters of exact duplicates large enough to hit the rarely taken paths: more than 64
    re-score candidates per row (several batches), more than RMAX tied candidates and
    candidate-list overflow (both -> exact fallback), k at its maximum."""
    rng = np.random.RandomState(99)
    bins = np.array([310, 150, 290, 40, 260, 333, 128, 257, 600, 90, 275, 267], dtype=np.int64)
    B = int(bins.sum())
    S = 24
    data = 1.0 + 0.02 * rng.standard_normal((B, S))
    perm = rng.permutation(B)
    at = 0
    for size in (70, 300, 700, 1500):               # rows of one cluster are bit-identical
        members = perm[at:at + size]
        data[members] = data[members[0]]
        at += size
    near = perm[at:at + 200]                        # near-duplicates: distances ~1e-14 apart
    data[near] = data[near[0]] + 1e-9 * rng.standard_normal((200, S))
    if order == "F":
        data = np.asfortranarray(data)
    sums = np.cumsum(bins)
    for k in (100, 256):
        idx, dst = wt.getReference(data, bins, sums, k, 1, 1)
        st = wt.newref_stats()
        with np.errstate(all="ignore"):
            want_i, want_d = wo.get_reference(data, bins, sums, k, 1, 1, fast=True)
        assert np.array_equal(idx, want_i), (order, k)
        assert same_bits(dst, want_d), (order, k)
        assert st["fallback_rows"] > 0 and st["fast_rows"] > 0, st     # both paths were exercised
