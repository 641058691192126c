"""BASELINE.json's full sizes, checked through size-independent properties (the oracle needs
hours there): newref 600 samples x 50 kb, batched test at 50 kb."""
import numpy as np
import pytest

from oracle import wc_oracle as wo

pytestmark = pytest.mark.gpu


@pytest.fixture(scope="module")
def wt():
    from wisecondor_amd import wisetools
    return wisetools


def test_newref_600x50kb_properties(wt):
    """Every row: k distinct in-range candidates from other chromosomes, ascending (distance, index)
    order; spot checks: stored distances are numpy's bits, and no unlisted candidate beats the k-th."""
    from wisecondor_amd import synth
    data, bins, sums = synth.corrected_matrix(50000, 600, seed=0)
    data = np.asfortranarray(data)                        # the prep-file layout: sequential summation
    B, k = data.shape[0], 100
    idx, dst = wt.getReference(data, bins, sums, k, 1, 1)
    assert idx.shape == (B, k) and dst.shape == (B, k)
    st = wt.newref_stats(0)
    assert st["fast_rows"] + st["fallback_rows"] == B
    chrom = np.repeat(np.arange(len(bins)), bins)
    n_other = B - np.asarray(bins)[chrom]                 # candidates per row
    assert (idx >= 0).all() and (idx < n_other[:, None]).all()
    assert (np.diff(dst, axis=1) >= 0).all()
    ties = np.diff(dst, axis=1) == 0
    assert (np.diff(idx, axis=1)[ties] > 0).all()         # ties keep the lower index first
    srt = np.sort(idx, axis=1)
    assert (np.diff(srt, axis=1) > 0).all()               # no candidate twice
    rng = np.random.RandomState(0)
    starts = np.concatenate([[0], np.cumsum(bins)])
    for row in rng.choice(B, 12, replace=False):
        c = chrom[row]
        others = np.concatenate([data[:starts[c]], data[starts[c + 1]:]])
        with np.errstate(all="ignore"):
            d = np.sum(np.power(others - data[row], 2), 1)      # wisetools.py:302 on the F-ordered rows
        order = np.argsort(d, kind="stable")[:k]
        assert np.array_equal(idx[row], order.astype(np.int32)), row
        assert np.array_equal(dst[row].view(np.uint64), d[order].view(np.uint64)), row


@pytest.mark.parametrize("kind", ["uncorrelated", "pipeline"])
def test_fast_path_equals_exact_path_on_every_row_600x50kb(wt, kind):
    """BASELINE config 4 (600 samples x 50 kb, 57 633 / ~55 k masked bins): the float16-bound fast path against
    the exact GPU path (float64 distances to every candidate in numpy's order, stable selection; pinned on the
    reference at the sizes the reference can run) for EVERY row -- indexes and distance bits -- on the
    kernel-level matrix (uncorrelated rows, C order: pairwise sums) and on a matrix made by the prep pipeline
    from 600 synthetic samples (correlated rows, Fortran order: sequential sums).  wisetools.py:298-325."""
    import torch
    from wisecondor_amd import _lib, distributed, synth
    if kind == "uncorrelated":
        data, bins, _ = synth.corrected_matrix(50000, 600, seed=0)
        order = _lib.SUM_PAIRWISE
    else:
        profile = synth.bin_profile(50000)
        samples = [synth.make_sample(profile, seed=i) for i in range(600)]
        _, _, _, data, _, _, bins = wt.prepReference(samples)
        order = wt.sum_order_of(data)
        assert order == _lib.SUM_SEQUENTIAL
        del samples
    bins = np.asarray(bins, dtype=np.int64)
    B = data.shape[0]
    assert B > 50000 and data.shape[1] == 600
    X = torch.from_numpy(np.ascontiguousarray(data)).cuda()
    job = distributed.NewrefJob(_lib.context(0), X, bins, 100, order)
    idx, dst = job.run()
    torch.cuda.synchronize()
    stats = wt.newref_stats(0)
    assert stats["fast_rows"] == B and stats["fallback_rows"] == 0, stats      # what is compared IS the fast path
    idx, dst = idx.clone(), dst.clone()
    ex_i, ex_d = torch.empty_like(idx), torch.empty_like(dst)
    job.st.exact(0, B, ex_i, ex_d)
    torch.cuda.synchronize()
    bad = (idx != ex_i).any(dim=1) | (dst.view(torch.int64) != ex_d.view(torch.int64)).any(dim=1)
    assert int(bad.sum()) == 0, torch.nonzero(bad)[:10].flatten().tolist()
    # and the exact path itself against numpy on a few rows (it is pinned on the reference at small sizes)
    rng = np.random.RandomState(1)
    chrom = np.repeat(np.arange(len(bins)), bins)
    starts = np.concatenate([[0], np.cumsum(bins)])
    host_i, host_d = ex_i.cpu().numpy(), ex_d.cpu().numpy()
    lay = np.asfortranarray(data) if order == _lib.SUM_SEQUENTIAL else np.ascontiguousarray(data)
    for row in rng.choice(B, 4, replace=False):
        c = chrom[row]
        others = np.concatenate([lay[:starts[c]], lay[starts[c + 1]:]])
        with np.errstate(all="ignore"):
            d = np.sum(np.power(others - lay[row], 2), 1)
        o = np.argsort(d, kind="stable")[:100]
        assert np.array_equal(host_i[row], o.astype(np.int32)), row
        assert np.array_equal(host_d[row].view(np.uint64), d[o].view(np.uint64)), row


@pytest.mark.parametrize("tag", ["c", "f"])
def test_fast_path_equals_the_reference_on_the_cfg4_slice(wt, golden, tag):
    """BASELINE config 4 pinned on the reference ITSELF: cfg4slice.npz holds the real getReference's output
    (tools/make_goldens.py --only cfg4slice; wisetools.py:298-325, 364-398) for 64 target rows of the
    600 x 50 kb kernel-level matrix against all candidates -- first / last bins of chromosomes, chr1, chr21 / 22 --
    in C order (numpy's pairwise row sums) and Fortran order (sequential).  The float16-bound FAST path (the
    statistics prove no row left it) must deliver those indexes and distance bits."""
    import torch
    from wisecondor_amd import _lib, distributed, synth
    g = golden("cfg4slice.npz")
    data, bins, _ = synth.corrected_matrix(50000, 600, seed=0)
    rows = g["rows"]
    assert tuple(g["shape"]) == data.shape and np.array_equal(data[rows[:4], :3], g["data_probe"])
    order = _lib.SUM_PAIRWISE if tag == "c" else _lib.SUM_SEQUENTIAL
    B = data.shape[0]
    X = torch.from_numpy(data).cuda()
    del data
    job = distributed.NewrefJob(_lib.context(0), X, np.asarray(bins, dtype=np.int64), 100, order)
    idx, dst = job.run()
    torch.cuda.synchronize()
    stats = wt.newref_stats(0)
    assert stats["fast_rows"] == B and stats["fallback_rows"] == 0, stats
    sel = torch.from_numpy(rows).cuda()
    got_i, got_d = idx[sel].cpu().numpy(), dst[sel].cpu().numpy()
    assert np.array_equal(got_i, g["idx_" + tag])
    assert np.array_equal(got_d.view(np.uint64), g["dst_" + tag].view(np.uint64))
    # the function seam on the same rows (getReference picks the order from the array's strides): part = row + 1 of B
    lay = np.ascontiguousarray(X.cpu().numpy()) if tag == "c" else np.asfortranarray(X.cpu().numpy())
    for n in (0, 33, 63):
        i1, d1 = wt.getReference(lay, bins, np.cumsum(bins), 100, int(rows[n]) + 1, B)
        assert np.array_equal(i1[0], g["idx_" + tag][n]) and np.array_equal(d1[0].view(np.uint64), g["dst_" + tag][n].view(np.uint64))


def test_batched_test_50kb_equals_single_samples(wt):
    """cfg5's shape per GPU (here 24 samples x 50 kb): every output of the batch equals the output
    of the same sample tested alone, and a second run of the batch is bit-identical."""
    from wisecondor_amd import synth
    from wisecondor_amd.wisecondor import zThreshold
    binsize = 50000
    profile = synth.bin_profile(binsize)
    refs = [synth.make_sample(profile, seed=i) for i in range(40)]
    _, chrom_bins, mask, corrected, comps, mean, masked_bins = wt.prepReference(refs)
    masked_bins = np.asarray(masked_bins, dtype=np.int64)
    idx, dst = wt.getReference(corrected, masked_bins, np.cumsum(masked_bins), 100, 1, 1)
    reference = wt.Reference(idx, dst, np.asarray(chrom_bins, dtype=np.int64), masked_bins, mask, mean,
                             comps, binsize=binsize, device=0)
    thr = float(zThreshold([int(v) for v in masked_bins], 1000, None))
    tests = []
    for i in range(24):
        events = [("7", 400, 900, 1.04)] if i % 3 == 0 else []
        if i == 1:
            events = [("2", 1000, 2500, 1.03)]          # a call longer than 1024 bins: radix-select median
        tests.append(synth.make_sample(profile, seed=500 + i, events=events))
    a = wt.test_batch(reference, tests, thr)
    b = wt.test_batch(reference, tests, thr)

    def same(u, v):
        if not np.array_equal(np.asarray(u["results_cwz"]).view(np.uint64), np.asarray(v["results_cwz"]).view(np.uint64)):
            return False
        if not np.array_equal(np.float64(u["asdef"]).view(np.uint64), np.float64(v["asdef"]).view(np.uint64)):
            return False
        if not np.array_equal(u["results_calls"], v["results_calls"]):
            return False
        for key in ("results_z", "results_r"):
            for cu, cv in zip(u[key], v[key]):
                if not np.array_equal(np.asarray(cu).view(np.uint64), np.asarray(cv).view(np.uint64)):
                    return False
        return True

    assert all(same(u, v) for u, v in zip(a, b))            # a second run is bit-identical
    planted = sum(len(a[i]["results_calls"]) for i in range(0, 24, 3))
    assert planted >= 8                                      # the planted gains are found
    for i in (0, 7, 23):                                     # a sample alone == the sample in the batch
        one = wt.test_batch(reference, [tests[i]], thr)[0]
        assert same(one, a[i]), i

    # effect size of the long call (median by radix selection) against numpy on the inflated ratios
    long_calls = [c for c in a[1]["results_calls"] if c[2] - c[1] > 1100]
    assert long_calls, a[1]["results_calls"]
    for chrom, start, end, _, effect in long_calls:
        r = np.asarray(a[1]["results_r"][int(chrom) - 1][int(start):int(end)])
        kept = np.sort(r[r != 0.0])                      # removed bins inflate to exactly 0
        # [start, end) misses the segment's last kept bin (the reference's end quirk), so the
        # median of the true set sits within one order statistic of this set's middle
        n = kept.shape[0]
        assert kept[n // 2 - 2] <= effect <= kept[n // 2 + 2], (chrom, start, end, effect, np.median(kept))


def _extreme(z, lo, hi, sign):
    """Largest sign * Stouffer value among the windows of z[lo:hi] -- prefix sums to find the
    near-maximal windows (vectorised over the window end), the reference's own expression
    np.sum(z[x:y+1]) / np.sqrt(len) to rank them, first in triangle (x-major) order on ties."""
    n = hi - lo
    P = np.concatenate([[0.0], np.cumsum(z[lo:hi])])
    best, cands = -np.inf, []
    for x in range(n):
        ln = np.arange(1, n - x + 1)
        v = sign * (P[x + ln] - P[x]) / np.sqrt(ln)
        m = v.max()
        if m >= best - 1e-9:
            best = max(best, m)
            cands += [(x, x + int(yy)) for yy in np.nonzero(v >= best - 1e-9)[0]]
    exact = [(sign * (np.sum(z[lo + x:lo + y + 1]) / np.sqrt(y - x + 1)), x, y) for x, y in cands]
    top = max(e[0] for e in exact)
    x, y = min((e[1], e[2]) for e in exact if e[0] == top)
    return sign * top, x, y


def _segments(z, thr, min_search, lo, hi):
    """triarray.py:59-84 on the windows of z[lo:hi] without the triangle."""
    out, n = [], hi - lo
    if n <= 0:
        return out
    cv, cx, cy = _extreme(z, lo, hi, +1)
    bv, bx, by = _extreme(z, lo, hi, -1)
    if abs(bv) > cv:
        cv, cx, cy = bv, bx, by
    if abs(cv) < thr:
        return out
    if cx > min_search:
        out += _segments(z, thr, min_search, lo, lo + cx)
    out.append((cv, (lo + cx, lo + cy)))
    if cy + 1 < n - min_search:
        out += _segments(z, thr, min_search, lo + cy + 1, hi)
    return out


@pytest.mark.parametrize("n", [6500, 8193, 12000])
def test_segments_regions_beyond_the_staged_sizes(wt, n):
    """Regions longer than the search stages in LDS (6 143 bins) and longer than the quiet
    certificate's block table covers (8 192 bins): prefix slice read from global memory, no
    certificate.  The oracle's triangle would need n^2 / 2 numpy calls; the restatement above
    ranks only the near-extreme windows exactly."""
    rng = np.random.RandomState(n)
    z = rng.standard_normal(n)
    z[n // 3:n // 3 + 150] += 0.8
    z[2 * n // 3:2 * n // 3 + 40] -= 1.5
    whole, got = wt.stouffer_segments([z, rng.standard_normal(50)], 5.0, 3)
    want = _segments(z, 5.0, 3, 0, n)
    assert len(want) >= 2
    assert [xy for _, xy in got[0]] == [xy for _, xy in want]
    assert np.array_equal(np.array([v for v, _ in got[0]]).view(np.int64), np.array([v for v, _ in want]).view(np.int64))
    assert np.float64(whole[0]).view(np.int64) == np.float64(np.sum(z) / np.sqrt(n)).view(np.int64)


def test_segment_longer_than_numpys_buffer(wt):
    """A call of ~9 500 bins: np.sum over more than 8 192 contiguous values is not one pairwise
    tree but buffer-sized pieces accumulated left to right (the whole-region value above 8 192
    bins as well); window value and coordinates must still be numpy's."""
    n = 12001
    rng = np.random.RandomState(77)
    z = rng.standard_normal(n)
    z[1000:10500] += 0.2
    whole, got = wt.stouffer_segments([z], 6.0, 3)
    want = _segments(z, 6.0, 3, 0, n)
    assert max(y - x for _, (x, y) in want) > 8192
    assert [xy for _, xy in got[0]] == [xy for _, xy in want]
    assert np.array_equal(np.array([v for v, _ in got[0]]).view(np.int64), np.array([v for v, _ in want]).view(np.int64))
    assert np.float64(whole[0]).view(np.int64) == np.float64(np.sum(z) / np.sqrt(n)).view(np.int64)
