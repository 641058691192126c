"""GPU parity of the HIP newref path (through the C ABI) against the golden
fixtures and the CPU oracle.  Indices must be identical, distances bit-equal."""
import numpy as np
import pytest

from oracle import wc_oracle as wo

pytestmark = pytest.mark.gpu


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


@pytest.mark.parametrize("name", ["plain", "ties", "fewcand", "special", "allsame", "deep"])
def test_golden_kernel_cases(golden, wt, name):
    g = golden("newref_kernel.npz")
    data, bins, k = g[name + "_data"], g[name + "_bins"], int(g[name + "_k"])
    sums = np.cumsum(bins)
    for parts in g[name + "_parts"]:
        for part in range(1, int(parts) + 1):
            idx, dst = wt.getReference(data, bins, sums, k, part, int(parts))
            assert np.array_equal(idx, g["%s_idx_%d_%d" % (name, part, parts)]), (name, part, parts)
            assert same_bits(dst, g["%s_dst_%d_%d" % (name, part, parts)]), (name, part, parts)


def test_golden_layout_cases(golden, wt):
    """Single-row concatenate pieces, pinned on the reference's own output."""
    g = golden("layout_cases.npz")
    for name in g["newref_names"]:
        data = g[name + "_data"]
        if bool(g[name + "_fortran"]):
            data = np.asfortranarray(data)
        bins = g[name + "_bins"]
        idx, dst = wt.getReference(data, bins, np.cumsum(bins), 3, 1, 1)
        assert np.array_equal(idx, g[name + "_idx"]), name
        assert same_bits(dst, g[name + "_dst"]), name


def test_golden_cfg1_prep_seam(golden, wt):
    g = golden("cfg1_pipeline.npz")
    data = g["prep_correctedData"]
    bins, sums = g["prep_maskedChromBins"], g["prep_maskedChromBinSums"]
    idx, dst = wt.getReference(data, bins, sums, 100, 1, 1)
    assert np.array_equal(idx, g["ref_indexes"])
    assert same_bits(dst, g["ref_distances"])
    st = wt.newref_stats()
    assert st["fast_rows"] + st["fallback_rows"] == data.shape[0]
    assert st["fallback_rows"] < data.shape[0] // 10, st
    cat_i, cat_d = [], []
    for part in range(1, 6):
        i, d = wt.getReference(data, bins, sums, 100, part, 5)
        cat_i.append(i)
        cat_d.append(d)
    assert np.array_equal(np.concatenate(cat_i), g["ref_indexes"])
    assert same_bits(np.concatenate(cat_d), g["ref_distances"])


@pytest.mark.parametrize("order", ["C", "F"])
@pytest.mark.parametrize("n_samples,binsize,k", [(16, 1000000, 100), (100, 1000000, 50), (257, 2000000, 100)])
def test_oracle_synthetic(wt, n_samples, binsize, k, order):
    """C-ordered input -> numpy pairwise bits; F-ordered (prep-file layout) -> sequential bits."""
    from wisecondor_amd import synth
    data, bins, sums = synth.corrected_matrix(binsize, n_samples, seed=3)
    if order == "F":
        data = np.asfortranarray(data)
    idx, dst = wt.getReference(data, bins, sums, k, 1, 1)
    st = wt.newref_stats()
    want_i, want_d = wo.get_reference(data, bins, sums, k, 1, 1, fast=True)
    assert np.array_equal(idx, want_i)
    assert same_bits(dst, want_d)
    assert st["fallback_rows"] <= data.shape[0] // 20, st


@pytest.mark.parametrize("order", ["C", "F"])
def test_more_samples_than_the_row_cache_holds(wt, order):
    """Above 2048 samples the target row of the float64 re-score is read from global memory
    instead of LDS, and numpy's pairwise tree is several levels deep."""
    rng = np.random.RandomState(11)
    bins = np.array([70, 1, 55, 64], dtype=np.int64)
    B, S, k = int(bins.sum()), 2500, 40
    data = 1.0 + 0.03 * rng.standard_normal((B, S))
    data[5] = data[150]
    if order == "F":
        data = np.asfortranarray(data)
    sums = np.cumsum(bins)
    idx, dst = wt.getReference(data, bins, sums, k, 1, 1)
    want_i, want_d = wo.get_reference(data, bins, sums, k, 1, 1, fast=True)
    assert np.array_equal(idx, np.asarray(want_i).reshape(-1, k))
    assert same_bits(dst, np.asarray(want_d, dtype=np.float64).reshape(-1, k))


@pytest.mark.parametrize("layout", [[1, 2], [1, 1], [3, 0, 1], [1, 40], [1, 2, 1], [1, 30, 1], [1, 0, 5, 1],
                                    [2, 3, 1], [1, 3, 2]])
@pytest.mark.parametrize("order", ["C", "F"])
def test_single_row_pieces(wt, layout, order):
    """Chromosomes with at most one bin before and at most one after them: the reference's
    chromData = concatenate(rows before, rows after) then consists of single-row pieces, which
    carry no layout -- it comes out C ordered and numpy sums its rows pairwise even when
    correctedData is Fortran ordered (everything else in such a file sums sample by sample).
    Found by the seed sweep."""
    rng = np.random.RandomState(31)
    bins = np.array(layout, dtype=np.int64)
    B, S, k = int(bins.sum()), 200, 5
    data = 1.0 + 0.03 * rng.standard_normal((B, S))
    if order == "F":
        data = np.asfortranarray(data)
    sums = np.cumsum(bins)
    for parts in (1, 2):
        for part in range(1, parts + 1):
            idx, dst = wt.getReference(data, bins, sums, k, part, parts)
            want_i, want_d = wo.get_reference(data, bins, sums, k, part, parts, fast=False)
            assert np.array_equal(idx, np.asarray(want_i).reshape(-1, k)), (part, parts)
            assert same_bits(dst, np.asarray(want_d, dtype=np.float64).reshape(-1, k)), (part, parts)


def test_oracle_hard_ties_and_outliers(wt):
    """Duplicated rows (exact ties across chromosomes), an outlier bin and a NaN bin."""
    from wisecondor_amd import synth
    data, bins, sums = synth.corrected_matrix(2000000, 24, seed=5)
    rng = np.random.RandomState(1)
    src = rng.randint(0, data.shape[0], size=40)
    dstrows = rng.randint(0, data.shape[0], size=40)
    data[dstrows] = data[src]
    data[17] *= 40.0
    data[333, 2] = np.nan
    idx, dst = wt.getReference(data, bins, sums, 30, 1, 1)
    with np.errstate(all="ignore"):
        want_i, want_d = wo.get_reference(data, bins, sums, 30, 1, 1, fast=True)
    assert np.array_equal(idx, want_i)
    assert same_bits(dst, want_d)


def test_large_slice_against_oracle(wt):
    """cfg2 shape (100 samples x 250 kb): a row slice checked against the oracle."""
    from wisecondor_amd import synth
    data, bins, sums = synth.corrected_matrix(250000, 100, seed=0)
    parts = 64
    part = 23
    idx, dst = wt.getReference(data, bins, sums, 100, part, parts)
    want_i, want_d = wo.get_reference(data, bins, sums, 100, part, parts, fast=True)
    assert np.array_equal(idx, want_i)
    assert same_bits(dst, want_d)


def test_error_reporting_through_the_c_abi(wt):
    """Bad arguments come back as error codes + messages; nothing exits or falls back."""
    from wisecondor_amd import _lib
    data = np.ones((10, 4))
    with pytest.raises(_lib.WisecondorHipError, match="chromosome sizes sum"):
        wt.getReference(data, [3, 3], [3, 10], 2)            # sizes do not add up to the row count
    with pytest.raises(_lib.WisecondorHipError, match="refsize above"):
        wt.getReference(data, [5, 5], [5, 10], 100000)
    with pytest.raises(ValueError):
        wt.getReference(np.ones((9, 4)), [5, 5], [5, 10], 2)  # matrix / layout mismatch caught on the host
    with pytest.raises(_lib.WisecondorHipError, match="mask has"):
        wt.Reference(np.zeros((4, 2), np.int32), np.ones((4, 2)), [3, 3], [2, 2], np.array([1, 1, 1, 0, 1, 0], np.uint8),
                     np.ones(4), np.zeros((0, 4)))
    with pytest.raises(_lib.WisecondorHipError, match="unsupported shape"):
        wt.Reference(np.zeros((4, 2000), np.int32), np.ones((4, 2000)), [2, 2], [2, 2], np.ones(4, np.uint8),
                     np.ones(4), np.zeros((0, 4)))           # refsize 2000 > 1024 in the test path
    # and the context is still usable afterwards
    idx, dst = wt.getReference(np.arange(40.0).reshape(10, 4), [5, 5], [5, 10], 2)
    assert idx.shape == (10, 2) and np.all(idx >= 0)


def test_thresholds_are_reproducible():
    """The admission thresholds (bf16 MFMA estimate -> 16-bit key codes -> order statistic) must
    come out identical on every call: a run-to-run difference once exposed a packed-float32
    hazard in the key-code epilogue (see the note on -fno-slp-vectorize in build.py)."""
    import torch
    from wisecondor_amd import _lib, distributed, synth
    data, bins, _ = synth.corrected_matrix(250000, 100, seed=3)
    X = torch.from_numpy(data).cuda()
    job = distributed.NewrefJob(_lib.context(0), X, bins, 100, _lib.SUM_SEQUENTIAL)
    st = job.st
    seen = []
    for it in range(3):
        st.prepare()
        st.thresholds(0, st.n_bins)
        t = torch.empty(st.n_bins, dtype=torch.float32, device="cuda")
        st.get_thr(0, st.n_bins, t)
        torch.cuda.synchronize()
        seen.append(t.cpu().numpy())
        if it == 0:
            job.run()          # dirty every workspace in between
    assert np.array_equal(seen[0], seen[1]) and np.array_equal(seen[0], seen[2])
    assert np.isfinite(seen[0]).all() and (seen[0] > 0).all()


def _check_listed_bounds(data, bins, order, step, kind):
    """Run prepare / thresholds / collect on `data` and check BOTH sides of the interval on every listed pair of
    every step-th row.  Returns (job, stages, pairs checked, worst relative key gap)."""
    import torch
    from wisecondor_amd import _lib, distributed
    X = torch.from_numpy(np.ascontiguousarray(data)).cuda()
    job = distributed.NewrefJob(_lib.context(0), X, bins, 100, order)
    st = job.st
    st.prepare()
    st.thresholds(0, st.n_bins)
    st.collect(0, st.n_bins, 0, 1)
    cap = st.cap
    cnt = torch.zeros(st.n_bins, dtype=torch.int32, device="cuda")
    lst = torch.zeros((st.n_bins, cap), dtype=torch.int64, device="cuda")
    st.export(0, st.n_bins, cap, cnt, lst)
    lo_t = torch.zeros(st.n_bins, dtype=torch.float32, device="cuda")
    slack_t = torch.zeros(st.n_bins, dtype=torch.float32, device="cuda")
    st.get_bounds(0, st.n_bins, lo_t, slack_t)
    torch.cuda.synchronize()
    slack = slack_t.cpu().numpy().astype(np.float64)
    cnt = cnt.cpu().numpy()
    rows = np.arange(0, st.n_bins, step)
    lst = lst[torch.from_numpy(rows).cuda()].cpu().numpy().view(np.uint64)
    checked, worst = 0, 0.0
    with np.errstate(all="ignore"):
        for at, i in enumerate(rows):
            n = min(int(cnt[i]), cap)
            if n == 0:
                continue
            e = lst[at, :n]
            j = (e & np.uint64(0xFFFFFFFF)).astype(np.int64)
            u = (e >> np.uint64(32)).astype(np.uint32)
            bits = np.where(u & np.uint32(0x80000000), u & np.uint32(0x7FFFFFFF), ~u)
            key = bits.astype(np.uint32).view(np.float32).astype(np.float64)
            d = ((data[j] - data[i]) ** 2).sum(1)
            ok = np.isfinite(d)                      # a non-finite distance is never admitted, whatever its key says
            assert (key[ok] <= d[ok] * (1 + 1e-12) + 1e-300).all(), (kind, i)
            hi = key + (slack[i] + slack[j]) * (1 + 1e-12) + 1e-300
            assert (d[ok] <= hi[ok]).all(), (kind, i)
            scale = (data[j] ** 2).sum(1) + (data[i] ** 2).sum()
            gap = ((d - key) / np.maximum(scale, 1e-300))[ok & np.isfinite(key)]
            if gap.size:
                worst = max(worst, float(gap.max()))
            checked += n
    return job, st, checked, worst


def _spoil(data, kind, seed=3):
    """The data kinds of the bound tests (in place on a copy)."""
    rng = np.random.RandomState(seed)
    data = data.copy()
    B, S = data.shape
    if kind == "wide":
        data = data * np.exp(rng.uniform(-4.0, 4.0, size=(B, 1)))      # per-row scale e^-4 .. e^4
        data += rng.standard_normal(data.shape) * 1e-3
    elif kind == "huge":
        data = data * 1e6
    elif kind == "tiny":
        data = data * 1e-9
    elif kind == "clamped":
        # values far beyond +-65504 / gam (the float16 image clamps them): single spikes, whole outlier rows
        # and one outlier SAMPLE column; the rows concerned lose their certificate, their bounds must still hold
        rows = rng.choice(B, 40, replace=False)
        data[rows[:20], rng.randint(0, S, 20)] += 1e4 * rng.choice([-1.0, 1.0], 20)
        data[rows[20:30]] *= 5e4
        data[rows[30:], :] = 1.0 + 3e5 * rng.standard_normal((10, S))
        data[:, S // 2] += 50.0 * (rng.rand(B) < 0.01)
    elif kind == "flushed":
        # centred values below the float16 subnormal flush (|a| gam < 6.1e-5, i.e. 1e-7 of the usual spread):
        # rows that sit on the per-sample centre the kernel subtracts (k_col_centre: trimmed mean over 128 strided
        # rows, restated here -- agreement to 1e-12 is plenty), and rows that mix such entries with normal ones
        n_rows = min(B, 128)
        step = B // n_rows
        sub = data[::step][:n_rows]
        c = sub.mean(0)
        rad = 8.0 * np.abs(sub - c).mean(0)
        keep = np.abs(sub - c) <= rad
        centre = (sub * keep).sum(0) / keep.sum(0)
        free = np.setdiff1d(np.arange(B), np.arange(n_rows) * step)       # not among the rows the centre is made of
        rows = rng.choice(free, 60, replace=False)
        data[rows[:30]] = centre + 2e-9 * rng.standard_normal((30, S))
        mix = rng.rand(30, S) < 0.5
        data[rows[30:]] = np.where(mix, centre + 2e-9 * rng.standard_normal((30, S)), data[rows[30:]])
    return data


@pytest.mark.parametrize("kind", ["noise", "wide", "huge", "tiny", "clamped", "flushed"])
def test_listed_keys_are_lower_bounds(kind):
    """Every decision of the fast path rests on key <= true distance <= key + slack_i + slack_j.
    The candidate lists expose the keys and wc_newref_get_bounds_dev the slacks: check both sides
    on every listed pair of the one-product float16 tiles, on plain noise, on rows whose magnitudes
    span four orders (16-bit operands lose the low bits of large values), on matrices far outside
    float16's own range (1e6 and 1e-9 times the usual magnitudes: the image is scaled), with values
    the image clamps (beyond +-65504 / gam) and with values it flushes to zero."""
    import torch
    from wisecondor_amd import _lib, synth
    data, bins, _ = synth.corrected_matrix(1000000, 200, seed=8)
    data = _spoil(data, kind)
    job, st, checked, worst = _check_listed_bounds(data, bins, _lib.SUM_PAIRWISE, 7, kind)
    assert checked > 10000
    # the slack relative to |a|^2 + |b|^2 of the raw rows; on plain noise the kernel's centred rows
    # have smaller norms than the raw ones, so this is far inside 3 * beta (the wide case centres
    # rows of very different scale on one common centre: no such yardstick)
    if kind == "noise":
        assert worst < 1e-3
    if kind in ("noise", "huge", "tiny"):
        # the fast path must actually carry these rows: a bound so loose that everything falls back
        # to the exact scan would pass the checks above
        from wisecondor_amd import wisetools as wt
        job.run()
        torch.cuda.synchronize()
        stats = wt.newref_stats()
        assert stats["fallback_rows"] == 0 and stats["fast_rows"] == st.n_bins, stats
        assert stats["rescored"] < 1.25 * 100 * st.n_bins, stats       # ~k candidates per row, not the whole list


@pytest.mark.parametrize("kind", ["noise", "pipeline", "clamped", "flushed"])
def test_listed_keys_are_lower_bounds_at_600_samples(kind):
    """The same interval at the sample count the headline configuration trusts it at (600 samples change
    beta, the accumulation chain and the clamp statistics), 28 783 bins (100 kb): plain noise, a matrix made by
    the prep pipeline (correlated rows, Fortran order), clamped and flushed values.  On the clean kinds the whole
    result is then held against the exact path for every row."""
    import torch
    from wisecondor_amd import _lib, synth
    from wisecondor_amd import wisetools as wt
    if kind == "pipeline":
        profile = synth.bin_profile(100000)
        samples = [synth.make_sample(profile, seed=i) for i in range(600)]
        _, _, _, data, _, _, bins = wt.prepReference(samples)
        bins = np.asarray(bins, dtype=np.int64)
        order = wt.sum_order_of(data)
        assert order == _lib.SUM_SEQUENTIAL
        data = np.ascontiguousarray(data)
    else:
        data, bins, _ = synth.corrected_matrix(100000, 600, seed=11)
        data = _spoil(data, kind, seed=5)
        order = _lib.SUM_PAIRWISE
    assert data.shape[0] >= 20000 and data.shape[1] == 600
    job, st, checked, worst = _check_listed_bounds(data, bins, order, 61, kind)
    assert checked > 100000
    idx, dst = job.run()
    torch.cuda.synchronize()
    stats = wt.newref_stats()
    if kind in ("noise", "pipeline"):
        assert stats["fallback_rows"] == 0 and stats["fast_rows"] == st.n_bins, stats
    elif kind == "clamped":
        assert 0 < stats["fallback_rows"] < 2000, stats       # the spoiled rows take the exact path -- and nobody else
    else:
        assert stats["fallback_rows"] < 2000, stats
    idx, dst = idx.clone(), dst.clone()
    ex_i, ex_d = torch.empty_like(idx), torch.empty_like(dst)
    st.exact(0, st.n_bins, ex_i, ex_d)
    torch.cuda.synchronize()
    assert torch.equal(idx, ex_i)
    assert torch.equal(dst.view(torch.int64), ex_d.view(torch.int64))
