"""GPU parity of the HIP `test` path (through the C ABI) against golden fixtures
produced by the reference and against the CPU oracle.

Contract: call coordinates, segment bounds and reference counts exact; z-scores,
ratios and effect sizes within 1e-9 relative (float64 everywhere; the only
non-bit-exact step is the PCA projection, whose BLAS summation order is not
reproducible), stated next to each assertion.
"""
import os
import warnings

import numpy as np
import pytest

from oracle import wc_oracle as wo

pytestmark = pytest.mark.gpu
KEYS = [str(c) for c in range(1, 23)] + ["X", "Y"]
NAMES = ["mild18", "gain5_gap", "gain5_past", "gain5_after", "loss2", "normal"]


def same_bits(a, b):
    a = np.asarray(a, dtype=np.float64)
    b = np.asarray(b, dtype=np.float64)
    if a.shape != b.shape:
        return False
    nan = np.isnan(a) & np.isnan(b)
    return bool(np.all(nan | (a.view(np.int64) == b.view(np.int64))))


def _split(flat, lengths):
    offs = np.concatenate([[0], np.cumsum(lengths)])
    return {k: flat[offs[i]:offs[i + 1]] for i, k in enumerate(KEYS)}


@pytest.fixture(scope="module")
def wt():
    from wisecondor_amd import wisetools
    return wisetools


@pytest.fixture(scope="module")
def cfg1(golden):
    return golden("cfg1_pipeline.npz")


@pytest.fixture(scope="module")
def reference(wt, cfg1):
    g = cfg1
    ref = wt.Reference(g["ref_indexes"], g["ref_distances"], g["ref_chromosome_sizes"], g["ref_masked_sizes"],
                       g["ref_mask"], g["ref_pca_mean"], g["ref_pca_components"], binsize=float(g["ref_binsize"]))
    yield ref
    ref.close()


def test_cutoff(wt, cfg1, reference):
    # the moments are taken in numpy's order (row-major compaction, pairwise within 8192-element
    # pieces, pieces left to right): the cutoff is the reference's double
    assert reference.cutoff == float(cfg1["cutoff"])
    cut, mask = wt.getOptimalCutoff(cfg1["ref_distances"], 3)
    assert cut == float(cfg1["cutoff"])
    _, want_mask = wo.get_optimal_cutoff(cfg1["ref_distances"], 3)
    assert mask.dtype == np.bool_ and mask.shape == want_mask.shape and np.array_equal(mask, want_mask)
    assert not mask.all() and mask.any()                    # the second iteration's cutoff does clip here
    c0, m0 = wt.getOptimalCutoff(cfg1["ref_distances"], 0)  # the loop never runs (wisetools.py:329-330)
    w0, wm0 = wo.get_optimal_cutoff(cfg1["ref_distances"], 0)
    assert c0 == w0 == float("inf") and m0.dtype == wm0.dtype and np.array_equal(m0, wm0)
    # both return values against the real reference's (tools/make_goldens.py --only cutoffmask)
    gm = np.load(os.path.join(os.path.dirname(__file__), "golden", "cutoff_mask.npz"), allow_pickle=False)
    for name, arr in (("cfg1", cfg1["ref_distances"]), ("small", gm["small"])):
        for repeats in (0, 1, 2, 3):
            got, got_mask = wt.getOptimalCutoff(arr, repeats)
            assert got == float(gm["%s_cutoff_%d" % (name, repeats)])
            want = np.unpackbits(gm["%s_mask_%d" % (name, repeats)])[:arr.size].astype(bool).reshape(arr.shape)
            assert np.array_equal(got_mask.astype(bool), want) and str(got_mask.dtype) == str(gm["%s_maskdtype_%d" % (name, repeats)])
    rng = np.random.RandomState(2)
    for shape in [(1, 1), (3, 7), (90, 100), (700, 100), (8192, 1), (8193, 1), (3000, 37)]:
        d = np.sort(rng.gamma(3.0, 0.1, size=shape), axis=1)
        d[rng.rand(*shape) < 0.01] = 1e10                 # padding entries of short lists
        for repeats in (1, 3):
            got, got_mask = wt.getOptimalCutoff(d, repeats)
            with warnings.catch_warnings():
                warnings.simplefilter("ignore")        # a one-element set clips itself away: numpy warns
                want, want_mask = wo.get_optimal_cutoff(d, repeats)
            assert (np.isnan(got) and np.isnan(want)) or got == want, (shape, repeats)
            assert got_mask.shape == want_mask.shape and np.array_equal(got_mask, want_mask), (shape, repeats)


@pytest.mark.parametrize("name", NAMES)
def test_prepare_and_pca(wt, cfg1, name):
    g = cfg1
    sample = _split(g["t_%s_sample" % name], g["sample_chrom_lengths"])
    x = wt.toNumpyRefFormat(sample, g["ref_chromosome_sizes"], g["ref_mask"])
    assert same_bits(x, g["t_%s_x" % name])          # integer sums and one division: bit-exact
    xp = wt.applyPCA(g["t_%s_x" % name], g["ref_pca_mean"], g["ref_pca_components"])
    assert np.allclose(xp, g["t_%s_xpca" % name], rtol=1e-12, atol=0)


@pytest.mark.parametrize("reps", [1, 2, 5])
def test_repeat_test_bits(wt, cfg1, reps):
    """From the golden PCA output the z-score repeats must reproduce numpy's bits."""
    g = cfg1
    data = np.stack([g["t_%s_xpca" % n] for n in NAMES])
    thr = float(g["t_mild18_threshold_z"])
    ms = g["ref_masked_sizes"]
    z, r, n, sd = wt.repeatTest(data, g["ref_indexes"], g["ref_distances"], ms, np.cumsum(ms),
                                float(g["cutoff"]), thr, reps)
    for row, name in enumerate(NAMES):
        assert np.array_equal(n[row], g["t_%s_rep%d_n" % (name, reps)]), name
        assert same_bits(z[row], g["t_%s_rep%d_z" % (name, reps)]), name
        if reps == 5:
            assert same_bits(r[row], g["t_%s_rep5_r" % name]), name
            assert sd[row] == float(g["t_%s_rep5_sd" % name]), name


def test_segments_golden(wt, golden):
    g = golden("segments.npz")
    for i in range(int(g["n_cases"])):
        z = g["z_%d" % i]
        thr = float(g["thresholds"][i])
        whole, segs = wt.stouffer_segments([z], thr, 3)
        want = g["seg_%d" % i]
        got = np.array([[v, x, y] for v, (x, y) in segs[0]], dtype=np.float64).reshape(-1, 3)
        assert np.array_equal(got[:, 1:], want[:, 1:]), (i, got, want)
        assert same_bits(got[:, 0], want[:, 0]), (i, got, want)
        tri = g["tri_%d" % i]
        assert same_bits([whole[0]], [tri[len(z) - 1]]), i      # getValue(0, n-1)
    # several regions in one call
    zs = [g["z_%d" % i] for i in range(14, int(g["n_cases"]))]
    _, segs = wt.stouffer_segments(zs, 3.0, 3)
    for z, s in zip(zs, segs):
        tri = wo.fill_tri(z)
        want = wo.segment_tri(tri, z.shape[0], 3.0, 3)
        assert [(x, y) for _, (x, y) in s] == [(x, y) for _, (x, y) in want]
        assert same_bits([v for v, _ in s], [v for v, _ in want])


def test_segments_golden_long_region(wt, golden):
    """8400 bins, one call longer than numpy's 8192-element buffer: the reference's own output."""
    g = golden("layout_cases.npz")
    z = g["long_z"]
    whole, segs = wt.stouffer_segments([z], float(g["long_thr"]), 3)
    want = g["long_seg"]
    got = np.array([[v, x, y] for v, (x, y) in segs[0]], dtype=np.float64).reshape(-1, 3)
    assert np.array_equal(got[:, 1:], want[:, 1:])
    assert same_bits(got[:, 0], want[:, 0])
    assert same_bits([whole[0]], [g["long_whole"]])


def test_triarr_mirror(wt, golden):
    from wisecondor_amd.triarray import TriArr
    g = golden("segments.npz")
    z = g["z_27"]
    tri = wt.fillTriMin(z, np.ones_like(z), 0)
    assert isinstance(tri, TriArr)
    want = g["tri_27"]
    n = len(z)
    assert tri.getValue(0, n - 1) == want[n - 1]
    assert tri.getValue(3, 9) == want[wo.tri_offset(n, 3, 9)]
    assert tri.linTo2D(wo.tri_offset(n, 13, 22)) == (13, 22)
    segs = tri.segmentTri(float(g["thresholds"][27]), 3)
    assert [(x, y) for _, (x, y) in segs] == [tuple(int(v) for v in row[1:]) for row in g["seg_27"]]


def test_fill_tri_min_golden(wt, golden):
    """-mineffectsize branch (wisetools.py:479-487): window values and segments of the filtered triangle."""
    g = golden("segments.npz")
    for i in range(int(g["mincase_n"])):
        z, r, eff = g["mincase_z_%d" % i], g["mincase_r_%d" % i], float(g["mincase_eff_%d" % i])
        tri = wt.fillTriMin(z, r, eff)
        want_tri = g["mincase_tri_%d" % i]
        n = len(z)
        rng = np.random.RandomState(i)
        for _ in range(min(12, n * (n + 1) // 2)):
            x = int(rng.randint(0, n))
            y = int(rng.randint(x, n))
            assert same_bits([tri.getValue(x, y)], [want_tri[wo.tri_offset(n, x, y)]]), (i, x, y)
        segs = tri.segmentTri(3.0, 3)
        want = g["mincase_seg_%d" % i]
        got = np.array([[v, x, y] for v, (x, y) in segs], dtype=np.float64).reshape(-1, 3)
        assert np.array_equal(got[:, 1:], want[:, 1:]), (i, got, want)
        assert same_bits(got[:, 0], want[:, 0]), i
    # batched, against the oracle, with ties and NaN ratios
    rng = np.random.RandomState(3)
    zs, rs = [], []
    for n in (5, 31, 77, 130):
        z = rng.standard_normal(n)
        r = np.round(1.0 + 0.05 * rng.standard_normal(n), 2)      # many equal ratios: median ties
        z[n // 2:n // 2 + 4] += 4
        r[n // 2:n // 2 + 4] += 0.1
        zs.append(z)
        rs.append(r)
    rs[2][5] = np.nan
    whole, segs = wt.stouffer_segments(zs, 3.0, 3, ratios=rs, mineffectsize=0.05)
    for z, r, w, s in zip(zs, rs, whole, segs):
        with np.errstate(all="ignore"):
            tri = wo.fill_tri_min(z, r, 0.05)
        want = wo.segment_tri(tri, len(z), 3.0, 3)
        assert [(x, y) for _, (x, y) in s] == [(x, y) for _, (x, y) in want]
        assert same_bits([v for v, _ in s], [v for v, _ in want])
        assert same_bits([w], [tri[len(z) - 1]])


@pytest.mark.parametrize("name", ["loss2", "gain5_gap"])
def test_cfg1_mineffectsize(wt, cfg1, reference, name):
    g = cfg1
    thr = float(g["t_mild18_threshold_z"])
    sample = _split(g["t_%s_sample" % name], g["sample_chrom_lengths"])
    out = wt.test_batch(reference, [sample], thr, mineffectsize=float(g["eff_mineffectsize"]))[0]
    want = g["eff_%s_results_calls" % name]
    got = out["results_calls"].reshape(-1, 5)
    assert np.array_equal(got[:, :3], want[:, :3]), (got, want)
    assert np.allclose(got[:, 3:], want[:, 3:], rtol=1e-9)
    assert np.allclose(out["results_cwz"], g["eff_%s_results_cwz" % name], rtol=1e-9, atol=1e-9)


def test_segments_random_vs_oracle(wt):
    rng = np.random.RandomState(11)
    regions = []
    for n in [1, 2, 3, 4, 5, 7, 33, 64, 65, 129, 200, 257, 300]:
        z = rng.standard_normal(n)
        if n > 20:
            a = rng.randint(0, n - 10)
            z[a:a + rng.randint(3, 10)] += rng.choice([-1, 1]) * 3.0
        regions.append(z)
    regions.append(np.zeros(40))                       # everything ties at zero
    regions.append(np.round(rng.standard_normal(60)))  # many exact ties between windows
    regions.append(np.array([2.0] * 50))               # monotone growth, single call
    thr = 3.5
    whole, segs = wt.stouffer_segments(regions, thr, 3)
    for z, w, s in zip(regions, whole, segs):
        tri = wo.fill_tri(z)
        want = wo.segment_tri(tri, z.shape[0], thr, 3)
        assert [(x, y) for _, (x, y) in s] == [(x, y) for _, (x, y) in want], z.shape
        assert same_bits([v for v, _ in s], [v for v, _ in want])
        assert same_bits([w], [tri[z.shape[0] - 1]])


def test_segments_long_regions_quiet_certificate(wt):
    """Regions of >= 2048 bins switch the quiet-job certificate on (block bounds on the prefix
    sums decide most jobs without a full window search): a quiet long region, a long region
    with calls (and therefore child ranges), a threshold-grazing one and short ones in the same
    call must still reproduce the reference's segments exactly."""
    rng = np.random.RandomState(23)
    quiet = rng.standard_normal(2100)
    busy = rng.standard_normal(2100)
    busy[300:420] += 0.9
    busy[1500:1530] -= 1.6
    graze = rng.standard_normal(2060) * 0.5
    regions = [quiet, busy, graze, rng.standard_normal(70), np.array([1.5] * 30)]
    thr = 5.0
    whole, segs = wt.stouffer_segments(regions, thr, 3)
    n_calls = 0
    for z, w, s in zip(regions, whole, segs):
        tri = wo.fill_tri(z)
        want = wo.segment_tri(tri, z.shape[0], thr, 3)
        assert [(x, y) for _, (x, y) in s] == [(x, y) for _, (x, y) in want], z.shape
        assert same_bits([v for v, _ in s], [v for v, _ in want])
        assert same_bits([w], [tri[z.shape[0] - 1]])
        n_calls += len(want)
    assert n_calls >= 3


@pytest.mark.parametrize("batch", [False, True])
def test_cfg1_whole_test(wt, cfg1, reference, batch):
    g = cfg1
    thr = float(g["t_mild18_threshold_z"])
    samples = [_split(g["t_%s_sample" % n], g["sample_chrom_lengths"]) for n in NAMES]
    if batch:
        outs = wt.test_batch(reference, samples, thr)
    else:
        outs = [wt.test_batch(reference, [s], thr)[0] for s in samples]
    for name, out in zip(NAMES, outs):
        want = g["t_%s_results_calls" % name]
        got = out["results_calls"].reshape(-1, 5)
        assert np.array_equal(got[:, :3], want[:, :3]), (name, got, want)   # chromosome, start, end: exact
        assert np.allclose(got[:, 3:], want[:, 3:], rtol=1e-9, atol=0), name
        assert np.allclose(np.concatenate(out["results_z"]), g["t_%s_results_z" % name], rtol=1e-9, atol=1e-11)
        assert np.allclose(np.concatenate(out["results_r"]), g["t_%s_results_r" % name], rtol=1e-9, atol=1e-12)
        assert np.allclose(out["results_cwz"], g["t_%s_results_cwz" % name], rtol=1e-9, atol=1e-9)
        assert np.isclose(out["asdef"], float(g["t_%s_asdef" % name]), rtol=1e-11)
        zero = np.concatenate(out["results_z"]) == 0
        assert np.array_equal(zero, g["t_%s_results_z" % name] == 0)        # same removed-bin pattern


def test_cfg1_options(wt, cfg1, reference):
    g = cfg1
    sample = _split(g["t_gain5_gap_sample"], g["sample_chrom_lengths"])
    out = wt.test_batch(reference, [sample], 4.0, minrefbins=40, repeats=2, chromosomes=[2, 5, 18])[0]
    want = g["opts_results_calls"]
    got = out["results_calls"].reshape(-1, 5)
    assert np.array_equal(got[:, :3], want[:, :3])
    assert np.allclose(got[:, 3:], want[:, 3:], rtol=1e-9)
    assert np.allclose(out["results_cwz"], g["opts_results_cwz"], rtol=1e-9)
    assert np.allclose(np.concatenate(out["results_z"]), g["opts_results_z"], rtol=1e-9, atol=1e-11)
    assert np.isclose(out["asdef"], float(g["opts_asdef"]), rtol=1e-11)


@pytest.mark.parametrize("opts", [dict(mineffectsize=0.02), dict(minrefbins=40, repeats=2, chromosomes=[2, 5, 18]),
                                  dict(repeats=1)])
def test_options_in_a_padded_batch(wt, cfg1, reference, opts):
    """45 samples in one call -- the z-score stage then runs on 48 sample columns (three of padding) -- with
    -mineffectsize, with a chromosome selection / minrefbins / two repeats, and with one repeat: every output equals
    the one-sample calls' (a different kernel path: no padding, the latency kernels), bit for bit."""
    g = cfg1
    thr = float(g["t_mild18_threshold_z"])
    six = [_split(g["t_%s_sample" % n], g["sample_chrom_lengths"]) for n in NAMES]
    samples = (six * 8)[:45]
    got = wt.test_batch(reference, samples, thr, **opts)
    want = [wt.test_batch(reference, [sm], thr, **opts)[0] for sm in six]
    for i, out in enumerate(got):
        w = want[i % len(six)]
        assert np.array_equal(np.asarray(out["results_calls"], dtype=np.float64).view(np.uint64),
                              np.asarray(w["results_calls"], dtype=np.float64).view(np.uint64)), (i, opts)
        assert same_bits(np.concatenate(out["results_z"]), np.concatenate(w["results_z"])), (i, opts)
        assert same_bits(np.concatenate(out["results_r"]), np.concatenate(w["results_r"])), (i, opts)
        assert same_bits(out["results_cwz"], w["results_cwz"]), (i, opts)
        assert same_bits([out["asdef"]], [w["asdef"]]), (i, opts)


def _seq_mean(v):
    """trySample's stdDevAvg: Python-style sequential float64 sum of the non-NaN terms / their count."""
    s, n = 0.0, 0
    for x in v:
        if x == x:
            s += float(x)
            n += 1
    return s / n if n else float("nan")


def test_std_dev_avg_parallel_form_is_exact(wt):
    """k_sd_fast (binade-wise integer maps, segmented scan) against the sequential sum, bit for bit:
    random magnitudes, forced rounding ties, sums that sit on powers of two, NaN gaps, zeros,
    huge and tiny terms; and it must really be the parallel form that answered."""
    rng = np.random.RandomState(11)
    rows = []
    for n in (1, 2, 7, 63, 64, 65, 1000, 1024, 1025, 11087, 20011, 57633):      # (20 011: the 13..24 values-per-thread form)
        rows.append((n, np.abs(0.03 + 0.01 * rng.standard_normal(n))))
        v = np.abs(rng.standard_normal(n)) * 10.0 ** rng.uniform(-6, 6, size=n)       # many binade crossings
        rows.append((n, v))
        v = np.ldexp(rng.randint(1, 1 << 10, size=n).astype(np.float64), -12)          # few mantissa bits: ties
        rows.append((n, v))
        v = np.full(n, 0.25)                                                            # exact powers of two along the way
        rows.append((n, v))
        v = np.abs(rng.standard_normal(n))
        v[rng.rand(n) < 0.2] = np.nan
        v[rng.rand(n) < 0.1] = 0.0
        rows.append((n, v))
    serial_total = 0
    for n, v in rows:
        got, serial = wt.stdDevAvg(v, return_serial_count=True)
        want = _seq_mean(v)
        assert (np.isnan(got) and np.isnan(want)) or got == want, (n, got, want)
        serial_total += serial
    assert serial_total <= 2, serial_total                   # the scan, not the fallback chain, did the work
    # ties that depend on the running sum's parity: terms of exactly half an ulp of the sum
    v = np.concatenate([[1.0], np.full(4000, 2.0 ** -53)])
    assert wt.stdDevAvg(v) == _seq_mean(v)
    v = np.concatenate([[1.0 + 2.0 ** -52], np.full(4000, 2.0 ** -53), [3.0], np.full(500, 2.0 ** -52)])
    assert wt.stdDevAvg(v) == _seq_mean(v)
    # a batch: every sample its own answer
    batch = np.abs(rng.standard_normal((70, 3000))) * 10.0 ** rng.uniform(-3, 3, size=(70, 1))
    got = wt.stdDevAvg(batch)
    assert all(got[i] == _seq_mean(batch[i]) for i in range(70))
    # terms the parallel form declines (inf, negative): the serial chain answers, same semantics
    v = np.abs(rng.standard_normal(500))
    v[100] = np.inf
    got, serial = wt.stdDevAvg(v, return_serial_count=True)
    assert got == _seq_mean(v) and serial == 1


@pytest.mark.parametrize("env", [{"WC_TEST_WALK": "0"}, {"WC_TEST_WALK": "0", "WC_TEST_TREE_TAIL": "0"},
                                 {"WC_TEST_WALK": "0", "WC_TEST_TREE_TAIL": "0", "WC_TEST_CELLS": "0"},
                                 {"WC_TEST_WALK": "0", "WC_TEST_TREE_TAIL": "0", "WC_CELL_PARTS": "1"}])
def test_every_segmentation_path_gives_the_walkers_outputs(wt, cfg1, reference, monkeypatch, env):
    """The batched `test` through the paths k_seg_walk replaced (the switches are read per call): the tree kernel
    after k_seg_quiet / k_seg_search / k_seg_classify, the host-driven levels with the cell search (k_seg_job /
    k_seg_merge; one workgroup per range or several), and those levels with the row-block kernels -- bit-identical
    calls, z, ratios and chromosome-wide values for a batch of 48 samples (the six cfg1 samples repeated)."""
    g = cfg1
    thr = float(g["t_mild18_threshold_z"])
    samples = [_split(g["t_%s_sample" % n], g["sample_chrom_lengths"]) for n in NAMES] * 8
    want = wt.test_batch(reference, samples, thr)
    for k_, v_ in env.items():
        monkeypatch.setenv(k_, v_)
    got = wt.test_batch(reference, samples, thr)
    for a, b in zip(want, got):
        assert np.array_equal(a["results_calls"], b["results_calls"])
        assert np.array_equal(np.asarray(a["results_cwz"]).view(np.uint64), np.asarray(b["results_cwz"]).view(np.uint64))
        assert np.array_equal(np.concatenate(a["results_z"]).view(np.uint64), np.concatenate(b["results_z"]).view(np.uint64))
        assert np.array_equal(np.concatenate(a["results_r"]).view(np.uint64), np.concatenate(b["results_r"]).view(np.uint64))


def test_long_regions_through_the_host_driven_levels(wt, monkeypatch):
    """wc_stouffer_segments on regions of 3 000 - 8 000 bins: the cell search with several workgroups per range (the
    default for few ranges), with one workgroup per range, and the row-block kernels give the same segments, bit for
    bit; every segment's value is its exact window value (the oracle's triangle is out of reach at this size)."""
    rng = np.random.RandomState(31)
    regions = []
    for n in (3000, 5000, 8000):
        z = rng.standard_normal(n)
        a = rng.randint(0, n - 700)
        z[a:a + 600] += 0.35
        z[n // 2:n // 2 + 25] -= 1.4
        regions.append(z)
    thr = 5.2
    whole, segs = wt.stouffer_segments(regions, thr, 3)
    found = 0
    for z, s in zip(regions, segs):
        found += len(s)
        xs = [x for _, (x, y) in s]
        assert xs == sorted(xs)
        for v, (x, y) in s:
            assert v == np.sum(z[x:y + 1]) / np.sqrt(y - x + 1) and abs(v) >= thr
        for (v0, (x0, y0)), (v1, (x1, y1)) in zip(s, s[1:]):
            assert y0 < x1
    assert found >= 3
    for env in ({"WC_CELL_PARTS": "1"}, {"WC_TEST_CELLS": "0"}):
        for k_, v_ in env.items():
            monkeypatch.setenv(k_, v_)
        whole2, segs2 = wt.stouffer_segments(regions, thr, 3)
        for k_ in env:
            monkeypatch.delenv(k_)
        assert same_bits(whole, whole2)
        for s, s2 in zip(segs, segs2):
            assert [(x, y) for _, (x, y) in s] == [(x, y) for _, (x, y) in s2]
            assert same_bits([v for v, _ in s], [v for v, _ in s2])


def test_whole_region_values_at_every_tree_shape(wt):
    """getValue(0, n - 1) = np.sum(z) / np.sqrt(n) (wisecondor.py:237) for region lengths around every change of shape of
    numpy's pairwise tree: k_region_whole sums regions of 129 .. 8 192 bins with every lane on its own node of the tree
    (pairwise_tree_lanes, round 6), shorter and longer ones the older ways -- all must carry numpy's bits."""
    rng = np.random.RandomState(11)
    lengths = list(range(120, 300)) + [511, 512, 513, 1023, 1024, 1025, 1031, 1032, 1033, 2047, 2048, 2049, 2057, 4095,
                                       4096, 4097, 4609, 4722, 8185, 8191, 8192, 8193, 8200, 9000] + \
        [int(v) for v in rng.randint(129, 8193, size=60)]
    regions = [rng.standard_normal(n) * rng.choice([1.0, 1e-3, 40.0]) for n in lengths]
    regions[3][:] = -0.0                                  # np.sum of nothing but -0.0 is +0.0 (it starts from its identity) ...
    regions[40][:] = -0.0                                 # ... also where most lanes of the fold hold nothing (160 bins: two nodes)
    whole, _ = wt.stouffer_segments(regions, np.inf, 3)
    want = np.array([np.sum(z) / np.sqrt(len(z)) for z in regions])
    assert same_bits(whole, want), [n for n, a, b in zip(lengths, whole, want) if not same_bits([a], [b])]


def test_late_repeats_in_one_launch(wt, monkeypatch, capfd):
    """Batches run repeats 3 .. as ONE launch of one workgroup (k_lat_repeats from repeat 3 on) -- as a rule nothing is
    queued by then.  Samples with scattered loud bins, where repeats 3 and 4 still have pairs queued: the outputs of that form,
    of a launch pair per repeat (WC_TEST_TAIL_REPEATS=0) and of the forced overflow (a cap of 0 pairs: the batch is
    repeated with launch pairs) are the same bits, and the later repeats really had pairs queued."""
    import re
    rng = np.random.RandomState(77)
    sizes = np.array([260, 240, 230, 210, 200, 190, 180, 170, 160, 150, 150, 140, 120, 110, 100, 90, 90, 80, 60, 60, 50, 50], dtype=np.int64)
    total = int(sizes.sum())
    offs = np.concatenate([[0], np.cumsum(sizes)])
    mask = np.ones(total, dtype=bool)
    B, S = total, 30
    corrected = 1.0 + 0.02 * rng.standard_normal((B, S))
    idx, dst = wt.getReference(np.asfortranarray(corrected), sizes, np.cumsum(sizes), 40, 1, 1)
    comps = np.linalg.qr(rng.standard_normal((B, 3)))[0].T
    mean = np.full(B, 1.0 / B) * (1 + 0.01 * rng.standard_normal(B))
    reference = wt.Reference(idx, dst, sizes, sizes, mask, mean, comps, binsize=1e6)
    samples = []
    r2 = np.random.RandomState(5)
    for s_ in range(48):
        lam = np.full(total, 2500.0) * (1 + 0.02 * r2.standard_normal(total)).clip(0.5)
        lam[r2.rand(total) < 0.1] *= 1.15                       # scattered loud bins: flags in the later repeats, too
        counts = r2.poisson(lam).astype(np.int32)
        samples.append({str(c + 1): counts[offs[c]:offs[c + 1]] for c in range(22)})
    thr = 5.0                                                   # (pairs queued per repeat here: 14 797, 1 016, 96, 0)
    monkeypatch.setenv("WC_TEST_VERBOSE", "1")
    capfd.readouterr()
    got = wt.test_batch(reference, samples, thr)
    err = capfd.readouterr().err
    queued = [int(v) for v in re.findall(r"pairs queued per repeat:((?: \d+)+)", err)[-1].split()]
    assert queued[2] > 0 and queued[3] > 0, queued          # repeats 3 and 4 had work
    assert "batch repeated with a launch pair" not in err
    monkeypatch.setenv("WC_TEST_TAIL_REPEATS", "0")
    want = wt.test_batch(reference, samples, thr)
    monkeypatch.delenv("WC_TEST_TAIL_REPEATS")
    monkeypatch.setenv("WC_TEST_TAIL_CAP", "0")
    capfd.readouterr()
    again = wt.test_batch(reference, samples, thr)
    assert "batch repeated with a launch pair" in capfd.readouterr().err
    for other in (want, again):
        for a, b in zip(got, other):
            assert same_bits(np.asarray(a["results_calls"], dtype=np.float64), np.asarray(b["results_calls"], dtype=np.float64))
            assert same_bits(np.concatenate([np.asarray(v) for v in a["results_z"]]), np.concatenate([np.asarray(v) for v in b["results_z"]]))
            assert same_bits(np.concatenate([np.asarray(v) for v in a["results_r"]]), np.concatenate([np.asarray(v) for v in b["results_r"]]))
            assert same_bits(np.asarray(a["results_cwz"], dtype=np.float64), np.asarray(b["results_cwz"], dtype=np.float64))
            assert same_bits([a["asdef"]], [b["asdef"]])
    reference.close()
