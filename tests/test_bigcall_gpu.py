"""The north-star's one-call shapes: 1 000 DISTINCT samples in ONE wc_test_batch call, at 250 kb and at 50 kb bins.

The reference tests one sample per invocation (wisecondor.py:193-238), so the batch size is this build's dimension
only -- and it selects code: k_zscore_tiled from 113 samples on, left-over sample tiles dealt to the XCDs, the
padding of the sample count to a multiple of 16, k_sd_fast's template width, the walker's workgroup order, the
grow-only scratch of the context.  Every output of the big call must be bit for bit what the same samples give in
eight 125-sample calls; the samples the real reference / the CPU oracle ran (tests/golden/cfg5_whole.npz, the
oracle's test_sample at 250 kb) are checked INSIDE the big call.

Also here: batches of growing size on one fresh context (a grow-only buffer that is re-reserved AFTER the prepare
kernel has written it loses its contents: ADVICE round 5), and a region whose loud cells overflow the cell search's
queues in ONE workgroup (the search must give the job up, not drop cells).
"""
import hashlib
import os
import sys

import numpy as np
import pytest

from oracle import wc_oracle as wo

pytestmark = pytest.mark.gpu
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "tools"))


def same_bits(a, b):
    a = np.asarray(a, dtype=np.float64)
    b = np.asarray(b, dtype=np.float64)
    if a.shape != b.shape:
        return False
    nan = np.isnan(a) & np.isnan(b)
    return bool(np.all(nan | (a.view(np.int64) == b.view(np.int64))))


def assert_same_outputs(a, b, tag):
    """Two result dicts of wisetools.test_batch: every member bit for bit."""
    for c, (x, y) in enumerate(zip(a["results_z"], b["results_z"])):
        assert same_bits(x, y), (tag, "results_z", c)
    for c, (x, y) in enumerate(zip(a["results_r"], b["results_r"])):
        assert same_bits(x, y), (tag, "results_r", c)
    assert same_bits(a["results_cwz"], b["results_cwz"]), (tag, "cwz")
    ca = np.asarray(a["results_calls"], dtype=np.float64).reshape(-1, 5)
    cb = np.asarray(b["results_calls"], dtype=np.float64).reshape(-1, 5)
    assert ca.shape == cb.shape and same_bits(ca, cb), (tag, "calls", ca, cb)
    assert same_bits([a["asdef"]], [b["asdef"]]), (tag, "asdef", a["asdef"], b["asdef"])


@pytest.fixture(scope="module")
def wt():
    from wisecondor_amd import wisetools
    return wisetools


@pytest.fixture(scope="module")
def case50(wt):
    import cfg5_case
    from wisecondor_amd import synth
    c = cfg5_case.build(wt, synth, n_test=1000)
    yield c
    c["reference"].close()


@pytest.fixture(scope="module")
def case250(wt):
    """A 250 kb reference (40 samples through the GPU prep + newref) and 1 000 distinct test samples, 5 % of them with
    a 1-5 % gain / loss over a quarter of a chromosome."""
    from wisecondor_amd import synth
    from wisecondor_amd.wisecondor import zThreshold
    binsize = 250000
    profile = synth.bin_profile(binsize)
    refs = [synth.make_sample(profile, seed=i) for i in range(40)]
    _, chrom_bins, mask, corrected, comps, mean, masked_bins = wt.prepReference(refs)
    masked_bins = np.asarray(masked_bins, dtype=np.int64)
    idx, dst = wt.getReference(corrected, masked_bins, np.cumsum(masked_bins), 100, 1, 1)
    reference = wt.Reference(idx, dst, np.asarray(chrom_bins, dtype=np.int64), masked_bins, mask, mean, comps,
                             binsize=binsize, device=0)
    rng = np.random.RandomState(4242)
    tests = []
    for i in range(1000):
        events = []
        if rng.rand() < 0.05:
            c = int(rng.randint(1, 23))
            n = len(profile[c - 1])
            a = int(rng.randint(0, max(1, n - n // 4)))
            events.append((str(c), a, a + n // 4, 1.0 + rng.choice([-1, 1]) * rng.uniform(0.01, 0.05)))
        tests.append(synth.make_sample(profile, seed=1000 + i, events=events))
    thr = float(zThreshold([int(v) for v in masked_bins], 1000, None))
    npz = dict(binsize=np.float64(binsize), indexes=idx, distances=dst, chromosome_sizes=np.asarray(chrom_bins),
               mask=mask, masked_sizes=masked_bins, pca_mean=mean, pca_components=comps)
    yield dict(reference=reference, threshold=thr, tests=tests, npz=npz, binsize=binsize)
    reference.close()


def _big_against_eight(wt, case):
    thr = case["threshold"]
    tests = case["tests"]
    assert len(tests) == 1000
    big = wt.test_batch(case["reference"], tests, thr)                 # ONE wc_test_batch call
    assert len(big) == 1000
    n_calls = 0
    for lo in range(0, 1000, 125):
        small = wt.test_batch(case["reference"], tests[lo:lo + 125], thr)
        for i, (a, b) in enumerate(zip(big[lo:lo + 125], small)):
            assert_same_outputs(a, b, lo + i)
            n_calls += len(np.asarray(b["results_calls"]).reshape(-1, 5))
    return big, n_calls


def test_1000_samples_in_one_call_at_250kb(wt, case250):
    big, n_calls = _big_against_eight(wt, case250)
    assert n_calls >= 30                                                # the planted events are called
    # two samples of the big call against the CPU oracle's test_sample (one with a planted event if there is one
    # among the first hundred): coordinates exact, values to the PCA's tolerance
    with_calls = [i for i in range(100) if len(np.asarray(big[i]["results_calls"]).reshape(-1, 5)) > 0]
    for i in sorted(set([999] + with_calls[:1])):
        with np.errstate(all="ignore"):
            want = wo.test_sample(case250["tests"][i], float(case250["binsize"]), case250["npz"])
        calls = np.asarray(big[i]["results_calls"], dtype=np.float64).reshape(-1, 5)
        wcalls = np.asarray(want["results_calls"], dtype=np.float64).reshape(-1, 5)
        assert np.array_equal(calls[:, :3], wcalls[:, :3]), (i, calls, wcalls)
        assert np.allclose(calls[:, 3:], wcalls[:, 3:], rtol=1e-6, atol=1e-12), i
        assert np.allclose(big[i]["results_cwz"], want["results_cwz"], rtol=1e-6, atol=1e-9), i
        assert np.isclose(big[i]["asdef"], float(want["asdef"]), rtol=1e-6, atol=0), i
        for c in range(22):
            wz = np.asarray(want["results_z"][c], dtype=np.float64)
            assert np.array_equal(big[i]["results_z"][c] == 0, wz == 0), (i, c)
            assert np.allclose(big[i]["results_z"][c], wz, rtol=1e-6, atol=1e-6), (i, c)


def test_1000_samples_in_one_call_at_50kb(wt, case50, golden):
    big, n_calls = _big_against_eight(wt, case50)
    assert n_calls >= 200
    # the samples the real reference's toolTest and the oracle ran (cfg5_whole.npz), inside the big call
    gw = golden("cfg5_whole.npz")
    ref = case50["reference"]
    assert case50["threshold"] == float(gw["threshold"])
    same_ref = hashlib.sha256(ref.distances.tobytes()).hexdigest() == str(gw["distances_sha256"])
    rtol = 1e-9 if same_ref else 1e-6
    wanted = sorted(set(int(i) for i in gw["ref_samples"]) | set(int(i) for i in gw["oracle_samples"]))
    checked = 0
    for i in wanted:
        out = big[i]
        calls = np.asarray(out["results_calls"], dtype=np.float64).reshape(-1, 5)
        for kind in ("ref", "oracle"):
            key = "%s%d_results_calls" % (kind, i)
            if key not in gw:
                continue
            want = gw[key].reshape(-1, 5)
            assert np.array_equal(calls[:, :3], want[:, :3]), (kind, i, calls[:, :3], want[:, :3])
            assert np.allclose(calls[:, 3:], want[:, 3:], rtol=rtol, atol=1e-12), (kind, i)
            assert np.allclose(out["results_cwz"], gw["%s%d_results_cwz" % (kind, i)], rtol=rtol, atol=1e-9), (kind, i)
            assert np.isclose(out["asdef"], float(gw["%s%d_asdef" % (kind, i)]), rtol=rtol, atol=0), (kind, i)
            checked += len(want)
    assert checked >= 50


@pytest.mark.parametrize("first,second", [(128, 144), (120, 130), (3, 7), (256, 288)])
def test_growing_batches_on_a_fresh_context(wt, case250, first, second):
    """A batch, then a LARGER one in the same context: the prepare kernel of the second call writes xt / xc before the
    repeats reserve their arrays, and a DevBuf that grows frees its memory -- with 113-128 samples followed by
    129-144 (Np 128 -> 144) the first reservation was a no-op and the second reallocated xt after it had been
    written.  Compared with the same samples in a context that has only ever seen the second size."""
    from wisecondor_amd import _lib
    thr = case250["threshold"]
    tests = case250["tests"]
    ctx_a, ctx_b = _lib.new_context(0), _lib.new_context(0)
    ref_a = ref_b = None
    try:
        ref_a = case250["reference"].clone(ctx_a)
        ref_b = case250["reference"].clone(ctx_b)
        wt.test_batch(ref_a, tests[:first], thr)
        got = wt.test_batch(ref_a, tests[200:200 + second], thr)
        want = wt.test_batch(ref_b, tests[200:200 + second], thr)
        for i, (a, b) in enumerate(zip(got, want)):
            assert_same_outputs(a, b, i)
        one = wt.test_batch(ref_b, tests[200:201], thr)                 # and against a one-sample call
        assert_same_outputs(got[0], one[0], "single")
    finally:
        for r in (ref_a, ref_b):
            if r is not None:
                r.close()
        _lib.destroy_context(ctx_a)
        _lib.destroy_context(ctx_b)


def _loud_sawtooth(n_bins=8100, up=200, cfrac=0.9):
    """A region built to fill the cell search's queues: a triangular wave (200 bins up, 200 down, amplitudes growing
    by 0.1 % per period so that the best window is unique) under an alternating component.  The wave makes every
    32 x 32 band cell on a flank a near-tie of the best window, the alternating part widens every block's range: at
    the FINAL cut 516 band cells still reach it (emulated on the host when the case was designed), against 320 queue
    slots of one workgroup."""
    z = np.zeros(n_bins)
    period = 2 * up
    for i in range(n_bins // period):
        a = 1.0 + 1e-3 * i
        z[i * period:i * period + up] = a
        z[i * period + up:(i + 1) * period] = -a
    c = cfrac * np.sqrt(up)
    return z + c * np.where(np.arange(n_bins) % 2 == 0, 1.0, -1.0)


def test_a_region_that_overflows_the_cell_queues(wt, monkeypatch):
    """cell_search<0> in ONE workgroup per job (what k_seg_walk always runs, and k_seg_job from 2 048 jobs on; forced
    here with WC_CELL_PARTS=1) on a region with more loud cells than its queues hold.  Dropping cells silently can
    miss the extreme; the search has to give the job up (exact scan).  Compared with the default split of the job
    over several workgroups and with the row-block kernels, and every segment's value with numpy's."""
    thr = 5.0
    z = _loud_sawtooth()
    extra = np.random.RandomState(3).standard_normal(700)              # a quiet second region beside it
    monkeypatch.setenv("WC_CELL_PARTS", "1")
    whole1, segs1 = wt.stouffer_segments([z, extra], thr, 3)
    monkeypatch.delenv("WC_CELL_PARTS")
    whole2, segs2 = wt.stouffer_segments([z, extra], thr, 3)
    monkeypatch.setenv("WC_TEST_CELLS", "0")
    whole3, segs3 = wt.stouffer_segments([z, extra], thr, 3)
    monkeypatch.delenv("WC_TEST_CELLS")
    assert len(segs1[0]) >= 20
    for other in (segs2, segs3):
        assert [xy for _, xy in segs1[0]] == [xy for _, xy in other[0]]
        assert same_bits([v for v, _ in segs1[0]], [v for v, _ in other[0]])
        assert segs1[1] == other[1]
    assert same_bits(whole1, whole2) and same_bits(whole1, whole3)
    for v, (x, y) in segs1[0]:
        assert v == np.sum(z[x:y + 1]) / np.sqrt(y - x + 1), (x, y)
    # the first decision of the recursion is the region's champion: no window beats it (prefix-sum values, 1e-9)
    P = np.concatenate([[0.0], np.cumsum(z)])
    best = 0.0
    for ln in range(1, len(z) + 1):
        v = (P[ln:] - P[:-ln]) / np.sqrt(ln)
        best = max(best, float(np.abs(v).max()))
    assert abs(max(abs(v) for v, _ in segs1[0]) - best) < 1e-9


@pytest.mark.parametrize("which", ["250kb", "50kb"])
def test_big_call_repeats_bit_for_bit(wt, case250, case50, monkeypatch, which):
    """The 1 000-sample call, several times, against ONE run of a second implementation of the segmentation (the
    host-driven levels, WC_TEST_WALK=0) and of the z-score outputs' layout (WC_ZSCORE_SM=0: bin-major + transposes):
    every member of every sample's result bit for bit, every time (round 5's defect only showed from run to run)."""
    case = case250 if which == "250kb" else case50
    thr = case["threshold"]
    monkeypatch.setenv("WC_TEST_WALK", "0")
    monkeypatch.setenv("WC_ZSCORE_SM", "0")
    want = wt.test_batch(case["reference"], case["tests"], thr)
    monkeypatch.delenv("WC_TEST_WALK")
    monkeypatch.delenv("WC_ZSCORE_SM")
    for _ in range(4):
        got = wt.test_batch(case["reference"], case["tests"], thr)
        for i, (a, b) in enumerate(zip(want, got)):
            assert_same_outputs(a, b, i)
