"""`test` for a few samples per call (BASELINE config 3): the latency mode -- fused small kernels, the
one-block segmentation tree with its riders, one hipGraph replay -- must give exactly what the
general path gives, including when it gives up (status words) and the call is repeated there."""
import os

import numpy as np
import pytest

pytestmark = pytest.mark.gpu


@pytest.fixture(scope="module")
def wt():
    from wisecondor_amd import wisetools
    return wisetools


def _genome(wt, seed, sizes, n_ref=24, k=40):
    rng = np.random.RandomState(seed)
    sizes = np.asarray(sizes, dtype=np.int64)
    total = int(sizes.sum())
    mask = rng.rand(total) > 0.03
    offs = np.concatenate([[0], np.cumsum(sizes)])
    msizes = np.array([int(mask[offs[i]:offs[i + 1]].sum()) for i in range(22)], dtype=np.int64)
    B = int(msizes.sum())
    corrected = 1.0 + 0.02 * rng.standard_normal((B, n_ref))
    idx, dst = wt.getReference(np.asfortranarray(corrected), msizes, np.cumsum(msizes), k, 1, 1)
    comps = np.linalg.qr(rng.standard_normal((B, 3)))[0].T
    mean = np.full(B, 1.0 / B) * (1 + 0.01 * rng.standard_normal(B))
    reference = wt.Reference(idx, dst, sizes, msizes, mask, mean, comps, binsize=1e6)
    return reference, rng, sizes, offs, total


def _sample(rng, sizes, offs, total, events):
    lam = np.full(total, 2500.0) * (1 + 0.02 * rng.standard_normal(total)).clip(0.5)
    for c, a, b, f in events:
        lam[offs[c] + a:offs[c] + b] *= f
    counts = rng.poisson(lam).astype(np.int32)
    return {str(c + 1): counts[offs[c]:offs[c + 1]] for c in range(22)}


def _run(wt, reference, samples, mode, **kw):
    old = os.environ.get("WC_TEST_LATENCY_MODE")
    try:
        if mode is None:
            os.environ.pop("WC_TEST_LATENCY_MODE", None)
        else:
            os.environ["WC_TEST_LATENCY_MODE"] = mode
        return [wt.test_batch(reference, samples, 4.5, **kw) for _ in range(3)][-1]     # size, capture, replay
    finally:
        if old is None:
            os.environ.pop("WC_TEST_LATENCY_MODE", None)
        else:
            os.environ["WC_TEST_LATENCY_MODE"] = old


def _same(a, b):
    assert len(a) == len(b)
    for x, y in zip(a, b):
        assert np.array_equal(np.asarray(x["results_calls"]), np.asarray(y["results_calls"]), equal_nan=True)
        assert np.array_equal(np.concatenate(x["results_z"]), np.concatenate(y["results_z"]), equal_nan=True)
        assert np.array_equal(np.concatenate(x["results_r"]), np.concatenate(y["results_r"]), equal_nan=True)
        assert np.array_equal(x["results_cwz"], y["results_cwz"], equal_nan=True)
        assert x["asdef"] == y["asdef"] or (x["asdef"] != x["asdef"] and y["asdef"] != y["asdef"])


@pytest.mark.parametrize("ns", [1, 3, 8])
def test_latency_mode_equals_general_path(wt, ns):
    """Regions from 2 bins to the longest the tree kernel takes (2 048 after masking), events of several
    sizes in the long one (a recursion a few levels deep), a zeroed stretch, one to eight samples."""
    sizes = [2100, 2, 40, 700, 64, 65, 129, 5] + [30] * 14
    reference, rng, sizes, offs, total = _genome(wt, 5 + ns, sizes)
    try:
        assert 1900 < max(int(v) for v in reference.masked_sizes) <= 2048
        samples = []
        for i in range(ns):
            events = [(0, 100 + 37 * i, 400 + 37 * i, 1.25), (0, 1200, 1230, 0.6), (0, 1700, 1702, 1.8),
                      (3, 50 * (i % 5), 50 * (i % 5) + 90, 0.8), (6, 10, 20, 1.5)]
            s = _sample(rng, sizes, offs, total, events)
            if i % 2:
                s["5"] = s["5"].copy()
                s["5"][:6] = 0
            samples.append(s)
        general = _run(wt, reference, samples, "0", minrefbins=10, repeats=4)
        assert sum(len(o["results_calls"]) for o in general) >= 4 * ns
        _same(_run(wt, reference, samples, "2", minrefbins=10, repeats=4), general)      # latency kernels, eager
        _same(_run(wt, reference, samples, None, minrefbins=10, repeats=4), general)     # ... as a graph replay
    finally:
        reference.close()


def test_latency_mode_gives_up_cleanly(wt):
    """More segments in one region than the tree kernel holds (128): the status word sends the call to
    the general path, and the caller sees the general path's result."""
    sizes = [1500] + [25] * 21
    reference, rng, sizes, offs, total = _genome(wt, 99, sizes)
    try:
        events = [(0, a, a + 3, 1.5 if (a // 8) % 2 else 0.5) for a in range(8, 1480, 8)]    # ~180 short events, alternating sign
        samples = [_sample(rng, sizes, offs, total, events)]
        general = _run(wt, reference, samples, "0", minrefbins=5, repeats=2)
        assert len(general[0]["results_calls"]) > 128
        _same(_run(wt, reference, samples, None, minrefbins=5, repeats=2), general)
        _same(_run(wt, reference, samples, "2", minrefbins=5, repeats=2), general)
    finally:
        reference.close()


def _run_tail(wt, reference, samples, tail, **kw):
    old = os.environ.get("WC_TEST_TREE_TAIL")
    try:
        if tail is None:
            os.environ.pop("WC_TEST_TREE_TAIL", None)
        else:
            os.environ["WC_TEST_TREE_TAIL"] = tail
        return wt.test_batch(reference, samples, 4.5, **kw)
    finally:
        if old is None:
            os.environ.pop("WC_TEST_TREE_TAIL", None)
        else:
            os.environ["WC_TEST_TREE_TAIL"] = old


def test_batch_tree_tail_equals_host_rounds(wt):
    """Batches (more than eight samples): after the first segmentation round the hot regions are finished
    by the tree kernel; the host-driven rounds (WC_TEST_TREE_TAIL=0) must give the same calls -- also
    when one sample makes the tree kernel give up (more than 128 segments in a region) and the round is
    started again on the host-driven path."""
    sizes = [1500, 2, 40, 700, 64, 65, 129, 5] + [30] * 14
    reference, rng, sizes, offs, total = _genome(wt, 321, sizes)
    try:
        samples = []
        for i in range(12):
            events = [(0, 100 + 31 * i, 300 + 31 * i, 1.25), (0, 900, 930, 0.6), (3, 40 * (i % 6), 40 * (i % 6) + 80, 0.8),
                      (6, 10, 20, 1.5)]
            samples.append(_sample(rng, sizes, offs, total, events))
        host = _run_tail(wt, reference, samples, "0", minrefbins=10, repeats=3)
        assert sum(len(o["results_calls"]) for o in host) >= 36
        _same(_run_tail(wt, reference, samples, None, minrefbins=10, repeats=3), host)
        busy = [(0, a, a + 3, 1.5 if (a // 8) % 2 else 0.5) for a in range(8, 1480, 8)]
        samples[5] = _sample(rng, sizes, offs, total, busy)
        host = _run_tail(wt, reference, samples, "0", minrefbins=5, repeats=2)
        assert len(host[5]["results_calls"]) > 128
        _same(_run_tail(wt, reference, samples, None, minrefbins=5, repeats=2), host)
    finally:
        reference.close()


@pytest.mark.parametrize("kw", [dict(chromosomes=[1, 5, 22], minrefbins=10, repeats=1),
                                dict(chromosomes=[4], minrefbins=10, repeats=5),
                                dict(minrefbins=39, repeats=2),          # nearly every bin is cleaned away
                                dict(minrefbins=41, repeats=2)])         # every bin is: empty regions
def test_latency_mode_options(wt, kw):
    """Chromosome subsets, a single repeat, regions emptied by the minrefbins filter: same answers from the
    latency kernels, their graph replay and the general path."""
    sizes = [300, 2, 40, 200, 64, 65, 129, 5] + [30] * 14
    reference, rng, sizes, offs, total = _genome(wt, 777, sizes)
    try:
        samples = [_sample(rng, sizes, offs, total, [(0, 40, 90, 1.3), (3, 20, 60, 0.7), (21, 3, 12, 1.6)])
                   for _ in range(3)]
        general = _run(wt, reference, samples, "0", **kw)
        _same(_run(wt, reference, samples, "2", **kw), general)
        _same(_run(wt, reference, samples, None, **kw), general)
    finally:
        reference.close()
