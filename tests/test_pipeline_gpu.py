"""distributed.TestPipeline: several batches of the batched `test` in flight on one GPU (a context, a stream and a
host thread per slot).  Every batch must come out exactly as a lone TestBatch computes it -- and as the reference does: the
batches are made of the four golden cfg3 samples (tests/golden/cfg3_250kb.npz), whose z-scores, calls and
stdDevAvg test_cfg3_gpu.py pins on the reference's own output."""
import numpy as np
import pytest

pytestmark = pytest.mark.gpu
KEYS = [str(c) for c in range(1, 23)] + ["X", "Y"]
NAMES = ["mild18", "strong5", "loss2", "normal"]


@pytest.fixture(scope="module")
def setup(golden):
    import torch
    from wisecondor_amd import wisetools as wt
    g = golden("cfg3_250kb.npz")
    corrected = np.asfortranarray(g["prep_correctedData"])
    bins = g["prep_maskedChromBins"]
    idx, dst = wt.getReference(corrected, bins, np.cumsum(bins), 100, 1, 1)
    ref = wt.Reference(idx, dst, g["ref_chromosome_sizes"], g["ref_masked_sizes"], g["ref_mask"],
                       g["ref_pca_mean"], g["ref_pca_components"], binsize=250000)
    lengths = g["sample_chrom_lengths"]
    offs = np.concatenate([[0], np.cumsum(lengths)])
    samples = [{k: g["t_%s_sample" % n][offs[i]:offs[i + 1]] for i, k in enumerate(KEYS)} for n in NAMES]
    counts = wt.samples_to_counts(samples, [int(v) for v in ref.chromosome_sizes])
    thr = float(g["t_mild18_threshold_z"])
    yield torch, ref, counts, thr, g
    ref.close()


def _batches(torch, counts):
    """Five batches of different order and two different sizes (33 and 40 samples: the wave-per-bin kernels)."""
    out = []
    for b, ns in enumerate((33, 33, 40, 33, 40)):
        rows = [(i + b) % 4 for i in range(ns)]
        out.append((rows, torch.from_numpy(np.ascontiguousarray(counts[rows])).cuda()))
    return out


def _snapshot(tb):
    return {k: getattr(tb, k).cpu().numpy().copy() for k in ("results_z", "results_r", "cwz", "calls", "n_calls", "asdef")}


@pytest.mark.parametrize("depth", [2, 4])
def test_pipeline_equals_lone_batches(setup, depth):
    torch, ref, counts, thr, g = setup
    from wisecondor_amd import distributed
    batches = _batches(torch, counts)
    lone = []
    for rows, c in batches:
        tb = distributed.TestBatch(ref, c, thr, max_calls=64)
        tb.run()
        torch.cuda.synchronize()
        lone.append(_snapshot(tb))
    pipe = distributed.TestPipeline(ref, thr, depth=depth, max_calls=64)
    try:
        for trip in range(2):                      # the second trip reuses every slot's buffers
            got = {}
            pipe.run([c for _, c in batches], consume=lambda b, tb: got.__setitem__(b, _snapshot(tb)))
            assert sorted(got) == list(range(len(batches)))
            for b, want in enumerate(lone):
                for key, arr in want.items():
                    if key == "calls":               # rows beyond a sample's n_calls are not written (a reused slot
                        continue                     # keeps what its previous batch left there)
                    assert arr.tobytes() == got[b][key].tobytes(), (trip, b, key)
                for i, n in enumerate(want["n_calls"]):
                    assert want["calls"][i, :n].tobytes() == got[b]["calls"][i, :n].tobytes(), (trip, b, i)
    finally:
        pipe.close()
    # and the content is the reference's: stdDevAvg and the call coordinates of every row of the first batch
    rows = batches[0][0]
    for i, s in enumerate(rows):
        name = NAMES[s]
        assert np.isclose(lone[0]["asdef"][i], float(g["t_%s_asdef" % name]), rtol=1e-11), name
        want = g["t_%s_results_calls" % name]
        n = int(lone[0]["n_calls"][i])
        assert n == want.shape[0], name
        assert np.array_equal(np.sort(lone[0]["calls"][i, :n, :3], axis=0), np.sort(want[:, :3], axis=0)), name


def test_pipeline_raises_in_the_caller(setup):
    torch, ref, counts, thr, g = setup
    from wisecondor_amd import distributed, _lib
    pipe = distributed.TestPipeline(ref, thr, depth=2, max_calls=1)        # one call row per sample: too few
    try:
        rows = [1] * 33                                                      # the strong sample has several calls
        c = torch.from_numpy(np.ascontiguousarray(counts[rows])).cuda()
        with pytest.raises(_lib.WisecondorHipError):
            pipe.run([c, c, c])
    finally:
        pipe.close()
