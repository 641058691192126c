"""CPU-only checks: the C ABI library loads and exports every declared symbol, the
host-side logic matches the oracle / goldens, and the product refuses to run
without a GPU instead of falling back to a CPU path."""
import os
import re

import numpy as np
import pytest

from oracle import wc_oracle as wo

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
KEYS = [str(c) for c in range(1, 23)] + ["X", "Y"]


@pytest.fixture(scope="module")
def lib():
    from wisecondor_amd import build, _lib
    build.build_library(verbose=False)
    return _lib.load()


def test_every_declared_symbol_is_exported(lib):
    from wisecondor_amd import _lib
    header = open(os.path.join(ROOT, "include", "wisecondor_hip.h")).read()
    declared = set(re.findall(r"\b(wc_[a-z0-9_]+)\s*\(", header))
    declared -= {"wc_ctx", "wc_reference"}
    assert declared, "no declarations parsed"
    for name in sorted(declared):
        assert hasattr(lib, name), "library does not export %s" % name
    assert declared == set(_lib.SIGNATURES), declared ^ set(_lib.SIGNATURES)
    assert lib.wc_version().startswith(b"wisecondor_hip")


def test_get_part_matches_reference_arithmetic(lib):
    from wisecondor_amd import wisetools as wt
    rng = np.random.RandomState(0)
    for _ in range(300):
        bins = int(rng.randint(1, 70000))
        parts = int(rng.randint(1, 130))
        p = int(rng.randint(0, parts))
        assert wt.getPart(p, parts, bins) == wo.get_part(p, parts, bins)
    from wisecondor_amd.distributed import row_range
    assert row_range(3, 8, 57633) == wo.get_part(3, 8, 57633)


def test_no_cpu_fallback_without_gpu(lib):
    import torch
    if torch.cuda.is_available():
        pytest.skip("GPU present")
    from wisecondor_amd import wisetools as wt, _lib
    data = np.ones((6, 3))
    with pytest.raises(_lib.WisecondorHipError):
        wt.getReference(data, [3, 3], [3, 6], 2)
    with pytest.raises(_lib.WisecondorHipError):
        wt.applyPCA(np.ones(4), np.ones(4), np.ones((1, 4)))


def test_product_does_not_import_oracle():
    pkg = os.path.join(ROOT, "wisecondor_amd")
    for name in os.listdir(pkg):
        if name.endswith(".py"):
            text = open(os.path.join(pkg, name)).read()
            assert "oracle" not in text.replace("# oracle", ""), "%s mentions the oracle" % name


def test_sum_order_detection():
    from wisecondor_amd import wisetools as wt, _lib
    a = np.zeros((50, 7))
    assert wt.sum_order_of(a) == _lib.SUM_PAIRWISE
    assert wt.sum_order_of(np.asfortranarray(a)) == _lib.SUM_SEQUENTIAL
    assert wt.sum_order_of(a.T.copy().T) == _lib.SUM_SEQUENTIAL       # what trainPCA returns
    assert wt.sum_order_of(np.zeros((50, 1))) == _lib.SUM_PAIRWISE
    # and numpy really does what the flag says
    rng = np.random.RandomState(1)
    t = rng.rand(300, 200)
    seq = np.zeros(300)
    for s in range(200):
        seq = seq + t[:, s]
    assert np.array_equal(np.sum(np.asfortranarray(t), 1), seq)
    assert np.array_equal(np.sum(t, 1), [wo.pairwise_sum(row) for row in t])


def test_scale_and_counts(golden):
    from wisecondor_amd import wisetools as wt
    g = golden("scale.npz")
    offs = np.concatenate([[0], np.cumsum(g["lengths"])])
    sample = {k: g["sample"][offs[i]:offs[i + 1]] for i, k in enumerate(KEYS)}
    scaled = wt.scaleSample(sample, 50000., 250000)
    assert np.array_equal(np.concatenate([scaled[k] for k in KEYS]), g["scaled"])
    assert all(scaled[k].dtype == np.int32 for k in KEYS)
    with pytest.raises(SystemExit):
        wt.scaleSample(sample, 50000., 120000)
    sizes = [len(sample[str(c)]) + (c % 3) - 1 for c in range(1, 23)]    # pad some, truncate others
    dense = wt.samples_to_counts([sample, sample], sizes)
    mask = np.ones(sum(sizes), dtype=bool)
    want = wo.to_numpy_ref_format(sample, sizes, mask)
    assert np.array_equal(dense[0] / dense[0].sum(), want)
    assert np.array_equal(dense[0], dense[1])


def test_prep_host_path(golden):
    from wisecondor_amd import wisetools as wt
    g = golden("cfg1_pipeline.npz")
    offs = np.concatenate([[0], np.cumsum(g["sample_chrom_lengths"])])
    samples = [{k: row[offs[i]:offs[i + 1]] for i, k in enumerate(KEYS)} for row in g["ref_samples"]]
    masked, bins, mask = wt.toNumpyArray(samples)
    assert np.array_equal(mask, g["prep_mask"]) and list(bins) == list(g["prep_chromosomeBins"])
    assert np.array_equal(masked, g["prep_maskedData"])
    corrected, pca = wt.trainPCA(masked)
    assert corrected.flags["F_CONTIGUOUS"] and not corrected.flags["C_CONTIGUOUS"]
    assert np.allclose(corrected, g["prep_correctedData"], rtol=1e-11, atol=0)
    assert np.allclose(pca.components_, g["prep_pca_components"], rtol=0, atol=1e-10)


def test_cli_surface():
    from wisecondor_amd import wisecondor as cli
    p = cli.buildParser()
    a = p.parse_args(["newref", "a.npz", "b.npz", "out.npz"])
    assert (a.infiles, a.outfile, a.refsize, a.binsize, a.cpus, a.parts) == (["a.npz", "b.npz"], "out.npz", 100, None, 1, 1)
    assert a.func is cli.toolNewref
    a = p.parse_args(["newrefpart", "prep.npz", "part", "3", "8", "-refsize", "50"])
    assert a.part == [3, 8] and a.refsize == 50 and a.func is cli.toolNewrefPart
    a = p.parse_args(["newrefpost", "prep.npz", "part", "8", "out.npz"])
    assert a.parts == 8 and a.func is cli.toolNewrefPost
    a = p.parse_args(["newrefprep", "a.npz", "prep.npz", "-binsize", "250000"])
    assert a.binsize == 250000 and a.prepfile == "prep.npz"
    a = p.parse_args(["test", "s.npz", "o.npz", "r.npz"])
    assert a.chromosomes == list(range(1, 23)) and a.minzscore is None and a.mineffectsize == 0
    assert (a.multitest, a.minrefbins, a.repeats) == (1000, 25, 5)
    a = p.parse_args(["test", "s.npz", "o.npz", "r.npz", "-chromosomes", "1,5,18", "-minzscore", "4.5"])
    assert a.chromosomes == [1, 5, 18] and a.minzscore == 4.5
    assert np.isclose(cli.zThreshold([2792], 1000, None), 4.820405826815299, rtol=1e-14)
    assert cli.zThreshold([2792], 1000, 3.0) == 3.0
    with pytest.raises(SystemExit):
        p.parse_args(["plot", "x", "y"]).func(None)


def test_shard_helpers():
    from wisecondor_amd.distributed import shard_samples
    got = [shard_samples(1000, r, 8) for r in range(8)]
    assert got[0][0] == 0 and got[-1][1] == 1000
    assert all(got[i][1] == got[i + 1][0] for i in range(7))
    got = [shard_samples(10, r, 4) for r in range(4)]
    assert [b - a for a, b in got] == [3, 3, 2, 2]


def test_shard_mode_choice():
    """Small jobs shard by row bands (no exchange), the 600 x 50 kb job by symmetric tiles."""
    from wisecondor_amd import distributed as d
    from wisecondor_amd import synth
    bins250 = synth.chrom_bins(250000)
    bins50 = synth.chrom_bins(50000)
    assert d.choose_shard_mode(int(sum(bins250)), 100, bins250, 8, 1024) == "rows"
    assert d.choose_shard_mode(int(sum(bins250)), 100, bins250, 2, 1024) == "rows"
    # 600 x 50 kb: the exchange only pays while a rank's share of the tile work is long (2 ranks)
    assert d.choose_shard_mode(int(sum(bins50)), 600, bins50, 2, 1024) == "tiles"
    assert d.choose_shard_mode(int(sum(bins50)), 600, bins50, 8, 1024) == "rows"
    assert d.choose_shard_mode(int(sum(bins50)), 4800, bins50, 8, 1024) == "tiles"
    assert d.choose_shard_mode(int(sum(bins50)), 600, bins50, 1, 1024) == "tiles"
    assert d.exchange_capacity(1024, 8) % 32 == 0 and d.exchange_capacity(1024, 1) == 1024
