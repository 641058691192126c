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


def test_shard_mode_override(monkeypatch):
    """WC_NEWREF_SHARD pins the multi-GPU shard mode; otherwise it is measured per job
    (NewrefJob.calibrate, exercised in tests/test_distributed_gloo.py)."""
    from wisecondor_amd import distributed as d
    monkeypatch.delenv("WC_NEWREF_SHARD", raising=False)
    assert d.forced_shard_mode() is None
    for mode in ("tiles", "rows"):
        monkeypatch.setenv("WC_NEWREF_SHARD", mode)
        assert d.forced_shard_mode() == mode
    monkeypatch.setenv("WC_NEWREF_SHARD", "auto")
    assert d.forced_shard_mode() is None
    assert d.exchange_capacity(1024, 8) % 32 == 0 and d.exchange_capacity(1024, 1) == 1024


def _run(cmd, env=None):
    import subprocess
    import sys
    full = dict(os.environ)
    full.pop("WORLD_SIZE", None)
    full.pop("RANK", None)
    if env:
        full.update(env)
    p = subprocess.run([sys.executable] + cmd, cwd=ROOT, env=full, capture_output=True, text=True, timeout=300)
    return p.returncode, p.stdout, p.stderr


def test_bench_starts_its_own_ranks():
    """`bench.py --gpus 2` without a torchrun environment starts two ranks itself (before anything
    touches the GPU) and reports the world size it really ran with; a launcher that started a
    different number of ranks than --gpus says is an error, not a silent one-rank run."""
    import json
    rc, out, err = _run(["bench.py", "--gpus", "2", "--launch-check"])
    assert rc == 0, err
    line = json.loads(out.strip().splitlines()[-1])
    assert line["n_gpus"] == 2 and line["ranks_requested"] == 2
    rc, out, err = _run(["bench.py", "--gpus", "2", "--launch-check"], env={"WORLD_SIZE": "3", "RANK": "0"})
    assert rc == 2 and "started 3 rank" in err
    rc, out, err = _run(["bench.py", "--launch-check"])
    assert rc == 0 and json.loads(out.strip().splitlines()[-1])["n_gpus"] == 1


def test_cpu_baseline_worker(tmp_path):
    """The worker processes of bench.py's cpu_baseline leg: a bounded row window of a part, equal to
    the oracle's own rows (they time the oracle, nothing else)."""
    import json
    from wisecondor_amd import synth
    data, bins, sums = synth.corrected_matrix(1000000, 12, seed=3)
    data = np.asfortranarray(data)
    idx, dst = wo.get_reference(data, bins, sums, 15, 1, 1, fast=True)
    np.save(str(tmp_path / "corrected.npy"), data)
    np.savez(str(tmp_path / "reference.npz"), bins=bins, k=15, binsize=1e6)
    nowait = {"WC_CPU_BASELINE_NO_WAIT": "1"}
    rc, out, err = _run(["oracle/cpu_baseline.py", "newref", str(tmp_path), "0", "1", "40"], env=nowait)
    assert rc == 0, err
    line = json.loads(out.strip().splitlines()[-1])
    assert line["rows"] == [0, 40] and line["seconds"] > 0
    assert np.array_equal(np.load(str(tmp_path / "newref_0.npy")), idx[:40])
    rc, out, err = _run(["oracle/cpu_baseline.py", "newref", str(tmp_path), "3", "4", "25"], env=nowait)
    lo, hi = wo.get_part(3, 4, data.shape[0])
    assert rc == 0 and json.loads(out.strip().splitlines()[-1])["rows"] == [lo, min(hi, lo + 25)]


def test_newref_file_names_and_rank_count(monkeypatch):
    from wisecondor_amd import wisecondor as cli
    import argparse
    names = cli.BuildFiles("out/dir/ref.npz")
    assert names.prep == "out/dir/ref_prep.npz" and names.part_base == "out/dir/ref_part"
    assert cli.BuildFiles("ref").prep == "ref_prep.npz"
    assert cli.BuildFiles.part_name("x_part", 7) == "x_part_7.npz"
    monkeypatch.setattr(cli, "_visible_gpus", lambda: 8)
    assert cli._rank_count(argparse.Namespace(cpus=1, gpus=None)) == 1
    assert cli._rank_count(argparse.Namespace(cpus=4, gpus=None)) == 4
    assert cli._rank_count(argparse.Namespace(cpus=64, gpus=None)) == 8       # no more ranks than GPUs
    assert cli._rank_count(argparse.Namespace(cpus=1, gpus=2)) == 2
    monkeypatch.setattr(cli, "_visible_gpus", lambda: 0)
    assert cli._rank_count(argparse.Namespace(cpus=4, gpus=None)) == 1


def test_prep_eigen_route_selection(monkeypatch):
    """Where prepReference solves trainPCA's eigenproblem (wisetools.py:89-101): csrc/eigh.hip from
    EIG_ON_GPU_FROM samples on (every size it takes), LAPACK on the fetched Gram matrix otherwise; WC_PREP_EIG forces either."""
    from wisecondor_amd import wisetools as wt
    monkeypatch.delenv("WC_PREP_EIG", raising=False)
    assert wt.EIG_ON_GPU_FROM == 3           # round 5: every size the solver takes stays on the GPU
    assert wt._eig_on_gpu(100, 3) and wt._eig_on_gpu(wt.EIG_ON_GPU_FROM, 3) and wt._eig_on_gpu(600, 3)
    assert not wt._eig_on_gpu(2, 1)
    assert not wt._eig_on_gpu(5000, 3) and not wt._eig_on_gpu(600, 9)      # beyond the solver: the host route
    monkeypatch.setenv("WC_PREP_EIG", "host")
    assert not wt._eig_on_gpu(600, 3)
    monkeypatch.setenv("WC_PREP_EIG", "gpu")
    assert wt._eig_on_gpu(3, 1) and wt._eig_on_gpu(100, 3)
    with pytest.raises(ValueError):
        wt._eig_on_gpu(2, 1)
    monkeypatch.setenv("WC_PREP_EIG", "maybe")
    with pytest.raises(ValueError):
        wt._eig_on_gpu(100, 3)


def test_part_files_go_to_the_rank_that_owns_their_rows():
    """`newref -gpus N`: part m is written by the rank whose row range (getPart for N parts, wisetools.py:358-361)
    holds the part's rows, so that no result all-gather is needed; a part count that is no multiple of the rank
    count leaves parts that straddle two ranges, and then every rank gathers all rows."""
    from wisecondor_amd import wisecondor as cli
    from wisecondor_amd import wisetools as wt
    from wisecondor_amd.distributed import row_range
    for parts, world, n_bins in [(8, 8, 57633), (16, 8, 57633), (2, 2, 2897), (4, 2, 11087), (24, 8, 55337), (6, 3, 1001)]:
        owners = cli.part_owners(parts, list(range(1, parts + 1)), n_bins, world)
        assert owners is not None, (parts, world, n_bins)
        covered = 0
        for m, r in owners.items():
            lo, hi = wt.getPart(m - 1, parts, n_bins)
            b, e = row_range(r, world, n_bins)
            assert b <= lo and hi <= e
            covered += hi - lo
        assert covered == n_bins
    assert cli.part_owners(3, [1, 2, 3], 11087, 2) is None            # part 2 straddles the two ranks' ranges
    assert cli.part_owners(3, [1, 3], 11087, 2) == {1: 0, 3: 1}       # ... but a resumed run that lacks 1 and 3 only does not


def test_bench_compact_line_from_a_full_record():
    """bench.py's LAST line must stay small (the driver's parser lost round 5's 21 KB line): the compact form of that
    very record (profiles/r05_bench_cfg2_unprofiled.json is the full round-5 line) is under 6 000 bytes, strict JSON,
    and carries the contract's keys with `roofline` / `cpu_baseline` as flat objects."""
    import json
    import bench
    full = json.load(open(os.path.join(ROOT, "profiles", "r05_bench_cfg2_unprofiled.json")))
    assert len(json.dumps(full)) > 15000
    line = json.dumps(bench.compact_line(full), allow_nan=False, separators=(",", ":"))
    assert len(line) < bench.LINE_LIMIT
    d = json.loads(line)
    for key in ("metric", "value", "unit", "n_gpus", "steps", "warmup", "ms_per_step", "higher_is_better", "scaling",
                "vs_baseline", "dtype", "data", "config", "roofline", "cpu_baseline"):
        assert key in d, key
    assert d["value"] == float("%.6g" % full["value"]) and d["ms_per_step"] == float("%.6g" % full["ms_per_step"])
    assert set(d["config"]) == {"workload", "parallelism", "world_size", "shard_mode"}
    assert d["roofline"]["kernel"] == "k_rescore" and 0 < d["roofline"]["frac"] <= 1
    assert all(not isinstance(v, (dict, list)) or k in ("other",) for k, v in d["roofline"].items())
    assert d["cpu_baseline"]["cores"] == 1 and d["cpu_baseline"]["kind"] == "port" and d["cpu_baseline"]["value"] > 0
    # NaN / infinity never reach the line
    full["value"] = float("nan")
    assert json.loads(json.dumps(bench.compact_line(full), allow_nan=False))["value"] is None
