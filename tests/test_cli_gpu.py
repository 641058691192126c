"""End-to-end run of the drop-in CLI (newref with parts, test) on the GPU, checked
against the oracle and the reference's .npz schema."""
import os

import numpy as np
import pytest

from oracle import wc_oracle as wo

pytestmark = pytest.mark.gpu
KEYS = [str(c) for c in range(1, 23)] + ["X", "Y"]


def _write_sample(path, flat, lengths, binsize):
    offs = np.concatenate([[0], np.cumsum(lengths)])
    sample = {k: np.asarray(flat[offs[i]:offs[i + 1]], dtype=np.int32) for i, k in enumerate(KEYS)}
    np.savez_compressed(path, arguments={"binsize": float(binsize)}, runtime={}, sample=sample, quality={})
    return sample


def test_newref_and_test_cli(tmp_path, golden):
    from wisecondor_amd import wisecondor as cli
    g = golden("cfg1_pipeline.npz")
    lengths = g["sample_chrom_lengths"]
    binsize = float(g["binsize"])
    infiles = []
    for i, row in enumerate(g["ref_samples"]):
        p = str(tmp_path / ("ref_%02d.npz" % i))
        _write_sample(p, row, lengths, binsize)
        infiles.append(p)
    refpath = str(tmp_path / "reference.npz")
    # run the prep step first and keep a copy: `newref` must pick the existing prep file up
    # (resume by file existence, wisecondor.py:43) and delete it at the end
    import shutil
    cli.main(["newrefprep"] + infiles + [str(tmp_path / "reference_prep.npz")])
    shutil.copy(str(tmp_path / "reference_prep.npz"), str(tmp_path / "kept_prep.npz"))
    cli.main(["newref"] + infiles + [refpath, "-parts", "3", "-refsize", "100"])
    assert not os.path.exists(str(tmp_path / "reference_prep.npz"))       # temp files removed like the reference does
    assert not os.path.exists(str(tmp_path / "reference_part_1.npz"))
    ref = np.load(refpath, allow_pickle=True)
    assert set(ref.files) == {"arguments", "runtime", "binsize", "indexes", "distances", "chromosome_sizes",
                              "mask", "masked_sizes", "pca_components", "pca_mean"}
    assert ref["indexes"].dtype == np.int32 and ref["distances"].dtype == np.float64
    assert ref["indexes"].shape == g["ref_indexes"].shape
    assert np.array_equal(ref["mask"], g["ref_mask"])
    assert np.array_equal(ref["masked_sizes"], g["ref_masked_sizes"])
    assert set(ref["runtime"].item()) == {"version", "datetime", "hostname", "username"}
    assert ref["arguments"].item()["refsize"] == 100

    # the prep seam: this run's correctedData -> the GPU selection must equal the oracle's
    pz = np.load(str(tmp_path / "kept_prep.npz"), allow_pickle=True)
    assert set(pz.files) == {"arguments", "runtime", "binsize", "chromosomeBins", "maskedData", "mask",
                             "maskedChromBins", "maskedChromBinSums", "correctedData", "pca_components", "pca_mean"}
    corrected = pz["correctedData"]
    assert corrected.flags["F_CONTIGUOUS"]                 # same layout as the reference's prep file
    assert np.allclose(corrected, g["prep_correctedData"], rtol=1e-10, atol=0)
    mbins = np.asarray(ref["masked_sizes"])
    want_i, want_d = wo.get_reference(corrected, mbins, np.cumsum(mbins), 100, 1, 1, fast=True)
    assert np.array_equal(ref["indexes"], want_i)
    assert np.array_equal(ref["distances"], want_d)
    # and it is the golden reference up to the PCA solver's rounding
    assert np.mean(ref["indexes"] == g["ref_indexes"]) > 0.99

    # test sub-command on two samples
    for name in ("gain5_gap", "loss2"):
        sp = str(tmp_path / ("test_%s.npz" % name))
        sample = _write_sample(sp, g["t_%s_sample" % name], lengths, binsize)
        op = str(tmp_path / ("out_%s.npz" % name))
        with pytest.raises(SystemExit) as e:
            cli.main(["test", sp, op, refpath])
        assert e.value.code == 0
        out = np.load(op, allow_pickle=True)
        assert set(out.files) == {"arguments", "runtime", "binsize", "results_r", "results_z", "results_cwz",
                                  "results_calls", "threshold_z", "asdef", "aasdef"}
        assert out["results_z"].dtype == object and len(out["results_z"]) == 22
        assert [len(a) for a in out["results_z"]] == [int(v) for v in ref["chromosome_sizes"]]
        want = wo.test_sample(sample, binsize, {k: ref[k] for k in ref.files})
        wc_ = np.asarray(want["results_calls"]).reshape(-1, 5)
        gc_ = np.asarray(out["results_calls"]).reshape(-1, 5)
        assert np.array_equal(gc_[:, :3], wc_[:, :3])
        assert np.allclose(gc_[:, 3:], wc_[:, 3:], rtol=1e-9)
        assert np.allclose(np.concatenate(list(out["results_z"])), np.concatenate(want["results_z"]), rtol=1e-9, atol=1e-11)
        assert np.isclose(float(out["threshold_z"]), want["threshold_z"], rtol=1e-14)
        assert np.isclose(float(out["asdef"]), want["asdef"], rtol=1e-11)
        # the golden (reference-produced) calls, coordinates included, come out of this reference too
        assert np.array_equal(gc_[:, :3], g["t_%s_results_calls" % name][:, :3])


def test_newrefpart_cluster_style(tmp_path, golden):
    """newrefprep / newrefpart m n / newrefpost as separate commands (README.md:135-142)."""
    from wisecondor_amd import wisecondor as cli
    g = golden("cfg1_pipeline.npz")
    lengths = g["sample_chrom_lengths"]
    infiles = []
    for i, row in enumerate(g["ref_samples"][:8]):
        p = str(tmp_path / ("ref_%02d.npz" % i))
        _write_sample(p, row, lengths, 1e6)
        infiles.append(p)
    prep = str(tmp_path / "r_prep.npz")
    cli.main(["newrefprep"] + infiles + [prep, "-binsize", "2000000"])
    for m in (1, 2):
        cli.main(["newrefpart", prep, str(tmp_path / "r_part"), str(m), "2", "-refsize", "60"])
    cli.main(["newrefpost", prep, str(tmp_path / "r_part"), "2", str(tmp_path / "r.npz")])
    ref = np.load(str(tmp_path / "r.npz"), allow_pickle=True)
    pz = np.load(prep, allow_pickle=True)
    assert float(ref["binsize"]) == 2000000
    assert pz["correctedData"].flags["F_CONTIGUOUS"]
    want_i, want_d = wo.get_reference(pz["correctedData"], pz["maskedChromBins"], pz["maskedChromBinSums"], 60, 1, 1,
                                      fast=True)
    assert np.array_equal(ref["indexes"], want_i) and np.array_equal(ref["distances"], want_d)
    part1 = np.load(str(tmp_path / "r_part_1.npz"), allow_pickle=True)
    assert set(part1.files) == {"arguments", "runtime", "indexes", "distances"}


def test_testbatch_equals_single_tests(tmp_path, golden):
    """The build-only `testbatch` sub-command writes what `test` writes, sample by sample."""
    from wisecondor_amd import wisecondor as cli
    g = golden("cfg1_pipeline.npz")
    lengths = g["sample_chrom_lengths"]
    refpath = str(tmp_path / "reference.npz")
    np.savez_compressed(refpath, arguments={}, runtime={}, binsize=float(g["ref_binsize"]),
                        indexes=g["ref_indexes"], distances=g["ref_distances"],
                        chromosome_sizes=g["ref_chromosome_sizes"], mask=g["ref_mask"],
                        masked_sizes=g["ref_masked_sizes"], pca_components=g["ref_pca_components"],
                        pca_mean=g["ref_pca_mean"])
    names = ["mild18", "gain5_gap", "loss2", "normal"]
    paths = []
    for n in names:
        p = str(tmp_path / ("s_%s.npz" % n))
        _write_sample(p, g["t_%s_sample" % n], lengths, float(g["binsize"]))
        paths.append(p)
    cli.main(["testbatch"] + paths + [str(tmp_path / "out"), refpath, "-batch", "3"])
    for n, p in zip(names, paths):
        single = str(tmp_path / ("single_%s.npz" % n))
        with pytest.raises(SystemExit):
            cli.main(["test", p, single, refpath])
        a = np.load(single, allow_pickle=True)
        b = np.load(str(tmp_path / "out" / ("s_%s_test.npz" % n)), allow_pickle=True)
        assert np.array_equal(a["results_calls"], b["results_calls"])
        assert np.array_equal(np.concatenate(list(a["results_z"])), np.concatenate(list(b["results_z"])))
        assert np.array_equal(a["results_cwz"], b["results_cwz"])
        assert float(a["asdef"]) == float(b["asdef"])
        want = g["t_%s_results_calls" % n]
        assert np.array_equal(np.asarray(b["results_calls"]).reshape(-1, 5)[:, :3], want[:, :3])


def test_newref_on_two_ranks_equals_one(tmp_path, golden, monkeypatch):
    """`newref -cpus 2 -gpus 2`: one process per rank (here two ranks sharing the box's GPU, gloo for
    the collectives; RCCL needs one GPU per rank), same part files, same reference as one rank."""
    from wisecondor_amd import wisecondor as cli
    g = golden("cfg1_pipeline.npz")
    lengths = g["sample_chrom_lengths"]
    infiles = []
    for i, row in enumerate(g["ref_samples"]):
        p = str(tmp_path / ("ref_%02d.npz" % i))
        _write_sample(p, row, lengths, float(g["binsize"]))
        infiles.append(p)
    one = str(tmp_path / "one.npz")
    cli.main(["newref"] + infiles + [one, "-refsize", "100", "-parts", "3"])
    monkeypatch.setenv("WC_RANKS_BACKEND", "gloo")
    monkeypatch.chdir(tmp_path)          # the rank processes must find the package from any working directory
    two = str(tmp_path / "two.npz")
    cli.main(["newref"] + infiles + [two, "-refsize", "100", "-cpus", "3", "-gpus", "2"])
    assert not os.path.exists(str(tmp_path / "two_prep.npz")) and not os.path.exists(str(tmp_path / "two_part_2.npz"))
    a = np.load(one, allow_pickle=True)
    b = np.load(two, allow_pickle=True)
    assert set(a.files) == set(b.files)
    for key in ("indexes", "distances", "mask", "masked_sizes", "pca_mean", "pca_components", "chromosome_sizes"):
        assert np.array_equal(a[key], b[key]), key
    assert b["arguments"].item()["parts"] == 3

    # testbatch on two ranks: every sample's file equals the one-rank file
    names = ["mild18", "gain5_gap", "loss2", "normal", "gain5_past"]
    paths = []
    for n in names:
        p = str(tmp_path / ("s_%s.npz" % n))
        _write_sample(p, g["t_%s_sample" % n], lengths, float(g["binsize"]))
        paths.append(p)
    cli.main(["testbatch"] + paths + [str(tmp_path / "o1"), one, "-batch", "2"])
    cli.main(["testbatch"] + paths + [str(tmp_path / "o2"), one, "-batch", "2", "-gpus", "2"])
    for n in names:
        x = np.load(str(tmp_path / "o1" / ("s_%s_test.npz" % n)), allow_pickle=True)
        y = np.load(str(tmp_path / "o2" / ("s_%s_test.npz" % n)), allow_pickle=True)
        assert np.array_equal(x["results_calls"], y["results_calls"])
        assert np.array_equal(np.concatenate(list(x["results_z"])), np.concatenate(list(y["results_z"])))
        assert float(x["asdef"]) == float(y["asdef"])
        want = g["t_%s_results_calls" % n]
        assert np.array_equal(np.asarray(y["results_calls"]).reshape(-1, 5)[:, :3], want[:, :3])


def _strict_json(text):
    """json.loads that refuses the NaN / Infinity tokens Python's encoder would happily write."""
    import json

    def refuse(token):
        raise ValueError("non-standard JSON token %s" % token)
    return json.loads(text, parse_constant=refuse)


def _run_bench(tmp_path, argv, timeout):
    """bench.py as the driver runs it: the LAST stdout line is the record the driver parses (compact: it lost a
    21 KB line in round 5), the full record goes to the --detail file.  Returns (line, detail)."""
    import json
    import subprocess
    import sys
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    env = {k: v for k, v in os.environ.items() if k not in ("WORLD_SIZE", "RANK", "LOCAL_RANK")}
    detail_path = os.path.join(str(tmp_path), "bench_detail.json")
    p = subprocess.run([sys.executable, "bench.py"] + argv + ["--detail", detail_path], cwd=root, env=env,
                       capture_output=True, text=True, timeout=timeout)
    assert p.returncode == 0, p.stderr[-2000:]
    last = p.stdout.rstrip("\n").splitlines()[-1]
    assert last.startswith("{") and len(last) < 8000, (len(last), last[:200])
    line = _strict_json(last)
    for key in ("metric", "value", "unit", "n_gpus", "steps", "warmup", "ms_per_step", "higher_is_better", "scaling",
                "vs_baseline", "dtype", "data", "config", "roofline", "cpu_baseline"):
        assert key in line, key
    assert set(line["config"]) >= {"workload", "parallelism", "world_size", "shard_mode"}
    assert set(line["roofline"]) >= {"kernel", "kernel_ms", "bound", "achieved", "peak", "unit", "frac", "traffic"}
    assert line["detail"] and "not written" not in line["detail"]
    detail = json.load(open(detail_path))
    assert detail["value"] == pytest.approx(line["value"], rel=1e-4) and detail["ms_per_step"] == pytest.approx(line["ms_per_step"], rel=1e-4)
    return line, detail


def test_bench_two_ranks_on_one_gpu(tmp_path):
    """bench.py --gpus 2 starts two ranks itself; with the gloo backend they share the one GPU.
    The JSON line must carry the world size it really ran with and the measured shard mode; every rank's
    stage times and collectives are in the detail record, their sums on the line."""
    line, detail = _run_bench(tmp_path, ["--gpus", "2", "--backend", "gloo", "--no-extra", "--workload", "cfg1", "--steps",
                                         "2", "--warmup", "1", "--test-samples", "32", "--no-cpu-baseline"], 900)
    assert line["n_gpus"] == 2 and line["config"]["world_size"] == 2
    assert line["config"]["shard_mode"] in ("tiles", "rows")
    assert len(line["multi_rank"]["stage_ms_sum_per_rank"]) == 2 and len(line["multi_rank"]["collective_ms_sum_per_rank"]) == 2
    assert all(v > 0 for v in line["multi_rank"]["stage_ms_sum_per_rank"])
    assert len(line["multi_rank"]["collective_bytes_rank0"]) >= 1
    mr = detail["multi_rank"]
    assert len(mr["per_rank"]) == 2 and mr["row_bands_per_rank"] >= 1
    for entry in mr["per_rank"]:
        assert entry["stages_ms"]["collected"] > 0 and len(entry["collectives"]) >= 1
        assert all(c["bytes"] > 0 and c["ms"] >= 0 for c in entry["collectives"])
    assert set(detail["config"]["shard_calibration_s"]) == {"tiles", "rows"}
    assert line["value"] > 0 and line["test"]["value"] > 0


def test_bench_eight_ranks_on_one_gpu(tmp_path):
    """bench.py --gpus 8 (the driver's scaling run) with the ranks sharing the one GPU over gloo: the launcher
    starts eight ranks, the shard mode is measured on them, and the line says so."""
    line, detail = _run_bench(tmp_path, ["--gpus", "8", "--backend", "gloo", "--no-extra", "--workload", "cfg1", "--steps",
                                         "2", "--warmup", "1", "--test-samples", "32", "--no-cpu-baseline"], 1500)
    assert line["n_gpus"] == 8 and line["config"]["world_size"] == 8
    assert line["config"]["shard_mode"] in ("tiles", "rows")
    assert len(line["multi_rank"]["stage_ms_sum_per_rank"]) == 8
    assert set(detail["config"]["shard_calibration_s"]) == {"tiles", "rows"}
    assert line["value"] > 0 and line["test"]["value"] > 0


def test_bench_line_is_complete_on_one_gpu(tmp_path):
    """The default bench command in a short form (cfg2, the 600 x 50 kb extra, no CPU leg): ONE compact last
    line (< 8 000 bytes, strict JSON) with the contract's keys, and a detail record with every object the
    measurement contract names, every fraction at most 1, no swallowed error."""
    short, line = _run_bench(tmp_path, ["--steps", "4", "--warmup", "1", "--test-samples", "32", "--no-cpu-baseline"], 900)
    assert short["n_gpus"] == 1 and short["steps"] == 4 and short["warmup"] == 1
    assert short["metric"] == "newref bin-pair distances/sec" and short["value"] > 1e10
    assert short["roofline"]["kernel"] in ("k_rescore", "k_gram_glds") and 0 < short["roofline"]["frac"] <= 1.0
    assert short["roofline"]["kernel_ms"] > 0 and short["roofline"]["bound"] in ("hbm", "mfma", "l2")
    assert short["test"]["value"] > 0 and short["test"]["ms_per_batch"] > 0 and short["test"]["latency_ms_per_call"] > 0
    assert short["test"]["whole_job_1000_samples"]["ms_per_call"] > 0
    assert short["test"]["whole_job_1000_samples"]["distinct_samples"] is True
    assert short["extra"]["ms_per_step"] > 0 and short["extra"]["test_50kb"]["ms_per_batch"] > 0
    assert short["extra"]["test_50kb"]["whole_job_1000_samples"]["ms_per_call"] > 0
    assert short["extra"]["emulated_world_8_projection"]["tiles"]["results_equal_single_rank"] is True
    assert short["cpu_baseline"] is None                        # --no-cpu-baseline
    assert line["n_gpus"] == 1 and line["steps"] == 4 and line["warmup"] == 1
    assert line["metric"] == "newref bin-pair distances/sec" and line["value"] > 1e10
    for roof in (line["roofline"], line["roofline_other"]):
        assert roof["bound"] in ("hbm", "mfma", "l2") and 0 < roof["frac"] <= 1.0, roof
        assert roof["achieved"] > 0 and roof["peak"] > 0 and roof["kernel_ms"] > 0
    extra = line["extra"]
    assert "error" not in extra, extra
    assert extra["ms_per_step"] > 0 and 0 < extra["k_gram_frac_of_mfma_peak"] <= 1.0
    assert line["roofline"]["kernel"].startswith("k_gram") or line["roofline"]["kernel"].startswith("float64 re-score")
    assert line["test"]["latency"]["launch_floor_us"] > 0 and line["test"]["latency"]["ms_per_call"] > 0
    assert 0 < extra["test_50kb"]["roofline"]["fp64_valu_frac"] <= 1.0
    assert extra["test_50kb"]["roofline"]["zscore_gather"]["l2_frac"] <= 1.0
    assert 0 < extra["rescore_roofline"]["frac"] <= 1.3             # at the HBM roof on uncorrelated rows (15 % of the gathers hit L2)
    assert "error" not in extra["test_50kb"], extra["test_50kb"]
    assert extra["test_50kb"]["value"] > 1000 and extra["test_50kb"]["calls_found"] > 0
    assert line["prep"]["ms"] > 0
    test = line["test"]
    assert test["value"] > 0 and 0 < test["roofline"]["frac"] <= 1.0
    assert test["single_sample_latency_ms"] < test["ms_per_batch"]
    for leg in (test, extra["test_50kb"]):          # one batch in flight on top, several in flight beside it
        one, two = leg["one_batch_in_flight"], leg["pipelined"]
        assert one["ms_per_batch"] > 0 and two["ms_per_batch"] > 0 and two["batches_timed"] >= 8
        assert leg["ms_per_batch"] == one["ms_per_batch"] and leg["value"] == one["value"]
        # (EVERY window x 4 operations over the batch time against the float64 peak: above 1 since the 50 kb batch takes
        #  less than a millisecond -- the search is faster than evaluating all windows at the peak would be)
        assert 0 < leg["roofline"]["algorithmic_fp64_frac"] < 4.0
    assert test["whole_job_1000_samples"]["samples_per_s"] > test["value"] * 0.8       # the big call amortises the fixed costs
    assert set(line["stages_ms"]) >= {"prepared", "thresholds", "collected", "picked", "rescored", "finished"}
    assert line["multi_rank"] is None
    assert "error" not in extra.get("ingest", {}), extra.get("ingest")
    # round 4: the node's worth of ranks one after the other (projections, labelled), the whole 50 kb cohort in one call
    em = extra["emulated_world_8"]
    assert "error" not in em, em
    for mode in ("tiles", "rows"):
        assert em[mode]["results_equal_single_rank"] is True and len(em[mode]["per_rank_ms"]) == 8
        assert em[mode]["projected_step_ms"] > em[mode]["max_rank_ms"] > 0
        assert em[mode]["projected_step_ms"] >= em[mode]["projected_step_ms_overlapped"] > 0.5 * em[mode]["max_rank_ms"]
    assert "PROJECT" in em["what"].upper()
    t50 = extra["test_50kb"]
    assert len(t50["emulated_world_8"]["per_rank_ms"]) == 8 and t50["whole_job_1000_samples"]["samples"] == 1000
    assert t50["whole_job_1000_samples"]["samples_per_s"] > 0.8 * t50["value"]
    assert line["rescore_stage"]["k_rescore_ms"] > 0 and line["roofline_other"] is not None
    assert extra["ingest"]["files_per_s"] > 100
