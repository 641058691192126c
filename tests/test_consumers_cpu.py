"""the reference's own consumers (`report`, `plot`) must read
the test .npz this build writes.  Needs the upstream sources (dev container only), so it
is skipped on the GPU box; the numbers come from the CPU oracle, the FILE from the
product's writer."""
import argparse
import contextlib
import io
import os
import sys

import numpy as np
import pytest

from oracle import wc_oracle as wo

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "tools"))
import ref_loader  # noqa: E402

pytestmark = pytest.mark.skipif(not ref_loader.available(), reason="upstream reference not present")
KEYS = [str(c) for c in range(1, 23)] + ["X", "Y"]


def test_reference_report_and_plot_read_our_output(tmp_path, golden):
    from wisecondor_amd import wisecondor as cli
    g = golden("cfg1_pipeline.npz")
    offs = np.concatenate([[0], np.cumsum(g["sample_chrom_lengths"])])
    sample = {k: g["t_loss2_sample"][offs[i]:offs[i + 1]] for i, k in enumerate(KEYS)}
    ref = dict(binsize=g["ref_binsize"], indexes=g["ref_indexes"], distances=g["ref_distances"],
               chromosome_sizes=g["ref_chromosome_sizes"], mask=g["ref_mask"], masked_sizes=g["ref_masked_sizes"],
               pca_mean=g["ref_pca_mean"], pca_components=g["ref_pca_components"])
    out = wo.test_sample(sample, float(g["binsize"]), ref)
    args = cli.buildParser().parse_args(["test", "in.npz", "out.npz", "ref.npz"])
    outfile = str(tmp_path / "sample_out.npz")
    cli.writeTestOutput(outfile, args, float(g["binsize"]), out, out["threshold_z"])
    back = np.load(outfile, allow_pickle=True)
    assert back["results_calls"].shape == (4, 5)
    assert back["results_z"].dtype == object and back["arguments"].item()["repeats"] == 5

    # an empty call list must look like the reference's np.array([])
    empty = dict(out)
    empty["results_calls"] = np.zeros((0, 5))
    cli.writeTestOutput(str(tmp_path / "none.npz"), args, float(g["binsize"]), empty, out["threshold_z"])
    assert np.load(str(tmp_path / "none.npz"), allow_pickle=True)["results_calls"].shape == (0,)

    # the same result through the native batch writer (csrc/npzio.cpp): the reference's report must
    # read that file as well
    from wisecondor_amd import ingest
    sizes = [int(v) for v in g["ref_chromosome_sizes"]]
    native = str(tmp_path / "sample_native.npz")
    calls = np.asarray(out["results_calls"], dtype=np.float64).reshape(1, -1, 5)
    ingest.write_results([native], [args], {"version": "t"}, float(g["binsize"]), float(out["threshold_z"]), sizes,
                         np.concatenate(out["results_z"])[None, :].copy(), np.concatenate(out["results_r"])[None, :].copy(),
                         np.asarray(out["results_cwz"], dtype=np.float64)[None, :].copy(), np.ascontiguousarray(calls),
                         np.array([calls.shape[1]], dtype=np.int32), np.array([out["asdef"]], dtype=np.float64))

    wt, wc, _ = ref_loader.load()
    convert_out = str(tmp_path / "sample.npz")
    quality = dict(mapped=1, unmapped=0, no_coordinate=0, filter_rmdup=0, filter_mapq=0, pre_retro=1,
                   post_retro=1, pair_fail=0)
    np.savez_compressed(convert_out, arguments={"binsize": float(g["binsize"])}, runtime={}, sample=sample,
                        quality=quality)
    buf = io.StringIO()
    with contextlib.redirect_stdout(buf):
        wc.toolReport(argparse.Namespace(testfile=convert_out, resultfile=outfile, mineffect=1.5))
    text = buf.getvalue()
    buf2 = io.StringIO()
    with contextlib.redirect_stdout(buf2):
        wc.toolReport(argparse.Namespace(testfile=convert_out, resultfile=native, mineffect=1.5))
    assert buf2.getvalue() == text
    assert "# Test results: #" in text
    assert "-157.33\t-49.85" in text            # the chr2 loss: z-score and effect in per cent
    assert "2:100000000-140000000" in text

    # `plot` reads results_z / results_calls / threshold_z / binsize and starts drawing; under
    # this container's matplotlib the reference then trips over a removed tick attribute
    # (Tick.label), which has nothing to do with the file
    try:
        with contextlib.redirect_stdout(io.StringIO()):
            wc.toolPlot(argparse.Namespace(infile=outfile, outfile=str(tmp_path / "plot"), cytofile=None,
                                           chromosomes=list(range(1, 23)), columns=2, filetype="png",
                                           size=[11.7, 8.3], mineffect=1.5))
        assert os.path.getsize(str(tmp_path / "plot_z.png")) > 10000
    except AttributeError as e:
        assert "label" in str(e)
