#!/usr/bin/env python3
"""Benchmark of the WISECONDOR hot path on MI355X.

    python bench.py --gpus N --steps K --warmup W

One step = one pass of the `newref` reference-bin selection over the whole workload
(BASELINE.json config 2 by default: 100 samples x 250 kb bins) with the corrected matrix
already resident in HBM, followed (timed separately) by batched `test` passes (PCA-apply ->
5 masked z-score repeats -> Stouffer segmentation) over --test-samples samples per GPU at the
same bin size.

With --gpus N > 1 and no torchrun environment the script starts N ranks of itself (fresh
processes, before anything touches the GPU); under torchrun it checks that WORLD_SIZE == N.

Rank 0 prints ONE JSON line: `value` is the newref metric of BASELINE.json (ordered
cross-chromosome bin-pair distances per second, whole job), `test` carries samples/s,
`roofline` describes the kernel that takes longest per step (timed live with events on the
launch stream), `stages_ms` every stage of the step, and `cpu_baseline` the CPU oracle timed
on this box (rank 0, N=1 only; one core and all cores).  Inputs are synthetic (seeded), see
wisecondor_amd/synth.py.
"""
import argparse
import json
import os
import re
import subprocess
import sys
import tempfile
import time

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)

WORKLOADS = {
    # name: (binsize, reference samples)
    "cfg1": (1000000, 16),
    "cfg2": (250000, 100),
    "cfg4": (50000, 600),
}
# MI355X_MICROARCH.md
PEAK_FP32_MFMA = 157.3e12   # v_mfma_f32_32x32x2_f32 dense
PEAK_BF16_MFMA = 2500e12    # v_mfma_f32_32x32x16_bf16 / _f16 dense (no sparsity)
PEAK_HBM = 8.0e12           # bytes/s, spec (6.3e12 achievable by a copy)
PEAK_L2 = 34.5e12           # bytes/s aggregate over the 8 XCDs
PEAK_FP64_VALU_OPS = 39.3e12  # float64 vector add / mul / max per second (78.6 TFLOP/s counts an FMA as two)
PEAK_MALL = 10.0e12         # bytes/s the Infinity Cache sustains towards the L2s (order of magnitude, MI355X_MICROARCH.md)
ROUND = "r06"
ROUNDS = ("r06", "r05", "r04", "r03", "r02", "r01")   # committed profile files: the newest that exists
PIPE_DEPTH = 4        # batches of the batched test in flight (distributed.TestPipeline)
LATENCY_LAUNCHES = 8        # kernels of one latency-mode `test` call


# --------------------------------------------------------------- rank launch ----
def spawn_ranks(args, argv):
    """Start args.gpus copies of this script, one per GPU, and wait.  Runs before torch is
    imported here: nothing in this process ever touches the GPU."""
    import socket
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    port = s.getsockname()[1]
    s.close()
    procs = []
    for rank in range(args.gpus):
        env = dict(os.environ)
        env.update({"RANK": str(rank), "LOCAL_RANK": str(rank), "WORLD_SIZE": str(args.gpus),
                    "MASTER_ADDR": "127.0.0.1", "MASTER_PORT": str(port),
                    "HSA_ENABLE_IPC_MODE_LEGACY": os.environ.get("HSA_ENABLE_IPC_MODE_LEGACY", "0")})
        procs.append(subprocess.Popen([sys.executable, os.path.abspath(__file__)] + argv, env=env))
    codes = []
    while True:
        codes = [p.poll() for p in procs]
        if all(c is not None for c in codes):
            break
        if any(c not in (None, 0) for c in codes):      # a dead rank leaves the others in a collective
            time.sleep(2.0)
            for p in procs:
                if p.poll() is None:
                    p.kill()
            codes = [p.wait() for p in procs]
            break
        time.sleep(0.05)
    return 1 if any(codes) else 0


def launch_check(args):
    """--launch-check: rendezvous + one all-reduce, no GPU work (tests of the rank launch on CPU)."""
    import torch
    import torch.distributed as dist
    world = int(os.environ.get("WORLD_SIZE", "1"))
    rank = int(os.environ.get("RANK", "0"))
    if world > 1:
        dist.init_process_group("gloo")
        t = torch.ones(1)
        dist.all_reduce(t)
        assert int(t.item()) == world
    if rank == 0:
        print(json.dumps({"launch_check": True, "n_gpus": world, "ranks_requested": args.gpus}))
    if world > 1:
        dist.barrier()
        dist.destroy_process_group()


# ------------------------------------------------------------------- inputs ----
def build_inputs(binsize, n_ref, n_test, seed0=0, device=0):
    """Pipeline-level synthetic inputs: reference samples -> prep (GPU, untimed) and test samples."""
    import numpy as np
    from wisecondor_amd import synth
    from wisecondor_amd import wisetools as wt
    profile = synth.bin_profile(binsize)
    samples = [synth.make_sample(profile, seed=seed0 + i) for i in range(n_ref)]
    _, chrom_bins, mask, corrected, comps, mean, masked_bins = wt.prepReference(samples, device=device)
    # the same prep again, device resident and timed (newrefprep's numerics; host counts in,
    # correctedData left in HBM for newref): what a caller pays between `convert`ed samples and stage A
    import time as _time
    import torch as _torch
    dense = wt.samples_to_counts(samples, chrom_bins)           # what an ingest loop holds (wisecondor_amd/ingest.py)
    wt.prepReference(None, device=device, device_out=True, counts=dense, chrom_bins=chrom_bins)
    _torch.cuda.synchronize()
    _t0 = _time.perf_counter()
    wt.prepReference(None, device=device, device_out=True, counts=dense, chrom_bins=chrom_bins)
    _torch.cuda.synchronize()
    prep_ms = 1e3 * (_time.perf_counter() - _t0)
    masked_bins = np.asarray(masked_bins, dtype=np.int64)
    tests = make_tests(profile, n_test)
    return dict(corrected=corrected, chrom_bins=np.asarray(chrom_bins, dtype=np.int64), mask=mask,
                masked_bins=masked_bins, pca_mean=mean, pca_components=comps, tests=tests, prep_ms=prep_ms)


def make_tests(profile, n_test):
    """n_test converted test samples (seeds 1000 ...); SURVEY.md 8(d): 5 % of them carry a 1-5 % gain / loss.
    The first m samples of make_tests(profile, n) are make_tests(profile, m)."""
    import numpy as np
    from wisecondor_amd import synth
    rng = np.random.RandomState(4242)
    tests = []
    for i in range(n_test):
        events = []
        if rng.rand() < 0.05:
            c = int(rng.randint(1, 23))
            n = len(profile[c - 1])
            a = int(rng.randint(0, max(1, n - n // 4)))
            f = 1.0 + rng.choice([-1, 1]) * rng.uniform(0.01, 0.05)
            events.append((str(c), a, a + n // 4, f))
        tests.append(synth.make_sample(profile, seed=1000 + i, events=events))
    return tests


def committed_traffic(workload):
    """HBM-side bytes per launch from the committed rocprofv3 PMC passes (profiles/<round>_traffic.json),
    or {} when no such file is there."""
    for rnd in ROUNDS:
        path = os.path.join(ROOT, "profiles", "%s_traffic.json" % rnd)
        if os.path.exists(path):
            t = json.load(open(path)).get(workload)
            if t:
                t = dict(t)
                t["source"] = "profiles/%s_traffic.json" % rnd
                t["measured_in_this_run"] = False
                return t
    return {}


def committed_latency_trace():
    """Kernel durations of one latency-mode replay from the committed rocprofv3 kernel trace
    (profiles/<round>_latency_trace.json, written by tools/refresh_profiles.sh), or None."""
    for rnd in ROUNDS:
        path = os.path.join(ROOT, "profiles", "%s_latency_trace.json" % rnd)
        if os.path.exists(path):
            t = json.load(open(path))
            t["source"] = "profiles/%s_latency_trace.json" % rnd
            t["measured_in_this_run"] = False
            return t
    return None


def committed_gather_roof(n_rows, mode=0):
    """What a kernel that does nothing but gather whole rows of an [n_rows, 1 KB] matrix reaches on this chip
    (tools/micro/gather_rate.*, 8 B per lane, every XCD gathering from the whole matrix; committed by
    tools/refresh_profiles.sh as profiles/<round>_gather_roof.txt): the figure of the nearest matrix size, or None."""
    import numpy as np
    path, rnd_found = None, None
    for rnd in ROUNDS:
        cand = os.path.join(ROOT, "profiles", "%s_gather_roof.txt" % rnd)
        if os.path.exists(cand):
            path, rnd_found = cand, rnd
            break
    if path is None:
        return None
    best = None
    for line in open(path):
        m = re.match(r"bins\s+(\d+) mode %d row 1024 B: [\d.]+ ms, ([\d.]+) TB/s" % mode, line)
        if m:
            rows, rate = int(m.group(1)), float(m.group(2))
            if best is None or abs(np.log(rows / float(n_rows))) < abs(np.log(best[0] / float(n_rows))):
                best = (rows, rate)
    if best is None:
        return None
    return {"matrix_rows": best[0], "TBps": best[1], "source": "profiles/%s_gather_roof.txt" % rnd_found,
            "measured_in_this_run": False,
            "what": ("random whole-row gathers of a matrix of this many 1 KB rows by a kernel that does nothing else "
                     "(the matrix misses the 4 MB L2 of an XCD; the guide's 34.5 TB/s is for L2-resident data)") if mode == 0 else
                    ("the same gathers with the 128-byte sample columns dealt to the XCDs (workgroup w only touches column "
                     "w % 8 of every row, a wave = 4 bins x 16 samples: k_zscore_tiled's access pattern, nothing but the loads)")}


def hbm_bytes(entry):
    """FETCH_SIZE is doubled (the gfx950 rule for wide streaming reads, MI355X_MICROARCH.md)."""
    if not entry or entry.get("fetch_kb_per_launch") is None:
        return None
    return 1024.0 * (2.0 * entry["fetch_kb_per_launch"] + entry.get("write_kb_per_launch", 0.0))


def committed_busy_file():
    for rnd in ROUNDS:
        path = os.path.join(ROOT, "profiles", "%s_pmc_busy.json" % rnd)
        if os.path.exists(path):
            return rnd, json.load(open(path))
    return None, None


def committed_l2(workload, kernel_prefix):
    """L2 hit rate and memory-side read requests of a kernel from the committed TCC counter pass, or None."""
    rnd, runs = committed_busy_file()
    for run, kernels in sorted((runs or {}).items()):
        if run.endswith(workload) and "T_" in run:
            for name, row in kernels.items():
                if name.startswith(kernel_prefix) and "l2_hit_rate" in row:
                    rd = row.get("counters", {}).get("TCC_EA0_RDREQ_sum")
                    return {"l2_hit_rate": row["l2_hit_rate"], "tcc_ea0_rdreq": rd,
                            "rdreq_bytes_at_128B": None if rd is None else rd * 128.0,
                            "source": "profiles/%s_pmc_busy.json (%s, %s)" % (rnd, run, name),
                            "measured_in_this_run": False}
    return None


def committed_busy(workload):
    """VALU / LDS busy fractions of the re-score kernels from the committed counter passes, or None."""
    rnd, runs = committed_busy_file()
    if runs is None:
        return None
    for run, kernels in sorted(runs.items()):
        if run.endswith(workload):
            out = {}
            for name, row in kernels.items():
                if name.startswith("k_rescore") or name.startswith("k_pick"):
                    out[name] = {n: round(row[n], 3) for n in ("valu_busy", "lds_busy", "waves_per_simd", "eff_clock_ghz")
                                 if n in row}
            if out:
                out["source"] = "profiles/%s_pmc_busy.json (%s)" % (rnd, run)
                out["measured_in_this_run"] = False
                return out
    return None


# ------------------------------------------------------------- cpu baseline ----
def cpu_model():
    try:
        for line in open("/proc/cpuinfo"):
            if line.startswith("model name"):
                return line.split(":", 1)[1].strip()
    except Exception:
        pass
    return "unknown"


def usable_cores():
    """Host cores this process may really use: the affinity mask, cut down to the cgroup CPU quota
    when the box runs under one (a container sees every core of the host in its mask)."""
    n = len(os.sched_getaffinity(0)) if hasattr(os, "sched_getaffinity") else (os.cpu_count() or 1)
    try:
        quota, period = open("/sys/fs/cgroup/cpu.max").read().split()[:2]
        if quota != "max":
            n = min(n, max(1, int(float(quota) / float(period))))
    except Exception:
        try:
            quota = int(open("/sys/fs/cgroup/cpu/cpu.cfs_quota_us").read())
            period = int(open("/sys/fs/cgroup/cpu/cpu.cfs_period_us").read())
            if quota > 0:
                n = min(n, max(1, quota // period))
        except Exception:
            pass
    return n


class CpuWorkers(object):
    """Child processes of oracle/cpu_baseline.py over one scratch directory: the workers load their
    inputs, report ready and wait for `go`, so that interpreter start-up and file reads of one worker
    do not eat into the timed work of another."""

    def __init__(self):
        self.tmp = tempfile.mkdtemp(prefix="wc_cpu_")
        self.script = os.path.join(ROOT, "oracle", "cpu_baseline.py")
        self.env = dict(os.environ, OMP_NUM_THREADS="1", OPENBLAS_NUM_THREADS="1", MKL_NUM_THREADS="1")

    def run(self, n_proc, what, rows=0):
        tmp = self.tmp
        for name in os.listdir(tmp):
            if name.startswith("ready_") or name == "go":
                os.remove(os.path.join(tmp, name))
        procs = [subprocess.Popen([sys.executable, self.script, what, tmp, str(p), str(n_proc), str(rows)],
                                  stdout=subprocess.PIPE, env=self.env) for p in range(n_proc)]
        while sum(name.startswith("ready_") for name in os.listdir(tmp)) < n_proc:
            if any(p.poll() not in (None, 0) for p in procs):
                raise RuntimeError("a cpu_baseline worker failed")
            time.sleep(0.02)
        t0 = time.perf_counter()
        open(os.path.join(tmp, "go"), "w").close()
        outs = [json.loads(p.communicate()[0].decode().strip().splitlines()[-1]) for p in procs]
        return time.perf_counter() - t0, outs

    def close(self):
        import shutil
        shutil.rmtree(self.tmp, ignore_errors=True)


def cpu_newref_leg(corrected, bins, k, binsize, idx_gpu, rows, all_cores=True):
    """The oracle's get_reference (`kind: port`: numpy temporaries + Python insertion scan, the
    reference's own structure) on `rows` target rows x all candidates: one core, then all cores with
    the reference's `newref -cpus N` process model (N processes over N row parts, wisecondor.py:47-56)."""
    import numpy as np
    B, S = corrected.shape
    cores = usable_cores()
    workers = max(1, min(cores, 256))
    w = CpuWorkers()
    try:
        np.save(os.path.join(w.tmp, "corrected.npy"), corrected)     # keeps the memory order (summation order)
        np.savez(os.path.join(w.tmp, "reference.npz"), bins=bins, k=k, binsize=binsize)
        sums = np.cumsum(bins)

        def pairs_of(lo, hi):
            return float(sum(B - bins[np.searchsorted(sums, r, side="right")] for r in range(lo, hi)))

        out = {"unit": "bin-pair distances/s", "kind": "port", "cpu_model": cpu_model(), "host_cores": cores}
        _, o1 = w.run(1, "newref", rows)
        lo, hi = o1[0]["rows"]
        ok = bool(np.array_equal(np.load(os.path.join(w.tmp, "newref_0.npy")), idx_gpu[lo:hi]))
        out.update({"value": pairs_of(lo, hi) / o1[0]["seconds"], "cores": 1,
                    "sample": "oracle get_reference on target rows [%d,%d) of %d x all candidates (%d samples), %.1f s"
                              % (lo, hi, B, S, o1[0]["seconds"]),
                    "matches_gpu_indices": ok})
        if all_cores:
            # wall time includes nothing but the work: the processes start from a common signal
            wall, oN = w.run(workers, "newref", rows)
            busy = max(o["seconds"] for o in oN)
            done = sum(pairs_of(*o["rows"]) for o in oN)
            out["all_cores"] = {"value": done / wall, "unit": "bin-pair distances/s", "cores": workers,
                                "sample": "%d processes (newref -cpus %d model), up to %d target rows each: %.1f s wall "
                                          "from a common start, slowest process %.1f s" % (workers, workers, rows, wall, busy)}
        return out
    finally:
        w.close()


def cpu_test_leg(inp, binsize, k, reference_arrays, tb_calls):
    """The oracle's test_sample (one np.sum per Stouffer window): one sample on one core, then one
    sample per core (the reference's `test` is single-process: N samples in N processes)."""
    import numpy as np
    cores = usable_cores()
    workers = max(1, min(cores, 256))
    w = CpuWorkers()
    try:
        np.savez(os.path.join(w.tmp, "reference.npz"), bins=inp["masked_bins"], k=k, binsize=binsize, **reference_arrays)
        n_tests = min(len(inp["tests"]), workers)
        for i in range(n_tests):
            np.savez(os.path.join(w.tmp, "sample_%d.npz" % i), **inp["tests"][i])
        _, t1 = w.run(1, "test")
        gpu_calls = tb_calls(0)
        cpu_calls = np.asarray(t1[0]["calls"], dtype=np.float64).reshape(-1, 5)
        out = {"value": 1.0 / t1[0]["seconds"], "unit": "samples/s", "cores": 1, "kind": "port",
               "sample": "oracle test_sample on 1 of the batch's samples, %.1f s" % t1[0]["seconds"],
               "matches_gpu_calls": bool(cpu_calls.shape == gpu_calls.shape and
                                         np.array_equal(cpu_calls[:, :3], gpu_calls[:, :3]))}
        if n_tests > 1:
            wall, tN = w.run(n_tests, "test")
            busy = max(o["seconds"] for o in tN)
            out["all_cores"] = {"value": n_tests / wall, "unit": "samples/s", "cores": n_tests,
                                "sample": "%d independent samples in %d processes: %.1f s wall from a common "
                                          "start, slowest process %.1f s" % (n_tests, n_tests, wall, busy)}
        return out
    finally:
        w.close()


def cpu_segments_leg(regions, thr, total_windows, gpu_segments):
    """the oracle's fillTri + segmentTri (the part that is ~90 % of the
    reference's `test`) on one sample's longest chromosomes, one process each, scaled to a whole
    sample by window count.  gpu_segments: per region the (value) list the GPU path called there."""
    import numpy as np
    w = CpuWorkers()
    try:
        for i, z in enumerate(regions):
            np.save(os.path.join(w.tmp, "region_%d.npy" % i), np.ascontiguousarray(z, dtype=np.float64))
        np.save(os.path.join(w.tmp, "region_thr.npy"), np.float64(thr))
        wall, outs = w.run(len(regions), "segments")
        cpu_s = sum(o["seconds"] for o in outs)
        win = sum(o["windows"] for o in outs)
        same = all(len(o["segments"]) == len(g) and
                   np.array_equal(np.array([v for v, _, _ in o["segments"]], dtype=np.float64).view(np.int64),
                                  np.asarray(g, dtype=np.float64).view(np.int64))
                   for o, g in zip(outs, gpu_segments))
        per_sample_s = cpu_s * total_windows / float(win)
        return {"value": 1.0 / per_sample_s, "unit": "samples/s", "cores": 1, "kind": "port", "cpu_model": cpu_model(),
                "sample": "oracle fill_tri + segment_tri on one sample's %d longest chromosomes (%s bins, %d of the "
                          "sample's %d windows), one process each: %.1f core-seconds, %.1f s wall; scaled to a whole "
                          "sample by window count (the z-score repeats, ~8 %% of the reference's test, not included)"
                          % (len(regions), "/".join(str(o["bins"]) for o in outs), win, int(total_windows), cpu_s, wall),
                "windows_per_core_second": win / cpu_s,
                "matches_gpu_segments": bool(same)}
    finally:
        w.close()


def run_cpu_baseline(inp, binsize, k, idx_gpu, reference_arrays, tb_calls, budget_s=8.0):
    """cpu_baseline of the headline workload: the newref leg (one core + all cores) with the test leg
    under `test`, and the oracle / reference time ratio measured in the development container."""
    corrected = inp["corrected"]
    B, S = corrected.shape
    # rows per process so that one process works for about budget_s (the numpy distance touches
    # ~B * S elements three times per target row)
    rows = int(max(8, min(B, budget_s * 1.0e9 / (float(B) * S))))
    out = cpu_newref_leg(corrected, inp["masked_bins"], k, binsize, idx_gpu, rows)
    out["test"] = cpu_test_leg(inp, binsize, k, reference_arrays, tb_calls)
    for rnd in ROUNDS:
        ratio_path = os.path.join(ROOT, "profiles", "%s_oracle_vs_reference.json" % rnd)
        if os.path.exists(ratio_path):      # measured in the development container, where the reference can run
            out["port_vs_reference"] = json.load(open(ratio_path))
            break
    return out


# ---------------------------------------------------------------- rooflines ----
GRAM_KERNELS = {
    "f16": ("k_gram_glds (symmetric distance tiles, ONE float16 matrix-core product per multiply, every row's "
            "representation error charged to its norm bounds, operand slabs by LDS-DMA, + candidate filter)", 1.0,
            PEAK_BF16_MFMA),
}


def gram_roofline(mode, flops, gram_ms, variants, traffic):
    """The distance-tile kernel against the matrix-core peak of the operand type it runs on: executed flops
    (products per multiply x 2 S per unordered pair) over the event time, with SURVEY.md 8(d)'s wording
    (algorithmic flops against the fp32 MFMA peak) beside it."""
    name, products, peak = GRAM_KERNELS[mode]
    achieved = products * flops / (gram_ms * 1e-3)
    roof = {"kernel": name, "bound": "mfma", "achieved": achieved / 1e12, "peak": peak / 1e12, "unit": "TFLOP/s",
            "frac": achieved / peak, "frac_executed_vs_16bit_mfma_peak": None if mode == "f32" else achieved / peak,
            "frac_algorithmic_vs_fp32_mfma": flops / (gram_ms * 1e-3) / PEAK_FP32_MFMA,
            "traffic": hbm_bytes(traffic),
            "traffic_unit": "HBM-side bytes per launch (rocprofv3 PMC, %s)" % traffic.get("source"),
            "kernel_ms": gram_ms, "algorithmic_flop_per_launch": flops, "executed_flop_per_launch": products * flops,
            "algorithmic_tflops": flops / (gram_ms * 1e-3) / 1e12,
            "note": "algorithmic work = 2 S flop per unordered cross-chromosome pair (SURVEY.md 8d); at the small "
                    "configs the tile's life is its epilogue (candidate filter + list appends), not its MFMAs"}
    others = {}
    for other, ms in sorted(variants.items()):
        _, pr, pk = GRAM_KERNELS[other]
        others[other] = {"kernel_ms": ms, "executed_tflops": pr * flops / (ms * 1e-3) / 1e12, "peak": pk / 1e12,
                         "frac": pr * flops / (ms * 1e-3) / pk}
    roof["other_tile_modes"] = others or None
    return roof


def test_roofline(prof, n_refs, windows, n_samples, ms_per_batch, n_bins=0):
    """`test` has no single binding roof: per-stage times of one batch (events on the launch stream inside
    the library), the z-score stage's gathered bytes against the L2 bandwidth (the sample matrix sits in
    L2 / Infinity Cache, not HBM), the window search's own float64 rate against the vector peak, and the
    SURVEY.md 8(d) byte model (which charges 8 B per window for a triangle that is never written)."""
    search_ms = float(prof[5])
    search_ops = 4.0 * float(prof[6]) + 6.0 * float(prof[7])   # sub, scale, max, min per window; ~6 per bound
    valu_frac = (search_ops / (search_ms * 1e-3) / PEAK_FP64_VALU_OPS) if search_ms > 0 else None
    test_bytes = 5.0 * n_refs * 12.0 + windows * 8.0
    byte_frac = n_samples * test_bytes / (ms_per_batch * 1e-3) / PEAK_HBM
    z_ms = float(prof[1])
    z_bytes = n_samples * n_refs * 8.0                          # first repeat: every reference value once
    tiled = n_samples >= 113                                    # (padded to 128 samples and more: k_zscore_tiled)
    gather_roof = committed_gather_roof(n_bins, 3 if tiled else 0) if n_bins else None
    untiled_roof = committed_gather_roof(n_bins, 0) if (n_bins and tiled) else None
    return {
        "bound": "fp64-valu" if byte_frac > 1.0 else "hbm",
        "frac": valu_frac if byte_frac > 1.0 else byte_frac,
        "fp64_valu_frac": valu_frac,
        # every window of the (never materialised) triangle x 4 float64 operations against the vector peak: what a
        # search that evaluated them all would have to sustain -- over the whole batch and over the search stage alone
        "algorithmic_fp64_frac": n_samples * windows * 4.0 / (ms_per_batch * 1e-3) / PEAK_FP64_VALU_OPS,
        "algorithmic_fp64_frac_of_search_stage": (n_samples * windows * 4.0 / (search_ms * 1e-3) / PEAK_FP64_VALU_OPS)
                                                 if search_ms > 0 else None,
        "fp64_valu_detail": {"kernels": "the window-search kernels of one batch (the cell search: k_seg_walk, or k_seg_job + "
                                        "k_seg_merge per recursion level; events on the launch stream): windows evaluated by "
                                        "value x 4 + bounds x 6 float64 operations, counted by the kernels",
                             "ms": search_ms, "windows_evaluated": float(prof[6]),
                             "certificate_evaluations": float(prof[7]), "fp64_ops": search_ops,
                             "peak_ops_per_s": PEAK_FP64_VALU_OPS},
        "zscore_gather": {"kernels": "%s + the later repeats' pair kernels" % ("k_zscore_tiled" if tiled else "k_zscore"), "ms": z_ms,
                          "gathered_bytes_per_batch": z_bytes,
                          "achieved_GBps": (z_bytes / (z_ms * 1e-3) / 1e9) if z_ms > 0 else None,
                          "l2_peak_GBps": PEAK_L2 / 1e9,
                          "l2_frac": (z_bytes / (z_ms * 1e-3) / PEAK_L2) if z_ms > 0 else None,
                          "measured_gather_roof": gather_roof,
                          "measured_gather_roof_untiled": untiled_roof,
                          "frac_of_measured_gather_roof": (z_bytes / (z_ms * 1e-3) / (gather_roof["TBps"] * 1e12))
                                                          if (z_ms > 0 and gather_roof) else None},
        "hbm_byte_model": {"model": "5 repeats x gathered refs x 12 B + 8 B per Stouffer window (SURVEY.md 8d); "
                                    "triangle never materialised", "bytes_per_sample": test_bytes,
                           "achieved_GBps": n_samples * test_bytes / (ms_per_batch * 1e-3) / 1e9,
                           "peak_GBps": PEAK_HBM / 1e9, "frac": byte_frac},
        "windows_decided_per_s": n_samples * windows / (ms_per_batch * 1e-3),
        "stage_ms_per_batch": {"prepare": float(prof[0]), "zscore_repeats": z_ms,
                               "reshape_clean": float(prof[2]), "segmentation": float(prof[3]),
                               "calls_outputs": float(prof[4]), "of_segmentation_search": search_ms},
    }


# ------------------------------------------------ a node's worth of ranks, one after the other ----
XGMI_LINK = 150.0e9          # bytes/s of one xGMI link as the projection prices it (153 GB/s peak per link)
COLLECTIVE_LATENCY_S = 25e-6  # assumed start-up cost of one RCCL collective on an 8-GPU node (not measured here)


def emulate_world_newref(ctx, X, bins, k, order, world, single_idx, single_dst):
    """What each rank of a `world`-GPU newref job would do, run serially on THIS GPU with the real HIP stages:
    per-rank milliseconds of every local stage in both shard modes, the bytes each collective moves, and a
    PROJECTED step time -- slowest rank + bytes over one xGMI link + collective latencies.  Nothing here is a
    multi-GPU measurement: RCCL between GPUs first runs in the driver's --gpus N runs."""
    import numpy as np
    import torch
    from wisecondor_amd import distributed
    st = distributed.HipStages(ctx, X, bins, k, order)
    B = st.n_bins
    ranges = [distributed.row_range(r, world, B) for r in range(world)]
    st.prepare()
    cap_x = distributed.exchange_capacity(st.cap, world)
    dev = X.device
    max_rows = max(e - b for b, e in ranges)

    def timed(fn):
        a, b_ = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        a.record()
        fn()
        b_.record()
        return a, b_

    # what the threshold all-gather delivers: every row's threshold
    thr_all = torch.empty(B, dtype=torch.float32, device=dev)
    st.thresholds(0, B)
    st.get_thr(0, B, thr_all)
    send_cnt = [[torch.zeros(max_rows, dtype=torch.int32, device=dev) for _ in range(world)] for _ in range(world)]
    send_lst = [[torch.zeros((max_rows, cap_x), dtype=torch.int64, device=dev) for _ in range(world)] for _ in range(world)]
    idx = torch.empty((B, k), dtype=torch.int32, device=dev)
    dst = torch.empty((B, k), dtype=torch.float64, device=dev)
    ev = {m: [dict() for _ in range(world)] for m in ("tiles", "rows")}
    exact_rows = []          # tile shard: rows of each rank that took the exact path (exchange-slot overflow included)

    def tile_share(r, export):
        rb, re = ranges[r]
        ev["tiles"][r]["prepare"] = timed(st.prepare)
        ev["tiles"][r]["thresholds"] = timed(lambda: st.thresholds(rb, re))
        for q, (b, e) in enumerate(ranges):
            if q != r:
                st.set_thr(b, e, thr_all[b:e])
        ev["tiles"][r]["collect"] = timed(lambda: st.collect(0, B, r, world))
        if export:
            def ex():
                for q, (b, e) in enumerate(ranges):
                    if q != r:
                        st.export(b, e, cap_x, send_cnt[r][q], send_lst[r][q])
            ev["tiles"][r]["export"] = timed(ex)

    for r in range(world):            # every rank's tile share once, to fill the exchange buffers
        tile_share(r, True)
    for r in range(world):            # again, now with the other ranks' lists to import, through to the rows' results
        rb, re = ranges[r]
        exp = ev["tiles"][r]["export"]
        tile_share(r, False)
        ev["tiles"][r]["export"] = exp

        def im():
            for q in range(world):
                if q != r:
                    st.import_(rb, re, cap_x, send_cnt[q][r], send_lst[q][r])
        ev["tiles"][r]["import"] = timed(im)
        ev["tiles"][r]["finish"] = timed(lambda: st.finish(rb, re, idx[rb:re], dst[rb:re]))
        from wisecondor_amd import wisetools as _wt
        exact_rows.append(int(_wt.newref_stats(dev.index or 0)["fallback_rows"]))     # (synchronises: the events are recorded)
    torch.cuda.synchronize()
    tiles_ok = bool(torch.equal(idx, single_idx) and torch.equal(dst.view(torch.int64), single_dst.view(torch.int64)))
    idx.zero_()
    for r in range(world):            # row shard: the rank's band against all columns, no exchange
        rb, re = ranges[r]
        ev["rows"][r]["prepare"] = timed(st.prepare)
        ev["rows"][r]["thresholds"] = timed(lambda: st.thresholds(rb, re))
        ev["rows"][r]["collect"] = timed(lambda: st.collect(rb, re, 0, 1))
        ev["rows"][r]["finish"] = timed(lambda: st.finish(rb, re, idx[rb:re], dst[rb:re]))
    torch.cuda.synchronize()
    rows_ok = bool(torch.equal(idx, single_idx) and torch.equal(dst.view(torch.int64), single_dst.view(torch.int64)))

    out = {"what": "every rank's share of a %d-GPU newref job run serially on ONE GPU (real HIP stages, events around "
                   "each); `projected_*` are PROJECTIONS, not measurements: slowest rank + the bytes one rank receives "
                   "over ONE %.0f GB/s xGMI link + %d collective start-ups of an assumed %.0f us"
                   % (world, XGMI_LINK / 1e9, 3, COLLECTIVE_LATENCY_S * 1e6),
           "world": world, "exchange_slots_per_row_and_source": cap_x}
    bytes_thr = float(B) * 4.0
    per_rank_recv_lists = [(world - 1) * ((ranges[r][1] - ranges[r][0]) * (4.0 + cap_x * 8.0)) for r in range(world)]
    bytes_result = float(B) * k * 12.0
    n_bands = distributed.band_count()

    def overlapped(ms, coll, mode):
        """The step as NewrefJob runs it: row bands, every collective on the communicator's stream beside the launch
        stream's kernels (two cursors; a collective starts when it is issued AND the previous one is done, and is
        priced like the serial projection: bytes over one link + a start-up each; a band's share of a stage = 1 / bands)."""
        comm_ms = lambda nbytes: 1e3 * (nbytes / XGMI_LINK + COLLECTIVE_LATENCY_S)
        t = ms["prepare"] + ms["thresholds"]
        c = 0.0                                                   # the communicator's stream is free from here
        x_done = []
        if mode == "tiles":
            t = c = t + comm_ms(coll["threshold_all_gather"])     # blocking: the tiles need every threshold
            t += ms["collect"]
            for i in range(n_bands):
                t += ms["export"] / n_bands
                c = max(c, t) + comm_ms(coll["list_all_to_all"] / n_bands)
                x_done.append(c)
        else:
            t += ms["collect"]
        for i in range(n_bands):
            if mode == "tiles":
                t = max(t, x_done[i]) + ms["import"] / n_bands
            t += ms["finish"] / n_bands
            c = max(c, t) + comm_ms(coll["result_all_gather"] / n_bands)
        return max(t, c)

    for mode in ("tiles", "rows"):
        per = []
        for r in range(world):
            ms = {name: a.elapsed_time(b_) for name, (a, b_) in ev[mode][r].items()}
            ms["total"] = sum(ms.values())
            per.append(ms)
        tot = [p_["total"] for p_ in per]
        coll = ({"threshold_all_gather": bytes_thr, "list_all_to_all": max(per_rank_recv_lists),
                 "result_all_gather": bytes_result} if mode == "tiles" else {"result_all_gather": bytes_result})
        comm_s = sum(coll.values()) / XGMI_LINK + len(coll) * COLLECTIVE_LATENCY_S
        out[mode] = {"per_rank_ms": per, "max_rank_ms": max(tot), "mean_rank_ms": float(np.mean(tot)),
                     "imbalance_max_over_mean": max(tot) / float(np.mean(tot)),
                     "bytes_received_per_rank_per_collective": coll,
                     "projected_comm_ms": 1e3 * comm_s, "projected_step_ms": max(tot) + 1e3 * comm_s,
                     "projected_step_ms_overlapped": max(overlapped(p_, coll, mode) for p_ in per),
                     "row_bands": n_bands,
                     "results_equal_single_rank": tiles_ok if mode == "tiles" else rows_ok}
        if mode == "tiles":
            out[mode]["exact_path_rows_per_rank"] = exact_rows
    out["what"] += ("; projected_step_ms_overlapped: the same prices with the schedule NewrefJob runs -- %d row bands per "
                    "rank, every collective in flight beside the kernels (still a PROJECTION)" % n_bands)
    return out


# ------------------------------------------------------------- the one line ----
LINE_LIMIT = 6000       # bytes: the driver's parser lost a 21 KB line in round 5


def _pick(d, *keys):
    return None if not isinstance(d, dict) else {k_: d.get(k_) for k_ in keys if k_ in d}


def _num(x, digits=6):
    if isinstance(x, float):
        if x != x or x in (float("inf"), float("-inf")):
            return None                         # no NaN / Infinity tokens on the line
        return float("%.*g" % (digits, x))
    if isinstance(x, dict):
        return {k_: _num(v, digits) for k_, v in x.items()}
    if isinstance(x, (list, tuple)):
        return [_num(v, digits) for v in x]
    return x


def compact_line(d):
    """The line the driver parses: the contract's keys, `roofline` and `cpu_baseline` as flat objects of scalars,
    a handful of the other legs' headline scalars.  Everything else stays in the detail record."""
    roof = dict(d.get("roofline") or {})
    name = str(roof.get("kernel", ""))
    short = "k_rescore" if name.startswith("float64 re-score") else (name.split(" ")[0] if name else None)
    r = {"kernel": short}
    for k_ in ("kernel_ms", "bound", "achieved", "peak", "unit", "frac", "traffic", "algorithmic_bytes_per_launch",
               "algorithmic_flop_per_launch", "gathered_bytes_per_launch", "frac_of_measured_gather_roof",
               "frac_algorithmic_vs_fp32_mfma"):
        if k_ in roof:
            r[k_] = roof[k_]
    other = d.get("roofline_other") or {}
    oname = str(other.get("kernel", ""))
    r["other"] = {"kernel": "k_rescore" if oname.startswith("float64 re-score") else (oname.split(" ")[0] or None),
                  "kernel_ms": other.get("kernel_ms"), "bound": other.get("bound"), "frac": other.get("frac")} if other else None
    cpu = d.get("cpu_baseline")
    c = None
    if isinstance(cpu, dict):
        c = _pick(cpu, "value", "unit", "cores", "cpu_model", "kind", "sample", "matches_gpu_indices", "error")
        if isinstance(cpu.get("all_cores"), dict):
            c["all_cores"] = _pick(cpu["all_cores"], "value", "cores")
        if isinstance(cpu.get("test"), dict):
            c["test"] = _pick(cpu["test"], "value", "unit", "cores", "matches_gpu_calls")
    t = d.get("test") or {}
    wj = t.get("whole_job_1000_samples") or {}
    test = {"value": t.get("value"), "unit": "samples/s", "ms_per_batch": t.get("ms_per_batch"),
            "samples_per_gpu": t.get("samples_per_gpu"), "pipelined_value": (t.get("pipelined") or {}).get("value"),
            "latency_ms_per_call": t.get("single_sample_latency_ms"),
            "whole_job_1000_samples": _pick(wj, "ms_per_call", "samples_per_s", "distinct_samples", "error"),
            "roofline_frac": (t.get("roofline") or {}).get("frac"), "roofline_bound": (t.get("roofline") or {}).get("bound"),
            "calls_found": t.get("calls_found")}
    ex = d.get("extra")
    extra = None
    if isinstance(ex, dict):
        extra = _pick(ex, "ms_per_step", "value", "shard_mode", "k_gram_ms", "k_gram_frac_of_mfma_peak", "rescore_ms", "error")
        rr = ex.get("rescore_roofline")
        if isinstance(rr, dict):
            extra["rescore_frac_of_hbm"] = rr.get("frac")
        em = ex.get("emulated_world_8")
        if isinstance(em, dict) and "error" not in em:
            extra["emulated_world_8_projection"] = {m: _pick(em.get(m) or {}, "max_rank_ms", "imbalance_max_over_mean",
                                                             "projected_step_ms_overlapped", "results_equal_single_rank")
                                                    for m in ("tiles", "rows") if m in em}
        t5 = ex.get("test_50kb")
        if isinstance(t5, dict):
            w5 = t5.get("whole_job_1000_samples") or {}
            extra["test_50kb"] = {"ms_per_batch": t5.get("ms_per_batch"), "value": t5.get("value"),
                                  "pipelined_value": (t5.get("pipelined") or {}).get("value"),
                                  "whole_job_1000_samples": _pick(w5, "ms_per_call", "samples_per_s"),
                                  "cpu_baseline_value": (t5.get("cpu_baseline") or {}).get("value"),
                                  "error": t5.get("error")}
        if isinstance(ex.get("cpu_baseline"), dict):
            extra["cpu_baseline"] = _pick(ex["cpu_baseline"], "value", "cores", "matches_gpu_indices", "error")
        if isinstance(ex.get("ingest"), dict):
            extra["ingest_files_per_s"] = ex["ingest"].get("files_per_s")
    mr = d.get("multi_rank")
    multi = None
    if isinstance(mr, dict) and mr.get("per_rank"):
        multi = {"row_bands_per_rank": mr.get("row_bands_per_rank"),
                 "stage_ms_sum_per_rank": [sum((e or {}).get("stages_ms", {}).values()) for e in mr["per_rank"]],
                 "collective_ms_sum_per_rank": [sum(c_["ms"] for c_ in (e or {}).get("collectives", [])) for e in mr["per_rank"]],
                 "collective_bytes_rank0": [[c_["name"], c_["bytes"]] for c_ in (mr["per_rank"][0] or {}).get("collectives", [])][:8]}
    cfg = d.get("config") or {}
    line = {k_: d.get(k_) for k_ in ("metric", "value", "unit", "n_gpus", "steps", "warmup", "ms_per_step",
                                      "higher_is_better", "scaling", "vs_baseline", "dtype", "data")}
    line["config"] = _pick(cfg, "workload", "parallelism", "world_size", "shard_mode")
    line["roofline"] = r
    line["cpu_baseline"] = c
    line["test"] = test
    line["extra"] = extra
    line["stages_ms"] = d.get("stages_ms")
    line["multi_rank"] = multi
    line["prep_ms"] = (d.get("prep") or {}).get("ms")
    line["detail"] = d.get("detail_file")
    return _num(line)


def emit(detail, path):
    """Write the full record to `path` (when it can be written), print the compact line LAST."""
    try:
        with open(path, "w") as fh:
            json.dump(detail, fh, indent=1)
        detail["detail_file"] = os.path.relpath(path, ROOT) if path.startswith(ROOT) else path
    except Exception as exc:
        detail["detail_file"] = "not written: %s" % exc
    line = json.dumps(compact_line(detail), allow_nan=False, separators=(",", ":"))
    if len(line) > LINE_LIMIT:          # never again an unparseable line: drop the optional objects, largest first
        slim = compact_line(detail)
        for k_ in ("multi_rank", "stages_ms", "extra", "test"):
            slim[k_] = None
            line = json.dumps(slim, allow_nan=False, separators=(",", ":"))
            if len(line) <= LINE_LIMIT:
                break
    sys.stdout.flush()
    print(line)
    sys.stdout.flush()


# --------------------------------------------------------------------- main ----
def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=20)
    ap.add_argument("--warmup", type=int, default=3)
    ap.add_argument("--workload", default="cfg2", choices=sorted(WORKLOADS))
    ap.add_argument("--test-samples", type=int, default=128, help="test samples per GPU in the batched test pass")
    ap.add_argument("--refsize", type=int, default=100)
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--no-extra", action="store_true", help="skip the extra 600 x 50 kb newref measurement")
    ap.add_argument("--backend", default="nccl", help="torch.distributed backend (nccl = RCCL; gloo only for "
                    "functional tests of the multi-rank path on a box with fewer GPUs than ranks)")
    ap.add_argument("--detail", default=os.path.join(ROOT, "bench_detail.json"),
                    help="where the full measurement record goes (stage tables, per-rank lists, emulated world, "
                         "every roofline object); the LAST stdout line is a compact summary of it")
    ap.add_argument("--launch-check", action="store_true",
                    help="only start the ranks, rendezvous (gloo) and report the world size: no GPU work")
    args = ap.parse_args()

    env_world = os.environ.get("WORLD_SIZE")
    if env_world is None and args.gpus > 1:
        sys.exit(spawn_ranks(args, sys.argv[1:]))
    world = int(env_world or "1")
    if world != args.gpus:
        print("bench.py: --gpus %d but the launcher started %d rank(s)" % (args.gpus, world), file=sys.stderr)
        sys.exit(2)
    if args.launch_check:
        launch_check(args)
        return

    import numpy as np
    import torch
    import torch.distributed as dist
    from wisecondor_amd import _lib, distributed
    from wisecondor_amd import wisetools as wt
    from wisecondor_amd.wisecondor import zThreshold

    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    if args.backend == "gloo":
        local_rank = local_rank % max(1, torch.cuda.device_count())   # ranks may share a GPU in tests
    if world > 1:
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        os.environ.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")
        torch.cuda.set_device(local_rank)
        if args.backend == "nccl":
            dist.init_process_group("nccl", device_id=torch.device("cuda", local_rank))
        else:
            dist.init_process_group(args.backend)
        if dist.get_world_size() != args.gpus:
            print("bench.py: process group has %d ranks, --gpus says %d" % (dist.get_world_size(), args.gpus),
                  file=sys.stderr)
            sys.exit(2)
    dev = torch.device("cuda", local_rank)
    torch.cuda.set_device(dev)
    lib = _lib.load()
    ctx = _lib.context(local_rank)

    binsize, n_ref = WORKLOADS[args.workload]
    k = args.refsize
    inp = build_inputs(binsize, n_ref, args.test_samples, seed0=0, device=local_rank)
    corrected = inp["corrected"]                       # Fortran-ordered, like the reference's prep file
    order = wt.sum_order_of(corrected)
    B, S = corrected.shape
    bins = np.ascontiguousarray(inp["masked_bins"])
    pairs = float(B) * B - float((bins.astype(np.float64) ** 2).sum())   # ordered cross-chromosome pairs
    X = torch.from_numpy(np.ascontiguousarray(corrected)).to(dev)
    job = distributed.NewrefJob(ctx, X, bins, k, order, rank=rank, world=world, passes=2 * args.steps + args.warmup + 1)
    tdev = dev if args.backend == "nccl" else torch.device("cpu")

    def sync_all():
        torch.cuda.synchronize()
        if world > 1:
            dist.barrier()
            torch.cuda.synchronize()

    def max_over_ranks(seconds):
        t = torch.tensor([seconds], device=tdev, dtype=torch.float64)
        if world > 1:
            dist.all_reduce(t, op=dist.ReduceOp.MAX)
        return float(t.item())

    def mean_stages(a_job, mark_sets):
        runs = []
        for m in mark_sets:
            a_job.last_marks = m
            runs.append(a_job.stage_ms())
        return {key: float(np.mean([r.get(key, 0.0) for r in runs])) for key in runs[0]}

    def mean_collectives(a_job, comm_sets):
        """Per collective of a multi-rank pass (issued blocking in the timing passes): name, bytes, mean ms."""
        runs = []
        for c in comm_sets:
            a_job.last_comm = c
            runs.append(a_job.collective_ms())
        if not runs or not runs[0]:
            return []
        return [{"name": runs[0][i]["name"], "bytes": runs[0][i]["bytes"],
                 "ms": float(np.mean([r[i]["ms"] for r in runs if i < len(r)]))} for i in range(len(runs[0]))]

    # ------------------------------------------------------------ newref ----
    idx, dst = job.run()                              # (the first pass of a multi-rank job also measures the shard mode)
    sync_all()
    # K passes with events between the stages (on the launch stream): kernel times for the roofline objects; the events
    # cost ~15 % of a 0.2 ms pass, so they stay out of `value`.  They run FIRST (round 6; behind the timed region
    # before): the chip's clocks settle over the first ~15 ms of load (consecutive groups of 20
    # passes read 0.212, 0.208, 0.203, 0.198 ms on a GPU that has just been idle), and this leg is load like any other.
    marks, comms = [], []
    for s in range(args.steps):
        idx, dst = job.run(timing=True)
        marks.append(job.last_marks)
        comms.append(job.last_comm)
    sync_all()
    for _ in range(max(1, args.warmup)):              # W untimed passes ...
        idx, dst = job.run()
    sync_all()
    t0 = time.perf_counter()
    for s in range(args.steps):
        idx, dst = job.run()                          # ... then the timed region: K plain passes, nothing else
    sync_all()
    t_newref = max_over_ranks(time.perf_counter() - t0)
    stages = mean_stages(job, marks)             # every interval booked under the stage that ended it (bands add up)
    collectives = mean_collectives(job, comms)
    gram_ms = stages.get("collected")
    pick_ms = stages.get("picked")
    rescore_ms = stages.get("rescored")
    finish_ms = None if rescore_ms is None else rescore_ms + (pick_ms or 0.0)
    per_rank = None
    if world > 1:
        # every rank's stage times and collectives on the one line: a scaling run is diagnosable from it
        per_rank = [None] * world
        dist.all_gather_object(per_rank, {"rank": rank, "stages_ms": stages, "collectives": collectives})
    stats = wt.newref_stats(local_rank)
    gram_mode = "f16"
    variants = {}
    ms_per_step = 1e3 * t_newref / args.steps
    value = pairs * args.steps / t_newref

    # -------------------------------------------------------------- test ----
    idx_h, dst_h = idx.cpu().numpy(), dst.cpu().numpy()
    reference = wt.Reference(idx_h, dst_h, inp["chrom_bins"], inp["masked_bins"], inp["mask"], inp["pca_mean"],
                             inp["pca_components"], binsize=binsize, device=local_rank)
    thr = float(zThreshold([int(v) for v in inp["masked_bins"]], 1000, None))
    counts_h = wt.samples_to_counts(inp["tests"], inp["chrom_bins"])
    tb = distributed.TestBatch(reference, torch.from_numpy(counts_h).to(dev), thr, max_calls=256)
    for _ in range(max(3, args.warmup // 2)):
        tb.run()
    sync_all()
    t0 = time.perf_counter()
    test_steps = max(10, args.steps // 2)
    for _ in range(test_steps):
        tb.run()
    sync_all()
    t_test = max_over_ranks(time.perf_counter() - t0)
    samples_per_s = world * args.test_samples * test_steps / t_test
    n_calls = int(tb.n_calls.sum().item())
    # the same batches with FOUR in flight (distributed.TestPipeline: a context, a stream and a host thread per slot; a
    # batch's narrow kernels run beside the others' wide ones) -- the throughput form of the batched test
    one_in_flight = {"ms_per_batch": 1e3 * t_test / test_steps, "value": samples_per_s, "unit": "samples/s"}
    pipe = distributed.TestPipeline(reference, thr, depth=PIPE_DEPTH, max_calls=256)
    pipe_batches = [tb.counts] * max(12 * PIPE_DEPTH, 2 * test_steps)
    pipe.run(pipe_batches[:2 * PIPE_DEPTH])
    sync_all()
    t0 = time.perf_counter()
    pipe.run(pipe_batches)
    sync_all()
    t_pipe = max_over_ranks(time.perf_counter() - t0) / len(pipe_batches)
    pipe.close()
    pipelined = {"ms_per_batch": 1e3 * t_pipe, "value": world * args.test_samples / t_pipe, "unit": "samples/s",
                 "batches_in_flight": PIPE_DEPTH, "batches_timed": len(pipe_batches)}
    # one more batch with the library's stage timer on (events on the launch stream between the
    # stages; the search kernels count the window evaluations they execute)
    prof = np.zeros(8)
    _lib.check(lib.wc_test_profile(ctx, 1))
    tb.run()
    _lib.check(lib.wc_test_profile_read(ctx, _lib.ptr(prof)))
    _lib.check(lib.wc_test_profile(ctx, 0))
    n_refs = float((reference.distances < reference.cutoff).sum())
    windows = float(sum(int(n) * (int(n) + 1) // 2 for n in inp["masked_bins"]))
    # BASELINE config 3: one sample per call (latency mode, nothing amortised over a batch)
    tb1 = distributed.TestBatch(reference, torch.from_numpy(counts_h[:1].copy()).to(dev), thr, max_calls=256)
    # on a stream of its own, as a service thread would call it: on the NULL stream the library has to hop
    # to an internal stream through an event first (a hipGraph cannot be captured on the NULL stream)
    side = torch.cuda.Stream()
    torch.cuda.synchronize()
    floor = np.zeros(2)
    with torch.cuda.stream(side):
        for _ in range(4):      # the first call sizes the workspaces, the second captures the hipGraph
            tb1.run()
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        for _ in range(50):
            tb1.run()
        torch.cuda.synchronize()
        single_ms = 1e3 * (time.perf_counter() - t0) / 50
        _lib.check(lib.wc_launch_floor_us(ctx, side.cuda_stream, LATENCY_LAUNCHES, 50, _lib.ptr(floor)))
    lat_kernels = committed_latency_trace()
    latency = {"workload": "cfg3: test of ONE sample x %d kb bins per call (PCA-apply, 5 z-score repeats, segmentation, "
                           "calls), one hipGraph replay + one synchronize per call" % (binsize // 1000),
               "ms_per_call": single_ms, "samples_per_s": 1e3 / single_ms,
               "launches_in_graph": LATENCY_LAUNCHES,
               "launch_floor_us": float(floor[0]),
               "launch_floor_note": "a chain of %d dependent EMPTY launches on the same stream, measured here with "
                                    "events: what the launch series costs before any kernel does work" % LATENCY_LAUNCHES,
               "kernel_us_sum": None if not lat_kernels else lat_kernels.get("kernel_us_sum"),
               "kernel_us": None if not lat_kernels else lat_kernels.get("kernels"),
               "kernel_source": None if not lat_kernels else lat_kernels.get("source"),
               "frac_of_call_in_launch_floor": float(floor[0]) * 1e-3 / single_ms}
    test_roof = test_roofline(prof, n_refs, windows, args.test_samples, 1e3 * t_test / test_steps, n_bins=B)
    # the north-star's whole `test` job (1000 samples at this bin size) in ONE call on this GPU: the per-batch
    # fixed costs (a few dozen small launches, one synchronize) spread over eight times the samples
    whole_job = None
    if world == 1 and not args.no_extra:
        try:
            from wisecondor_amd import synth as _synth
            big_h = wt.samples_to_counts(make_tests(_synth.bin_profile(binsize), 1000), inp["chrom_bins"])
            assert np.array_equal(big_h[:counts_h.shape[0]], counts_h)       # the batch's samples lead the cohort
            big = torch.from_numpy(big_h).to(dev)
            tbw = distributed.TestBatch(reference, big, thr, max_calls=256)
            for _ in range(2):
                tbw.run()
            torch.cuda.synchronize()
            t0 = time.perf_counter()
            for _ in range(3):
                tbw.run()
            torch.cuda.synchronize()
            tw = (time.perf_counter() - t0) / 3
            whole_job = {"what": "1000 DISTINCT samples x %d kb in one wc_test_batch_dev call" % (binsize // 1000),
                         "samples": 1000, "distinct_samples": True, "ms_per_call": 1e3 * tw, "samples_per_s": 1000 / tw,
                         "calls_found": int(tbw.n_calls.sum().item())}
            del tbw, big
        except Exception as exc:
            whole_job = {"error": "%s: %s" % (type(exc).__name__, exc)}

    # ------------------------------------------- extra: newref at 600 x 50 kb ----
    # BASELINE.json config 4 (the at-scale shape), kernel-level synthetic matrix; reported
    # beside the headline so that the MFMA kernel is also seen where it is not launch bound.
    extra = None
    if not args.no_extra and args.workload != "cfg4":
        try:
            from wisecondor_amd import synth
            xb, xs = WORKLOADS["cfg4"]
            xdata, xbins, _ = synth.corrected_matrix(xb, xs, seed=0)
            XB = int(xdata.shape[0])
            xpairs = float(XB) * XB - float((xbins.astype(np.float64) ** 2).sum())
            XX = torch.from_numpy(xdata).to(dev)
            del xdata
            xjob = distributed.NewrefJob(ctx, XX, xbins, k, _lib.SUM_SEQUENTIAL, rank=rank, world=world, passes=8)
            xsteps = 10
            for _ in range(3):
                xjob.run()
            sync_all()
            xmarks = []
            t0 = time.perf_counter()
            for q in range(xsteps):
                xjob.run(timing=True)
                xmarks.append(xjob.last_marks)
            sync_all()
            xt = max_over_ranks(time.perf_counter() - t0)
            xstages = mean_stages(xjob, xmarks)
            xk_ms = xstages["collected"]
            # tiles mode: every unordered pair once on the node; rows mode: every ordered pair
            xflops = (xpairs if xjob.mode == "rows" else xpairs / 2.0) * 2.0 * xs / world
            extra = {"workload": "cfg4: newref %d samples x %d kb bins (%d bins), kernel-level synthetic matrix"
                                 % (xs, xb // 1000, XB),
                     "value": xpairs * xsteps / xt, "unit": "bin-pair distances/s", "ms_per_step": 1e3 * xt / xsteps,
                     "steps": xsteps, "shard_mode": xjob.mode or "single", "shard_calibration_s": xjob.calibration,
                     "stages_ms": xstages, "k_gram_ms": xk_ms,
                     "k_gram_algorithmic_tflops": xflops / (xk_ms * 1e-3) / 1e12}
            # the re-score stage on this matrix: uncorrelated rows share no candidates, the 295 MB float64
            # image does not fit the 32 MB of L2, and the gathers come from HBM / Infinity Cache
            xr_ms = xstages.get("rescored")
            if xr_ms:
                xr_bytes = float(XB) / world * k * xs * 8.0 + float(XB) / world * k * 12.0
                extra["rescore_ms"] = xr_ms
                extra["rescore_roofline"] = {"bound": "hbm", "achieved": xr_bytes / (xr_ms * 1e-3) / 1e9,
                                             "peak": PEAK_HBM / 1e9, "unit": "GB/s",
                                             "frac": xr_bytes / (xr_ms * 1e-3) / PEAK_HBM,
                                             "algorithmic_bytes_per_launch": xr_bytes,
                                             "l2_counters": committed_l2("cfg4", "k_rescore"),
                                             "note": "k_rescore alone (k_pick's 0.1 ms not included); l2_counters: the committed TCC pass of k_rescore on this "
                                                     "matrix (uncorrelated rows share no candidates)"}
            _, xprod, xpeak = GRAM_KERNELS[gram_mode]
            extra["k_gram_mode"] = gram_mode
            extra["k_gram_executed_tflops"] = xprod * xflops / (xk_ms * 1e-3) / 1e12
            extra["k_gram_frac_of_mfma_peak"] = xprod * xflops / (xk_ms * 1e-3) / xpeak
            extra["k_gram_frac_algorithmic_vs_fp32_mfma"] = xflops / (xk_ms * 1e-3) / PEAK_FP32_MFMA
            if world == 1:
                xidx, xdst = xjob.run()
                torch.cuda.synchronize()
                try:
                    extra["emulated_world_8"] = emulate_world_newref(ctx, XX, xbins, k, _lib.SUM_SEQUENTIAL, 8,
                                                                     xidx.clone(), xdst.clone())
                    extra["emulated_world_8"]["single_rank_ms_per_step"] = extra["ms_per_step"]
                except Exception as exc:
                    extra["emulated_world_8"] = {"error": "%s: %s" % (type(exc).__name__, exc)}
                xidx, xdst = xjob.run()
                torch.cuda.synchronize()
                if rank == 0 and not args.no_cpu_baseline:
                    # >= 8 target rows x the full candidate set on the host
                    try:
                        extra["cpu_baseline"] = cpu_newref_leg(XX.cpu().numpy(), xbins, k, xb, xidx.cpu().numpy(), 8)
                    except Exception as exc:
                        extra["cpu_baseline"] = {"error": "%s: %s" % (type(exc).__name__, exc)}
            xt_traffic = committed_traffic("cfg4")
            if world == 1 and xt_traffic:
                extra["k_gram_hbm_bytes_per_launch"] = hbm_bytes(xt_traffic)
                extra["k_gram_compulsory_bytes"] = float(XB) * ((xs + 31) // 32 * 32) * 4.0
                extra["traffic_source"] = xt_traffic.get("source")
            del xjob, XX
            # the main job's context state was replaced by the extra run; nothing below needs it
        except Exception as exc:      # the headline line must still be printed
            extra = {"workload": "cfg4", "error": "%s: %s" % (type(exc).__name__, exc)}
        # BASELINE.json config 5, one GPU's share: batched test of 125 samples at 50 kb bins (reference from
        # 100 samples through the GPU prep + newref, untimed)
        try:
            n5 = 1000 if (world == 1 and rank == 0) else 125          # one GPU: the north-star's whole cohort, in eight shares
            inp5 = build_inputs(50000, 100, n5, seed0=500, device=local_rank)
            bins5 = np.ascontiguousarray(inp5["masked_bins"])
            X5 = torch.from_numpy(np.ascontiguousarray(inp5["corrected"])).to(dev)
            job5 = distributed.NewrefJob(ctx, X5, bins5, k, wt.sum_order_of(inp5["corrected"]))
            idx5, dst5 = job5.run()
            torch.cuda.synchronize()
            ref5 = wt.Reference(idx5.cpu().numpy(), dst5.cpu().numpy(), inp5["chrom_bins"], inp5["masked_bins"],
                                inp5["mask"], inp5["pca_mean"], inp5["pca_components"], binsize=50000, device=local_rank)
            thr5 = float(zThreshold([int(v) for v in inp5["masked_bins"]], 1000, None))
            counts5 = torch.from_numpy(wt.samples_to_counts(inp5["tests"], inp5["chrom_bins"])).to(dev)
            tb5 = distributed.TestBatch(ref5, counts5[:125].contiguous(), thr5, max_calls=256)
            for _ in range(3):
                tb5.run()
            sync_all()
            t0 = time.perf_counter()
            for _ in range(10):
                tb5.run()
            sync_all()
            t5 = max_over_ranks(time.perf_counter() - t0) / 10
            pipe5 = distributed.TestPipeline(ref5, thr5, depth=PIPE_DEPTH, max_calls=256)
            b5 = [tb5.counts] * (6 * PIPE_DEPTH)
            pipe5.run(b5[:PIPE_DEPTH])
            sync_all()
            t0 = time.perf_counter()
            pipe5.run(b5)
            sync_all()
            t5p = max_over_ranks(time.perf_counter() - t0) / len(b5)
            pipe5.close()
            prof5 = np.zeros(8)
            _lib.check(lib.wc_test_profile(ctx, 1))
            tb5.run()
            _lib.check(lib.wc_test_profile_read(ctx, _lib.ptr(prof5)))
            _lib.check(lib.wc_test_profile(ctx, 0))
            n_refs5 = float((ref5.distances < ref5.cutoff).sum())
            windows5 = float(sum(int(n) * (int(n) + 1) // 2 for n in bins5))
            extra = dict(extra or {})
            extra["test_50kb"] = {"workload": "cfg5, one GPU's share: batched test of 125 samples x 50 kb bins (%d masked bins)"
                                              % int(bins5.sum()),
                                  "value": world * 125 / t5, "unit": "samples/s", "ms_per_batch": 1e3 * t5,
                                  "what": "value / ms_per_batch: ONE batch in flight; `pipelined` beside it",
                                  "one_batch_in_flight": {"ms_per_batch": 1e3 * t5, "value": world * 125 / t5},
                                  "pipelined": {"ms_per_batch": 1e3 * t5p, "value": world * 125 / t5p,
                                                "batches_in_flight": PIPE_DEPTH, "batches_timed": len(b5)},
                                  "samples_per_gpu": 125, "calls_found": int(tb5.n_calls.sum().item()),
                                  "roofline": test_roofline(prof5, n_refs5, windows5, 125, 1e3 * t5, n_bins=int(bins5.sum()))}
            if n5 == 1000:
                # (a) every rank's share of the 8-GPU job (125 of the 1 000 samples, no collective) one after the
                # other on this GPU; (b) the whole cohort in ONE call on this GPU
                shares = []
                for r8 in range(8):
                    tbr = distributed.TestBatch(ref5, counts5[125 * r8:125 * (r8 + 1)].contiguous(), thr5, max_calls=256)
                    tbr.run()
                    torch.cuda.synchronize()
                    t0 = time.perf_counter()
                    for _ in range(3):
                        tbr.run()
                    torch.cuda.synchronize()
                    shares.append(1e3 * (time.perf_counter() - t0) / 3)
                    del tbr
                extra["test_50kb"]["emulated_world_8"] = {
                    "what": "cfg5 on 8 GPUs = eight independent shares of 125 samples (sample shard, no collective): each "
                            "share timed on THIS GPU, one after the other; projected_* is a PROJECTION (slowest share), not "
                            "a multi-GPU measurement",
                    "per_rank_ms": shares, "max_rank_ms": max(shares), "mean_rank_ms": float(np.mean(shares)),
                    "imbalance_max_over_mean": max(shares) / float(np.mean(shares)),
                    "projected_samples_per_s_8_gpus": 1000.0 / (max(shares) * 1e-3)}
                tbw5 = distributed.TestBatch(ref5, counts5, thr5, max_calls=256)
                for _ in range(2):
                    tbw5.run()
                torch.cuda.synchronize()
                t0 = time.perf_counter()
                for _ in range(3):
                    tbw5.run()
                torch.cuda.synchronize()
                tw5 = (time.perf_counter() - t0) / 3
                extra["test_50kb"]["whole_job_1000_samples"] = {
                    "what": "BASELINE config 5 whole: 1000 distinct samples x 50 kb in ONE wc_test_batch_dev call on one GPU",
                    "samples": 1000, "ms_per_call": 1e3 * tw5, "samples_per_s": 1000.0 / tw5,
                    "calls_found": int(tbw5.n_calls.sum().item())}
                del tbw5
            if rank == 0 and world == 1 and not args.no_cpu_baseline:
                # one sample x the three longest chromosomes on the host
                try:
                    counts5 = wt.samples_to_counts(inp5["tests"][:1], inp5["chrom_bins"])
                    data5 = np.empty((1, ref5.n_bins))
                    _lib.check(lib.wc_prepare_samples(ref5.ctx, ref5.handle, _lib.ptr(counts5), 1, _lib.ptr(data5), None))
                    z5, r5, n5, _ = wt.repeatTest(data5[0], None, None, None, None, None, thr5, 5, reference=ref5)
                    keep5 = n5 >= 25
                    offs5 = np.concatenate([[0], np.cumsum(bins5)])
                    longest = sorted(np.argsort(-bins5)[:3])
                    regions = [z5[offs5[c]:offs5[c + 1]][keep5[offs5[c]:offs5[c + 1]]] for c in longest]
                    nc0 = int(tb5.n_calls[0].item())
                    calls0 = tb5.calls[0, :nc0].cpu().numpy()
                    gsegs = [calls0[calls0[:, 0] == c + 1][:, 3] for c in longest]
                    clean_windows = float(sum(int(keep5[offs5[c]:offs5[c + 1]].sum()) *
                                              (int(keep5[offs5[c]:offs5[c + 1]].sum()) + 1) // 2 for c in range(22)))
                    extra["test_50kb"]["cpu_baseline"] = cpu_segments_leg(regions, thr5, clean_windows, gsegs)
                except Exception as exc:
                    extra["test_50kb"]["cpu_baseline"] = {"error": "%s: %s" % (type(exc).__name__, exc)}
            ref5.close()
            del tb5, job5, X5
        except Exception as exc:
            extra = dict(extra or {})
            extra["test_50kb"] = {"error": "%s: %s" % (type(exc).__name__, exc)}

    # ---------------------------------------------------- extra: ingest ----
    # `testbatch` end to end -- converted-sample files on local disk -> GPU batches ->
    # one result file per sample -- in files/s, beside the samples/s of the kernels alone
    if rank == 0 and world == 1 and not args.no_extra:
        try:
            import contextlib
            import io
            import shutil
            from wisecondor_amd import wisecondor as cli
            tmp_io = tempfile.mkdtemp(prefix="wc_ingest_")
            n_files = 4096        # the box spreads a process's threads over its cores only after ~1 s of load
            refpath = os.path.join(tmp_io, "reference.npz")
            np.savez(refpath, arguments={}, runtime={}, binsize=float(binsize), indexes=idx_h, distances=dst_h,
                     chromosome_sizes=inp["chrom_bins"], mask=inp["mask"], masked_sizes=inp["masked_bins"],
                     pca_components=inp["pca_components"], pca_mean=inp["pca_mean"])
            paths_io = []
            for i in range(n_files):
                p_io = os.path.join(tmp_io, "s_%04d.npz" % i)
                if i < 64:
                    np.savez_compressed(p_io, arguments={"binsize": float(binsize)}, runtime={},
                                        sample=inp["tests"][i % len(inp["tests"])], quality={})
                else:
                    shutil.copyfile(paths_io[i % 64], p_io)
                paths_io.append(p_io)
            io_threads = usable_cores()
            buf_io = io.StringIO()
            t0 = time.perf_counter()
            with contextlib.redirect_stdout(buf_io):
                cli.main(["testbatch"] + paths_io + [os.path.join(tmp_io, "out"), refpath, "-batch", "512", "-io",
                                                     str(io_threads)])
            wall_io = time.perf_counter() - t0
            rep = [ln for ln in buf_io.getvalue().splitlines() if ln.startswith("rank 0")][-1]
            files_per_s = float(rep.split("(")[1].split(" files/s")[0])
            extra = dict(extra or {})
            extra["ingest"] = {"what": "testbatch end to end: %d converted-sample files (%d kb bins) on local disk -> native "
                                       "decode pool -> GPU batches of 512 -> native encode pool (zlib level 1, run-length) -> %d result "
                                       "files" % (n_files, binsize // 1000, n_files),
                               "files_per_s": files_per_s, "unit": "files/s", "io_threads": io_threads,
                               "wall_s_incl_reference_load": wall_io, "report": rep}
            shutil.rmtree(tmp_io, ignore_errors=True)
        except Exception as exc:
            extra = dict(extra or {})
            extra["ingest"] = {"error": "%s: %s" % (type(exc).__name__, exc)}

    # ------------------------------------------------------- cpu baseline ----
    cpu = None
    if rank == 0 and world == 1 and not args.no_cpu_baseline:
        ref_arrays = dict(indexes=reference.indexes, distances=reference.distances,
                          chromosome_sizes=inp["chrom_bins"], mask=inp["mask"], masked_sizes=inp["masked_bins"],
                          pca_mean=inp["pca_mean"], pca_components=inp["pca_components"])

        def gpu_calls(i):
            nc = int(tb.n_calls[i].item())
            return tb.calls[i, :nc].cpu().numpy()
        try:
            cpu = run_cpu_baseline(inp, binsize, k, idx_h, ref_arrays, gpu_calls)
        except Exception as exc:
            cpu = {"error": "%s: %s" % (type(exc).__name__, exc)}

    if rank == 0:
        traffic = committed_traffic(args.workload) if world == 1 else {}
        # ---- k_gram: algorithmic work = one multiply-add per sample per unordered pair
        flops = (pairs if job.mode == "rows" else pairs / 2.0) * 2.0 * S / world   # rows mode: ordered pairs
        roof_gram = gram_roofline(gram_mode, flops, gram_ms, variants, traffic)
        # ---- the float64 re-score stage: algorithmic bytes = the candidate rows it must read
        # (rows x k x S x 8 B) + the output it writes.  The rows are gathered, and
        # neighbouring targets share candidates, so L2 / Infinity Cache serve most of the reads:
        # HBM is NOT what binds this kernel (see binding / hbm_counter_frac / l2_model_frac).
        # ---- the float64 re-score: `roofline` candidates are SINGLE kernels; k_rescore is one (kernel_ms = the
        # event interval around it).
        # Algorithmic bytes = the candidate rows it must read (rows x k x S x 8 B) + the output.
        # The rows are gathered, and neighbouring targets share candidates, so L2 / Infinity Cache serve most of
        # the reads: which roof binds is read from the committed counters.
        roof_finish, rescore_stage = None, None
        if rescore_ms:
            rows_here = B / world
            fbytes = rows_here * k * S * 8.0 + rows_here * k * 12.0
            gathered = float(stats.get("rescored", 0)) * S * 8.0      # what the kernel really pulls through L2
            kt = (traffic.get("k_finish") or {}).get("kernels", {}).get("k_rescore")
            ft = hbm_bytes(kt)
            hbm_alg = fbytes / (rescore_ms * 1e-3) / PEAK_HBM
            hbm_cnt = None if ft is None else ft / (rescore_ms * 1e-3) / PEAK_HBM
            l2_frac = gathered / (rescore_ms * 1e-3) / PEAK_L2
            groof = committed_gather_roof(B)
            busy = committed_busy(args.workload)
            # when the memory-side counters see less than half of what the HBM roof would allow (the image is
            # re-read out of L2 / Infinity Cache), the kernel is priced against the L2 gather bandwidth instead
            hbm_binds = hbm_cnt is not None and hbm_cnt >= 0.5 and hbm_alg <= 1.0
            common = {"kernel": "float64 re-score: k_rescore (exact float64 distances of the picked candidates in numpy's "
                                "order, counting order, output rows), alone: events around it on the launch stream",
                      "kernel_ms": rescore_ms, "algorithmic_bytes_per_launch": fbytes,
                      "gathered_bytes_per_launch": gathered, "traffic": ft,
                      "traffic_unit": "HBM-side bytes per launch of k_rescore (rocprofv3 PMC, %s; measured_in_this_run: false)"
                                      % traffic.get("source"),
                      "hbm_algorithmic_frac": hbm_alg, "hbm_counter_frac": hbm_cnt, "l2_gather_frac": l2_frac,
                      "measured_gather_roof": groof,
                      "frac_of_measured_gather_roof": None if not groof else gathered / (rescore_ms * 1e-3) / (groof["TBps"] * 1e12),
                      "issue_busy": busy}
            if hbm_binds:
                roof_finish = dict(common, bound="hbm", achieved=fbytes / (rescore_ms * 1e-3) / 1e9, peak=PEAK_HBM / 1e9,
                                   unit="GB/s", frac=hbm_alg)
            else:
                roof_finish = dict(common, bound="l2", achieved=gathered / (rescore_ms * 1e-3) / 1e9, peak=PEAK_L2 / 1e9,
                                   unit="GB/s", frac=l2_frac,
                                   binding="the candidate rows are gathered out of L2 / Infinity Cache (neighbouring "
                                           "targets share candidates), not HBM: hbm_algorithmic_frac prices the "
                                           "SURVEY 8(d) bytes against the HBM peak as the contract words it and may "
                                           "exceed 1; `frac` is the gathered bytes over the aggregate L2 bandwidth; "
                                           "issue_busy holds the committed VALU / LDS busy fractions -- the kernel is "
                                           "issue and gather-latency bound")
            rescore_stage = {"what": "k_pick (k-th key, bound, certificate, compaction) + k_rescore; the exact-path "
                                     "launches come after", "stage_ms": finish_ms, "k_pick_ms": pick_ms,
                             "k_rescore_ms": rescore_ms,
                             "l2_gather_frac_of_stage": None if not finish_ms else gathered / (finish_ms * 1e-3) / PEAK_L2}
        # `roofline` is the single kernel that takes longest per step (k_gram_glds against k_rescore)
        if roof_finish and rescore_ms > gram_ms:
            dominant, other = roof_finish, roof_gram
        else:
            dominant, other = roof_gram, roof_finish
        shard = {"rows": "row bands (all-gather only)", "tiles": "symmetric tiles (threshold all-gather, list "
                 "all-to-all, result all-gather)"}.get(job.mode, "single rank, symmetric tiles")
        out = {
            "metric": "newref bin-pair distances/sec",
            "value": value,
            "unit": "bin-pair distances/s",
            "n_gpus": world,
            "steps": args.steps,
            "warmup": args.warmup,
            "ms_per_step": ms_per_step,
            "higher_is_better": True,
            "scaling": "strong",
            "vs_baseline": None,
            # every delivered index and distance is decided in float64 (numpy's bits); the matrix
            # cores only bound the distances to pick the candidates that get re-scored
            "dtype": "f64",
            "dtype_detail": "f16 MFMA (one product per multiply, f32 accumulate, per-row representation error in the "
                            "bounds) distance bounds + f64 exact re-score",
            "timing": "value / ms_per_step: %d plain passes between two synchronizes; stages_ms and the roofline "
                      "kernel times: %d passes with events between the stages, run BEFORE the %d warm-up passes and the timed "
                      "region (they are load like any other: the clocks settle over the first ~15 ms)" % (args.steps, args.steps, args.warmup),
            "data": "synthetic",
            "config": {"workload": "%s: newref %d samples x %d kb bins (%d masked bins, refsize %d), "
                                   "then batched test of %d samples/GPU at the same bin size"
                                   % (args.workload, S, binsize // 1000, B, k, args.test_samples),
                       "parallelism": "newref sharded by %s + sample-sharded test, %d rank(s)" % (shard, world),
                       "world_size": dist.get_world_size() if world > 1 else 1,
                       "shard_mode": job.mode or "single", "shard_calibration_s": job.calibration},
            "stages_ms": stages,
            "multi_rank": None if world == 1 else {
                "what": "diagnostic passes of the same job with events between the stages and every collective waited for "
                        "where it is issued (its own duration incl. the wait for the slowest peer); the timed passes "
                        "behind `value` keep the collectives in flight beside the kernels (row bands, async_op)",
                "row_bands_per_rank": getattr(job, "n_bands", None), "per_rank": per_rank},
            "prep": {"what": "newrefprep numerics on the GPU (normalise, mask, float64 MFMA Gram, the leading eigenpairs by %s, "
                             "components, correctedData left in HBM), %d samples, dense int32 host counts in"
                             % ("csrc/eigh.hip (tridiagonalisation + Sturm multisection + inverse iteration)"
                                if wt._eig_on_gpu(S, 3) else "host LAPACK (fewer than %d samples)" % wt.EIG_ON_GPU_FROM, S),
                     "ms": inp.get("prep_ms")},
            "test": {"metric": "test samples/sec", "value": one_in_flight["value"], "unit": "samples/s",
                     "ms_per_batch": one_in_flight["ms_per_batch"],
                     "what": "value / ms_per_batch: ONE batch in flight (a lone wc_test_batch_dev call after the other); "
                             "`pipelined` beside it: several batches in flight (TestPipeline); the roofline and stage "
                             "times are those of a lone batch",
                     "one_batch_in_flight": one_in_flight, "pipelined": pipelined,
                     "samples_per_gpu": args.test_samples,
                     "single_sample_latency_ms": single_ms, "latency": latency, "whole_job_1000_samples": whole_job,
                     "roofline": test_roof,
                     "calls_found": n_calls},
            "roofline": dominant,
            "roofline_other": other,
            "rescore_stage": rescore_stage,
            "newref_stats": stats,
            "extra": extra,
            "cpu_baseline": cpu,
        }
        emit(out, args.detail)
    if world > 1:
        dist.barrier()
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
