#!/usr/bin/env python3
"""Benchmark of the WISECONDOR hot path on MI355X.

    python bench.py --gpus N --steps K --warmup W

One step = one pass of the `newref` reference-bin selection over the whole
workload (BASELINE.json config 2 by default: 100 samples x 250 kb bins) with
the corrected matrix already resident in HBM, followed (timed separately) by
one batched `test` pass (PCA-apply -> 5 masked z-score repeats -> Stouffer
segmentation) over --test-samples samples per GPU at the same bin size.

Rank 0 prints ONE JSON line: `value` is the newref metric of BASELINE.json
(ordered cross-chromosome bin-pair distances per second, whole job), the `test`
object carries samples/s, `roofline` describes the dominant kernel (the
symmetric distance-tile kernel: on the bf16 matrix cores with hi/lo operand pairs by
default, with the float32-matrix-core variant timed beside it) and `cpu_baseline`
the CPU oracle timed on this box (rank 0, N=1 only).  Inputs are synthetic (seeded), see
wisecondor_amd/synth.py.
"""
import argparse
import json
import os
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)

WORKLOADS = {
    # name: (binsize, reference samples)
    "cfg1": (1000000, 16),
    "cfg2": (250000, 100),
    "cfg4": (50000, 600),
}
PEAK_FP32_MFMA = 157.3e12  # MI355X_MICROARCH.md: v_mfma_f32_32x32x2_f32 dense peak
PEAK_BF16_MFMA = 2500e12   # MI355X_MICROARCH.md: v_mfma_f32_32x32x16_bf16 dense peak (no sparsity)
SPLIT_PRODUCTS = 3.0       # k_gram<split>: hi.hi + hi.lo + lo.hi per float32 multiply


def build_inputs(binsize, n_ref, n_test, seed0=0):
    """Pipeline-level synthetic inputs: reference samples -> prep (host, untimed) and test samples."""
    from wisecondor_amd import synth
    from wisecondor_amd import wisetools as wt
    import contextlib
    import io
    profile = synth.bin_profile(binsize)
    samples = [synth.make_sample(profile, seed=seed0 + i) for i in range(n_ref)]
    with contextlib.redirect_stdout(io.StringIO()):
        masked, chrom_bins, mask = wt.toNumpyArray(samples)
        corrected, pca = wt.trainPCA(masked)
    offs = np.concatenate([[0], np.cumsum(chrom_bins)])
    masked_bins = np.array([int(mask[offs[i]:offs[i + 1]].sum()) for i in range(22)], dtype=np.int64)
    rng = np.random.RandomState(4242)
    tests = []
    for i in range(n_test):
        events = []
        if rng.rand() < 0.05:  # 5 % of the test samples carry a 1-5 % gain/loss
            c = int(rng.randint(1, 23))
            n = len(profile[c - 1])
            a = int(rng.randint(0, max(1, n - n // 4)))
            f = 1.0 + rng.choice([-1, 1]) * rng.uniform(0.01, 0.05)
            events.append((str(c), a, a + n // 4, f))
        tests.append(synth.make_sample(profile, seed=1000 + i, events=events))
    return dict(corrected=corrected, chrom_bins=np.asarray(chrom_bins, dtype=np.int64), mask=mask,
                masked_bins=masked_bins, pca_mean=pca.mean_, pca_components=pca.components_,
                tests=tests)


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
    args = ap.parse_args()

    import torch
    import torch.distributed as dist
    from wisecondor_amd import _lib, distributed
    from wisecondor_amd import wisetools as wt
    from wisecondor_amd.wisecondor import zThreshold

    world = int(os.environ.get("WORLD_SIZE", "1"))
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
    dev = torch.device("cuda", local_rank)
    torch.cuda.set_device(dev)
    lib = _lib.load()
    ctx = _lib.context(local_rank)

    binsize, n_ref = WORKLOADS[args.workload]
    k = args.refsize
    inp = build_inputs(binsize, n_ref, args.test_samples, seed0=0)
    corrected = inp["corrected"]                       # Fortran-ordered, like the reference's prep file
    order = wt.sum_order_of(corrected)
    B, S = corrected.shape
    bins = np.ascontiguousarray(inp["masked_bins"])
    pairs = float(B) * B - float((bins.astype(np.float64) ** 2).sum())   # ordered cross-chromosome pairs
    X = torch.from_numpy(np.ascontiguousarray(corrected)).to(dev)

    job = distributed.NewrefJob(ctx, X, bins, k, order, rank=rank, world=world)
    stream = torch.cuda.current_stream()

    def sync_all():
        torch.cuda.synchronize()
        if world > 1:
            dist.barrier()
            torch.cuda.synchronize()

    # ------------------------------------------------------------ newref ----
    ev = [torch.cuda.Event(enable_timing=True) for _ in range(3 * args.steps)]
    for _ in range(args.warmup):
        idx, dst = job.run()
    sync_all()
    t0 = time.perf_counter()
    for s in range(args.steps):
        idx, dst = job.run(collect_events=(ev[3 * s], ev[3 * s + 1], ev[3 * s + 2]))
    sync_all()
    t_newref = time.perf_counter() - t0
    kernel_ms = float(np.mean([ev[3 * s].elapsed_time(ev[3 * s + 1]) for s in range(args.steps)]))
    finish_ms = None
    if world == 1 or job.mode == "rows":      # tiles mode: the exchange sits between collect and finish
        finish_ms = float(np.mean([ev[3 * s + 1].elapsed_time(ev[3 * s + 2]) for s in range(args.steps)]))
    stats = wt.newref_stats(local_rank)
    split = os.environ.get("WC_GRAM_MODE", "") != "f32"
    f32_kernel_ms = None
    if split and world == 1:
        # the same pass with the distance tiles on the float32 matrix cores (north-star wording),
        # for the record: a few steps, kernel time only
        os.environ["WC_GRAM_MODE"] = "f32"
        fev = [torch.cuda.Event(enable_timing=True) for _ in range(8)]
        job.run()
        for s4 in range(4):
            job.run(collect_events=(fev[2 * s4], fev[2 * s4 + 1]))
        torch.cuda.synchronize()
        f32_kernel_ms = float(np.mean([fev[2 * s4].elapsed_time(fev[2 * s4 + 1]) for s4 in range(4)]))
        del os.environ["WC_GRAM_MODE"]
        idx, dst = job.run()                      # leave the context in the default mode
        torch.cuda.synchronize()
    tdev = dev if args.backend == "nccl" else torch.device("cpu")
    tmax = torch.tensor([t_newref], device=tdev, dtype=torch.float64)
    if world > 1:
        dist.all_reduce(tmax, op=dist.ReduceOp.MAX)
    t_newref = float(tmax.item())
    ms_per_step = 1e3 * t_newref / args.steps
    value = pairs * args.steps / t_newref

    # -------------------------------------------------------------- test ----
    reference = wt.Reference(idx.cpu().numpy(), dst.cpu().numpy(), inp["chrom_bins"], inp["masked_bins"],
                             inp["mask"], inp["pca_mean"], inp["pca_components"], binsize=binsize,
                             device=local_rank)
    thr = float(zThreshold([int(v) for v in inp["masked_bins"]], 1000, None))
    counts_h = wt.samples_to_counts(inp["tests"], inp["chrom_bins"])
    tb = distributed.TestBatch(reference, torch.from_numpy(counts_h).to(dev), thr, max_calls=256)
    for _ in range(max(1, args.warmup // 2)):
        tb.run()
    sync_all()
    t0 = time.perf_counter()
    test_steps = max(1, args.steps // 4)
    for _ in range(test_steps):
        tb.run()
    sync_all()
    t_test = time.perf_counter() - t0
    tmax = torch.tensor([t_test], device=tdev, dtype=torch.float64)
    if world > 1:
        dist.all_reduce(tmax, op=dist.ReduceOp.MAX)
    t_test = float(tmax.item())
    samples_per_s = world * args.test_samples * test_steps / t_test
    n_calls = int(tb.n_calls.sum().item())
    # SURVEY.md 8(d) traffic model of the test path, per sample: the z-score gathers
    # (repeats x sum_i n_i x (4 B index + 8 B value)) plus one float64 per Stouffer window (the
    # materialised-triangle model; the kernels never write the triangle, so this can exceed
    # what actually moves -- windows/s is reported beside it)
    n_refs = float((reference.distances < reference.cutoff).sum())
    windows = float(sum(int(n) * (int(n) + 1) // 2 for n in inp["masked_bins"]))
    test_bytes = 5.0 * n_refs * 12.0 + windows * 8.0
    # BASELINE config 3: one sample per call (latency mode, nothing amortised over a batch)
    tb1 = distributed.TestBatch(reference, torch.from_numpy(counts_h[:1].copy()).to(dev), thr, max_calls=256)
    tb1.run()
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(10):
        tb1.run()
    torch.cuda.synchronize()
    single_ms = 1e3 * (time.perf_counter() - t0) / 10

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
            xjob = distributed.NewrefJob(ctx, XX, xbins, k, _lib.SUM_SEQUENTIAL, rank=rank, world=world)
            xsteps = 3
            xev = [torch.cuda.Event(enable_timing=True) for _ in range(2 * xsteps)]
            xjob.run()
            sync_all()
            t0 = time.perf_counter()
            for q in range(xsteps):
                xjob.run(collect_events=(xev[2 * q], xev[2 * q + 1]))
            sync_all()
            xt = time.perf_counter() - t0
            tmax = torch.tensor([xt], device=tdev, dtype=torch.float64)
            if world > 1:
                dist.all_reduce(tmax, op=dist.ReduceOp.MAX)
            xt = float(tmax.item())
            xk_ms = float(np.mean([xev[2 * q].elapsed_time(xev[2 * q + 1]) for q in range(xsteps)]))
            # tiles mode: every unordered pair once on the node; rows mode: every ordered pair
            xflops = (xpairs if xjob.mode == "rows" else xpairs / 2.0) * 2.0 * xs / world
            extra = {"workload": "cfg4: newref %d samples x %d kb bins (%d bins), kernel-level synthetic matrix"
                                 % (xs, xb // 1000, XB),
                     "value": xpairs * xsteps / xt, "unit": "bin-pair distances/s", "ms_per_step": 1e3 * xt / xsteps,
                     "steps": xsteps, "shard_mode": xjob.mode or "tiles", "k_gram_ms": xk_ms,
                     "k_gram_algorithmic_tflops": xflops / (xk_ms * 1e-3) / 1e12}
            if os.environ.get("WC_GRAM_MODE", "") != "f32":
                extra["k_gram_executed_tflops"] = SPLIT_PRODUCTS * xflops / (xk_ms * 1e-3) / 1e12
                extra["k_gram_frac_of_bf16_mfma_peak"] = SPLIT_PRODUCTS * xflops / (xk_ms * 1e-3) / PEAK_BF16_MFMA
            else:
                extra["k_gram_frac_of_fp32_mfma_peak"] = xflops / (xk_ms * 1e-3) / PEAK_FP32_MFMA
            del xjob, XX
            # the main job's context state was replaced by the extra run; nothing below needs it
        except Exception as exc:      # the headline line must still be printed
            extra = {"workload": "cfg4", "error": "%s: %s" % (type(exc).__name__, exc)}

    # ------------------------------------------------------- cpu baseline ----
    cpu = None
    if rank == 0 and world == 1 and not args.no_cpu_baseline:
        from oracle import wc_oracle as wo
        import contextlib
        import io
        sums = np.cumsum(bins)
        parts = max(1, int(round(B * float(B) * S / 1.2e10)))      # ~10-20 s of numpy work per slice
        lo, hi = wo.get_part(0, parts, B)
        t0 = time.perf_counter()
        with contextlib.redirect_stdout(io.StringIO()), np.errstate(all="ignore"):
            ci, cd = wo.get_reference(corrected, bins, sums, k, 1, parts)
        t_cpu = time.perf_counter() - t0
        ok = bool(np.array_equal(ci, idx[lo:hi].cpu().numpy()))
        slice_pairs = float(sum(B - bins[np.searchsorted(sums, r, side="right")] for r in range(lo, hi)))
        cpu = {"value": slice_pairs / t_cpu, "unit": "bin-pair distances/s", "cores": 1, "kind": "port",
               "sample": "oracle get_reference (numpy distance + Python insertion top-k, the reference's own "
                         "structure) on target rows [%d,%d) of %d x all candidates, %.1f s" % (lo, hi, B, t_cpu),
               "matches_gpu_indices": ok}
        # the reference's `test` on ONE of the batch's samples (fillTri dominates: one np.sum per window)
        ref_dict = dict(binsize=np.float64(binsize), indexes=reference.indexes, distances=reference.distances,
                        chromosome_sizes=inp["chrom_bins"], mask=inp["mask"], masked_sizes=inp["masked_bins"],
                        pca_mean=inp["pca_mean"], pca_components=inp["pca_components"])
        t0 = time.perf_counter()
        with np.errstate(all="ignore"):
            o = wo.test_sample(inp["tests"][0], binsize, ref_dict)
        t_cpu_test = time.perf_counter() - t0
        nc = int(tb.n_calls[0].item())
        gpu_calls = tb.calls[0, :nc].cpu().numpy()
        cpu_calls = np.asarray(o["results_calls"], dtype=np.float64).reshape(-1, 5)
        cpu["test"] = {"value": 1.0 / t_cpu_test, "unit": "samples/s", "cores": 1, "kind": "port",
                       "sample": "oracle test_sample on 1 of the batch's samples, %.1f s" % t_cpu_test,
                       "matches_gpu_calls": bool(cpu_calls.shape == gpu_calls.shape and
                                                 np.array_equal(cpu_calls[:, :3], gpu_calls[:, :3]))}

    if rank == 0:
        traffic = None
        finish_traffic = None
        tpath = os.path.join(ROOT, "profiles", "r01_traffic.json")
        if world == 1 and os.path.exists(tpath):
            t = json.load(open(tpath)).get(args.workload)
            if t:   # HBM-side bytes per launch from the committed PMC passes of this same workload
                traffic = 1024.0 * (2.0 * t["fetch_kb_per_launch"] + t["write_kb_per_launch"])
                tf = t.get("k_finish") or {}
                if tf.get("fetch_kb_per_launch") is not None:
                    finish_traffic = 1024.0 * (2.0 * tf["fetch_kb_per_launch"] + tf["write_kb_per_launch"])
        # algorithmic work of the dominant kernel: one multiply-add per sample per unordered pair
        flops = (pairs if job.mode == "rows" else pairs / 2.0) * 2.0 * S / world   # rows mode: ordered pairs
        if split:
            # the tiles run on the bf16 matrix cores: three bf16 products stand for one float32
            # multiply, so the executed work is 3x the algorithmic work and the peak is the bf16 one
            achieved = SPLIT_PRODUCTS * flops / (kernel_ms * 1e-3)
            roof = {"kernel": "k_gram<split> (symmetric distance tiles on the bf16 matrix cores, hi/lo operand "
                              "pairs, + candidate filter)",
                    "bound": "mfma", "achieved": achieved / 1e12, "peak": PEAK_BF16_MFMA / 1e12, "unit": "TFLOP/s",
                    "frac": achieved / PEAK_BF16_MFMA, "traffic": traffic,
                    "traffic_unit": "bytes per launch (rocprofv3 PMC, profiles/r01_traffic.json)",
                    "kernel_ms": kernel_ms, "algorithmic_flop_per_launch": flops,
                    "executed_flop_per_launch": SPLIT_PRODUCTS * flops,
                    "algorithmic_tflops": flops / (kernel_ms * 1e-3) / 1e12,
                    "algorithmic_rate_over_fp32_mfma_peak": flops / (kernel_ms * 1e-3) / PEAK_FP32_MFMA,
                    "fp32_mfma_variant": None if f32_kernel_ms is None else {
                        "kernel_ms": f32_kernel_ms, "achieved": flops / (f32_kernel_ms * 1e-3) / 1e12,
                        "peak": PEAK_FP32_MFMA / 1e12, "frac": flops / (f32_kernel_ms * 1e-3) / PEAK_FP32_MFMA,
                        "note": "WC_GRAM_MODE=f32: same tiles with v_mfma_f32_32x32x2_f32"}}
        else:
            achieved = flops / (kernel_ms * 1e-3)
            roof = {"kernel": "k_gram (symmetric fp32 MFMA distance tiles + candidate filter)",
                    "bound": "mfma", "achieved": achieved / 1e12, "peak": PEAK_FP32_MFMA / 1e12, "unit": "TFLOP/s",
                    "frac": achieved / PEAK_FP32_MFMA, "traffic": traffic,
                    "traffic_unit": "bytes per launch (rocprofv3 PMC, profiles/r01_traffic.json)",
                    "kernel_ms": kernel_ms, "algorithmic_flop_per_launch": flops}
        # the re-score stage (B * k * S * 8 B read, B * k * 12 B written), bound by memory
        roof_finish = None
        if finish_ms:
            rows_here = B / world
            fbytes = rows_here * k * S * 8.0 + rows_here * k * 12.0
            roof_finish = {"kernel": "k_finish (per row: k-th key, candidate compaction, float64 re-score in numpy "
                                     "order, counting order; events around this kernel alone, the exact-path launches come after)",
                           "bound": "hbm", "achieved": fbytes / (finish_ms * 1e-3) / 1e9, "peak": 8000.0,
                           "unit": "GB/s", "frac": fbytes / (finish_ms * 1e-3) / 8.0e12, "traffic": finish_traffic,
                           "traffic_unit": "bytes per launch (rocprofv3 PMC, profiles/r01_traffic.json): the "
                                           "gathers are served by L2 / Infinity Cache, a fraction reaches HBM",
                           "kernel_ms": finish_ms, "algorithmic_bytes_per_launch": fbytes}
        # `roofline` is the kernel that takes longer per step
        if roof_finish and finish_ms > kernel_ms:
            dominant, other = roof_finish, roof
        else:
            dominant, other = roof, roof_finish
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
            "dtype_detail": ("bf16x3 MFMA (hi/lo pairs, f32 accumulate) distance bounds + f64 exact re-score" if split
                             else "f32 MFMA distance bounds + f64 exact re-score"),
            "data": "synthetic",
            "config": {"workload": "%s: newref %d samples x %d kb bins (%d masked bins, refsize %d), "
                                   "then batched test of %d samples/GPU at the same bin size"
                                   % (args.workload, S, binsize // 1000, B, k, args.test_samples),
                       "parallelism": "newref sharded by %s + sample-sharded test, %d rank(s)"
                                      % ({"rows": "row bands (all-gather only)", "tiles": "symmetric tiles "
                                          "(threshold all-gather, list all-to-all, result all-gather)"}
                                         .get(job.mode or "tiles"), world)},
            "test": {"metric": "test samples/sec", "value": samples_per_s, "unit": "samples/s",
                     "ms_per_batch": 1e3 * t_test / test_steps, "samples_per_gpu": args.test_samples,
                     "single_sample_latency_ms": single_ms,
                     "roofline": {"bound": "hbm", "model": "5 repeats x gathered refs x 12 B + 8 B per Stouffer window "
                                  "(SURVEY.md 8d); triangle never materialised",
                                  "bytes_per_sample": test_bytes, "achieved": samples_per_s / world * test_bytes / 1e9,
                                  "peak": 8000.0, "unit": "GB/s",
                                  "frac": samples_per_s / world * test_bytes / 8.0e12,
                                  "windows_per_s": samples_per_s * windows},
                     "calls_found": n_calls},
            "roofline": dominant,
            "roofline_other": other,
            "newref_stats": stats,
            "extra": extra,
            "cpu_baseline": cpu,
        }
        print(json.dumps(out))
    if world > 1:
        dist.barrier()
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
