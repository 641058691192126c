"""Two and three ranks sharing the one GPU of the test box (gloo for the collectives,
staged through host memory): the real HIP stages with the real exchange kernels
(export / import of candidate lists, threshold hand-over) must reproduce the single-rank
result bit for bit.  RCCL itself needs one GPU per rank and is exercised by bench.py."""
import os
import socket

import numpy as np
import pytest

from oracle import wc_oracle as wo

pytestmark = pytest.mark.gpu


def _free_port():
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    port = s.getsockname()[1]
    s.close()
    return port


RENDEZVOUS_FAILED = "RENDEZVOUS_FAILED"


def _init_group(dist, rank, world, port):
    """The gloo process group of a worker.  ONLY a failure in here (the store's port taken between the probe and the
    bind, a peer that did not connect in time on a busy box) is marked as retryable: the marker is what _spawn looks for."""
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    try:
        dist.init_process_group("gloo", rank=rank, world_size=world)
    except Exception as exc:
        raise RuntimeError("%s rank %d of %d on port %d: %s: %s" % (RENDEZVOUS_FAILED, rank, world, port,
                                                                     type(exc).__name__, exc)) from exc


def _spawn(fn, args_of_port, nprocs, log_dir=None):
    """mp.spawn; ONE more try on a fresh port if -- and only if -- the process group did not come up (the marker of
    _init_group in the failure text).  Anything a worker raises after the rendezvous (an assertion, a HIP error, a
    collective that fails) fails the test at once.  The first failure's full text is printed and kept in log_dir."""
    import torch.multiprocessing as mp
    try:
        mp.spawn(fn, args=args_of_port(_free_port()), nprocs=nprocs, join=True)
    except Exception as exc:
        text = "%s: %s" % (type(exc).__name__, exc)
        print("spawn failed:\n" + text)
        if log_dir is not None:
            with open(os.path.join(str(log_dir), "first_spawn_failure.txt"), "w") as fh:
                fh.write(text)
        if RENDEZVOUS_FAILED not in text:
            raise
        print("the process group did not come up; one retry on a fresh port")
        mp.spawn(fn, args=args_of_port(_free_port()), nprocs=nprocs, join=True)


def _worker(rank, world, port, path_in, out_dir, mode):
    import torch
    import torch.distributed as dist
    from wisecondor_amd import _lib
    from wisecondor_amd.distributed import NewrefJob
    _init_group(dist, rank, world, port)
    try:
        z = np.load(path_in)
        X = torch.from_numpy(np.ascontiguousarray(z["data"])).cuda()
        job = NewrefJob(_lib.context(0), X, z["bins"], int(z["k"]), int(z["order"]), rank=rank, world=world, dist=dist,
                        mode=mode)
        for _ in range(2):
            idx, dst = job.run()
        torch.cuda.synchronize()
        np.savez(os.path.join(out_dir, "rank%d.npz" % rank), idx=idx.cpu().numpy(), dst=dst.cpu().numpy())
    finally:
        dist.destroy_process_group()


@pytest.mark.parametrize("world,order,mode", [(2, 1, "tiles"), (3, 0, "tiles"), (2, 0, "rows"), (3, 1, "rows")])
def test_hip_multi_rank_on_one_gpu(tmp_path, world, order, mode):
    import torch.multiprocessing as mp
    from wisecondor_amd import synth
    data, bins, sums = synth.corrected_matrix(1000000, 40, seed=9)
    data[5] = data[900]                       # a tie across ranks' row ranges
    data[2000] *= 25.0                        # an outlier row that needs the exact fallback
    rng = np.random.RandomState(4)
    members = rng.permutation(data.shape[0])[:900]
    data[members] = data[members[0]]          # 900 identical rows: per-source exchange slots overflow -> fallback
    k = 100
    path_in = str(tmp_path / "in.npz")
    np.savez(path_in, data=data, bins=bins, k=k, order=order)
    mp.get_context("spawn")
    _spawn(_worker, lambda port: (world, port, path_in, str(tmp_path), mode), world, tmp_path)
    src = data if order == 0 else np.asfortranarray(data)
    with np.errstate(all="ignore"):
        want_i, want_d = wo.get_reference(src, bins, sums, k, 1, 1, fast=True)
    for r in range(world):
        got = np.load(str(tmp_path / ("rank%d.npz" % r)))
        assert np.array_equal(got["idx"], want_i), r
        assert np.array_equal(got["dst"], want_d), r


@pytest.mark.parametrize("world,mode", [(2, "tiles"), (3, "rows"), (3, "tiles")])
def test_hip_multi_rank_ragged_layout(tmp_path, world, mode):
    """A layout with one-bin chromosomes at both ends (their rows sum pairwise even in the
    Fortran-ordered file), an empty chromosome and row bands that cut through
    chromosomes; fewer candidates than k for some rows."""
    import torch.multiprocessing as mp
    rng = np.random.RandomState(8)
    bins = np.array([1, 130, 0, 97, 2, 1], dtype=np.int64)
    B = int(bins.sum())
    data = 1.0 + 0.03 * rng.standard_normal((B, 33))
    data[3] = data[200]
    k = 120
    path_in = str(tmp_path / "in.npz")
    np.savez(path_in, data=data, bins=bins, k=k, order=1)
    mp.get_context("spawn")
    _spawn(_worker, lambda port: (world, port, path_in, str(tmp_path), mode), world, tmp_path)
    with np.errstate(all="ignore"):
        want_i, want_d = wo.get_reference(np.asfortranarray(data), bins, np.cumsum(bins), k, 1, 1, fast=True)
    for r in range(world):
        got = np.load(str(tmp_path / ("rank%d.npz" % r)))
        assert np.array_equal(got["idx"], want_i), r
        assert np.array_equal(got["dst"].view(np.int64), np.asarray(want_d).view(np.int64)), r


def _worker8(rank, world, port, path_in, out_dir, mode, xcap):
    import hashlib
    import torch
    import torch.distributed as dist
    from wisecondor_amd import _lib
    from wisecondor_amd import wisetools as wt
    from wisecondor_amd.distributed import NewrefJob
    if xcap:
        os.environ["WC_EXCHANGE_CAP"] = str(xcap)
    _init_group(dist, rank, world, port)
    try:
        z = np.load(path_in)
        X = torch.from_numpy(np.ascontiguousarray(z["data"])).cuda()
        job = NewrefJob(_lib.context(0), X, z["bins"], int(z["k"]), int(z["order"]), rank=rank, world=world, dist=dist,
                        mode=mode)
        for _ in range(2):
            idx, dst = job.run()
        torch.cuda.synchronize()
        idx, dst = idx.cpu().numpy(), dst.cpu().numpy()
        over = 0
        if mode == "tiles":
            over = int(sum(int((job.recv_cnt[i][r] > job.cap_x).sum()) for i in range(job.n_bands) for r in range(world) if r != rank))
        out = dict(sha=hashlib.sha256(idx.tobytes() + dst.tobytes()).hexdigest(), overflow=over, cap_x=getattr(job, "cap_x", 0),
                   fallback=wt.newref_stats()["fallback_rows"])
        if rank == 0:
            out.update(idx=idx, dst=dst)
        np.savez(os.path.join(out_dir, "rank%d.npz" % rank), **out)
    finally:
        dist.destroy_process_group()


@pytest.mark.parametrize("mode,xcap", [("tiles", 0), ("tiles", 40), ("rows", 0)])
def test_eight_ranks_on_one_gpu(tmp_path, mode, xcap):
    """The world size the node run uses: 8 ranks (sharing the one GPU of the test box, gloo) at
    20 k bins x 200 samples in both shard modes -- exchange_capacity() at world 8 (192 slots per row
    and source), eight-way tile dealing, eight row bands -- must reproduce the single-rank result bit
    for bit, and a row sample of it the CPU oracle.  xcap = 40 shrinks the exchange slots so that
    hundreds of rows overflow them and take the import -> exact-fallback path."""
    import torch
    import torch.multiprocessing as mp
    from wisecondor_amd import _lib, synth
    from wisecondor_amd.distributed import NewrefJob, exchange_capacity
    assert exchange_capacity(1024, 8) == 192
    data, bins, sums = synth.corrected_matrix(140000, 200, seed=11)
    assert data.shape[0] >= 20000
    data[5] = data[9000]                      # a tie across ranks' row ranges
    data[12000] *= 25.0                       # an outlier row that needs the exact fallback
    k = 100
    order = 1
    path_in = str(tmp_path / "in.npz")
    np.savez(path_in, data=data, bins=bins, k=k, order=order)
    mp.get_context("spawn")
    _spawn(_worker8, lambda port: (8, port, path_in, str(tmp_path), mode, xcap), 8, tmp_path)
    one = NewrefJob(_lib.context(0), torch.from_numpy(data).cuda(), bins, k, order)
    idx1, dst1 = one.run()
    torch.cuda.synchronize()
    idx1, dst1 = idx1.cpu().numpy(), dst1.cpu().numpy()
    got = [np.load(str(tmp_path / ("rank%d.npz" % r))) for r in range(8)]
    assert len({str(g["sha"]) for g in got}) == 1                      # every rank ends with the same full result
    assert np.array_equal(got[0]["idx"], idx1) and np.array_equal(got[0]["dst"].view(np.int64), dst1.view(np.int64))
    if xcap:
        assert sum(int(g["overflow"]) for g in got) > 100, [int(g["overflow"]) for g in got]
        assert sum(int(g["fallback"]) for g in got) > 100
    # the oracle on a sample of rows (the reference's own loop, wisetools.py:298-325)
    F = np.asfortranarray(data)
    rows = np.r_[5, 9000, 12000, np.arange(17, data.shape[0], 1009)]
    off = np.concatenate([[0], sums])
    for r in rows:
        c = int(np.searchsorted(sums, r, side="right"))
        others = np.concatenate((F[:off[c]], F[off[c + 1]:]))
        with np.errstate(all="ignore"):
            wi, wd = wo.get_ref_for_bins(k, int(r), int(r) + 1, F, others)
        assert np.array_equal(idx1[r], wi[0]) and np.array_equal(dst1[r], wd[0]), r


def _worker_rccl(rank, world, port, path_in, out_dir):
    import torch
    import torch.distributed as dist
    from wisecondor_amd import _lib
    from wisecondor_amd.distributed import NewrefJob
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    os.environ.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")
    torch.cuda.set_device(0)
    dist.init_process_group("nccl", rank=rank, world_size=world, device_id=torch.device("cuda", 0))
    try:
        z = np.load(path_in)
        X = torch.from_numpy(np.ascontiguousarray(z["data"])).cuda()
        out = {}
        for mode in ("tiles", "rows", None):
            job = NewrefJob(_lib.context(0), X, z["bins"], int(z["k"]), int(z["order"]), rank=rank, world=world, dist=dist,
                            mode=mode, passes=3, collectives=True)
            for _ in range(2):
                idx, dst = job.run()
            torch.cuda.synchronize()
            out["idx_%s" % mode] = idx.cpu().numpy()
            out["dst_%s" % mode] = dst.cpu().numpy()
            out["mode_%s" % mode] = job.mode
        np.savez(os.path.join(out_dir, "rccl.npz"), **out)
    finally:
        dist.destroy_process_group()


def test_rccl_collectives_in_a_world_of_one(tmp_path):
    """The RCCL calls themselves (backend "nccl"): the box has one GPU, so the world has one rank, but the job
    takes the multi-rank route -- threshold all-gather on a float view, the list all-to-all and the in-place
    result all-gather on BYTE views, the MAX / MIN all-reduces of the calibration on device tensors -- exactly
    the calls and dtypes the 8-GPU run makes.  Results equal the plain single-rank pass bit for bit."""
    import torch
    import torch.multiprocessing as mp
    from wisecondor_amd import _lib, synth
    from wisecondor_amd.distributed import NewrefJob
    data, bins, sums = synth.corrected_matrix(1000000, 40, seed=21)
    data[7] = data[1500]
    path_in = str(tmp_path / "in.npz")
    np.savez(path_in, data=data, bins=bins, k=100, order=1)
    mp.get_context("spawn")
    mp.spawn(_worker_rccl, args=(1, _free_port(), path_in, str(tmp_path)), nprocs=1, join=True)
    got = np.load(str(tmp_path / "rccl.npz"), allow_pickle=True)
    one = NewrefJob(_lib.context(0), torch.from_numpy(data).cuda(), bins, 100, 1)
    idx1, dst1 = one.run()
    torch.cuda.synchronize()
    idx1, dst1 = idx1.cpu().numpy(), dst1.cpu().numpy()
    for mode in ("tiles", "rows", "None"):
        assert np.array_equal(got["idx_%s" % mode], idx1), mode
        assert np.array_equal(got["dst_%s" % mode].view(np.int64), dst1.view(np.int64)), mode
    assert str(got["mode_None"]) in ("tiles", "rows")          # the calibration ran over RCCL and decided


def test_world_of_three_emulated_in_this_process():
    """Every rank's share of a three-rank job run one after the other in THIS process (bench.py's
    emulate_world_newref: thresholds of the band, the tile deal, export of the lists for foreign rows, import at
    the owner, finish of the band; and the row-band mode): both modes must reproduce the single-rank pass -- the
    same exchange kernels as the multi-process tests above, visible to a profiler of this process -- and the launch
    floor measurement the latency object of the bench line uses."""
    import sys
    import torch
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    if root not in sys.path:
        sys.path.insert(0, root)
    import bench
    from wisecondor_amd import _lib, synth, distributed
    data, bins, sums = synth.corrected_matrix(1000000, 16, seed=3)
    X = torch.from_numpy(data).cuda()
    ctx = _lib.context(0)
    job = distributed.NewrefJob(ctx, X, bins, 100, _lib.SUM_PAIRWISE)
    idx, dst = job.run()
    torch.cuda.synchronize()
    em = bench.emulate_world_newref(ctx, X, bins, 100, _lib.SUM_PAIRWISE, 3, idx, dst)
    for mode in ("tiles", "rows"):
        assert em[mode]["results_equal_single_rank"] is True, mode
        assert len(em[mode]["per_rank_ms"]) == 3
    floor = np.zeros(2)
    _lib.check(_lib.load().wc_launch_floor_us(ctx, torch.cuda.current_stream().cuda_stream, 8, 10, _lib.ptr(floor)))
    assert 0.0 < floor[0] < 1e4
