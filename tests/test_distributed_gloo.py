"""world_size-2 (and 3) gloo runs of the multi-GPU newref choreography on CPU.

The HIP stages cannot run here, so the driver (wisecondor_amd.distributed.NewrefJob:
threshold all-gather, round-robin tile deal, candidate-list all-to-all, owner-side
finish, result all-gather; and the row-band mode with the all-gather only) is exercised with a numpy stand-in for the four stages
that follows the same contract (test infrastructure only).  The multi-rank result
must equal the single-rank result and the oracle, bit for bit.
"""
import os
import socket

import numpy as np
import pytest
import torch
import torch.distributed as dist
import torch.multiprocessing as mp

from oracle import wc_oracle as wo
from wisecondor_amd.distributed import NewrefJob

TILE = 8


class NumpyStages(object):
    """Same contract as HipStages, exact float64 distances as keys."""

    def __init__(self, X, bins, k):
        self.X = np.asarray(X)
        self.bins = np.asarray(bins, dtype=np.int64)
        self.k = k
        self.n_bins, self.n_samples = self.X.shape
        self.off = np.concatenate([[0], np.cumsum(self.bins)])
        self.chrom = np.repeat(np.arange(len(bins)), self.bins)

    def prepare(self):
        self.cap = 64
        self.thr = np.full(self.n_bins, -np.inf, dtype=np.float32)
        self.lists = [[] for _ in range(self.n_bins)]

    def empty(self, shape, dtype):
        return torch.empty(shape, dtype=dtype)

    def thresholds(self, rb, re):
        # owner-side rule that depends on the row only: the distance to the 30th nearest candidate
        for i in range(rb, re):
            d = self._dist_row(i)
            self.thr[i] = np.float32(np.sort(d[np.isfinite(d)])[min(29, np.isfinite(d).sum() - 1)] * 1.0001)

    def _dist_row(self, i):
        d = np.sum(np.power(self.X - self.X[i], 2), 1)
        d[self.chrom == self.chrom[i]] = np.inf
        return d

    def get_thr(self, rb, re, out):
        out[:re - rb] = torch.from_numpy(self.thr[rb:re].copy())

    def set_thr(self, rb, re, src):
        self.thr[rb:re] = src[:re - rb].numpy()

    def collect(self, rb, re, rank, ranks):
        nb = (self.n_bins + TILE - 1) // TILE
        serial = 0
        for I in range(nb):
            for J in range(I, nb):
                mine = serial % ranks == rank
                serial += 1
                if not mine:
                    continue
                for i in range(I * TILE, min((I + 1) * TILE, self.n_bins)):
                    for j in range(J * TILE, min((J + 1) * TILE, self.n_bins)):
                        if self.chrom[i] == self.chrom[j] or (I == J and j <= i):
                            continue
                        d = float(np.sum(np.power(self.X[j] - self.X[i], 2)))
                        if d <= self.thr[i] and rb <= i < re:
                            self.lists[i].append(j)
                        if d <= self.thr[j] and rb <= j < re:
                            self.lists[j].append(i)

    def export(self, rb, re, cap, cnt, lst):
        for r in range(rb, re):
            n = len(self.lists[r])
            cnt[r - rb] = n
            lst[r - rb, :min(n, cap)] = torch.tensor(self.lists[r][:cap], dtype=torch.int64)

    def import_(self, rb, re, cap, cnt, lst):
        for r in range(rb, re):
            n = int(cnt[r - rb])
            assert n <= cap
            self.lists[r].extend(int(v) for v in lst[r - rb, :n])

    def finish(self, rb, re, idx, dst):
        for r in range(rb, re):
            cand = np.array(sorted(set(self.lists[r])), dtype=np.int64)
            assert len(cand) == len(self.lists[r]), "duplicate candidate: a tile was processed twice"
            d = np.sum(np.power(self.X[cand] - self.X[r], 2), 1)
            order = np.argsort(d, kind="stable")[:self.k]
            c = self.chrom[r]
            lo, hi = self.off[c], self.off[c + 1]
            loc = np.where(cand[order] < lo, cand[order], cand[order] - (hi - lo))
            idx[r - rb] = torch.from_numpy(loc.astype(np.int32))
            dst[r - rb] = torch.from_numpy(d[order])


def _free_port():
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    port = s.getsockname()[1]
    s.close()
    return port


class FailingStages(NumpyStages):
    """A rank whose tile collection breaks (the stand-in for an out-of-memory or GPU fault on one rank)."""

    def collect(self, rb, re, rank, ranks):
        if ranks > 1:
            raise MemoryError("collect failed on this rank")
        return NumpyStages.collect(self, rb, re, rank, ranks)


def _worker(rank, world, port, X, bins, k, out_dir, mode, passes=1, fail_rank=-1, bands=None, gather=True):
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    try:
        st = (FailingStages if rank == fail_rank else NumpyStages)(X, bins, k)
        job = NewrefJob(None, None, bins, k, 0, rank=rank, world=world, stages=st, dist=dist, mode=mode, passes=passes,
                        bands=bands, gather=gather)
        try:
            for _ in range(2):          # second run reuses the exchange buffers
                idx, dst = job.run()
        except Exception as exc:
            if fail_rank < 0:
                raise
            np.savez(os.path.join(out_dir, "rank%d.npz" % rank), error=type(exc).__name__)
            return
        np.savez(os.path.join(out_dir, "rank%d.npz" % rank), idx=idx.numpy(), dst=dst.numpy(), mode=job.mode,
                 measured=sorted((job.calibration or {}).keys()), n_bands=job.n_bands)
    finally:
        dist.destroy_process_group()


def test_a_failure_on_one_rank_stops_the_calibration_on_all(tmp_path):
    """calibrate(): a real fault on one rank (not "this backend lacks the collective") must not be
    swallowed into the row shard, and must not leave the other ranks waiting in a collective: after
    each mode the ranks agree on the outcome, so all of them stop together."""
    rng = np.random.RandomState(5)
    bins = np.array([9, 14, 7], dtype=np.int64)
    X = 1.0 + 0.05 * rng.standard_normal((int(bins.sum()), 6))
    mp.spawn(_worker, args=(3, _free_port(), X, bins, 5, str(tmp_path), None, 3, 1), nprocs=3, join=True)
    errors = [str(np.load(os.path.join(str(tmp_path), "rank%d.npz" % r))["error"]) for r in range(3)]
    assert errors[1] == "MemoryError" and errors[0] == errors[2] == "PeerFailure", errors


def test_a_one_shot_job_does_not_calibrate(tmp_path):
    """passes=1 (the CLI): the symmetric tile shard without the four calibration passes."""
    rng = np.random.RandomState(6)
    bins = np.array([9, 14, 7], dtype=np.int64)
    X = 1.0 + 0.05 * rng.standard_normal((int(bins.sum()), 6))
    mp.spawn(_worker, args=(2, _free_port(), X, bins, 5, str(tmp_path), None, 1), nprocs=2, join=True)
    for r in range(2):
        got = np.load(os.path.join(str(tmp_path), "rank%d.npz" % r))
        assert str(got["mode"]) == "tiles" and list(got["measured"]) == []


@pytest.mark.parametrize("world,mode", [(2, "tiles"), (3, "tiles"), (2, "rows"), (3, None), (8, "tiles"), (8, "rows")])
def test_multi_rank_equals_single_rank_and_oracle(tmp_path, world, mode):
    rng = np.random.RandomState(5)
    bins = np.array([9, 14, 7, 12, 11], dtype=np.int64)
    X = 1.0 + 0.05 * rng.standard_normal((int(bins.sum()), 6))
    k = 10
    mp.spawn(_worker, args=(world, _free_port(), X, bins, k, str(tmp_path), mode, 3 if mode is None else 1),
             nprocs=world, join=True)
    st = NumpyStages(X, bins, k)
    one_i, one_d = NewrefJob(None, None, bins, k, 0, rank=0, world=1, stages=st).run()
    want_i, want_d = wo.get_reference(X, bins, np.cumsum(bins), k, 1, 1, fast=True)
    assert np.array_equal(one_i.numpy(), want_i) and np.array_equal(one_d.numpy(), want_d)
    modes = set()
    for r in range(world):
        got = np.load(os.path.join(str(tmp_path), "rank%d.npz" % r))
        assert np.array_equal(got["idx"], want_i), r
        assert np.array_equal(got["dst"], want_d), r
        modes.add(str(got["mode"]))
        if mode is None:        # the shard mode was measured (both modes timed), not assumed
            assert list(got["measured"]) == ["rows", "tiles"]
    assert len(modes) == 1 and modes <= {"tiles", "rows"}      # every rank reached the same decision


@pytest.mark.parametrize("world,mode,bands", [(2, "tiles", 1), (3, "tiles", 3), (3, "rows", 2), (8, "tiles", 7), (2, "rows", 64)])
def test_band_pipeline_any_band_count(tmp_path, world, mode, bands):
    """The multi-rank step is a pipeline over row bands with the collectives in flight (async_op): any band
    count -- one (a single exchange and gather), more bands than a rank has rows, bands that leave the shorter
    ranks' last band empty -- gives the single-rank result."""
    rng = np.random.RandomState(15)
    bins = np.array([9, 14, 7, 12, 11], dtype=np.int64)
    X = 1.0 + 0.05 * rng.standard_normal((int(bins.sum()), 6))
    k = 10
    mp.spawn(_worker, args=(world, _free_port(), X, bins, k, str(tmp_path), mode, 1, -1, bands), nprocs=world, join=True)
    want_i, want_d = wo.get_reference(X, bins, np.cumsum(bins), k, 1, 1, fast=True)
    for r in range(world):
        got = np.load(os.path.join(str(tmp_path), "rank%d.npz" % r))
        assert np.array_equal(got["idx"], want_i) and np.array_equal(got["dst"], want_d), r
        assert 1 <= int(got["n_bands"]) <= bands


@pytest.mark.parametrize("world,mode", [(2, "tiles"), (3, "rows"), (8, "tiles")])
def test_owners_keep_their_rows_without_the_result_gather(tmp_path, world, mode):
    """gather=False (the command line tool: a rank writes the part files of the rows it owns, the reference's
    workers exchange nothing but files, wisecondor.py:47-56): run() returns the rank's own rows, which stacked in
    rank order are the whole result."""
    from wisecondor_amd.distributed import row_range
    rng = np.random.RandomState(25)
    bins = np.array([9, 14, 7, 12, 11], dtype=np.int64)
    X = 1.0 + 0.05 * rng.standard_normal((int(bins.sum()), 6))
    k = 10
    mp.spawn(_worker, args=(world, _free_port(), X, bins, k, str(tmp_path), mode, 1, -1, None, False), nprocs=world, join=True)
    want_i, want_d = wo.get_reference(X, bins, np.cumsum(bins), k, 1, 1, fast=True)
    got = [np.load(os.path.join(str(tmp_path), "rank%d.npz" % r)) for r in range(world)]
    for r in range(world):
        b, e = row_range(r, world, X.shape[0])
        assert got[r]["idx"].shape == (e - b, k)
    assert np.array_equal(np.concatenate([g["idx"] for g in got]), want_i)
    assert np.array_equal(np.concatenate([g["dst"] for g in got]), want_d)
