"""One-process-per-GPU drivers of the hot path (torch.distributed over RCCL/xGMI).

newref  The reference parallelises `newref` by splitting the target rows into
        parts and gathering part files (wisecondor.py:47-61, 135-158).  Here the
        symmetric distance-tile space is dealt round-robin to the ranks, which
        halves the MFMA work compared with row parts; the price is one exchange
        step: every rank holds partial candidate lists for ALL rows, so the
        lists of each row range travel to the rank that owns it (all-to-all),
        the owner finishes its rows (float64 re-score, stable top-k) and the
        [rows, k] results are all-gathered.  Thresholds are computed per owner
        and all-gathered first.  Row ownership is the reference's getPart.
test    Samples are independent (one per invocation in the reference,
        wisecondor.py:193-199): the batch is sharded by sample, no collective.

torch is used for device memory, streams and the collectives only; all
arithmetic happens behind the C ABI (include/wisecondor_hip.h).
"""
import os

import numpy as np

from . import _lib


def row_range(rank, world, n_bins):
    """Rows owned by `rank`: getPart(rank, world, bins) (wisetools.py:358-361)."""
    per = n_bins / float(world)
    return int(per * rank), int(per * (rank + 1))


class HipStages(object):
    """The four newref stages of the C ABI on torch CUDA tensors / the current stream."""

    def __init__(self, ctx, X, chrom_bins, k, sum_order):
        import torch
        self.torch = torch
        self.lib = _lib.load()
        self.ctx = ctx
        self.X = X
        self.bins = np.ascontiguousarray(chrom_bins, dtype=np.int64)
        self.k = int(k)
        self.order = int(sum_order)
        self.n_bins, self.n_samples = int(X.shape[0]), int(X.shape[1])
        self.device = X.device

    def _stream(self):
        return self.torch.cuda.current_stream().cuda_stream

    def prepare(self):
        _lib.check(self.lib.wc_newref_prepare_dev(self.ctx, self._stream(), self.X.data_ptr(), self.n_bins,
                                                  self.n_samples, _lib.ptr(self.bins), len(self.bins), self.k,
                                                  self.order))
        self.cap = int(self.lib.wc_newref_list_capacity(self.ctx))

    def full_pass(self, idx, dst):
        """All four stages for every row in one C call (wc_get_reference_dev)."""
        _lib.check(self.lib.wc_get_reference_dev(self.ctx, self._stream(), self.X.data_ptr(), self.n_bins,
                                                 self.n_samples, _lib.ptr(self.bins), len(self.bins), self.k,
                                                 self.order, 0, self.n_bins, idx.data_ptr(), dst.data_ptr()))
        self.cap = int(self.lib.wc_newref_list_capacity(self.ctx))

    def thresholds(self, rb, re):
        _lib.check(self.lib.wc_newref_thresholds_dev(self.ctx, self._stream(), rb, re))

    def get_thr(self, rb, re, out):
        _lib.check(self.lib.wc_newref_get_thresholds_dev(self.ctx, self._stream(), rb, re, out.data_ptr()))

    def get_bounds(self, rb, re, lo, slack):
        _lib.check(self.lib.wc_newref_get_bounds_dev(self.ctx, self._stream(), rb, re, lo.data_ptr(), slack.data_ptr()))

    def set_thr(self, rb, re, src):
        _lib.check(self.lib.wc_newref_set_thresholds_dev(self.ctx, self._stream(), rb, re, src.data_ptr()))

    def collect(self, rb, re, tile_rank, tile_ranks):
        _lib.check(self.lib.wc_newref_collect_dev(self.ctx, self._stream(), rb, re, tile_rank, tile_ranks))

    def export(self, rb, re, cap, cnt, lst):
        _lib.check(self.lib.wc_newref_export_lists_dev(self.ctx, self._stream(), rb, re, cap, cnt.data_ptr(),
                                                       lst.data_ptr()))

    def import_(self, rb, re, cap, cnt, lst):
        _lib.check(self.lib.wc_newref_import_lists_dev(self.ctx, self._stream(), rb, re, cap, cnt.data_ptr(),
                                                       lst.data_ptr()))

    def finish(self, rb, re, idx, dst):
        _lib.check(self.lib.wc_newref_finish_dev(self.ctx, self._stream(), rb, re, idx.data_ptr(), dst.data_ptr()))

    def rescore(self, rb, re, idx, dst):
        """The per-row fast path alone (finish == rescore + fallback), for callers that time them apart."""
        _lib.check(self.lib.wc_newref_rescore_dev(self.ctx, self._stream(), rb, re, idx.data_ptr(), dst.data_ptr()))

    def pick(self, rb, re, idx, dst):
        _lib.check(self.lib.wc_newref_pick_dev(self.ctx, self._stream(), rb, re, idx.data_ptr(), dst.data_ptr()))

    def rescore_pairs(self, rb, re, idx, dst):
        _lib.check(self.lib.wc_newref_rescore_pairs_dev(self.ctx, self._stream(), rb, re, idx.data_ptr(),
                                                        dst.data_ptr()))

    def fallback(self, rb, re, idx, dst):
        _lib.check(self.lib.wc_newref_fallback_dev(self.ctx, self._stream(), rb, re, idx.data_ptr(), dst.data_ptr()))

    def exact(self, rb, re, idx, dst):
        """Every row of [rb, re) by the exact path (float64 distances to all candidates, stable selection) on a
        prepared job: what the full-size tests hold the fast path against."""
        _lib.check(self.lib.wc_newref_exact_dev(self.ctx, self._stream(), rb, re, idx.data_ptr(), dst.data_ptr()))

    def empty(self, shape, dtype):
        return self.torch.empty(shape, dtype=dtype, device=self.device)


def forced_shard_mode():
    """'tiles' / 'rows' when WC_NEWREF_SHARD pins the multi-GPU shard mode, else None
    ('measure' forces the calibration even for a job that runs once)."""
    env = os.environ.get("WC_NEWREF_SHARD", "auto")
    return env if env in ("tiles", "rows") else None


CALIBRATE_FROM_PASSES = 3      # a job that will run fewer passes is not worth four calibration passes


class PeerFailure(RuntimeError):
    """Another rank failed in a step all ranks take together; this rank stops with it."""


def _unsupported(exc):
    """An exchange collective this backend does not offer (argument checks fail on every rank alike,
    before any communication) -- as opposed to a real fault (out of memory, a GPU or RCCL error)."""
    text = str(exc).lower()
    return isinstance(exc, NotImplementedError) or (
        isinstance(exc, RuntimeError) and any(w in text for w in ("not support", "unsupported", "not implemented")))


def exchange_capacity(cap, world):
    """Per (source rank, row) slot count of the candidate exchange.

    A rank holds ~1/world of a row's candidates, whose total is ~cap/2 with a row-to-row
    spread of ~20 % (it comes from a sampled order statistic): 1.75 x mean + 8 sigma
    (Poisson) + slack, in steps of 32.  A row that still overflows is marked by the
    importer and takes the exact fallback."""
    forced = os.environ.get("WC_EXCHANGE_CAP")      # tests: a small value drives rows into the overflow -> exact path
    if forced:
        return int(min(cap, max(1, int(forced))))
    mean = cap / 2.0 / world
    return int(min(cap, 32 * int(np.ceil((1.75 * mean + 8.0 * np.sqrt(mean) + 16.0) / 32.0))))


def _align(n, a):
    return (n + a - 1) // a * a


class _Pending(object):
    """A collective in flight.  RCCL: the work runs on the communicator's stream, ordered after what the launch
    stream held when it was issued; wait() makes the launch stream wait for it (the host does not block).
    gloo with device tensors (CPU tests, ranks sharing a GPU): the tensors were staged to the host before the
    collective started; wait() blocks the host and copies the result back."""

    def __init__(self, work, after=None):
        self.work, self.after = work, after

    def wait(self):
        if self.work is not None:
            self.work.wait()
        if self.after is not None:
            self.after()
        self.work = self.after = None


def band_count(default=4):
    """Row bands per rank of the multi-rank pipeline (WC_NEWREF_BANDS; 1 = one exchange, one gather)."""
    try:
        return max(1, int(os.environ.get("WC_NEWREF_BANDS", default)))
    except ValueError:
        return default


class NewrefJob(object):
    """getReference for all rows, on `world` ranks.

    A rank's rows (the reference's getPart) are cut into `bands` row bands and the multi-rank step is a
    software pipeline over them, every collective issued with async_op=True so that it runs on the
    communicator's stream beside the launch stream's kernels:

      tiles  thresholds all-gather (small, blocking); the rank's share of the symmetric tile space; then per
             band: export of the band's foreign lists -> all-to-all (async) -- and, one band behind, import ->
             re-score of the band's own rows -> all-gather of the band's results (async).  The exchange of
             band i + 1 and the result gather of band i - 1 travel while band i is re-scored.
      rows   the rank's row band against all columns; per band: re-score -> all-gather (async).

    Counts + lists and indexes + distances share one byte buffer per band so that they travel in one
    collective.  gather=False (the command line tool: every rank writes the part files of the rows it owns,
    wisecondor.py:47-56 -- the reference's workers exchange nothing but files) skips the result gathers and
    run() returns the rank's own rows."""

    def __init__(self, ctx, X, chrom_bins, k, sum_order, rank=0, world=1, stages=None, dist=None, mode=None,
                 passes=1, collectives=False, bands=None, gather=True):
        """`passes`: how many times the caller expects to run() this job.  The shard mode of a
        multi-rank job is measured (calibrate: four extra passes) only from CALIBRATE_FROM_PASSES
        on or with WC_NEWREF_SHARD=measure; a one-shot job (the CLI) takes the symmetric tile shard."""
        import torch
        self.passes = int(passes)
        # collectives=True: take the multi-rank route (exchange buffers, every collective) even in a world of
        # one rank -- a one-GPU box can then run the RCCL calls themselves (tests/test_distributed_gpu.py)
        self.single = int(world) == 1 and not collectives
        self.torch = torch
        self.rank, self.world = int(rank), int(world)
        self.st = stages if stages is not None else HipStages(ctx, X, chrom_bins, k, sum_order)
        self.k = int(k)
        self.n_bins = int(self.st.n_bins)
        if dist is None and not self.single:
            import torch.distributed as dist
        self.dist = dist
        self.mode = mode            # None: measured at the first run (calibrate), unless WC_NEWREF_SHARD pins it
        self.gather = bool(gather)
        self.calibration = None
        self._checked = False       # calibration warm-up passes: ranks agree on every local step's outcome (_local)
        self._marks = None
        self.last_marks = None
        self.comm_log = None        # run(timing=True): one entry per collective {name, bytes, ms (blocking)}
        self.chrom_bins = np.asarray(chrom_bins, dtype=np.int64)
        self.ranges = [row_range(r, self.world, self.n_bins) for r in range(self.world)]
        self.max_rows = max(e - b for b, e in self.ranges)
        t = torch
        e = self.st.empty
        if self.single:
            self.idx = e((self.n_bins, self.k), t.int32)
            self.dst = e((self.n_bins, self.k), t.float64)
            return
        # row bands: band i of rank r = rows [b_r + i * band_rows, ...) clipped to the rank's range (the same band
        # length on every rank: equal blocks per collective; a rank with one row less has a shorter last band)
        want = band_count() if bands is None else max(1, int(bands))
        self.band_rows = max(1, -(-self.max_rows // want))
        self.n_bands = -(-self.max_rows // self.band_rows)
        n = self.band_rows * self.k
        # result slots per band: [dst f64 | idx i32] of band_rows rows -- `own` is what finish writes and the
        # all-gather sends, `res` [band, rank, bytes] what it delivers
        self.res_bytes = _align(n * 12, 16)
        self.own = e((self.n_bands, self.res_bytes), t.uint8)
        self.own_dst = [self.own[i, :n * 8].view(t.float64).view(self.band_rows, self.k) for i in range(self.n_bands)]
        self.own_idx = [self.own[i, n * 8:n * 12].view(t.int32).view(self.band_rows, self.k) for i in range(self.n_bands)]
        if self.gather:
            self.res = e((self.n_bands, self.world, self.res_bytes), t.uint8)
            self.res_dst = self.res[:, :, :n * 8].view(t.float64).view(self.n_bands, self.world, self.band_rows, self.k)
            self.res_idx = self.res[:, :, n * 8:n * 12].view(t.int32).view(self.n_bands, self.world, self.band_rows, self.k)
            # where row g of the [bins, k] result sits in `res`: (band, rank, row of the band)
            owner = np.zeros(self.n_bins, dtype=np.int64)
            local = np.zeros(self.n_bins, dtype=np.int64)
            for r, (b, en) in enumerate(self.ranges):
                owner[b:en] = r
                local[b:en] = np.arange(en - b)
            dev = self.own.device
            self._map = tuple(t.from_numpy(np.ascontiguousarray(a)).to(dev)
                              for a in (local // self.band_rows, owner, local % self.band_rows))
        self.thr_all = e((self.world, self.max_rows), t.float32)
        self.buffers_ready = False

    def band(self, r, i):
        """Global row range of band i of rank r (empty when the rank's range ends before it)."""
        b, e = self.ranges[r]
        lo = min(e, b + i * self.band_rows)
        return lo, min(e, lo + self.band_rows)

    # Collectives: RCCL moves device tensors directly; under gloo (CPU tests, or two ranks
    # sharing one GPU in tests) device tensors are staged through host memory.
    def _needs_staging(self, t):
        return self.dist.get_backend() == "gloo" and t.device.type != "cpu"

    def _timed_collective(self, name, nbytes, issue):
        """Issue a collective; in a timing run (run(timing=True)) it is waited for on the spot between two events
        on the launch stream, so that its own duration (with the wait for the slowest peer) is on record."""
        if self.comm_log is None or self._marks is None:
            return issue()
        ev = self.torch.cuda.Event
        cuda = getattr(self.st, "device", None) is not None and self.st.device.type == "cuda"
        a, b = (ev(enable_timing=True), ev(enable_timing=True)) if cuda else (None, None)
        import time
        t0 = time.perf_counter()
        if cuda:
            a.record()
        pending = issue()
        pending.wait()
        if cuda:
            b.record()
        self.comm_log.append({"name": name, "bytes": int(nbytes), "events": (a, b),
                              "host_ms": 1e3 * (time.perf_counter() - t0)})
        return _Pending(None)

    def _all_gather(self, out, own, name="all_gather"):
        """All-gather of `own` (every rank's block) into `out`, in flight when this returns."""
        def issue():
            if self._needs_staging(out):
                o = out.cpu()
                work = self.dist.all_gather_into_tensor(o, own.cpu(), async_op=True)
                return _Pending(work, lambda: out.copy_(o))
            return _Pending(self.dist.all_gather_into_tensor(out, own, async_op=True))
        return self._timed_collective(name, out.numel() * out.element_size(), issue)

    def _all_to_all(self, out, inp, name="all_to_all"):
        def issue():
            if self._needs_staging(inp):
                o = out.cpu()
                work = self.dist.all_to_all_single(o, inp.cpu(), async_op=True)
                return _Pending(work, lambda: out.copy_(o))
            return _Pending(self.dist.all_to_all_single(out, inp, async_op=True))
        return self._timed_collective(name, out.numel() * out.element_size(), issue)

    def _alloc_exchange(self):
        t = self.torch
        e = self.st.empty
        self.cap_x = exchange_capacity(self.st.cap, self.world)
        # per band and destination rank: [counts i32 x band_rows | lists i64 x band_rows x cap_x]
        off = _align(self.band_rows * 4, 16)
        self.x_bytes = off + self.band_rows * self.cap_x * 8
        self.send = e((self.n_bands, self.world, self.x_bytes), t.uint8)
        self.recv = e((self.n_bands, self.world, self.x_bytes), t.uint8)

        def views(buf):
            cnt = [[buf[i, r, :self.band_rows * 4].view(t.int32) for r in range(self.world)] for i in range(self.n_bands)]
            lst = [[buf[i, r, off:].view(t.int64).view(self.band_rows, self.cap_x) for r in range(self.world)]
                   for i in range(self.n_bands)]
            return cnt, lst
        self.send_cnt, self.send_lst = views(self.send)
        self.recv_cnt, self.recv_lst = views(self.recv)
        self.send_cnt_all = self.send[:, :, :off]
        self.buffers_ready = True

    def _assemble(self):
        """The bands' slots into the [bins, k] result (one gather per array; buffers kept across passes)."""
        if not self.gather:
            rows = self.ranges[self.rank][1] - self.ranges[self.rank][0]
            idx = self.torch.cat(self.own_idx)[:rows]
            dst = self.torch.cat(self.own_dst)[:rows]
            return idx, dst
        bi, ri, ti = self._map
        self.idx_full = self.res_idx[bi, ri, ti]
        self.dst_full = self.res_dst[bi, ri, ti]
        return self.idx_full, self.dst_full

    def calibrate(self):
        """Pick the shard mode by measurement: one warm-up and one timed pass of each mode on
        this job's own data, the slower rank's time counts (all-reduce MAX), ties go to the
        symmetric tile shard.  Every rank reaches the same decision: after each mode the ranks
        agree (all-reduce MIN) on ok / collective-not-offered / failed, so a failure on one rank
        stops all of them together instead of leaving its peers in the next collective.
        `calibration` keeps the measured seconds per pass."""
        import time
        times = {}
        marks, self._marks = self._marks, None           # the caller's timing marks are not for these passes
        for mode in ("tiles", "rows"):
            self.mode = mode
            state, err, elapsed = 2, None, 0.0           # 2 ok, 1 the backend lacks an exchange collective, 0 failed
            try:
                # warm-up pass (buffers, tile lists, communicator) with the ranks agreeing on every local step
                # before the next collective: whatever can fail on one rank alone (memory, a kernel fault) fails here
                self._checked = True
                try:
                    self._run()
                finally:
                    self._checked = False
                self._sync()
            except Exception as exc:
                err = exc
                state = 1 if (mode == "tiles" and _unsupported(exc)) else 0
            state = self._min_over_ranks(state)
            if state == 0:
                self.mode = None
                self._marks = marks
                if err is not None:
                    raise err
                raise PeerFailure("newref calibration (%s shard): another rank failed" % mode)
            if state == 1:
                times[mode] = None
                times["tiles_error"] = "%s: %s" % (type(err).__name__, err) if err is not None else "on another rank"
                continue
            # the TIMED pass runs as a production pass does (collectives in flight beside the kernels; the
            # agreement's all-reduces and host synchronisations would be what is compared otherwise).  The same
            # buffers, lists and communicator have just worked; should this pass still fail on one rank, that rank
            # must NOT enter the agreement below while its peers sit in the pass's collectives: the error leaves
            # calibrate() at once, the process ends non-zero and the launcher (ranks.launch, torchrun) stops the peers
            self.dist.barrier()
            t0 = time.perf_counter()
            self._run()
            self._sync()
            elapsed = time.perf_counter() - t0
            times[mode] = self._max_over_ranks(elapsed)
        self.mode = None
        self._marks = marks
        self.calibration = times
        return "rows" if times["tiles"] is None or times["rows"] < times["tiles"] else "tiles"

    def _min_over_ranks(self, value):
        on_gpu = self.dist.get_backend() != "gloo"
        t = self.torch.tensor([int(value)], dtype=self.torch.int32,
                              device=self.st.device if on_gpu else "cpu")
        self.dist.all_reduce(t, op=self.dist.ReduceOp.MIN)
        return int(t.item())

    def _sync(self):
        if getattr(self.st, "device", None) is not None and self.st.device.type == "cuda":
            self.torch.cuda.synchronize()

    def _max_over_ranks(self, seconds):
        on_gpu = self.dist.get_backend() != "gloo"
        t = self.torch.tensor([seconds], dtype=self.torch.float64,
                              device=self.st.device if on_gpu else "cpu")
        self.dist.all_reduce(t, op=self.dist.ReduceOp.MAX)
        return float(t.item())

    def _mark(self, name):
        """Record a timing event on the launch stream (only while `timing` is on): the time since the previous
        mark is booked under `name`."""
        if self._marks is not None:
            ev = self.torch.cuda.Event(enable_timing=True)
            ev.record()
            self._marks.append((name, ev))

    STAGES = ("prepared", "thresholds", "collected", "exported", "exchanged", "picked", "rescored", "finished", "gathered")

    def stage_ms(self):
        """Milliseconds of the last timed run's stages (call after a synchronize): every interval between two
        marks is booked under the later mark's name and the bands' intervals add up -- prepared, thresholds,
        collected (the tile kernel), [exported, exchanged (import)], picked, rescored, finished (exact path),
        [gathered (assembly)]; `comm` = the collectives, each waited for on the spot in a timing run."""
        m = self.last_marks or []
        out = {}
        for (_, a), (name, b) in zip(m, m[1:]):
            out[name] = out.get(name, 0.0) + a.elapsed_time(b)
        return out

    def collective_ms(self):
        """The collectives of the last run(timing=True), each waited for on the spot (its own duration incl. the
        wait for the slowest peer): [{name, bytes, ms}] (call after a synchronize)."""
        out = []
        for c in self.last_comm or []:
            a, b = c["events"]
            out.append({"name": c["name"], "bytes": c["bytes"],
                        "ms": a.elapsed_time(b) if a is not None else c["host_ms"]})
        return out

    last_comm = None

    def run(self, timing=False):
        """One pass.  timing=True records torch events between the stages on the launch stream (stage_ms() after
        a synchronize) and, on several ranks, waits for every collective where it is issued (collective_ms()):
        a diagnostic pass -- the production pass keeps the collectives in flight beside the kernels."""
        self._marks = [] if timing else None
        self.comm_log = [] if timing else None
        try:
            return self._run()
        finally:
            self.last_marks, self._marks = self._marks, None
            self.last_comm, self.comm_log = self.comm_log, None

    def _run(self):
        st = self.st
        if self.single and self._marks is None and hasattr(st, "full_pass"):
            st.full_pass(self.idx, self.dst)
            return self.idx, self.dst
        self._mark("start")
        self._local(st.prepare)
        self._mark("prepared")
        if self.single:
            st.thresholds(0, self.n_bins)
            self._mark("thresholds")
            st.collect(0, self.n_bins, 0, 1)
            self._mark("collected")
            self._finish(0, self.n_bins, self.idx, self.dst)
            return self.idx, self.dst

        rb, re = self.ranges[self.rank]
        if self.mode is None:
            measure = self.passes >= CALIBRATE_FROM_PASSES or os.environ.get("WC_NEWREF_SHARD") == "measure"
            self.mode = forced_shard_mode() or (self.calibrate() if measure else "tiles")
            self._untried = not measure and forced_shard_mode() is None       # one-shot job: the tile shard, untested
            st.prepare()
        bands = range(self.n_bands)
        gathers = []

        def finish_rows(i):
            b, e = self.band(self.rank, i)
            if e > b:
                self._finish(b, e, self.own_idx[i], self.own_dst[i])

        def gather_band(i):
            # (a collective is never issued inside _local: a rank whose local step failed must not have entered it)
            if self.gather:
                gathers.append(self._all_gather(self.res[i].view(-1), self.own[i], "result_all_gather[%d]" % i))
                self._mark("comm")

        def finish_all():
            for g in gathers:
                g.wait()
            out = self._assemble()
            self._mark("gathered")
            return out

        if self.mode == "rows":
            # row band of this rank against all columns: no exchange; the result gather of band i travels while
            # band i + 1 is re-scored
            def head():
                st.thresholds(rb, re)
                self._mark("thresholds")
                st.collect(rb, re, 0, 1)
                self._mark("collected")
            self._local(head)
            for i in bands:
                self._local(lambda i=i: finish_rows(i))
                gather_band(i)
            return finish_all()

        # thresholds: owner computes, everyone needs them for the tiles it was dealt
        def own_thresholds():
            if not self.buffers_ready:
                self._alloc_exchange()
            st.thresholds(rb, re)
            self.thr_all[self.rank].zero_()
            st.get_thr(rb, re, self.thr_all[self.rank])
        self._local(own_thresholds)
        self._mark("thresholds")
        self._all_gather(self.thr_all.view(-1), self.thr_all[self.rank].clone(), "threshold_all_gather").wait()
        self._mark("comm")

        # this rank's share of the symmetric tile space, candidates for all rows
        def tiles():
            for r, (b, e) in enumerate(self.ranges):
                if r != self.rank:
                    st.set_thr(b, e, self.thr_all[r])
            self._mark("thresholds")
            st.collect(0, self.n_bins, self.rank, self.world)
            self._mark("collected")
            self.send_cnt_all.zero_()
        self._local(tiles)

        # the lists of foreign rows travel to their owners band by band
        def export_band(i):
            for r in range(self.world):
                b, e = self.band(r, i)
                if r != self.rank and e > b:
                    st.export(b, e, self.cap_x, self.send_cnt[i][r], self.send_lst[i][r])
            self._mark("exported")

        exchanges = []
        try:
            for i in bands:
                self._local(lambda i=i: export_band(i))
                exchanges.append(self._all_to_all(self.recv[i].view(-1), self.send[i].view(-1), "list_all_to_all[%d]" % i))
                self._mark("comm")
                if i == 0 and getattr(self, "_untried", False):
                    exchanges[0].wait()          # a backend without the collective says so here at the latest
        except Exception as exc:
            # a one-shot job takes the tile shard without having tried it: a backend that does not offer the
            # exchange collective says so on every rank alike, and the row shard (all-gathers only) takes over
            if not (getattr(self, "_untried", False) and _unsupported(exc)):
                raise
            self.mode, self._untried = "rows", False
            st.prepare()
            return self._run()
        self._untried = False

        # owners merge what they received and finish their rows, one band behind the exchange
        def own_band(i):
            b, e = self.band(self.rank, i)
            if e > b:
                for r in range(self.world):
                    if r != self.rank:
                        st.import_(b, e, self.cap_x, self.recv_cnt[i][r], self.recv_lst[i][r])
            self._mark("exchanged")
            finish_rows(i)
        for i in bands:
            exchanges[i].wait()
            self._local(lambda i=i: own_band(i))
            gather_band(i)
        return finish_all()

    def _local(self, work):
        """This rank's work between two collectives.  During the calibration warm-up passes (`_checked`) the
        ranks agree on its outcome before any of them enters the next collective: a rank that failed
        would otherwise leave its peers waiting there."""
        if not self._checked:
            return work()
        err = None
        try:
            work()
        except Exception as exc:
            err = exc
        if self._min_over_ranks(0 if err is not None else 2) == 0:
            if err is not None:
                raise err
            raise PeerFailure("newref: another rank failed before a collective")

    def _finish(self, rb, re, idx, dst):
        if self._marks is None:
            self.st.finish(rb, re, idx, dst)
            return
        if hasattr(self.st, "pick"):
            self.st.pick(rb, re, idx, dst)
            self._mark("picked")
            self.st.rescore_pairs(rb, re, idx, dst)
        else:
            self.st.rescore(rb, re, idx, dst)
        self._mark("rescored")
        self.st.fallback(rb, re, idx, dst)
        self._mark("finished")


class TestBatch(object):
    """Device-resident batched `test` of this rank's sample shard (no collective)."""

    def __init__(self, reference, counts, threshold, minrefbins=25, repeats=5, chromosomes=None,
                 max_calls=64, mineffectsize=0.0):
        import torch
        self.torch = torch
        self.lib = _lib.load()
        self.ref = reference
        self.counts = counts
        self.ns = int(counts.shape[0])
        self.thr = float(threshold)
        self.minrefbins, self.repeats, self.max_calls = int(minrefbins), int(repeats), int(max_calls)
        self.mineffectsize = float(mineffectsize)
        self.sel = np.ascontiguousarray(chromosomes if chromosomes is not None else range(1, 23), dtype=np.int32)
        dev = counts.device
        f64 = torch.float64
        self.results_z = torch.empty((self.ns, reference.n_total), dtype=f64, device=dev)
        self.results_r = torch.empty((self.ns, reference.n_total), dtype=f64, device=dev)
        self.cwz = torch.empty((self.ns, len(self.sel)), dtype=f64, device=dev)
        self.calls = torch.zeros((self.ns, self.max_calls, 5), dtype=f64, device=dev)
        self.n_calls = torch.zeros((self.ns,), dtype=torch.int32, device=dev)
        self.asdef = torch.empty((self.ns,), dtype=f64, device=dev)

    def run(self):
        stream = self.torch.cuda.current_stream().cuda_stream
        _lib.check(self.lib.wc_test_batch_dev(
            self.ref.ctx, stream, self.ref.handle, self.counts.data_ptr(), self.ns, self.thr, self.minrefbins,
            self.repeats, self.mineffectsize, _lib.ptr(self.sel), len(self.sel), self.max_calls,
            self.results_z.data_ptr(),
            self.results_r.data_ptr(), self.cwz.data_ptr(), self.calls.data_ptr(), self.n_calls.data_ptr(),
            self.asdef.data_ptr()))


class TestPipeline(object):
    """Several batches of the batched `test` in flight on ONE GPU.

    wc_test_batch_dev reads counts back during the segmentation rounds, i.e. it blocks its host thread, and a good
    part of a batch's kernels leave most of the chip idle (the tree walk of the hot regions, stdDevAvg, the prefix
    set-up: 1-3 waves per SIMD).  `depth` slots -- each a context of its own (scratch, side stream), its own copy of
    the reference, a stream and a host thread -- take the batches round robin, so the narrow kernels of one batch
    run beside the wide ones of another: 128 x 250 kb 0.57 -> 0.37 ms per batch, 125 x 50 kb 2.7 -> 2.0 ms at depth 4
    (depth 2: 0.40-0.59 / 2.2-2.7 depending on which hardware queues the two streams land on; 6 and 8 add nothing).
    Every batch is computed by the same kernels as a lone TestBatch: results are identical.
    The reference has no counterpart (one `test` process per sample, wisecondor.py:173-268)."""

    def __init__(self, reference, threshold, depth=4, **batch_args):
        import torch
        self.torch = torch
        self.depth = max(1, int(depth))
        self.threshold = float(threshold)
        self.batch_args = batch_args
        self.slots = []
        for i in range(self.depth):
            ctx = None if i == 0 else _lib.new_context(reference.device)
            ref = reference if i == 0 else reference.clone(ctx)
            self.slots.append(dict(ctx=ctx, ref=ref, stream=torch.cuda.Stream(device=reference.device), tb={}))

    def _batch(self, slot, counts):
        key = (int(counts.shape[0]), int(counts.shape[1]))
        tb = slot["tb"].get(key)
        if tb is None:
            tb = slot["tb"][key] = TestBatch(slot["ref"], counts, self.threshold, **self.batch_args)
        tb.counts = counts
        return tb

    def run(self, batches, consume=None):
        """batches: int32 count tensors [samples, bins] on the device.  consume(index, TestBatch), if given, is
        called from the slot's thread once that batch's results are complete (the slot's output buffers are reused
        by its next batch of the same shape).  Returns after every batch has finished."""
        import threading
        errors = []
        ready = self.torch.cuda.Event()
        ready.record()                               # the batches' producers on the caller's stream

        def work(i):
            slot = self.slots[i]
            try:
                with self.torch.cuda.stream(slot["stream"]):
                    slot["stream"].wait_event(ready)
                    for b in range(i, len(batches), self.depth):
                        tb = self._batch(slot, batches[b])
                        tb.run()
                        if consume is not None:
                            slot["stream"].synchronize()
                            consume(b, tb)
                    slot["stream"].synchronize()
            except BaseException as exc:             # re-raised in the caller's thread
                errors.append(exc)

        threads = [threading.Thread(target=work, args=(i,)) for i in range(min(self.depth, len(batches)))]
        for t in threads:
            t.start()
        for t in threads:
            t.join()
        if errors:
            raise errors[0]

    def close(self):
        """Frees the extra slots' references and contexts (slot 0 is the caller's reference)."""
        for slot in self.slots[1:]:
            slot["ref"].close()
            _lib.destroy_context(slot["ctx"])
        self.slots = self.slots[:1]

    def __enter__(self):
        return self

    def __exit__(self, *exc):
        self.close()
        return False

    def __del__(self):
        try:
            self.close()
        except Exception:       # interpreter shutdown: the library may already be gone
            pass


def shard_samples(n_samples, rank, world):
    """Contiguous sample shard of `rank` (sizes differ by at most one)."""
    per, extra = divmod(n_samples, world)
    lo = rank * per + min(rank, extra)
    return lo, lo + per + (1 if rank < extra else 0)
