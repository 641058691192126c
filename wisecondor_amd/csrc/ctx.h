// Per-device context: cached workspaces for the newref and test paths.
#pragma once
#include "common.h"

#define WC_MAX_CHROM 64

struct NewrefState {
    // problem
    int64_t n_bins = 0, n_samples = 0;
    int64_t bins_pad = 0;   // rows padded to the 128-row tile
    int64_t k_pad16 = 0;    // samples padded to the 64-wide float16 k-slab of the threshold tiles
    int n_chrom = 0, k = 0, sum_order = 0;
    int64_t chrom_off[WC_MAX_CHROM + 1] = {0};
    const double *corrected = nullptr;  // device, caller owned
    // tuning
    int64_t n_sample_cols = 0;      // M, multiple of 128
    int64_t cap = 0;                // candidate-list capacity per row
    int64_t expect = 0;             // expected candidates per row under the sampled threshold
    float beta = 0.f;               // relative half-width of the key error interval
    bool prepared = false;
    double tau = 0.0;               // weight of the float16 representation-error split (k_convert)
    // device buffers
    wc::DevBuf col_partial, col_mean, norm_lo, norm_hi, chrom_of_row, chrom_range, chrom_off_dev;
    wc::DevBuf sample_rows, sample_slot, s32, s_norm_lo, s_chrom, s_range, a16, s16;
    wc::DevBuf keys1, thr, cnt, list, tiles;
    wc::DevBuf fb_rows, fb_count, fb_scratch, stats, tiles0, pw_prog, pairs, x64, m2;
    bool fb_dirty = true;       // fb_count holds a finished pass's counts (k_convert zeroes it for the next job)
    int64_t s_pad = 0;      // samples padded to whole 16-sample chunks (x64 row stride)
    bool exact_only = false; // refsize beyond the candidate lists' design size: every row takes the exact path
    bool x64_pad = false;   // the padded float64 image exists (pair engine usable)
    int pw_leaves = 0;
    int64_t pw_for = -1;
    // host-side caches so that repeated calls on the same layout enqueue kernels only
    std::vector<int64_t> sample_key, tiles0_key, tiles1_key, chrom_key;
    int64_t tiles0_n = 0, tiles1_n = 0;
};

// Workspaces of the batched test path (grow-only, reused across calls).
struct TestState {
    std::vector<int> sel_host;      // chromosome selection currently held by `sel`
    wc::DevBuf counts, totals, raw, proj, data, xt, xc, zt, rt, nt, sdt, z, r, n, sd_avg;
    wc::DevBuf zc, rc, gpos, clean_n, regions, sel;
    wc::DevBuf res_z, res_r, cwz, calls, n_calls;
    // Stouffer search
    wc::DevBuf zs, rs2, ns2, sds, sub, tmin, tmax, tmin2, tmax2, cell_state, cell_rec, prefix, reg_abs, reg_flag, rs, jobs_a, jobs_b, job_cnt, partial, cbound, cuts, job_res, hot, cand, cand_cnt;
    wc::DevBuf seg, seg_cnt, out_val, out_x, out_y, out_n, whole, effect, misc, misc2, reduce_tmp, win_bits, bit_off, pairs_a, pairs_b, cut_vals;
    wc::DevBuf walk_hot;             // k_seg_walk's early starters: [0] count, [16 ..] the list, [16 + 4096 ..] every region's place in it
    void *walk_hot_clean = nullptr;  // the buffer whose count has been zeroed once (k_walk_rows resets it after every walk)
    bool tail_used = false, no_tail = false;   // run_repeat: the late repeats ran as ONE workgroup (k_lat_repeats); no_tail: the batch is being repeated without that
    int *tail_flag = nullptr;                  // ... its overflow word (device)
    int64_t rs_len = 0;
    int64_t last_segs = 0;       // segments of the last segmentation call; negative: -(bound), the count is on the device
    int lat_left = 1;
    // latency mode: the captured call (hipGraph), the arguments it was captured for, and whether
    // an eager call of that shape has sized the workspaces
    hipGraphExec_t lat_exec = nullptr;
    std::vector<int64_t> lat_key, lat_fail_key;   // lat_fail_key: a shape whose capture failed stays on the eager latency kernels
    bool lat_warm = false;
    unsigned long long lat_epoch = 0;   // wc::realloc_epoch() at capture time            // latency mode: index of the counter that holds the jobs left after the last round
    // optional stage timing of wc_test_batch_dev (wc_test_profile): events on the launch stream
    // between the stages, and counts of the window evaluations the search kernels executed
    bool profile = false;
    std::vector<hipEvent_t> prof_ev;    // event pool, grown on demand
    std::vector<int> prof_tag;          // tag of every mark of the last batch, in record order
    wc::DevBuf sd_fail;                 // per-sample flags of k_sd_fast (1: the serial kernel takes the sample)
    bool tree_done = false;             // run_stouffer: the tree kernel finished the recursion and wrote the call rows
    bool tree_pending = false;          // ... but its status words are still on their way to pinned memory (deferred check)
    bool sm_out = false;                // run_repeat: zt / rt / nt / sdt of the last call are sample-major [Ns, B] (tiled first repeat)
    bool no_tree = false;               // repeat of a batch the tree kernel passed on: host-driven rounds only
    int64_t tree_seg_cap = 0;
    bool lat_ride = false;              // latency mode: stdDevAvg rides in k_seg_tree's grid (run_repeat -> run_seg_lat)
    double *lat_ride_out2 = nullptr;
    wc::DevBuf prof_work;               // u64[2]: windows evaluated by k_seg_search, evaluations by k_seg_quiet
    void mark(int tag, hipStream_t stream) {
        if (!profile) return;
        const size_t at = prof_tag.size();
        if (at >= prof_ev.size()) {
            hipEvent_t e = nullptr;
            if (hipEventCreate(&e) != hipSuccess) return;
            prof_ev.push_back(e);
        }
        if (hipEventRecord(prof_ev[at], stream) == hipSuccess) prof_tag.push_back(tag);
    }
};

// state carried from wc_newref_prep_gram to wc_newref_prep_finish
struct PrepState {
    int64_t S = 0, Btot = 0, B = 0;
    bool ready = false;
    wc::DevBuf eig_ws;          // eigh.hip: working copy, reflectors, tridiagonal, vectors
};

struct wc_ctx {
    int device = 0;
    PrepState prep;
    NewrefState nr;
    TestState ts;
    wc::DevBuf tmp_a, tmp_b, tmp_c, tmp_d;  // host-pointer API staging
    void *pinned = nullptr;
    size_t pinned_bytes = 0;
    int64_t last_stats[8] = {0};
    // side stream for work that only feeds an output (overlaps with the main stream)
    hipStream_t lat_stream = nullptr;      // latency-mode calls of the test path run (and are captured) here
    hipEvent_t ev_lat_in = nullptr, ev_lat_out = nullptr;
    hipStream_t side = nullptr;
    hipEvent_t ev_fork = nullptr, ev_join = nullptr;
    bool side_pending = false;
    hipStream_t side2 = nullptr;           // big batches: the inflated arrays and the whole-region values beside stdDevAvg, not behind it
    hipEvent_t ev_join2 = nullptr;
    bool side2_pending = false;
    bool side_fresh = false;               // the side stream was forked from the launch stream and NOTHING has been enqueued on the
                                           // launch stream since: the next side_begin needs no second event record + wait
    // small pinned host block for count read-backs (a pageable destination costs a staged copy)
    int ensure_pinned(size_t bytes) {
        if (pinned_bytes >= bytes) return WC_OK;
        if (pinned) (void)hipHostFree(pinned);
        pinned = nullptr;
        pinned_bytes = 0;
        WC_HIP(hipHostMalloc(&pinned, bytes, hipHostMallocDefault));
        pinned_bytes = bytes;
        return WC_OK;
    }
    int ensure_side_stream() {
        if (side) return WC_OK;
        // lowest priority: what runs here only feeds outputs (stdDevAvg, the inflated arrays, whole-region values) and
        // must not take CUs from the launch stream's critical path
        int prio_least = 0, prio_greatest = 0;
        if (hipDeviceGetStreamPriorityRange(&prio_least, &prio_greatest) != hipSuccess) prio_least = 0;
        WC_HIP(hipStreamCreateWithPriority(&side, hipStreamNonBlocking, prio_least));
        WC_HIP(hipEventCreateWithFlags(&ev_fork, hipEventDisableTiming));
        WC_HIP(hipEventCreateWithFlags(&ev_join, hipEventDisableTiming));
        WC_HIP(hipStreamCreateWithPriority(&side2, hipStreamNonBlocking, prio_least));
        WC_HIP(hipEventCreateWithFlags(&ev_join2, hipEventDisableTiming));
        return WC_OK;
    }

    std::vector<wc::DevBuf *> all_buffers() {
        return {&nr.col_partial, &nr.col_mean, &nr.norm_lo, &nr.norm_hi, &nr.chrom_of_row, &nr.chrom_range,
                &nr.chrom_off_dev, &nr.sample_rows, &nr.sample_slot, &nr.s32, &nr.s_norm_lo, &nr.s_chrom, &nr.s_range, &nr.keys1,
                &nr.thr, &nr.cnt, &nr.list, &nr.tiles, &nr.fb_rows, &nr.fb_count, &nr.fb_scratch,
                &nr.stats, &nr.tiles0, &nr.pw_prog, &nr.pairs, &nr.x64, &nr.m2, &nr.a16, &nr.s16, &tmp_a, &tmp_b, &tmp_c, &tmp_d,
                &ts.counts, &ts.totals, &ts.raw, &ts.proj, &ts.data, &ts.xt, &ts.xc, &ts.zt, &ts.rt, &ts.nt,
                &ts.sdt, &ts.z, &ts.r, &ts.n, &ts.sd_avg, &ts.zc, &ts.rc, &ts.gpos, &ts.clean_n, &ts.regions,
                &ts.sel, &ts.res_z, &ts.res_r, &ts.cwz, &ts.calls, &ts.n_calls, &ts.zs, &ts.rs2, &ts.ns2, &ts.sds, &ts.sub, &ts.tmin, &ts.tmax, &ts.tmin2, &ts.tmax2, &ts.cell_state, &ts.cell_rec, &ts.prefix, &ts.reg_abs,
                &ts.reg_flag, &ts.rs, &ts.jobs_a, &ts.jobs_b, &ts.job_cnt, &ts.partial, &ts.cbound, &ts.cuts, &ts.job_res, &ts.hot,
                &ts.cand, &ts.cand_cnt, &ts.seg, &ts.seg_cnt, &ts.out_val, &ts.out_x, &ts.out_y, &ts.out_n,
                &ts.whole, &ts.effect, &ts.misc, &ts.misc2, &ts.reduce_tmp, &ts.win_bits, &ts.bit_off, &ts.pairs_a, &ts.pairs_b, &ts.cut_vals, &ts.prof_work, &ts.sd_fail, &ts.walk_hot, &prep.eig_ws};
    }
};

namespace wc {
// eigh.hip: leading eigenpairs of a device-resident symmetric matrix (host outputs)
int sym_eigh_leading(wc_ctx *ctx, const double *matrix_dev, int64_t n, int n_pairs, double *eigvals_out,
                     double *eigvecs_out);
}  // namespace wc
