// test path on gfx950: sample preparation + PCA-apply, masked z-score repeats,
// minrefbins cleaning, Stouffer window segmentation, call mapping, inflation.
//
// Replaces (per batch of samples) toNumpyRefFormat / applyPCA / getOptimalCutoff /
// trySample / repeatTest / fillTri / TriArr.segmentTri and the numeric part of
// toolTest (wisetools.py:104-113, 267-278, 328-336, 407-448, 466-472;
// triarray.py:59-84; wisecondor.py:199-268).  Everything is float64: these stages
// are gather/HBM or fp64-ALU bound, and the reference's flags and segment bounds
// hinge on exact comparisons.
#include "ctx.h"

#include <algorithm>
#include <cstring>

struct wc_reference {
    wc_ctx *ctx = nullptr;
    unsigned long long serial = 0;         // unique per created reference (a freed handle's address may be reused)
    int64_t B = 0, Btot = 0;
    int k = 0, n_chrom = 0, n_comp = 0;
    int64_t moff[WC_MAX_CHROM + 1] = {0};  // masked-bin offsets per chromosome
    int64_t goff[WC_MAX_CHROM + 1] = {0};  // genomic-bin offsets per chromosome
    double cutoff = 0.0;
    wc::DevBuf gidx, nref, pca_mean, pca_comp, m2g, g2m, moff_dev, goff_dev;
    wc::DevBuf users_off, users;   // reverse reference lists: bins that use bin g (CSR over g)
};

namespace {

// Development aid: with WC_DEBUG_TIMES=1 the latency-mode kernels leave s_memtime stamps of their
// phases (workgroup `DBG_BLOCK`, thread 0) in a device array read back by wc_debug_times.
__device__ unsigned long long g_dbg[64];
__device__ int g_dbg_on = 0;
#define WC_STAMP_AT(slot, blk)                                                              \
    do {                                                                                    \
        if (g_dbg_on && threadIdx.x == 0 && (int)(blk) == g_dbg_on - 1) g_dbg[slot] = clock64(); \
    } while (0)
#define WC_STAMP(slot) WC_STAMP_AT(slot, blockIdx.x)

constexpr int MAX_COMP = 8;
constexpr int SHORT_SEG = 1024;  // segments up to this length take the counting median in k_call_post
constexpr int ROWS_HALF = 64;  // window rows per side handled by one search workgroup (one lane each)
constexpr int CAND_CAP = 64;

struct Region { long long off; int n; int pad; };
struct Job { int region, lo, hi, pad; };
struct Extreme { double maxv, minv; int max_x, max_y, min_x, min_y; };
struct Seg { double val; int region, x, y, pad; };

inline int64_t cdiv(int64_t a, int64_t b) { return (a + b - 1) / b; }

// ------------------------------------------------------------- reductions ----
// getOptimalCutoff (wisetools.py:328-336): three rounds of mean + 3 sd over distances[mask].
// The cutoff decides list membership by `distance < cutoff`, so the moments are taken in
// numpy's own order rather than "some deterministic order": distances[mask] is the row-major
// compaction of the kept values; np.mean / np.std reduce it with add.reduce, i.e. pairwise
// within pieces of 8192 elements, the piece sums accumulated left to right (np.std: subtract
// the mean, multiply the difference with itself, add.reduce, divide, sqrt).
__global__ void k_cut_count(const double *__restrict__ d, int64_t rows, int k, double cutoff, int *__restrict__ cnt) {
    const int64_t r = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (r >= rows) return;
    int c = 0;
    for (int j = 0; j < k; ++j) c += d[r * k + j] < cutoff;
    cnt[r] = c;
}

// exclusive scan of the per-row counts, one workgroup (rows <= a few hundred thousand)
__global__ __launch_bounds__(1024) void k_cut_scan(const int *__restrict__ cnt, int64_t rows, long long *__restrict__ off) {
    __shared__ long long part[1024];
    const int tid = threadIdx.x;
    const int64_t per = (rows + 1023) / 1024, lo = tid * per, hi = lo + per < rows ? lo + per : rows;
    long long s = 0;
    for (int64_t r = lo; r < hi; ++r) s += cnt[r];
    part[tid] = s;
    wc_sync();
    if (tid == 0) {
        long long run = 0;
        for (int t = 0; t < 1024; ++t) { const long long v = part[t]; part[t] = run; run += v; }
        off[rows] = run;
    }
    wc_sync();
    long long run = part[tid];
    for (int64_t r = lo; r < hi; ++r) { off[r] = run; run += cnt[r]; }
}

__global__ void k_cut_compact(const double *__restrict__ d, int64_t rows, int k, double cutoff,
                              const long long *__restrict__ off, double *__restrict__ out) {
    const int64_t r = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (r >= rows) return;
    long long at = off[r];
    for (int j = 0; j < k; ++j) {
        const double v = d[r * k + j];
        if (v < cutoff) out[at++] = v;
    }
}

// one wave per 8192-element piece: numpy's pairwise tree of the piece (mode 1: of (v - mean)^2)
__global__ __launch_bounds__(64) void k_cut_pieces(const double *__restrict__ v, const long long *__restrict__ n_ptr,
                                                   double mean, int mode, double *__restrict__ piece) {
    __shared__ wc::PwWaveScratch sc;
    const long long n = *n_ptr, lo = (long long)blockIdx.x * WC_NPY_BUFSIZE;
    if (lo >= n) return;
    const long long m = n - lo < WC_NPY_BUFSIZE ? n - lo : WC_NPY_BUFSIZE;
    const double *p = v + lo;
    const double s = wc::pairwise_tree_wave(
        [&](int64_t i) {
            const double x = p[i];
            if (mode == 0) return x;
            const double t = x - mean;
            return t * t;
        },
        (int64_t)m, (int)threadIdx.x, sc);
    if (threadIdx.x == 0) piece[blockIdx.x] = s;
}

__global__ void k_cut_fold(const double *__restrict__ piece, const long long *__restrict__ n_ptr, double *__restrict__ out) {
    if (blockIdx.x != 0 || threadIdx.x != 0) return;
    const long long n = *n_ptr, pieces = (n + WC_NPY_BUFSIZE - 1) / WC_NPY_BUFSIZE;
    double res = 0.0;
    if (pieces == 1) res = piece[0];
    else for (long long c = 0; c < pieces; ++c) res = res + piece[c];
    out[0] = res;
    out[1] = (double)n;
}

// the mask getOptimalCutoff hands back (wisetools.py:332): distances < the cutoff of the iteration before
__global__ void k_cut_mask(const double *__restrict__ d, int64_t count, double cutoff, uint8_t *__restrict__ mask) {
    const int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (i < count) mask[i] = d[i] < cutoff ? 1 : 0;
}

int device_cutoff(wc_ctx *ctx, const double *d_dev, int64_t count, int repeats, hipStream_t stream, double *out,
                  int k = 1, double *previous = nullptr) {
    int rc;
    WC_CHECK(k > 0 && count % k == 0, WC_E_ARG, "cutoff: distances are not [rows, k]");
    const int64_t rows = count / k;
    const int64_t max_pieces = count / WC_NPY_BUFSIZE + 1;
    TestState &ts = ctx->ts;
    if ((rc = ts.reduce_tmp.reserve(sizeof(double) * (max_pieces + 4) + sizeof(long long) * (rows + 2) + sizeof(int) * rows)))
        return rc;
    if ((rc = ts.cut_vals.reserve(sizeof(double) * count))) return rc;
    double *piece = ts.reduce_tmp.as<double>();
    double *res = piece + max_pieces;                              // {sum, n}
    long long *off = (long long *)(res + 4);
    int *cnt = (int *)(off + rows + 2);
    double cutoff = INFINITY;
    const unsigned gr = (unsigned)cdiv(rows, 256);
    for (int it = 0; it < repeats; ++it) {
        double h[2];
        if (previous) *previous = cutoff;
        hipLaunchKernelGGL(k_cut_count, dim3(gr), dim3(256), 0, stream, d_dev, rows, k, cutoff, cnt);
        hipLaunchKernelGGL(k_cut_scan, dim3(1), dim3(1024), 0, stream, (const int *)cnt, rows, off);
        hipLaunchKernelGGL(k_cut_compact, dim3(gr), dim3(256), 0, stream, d_dev, rows, k, cutoff,
                           (const long long *)off, ts.cut_vals.as<double>());
        double mean = 0.0;
        for (int mode = 0; mode < 2; ++mode) {
            hipLaunchKernelGGL(k_cut_pieces, dim3((unsigned)max_pieces), dim3(64), 0, stream,
                               (const double *)ts.cut_vals.as<double>(), (const long long *)(off + rows), mean, mode, piece);
            hipLaunchKernelGGL(k_cut_fold, dim3(1), dim3(1), 0, stream, (const double *)piece,
                               (const long long *)(off + rows), res);
            WC_HIP(hipMemcpyAsync(h, res, sizeof(h), hipMemcpyDeviceToHost, stream));
            WC_HIP(hipStreamSynchronize(stream));
            if (mode == 0) mean = h[0] / h[1];
        }
        const double sd = sqrt(h[0] / h[1]);
        cutoff = mean + 3 * sd;
    }
    *out = cutoff;
    return WC_OK;
}

// Per-bin reference lists: index[i][distances[i] < cutoff] (wisetools.py:424) mapped
// from "other chromosomes" positions (wisetools.py:420-421) to masked-bin numbers.
__global__ void k_ref_lists(const int *__restrict__ idx, const double *__restrict__ dist, int64_t B, int k,
                            const int64_t *__restrict__ moff, int n_chrom, double cutoff,
                            int *__restrict__ gidx, int *__restrict__ nref) {
    int64_t b = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (b >= B) return;
    int c = 0;
    while (c + 1 < n_chrom && b >= moff[c + 1]) ++c;
    const int64_t cs = moff[c], ce = moff[c + 1], len = B - (ce - cs);
    int n = 0;
    for (int r = 0; r < k; ++r) {
        if (!(dist[b * k + r] < cutoff)) continue;
        int64_t p = idx[b * k + r];
        if (p < 0) p += len;  // numpy negative indexing (the -1 padding picks the last element)
        int64_t g = p < cs ? p : p + (ce - cs);
        if (p < 0 || g >= B) g = -1;  // numpy would raise IndexError; dropped here
        gidx[b * k + n] = (int)g;
        ++n;
    }
    nref[b] = n;
    for (int r = n; r < k; ++r) gidx[b * k + r] = -1;
}

// Reverse lists: for every bin g the bins whose reference list holds g (order irrelevant).  A
// new flag on (g, sample) can only change the results of those bins (wisetools.py:424-427).
__global__ void k_count_users(const int *__restrict__ gidx, const int *__restrict__ nref, int64_t B, int k,
                              int *__restrict__ cnt) {
    int64_t t = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (t >= B * k) return;
    const int64_t b = t / k;
    const int r = (int)(t - b * k);
    if (r >= nref[b]) return;
    const int g = gidx[t];
    if (g >= 0) atomicAdd(&cnt[g], 1);
}

__global__ void k_fill_users(const int *__restrict__ gidx, const int *__restrict__ nref, int64_t B, int k,
                             const int *__restrict__ off, int *__restrict__ cursor, int *__restrict__ users) {
    int64_t t = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (t >= B * k) return;
    const int64_t b = t / k;
    const int r = (int)(t - b * k);
    if (r >= nref[b]) return;
    const int g = gidx[t];
    if (g >= 0) users[off[g] + atomicAdd(&cursor[g], 1)] = (int)b;
}

// -------------------------------------------------------- sample preparation ----
// Per-sample read totals, TOT_SPLIT workgroups per sample over contiguous slices of the bins; the
// partial sums are integers (exact in any order), k_normalize adds the slices up.
constexpr int TOT_SPLIT = 16;
__global__ __launch_bounds__(256) void k_sample_totals(const int *__restrict__ counts, int64_t Btot,
                                                       long long *__restrict__ partial) {
    __shared__ long long sh[4];
    const int *row = counts + (int64_t)blockIdx.x * Btot;
    const int64_t per = (Btot + TOT_SPLIT - 1) / TOT_SPLIT;
    const int64_t lo = (int64_t)blockIdx.y * per, hi = lo + per < Btot ? lo + per : Btot;
    long long s = 0;
    for (int64_t g0 = lo + threadIdx.x; g0 < hi; g0 += 256 * 8) {      // eight loads of a trip in flight together
        int v[8];
#pragma unroll
        for (int e = 0; e < 8; ++e) v[e] = g0 + 256 * e < hi ? row[g0 + 256 * e] : 0;
#pragma unroll
        for (int e = 0; e < 8; ++e) s += v[e];
    }
    for (int o = 32; o > 0; o >>= 1) s += __shfl_xor(s, o);
    if ((threadIdx.x & 63) == 0) sh[threadIdx.x >> 6] = s;
    wc_sync();
    if (threadIdx.x == 0) partial[(int64_t)blockIdx.x * TOT_SPLIT + blockIdx.y] = (sh[0] + sh[1]) + (sh[2] + sh[3]);
}

// toNumpyRefFormat (wisetools.py:275-276): x = counts / total, masked bins only
__global__ void k_normalize(const int *__restrict__ counts, int64_t Btot, const int *__restrict__ m2g, int64_t B,
                            const long long *__restrict__ partial, double *__restrict__ raw) {
    int64_t b = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    int64_t i = blockIdx.y;
    if (b >= B) return;
    long long t = 0;
#pragma unroll
    for (int q = 0; q < TOT_SPLIT; ++q) t += partial[i * TOT_SPLIT + q];
    raw[i * B + b] = (double)counts[i * Btot + m2g[b]] / (double)t;
}

// applyPCA step 1 (wisetools.py:109): t = (x - mean) . components^T
// Projection on the components: PROJ_SPLIT workgroups per sample, each over a contiguous
// slice of the bins; k_pca_apply adds the slice sums in slice order (deterministic).
constexpr int PROJ_SPLIT = 8;
// raw != NULL: the normalised vectors as k_normalize wrote them; raw == NULL (batches): the same values taken from
// the counts on the fly (one IEEE division per element, the expression of k_normalize)
__global__ __launch_bounds__(256) void k_pca_project(const double *__restrict__ raw, int64_t B,
                                                     const double *__restrict__ mean, const double *__restrict__ comp,
                                                     int n_comp, double *__restrict__ proj,
                                                     const int *__restrict__ counts = nullptr, int64_t Btot = 0,
                                                     const int *__restrict__ m2g = nullptr,
                                                     const long long *__restrict__ partial = nullptr) {
    __shared__ double sh[MAX_COMP][256];
    const double *x = raw ? raw + (int64_t)blockIdx.x * B : nullptr;
    const int *cnt = counts ? counts + (int64_t)blockIdx.x * Btot : nullptr;
    long long tot = 0;
    if (!raw)
#pragma unroll
        for (int q = 0; q < TOT_SPLIT; ++q) tot += partial[(int64_t)blockIdx.x * TOT_SPLIT + q];
    const int64_t per = (B + PROJ_SPLIT - 1) / PROJ_SPLIT;
    const int64_t b_lo = (int64_t)blockIdx.y * per, b_hi = b_lo + per < B ? b_lo + per : B;
    double acc[MAX_COMP];
#pragma unroll
    for (int c = 0; c < MAX_COMP; ++c) acc[c] = 0.0;
    // a thread's terms in the order b, b + 256, ... as ever; the loads of FOUR trips are requested together (index,
    // then count / mean / components: one trip at a time was 27 dependent round trips per thread at 50 kb, 68 us)
    for (int64_t b0 = b_lo + threadIdx.x; b0 < b_hi; b0 += 4 * 256) {
        int gi[4];
        double xr[4], mn[4], cp[4][MAX_COMP];
#pragma unroll
        for (int e = 0; e < 4; ++e) gi[e] = (!raw && b0 + 256 * e < b_hi) ? m2g[b0 + 256 * e] : 0;
#pragma unroll
        for (int e = 0; e < 4; ++e) {
            const int64_t b = b0 + 256 * e;
            const bool in = b < b_hi;
            mn[e] = in ? mean[b] : 0.0;
            xr[e] = (in && raw) ? x[b] : 0.0;
#pragma unroll
            for (int c = 0; c < MAX_COMP; ++c) cp[e][c] = (in && c < n_comp) ? comp[(int64_t)c * B + b] : 0.0;
        }
        int cv[4];
#pragma unroll
        for (int e = 0; e < 4; ++e) cv[e] = (!raw && b0 + 256 * e < b_hi) ? cnt[gi[e]] : 0;
#pragma unroll
        for (int e = 0; e < 4; ++e) {
            if (b0 + 256 * e >= b_hi) continue;
            const double xv = raw ? xr[e] : (double)cv[e] / (double)tot;
            const double d = xv - mn[e];
#pragma unroll
            for (int c = 0; c < MAX_COMP; ++c)
                if (c < n_comp) acc[c] += d * cp[e][c];
        }
    }
#pragma unroll
    for (int c = 0; c < MAX_COMP; ++c) sh[c][threadIdx.x] = acc[c];
    wc_sync();
    for (int o = 128; o > 0; o >>= 1) {
        if ((int)threadIdx.x < o)
#pragma unroll
            for (int c = 0; c < MAX_COMP; ++c) sh[c][threadIdx.x] += sh[c][threadIdx.x + o];
        wc_sync();
    }
    if ((int)threadIdx.x < n_comp)
        proj[((int64_t)blockIdx.x * PROJ_SPLIT + blockIdx.y) * MAX_COMP + threadIdx.x] = sh[threadIdx.x][0];
}

__global__ void k_pca_apply(const double *__restrict__ raw, int64_t B, const double *__restrict__ mean,
                            const double *__restrict__ comp, int n_comp, const double *__restrict__ proj,
                            double *__restrict__ out) {
    int64_t b = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    int64_t i = blockIdx.y;
    if (b >= B) return;
    double rec = 0.0;
    for (int c = 0; c < n_comp; ++c) {
        double t = 0.0;
        for (int q = 0; q < PROJ_SPLIT; ++q) t += proj[(i * PROJ_SPLIT + q) * MAX_COMP + c];
        rec += t * comp[(int64_t)c * B + b];
    }
    rec += mean[b];
    out[i * B + b] = raw[i * B + b] / rec;
}

// Batches: normalisation (k_normalize's division), x / reconstruction (k_pca_apply's arithmetic, term by term) and the
// transpose into the repeats' bin-major arrays (k_transpose with its two outputs and the cleared flag words) in ONE
// launch: 32 bins x 32 samples per workgroup through LDS -- the sample-major intermediates (raw, data: 2 x 55 MB per
// 125 x 50 kb batch, written and read again) are never made.
__global__ void k_pca_apply_t(const int *__restrict__ counts, int64_t Btot, const int *__restrict__ m2g, int64_t B,
                              int64_t Ns_real, int64_t Ns, const long long *__restrict__ partial, const double *__restrict__ mean,
                              const double *__restrict__ comp, int n_comp, const double *__restrict__ proj,
                              double *__restrict__ xt, double *__restrict__ xc, int *__restrict__ zero,
                              int64_t n_zero) {
    __shared__ double tile[32][33];
    __shared__ double s_t[32][MAX_COMP], s_tot[32];
    const int tx = threadIdx.x, ty = threadIdx.y, tid = ty * 32 + tx;
    if (zero) {
        const int64_t nthreads = (int64_t)gridDim.x * gridDim.y * 256;
        const int64_t me = ((int64_t)blockIdx.y * gridDim.x + blockIdx.x) * 256 + tid;
        for (int64_t t = me; t < n_zero; t += nthreads) zero[t] = 0;
    }
    const int64_t b0 = (int64_t)blockIdx.x * 32, i0 = (int64_t)blockIdx.y * 32;
    // (samples Ns_real .. Ns - 1 are the padding of the bin-major rows to whole 128-byte lines: copies of sample 0)
    auto src = [&](const int64_t i) { return i < Ns_real ? i : 0; };
    if (tid < 32) {
        long long t = 0;
        if (i0 + tid < Ns)
#pragma unroll
            for (int q = 0; q < TOT_SPLIT; ++q) t += partial[src(i0 + tid) * TOT_SPLIT + q];
        s_tot[tid] = (double)t;
    }
    {
        const int sm = tid >> 3, c = tid & 7;                 // MAX_COMP == 8
        double t = 0.0;
        if (c < n_comp && i0 + sm < Ns)
            for (int q = 0; q < PROJ_SPLIT; ++q) t += proj[(src(i0 + sm) * PROJ_SPLIT + q) * MAX_COMP + c];
        s_t[sm][c] = t;
    }
    wc_sync();
    const int64_t b = b0 + tx;
    const double mean_b = b < B ? mean[b] : 0.0;
    const int g = b < B ? m2g[b] : 0;
    double cb[MAX_COMP];
#pragma unroll
    for (int c = 0; c < MAX_COMP; ++c) cb[c] = (c < n_comp && b < B) ? comp[(int64_t)c * B + b] : 0.0;
    for (int j = ty; j < 32; j += 8) {
        const int64_t i = i0 + j;
        if (i < Ns && b < B) {
            const double raw = (double)counts[src(i) * Btot + g] / s_tot[j];
            double rec = 0.0;
#pragma unroll
            for (int c = 0; c < MAX_COMP; ++c)
                if (c < n_comp) rec += s_t[j][c] * cb[c];
            rec += mean_b;
            tile[j][tx] = raw / rec;
        }
    }
    wc_sync();
    for (int j = ty; j < 32; j += 8) {
        const int64_t bb = b0 + j, i = i0 + tx;
        if (bb < B && i < Ns) {
            const double v = tile[tx][j];
            xt[bb * Ns + i] = v;
            xc[bb * Ns + i] = v;
        }
    }
    // row B of xc = -1.0: what a list index of -1 reads in k_zscore_tiled (was a k_fill launch in front of that kernel)
    if (blockIdx.x == 0 && ty == 0 && i0 + tx < Ns) xc[B * Ns + i0 + tx] = -1.0;
}

// Latency mode (a few samples per call): the four preparation launches as two.
// k_lat_project: one workgroup per (sample, bin slice) -- the sample's total (every slice computes it:
// 50 KB of counts, cheaper than a launch), the slice's normalised values on the fly and the
// projection partial sums with k_pca_project's arithmetic exactly (256 strided accumulators per
// slice, the same tree), so a sample comes out bit-identical to the batch kernels.
__global__ __launch_bounds__(256) void k_lat_project(const int *__restrict__ counts, int64_t Btot,
                                                     const int *__restrict__ m2g, int64_t B,
                                                     const double *__restrict__ mean, const double *__restrict__ comp,
                                                     int n_comp, double *__restrict__ totals,
                                                     double *__restrict__ proj) {
    __shared__ long long sh_t[256];
    __shared__ double sh[MAX_COMP][256];
    const int tid = threadIdx.x;
    const int64_t i = blockIdx.x;
    const int *row = counts + i * Btot;
    long long acc_t = 0;
    {
        // thirty-two loads of a trip in flight together (one memory round trip per 8192 bins: two trips at
        // 250 kb instead of twelve)
        long long a0 = 0, a1 = 0, a2 = 0, a3 = 0;
        for (int64_t g0 = tid; g0 < Btot; g0 += 8192) {
            int v[32];
#pragma unroll
            for (int e = 0; e < 32; ++e) v[e] = g0 + 256 * e < Btot ? row[g0 + 256 * e] : 0;
#pragma unroll
            for (int e = 0; e < 32; e += 4) { a0 += v[e]; a1 += v[e + 1]; a2 += v[e + 2]; a3 += v[e + 3]; }
        }
        acc_t = (a0 + a1) + (a2 + a3);
    }
    sh_t[tid] = acc_t;
    wc_sync();
    for (int o = 128; o > 0; o >>= 1) {
        if (tid < o) sh_t[tid] += sh_t[tid + o];
        wc_sync();
    }
    const double total = (double)sh_t[0];          // integer sums are exact in any order
    if (blockIdx.y == 0 && tid == 0) totals[i] = total;
    const int64_t per = (B + PROJ_SPLIT - 1) / PROJ_SPLIT;
    const int64_t b_lo = (int64_t)blockIdx.y * per, b_hi = b_lo + per < B ? b_lo + per : B;
    double acc[MAX_COMP];
#pragma unroll
    for (int c = 0; c < MAX_COMP; ++c) acc[c] = 0.0;
    // the same terms in the same order per accumulator as the loop `for b: acc[c] += d * comp[c][b]`, with
    // the loads of eight consecutive trips issued together (index, then count / mean / components)
    for (int64_t b0 = b_lo + tid; b0 < b_hi; b0 += 2048) {
        int gi[8];
        double mn[8], cp[8][MAX_COMP];
#pragma unroll
        for (int e = 0; e < 8; ++e) gi[e] = b0 + 256 * e < b_hi ? m2g[b0 + 256 * e] : -1;
#pragma unroll
        for (int e = 0; e < 8; ++e) {
            const int64_t b = b0 + 256 * e;
            const bool in = b < b_hi;
            mn[e] = in ? mean[b] : 0.0;
#pragma unroll
            for (int c = 0; c < MAX_COMP; ++c) cp[e][c] = (in && c < n_comp) ? comp[(int64_t)c * B + b] : 0.0;
        }
        int cv[8];
#pragma unroll
        for (int e = 0; e < 8; ++e) cv[e] = gi[e] >= 0 ? row[gi[e]] : 0;
#pragma unroll
        for (int e = 0; e < 8; ++e) {
            if (gi[e] < 0) continue;
            const double xb = (double)cv[e] / total;
            const double d = xb - mn[e];
#pragma unroll
            for (int c = 0; c < MAX_COMP; ++c)
                if (c < n_comp) acc[c] += d * cp[e][c];
        }
    }
#pragma unroll
    for (int c = 0; c < MAX_COMP; ++c) sh[c][tid] = acc[c];
    wc_sync();
    for (int o = 128; o > 0; o >>= 1) {
        if (tid < o)
#pragma unroll
            for (int c = 0; c < MAX_COMP; ++c)
                if (c < n_comp) sh[c][tid] += sh[c][tid + o];
        wc_sync();
    }
    if (tid < n_comp) proj[(i * PROJ_SPLIT + blockIdx.y) * MAX_COMP + tid] = sh[tid][0];
}

// k_lat_apply: normalise again (one division, the same bits), reconstruct, divide; writes the sample-major
// result and the repeats' bin-major working arrays, and clears the repeats' counters and dirty map.
__global__ void k_lat_apply(const int *__restrict__ counts, int64_t Btot, const int *__restrict__ m2g, int64_t B,
                            int64_t Ns, const double *__restrict__ totals, const double *__restrict__ mean,
                            const double *__restrict__ comp, int n_comp, const double *__restrict__ proj,
                            double *__restrict__ data, double *__restrict__ xt, double *__restrict__ xc,
                            int *__restrict__ zero, int64_t n_zero) {
    const int64_t b = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    const int64_t i = blockIdx.y;
    {
        const int64_t nthreads = (int64_t)gridDim.x * gridDim.y * blockDim.x;
        for (int64_t t = i * gridDim.x * blockDim.x + b; t < n_zero; t += nthreads) zero[t] = 0;
    }
    if (b >= B) return;
    double rec = 0.0;
    for (int c = 0; c < n_comp; ++c) {
        double t = 0.0;
        for (int q = 0; q < PROJ_SPLIT; ++q) t += proj[(i * PROJ_SPLIT + q) * MAX_COMP + c];
        rec += t * comp[(int64_t)c * B + b];
    }
    rec += mean[b];
    const double v = ((double)counts[i * Btot + m2g[b]] / totals[i]) / rec;
    data[i * B + b] = v;
    xt[b * Ns + i] = v;
    xc[b * Ns + i] = v;
}

// [R, C] -> [C, R]
// out = in^T (and out2, when given: the repeats' working copy); `zero`/`n_zero`: an int array the
// same launch clears (the repeats' flags and counts), so that no separate memset is needed
__global__ void k_transpose(const double *__restrict__ in, int64_t R, int64_t C, double *__restrict__ out,
                            double *__restrict__ out2, int *__restrict__ zero, int64_t n_zero) {
    __shared__ double tile[32][33];
    if (zero) {
        const int64_t nthreads = (int64_t)gridDim.x * gridDim.y * 256;
        const int64_t me = ((int64_t)blockIdx.y * gridDim.x + blockIdx.x) * 256 + threadIdx.y * 32 + threadIdx.x;
        for (int64_t t = me; t < n_zero; t += nthreads) zero[t] = 0;
    }
    int64_t c0 = (int64_t)blockIdx.x * 32, r0 = (int64_t)blockIdx.y * 32;
    for (int j = threadIdx.y; j < 32; j += 8) {
        int64_t r = r0 + j, c = c0 + threadIdx.x;
        if (r < R && c < C) tile[j][threadIdx.x] = in[r * C + c];
    }
    wc_sync();
    for (int j = threadIdx.y; j < 32; j += 8) {
        int64_t c = c0 + j, r = r0 + threadIdx.x;
        if (r < R && c < C) {
            out[c * R + r] = tile[threadIdx.x][j];
            if (out2) out2[c * R + r] = tile[threadIdx.x][j];
        }
    }
}

// ------------------------------------------------------------------ z-score ----
__device__ inline double combine8(const double *r) {
    return ((r[0] + r[1]) + (r[2] + r[3])) + ((r[4] + r[5]) + (r[6] + r[7]));
}

// numpy's pairwise sum over kept values v_0..v_{m-1} (m <= 128) streamed in order,
// WITHOUT knowing m in advance: values fill a pending group of eight; a complete
// group is committed to the eight strided accumulators; whatever is pending at the
// end is numpy's tail (added sequentially after the accumulators are combined).
// m < 8 degenerates to the plain left-to-right sum numpy uses there.
struct StreamSum {
    double r[8], p[8];
    int pos;
    __device__ inline void init() {
#pragma unroll
        for (int j = 0; j < 8; ++j) { r[j] = 0.0; p[j] = 0.0; }
        pos = 0;
    }
    __device__ inline void push(double v) {
        const int slot = pos & 7;
#pragma unroll
        for (int j = 0; j < 8; ++j) p[j] = (slot == j) ? v : p[j];
        if (slot == 7) {
#pragma unroll
            for (int j = 0; j < 8; ++j) r[j] = r[j] + p[j];   // first group: 0 + v == v exactly (v >= 0)
        }
        ++pos;
    }
    __device__ inline double finish() const {
        double res = ((r[0] + r[1]) + (r[2] + r[3])) + ((r[4] + r[5]) + (r[6] + r[7]));
        const int tail = pos & 7;
#pragma unroll
        for (int j = 0; j < 8; ++j)
            if (j < tail) res = res + p[j];
        return res;
    }
};

// trySample (wisetools.py:407-435) for one (bin b, sample i) pair, gid = b * Ns + i (the arrays
// are bin-major so that a wave reads 64 consecutive samples of one reference bin).
// Two passes over the kept references in numpy's pairwise order (StreamSum).  While every
// active lane of the wave has kept all of its values so far, element r goes to accumulator
// r & 7: whole groups of eight are added straight into the eight strided accumulators (eight
// independent loads in flight, no slot selection).  The first dropped value anywhere in the
// wave ends that: the rest of the list takes the general push().
__device__ inline double zscore_one(const int64_t b, const int64_t i, const int64_t gid,
                                  const double *__restrict__ XT, const double *__restrict__ XC,
                                  const int *__restrict__ gidx, const int *__restrict__ nref, int k, int64_t Ns,
                                  double *__restrict__ zT, double *__restrict__ rT, double *__restrict__ nT,
                                  double *__restrict__ sdT, const int64_t osm = 0) {
    // osm: 0 -- the outputs are bin-major like the inputs (index gid); otherwise they are SAMPLE-major with rows of
    // osm bins (index i * osm + b): what the per-sample consumers read, written by the producer (no transposes)
    const int *lst = gidx + b * k;
    const int n = nref[b];
    StreamSum acc;
    acc.init();
    int fast_end = 0;      // references [0, fast_end) were consumed in whole, fully kept groups
    {
        bool ok = true;
        while (ok) {
            const int r0 = fast_end;
            double v[8];
            bool mine = r0 + 8 <= n;
#pragma unroll
            for (int e = 0; e < 8; ++e) {
                int g = mine ? lst[r0 + e] : -1;
                v[e] = g >= 0 ? XC[(int64_t)g * Ns + i] : -1.0;
                mine = mine && (v[e] >= 0.0);
            }
            ok = __all(mine);
            if (ok) {
#pragma unroll
                for (int e = 0; e < 8; ++e) acc.r[e] = acc.r[e] + v[e];   // first group: 0 + v == v exactly (v >= 0)
                fast_end = r0 + 8;
            }
        }
        acc.pos = fast_end;
    }
    for (int r = fast_end; r < n; ++r) {
        int g = lst[r];
        double v = g >= 0 ? XC[(int64_t)g * Ns + i] : -1.0;
        if (v >= 0.0) acc.push(v);  // flagged (-1), negative and NaN values are dropped (wisetools.py:425)
    }
    const int m = acc.pos;
    const double mean = acc.finish() / (double)m;
    acc.init();
    for (int r0 = 0; r0 < fast_end; r0 += 8) {
        double v[8];
#pragma unroll
        for (int e = 0; e < 8; ++e) v[e] = XC[(int64_t)lst[r0 + e] * Ns + i];
#pragma unroll
        for (int e = 0; e < 8; ++e) {
            double dv = v[e] - mean;
            double sq = dv * dv;
            acc.r[e] = acc.r[e] + sq;
        }
    }
    acc.pos = fast_end;
    for (int r = fast_end; r < n; ++r) {
        int g = lst[r];
        double v = g >= 0 ? XC[(int64_t)g * Ns + i] : -1.0;
        if (v >= 0.0) {
            double dv = v - mean;
            double sq = dv * dv;
            acc.push(sq);
        }
    }
    const double var = acc.finish() / (double)m;
    const double sd = sqrt(var);
    const double x = XT[gid];
    const int64_t oid = osm ? i * osm + b : gid;
    const double zv = (x - mean) / sd;
    zT[oid] = zv;
    rT[oid] = x / mean;
    nT[oid] = (double)m;
    sdT[oid] = sd;
    return zv;
}

// The same for a wave that holds ONE bin and 64 samples (b wave-uniform) when every lane keeps
// all of its values, the normal case in the first repeat: the G full groups of eight stay in
// registers, so the references are gathered once instead of once per pass.
// The wave's life is a latency chain, so everything it will need is requested at once: the bin's
// whole index list by 16-index scalar loads (the list is padded with -1 beyond its length and the
// array by one such load at its end, so no load depends on the length), then every gather -- an
// index of -1 reads row 0 and is masked afterwards (no branch between a scalar load and its gather:
// one index at a time, each waited for, was ~100 dependent round trips per wave), the pair's own
// value with them.  Anything else (a dropped value anywhere in the wave, a list longer than
// 8 G + 7, a list stride that is not a multiple of four) takes zscore_one.
typedef int int16v __attribute__((ext_vector_type(16), aligned(16)));   // list rows start 16-byte aligned (k % 4 == 0)
template <int G>
__device__ inline void zscore_wave(const int b, const int64_t i, const int64_t gid, const double *__restrict__ XT,
                                   const double *__restrict__ XC, const int *__restrict__ gidx,
                                   const int *__restrict__ nref, int k, int64_t Ns, double *__restrict__ zT,
                                   double *__restrict__ rT, double *__restrict__ nT, double *__restrict__ sdT) {
    constexpr int NL = (8 * G + 7 + 15) / 16;      // 16-index loads that cover 8 G + 7 list slots
    const int *lst = gidx + (int64_t)b * k;
    const int n = nref[b];
    const int ng = n >> 3;
    if (ng > G || (k & 3)) {
        zscore_one(b, i, gid, XT, XC, gidx, nref, k, Ns, zT, rT, nT, sdT);
        return;
    }
    double v[16 * NL];
    const double x = XT[gid];
    {
        const int16v *l16 = reinterpret_cast<const int16v *>(lst);
        unsigned int absent[NL];                                    // per 16 slots: bit e set = no reference there
        const unsigned int ioff = (unsigned int)i * 8u;             // scalar row base + this lane's 32-bit byte offset
#pragma unroll
        for (int t = 0; t < NL; ++t) {
            const int16v g16 = l16[t];
            unsigned int bits = 0u;
#pragma unroll
            for (int e = 0; e < 16; ++e) {
                const int g = 16 * t + e < k ? g16[e] : -1;        // slots beyond this bin's list belong to the next bin
                bits |= g < 0 ? (1u << e) : 0u;
                const char *rowp = reinterpret_cast<const char *>(XC + (int64_t)(g < 0 ? 0 : g) * Ns);
                v[16 * t + e] = *reinterpret_cast<const double *>(rowp + ioff);
            }
            absent[t] = bits;
        }
        __builtin_amdgcn_sched_barrier(0);                          // every gather is requested before the first one is waited for
#pragma unroll
        for (int t = 0; t < NL; ++t)
#pragma unroll
            for (int e = 0; e < 16; ++e)
                if ((absent[t] >> e) & 1u) v[16 * t + e] = -1.0;
    }
    bool mine = true;
#pragma unroll
    for (int q = 0; q < G; ++q) {
        if (q < ng) {
#pragma unroll
            for (int e = 0; e < 8; ++e) mine = mine && (v[8 * q + e] >= 0.0);
        }
    }
    if (!__all(mine)) {
        zscore_one(b, i, gid, XT, XC, gidx, nref, k, Ns, zT, rT, nT, sdT);
        return;
    }
    StreamSum acc;
    acc.init();
#pragma unroll
    for (int q = 0; q < G; ++q) {
        if (q < ng) {
#pragma unroll
            for (int e = 0; e < 8; ++e) acc.r[e] = acc.r[e] + v[8 * q + e];   // first group: 0 + v == v exactly (v >= 0)
        }
    }
    acc.pos = 8 * ng;
    // the incomplete last group: slots 8 ng .. 8 ng + 6 (beyond the list: -1, dropped)
    double tailv[7];
#pragma unroll
    for (int e = 0; e < 7; ++e) tailv[e] = -1.0;
#pragma unroll
    for (int q = 0; q <= G; ++q) {
        if (q == ng) {
#pragma unroll
            for (int e = 0; e < 7; ++e) tailv[e] = v[8 * q + e];
        }
    }
#pragma unroll
    for (int e = 0; e < 7; ++e)
        if (tailv[e] >= 0.0) acc.push(tailv[e]);  // flagged (-1), negative and NaN values are dropped (wisetools.py:425)
    const int m = acc.pos;
    const double mean = acc.finish() / (double)m;
    acc.init();
#pragma unroll
    for (int q = 0; q < G; ++q) {
        if (q < ng) {
#pragma unroll
            for (int e = 0; e < 8; ++e) {
                const double dv = v[8 * q + e] - mean;
                const double sq = dv * dv;
                acc.r[e] = acc.r[e] + sq;
            }
        }
    }
    acc.pos = 8 * ng;
#pragma unroll
    for (int e = 0; e < 7; ++e)
        if (tailv[e] >= 0.0) {
            const double dv = tailv[e] - mean;
            const double sq = dv * dv;
            acc.push(sq);
        }
    const double var = acc.finish() / (double)m;
    const double sd = sqrt(var);
    zT[gid] = (x - mean) / sd;
    rT[gid] = x / mean;
    nT[gid] = (double)m;
    sdT[gid] = sd;
}

// First repeat of a batch (>= 32 samples): every (bin, sample) pair, wave-uniform mapping -- a
// wave holds one bin and 64 consecutive samples, so the bin, its reference list and the list
// length are wave-uniform: scalar loads of the indexes, scalar row base + lane offset for the
// gathers.  (Smaller batches: k_zscore_pairs over all pairs.)
__global__ __launch_bounds__(256, 2) void k_zscore(const double *__restrict__ XT, const double *__restrict__ XC,
                                                   const int *__restrict__ gidx, const int *__restrict__ nref, int k,
                                                   int64_t B, int64_t Ns, double *__restrict__ zT,
                                                   double *__restrict__ rT, double *__restrict__ nT,
                                                   double *__restrict__ sdT) {
    const int64_t n_sg = (Ns + 63) / 64;
    const int64_t wave = (int64_t)blockIdx.x * 4 + (threadIdx.x >> 6);
    const int wb = __builtin_amdgcn_readfirstlane((int)(wave / n_sg));
    if (wb >= B) return;
    const int64_t i = (wave - (int64_t)wb * n_sg) * 64 + (threadIdx.x & 63);
    if (i >= Ns) return;
    zscore_wave<12>(wb, i, (int64_t)wb * Ns + i, XT, XC, gidx, nref, k, Ns, zT, rT, nT, sdT);
}

// The first repeat with SAMPLE TILES DEALT TO XCDs (round 5).  k_zscore above lets every XCD gather rows of the whole
// [bins, samples] matrix -- 55 MB at 125 x 50 kb against 4 MB of L2 per XCD -- and a gather micro-benchmark
// (tools/micro/gather_rate.*) tops out at 8.7 TB/s for that whatever the access width; it reaches 24-26 TB/s when a
// workgroup only ever touches the 128-byte column (16 samples) blockIdx % 8 of every row: workgroups go to the XCDs
// round robin, so each XCD's L2 then holds one column of the matrix.  A wave here = FOUR bins x the 16 samples of one
// tile, one lane per pair (64 pairs per wave like k_zscore: the forms with fewer lost to the per-bin work, see
// EXPERIMENTS.md): the bin and its list are per lane (16-byte index loads, a 64-bit address per gather) instead of
// wave-uniform, everything else is zscore_wave.  Needs the sample stride to be a multiple of 16 (whole 128-byte
// columns; the caller pads).
typedef int int4v __attribute__((ext_vector_type(4), aligned(16)));
__global__ void k_fill(double *__restrict__ p, int64_t n, double value) {
    const int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (i < n) p[i] = value;
}
// x of lane `from` of the caller's row of 16 lanes (DPP row_newbcast; `from` a constant after unrolling)
__device__ inline int row_lane(const int x, const int from) {
#define WC_ROW_LANE(N) case N: return __builtin_amdgcn_mov_dpp(x, 0x150 + N, 0xf, 0xf, true);
    switch (from) {
        WC_ROW_LANE(0) WC_ROW_LANE(1) WC_ROW_LANE(2) WC_ROW_LANE(3) WC_ROW_LANE(4) WC_ROW_LANE(5) WC_ROW_LANE(6) WC_ROW_LANE(7)
        WC_ROW_LANE(8) WC_ROW_LANE(9) WC_ROW_LANE(10) WC_ROW_LANE(11) WC_ROW_LANE(12) WC_ROW_LANE(13) WC_ROW_LANE(14)
    default: return __builtin_amdgcn_mov_dpp(x, 0x15f, 0xf, 0xf, true);
    }
#undef WC_ROW_LANE
}
template <int G, int WAVES, int NL, bool SM>   // NL: 4-slot groups of the list held in registers (list stride k <= 4 NL <= 8 G + 8); SM: sample-major outputs
__global__ __launch_bounds__(64 * WAVES, 2) void k_zscore_tiled(const double *__restrict__ XT, const double *__restrict__ XC,
                                                         const int *__restrict__ gidx, const int *__restrict__ nref, int k,
                                                         int B, int Ns, double *__restrict__ zT, double *__restrict__ rT,
                                                         double *__restrict__ nT, double *__restrict__ sdT,
                                                         double thr, unsigned int *__restrict__ hits,
                                                         int *__restrict__ hit_count) {
    const int64_t osm = SM ? B : 0;
    // hits (SM only): the pairs whose |z| reaches the threshold are appended to a list -- one in a thousand -- and the
    // flag pass (k_flag_pairs over that list) no longer reads every z-score of the batch again (k_flag: 39 us of a
    // 125 x 50 kb batch, 0.18 ms of 1 000 x 50 kb).  The flags themselves cannot be set here: the other waves of this
    // launch still gather from XC, and the reference applies them after the whole repeat (wisetools.py:446).
    auto note_hit = [&](const bool on, const unsigned int pair) {
        if (!SM || !hits) return;
        const unsigned long long m = __ballot(on);
        if (!m) return;
        int base = 0;
        const int lane_ = threadIdx.x & 63;
        if (lane_ == (int)__ffsll((long long)m) - 1) base = atomicAdd(hit_count, __popcll(m));
        base = __shfl(base, __ffsll((long long)m) - 1);
        if (on) hits[base + __popcll(m & ((1ull << lane_) - 1ull))] = pair;
    };
    // SM: the four outputs are written SAMPLE-major [Ns, B] -- the layout k_clean, k_inflate and the
    // stdDevAvg kernels read -- instead of bin-major: k_transpose3 and k_transpose (0.14 ms of a 125 x 50 kb batch,
    // 0.9 ms of 1 000 x 50 kb) are gone.  A wave's 4 bins x 16 samples are 16 runs of 32 bytes then; the lanes are
    // permuted first (lane 4 s + q holds bin q of sample s) so that four neighbouring lanes write one run, and the
    // four workgroups that complete a 128-byte line are neighbours on the same XCD (their writes merge in its L2).
    // tiles 0 .. 8 F - 1: tile t belongs to XCD t % 8; the R = n_tiles % 8 tiles left over are dealt to all XCDs by
    // (tile, workgroup) items, so that no XCD gets a whole extra tile
    const int n_tiles = Ns >> 4;
    const int groups = (B + 4 * WAVES - 1) / (4 * WAVES);     // a workgroup: WAVES waves x 4 bins
    const int xcd = (int)(blockIdx.x & 7u);
    const int j = (int)(blockIdx.x >> 3);
    const int full = (n_tiles >> 3) * groups, R = n_tiles & 7;
    int tile, grp;
    if (j < full) {
        tile = xcd + 8 * (j / groups);
        grp = j % groups;
    } else {
        const int m = (j - full) * 8 + xcd;
        if (m >= R * groups) return;
        tile = (n_tiles & ~7) + m % R;
        grp = m / R;
    }
    const int lane = threadIdx.x & 63;
    const int b0 = (grp * WAVES + (threadIdx.x >> 6)) * 4;
    if (b0 >= B) return;
    const int q2 = lane >> 4, sm = lane & 15;
    const bool live = b0 + q2 < B;
    const int b = live ? b0 + q2 : B - 1;
    const int64_t i = (int64_t)tile * 16 + sm, gid = (int64_t)b * Ns + i;
    if (k & 3) {
        double zf = 0.0;
        if (live) zf = zscore_one(b, i, gid, XT, XC, gidx, nref, k, Ns, zT, rT, nT, sdT, osm);
        note_hit(live && fabs(zf) >= thr, (unsigned int)gid);
        return;
    }
    const int n = nref[b];                         // (not needed before the sums: the list is read whatever its length)
    double v[4 * NL];
    const double x = XT[gid];
    {
        // the 16 lanes of a bin share its list: lane sm holds slots 8 sm .. 8 sm + 7 (two 16-byte loads), a gather
        // fetches its index from its owner by a row broadcast (DPP) -- 8 index registers per lane instead of 8 G + 8.
        // An index of -1 (the list's padding; a reference numpy would reject) reads ROW B of XC, which the caller has
        // filled with -1.0: the value is then dropped like a flagged one, no masks to keep.
        const int4v *l4 = reinterpret_cast<const int4v *>(gidx + (int64_t)b * k) + 2 * sm;
        const int4v ga = l4[0], gb = l4[1];
        const char *base = reinterpret_cast<const char *>(XC);      // scalar base + a 32-bit byte offset per gather (the caller checks the size)
        const unsigned int ioff = (unsigned int)i * 8u, row_bytes = (unsigned int)Ns * 8u;
        int own[8] = {ga[0], ga[1], ga[2], ga[3], gb[0], gb[1], gb[2], gb[3]};
#pragma unroll
        for (int e = 0; e < 8; ++e) {                               // the owner turns its indexes into rows' byte offsets once
            const unsigned int g = (unsigned int)own[e];
            own[e] = (int)__umul24(g < (unsigned int)B ? g : (unsigned int)B, row_bytes);   // (both below 2^24: the caller checks)
        }
#pragma unroll
        for (int r = 0; r < 4 * NL; ++r) {
            v[r] = *reinterpret_cast<const double *>(base + ((unsigned int)row_lane(own[r & 7], r >> 3) + ioff));   // from lane r >> 3 of the row
            __builtin_amdgcn_sched_barrier(0);                      // (offsets computed ahead of their gathers end up in scratch)
        }
    }
    // flagged (-1), negative, NaN: dropped (wisetools.py:425) -- the general form does that; told by the high words:
    // a non-negative finite value's is below 0x7ff00000 as an unsigned number (infinities go the general way too)
    const int ng = n >> 3;
    unsigned int top = ng <= G ? 0u : ~0u;         // (a longer list: the general form)
#pragma unroll
    for (int q = 0; q < G; ++q)
        if (q < ng) {
#pragma unroll
            for (int e = 0; e < 8; ++e) top = max(top, (unsigned int)__double2hiint(v[8 * q + e]));
        }
    if (!__all(top < 0x7ff00000u || !live)) {
        double zf = 0.0;
        if (live) zf = zscore_one(b, i, gid, XT, XC, gidx, nref, k, Ns, zT, rT, nT, sdT, osm);
        note_hit(live && fabs(zf) >= thr, (unsigned int)gid);
        return;
    }
    // numpy's pairwise sum of m = 8 ng + (kept tail values) numbers: eight strided accumulators over the whole groups,
    // combined as a tree, then the tail's values one after the other (StreamSum with nothing dropped in the groups)
    double r8[8];
#pragma unroll
    for (int e = 0; e < 8; ++e) r8[e] = 0.0;
#pragma unroll
    for (int q = 0; q < G; ++q)
        if (q < ng) {
#pragma unroll
            for (int e = 0; e < 8; ++e) r8[e] = r8[e] + v[8 * q + e];   // first group: 0 + v == v exactly (v >= 0)
        }
    // the incomplete last group: slots 8 ng .. n - 1
    double tailv[7];
#pragma unroll
    for (int e = 0; e < 7; ++e) tailv[e] = -1.0;
#pragma unroll
    for (int q = 0; q <= G; ++q)
        if (q == ng) {
#pragma unroll
            for (int e = 0; e < 7; ++e) tailv[e] = (e < (n & 7) && 8 * q + e < 4 * NL) ? v[8 * q + e < 4 * NL ? 8 * q + e : 0] : -1.0;
        }
    int m = 8 * ng;
    double sum = ((r8[0] + r8[1]) + (r8[2] + r8[3])) + ((r8[4] + r8[5]) + (r8[6] + r8[7]));
#pragma unroll
    for (int e = 0; e < 7; ++e)
        if (tailv[e] >= 0.0) {                    // flagged (-1), negative and NaN values are dropped (wisetools.py:425)
            sum = sum + tailv[e];
            ++m;
        }
    const double mean = sum / (double)m;
#pragma unroll
    for (int e = 0; e < 8; ++e) r8[e] = 0.0;
#pragma unroll
    for (int q = 0; q < G; ++q)
        if (q < ng) {
#pragma unroll
            for (int e = 0; e < 8; ++e) {
                const double dv = v[8 * q + e] - mean;
                const double sq = dv * dv;
                r8[e] = r8[e] + sq;
            }
        }
    sum = ((r8[0] + r8[1]) + (r8[2] + r8[3])) + ((r8[4] + r8[5]) + (r8[6] + r8[7]));
#pragma unroll
    for (int e = 0; e < 7; ++e)
        if (tailv[e] >= 0.0) {
            const double dv = tailv[e] - mean;
            const double sq = dv * dv;
            sum = sum + sq;
        }
    const double var = sum / (double)m;
    const double sd = sqrt(var);
    if (!SM) {
        if (live) {
            zT[gid] = (x - mean) / sd;
            rT[gid] = x / mean;
            nT[gid] = (double)m;
            sdT[gid] = sd;
        }
        return;
    }
    const double zmine = (x - mean) / sd;
    note_hit(live && fabs(zmine) >= thr, (unsigned int)gid);
    // sample-major: destination lane d = 4 s + q takes the results of source lane 16 q + s
    const int src = ((lane & 3) << 4) | (lane >> 2);
    const double zo = __shfl(zmine, src), ro = __shfl(x / mean, src), no = __shfl((double)m, src),
                 so = __shfl(sd, src);
    const int ob = b0 + (lane & 3);
    if (ob < B) {
        const int64_t oid = ((int64_t)tile * 16 + (lane >> 2)) * osm + ob;
        zT[oid] = zo;
        rT[oid] = ro;
        nT[oid] = no;
        sdT[oid] = so;
    }
}

// numpy's pairwise sum of a stream of kept values by an aligned group of eight lanes: lane s
// owns numpy's accumulator r[s].  Kept values get consecutive stream indexes c; the value with
// index c belongs to lane c & 7, which holds it as `pend` until its group of eight is known to
// be complete (the next value for that lane arrives, or the stream ends with >= 8 (c/8 + 1)
// values) -- the incomplete last group is numpy's tail, added sequentially after the
// accumulators are combined.  Fewer than eight values degenerate to numpy's plain
// left-to-right sum.  All eight lanes return the sum.
struct GroupSum {
    double r, pend;
    int pend_idx, pos;      // stream index of `pend` (-1: none), values seen so far
    __device__ inline void init() { r = 0.0; pend = 0.0; pend_idx = -1; pos = 0; }
    // one trip: lane j of the group offers `x` (kept or not); `sub` = lane & 7, `gbase` = lane & ~7
    __device__ inline void trip(double x, bool kept, int sub, int gbase) {
        const unsigned int mask = (unsigned int)(__ballot(kept) >> gbase) & 0xFFu;
        const int cnt = __popc(mask);
        const int q = (sub - pos) & 7;          // this lane's value is the q-th kept one of the trip, if any
        int src = 0, seen = 0;
#pragma unroll
        for (int e = 0; e < 8; ++e) {
            const bool on = (mask >> e) & 1u;
            src = (on && seen == q) ? e : src;
            seen += on;
        }
        const double val = __shfl(x, gbase + src);
        if (q < cnt) {
            if (pend_idx >= 0) r = r + pend;    // its group is complete: a later value for this lane exists
            pend = val;
            pend_idx = pos + q;
        }
        pos += cnt;
    }
    __device__ inline double finish(int sub, int gbase) {
        const int body = pos & ~7;
        if (pend_idx >= 0 && pend_idx < body) { r = r + pend; pend_idx = -1; }
        double res = r + __shfl_xor(r, 1);
        res = res + __shfl_xor(res, 2);
        res = res + __shfl_xor(res, 4);
        const int tail = pos & 7;
        for (int e = 0; e < tail; ++e) res = res + __shfl(pend, gbase + e);
        return res;
    }
};

// Later repeats: flags only accumulate, so a (bin, sample) pair whose references got no new
// flag in the previous repeat has exactly the same reference set and the same results bit for
// bit.  k_flag queued the pairs that did (through the reverse reference lists, each pair once:
// the `dirty` bit); only those are recomputed.  Eight lanes per pair: each loads every eighth
// reference (all of a pair's loads in flight at once; the values stay in registers for the
// second pass) and the group sums in numpy's order (GroupSum).
// pairs == nullptr: every pair 0 .. n_all - 1 (first repeat of a small batch, where whole waves
// per bin would be mostly empty and one thread per pair waits out ~26 dependent gathers).
// one (bin, sample) pair by an aligned group of eight lanes (sub = lane & 7, gbase = lane & ~7)
__device__ __forceinline__ void zscore_pair8(const unsigned int gid, const int sub, const int gbase,
                                    const double *__restrict__ XT, const double *__restrict__ XC,
                                    const int *__restrict__ gidx, const int *__restrict__ nref, int k, int64_t Ns,
                                    double *__restrict__ zT, double *__restrict__ rT, double *__restrict__ nT,
                                    double *__restrict__ sdT, const int64_t osm = 0) {
    const int64_t b = gid / Ns, i = gid - b * Ns;
    const int *lst = gidx + b * k;
    const int n = nref[b];
    // references 8 t + sub, t = 0..15 (a reference list holds at most 128 entries here; longer
    // lists take k_zscore_big); beyond the list: dropped
    double v[16];
    {
        int g[16];
#pragma unroll
        for (int t8 = 0; t8 < 16; ++t8) {
            const int r = 8 * t8 + sub;
            g[t8] = r < n ? lst[r] : -1;
        }
#pragma unroll
        for (int t8 = 0; t8 < 16; ++t8) v[t8] = g[t8] >= 0 ? XC[(int64_t)g[t8] * Ns + i] : -1.0;
    }
    // Nothing dropped in this pair (the usual case before any flag): kept value number c is reference c,
    // so lane `sub` IS numpy's accumulator r[sub] -- the body of the list is a plain chain of adds per
    // lane and the stream bookkeeping of GroupSum (ballots, lane searches, shuffles per trip) is not
    // needed; combine and tail are GroupSum::finish's.
    bool all_kept = true;
#pragma unroll
    for (int t8 = 0; t8 < 16; ++t8) all_kept = all_kept && (8 * t8 + sub >= n || v[t8] >= 0.0);
    const bool group_kept = ((unsigned int)(__ballot(all_kept) >> gbase) & 0xFFu) == 0xFFu;
    int m;
    double mean, var;
    if (group_kept) {
        const int nb8 = n >> 3, tail = n & 7;
        auto group_sum = [&](auto term) {
            double r = 0.0, pend = 0.0;
#pragma unroll
            for (int t8 = 0; t8 < 16; ++t8) {
                const double x = term(t8);
                if (t8 < nb8) r = r + x;
                pend = t8 == nb8 ? x : pend;
            }
            double res = r + __shfl_xor(r, 1);
            res = res + __shfl_xor(res, 2);
            res = res + __shfl_xor(res, 4);
            for (int e = 0; e < tail; ++e) res = res + __shfl(pend, gbase + e);
            return res;
        };
        m = n;
        mean = group_sum([&](int t8) { return v[t8]; }) / (double)m;
        var = group_sum([&](int t8) {
                  const double dv = v[t8] - mean;
                  const double sq = dv * dv;
                  return sq;
              }) / (double)m;
    } else {
        GroupSum acc;
        acc.init();
#pragma unroll
        for (int t8 = 0; t8 < 16; ++t8)      // (a predicate, not a break: the loop unrolls and v[] stays in registers)
            if (8 * t8 < n) acc.trip(v[t8], v[t8] >= 0.0, sub, gbase);   // flagged (-1), negative and NaN values are dropped (wisetools.py:425)
        m = acc.pos;
        mean = acc.finish(sub, gbase) / (double)m;
        acc.init();
#pragma unroll
        for (int t8 = 0; t8 < 16; ++t8)
            if (8 * t8 < n) {
                const double dv = v[t8] - mean;
                const double sq = dv * dv;
                acc.trip(sq, v[t8] >= 0.0, sub, gbase);
            }
        var = acc.finish(sub, gbase) / (double)m;
    }
    if (sub == 0) {
        const double sd = sqrt(var);
        const double x = XT[gid];
        const int64_t oid = osm ? i * osm + b : (int64_t)gid;       // (osm: sample-major outputs, see zscore_one)
        zT[oid] = (x - mean) / sd;
        rT[oid] = x / mean;
        nT[oid] = (double)m;
        sdT[oid] = sd;
    }
}

__global__ __launch_bounds__(256) void k_zscore_pairs(const unsigned int *__restrict__ pairs,
                                                      const int *__restrict__ count, int64_t n_all,
                                                      unsigned int *__restrict__ dirty,
                                                      const double *__restrict__ XT, const double *__restrict__ XC,
                                                      const int *__restrict__ gidx, const int *__restrict__ nref,
                                                      int k, int64_t Ns, double *__restrict__ zT,
                                                      double *__restrict__ rT, double *__restrict__ nT,
                                                      double *__restrict__ sdT, int64_t osm) {
    const int64_t n_pairs = pairs ? (int64_t)*count : n_all;
    const int lane = threadIdx.x & 63, sub = lane & 7, gbase = lane & ~7;
    for (int64_t t = ((int64_t)blockIdx.x * 256 + threadIdx.x) >> 3; t < n_pairs; t += (int64_t)gridDim.x * 32) {
        const unsigned int gid = pairs ? pairs[t] : (unsigned int)t;
        if (pairs && sub == 0) atomicAnd(&dirty[gid >> 5], ~(1u << (gid & 31)));
        zscore_pair8(gid, sub, gbase, XT, XC, gidx, nref, k, Ns, zT, rT, nT, sdT, osm);
    }
}

// Reference lists longer than 128 entries (-refsize above 128): numpy's pairwise sum is then a
// tree whose shape depends on the NUMBER of kept values, so the kept values are first compacted, in
// list order, into an LDS buffer and then summed with wc::pairwise_sum over that buffer (mean,
// then the squared deviations).  One wave per (bin, sample) pair, grid-strided; `pairs` as in
// k_zscore_pairs (NULL: every pair).  Slower than the <= 128 kernels, same bits as numpy.
constexpr int BIG_K = 1024;         // longest reference list (LDS buffer of a wave)
__global__ __launch_bounds__(256) void k_zscore_big(const unsigned int *__restrict__ pairs,
                                                    const int *__restrict__ count, int64_t n_all,
                                                    unsigned int *__restrict__ dirty,
                                                    const double *__restrict__ XT, const double *__restrict__ XC,
                                                    const int *__restrict__ gidx, const int *__restrict__ nref,
                                                    int k, int64_t Ns, double *__restrict__ zT,
                                                    double *__restrict__ rT, double *__restrict__ nT,
                                                    double *__restrict__ sdT) {
    __shared__ double kept[4][BIG_K];
    const int64_t n_pairs = pairs ? (int64_t)*count : n_all;
    const int lane = threadIdx.x & 63, w = threadIdx.x >> 6, sub = lane & 7;
    double *buf = kept[w];
    for (int64_t t = (int64_t)blockIdx.x * 4 + w; t < n_pairs; t += (int64_t)gridDim.x * 4) {
        const unsigned int gid = pairs ? pairs[t] : (unsigned int)t;
        if (pairs && lane == 0) atomicAnd(&dirty[gid >> 5], ~(1u << (gid & 31)));
        const int64_t b = gid / Ns, i = gid - b * Ns;
        const int *lst = gidx + b * k;
        const int n = nref[b];
        int m = 0;
        __builtin_amdgcn_wave_barrier();
        for (int base = 0; base < n; base += 64) {
            const int r = base + lane;
            const int g = r < n ? lst[r] : -1;
            const double v = g >= 0 ? XC[(int64_t)g * Ns + i] : -1.0;
            const bool keep = v >= 0.0;          // flagged (-1), negative and NaN values are dropped (wisetools.py:425)
            const unsigned long long mask = __ballot(keep);
            if (keep) buf[m + __popcll(mask & ((1ull << lane) - 1ull))] = v;
            m += __popcll(mask);
        }
        __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
        __builtin_amdgcn_wave_barrier();
        __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
        // (m <= BIG_K = 1 024 kept values: numpy's tree is at most four splits deep -- unrolled at compile time; the
        //  generic walk with its runtime-indexed stack took 498 registers and 304 bytes of scratch per lane here)
        const double mean = wc::pw_node_group8<4>([&](int64_t e) { return buf[e]; }, 0, m, sub) / (double)m;
        const double var = wc::pw_node_group8<4>(
                               [&](int64_t e) {
                                   const double dv = buf[e] - mean;
                                   const double sq = dv * dv;
                                   return sq;
                               },
                               0, m, sub) / (double)m;
        if (lane == 0) {
            const double sd = sqrt(var);
            const double x = XT[gid];
            zT[gid] = (x - mean) / sd;
            rT[gid] = x / mean;
            nT[gid] = (double)m;
            sdT[gid] = sd;
        }
    }
}

// testCopy[abs(z) >= threshold] = -1 (wisetools.py:446).  A NEW flag on (bin g, sample i)
// queues every bin that uses g as a reference, for the same sample, for the next repeat: the
// wave expands its new flags one after the other, 64 users per trip.
// (zv, xcv: this lane's z-score and working value, already loaded)
__device__ inline void flag_wave_vals(const int64_t gid, const bool valid, const double zv, const double xcv, double thr,
                                      int64_t Ns, double *__restrict__ XC, const int *__restrict__ users_off,
                                      const int *__restrict__ users, unsigned int *__restrict__ dirty,
                                      unsigned int *__restrict__ next_pairs, int *__restrict__ next_count) {
    const int lane = threadIdx.x & 63;
    bool hit = false;
    if (valid && fabs(zv) >= thr && xcv != -1.0) {
        XC[gid] = -1.0;
        hit = next_pairs != nullptr;      // last repeat: nothing follows
    }
    unsigned long long todo = __ballot(hit);
    while (todo) {
        const int l = __ffsll((long long)todo) - 1;
        todo &= todo - 1;
        const int64_t fg = __shfl(gid, l);
        const int64_t g = fg / Ns, i = fg - g * Ns;
        const int u1 = users_off[g + 1];
        for (int u0 = users_off[g]; u0 < u1; u0 += 64) {
            const int u = u0 + lane;
            unsigned int ug = 0u;
            bool won = false;        // this lane set the pair's bit: it queues the pair
            if (u < u1) {
                ug = (unsigned int)((int64_t)users[u] * Ns + i);
                const unsigned int bit = 1u << (ug & 31);
                won = !(atomicOr(&dirty[ug >> 5], bit) & bit);
            }
            const unsigned long long winners = __ballot(won);
            if (winners) {            // one reservation per trip
                int base = 0;
                if (lane == 0) base = atomicAdd(next_count, __popcll(winners));
                base = __shfl(base, 0);
                if (won) next_pairs[base + __popcll(winners & ((1ull << lane) - 1ull))] = ug;
            }
        }
    }
}

__device__ inline void flag_wave(const int64_t gid, const bool valid, const double *__restrict__ zT, double thr,
                                 int64_t Ns, double *__restrict__ XC, const int *__restrict__ users_off,
                                 const int *__restrict__ users, unsigned int *__restrict__ dirty,
                                 unsigned int *__restrict__ next_pairs, int *__restrict__ next_count,
                                 const int64_t osm = 0) {
    int64_t zid = gid;
    if (osm) {                                    // sample-major z (see zscore_one)
        const int64_t b = gid / Ns;
        zid = (gid - b * Ns) * osm + b;
    }
    flag_wave_vals(gid, valid, valid ? zT[zid] : 0.0, valid ? XC[gid] : -1.0, thr, Ns, XC, users_off, users, dirty,
                   next_pairs, next_count);
}

__global__ __launch_bounds__(256) void k_flag(const double *__restrict__ zT, double thr, int64_t n, int64_t Ns,
                                              double *__restrict__ XC, const int *__restrict__ users_off,
                                              const int *__restrict__ users, unsigned int *__restrict__ dirty,
                                              unsigned int *__restrict__ next_pairs, int *__restrict__ next_count) {
    const int64_t gid = (int64_t)blockIdx.x * 256 + threadIdx.x;
    flag_wave(gid, gid < n, zT, thr, Ns, XC, users_off, users, dirty, next_pairs, next_count);
}
// the same over SAMPLE-major z-scores [Ns, B]: workgroup (x, i) reads 256 consecutive bins of sample i; the working
// value of a pair (bin-major, a line of its own per lane) is only read where |z| reaches the threshold
__global__ __launch_bounds__(256) void k_flag_sm(const double *__restrict__ zS, double thr, int64_t B, int64_t Ns,
                                                 double *__restrict__ XC, const int *__restrict__ users_off,
                                                 const int *__restrict__ users, unsigned int *__restrict__ dirty,
                                                 unsigned int *__restrict__ next_pairs, int *__restrict__ next_count) {
    const int64_t b = (int64_t)blockIdx.x * 256 + threadIdx.x, i = blockIdx.y;
    const bool valid = b < B;
    const double zv = valid ? zS[i * B + b] : 0.0;
    const int64_t gid = valid ? b * Ns + i : 0;
    const double xcv = (valid && fabs(zv) >= thr) ? XC[gid] : -1.0;
    flag_wave_vals(gid, valid, zv, xcv, thr, Ns, XC, users_off, users, dirty, next_pairs, next_count);
}

// the same over the pairs this repeat recomputed (nothing else can have changed)
__global__ __launch_bounds__(256) void k_flag_pairs(const unsigned int *__restrict__ pairs,
                                                    const int *__restrict__ count, const double *__restrict__ zT,
                                                    double thr, int64_t Ns, double *__restrict__ XC,
                                                    const int *__restrict__ users_off, const int *__restrict__ users,
                                                    unsigned int *__restrict__ dirty,
                                                    unsigned int *__restrict__ next_pairs,
                                                    int *__restrict__ next_count, int64_t osm) {
    const int64_t n = *count;
    const int lane = threadIdx.x & 63;
    for (int64_t t0 = (int64_t)blockIdx.x * 256 + (threadIdx.x & ~63); t0 < n; t0 += (int64_t)gridDim.x * 256) {
        const int64_t t = t0 + lane;
        flag_wave(t < n ? (int64_t)pairs[t] : 0, t < n, zT, thr, Ns, XC, users_off, users, dirty, next_pairs,
                  next_count, osm);
    }
}

// The first repeat's flags from the tiled kernel's hit list: a WAVE per listed pair (k_flag_pairs holds a pair per
// lane and expands a wave's hits one after the other -- 64 serial expansions per wave when the list is nothing but
// hits: 40 us for the ~7 000 hits of a 125 x 50 kb batch).
__global__ __launch_bounds__(256) void k_flag_hits(const unsigned int *__restrict__ hits, const int *__restrict__ count,
                                                   const double *__restrict__ zT, double thr, int64_t Ns,
                                                   double *__restrict__ XC, const int *__restrict__ users_off,
                                                   const int *__restrict__ users, unsigned int *__restrict__ dirty,
                                                   unsigned int *__restrict__ next_pairs, int *__restrict__ next_count,
                                                   int64_t osm) {
    const int n = *count;
    const int lane = threadIdx.x & 63;
    const int n_waves = (int)gridDim.x * 4;
    for (int t = (int)blockIdx.x * 4 + (threadIdx.x >> 6); t < n; t += n_waves) {
        const int64_t gid = hits[t];
        const int64_t g = gid / Ns, i = gid - g * Ns;
        const double zv = zT[osm ? i * osm + g : gid];
        if (!(fabs(zv) >= thr) || XC[gid] == -1.0) continue;        // (wave-uniform)
        // every lane has read XC[gid] before any lane writes it: the loads above return before the store is issued
        __builtin_amdgcn_s_waitcnt(0x0070);
        __builtin_amdgcn_wave_barrier();
        if (lane == 0) XC[gid] = -1.0;
        if (!next_pairs) continue;                                  // last repeat: nothing follows
        const int u1 = users_off[g + 1];
        for (int u0 = users_off[g]; u0 < u1; u0 += 64) {
            const int u = u0 + lane;
            unsigned int ug = 0u;
            bool won = false;        // this lane set the pair's bit: it queues the pair
            if (u < u1) {
                ug = (unsigned int)((int64_t)users[u] * Ns + i);
                const unsigned int bit = 1u << (ug & 31);
                won = !(atomicOr(&dirty[ug >> 5], bit) & bit);
            }
            const unsigned long long winners = __ballot(won);
            if (winners) {            // one reservation per trip
                int base = 0;
                if (lane == 0) base = atomicAdd(next_count, __popcll(winners));
                base = __shfl(base, 0);
                if (won) next_pairs[base + __popcll(winners & ((1ull << lane) - 1ull))] = ug;
            }
        }
    }
}

// Latency mode: repeats 2 .. `repeats` in ONE launch by one workgroup.  After the first repeat a
// sample usually has a handful of new flags, i.e. some hundred queued pairs, then none: a launch per
// repeat and step costs more than the work.  Per repeat: recompute the queued pairs (eight lanes
// each, as k_zscore_pairs), workgroup barrier, look for new flags among them and queue their
// users (as k_flag_pairs), barrier.  More queued pairs than LAT_PAIR_CAP in a repeat: *overflow
// is raised and the caller repeats the call on the general path.
constexpr int LAT_PAIR_CAP = 1 << 16;
constexpr int TAIL_PAIR_CAP = 2048;             // batches: pairs a late repeat may hold for the one-workgroup form (16 trips)
__global__ __launch_bounds__(1024) void k_lat_repeats(unsigned int *__restrict__ pairs_a,
                                                      unsigned int *__restrict__ pairs_b, int *__restrict__ pair_counts,
                                                      int repeats, unsigned int *__restrict__ dirty,
                                                      const double *__restrict__ XT, double *__restrict__ XC,
                                                      const int *__restrict__ gidx, const int *__restrict__ nref, int k,
                                                      int64_t Ns, double thr, const int *__restrict__ users_off,
                                                      const int *__restrict__ users, double *__restrict__ zT,
                                                      double *__restrict__ rT, double *__restrict__ nT,
                                                      double *__restrict__ sdT, int *__restrict__ overflow,
                                                      int first_n, int it0 = 1, int64_t osm = 0, int cap = LAT_PAIR_CAP) {
    // (batches: repeats it0 + 1 .. of a whole batch by this ONE workgroup -- after the second repeat there is as a rule
    //  nothing queued at all (125 x 50 kb: 12 688 threshold hits, 3 280 pairs for repeat 2, then 0, 0, 0), and two
    //  launches per empty repeat were 43 us of a 327 us batch; more than `cap` pairs: *overflow, the caller repeats the
    //  batch with a launch pair per repeat)
    const int tid = threadIdx.x, lane = tid & 63, sub = lane & 7, gbase = lane & ~7;
    if (first_n > 0) {
        // the flag pass of the first repeat (k_flag's job) for up to 16 384 pairs: the z-scores and working
        // values of all of a thread's pairs are requested together, one memory round trip
        constexpr int FT = 16;
        double zv[FT], xv[FT];
#pragma unroll
        for (int e = 0; e < FT; ++e) {
            const int t = e * 1024 + tid;
            zv[e] = t < first_n ? zT[t] : 0.0;
            xv[e] = t < first_n ? XC[t] : -1.0;
        }
#pragma unroll
        for (int e = 0; e < FT; ++e) {
            const int t = e * 1024 + tid;
            if (e * 1024 < first_n)
                flag_wave_vals(t, t < first_n, zv[e], xv[e], thr, Ns, XC, users_off, users, dirty, pairs_a, pair_counts + 1);
        }
        __threadfence_block();
        wc_sync();
    }
    for (int it = it0; it < repeats; ++it) {
        unsigned int *cur = (it & 1) ? pairs_a : pairs_b;
        unsigned int *next = it + 1 < repeats ? ((it & 1) ? pairs_b : pairs_a) : nullptr;
        const int n = pair_counts[it];
        if (n == 0) return;                      // nothing queued: every later repeat is identical
        if (n > cap) {
            if (tid == 0) *overflow = 1;
            return;
        }
        for (int t0 = 0; t0 < n; t0 += 128) {            // 128 pairs per trip, whole waves stay together
            const int t = t0 + (tid >> 3);
            if (t < n) {
                const unsigned int gid = cur[t];
                if (sub == 0) atomicAnd(&dirty[gid >> 5], ~(1u << (gid & 31)));
                zscore_pair8(gid, sub, gbase, XT, XC, gidx, nref, k, Ns, zT, rT, nT, sdT, osm);
            }
        }
        __threadfence_block();
        wc_sync();
        for (int t0 = 0; t0 < n; t0 += 1024) {
            const int t = t0 + tid;
            flag_wave(t < n ? (int64_t)cur[t] : 0, t < n, zT, thr, Ns, XC, users_off, users, dirty, next,
                      pair_counts + it + 1, osm);
        }
        __threadfence_block();
        wc_sync();
    }
}

// stdDevAvg (wisetools.py:428-435): the reference adds the non-NaN sds bin by bin in a
// Python loop, i.e. a strictly sequential sum per sample.  One lane per sample does the
// adds; all 256 threads of the workgroup stream 128-bin chunks of sds through LDS (the
// next chunk's loads in flight during the adds), so the chain is add-latency bound, not
// load-latency bound.  Skipped (NaN) terms add +0.0, exact for a sum of sds >= 0.
// SPB samples per workgroup (64, or 16 for latency mode): the 256 threads stage chunks of
// 128 * 64 / SPB bins, i.e. few samples get long chunks, whose sums cover the latency of the
// next chunk's loads (one sample: 512-bin chunks, 22 load round trips instead of 87).
template <int SPB>
__global__ __launch_bounds__(256) void k_sd_avg(const double *__restrict__ sdT, int64_t B, int64_t Ns,
                                                double *__restrict__ out, const int *__restrict__ only,
                                                double *__restrict__ out2, int64_t sb, int64_t si) {
    // (sb, si): strides per bin and per sample -- (Ns, 1) for the bin-major array, (1, B) for the sample-major one
    constexpr int PH = 256 / SPB;                 // bin phases: thread (ss, bq) stages bins bq, bq + PH, ...
    constexpr int CH = 32 * PH;                   // bins per chunk
    __shared__ double buf[CH][SPB];
    __shared__ int cnts[PH][SPB];
    const int ss = threadIdx.x % SPB, bq = threadIdx.x / SPB;
    const int64_t i = (int64_t)blockIdx.x * SPB + ss;
    // `only`: per-sample flags of k_sd_fast -- just the samples it could not finish are summed here
    if (only && !wc_sync_or(i < Ns && only[i] != 0)) return;
    const bool live = i < Ns && (!only || only[i] != 0);
    // The staging threads do everything that is not the chain: NaN terms (and bins past the
    // end) become +0.0 -- exact for a sum of sds >= +0 -- and are left out of the count, so the
    // adding lanes issue one LDS read and one add per bin.
    double pre[32];
    int cnt = 0;
    auto fetch = [&](int64_t base) {
#pragma unroll
        for (int u = 0; u < 32; ++u) {
            const int64_t b = base + bq + PH * u;
            const double x = (live && b < B) ? sdT[b * sb + i * si] : NAN;
            const bool ok = x == x;
            cnt += ok;
            pre[u] = ok ? x : 0.0;
        }
    };
    fetch(0);
    double s = 0.0;
    for (int64_t b0 = 0; b0 < B; b0 += CH) {
        wc_sync();
#pragma unroll
        for (int u = 0; u < 32; ++u) buf[bq + PH * u][ss] = pre[u];
        wc_sync();
        if (b0 + CH < B) fetch(b0 + CH);
        if (threadIdx.x < SPB) {
            for (int bb = 0; bb < CH; bb += 16) {
                double v[16];
#pragma unroll
                for (int u = 0; u < 16; ++u) v[u] = buf[bb + u][ss];
#pragma unroll
                for (int u = 0; u < 16; ++u) s = s + v[u];
            }
        }
    }
    cnts[bq][ss] = cnt;
    wc_sync();
    if (threadIdx.x < SPB && live) {
        long long c = 0;
        for (int q = 0; q < PH; ++q) c += cnts[q][ss];
        out[i] = s / (double)c;
        if (out2) out2[i] = s / (double)c;
    }
}

// stdDevAvg without the serial chain.  The reference's sum is sequential, s <- fl(s + x_b) bin by
// bin, and 11 087 dependent float64 adds take ~100 us on one lane -- the longest thing on a single
// sample's critical path.  The same BITS can be had in parallel: while s stays inside one binade
// [2^e, 2^(e+1)), with u = 2^(e-52) the ulp there, S = s / u is an integer and
//     fl(s + x) = u (S + q + c),   q = floor(x / u), r = x / u - q,
//     c = 1 if r > 1/2, 0 if r < 1/2, and on a tie (r = 1/2) whatever makes S + q + c even.
// So every element is a map S -> S + A + H[S & 1] (ties depend on the parity of S only), such maps
// are closed under composition -- (A1, H1) then (A2, H2) is (A1 + A2, p -> H1[p] + H2[(p + A1 + H1[p]) & 1])
// -- and composition is associative: a segmented parallel scan composes all elements of a binade
// at once.  The binades come from an ordinary parallel prefix sum (the few elements at which its
// exponent changes are "boundary" elements, added one by one in plain float64 by one thread, which
// also checks that the exact running sum really has the exponent the segment assumed and really
// stayed below 2^53 units).  Any check that fails, a non-finite term, more than SD_MAX_BOUND
// boundaries: fail[sample] is raised and the serial kernel (k_sd_avg) computes that sample.
// One workgroup of 1024 threads per sample; NaN terms are skipped as in the reference
// (wisetools.py:428-430).
constexpr int SD_MAX_BOUND = 96;
struct SdMap { long long A; int H0, H1; };
__device__ inline SdMap sd_compose(const SdMap &f, const SdMap &g) {     // f first, then g
    SdMap r;
    r.A = f.A + g.A;
    const int p0 = (int)((0 + f.A + f.H0) & 1), p1 = (int)((1 + f.A + f.H1) & 1);
    r.H0 = f.H0 + (p0 ? g.H1 : g.H0);
    r.H1 = f.H1 + (p1 ? g.H1 : g.H0);
    return r;
}
__device__ inline int sd_exponent(double v) {          // unbiased exponent of a positive normal double
    return (int)((__double_as_longlong(v) >> 52) & 0x7FF) - 1023;
}
// The map of adding x > 0 to a sum that stays in the binade of unbiased exponent eb (so x < 2^eb): with
// u = 2^(eb - 52) the sum's unit, x / u = q + r, q integer; round to nearest adds q (+ 1 when r > 1/2),
// and a tie (r = 1/2) goes to the even neighbour, which depends on the sum's parity.  All in integer
// arithmetic on x's significand: x = mant * 2^(E - 1075), x / u = mant / 2^s with s = (eb + 1023) - E >= 1.
__device__ inline SdMap sd_element_map(double xv, int eb) {
    const unsigned long long bits = (unsigned long long)__double_as_longlong(xv);
    const int ex = (int)(bits >> 52) & 0x7FF;
    const unsigned long long mant = (bits & 0xFFFFFFFFFFFFFull) | (ex ? 0x10000000000000ull : 0ull);
    const int s = (eb + 1023) - (ex ? ex : 1);
    SdMap m;
    m.A = 0; m.H0 = 0; m.H1 = 0;
    if (s >= 1 && s <= 53) {                             // s > 53: x < u / 2, the sum does not move (s < 1 cannot happen: x < 2^eb)
        const unsigned long long q = mant >> s, rem = mant & ((1ull << s) - 1ull), half = 1ull << (s - 1);
        m.A = (long long)q;
        if (rem > half) m.A += 1;
        else if (rem == half) { m.H0 = (int)(q & 1ull); m.H1 = (int)((q + 1ull) & 1ull); }
    }
    return m;
}
// REG > 0: at most REG terms per thread, fetched ONCE into registers by a fully unrolled loop (one
// memory round trip instead of one per term and walk: 26 -> 8 us at 11 087 bins); REG == 0: any
// count up to PER_MAX, re-read in each walk.
struct SdShared {
    double sh_p[1024];                    // scan of the threads' approximate sums
    long long sh_A[1024];
    int sh_H0[1024], sh_H1[1024], sh_flag[1024], sh_nb[1024], sh_cnt[1024];
    int sh_e[1024];                       // binade every thread's own walk ENDS in (checked against its successor's start)
    long long b_A[SD_MAX_BOUND + 1];      // map of the segment that ENDS before boundary k (k = n: the tail)
    int b_H0[SD_MAX_BOUND + 1], b_H1[SD_MAX_BOUND + 1], b_e[SD_MAX_BOUND + 1];
    double b_x[SD_MAX_BOUND + 1];
    // boundaries as the threads meet them (unordered): owner thread, its how-manieth, the map of the thread's own
    // terms since its previous boundary (or its start), the boundary term, the binade after it
    long long r_A[SD_MAX_BOUND];
    int r_H0[SD_MAX_BOUND], r_H1[SD_MAX_BOUND], r_e[SD_MAX_BOUND], r_tid[SD_MAX_BOUND], r_j[SD_MAX_BOUND];
    double r_x[SD_MAX_BOUND];
    int s_nrec;
    int s_bad;
    double sh_wp[16];
    int sh_wc[16];
    long long sh_wA[16];
    int sh_wH0[16], sh_wH1[16], sh_wf[16], sh_wn[16];
};
// `stage` (REG == 0, sample-major input): 16 wave-private transposition buffers of 64 x 9 doubles.  A thread's elements
// are contiguous in memory (55 of them at 50 kb), so a load instruction of the wave used to touch 64 different 128-byte
// lines for 8 bytes each -- with sixteen waves in flight the 32 KB L1 kept none of them for the next of a thread's
// eight loads: ~8 000 line requests per wave and walk.  Now the wave's lanes 8 g .. 8 g + 7 fetch the eight consecutive
// elements of ONE thread (trip i: thread 8 i + g of the wave): eight 64-byte runs per instruction, every byte used,
// and the values go to their owners through the buffer.
constexpr int SD_STAGE_DOUBLES = 16 * 64 * 9;
template <int REG, bool STAGE = false>
__device__ inline void sd_fast_block(const int64_t i, SdShared *sm, const double *__restrict__ sdT, int64_t B, int64_t Ns,
                                     double *__restrict__ out, int *__restrict__ fail, double *__restrict__ out2,
                                     int64_t sb, int64_t si, double *stage = nullptr) {     // element (bin b, sample i) at sdT[b * sb + i * si]
    constexpr int PER_MAX = 64;                      // elements per thread (B <= 65536)
    // the workgroup's scratch (caller-provided LDS: static in k_sd_fast, the dynamic region in k_seg_tree)
    double *sh_p = sm->sh_p;
    long long *sh_A = sm->sh_A;
    int *sh_H0 = sm->sh_H0, *sh_H1 = sm->sh_H1, *sh_flag = sm->sh_flag, *sh_nb = sm->sh_nb, *sh_cnt = sm->sh_cnt;
    long long *b_A = sm->b_A;
    int *b_H0 = sm->b_H0, *b_H1 = sm->b_H1, *b_e = sm->b_e;
    double *b_x = sm->b_x;
    int &s_bad = sm->s_bad;
    double *sh_wp = sm->sh_wp;
    int *sh_wc = sm->sh_wc;
    long long *sh_wA = sm->sh_wA;
    int *sh_wH0 = sm->sh_wH0, *sh_wH1 = sm->sh_wH1, *sh_wf = sm->sh_wf, *sh_wn = sm->sh_wn;
    const int tid = threadIdx.x;
    const int per = (int)((B + 1023) / 1024);
    if (per > PER_MAX) { if (tid == 0) fail[i] = 1; return; }
    if (tid == 0) { s_bad = 0; sm->s_nrec = 0; }
    const int64_t lo = (int64_t)tid * per, hi = lo + per < B ? lo + per : B;
    constexpr int NREG = REG > 0 ? REG : 1;
    double xr[NREG];
    // STAGE: the thread's elements e0 .. e0 + 7 through the wave's transposition buffer (see SD_STAGE_DOUBLES; sb == 1)
    auto load8 = [&](const int e0, double (&b8)[8]) {
        const int lane = tid & 63, g = lane >> 3, u_ = lane & 7;
        double *buf = stage + (tid >> 6) * (64 * 9);
        const double *base = sdT + i * si;
        const int64_t wave_lo = (int64_t)(tid & ~63) * per;          // first element of the wave's first thread
        double got[8];
#pragma unroll
        for (int t8 = 0; t8 < 8; ++t8) {                             // trip t8: the run of the wave's thread 8 t8 + g
            const int64_t e = wave_lo + (int64_t)(8 * t8 + g) * per + e0 + u_;
            got[t8] = (e0 + u_ < per && e < B) ? base[e] : 0.0;
        }
#pragma unroll
        for (int t8 = 0; t8 < 8; ++t8) buf[(8 * t8 + g) * 9 + u_] = got[t8];
        __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
        __builtin_amdgcn_wave_barrier();
#pragma unroll
        for (int u = 0; u < 8; ++u) b8[u] = buf[lane * 9 + u];
        __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
        __builtin_amdgcn_wave_barrier();                             // (the buffer is rewritten by the next call)
    };
    if constexpr (REG > 0 && STAGE) {
#pragma unroll
        for (int r = 0; r < (NREG + 7) / 8; ++r) {
            double b8[8];
            load8(8 * r, b8);
#pragma unroll
            for (int u = 0; u < 8; ++u)
                if (8 * r + u < NREG) xr[8 * r + u] = b8[u];
        }
    } else if (REG > 0) {
#pragma unroll
        for (int e = 0; e < NREG; ++e) xr[e] = lo + e < hi ? sdT[(lo + e) * sb + i * si] : 0.0;
    }
    const int n_mine = (int)(hi > lo ? hi - lo : 0);
    // f(value) for this thread's elements in order: from the registers (REG > 0), or re-read eight at a time
    // (REG == 0: one load per iteration was a memory round trip per element and walk -- 0.4 ms per
    // 125-sample batch at 50 kb)
    auto for_each = [&](auto f) {
        if (REG > 0) {
#pragma unroll
            for (int e = 0; e < NREG; ++e)           // (a predicate, not a break: the loop unrolls, xr[] stays in registers)
                if (e < n_mine) f(xr[e]);
        } else if constexpr (STAGE) {
            for (int e0 = 0; e0 < per; e0 += 8) {                    // (the same trips in every lane; sb == 1: the caller checks)
                double b8[8];
                load8(e0, b8);
#pragma unroll
                for (int u = 0; u < 8; ++u)
                    if (e0 + u < n_mine) f(b8[u]);
            }
        } else {
            for (int e0 = 0; e0 < n_mine; e0 += 8) {
                double b8[8];
#pragma unroll
                for (int u = 0; u < 8; ++u) b8[u] = e0 + u < n_mine ? sdT[(lo + e0 + u) * sb + i * si] : 0.0;
#pragma unroll
                for (int u = 0; u < 8; ++u)
                    if (e0 + u < n_mine) f(b8[u]);
            }
        }
    };
    int cnt = 0;
    bool bad = false;
    double loc = 0.0;
    for_each([&](double v) {
        if (v == v) {
            ++cnt;
            if (!(v >= 0.0) || v > 1.7e308) bad = true;
            loc += v;
        }
    });
    // approximate prefix sums: inclusive scan of the threads' sums (wave shuffles, then the sixteen
    // wave totals), then along the thread's elements
    {
        const int lane = tid & 63, wv = tid >> 6;
        double incl = loc;
        int inc_c = cnt;
        for (int o = 1; o < 64; o <<= 1) {
            const double up = __shfl_up(incl, o);
            const int upc = __shfl_up(inc_c, o);
            if (lane >= o) { incl += up; inc_c += upc; }
        }
        if (lane == 63) { sh_wp[wv] = incl; sh_wc[wv] = inc_c; }
        wc_sync();
        double base = 0.0;
        int basec = 0;
        for (int q = 0; q < wv; ++q) { base += sh_wp[q]; basec += sh_wc[q]; }
        sh_p[tid] = base + incl;
        sh_cnt[tid] = basec + inc_c;
        wc_sync();
    }
    double before = tid > 0 ? sh_p[tid - 1] : 0.0;
    const int total_cnt = sh_cnt[1023];
    // element maps, the thread's own segmented fold: `head` = composition up to (not including) the
    // thread's first boundary, `tail` = composition after its last boundary
    SdMap ident; ident.A = 0; ident.H0 = 0; ident.H1 = 0;
    SdMap run = ident;
    int n_bound = 0;
    for_each([&](double raw_v) {
        const double xv = raw_v != raw_v ? 0.0 : raw_v;      // NaN: not part of the sum
        const double after = before + xv;
        bool boundary = false;
        if (xv > 0.0) {
            if (before <= 0.0) boundary = true;
            else {
                const int eb = sd_exponent(before), ea = sd_exponent(after);
                if (eb != ea) boundary = true;
                else if (eb < -900 || eb > 50) bad = true;     // denormal / enormous sums: not worth the care
                else run = sd_compose(run, sd_element_map(xv, eb));
            }
        }
        if (boundary) {
            // recorded once, here: its place among all boundaries is known after the scan below (a second walk
            // that published them in order cost as much as this one)
            const int slot = atomicAdd(&sm->s_nrec, 1);
            if (slot < SD_MAX_BOUND) {
                sm->r_tid[slot] = tid; sm->r_j[slot] = n_bound;
                sm->r_A[slot] = run.A; sm->r_H0[slot] = run.H0; sm->r_H1[slot] = run.H1;
                sm->r_x[slot] = xv;
                sm->r_e[slot] = sd_exponent(after);       // the exponent the following segment assumes
            }
            ++n_bound;
            run = ident;
        }
        before = after;
    });
    // A thread's walk ends on its own sequentially rounded sum, its successor starts from the scan's value of
    // the same prefix: different roundings.  If the two sit on opposite sides of a power of two, the junction
    // is a binade boundary nobody recorded (the maps left and right of it assume different units): the serial
    // kernel takes the sample.  (Practically unreachable -- the sum would have to sit within an ulp of 2^k.)
    sm->sh_e[tid] = before > 0.0 ? sd_exponent(before) : -100000;
    wc_sync();
    if (tid > 0) {
        const double start = sh_p[tid - 1];
        if ((start > 0.0 ? sd_exponent(start) : -100000) != sm->sh_e[tid - 1]) bad = true;
    }
    // block-wide segmented scan over the threads: value = (flag: holds a boundary, map: its tail --
    // or its whole fold when it holds none)
    if (bad) s_bad = 1;
    {
        // inclusive segmented scan: within the wave by shuffles, then over the sixteen wave results
        const int lane = tid & 63, wv = tid >> 6;
        SdMap val = run;
        int flag = n_bound > 0, nb = n_bound;
        for (int o = 1; o < 64; o <<= 1) {
            SdMap left;
            left.A = __shfl_up(val.A, o);
            left.H0 = __shfl_up(val.H0, o);
            left.H1 = __shfl_up(val.H1, o);
            const int lf = __shfl_up(flag, o), lnb = __shfl_up(nb, o);
            if (lane >= o) {
                if (!flag) { val = sd_compose(left, val); flag = lf; }
                nb += lnb;
            }
        }
        if (lane == 63) { sh_wA[wv] = val.A; sh_wH0[wv] = val.H0; sh_wH1[wv] = val.H1; sh_wf[wv] = flag; sh_wn[wv] = nb; }
        wc_sync();
        // carry of the earlier waves (their open segment), composed in order
        SdMap carryw; carryw.A = 0; carryw.H0 = 0; carryw.H1 = 0;
        int cf = 0, cn = 0;
        for (int q = 0; q < wv; ++q) {
            SdMap m; m.A = sh_wA[q]; m.H0 = sh_wH0[q]; m.H1 = sh_wH1[q];
            if (sh_wf[q]) { carryw = m; cf = 1; }
            else carryw = sd_compose(carryw, m);
            cn += sh_wn[q];
        }
        if (!flag) { val = sd_compose(carryw, val); flag = cf; }
        nb += cn;
        sh_A[tid] = val.A; sh_H0[tid] = val.H0; sh_H1[tid] = val.H1;
        sh_flag[tid] = flag;
        sh_nb[tid] = nb;
        wc_sync();
    }
    const int n_total_bound = sh_nb[1023];
    if (n_total_bound > SD_MAX_BOUND) { if (tid == 0) fail[i] = 1; return; }
    // every recorded boundary to its place: boundary j of thread t is number (boundaries before t) + j; the map of
    // the segment that ends just before it is the thread's own terms since its previous boundary, for its first
    // one behind the open segment of all earlier threads (the scan's value at t - 1)
    if (tid < n_total_bound) {
        const int t = sm->r_tid[tid], j = sm->r_j[tid];
        SdMap m; m.A = sm->r_A[tid]; m.H0 = sm->r_H0[tid]; m.H1 = sm->r_H1[tid];
        int k = j;
        if (t > 0) {
            k += sh_nb[t - 1];
            if (j == 0) {
                SdMap carry; carry.A = sh_A[t - 1]; carry.H0 = sh_H0[t - 1]; carry.H1 = sh_H1[t - 1];
                m = sd_compose(carry, m);
            }
        }
        b_A[k] = m.A; b_H0[k] = m.H0; b_H1[k] = m.H1;
        b_x[k] = sm->r_x[tid];
        b_e[k] = sm->r_e[tid];
    }
    // the tail: the open segment through the last thread
    if (tid == 1023) { b_A[n_total_bound] = sh_A[1023]; b_H0[n_total_bound] = sh_H0[1023]; b_H1[n_total_bound] = sh_H1[1023]; }
    wc_sync();
    // The fold over the boundaries is a dependent chain of ~n_total_bound steps: the first wave runs it with every
    // lane computing the same values (no divergence), the records of the boundaries in REGISTERS -- lane l holds
    // record l, a step reads it with v_readlane -- instead of five dependent LDS reads per step (records beyond 63
    // are read from LDS)
    if (tid < 64) {
        const int lane = tid;
        const bool mine = lane <= n_total_bound;
        const long long rA = mine ? b_A[lane] : 0;
        const int rH0 = mine ? b_H0[lane] : 0, rH1 = mine ? b_H1[lane] : 0;
        const double rx = lane < n_total_bound ? b_x[lane] : 0.0;
        const int re = lane < n_total_bound ? b_e[lane] : 0;
        auto rl = [](int v, int src) { return __builtin_amdgcn_readlane(v, src); };
        bool ok = !s_bad;
        double sum = 0.0;
        int e_cur = 0;
        for (int k = 0; k <= n_total_bound && ok; ++k) {
            long long A;
            int H0, H1, eb = 0;
            double xb = 0.0;
            if (k < 64) {
                const unsigned int alo = (unsigned int)rl((int)(unsigned int)(rA & 0xFFFFFFFFll), k);
                const int ahi = rl((int)(rA >> 32), k);
                A = ((long long)ahi << 32) | (long long)alo;
                H0 = rl(rH0, k); H1 = rl(rH1, k);
                const long long xbits = __double_as_longlong(rx);
                const unsigned int xlo = (unsigned int)rl((int)(unsigned int)(xbits & 0xFFFFFFFFll), k);
                const int xhi = rl((int)(xbits >> 32), k);
                xb = __longlong_as_double(((long long)xhi << 32) | (long long)xlo);
                eb = rl(re, k);
            } else {
                A = b_A[k]; H0 = b_H0[k]; H1 = b_H1[k];
                if (k < n_total_bound) { xb = b_x[k]; eb = b_e[k]; }
            }
            // the segment before boundary k (k = 0: before the first positive term, the identity)
            if (k > 0 || n_total_bound == 0) {
                if (A != 0 || H0 != 0 || H1 != 0) {
                    if (!(sum > 0.0) || sd_exponent(sum) != e_cur) { ok = false; break; }
                    // in units of u = 2^(e_cur - 52) the sum IS its 53-bit significand (it is a normal number of
                    // that binade): integer arithmetic on the bits, no scaling and no conversions on this chain
                    const long long S = (__double_as_longlong(sum) & 0xFFFFFFFFFFFFFll) | (1ll << 52);
                    const long long S2 = S + A + ((S & 1) ? H1 : H0);
                    if (A < 0 || A >= (1ll << 53) || S2 >= (1ll << 53)) { ok = false; break; }
                    sum = __longlong_as_double(((long long)(e_cur + 1023) << 52) | (S2 & 0xFFFFFFFFFFFFFll));
                }
            }
            if (k < n_total_bound) {
                sum = sum + xb;                                         // the boundary term: a plain float64 add
                if (!(sum > 0.0) || sd_exponent(sum) != eb) { ok = false; break; }
                e_cur = eb;
            }
        }
        if (tid == 0) {
            if (ok) {
                out[i] = sum / (double)total_cnt;
                if (out2) out2[i] = sum / (double)total_cnt;
            }
            fail[i] = ok ? 0 : 1;                  // 1: the serial kernel computes this sample
        }
    }
}

template <int REG, bool STAGE = false>
__global__ __launch_bounds__(1024) void k_sd_fast(const double *__restrict__ sdT, int64_t B, int64_t Ns,
                                                  double *__restrict__ out, int *__restrict__ fail,
                                                  double *__restrict__ out2, int64_t sb, int64_t si) {
    __shared__ SdShared sm;
    if constexpr (STAGE) {
        __shared__ double stage[SD_STAGE_DOUBLES];
        sd_fast_block<REG, true>(blockIdx.x, &sm, sdT, B, Ns, out, fail, out2, sb, si, stage);
    } else {
        sd_fast_block<REG>(blockIdx.x, &sm, sdT, B, Ns, out, fail, out2, sb, si);
    }
}

// what the latency mode appends to another kernel's grid (k_seg_tree): `blocks` workgroups, one per sample
struct SdRider {
    int blocks;
    const double *sdT;
    int64_t B, Ns;
    double *out, *out2;
    int *fail;
    int64_t sb, si;
};

void launch_sd_fast(hipStream_t stream, const double *sdT, int64_t B, int64_t Ns, double *out, int *fail, double *out2,
                    int64_t sb, int64_t si) {
    const int64_t per = (B + 1023) / 1024;
    // (a sample's values contiguous -- every batch: the cooperative loads; WC_TEST_SD_STAGE=0: a lane per run, round 5's form)
    const bool staged = sb == 1 && !(getenv("WC_TEST_SD_STAGE") && getenv("WC_TEST_SD_STAGE")[0] == '0');
    if (per <= 12 && staged)
        hipLaunchKernelGGL((k_sd_fast<12, true>), dim3((unsigned)Ns), dim3(1024), 0, stream, sdT, B, Ns, out, fail, out2, sb, si);
    else if (per <= 12)
        hipLaunchKernelGGL(k_sd_fast<12>, dim3((unsigned)Ns), dim3(1024), 0, stream, sdT, B, Ns, out, fail, out2, sb, si);
    else if (per <= 24 && staged)
        hipLaunchKernelGGL((k_sd_fast<24, true>), dim3((unsigned)Ns), dim3(1024), 0, stream, sdT, B, Ns, out, fail, out2, sb, si);
    else if (per <= 24)
        hipLaunchKernelGGL(k_sd_fast<24>, dim3((unsigned)Ns), dim3(1024), 0, stream, sdT, B, Ns, out, fail, out2, sb, si);
    else if (staged)
        hipLaunchKernelGGL((k_sd_fast<0, true>), dim3((unsigned)Ns), dim3(1024), 0, stream, sdT, B, Ns, out, fail, out2, sb, si);
    else
        hipLaunchKernelGGL(k_sd_fast<0>, dim3((unsigned)Ns), dim3(1024), 0, stream, sdT, B, Ns, out, fail, out2, sb, si);
}

// --------------------------------------------------------------- cleaning ----
// Keep bins with refSizes >= minrefbins (wisecondor.py:215-222); one wave per
// (sample, selected chromosome) compacts z, r and the genomic position in order.
// zT / rT / nT here: the sample-major [Ns, B] copies (see k_inflate)
// (si, sb): strides of the z / ratio / count arrays per sample and per bin -- (B, 1) for the
// sample-major copies, (1, Ns) to read the repeats' bin-major arrays directly (small batches)
__global__ __launch_bounds__(1024) void k_clean(const double *__restrict__ zT, const double *__restrict__ rT,
                                                const double *__restrict__ nT, int64_t B, int64_t Ns,
                                                const int64_t *__restrict__ moff, const int64_t *__restrict__ goff,
                                                const int *__restrict__ m2g, const int *__restrict__ sel, int n_sel,
                                                double minref, double *__restrict__ zc, double *__restrict__ rc,
                                                int *__restrict__ gpos, Region *__restrict__ regions,
                                                int64_t str_i, int64_t str_b) {
    // one workgroup (up to 1 024 threads) per (sample, chromosome): CL_U x blockDim bins per trip -- every load of a
    // trip requested before the first is used -- the kept ones compacted in order (ballot within a wave, the
    // wave counts of the trip's CL_U slices through LDS: one barrier per trip).  One wave per region walked a 50 kb
    // chromosome in 74 dependent trips (0.22 ms per 125-sample batch), 1 024 bins per trip in five (0.13 ms).
    constexpr int CL_U = 4;
    __shared__ int s_cnt[2][CL_U][16];
    const int tid = threadIdx.x, lane = tid & 63, w = tid >> 6;
    const int64_t i = blockIdx.y;
    const int si = blockIdx.x;
    const int c = sel[si];
    const int64_t cs = moff[c], ce = moff[c + 1];
    int count = 0, trip = 0;
    const int nt = (int)blockDim.x, nwaves = nt >> 6;      // 64 .. 1 024 threads: the launch sizes the workgroup to the regions
    const int goff_c = (int)goff[c];
    for (int64_t base = cs; base < ce; base += (int64_t)CL_U * nt, trip ^= 1) {
        bool keep[CL_U];
        double zv[CL_U], rv[CL_U], nv[CL_U];
        int gp[CL_U];
#pragma unroll
        for (int u = 0; u < CL_U; ++u) {
            const int64_t b = base + (int64_t)u * nt + tid;
            const bool in = b < ce;
            const int64_t at = i * str_i + (in ? b : cs) * str_b;
            nv[u] = nT[at];
            zv[u] = zT[at];
            rv[u] = rT[at];
            gp[u] = m2g[in ? b : cs] - goff_c;
            keep[u] = in;
        }
        unsigned long long mask[CL_U];
#pragma unroll
        for (int u = 0; u < CL_U; ++u) {
            keep[u] = keep[u] && nv[u] >= minref;
            mask[u] = __ballot(keep[u]);
            if (lane == 0) s_cnt[trip][u][w] = __popcll(mask[u]);
            if (tid < 16 && tid >= nwaves) s_cnt[trip][u][tid] = 0;  // waves this workgroup does not have
        }
        wc_sync();                     // (the other buffer is written in the next trip: one barrier per trip)
        int before = count;
#pragma unroll
        for (int u = 0; u < CL_U; ++u) {
            int mine = 0, total = 0;
#pragma unroll
            for (int q = 0; q < 16; ++q) {
                const int n = s_cnt[trip][u][q];
                mine += q < w ? n : 0;
                total += n;
            }
            if (keep[u]) {
                const int at = before + mine + __popcll(mask[u] & ((1ull << lane) - 1ull));
                zc[i * B + cs + at] = zv[u];
                rc[i * B + cs + at] = rv[u];
                gpos[i * B + cs + at] = gp[u];
            }
            before += total;
        }
        count = before;
    }
    if (tid == 0) {
        Region rg;
        rg.off = i * B + cs;
        rg.n = count;
        rg.pad = c;
        regions[i * n_sel + si] = rg;
    }
}

// inflateArrayMulti + per-chromosome split (wisetools.py:281-295, wisecondor.py:260-268)
// zs / rs / ns are sample-major [Ns, B] (transposed back once after the repeats: the bin-major
// arrays would be read with a stride of Ns doubles here)
__global__ void k_inflate(const double *__restrict__ zs, const double *__restrict__ rs, const double *__restrict__ ns,
                          int64_t B, int64_t Btot, const int *__restrict__ g2m, double minref,
                          double *__restrict__ res_z, double *__restrict__ res_r, int64_t si, int64_t sb) {
    int64_t g = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    int64_t i = blockIdx.y;
    if (g >= Btot) return;
    int m = g2m[g];
    double z = 0.0, r = 0.0;
    if (m >= 0 && ns[i * si + m * sb] >= minref) {
        z = zs[i * si + m * sb];
        r = rs[i * si + m * sb] - 1.0;
    }
    if (res_z) res_z[i * Btot + g] = z;
    if (res_r) res_r[i * Btot + g] = r;
}

// ------------------------------------------------------- Stouffer windows ----
__global__ void k_fill_rs(double *rs, int64_t n) {
    int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (i < n) rs[i] = i > 0 ? 1.0 / sqrt((double)i) : 0.0;
}

// The regions k_seg_walk should start first (see there): a region whose whole-region value |sum z| / sqrt(n), or one of
// the 64-bin windows the prefix pass works in, reaches `cut` holds an aberration of some length -- hundreds of loud cells,
// candidate windows of thousands of bins: up to ten times the mean region's work (a spike of a bin or two costs nothing
// and is not looked for).  list[0 .. *count - 1]: their numbers; index[region] = its place in the list + 1, or 0.  *count is reset by
// k_walk_rows (the launch after the walk); a list entry only counts if index[] of THIS batch points back at it.
struct WalkHot {
    int *list, *count, *index;
    int cap;
    double cut;
    const int *tail_flag;     // (rides along: the late repeats' overflow word, copied to misc[1] for the host)
};
// Prefix sums, sum |z| and a finiteness flag per region; one wave per region (any summation
// order satisfies window_eps' bound).
__global__ __launch_bounds__(256) void k_region_prefix(const double *__restrict__ z,
                                                       const Region *__restrict__ regions, int64_t n_regions,
                                                       double *__restrict__ prefix, double *__restrict__ reg_abs,
                                                       int *__restrict__ reg_flag, int *__restrict__ counters,
                                                       int *__restrict__ out_n, int *__restrict__ misc,
                                                       const WalkHot hot = WalkHot{nullptr, nullptr, nullptr, 0, 0.0, nullptr}) {
    const int lane = threadIdx.x & 63;
    int64_t r = (int64_t)blockIdx.x * 4 + (threadIdx.x >> 6);
    // first kernel of a segmentation call: its counters start at zero (no separate memsets)
    if (blockIdx.x == 0 && threadIdx.x < 8) counters[threadIdx.x] = 0;
    if (blockIdx.x == 0 && threadIdx.x >= 8 && threadIdx.x < 12 && misc) misc[threadIdx.x - 8] = (threadIdx.x == 9 && hot.tail_flag) ? *hot.tail_flag : 0;   // ([1]: the late repeats' overflow word, see run_repeat)
    if (r >= n_regions) return;
    if (lane == 0) out_n[r] = 0;
    const Region rg = regions[r];
    const double *zz = z + rg.off;
    double *P = prefix + rg.off + r;
    // 64 consecutive bins per trip (coalesced), inclusive wave scan, running total carried along
    double a = 0.0, run = 0.0, loud64 = 0.0;
    int finite = 1;
    if (lane == 0) P[0] = 0.0;
    // four trips at a time: their loads and wave scans are independent, only the running total chains them
    // (the same additions in the same order as one trip after the other)
    for (int t0 = 0; t0 < rg.n; t0 += 256) {
        double v[4], incl[4];
#pragma unroll
        for (int u = 0; u < 4; ++u) {
            const int t = t0 + 64 * u + lane;
            v[u] = t < rg.n ? zz[t] : 0.0;
        }
#pragma unroll
        for (int u = 0; u < 4; ++u) {
            if (!isfinite(v[u])) finite = 0;
            a += fabs(v[u]);
            incl[u] = v[u];
        }
        for (int o = 1; o < 64; o <<= 1) {
#pragma unroll
            for (int u = 0; u < 4; ++u) {
                const double up = __shfl_up(incl[u], o);
                if (lane >= o) incl[u] += up;
            }
        }
#pragma unroll
        for (int u = 0; u < 4; ++u) {
            const int t = t0 + 64 * u + lane;
            if (t < rg.n) P[t + 1] = run + incl[u];
            const double trip = __shfl(incl[u], 63);
            loud64 = fmax(loud64, fabs(trip));              // (the trip's 64 bins as one window, see WalkHot)
            run += trip;
        }
    }
    for (int o = 32; o > 0; o >>= 1) {
        a += __shfl_xor(a, o);
        finite &= __shfl_xor(finite, o);
    }
    if (lane == 0) {
        reg_abs[r] = a;
        reg_flag[r] = finite;
        if (hot.index) {
            int idx = 0;
            if (finite && rg.n > 0 && (fabs(run) >= hot.cut * sqrt((double)rg.n) || loud64 >= hot.cut * 8.0)) {
                const int at = atomicAdd(hot.count, 1);
                if (at < hot.cap) { hot.list[at] = (int)r; idx = at + 1; }
            }
            hot.index[r] = idx;
        }
    }
}

// fillTriMin's filter (wisetools.py:479-487): a window keeps its value only if
// abs(median(ratio[x..y]) - 1) >= mineffectsize, otherwise it is exactly 0.  One bit per
// window, packed region by region in triangle order; a NULL bitmap means "no filter".
struct WindowMask {
    const unsigned int *bits;   // NULL -> every window valid
    long long base;             // bit offset of this region's triangle
    int n;                      // region length
    __device__ inline bool valid(int x, int y) const {
        if (!bits) return true;
        long long lin = base + (long long)x * n - ((long long)x * (x - 1)) / 2 + (y - x);
        return (bits[lin >> 5] >> (lin & 31)) & 1u;
    }
};

// -mineffectsize in O(1) per window (wisetools.py:479-487: a window counts only if
// abs(median(ratio[x..y]) - 1) >= threshold).  One LANE per first bin x; the lanes of a wave walk their
// windows' last bins in lock step (coalesced loads, every lane at the same window length m).  The
// comparison is monotone in the median, so there are two doubles u_hi >= 1 >= u_lo (found on the host
// with the very same expression) with  "valid  <=>  median >= u_hi  or  median <= u_lo",  and an order
// statistic against a fixed value is a COUNT: with c = #(elements >= u_hi),
//   m odd:   median >= u_hi  <=>  c >= (m + 1) / 2;
//   m even:  c >= m / 2 + 1 -> both middle elements >= u_hi -> valid;  c <= m / 2 - 1 -> not (on this side);
//            c == m / 2 -> the middle elements are max(elements < u_hi) and min(elements >= u_hi): the
//            median (a + b) / 2 is formed exactly as numpy forms it and compared as the reference does
// (and the mirror image with #(elements <= u_lo)).  Per lane: two counters, four running extremes, a NaN
// flag (np.median of a window with a NaN is NaN: the comparison fails).  The sorted-insert kernel below
// (O(n) per window) is kept as WC_MINEFFECT=sorted for cross-checks.
__global__ __launch_bounds__(256) void k_window_valid_count(const double *__restrict__ ratio,
                                                           const Region *__restrict__ regions,
                                                           const long long *__restrict__ bit_off, double min_effect,
                                                           double u_hi, double u_lo,
                                                           unsigned int *__restrict__ bits) {
    const Region rg = regions[blockIdx.y];
    const int x = (int)(blockIdx.x * blockDim.x + threadIdx.x);
    const int x0 = (int)(blockIdx.x * blockDim.x + (threadIdx.x & ~63));     // the wave's first start
    if (x0 >= rg.n) return;
    const bool live = x < rg.n;
    const double *rr = ratio + rg.off;
    const long long row_base = bit_off[blockIdx.y] + (long long)x * rg.n - ((long long)x * (x - 1)) / 2;
    const long long first_word = row_base >> 5, last_word = (row_base + (rg.n - 1 - x)) >> 5;
    int c_hi = 0, c_lo = 0;
    bool nan = false;
    double below_hi = -INFINITY, from_hi = INFINITY;     // max of the elements < u_hi, min of those >= u_hi
    double upto_lo = -INFINITY, above_lo = INFINITY;     // max of the elements <= u_lo, min of those > u_lo
    unsigned int word = 0;
    long long word_idx = first_word;
    const int steps = rg.n - x0;                         // the wave's longest row
    for (int d = 0; d < steps; ++d) {
        const int y = x + d;
        if (!live || y >= rg.n) continue;
        const double v = rr[y];
        if (v != v) nan = true;
        if (v >= u_hi) { ++c_hi; from_hi = fmin(from_hi, v); } else if (v == v) below_hi = fmax(below_hi, v);
        if (v <= u_lo) { ++c_lo; upto_lo = fmax(upto_lo, v); } else if (v == v) above_lo = fmin(above_lo, v);
        const int m = d + 1;
        bool ok;
        if (m & 1) {
            const int k = (m + 1) >> 1;
            ok = c_hi >= k || c_lo >= k;
        } else {
            const int h = m >> 1;
            ok = c_hi > h || c_lo > h;
            if (!ok && c_hi == h) ok = fabs((below_hi + from_hi) / 2.0 - 1.0) >= min_effect;
            if (!ok && c_lo == h) ok = fabs((upto_lo + above_lo) / 2.0 - 1.0) >= min_effect;
        }
        if (nan) ok = false;
        const long long lin = row_base + d;
        if ((lin >> 5) != word_idx) {
            if (word) {
                if (word_idx == first_word || word_idx == last_word) atomicOr(&bits[word_idx], word);
                else bits[word_idx] = word;              // a word inside the row: nobody else writes it (zeroed by the caller)
            }
            word = 0;
            word_idx = lin >> 5;
        }
        if (ok) word |= 1u << (lin & 31);
    }
    if (live && word) atomicOr(&bits[word_idx], word);
}

// One wave per (region, first bin x): the windows [x, y], y = x..n-1, by inserting
// ratio[y] into a sorted LDS array (count-smaller + shift) and reading the median off
// the middle.  np.median averages the two middle values for even lengths; a NaN in the
// window makes the median NaN, which fails the comparison (window zeroed).
__global__ __launch_bounds__(64) void k_window_valid(const double *__restrict__ ratio,
                                                     const Region *__restrict__ regions,
                                                     const long long *__restrict__ bit_off, double min_effect,
                                                     unsigned int *__restrict__ bits) {
    extern __shared__ double sorted[];
    const int lane = threadIdx.x;
    const int x = blockIdx.x;
    const Region rg = regions[blockIdx.y];
    if (x >= rg.n) return;
    const double *rr = ratio + rg.off;
    const long long row_base = bit_off[blockIdx.y] + (long long)x * rg.n - ((long long)x * (x - 1)) / 2;
    int len = 0, nans = 0;
    unsigned int word = 0;
    long long word_idx = row_base >> 5;
    for (int y = x; y < rg.n; ++y) {
        const double v = rr[y];
        if (v != v) {
            ++nans;
        } else {
            int cnt = 0;
            for (int t = lane; t < len; t += 64) cnt += (sorted[t] < v);
            for (int o = 32; o > 0; o >>= 1) cnt += __shfl_xor(cnt, o);
            const int pos = cnt;
            for (int hi = len; hi > pos; hi -= 64) {
                const int lo = (hi - 64 > pos) ? hi - 64 : pos;
                const int t = lo + lane;
                double val = 0.0;
                if (t < hi) val = sorted[t];
                wc_sync();
                if (t < hi) sorted[t + 1] = val;
                wc_sync();
            }
            if (lane == 0) sorted[pos] = v;
            ++len;
            wc_sync();
        }
        double med = NAN;
        if (nans == 0) med = (len & 1) ? sorted[len / 2] : (sorted[len / 2 - 1] + sorted[len / 2]) / 2.0;
        const bool ok = fabs(med - 1.0) >= min_effect;
        const long long lin = row_base + (y - x);
        if ((lin >> 5) != word_idx) {
            if (lane == 0 && word) atomicOr(&bits[word_idx], word);
            word = 0;
            word_idx = lin >> 5;
        }
        if (ok) word |= 1u << (lin & 31);
    }
    if (lane == 0 && word) atomicOr(&bits[word_idx], word);
}

// bit offset of every region's triangle in the window bitmap (exclusive scan of n(n+1)/2)
__global__ void k_bit_offsets(const Region *__restrict__ regions, int64_t n_regions, long long *__restrict__ bit_off) {
    if (blockIdx.x != 0 || threadIdx.x != 0) return;
    long long at = 0;
    for (int64_t r = 0; r < n_regions; ++r) {
        bit_off[r] = at;
        long long n = regions[r].n;
        at += n * (n + 1) / 2;
    }
}

// Exact value of window [x, y]: np_sum(z[x:y+1]) / np_sqrt(y-x+1) (wisetools.py:471)
template <bool GROUP8>
__device__ inline double window_exact(const double *__restrict__ zz, int x, int y, int sub, const WindowMask &wm) {
    if (!wm.valid(x, y)) return 0.0;   // uniform per 8-lane group: all its lanes evaluate the same window
    const double *p = zz + x;
    double s = wc::pairwise_sum<GROUP8>([&](int64_t t) { return p[t]; }, (int64_t)(y - x + 1), sub);
    s = s + 0.0;        // np.sum starts from its identity +0.0: a sum of nothing but -0.0 comes out as +0.0 (-fno-fast-math keeps this add)
    return s / sqrt((double)(y - x + 1));
}

// The same by a whole wave (leaves of numpy's tree summed eight at a time); every lane
// returns the value.
// SHORT: the caller's windows never exceed numpy's 8 192-element buffer (one pairwise tree, no loop over pieces)
template <bool SHORT = false>
__device__ inline double window_exact_wave(const double *__restrict__ zz, int x, int y, int lane,
                                           const WindowMask &wm, wc::PwWaveScratch &sc) {
    if (!wm.valid(x, y)) return 0.0;
    const double *p = zz + x;
    const int64_t len = (int64_t)(y - x + 1);
    double s;
    // (129 .. 8 192 bins: every lane finds its own node of numpy's tree -- the walk through the LDS stack takes ~50 us for
    //  a 4 700-bin window, and the long windows of an aberrant region were what the slowest workgroups of a batch spent
    //  their lives on: 110 of 180-245 us)
    if (len > WC_PW_BLOCK && len <= WC_NPY_BUFSIZE) s = wc::pairwise_tree_lanes<true>([&](int64_t t) { return p[t]; }, (int)len, lane);
    else if constexpr (SHORT) s = wc::pairwise_tree_wave([&](int64_t t) { return p[t]; }, len, lane, sc);
    else s = wc::pairwise_sum_wave([&](int64_t t) { return p[t]; }, len, lane, sc);
    s = s + 0.0;        // (np.sum's identity, see window_exact)
    return s / sqrt((double)(y - x + 1));
}

__global__ __launch_bounds__(256) void k_region_whole(const double *__restrict__ z, const Region *__restrict__ regions,
                                                      int64_t n_regions, const unsigned int *__restrict__ bits,
                                                      const long long *__restrict__ bit_off,
                                                      double *__restrict__ whole, double *__restrict__ whole2) {
    __shared__ wc::PwWaveScratch sc[4];
    const int w = threadIdx.x >> 6, lane = threadIdx.x & 63;       // a wave per region
    const int64_t r = (int64_t)blockIdx.x * 4 + w;
    if (r >= n_regions) return;
    const Region rg = regions[r];
    const WindowMask wm{bits, bits ? bit_off[r] : 0, rg.n};
    double v;
    if (!bits && rg.n > WC_PW_BLOCK && rg.n <= WC_NPY_BUFSIZE) {
        // (round 6) every lane sums its own node of numpy's tree: no LDS stack walk (40 us per 4 700-bin region before)
        const double *p = z + rg.off;
        v = (wc::pairwise_tree_lanes([&](int64_t t) { return p[t]; }, rg.n, lane) + 0.0) / sqrt((double)rg.n);   // (+ 0.0: np.sum's identity)
    } else {
        v = rg.n > 0 ? window_exact_wave(z + rg.off, 0, rg.n - 1, lane, wm, sc[w]) : NAN;
    }
    if (lane == 0) {
        whole[r] = v;
        if (whole2) whole2[r] = v;             // the caller's results_cwz, when it wants no separate copy
    }
}

__global__ void k_init_jobs(const Region *__restrict__ regions, int64_t n_regions, Job *__restrict__ jobs,
                            int *__restrict__ counters) {
    int64_t r = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (r == 0) { counters[1] = (int)n_regions; counters[5] = 0; counters[6] = 0; counters[7] = 0; }   // [1] / [5]: job counts of even / odd rounds (device-side loop)
    if (r >= n_regions) return;
    Job j;
    j.region = (int)r;
    j.lo = 0;
    j.hi = regions[r].n;
    j.pad = 0;
    jobs[r] = j;
}

struct ScanCtx {
    const double *P;   // prefix of the region
    int lo, hi, L, half, chunk;
    WindowMask wm;
    int clip_lo = 0, clip_hi = 0x7fffffff;   // scan only the windows inside [clip_lo, clip_hi) (a sub-range of the job)
};

// Rows handled by (job, chunk): ROWS_HALF rows from the top of the triangle and the
// ROWS_HALF mirrored rows from the bottom, so every workgroup sees ~ the same work.
// A lane owns one start bin x (its prefix value stays in a register); the four waves of
// the workgroup (nw waves) take the window lengths len = 1 + w, 1 + w + nw, ...: within a wave the length
// is uniform, so rs[len] is one scalar load and P[x + len] one coalesced vector load.
template <class F>
__device__ inline void scan_chunk(const ScanCtx &c, const double *__restrict__ rs, int tid, F f,
                                  unsigned int skip = 0u) {     // bit (side * 16 + wave): nothing to find there
    const int lane = tid & 63;
    const int w = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int nw = (int)(blockDim.x >> 6);
    for (int side = 0; side < 2; ++side) {
        if ((skip >> (side * 16 + w)) & 1u) continue;
        int xr = c.chunk * ROWS_HALF + lane;
        bool live;
        if (side == 0) {
            live = xr < c.half;
        } else {
            xr = c.L - 1 - xr;
            live = xr >= c.half;
        }
        // longest window among the rows of this block (lane 0 has the smallest / largest start)
        const int xr_min = side == 0 ? c.chunk * ROWS_HALF : c.L - 1 - (c.chunk * ROWS_HALF + 63);
        int max_len = c.L - (xr_min < 0 ? 0 : xr_min);
        const int x = c.lo + (live ? xr : 0);
        const int end = c.hi < c.clip_hi ? c.hi : c.clip_hi;
        live = live && x >= c.clip_lo && x < end;
        if (end - c.clip_lo < max_len) max_len = end - c.clip_lo > 0 ? end - c.clip_lo : 0;
        const double px = c.P[x];
        const int room = live ? end - x : 0;             // windows [x, x + len - 1] with len <= room
        // four window lengths per trip: their table factors and prefix values are requested
        // together (a trip per length waits out one load latency per window)
        for (int len0 = 1 + w; len0 <= max_len; len0 += 4 * nw) {
            double r[4], pv[4];
#pragma unroll
            for (int u = 0; u < 4; ++u) {
                const int len = len0 + u * nw;
                r[u] = rs[len];                                    // the table is padded by 4 * 16 lengths
                pv[u] = c.P[x + (len < room ? len : room)];         // always inside the job's slice
            }
#pragma unroll
            for (int u = 0; u < 4; ++u) {
                const int len = len0 + u * nw;
                if (len <= room && len <= max_len) {
                    const int y = x + len - 1;
                    double v = (pv[u] - px) * r[u];
                    if (!c.wm.valid(x, y)) v = 0.0;
                    f(v, x, y);
                }
            }
        }
    }
}

// error bound of the prefix-sum window value against numpy's exact value
__device__ inline double window_eps(int n, double abs_sum) {
    return (2.0 * n + 64.0) * 1.1102230246251565e-16 * abs_sum;
}

// Minimum and maximum of every 32-entry block of the concatenated prefix array (blocks are
// aligned to the array, not to the regions: a block that straddles a region boundary bounds a
// superset, which is still a bound).
constexpr int QB = 32;              // window ends per end block
constexpr int Q_WORK = 1024;        // undecided (row, end block) pairs a block can queue (10 KB of LDS in all: 8 blocks per CU)
constexpr int Q_BLOCKS = 256;       // end blocks of one job held in LDS (jobs up to ~8 k bins)
// One workgroup = 1 024 consecutive entries: a thread loads four (two 16-byte loads), eight lanes reduce a block of 32,
// 32 lanes a second-level block of 128 (tmin2 / tmax2: minimum and maximum of every 128-entry block, for the
// bound-driven search's long windows) -- shuffles only.  (Round 5's form took one entry per thread: 27 000 workgroups
// for a 125 x 50 kb batch, 48 us for 55 MB.)
typedef double f64x2_t __attribute__((ext_vector_type(2)));
__global__ __launch_bounds__(256) void k_block_minmax(const double *__restrict__ prefix, int64_t total,
                                                      double *__restrict__ tmin, double *__restrict__ tmax,
                                                      double *__restrict__ tmin2, double *__restrict__ tmax2) {
    const int tid = threadIdx.x;
    const int64_t at = ((int64_t)blockIdx.x * 256 + tid) * 4;
    double mn = INFINITY, mx = -INFINITY;
    if (at + 4 <= total) {
        const f64x2_t a = *reinterpret_cast<const f64x2_t *>(prefix + at), b = *reinterpret_cast<const f64x2_t *>(prefix + at + 2);
        mn = fmin(fmin(a.x, a.y), fmin(b.x, b.y));
        mx = fmax(fmax(a.x, a.y), fmax(b.x, b.y));
    } else {
        for (int e = 0; e < 4; ++e)
            if (at + e < total) { mn = fmin(mn, prefix[at + e]); mx = fmax(mx, prefix[at + e]); }
    }
    for (int o = 1; o < 8; o <<= 1) {
        mn = fmin(mn, __shfl_xor(mn, o));
        mx = fmax(mx, __shfl_xor(mx, o));
    }
    const int64_t k = (int64_t)blockIdx.x * 32 + (tid >> 3);
    if ((tid & 7) == 0 && k * QB < total) { tmin[k] = mn; tmax[k] = mx; }
    if (!tmin2) return;
    for (int o = 8; o < 32; o <<= 1) {
        mn = fmin(mn, __shfl_xor(mn, o));
        mx = fmax(mx, __shfl_xor(mx, o));
    }
    const int64_t k2 = (int64_t)blockIdx.x * 8 + (tid >> 5);
    if ((tid & 31) == 0 && k2 * (4 * QB) < total) { tmin2[k2] = mn; tmax2[k2] = mx; }
}

// (second level: beyond 512 bins a 128-end block loosens the bound by less than a 32-end block does at 64)
constexpr int QB2 = 4 * QB;         // window ends per second-level block
constexpr int FAR2 = 512;           // window lengths from here on are bounded per second-level block
// Quiet-job certificate.  Most jobs (a chromosome of a sample, or a child range) hold no call:
// every window stays below the threshold.  Proving that needs far fewer evaluations than
// finding the extremes.  A lane owns a window start x (as in the search); for a 32-entry block
// of window ends that begins past x + 32,
//     (P[y'] - P[x]) / sqrt(len) <= (max P over the block - P[x]) / sqrt(min len)
// and symmetrically from below (float rounding is monotone, so the bound also covers the
// values as the search would compute them): one bound stands for 32 windows.  Blocks whose
// bound does not settle it, and the ends before the first such block, are evaluated window by
// window.  A job keeps Job::pad == 0 (search and classify skip it) unless some window could
// reach the threshold: |v| + eps < thr is the same test k_seg_classify applies to the extremes.
// The certificate for one job by the calling workgroup (256 threads), over the row blocks
// first_chunk, first_chunk + chunk_stride, ...; returns (to every thread) 0 when every window of
// those rows stays below the threshold, 1 when some window could reach it or the job is too long
// for the staged block table.
__device__ inline int quiet_body(const Job job, const Region rg, const double abs_sum, const double *__restrict__ prefix,
                                 const double *__restrict__ rs, double thr,
                                 const double *__restrict__ tmin, const double *__restrict__ tmax,
                                 unsigned long long *__restrict__ work, int first_chunk, int chunk_stride,
                                 long long tab_k0 = 0) {     // tmin / tmax start at end block tab_k0 (a region-local table)
    __shared__ int s_found, s_nwork;
    __shared__ double s_px[2 * ROWS_HALF];        // P[x] of the block's rows (side * 64 + lane)
    __shared__ long long s_ax[2 * ROWS_HALF];     // their absolute prefix indexes
    __shared__ int s_work[Q_WORK];                // undecided (row, end block) pairs
    __shared__ double s_tmx[Q_BLOCKS], s_tmn[Q_BLOCKS];   // block maxima / minima the job touches
    __shared__ double s_pn[2][2 * ROWS_HALF];             // rows of a side and their near ends
    __shared__ double s_b8x[2][16], s_b8n[2][16];         // maxima / minima of every eight of them
    const int tid = threadIdx.x;
    const bool on = tid < 256;                    // larger workgroups: the first 256 threads work, all keep the barriers
    const int L = job.hi - job.lo, half = (L + 1) / 2;
    const double eps = window_eps(rg.n, abs_sum);
    const double T = thr - eps, T2c = T * T * (1.0 - 3e-6);   // a window below T in magnitude cannot reach thr
    const long long base = rg.off + job.region + job.lo;                      // absolute index of the job's P[0]
    const long long a_hi = base + L;                                          // absolute index of the last end
    const long long k_last = a_hi / QB;
    const long long k_base = base / QB;
    const int lane = tid & 63;
    const int w = __builtin_amdgcn_readfirstlane(tid >> 6);
    if (k_last - k_base >= Q_BLOCKS) {           // more end blocks than the staged table holds: full search
        return 1;
    }
    if (tid == 0) s_found = 0;
    for (int i = tid; on && i <= (int)(k_last - k_base); i += 256) {
        s_tmx[i] = tmax[k_base + i - tab_k0];
        s_tmn[i] = tmin[k_base + i - tab_k0];
    }
    for (int chunk = first_chunk; chunk * ROWS_HALF < half; chunk += chunk_stride) {
        wc_sync();                          // previous chunk's queue and rows are done with
        if (s_found) break;
        if (tid == 0) s_nwork = 0;
        if (on) {   // the 64 rows of each side and the 64 prefix entries after them (clipped to the job)
            const int side = tid >> 7, t = tid & 127;
            const int xr_lo = side == 0 ? chunk * ROWS_HALF : L - 1 - (chunk * ROWS_HALF + 63);
            const long long ai = base + (xr_lo < 0 ? 0 : xr_lo) + t;
            const bool valid = ai <= a_hi;
            const double pv = valid ? prefix[ai] : 0.0;
            s_pn[side][t] = pv;
            // minimum and maximum of every eight staged entries (entries past the job's end take no part)
            double mx8 = valid ? pv : -INFINITY, mn8 = valid ? pv : INFINITY;
            for (int o = 1; o < 8; o <<= 1) {
                mx8 = fmax(mx8, __shfl_xor(mx8, o));
                mn8 = fmin(mn8, __shfl_xor(mn8, o));
            }
            if ((t & 7) == 0) { s_b8x[side][t >> 3] = mx8; s_b8n[side][t >> 3] = mn8; }
        }
        wc_sync();
        bool found = false;
        int evals = 0;                            // profiling only: window / bound evaluations of this lane
        for (int side = 0; on && side < 2; ++side) {
            int xr = chunk * ROWS_HALF + lane;
            bool live;
            if (side == 0) {
                live = xr < half;
            } else {
                xr = L - 1 - xr;
                live = xr >= half;
            }
            const long long ax = base + (live ? xr : 0);
            // rows of this side and their near ends sit in LDS: s_pn[side][t] = P[a0 + t], t < 128
            const int xr_lo = side == 0 ? chunk * ROWS_HALF : L - 1 - (chunk * ROWS_HALF + 63);
            const long long a0 = base + (xr_lo < 0 ? 0 : xr_lo);
            const int xo = (int)(ax - a0);                              // 0..63 for live lanes
            const double px = live ? s_pn[side][xo] : 0.0;
            if (w == 0) { s_px[side * ROWS_HALF + lane] = px; s_ax[side * ROWS_HALF + lane] = ax; }
            const long long k_far = ax / QB + 2;                        // first end block that starts past ax + 32
            // near ends, one by one; the four waves take every fourth window length (uniform per
            // wave: its 1/sqrt(len) is a scalar load)
            long long y_near = k_far * QB - 1;
            if (y_near > a_hi) y_near = a_hi;
            const int near = live ? (int)(y_near - ax) : 0;            // <= 63
            // ... the ends in the row's own eight-entry block and the next one (8-15 window lengths) by value,
            // the eight-entry blocks after them up to y_near by the same bound as the far blocks (a block may reach
            // past y_near: a superset, still a bound); a block the bound does not settle is evaluated on the spot
            // -- rare at these lengths, the bound is within a factor ~1.5 of the values
            const int b_own = xo >> 3;
            const int exact = 8 * (b_own + 2) - 1 - xo;                  // 8..15
            const int n_exact = near < exact ? near : exact;
            if (w == 0 && live) evals += n_exact + (near > exact ? (near - exact + 7) / 8 : 0) +
                                         (int)(k_last >= k_far ? k_last - k_far + 1 : 0);
#pragma unroll
            for (int len = 1 + w; len <= 15; len += 4) {
                const double r = rs[len];
                if (len <= n_exact) {
                    const double v = (s_pn[side][xo + len] - px) * r;
                    if (!(fabs(v) + eps < thr)) found = true;
                }
            }
#pragma unroll
            for (int jb = 0; jb < 2; ++jb) {
                const int bi = b_own + 2 + w + 4 * jb;                   // ends 8 bi .. 8 bi + 7 of the staged stretch
                if (live && 8 * bi <= xo + near) {
                    const double t2 = T2c * (double)(8 * bi - xo);       // the shortest window of the pair
                    const double up = s_b8x[side][bi] - px, dn = s_b8n[side][bi] - px;
                    if ((up > 0.0 && up * up >= t2) || (dn < 0.0 && dn * dn >= t2)) {
                        for (int e = 0; e < 8; ++e) {
                            const int len = 8 * bi + e - xo;
                            if (len <= near) {
                                const double v = (s_pn[side][xo + len] - px) * rs[len];
                                if (!(fabs(v) + eps < thr)) found = true;
                            }
                        }
                    }
                }
            }
            // far ends, one bound per block; the four waves take every fourth block.  The bound
            // |extreme - P[x]| / sqrt(min len) is compared in squared form (no table gather, no
            // reciprocal square root in the loop).  Undecided (row, block) pairs are queued:
            // evaluating them here would keep a whole wave waiting for the few lanes that need it.
            const long long k0 = a0 / QB + 2;
            if (!(T > 0.0)) found = true;        // nothing can be certified below a non-positive bound
            for (long long k = k0 + w; k <= k_last; k += 4) {
                const double mx = s_tmx[k - k_base], mn = s_tmn[k - k_base];     // wave-uniform LDS reads
                if (live && k >= k_far) {
                    // (max - P[x]) / sqrt(min len) + eps < thr, squared: no reciprocal square root,
                    // no division (the 3e-6 margin dwarfs every rounding on the way)
                    const double t2 = T2c * (double)(int)(k * QB - ax);
                    const double up = mx - px, dn = mn - px;
                    if ((up > 0.0 && up * up >= t2) || (dn < 0.0 && dn * dn >= t2)) {
                        const int at = atomicAdd(&s_nwork, 1);
                        if (at < Q_WORK) s_work[at] = ((side * ROWS_HALF + lane) << 24) | (int)(k - k_base);
                        else found = true;                              // queue full: give up the certificate
                    }
                }
            }
        }
        wc_sync();
        // queued pairs: 32 ends each, eight pairs per trip
        const int nwork = s_nwork < Q_WORK ? s_nwork : Q_WORK;
        for (int wk = tid >> 5; on && wk < nwork; wk += 8) {
            const int row = s_work[wk] >> 24;
            const long long ay = (k_base + (s_work[wk] & 0xFFFFFF)) * QB + (tid & 31);
            if (ay <= a_hi) {
                const double v = (prefix[ay] - s_px[row]) * rs[ay - s_ax[row]];
                if (!(fabs(v) + eps < thr)) found = true;
            }
        }
        if (found) s_found = 1;
        if (work) {
            if (tid < 64) {
                for (int o = 32; o > 0; o >>= 1) evals += __shfl_xor(evals, o);
                if (tid == 0) atomicAdd(work + 1, (unsigned long long)evals + 32ull * (unsigned long long)nwork);
            }
        }
    }
    wc_sync();
    const int found = s_found;
    wc_sync();                              // the shared tables may be reused by the caller's next job
    return found;
}

__global__ __launch_bounds__(256) void k_seg_quiet(Job *__restrict__ jobs, int n_jobs,
                                                   const Region *__restrict__ regions,
                                                   const double *__restrict__ prefix, const double *__restrict__ rs,
                                                   const double *__restrict__ reg_abs,
                                                   const int *__restrict__ reg_flag, double thr,
                                                   const double *__restrict__ tmin,
                                                   const double *__restrict__ tmax,
                                                   unsigned long long *__restrict__ work,
                                                   const int *__restrict__ n_jobs_dev, int n_regions) {
    // gridDim.x workgroups share a job: each takes the row blocks blockIdx.x, + gridDim.x, ...
    // (many jobs: one workgroup per job, the block table is staged once; few jobs: all row
    // blocks in parallel)
    const int j = blockIdx.y, tid = threadIdx.x;
    if (n_jobs_dev) n_jobs = *n_jobs_dev < n_jobs ? *n_jobs_dev : n_jobs;   // device-side round loop: n_jobs is the grid's bound
    if (j >= n_jobs) return;
    // In a first round job j IS region j: the region's words are requested together with the job instead of after
    // it (a workgroup lives ~7 us here, a dependent load is more than one of them)
    const bool guess = j < n_regions;
    Region rg = guess ? regions[j] : Region{0, 0, 0};
    double abs_sum = guess ? reg_abs[j] : 0.0;
    int flag = guess ? reg_flag[j] : 0;
    const Job job = jobs[j];
    const int L = job.hi - job.lo, half = (L + 1) / 2;
    if (L <= 0 || (int)blockIdx.x * ROWS_HALF >= half) return;
    if (!guess || job.region != j) {
        rg = regions[job.region];
        abs_sum = reg_abs[job.region];
        flag = reg_flag[job.region];
    }
    if (!flag) return;                            // non-finite region: classify sends it to the brute path
    if (job.pad) return;                          // a sibling workgroup already found a window
    if (quiet_body(job, rg, abs_sum, prefix, rs, thr, tmin, tmax, work, blockIdx.x, gridDim.x) && tid == 0)
        jobs[j].pad = 1;
}

// ------------------------------------------------------------ bound-driven search ----
// The host-driven rounds (regions beyond the tree kernel, i.e. bin sizes below ~125 kb; every call of
// wc_stouffer_segments without -mineffectsize) do not evaluate every window of a job that may hold a call.
// The same per-(row, 32-end block) bounds that certify a quiet job also LOCATE the extremes of a loud one:
//     ub(x, k) = max(max P over block k - P[x], 0) * rs[min len]  >=  every window value of the pair
// (subtraction, scaling and rounding are monotone, so the bound covers the values exactly as the search would
// compute them), the same from below, and for a whole block  (max P - P[x]) * rs[min len + 32]  is a LOWER
// bound of the pair's best window (the window that ends on the block's maximum is at most 31 bins longer).
//   k_seg_seed     per job, from the block tables alone: the window from the minimum of block a to the maximum
//                  of block b is worth at least (max_b - min_a) * rs[(b - a + 1) 32]; the best such value (and
//                  its mirror image) seeds the job's cut max(seed, thr - eps): nothing below it can be the job's
//                  extreme AND matter.  L^2 / 2048 products per job.
//   k_seg_bound    per (job, row block): near windows (< 64 bins) by value, far pairs by bound; a pair whose bound
//                  reaches the cut first raises the cut to its own lower bound, then is queued and evaluated
//                  window by window after the sweep.  The cut only ever rises (to values of windows that exist),
//                  so whatever was skipped lies below the job's final extreme or below thr - eps: partial[] ends
//                  up holding, per row block, extremes that are EXACTLY the full search's whenever they can matter
//                  (>= thr - eps in magnitude), and k_seg_classify reads them as before.  Per row block it also
//                  leaves an upper bound of everything it did NOT evaluate (ChunkBound).
//   k_seg_bcollect the windows within 2 eps of the extremes (k_seg_collect's job): the row blocks whose values or
//                  residual bounds reach the cut, and in them the pairs whose bound does -- only for a side that
//                  can hold a call.
// On a 50 kb batch ~4 % of the windows are touched: 1 bound per 32 windows + the pairs around the extremes and
// the few that reach thr - eps by chance.
struct ChunkBound { double ubmax, lbmin; };
// 1 / sqrt(len) from above and from below for the BOUNDS (the values themselves use the table rs[], as the search
// does): the hardware's float reciprocal square root (1 ulp) widened by 1e-6 -- no table gather in the sweep, whose
// iterations would otherwise each wait out a memory round trip
__device__ inline double rs_above(int len) { return (double)__builtin_amdgcn_rsqf((float)len) * (1.0 + 1e-6); }
__device__ inline double rs_below(int len) { return (double)__builtin_amdgcn_rsqf((float)len) * (1.0 - 1e-6); }
constexpr int BS_QUEUE = 2048;      // (row, end block) pairs a row block can queue for evaluation; more: the block is scanned in full

// One row block of a job by a 256-thread workgroup.  MODE 0: bounds + evaluation of what reaches the running cut
// (k_seg_bound; cut_hi / cut_lo come in as the job's cuts so far and go out raised); 2: candidates
// (k_seg_bcollect; fixed cuts, emit(v, x, y) with region coordinates).
// s_tmx / s_tmn: the job's slice of the block tables (table_ok: it fits).  Every thread returns its share
// of vmax / vmin (values seen) and, MODE 0, of ubmax / lbmin (bounds of the pairs that were NOT evaluated).
template <int MODE, class F>
__device__ inline void bscan_chunk(const Job &job, const int chunk, const long long base, const double *__restrict__ prefix,
                                   const double *__restrict__ rs, double &cut_hi, double &cut_lo,
                                   const double *s_tmx, const double *s_tmn, const long long k_base, const long long k_last,
                                   const double *s_tmx2, const double *s_tmn2,
                                   const bool table_ok, double &vmax, double &vmin, double &ubmax, double &lbmin, F emit,
                                   unsigned long long *__restrict__ work) {
    __shared__ double s_pn[2][2 * ROWS_HALF];     // rows of a side and their near ends
    __shared__ double s_px[2 * ROWS_HALF];        // P[x] of the block's rows (side * 64 + lane)
    __shared__ long long s_ax[2 * ROWS_HALF];     // their absolute prefix indexes
    __shared__ double s_b0x[2][16], s_b0n[2][16]; // maximum / minimum of every 8 entries of s_pn (entries of the job only)
    __shared__ int s_work[BS_QUEUE];
    __shared__ int s_nwork;
    __shared__ unsigned long long s_cut[2];       // the running cuts, ordered bit patterns: [0] cut_hi, [1] -cut_lo
    const int tid = threadIdx.x, lane = tid & 63;
    const int w = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int L = job.hi - job.lo, half = (L + 1) / 2;
    const long long a_hi = base + L;              // absolute index of the last end
    bool full = !table_ok;
    int evals = 0, wins = 0;                      // profiling only: bounds / windows by value of this thread
    wc_sync();                              // the previous chunk's shared state is done with
    if (tid == 0) {
        s_nwork = 0;
        s_cut[0] = wc::f64_ordered(cut_hi);
        s_cut[1] = wc::f64_ordered(-cut_lo);
    }
    if (!full) {
        {   // the 64 rows of each side and the 64 prefix entries after them (clipped to the job)
            const int side = tid >> 7, t = tid & 127;
            const int xr_lo = side == 0 ? chunk * ROWS_HALF : L - 1 - (chunk * ROWS_HALF + 63);
            const long long ai = base + (xr_lo < 0 ? 0 : xr_lo) + t;
            s_pn[side][t] = ai <= a_hi ? prefix[ai] : 0.0;
        }
        wc_sync();
        if (tid < 32) {      // 8-entry blocks of the staged stretches: the near windows beyond 8..15 bins go by bound too
            const int side = tid >> 4, b = tid & 15;
            const int xr_lo = side == 0 ? chunk * ROWS_HALF : L - 1 - (chunk * ROWS_HALF + 63);
            const long long a0 = base + (xr_lo < 0 ? 0 : xr_lo);
            double mx = -INFINITY, mn = INFINITY;
            for (int e = 0; e < 8; ++e)
                if (a0 + 8 * b + e <= a_hi) { mx = fmax(mx, s_pn[side][8 * b + e]); mn = fmin(mn, s_pn[side][8 * b + e]); }
            s_b0x[side][b] = mx;
            s_b0n[side][b] = mn;
        }
        wc_sync();
        for (int side = 0; side < 2; ++side) {
            int xr = chunk * ROWS_HALF + lane;
            bool live;
            if (side == 0) {
                live = xr < half;
            } else {
                xr = L - 1 - xr;
                live = xr >= half;
            }
            const long long ax = base + (live ? xr : 0);
            const int xr_lo = side == 0 ? chunk * ROWS_HALF : L - 1 - (chunk * ROWS_HALF + 63);
            const long long a0 = base + (xr_lo < 0 ? 0 : xr_lo);
            const int xo = (int)(ax - a0);                              // 0..63 for live lanes
            const double px = live ? s_pn[side][xo] : 0.0;
            if (w == 0) { s_px[side * ROWS_HALF + lane] = px; s_ax[side * ROWS_HALF + lane] = ax; }
            const long long k_far = ax / QB + 2;                        // first end block that starts past ax + 32
            long long y_near = k_far * QB - 1;
            if (y_near > a_hi) y_near = a_hi;
            const int near = live ? (int)(y_near - ax) : 0;             // <= 63
            const int x = job.lo + xr;
            // the cut as the workgroup has raised it so far (MODE 0), re-read now and then: a stale value only costs
            // evaluations
            double chi = cut_hi, clo = cut_lo;
            auto refresh = [&]() {
                if (MODE == 0) {
                    chi = wc::f64_from_ordered(__hip_atomic_load(&s_cut[0], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP));
                    clo = -wc::f64_from_ordered(__hip_atomic_load(&s_cut[1], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP));
                }
            };
            refresh();
            auto window = [&](const double v, const int len) {
                ++wins;
                if (MODE == 2) {
                    if (v >= cut_hi || v <= cut_lo) emit(v, x, x + len - 1);
                } else {
                    vmax = fmax(vmax, v);
                    vmin = fmin(vmin, v);
                }
            };
            // near ends: the first 8..15 by value (the four waves take every fourth length: uniform per wave, its
            // 1/sqrt(len) is a scalar load), the rest per 8-entry block of the staged stretch by bound, a block that
            // reaches the cut window by window on the spot
            const int fb = (xo + 16) >> 3;                              // first 8-block that starts at least 9 past the row
            const int n_exact = near < 8 * fb - 1 - xo ? near : 8 * fb - 1 - xo;
#pragma unroll 4
            for (int len = 1 + w; len <= 15; len += 4) {
                const double r = rs[len];
                if (len <= n_exact) window((s_pn[side][xo + len] - px) * r, len);
            }
            for (int b = fb + w; b < 16; b += 4) {
                const int minlen = 8 * b - xo;                          // 9..
                if (minlen > near) break;
                const double r = rs_above(minlen);
                const double ub = fmax(s_b0x[side][b] - px, 0.0) * r, lb = fmin(s_b0n[side][b] - px, 0.0) * r;
                ++evals;
                if (ub >= chi || lb <= clo) {
                    // queued like the far pairs (a wave that evaluated on the spot would walk the eight windows
                    // whenever ANY of its lanes has such a block)
                    const int at = atomicAdd(&s_nwork, 1);
                    if (at < BS_QUEUE) s_work[at] = ((side * ROWS_HALF + lane) << 24) | (1 << 23) | b;
                } else if (MODE == 0) {
                    ubmax = fmax(ubmax, ub);
                    lbmin = fmin(lbmin, lb);
                }
            }
            // far ends, one bound per block: 32-end blocks up to lengths of FAR2, 128-end blocks beyond (the first
            // second-level block of a row is the first one that starts at least FAR2 past it)
            // (32-bit indexes relative to the start of the job's first second-level block: the sweep's index
            // arithmetic stays off the 64-bit vector path)
            const long long org = (k_base / 4) * QB2;
            const int rax = (int)(ax - org), ra0 = (int)(a0 - org), rhi = (int)(a_hi - org);
            const int d1 = (int)(k_base & 3);                           // first-level block rel / 32 is table entry rel / 32 - d1
            const int r_far = rax / QB + 2, r_last = rhi / QB;
            const int r0 = ra0 / QB + 2;
            const int r2_first = (rax + FAR2 + QB2 - 1) / QB2;          // this row's first second-level block
            const int r1_end = r2_first * 4;                            // ... and the first-level blocks before it
            const int r2_last = rhi / QB2;
            const int r1_stop = ((ra0 + 63 + FAR2 + QB2 - 1) / QB2) * 4;     // wave-uniform: beyond every lane's r1_end
            auto pair = [&](const double mx, const double mn, const int minlen, const int span, const bool whole,
                            const int first_blk, const int n_blk) {
                const double r = rs_above(minlen);
                const double up = mx - px, dn = mn - px;
                const double ub = fmax(up, 0.0) * r, lb = fmin(dn, 0.0) * r;
                ++evals;
                if (ub >= chi || lb <= clo) {
                    if (MODE == 0 && whole) {
                        // a whole block inside the job: the window that ends on the block's extreme is less than
                        // `span` bins longer than minlen, so its value is at least up * rs[minlen + span] -- raise the
                        // cut to it before anybody queues more of what it rules out
                        const double r2 = rs_below(minlen + span);
                        if (ub >= chi && up > 0.0) atomicMax(&s_cut[0], wc::f64_ordered(up * r2));
                        if (lb <= clo && dn < 0.0) atomicMax(&s_cut[1], wc::f64_ordered(-(dn * r2)));
                        refresh();
                    }
                    const int at = atomicAdd(&s_nwork, n_blk);          // evaluated as n_blk first-level blocks
                    for (int q = 0; q < n_blk; ++q)
                        if (at + q < BS_QUEUE) s_work[at + q] = ((side * ROWS_HALF + lane) << 24) | (first_blk + q);
                } else if (MODE == 0) {
                    ubmax = fmax(ubmax, ub);
                    lbmin = fmin(lbmin, lb);
                }
            };
            int since = 0;
            for (int r1 = r0 + w; r1 <= r_last && r1 < r1_stop; r1 += 4) {
                const double mx = s_tmx[r1 - d1], mn = s_tmn[r1 - d1];           // wave-uniform LDS reads
                if (live && r1 >= r_far && r1 < r1_end) pair(mx, mn, r1 * QB - rax, QB, r1 < r_last, r1 - d1, 1);
                if ((++since & 7) == 0) refresh();
            }
            for (int r2 = (ra0 + FAR2) / QB2 + w; r2 <= r2_last; r2 += 4) {
                const double mx = s_tmx2[r2], mn = s_tmn2[r2];
                if (live && r2 >= r2_first) pair(mx, mn, r2 * QB2 - rax, QB2, r2 < r2_last, r2 * 4 - d1, 4);
                if ((++since & 7) == 0) refresh();
            }
        }
        wc_sync();
        const int nwork = s_nwork;
        if (nwork > BS_QUEUE) {
            full = true;                          // workgroup-uniform: scan the whole block instead
        } else {
            for (int wk = tid >> 5; wk < nwork; wk += 8) {      // 32 ends per pair, eight pairs per trip
                const int row = s_work[wk] >> 24;
                long long ay = (k_base + (s_work[wk] & 0x7FFFFF)) * QB + (tid & 31);
                long long last = a_hi;
                if (s_work[wk] & (1 << 23)) {
                    // a near block: eight entries of the row's side, up to the row's last near end
                    const int xr_lo = (row >> 6) == 0 ? chunk * ROWS_HALF : L - 1 - (chunk * ROWS_HALF + 63);
                    ay = base + (xr_lo < 0 ? 0 : xr_lo) + 8 * (s_work[wk] & 15) + (tid & 31);
                    const long long y_near = (s_ax[row] / QB + 2) * QB - 1;
                    last = (tid & 31) < 8 ? (y_near < a_hi ? y_near : a_hi) : -1;
                }
                if (ay <= last) {
                    ++wins;
                    const int len = (int)(ay - s_ax[row]);
                    const double v = (prefix[ay] - s_px[row]) * rs[len];
                    if (MODE == 2) {
                        const int x = job.lo + (int)(s_ax[row] - base);
                        if (v >= cut_hi || v <= cut_lo) emit(v, x, x + len - 1);
                    } else {
                        vmax = fmax(vmax, v);
                        vmin = fmin(vmin, v);
                    }
                }
            }
        }
    }
    if (full) {
        // every window of the block (jobs longer than the staged tables cover, queue overflow): the plain scan
        ScanCtx c;
        c.lo = job.lo; c.hi = job.hi; c.L = L; c.half = half; c.chunk = chunk;
        c.P = prefix + (base - job.lo);
        c.wm = WindowMask{nullptr, 0, 0};
        scan_chunk(c, rs, tid, [&](double v, int x, int y) {
            ++wins;
            if (MODE == 2) {
                if (v >= cut_hi || v <= cut_lo) emit(v, x, y);
            } else {
                vmax = fmax(vmax, v);
                vmin = fmin(vmin, v);
            }
        });
        if (MODE == 0) { ubmax = -INFINITY; lbmin = INFINITY; }    // nothing left unevaluated in this block
    }
    if (MODE == 0) {
        wc_sync();
        cut_hi = wc::f64_from_ordered(s_cut[0]);                    // (the values seen join it in the caller's reduce)
        cut_lo = -wc::f64_from_ordered(s_cut[1]);
    }
    if (work) {
        // spread over 64 counter pairs (wc_test_profile_read adds them up): one address for every wave of the grid
        // would serialise the launch
        for (int o = 32; o > 0; o >>= 1) { evals += __shfl_xor(evals, o); wins += __shfl_xor(wins, o); }
        const int slot = (int)((blockIdx.x * 7u + blockIdx.y * 13u + (unsigned)w) & 63u);
        if (lane == 0) {
            atomicAdd(work + 2 * slot, (unsigned long long)wins);
            atomicAdd(work + 2 * slot + 1, (unsigned long long)evals);
        }
    }
}

// workgroup reduction of four doubles (max, min, max, min); every thread gets the results
__device__ inline void block_minmax4(double &a_max, double &a_min, double &b_max, double &b_min, int tid) {
    __shared__ double red[4][4];
    for (int o = 32; o > 0; o >>= 1) {
        a_max = fmax(a_max, __shfl_xor(a_max, o));
        a_min = fmin(a_min, __shfl_xor(a_min, o));
        b_max = fmax(b_max, __shfl_xor(b_max, o));
        b_min = fmin(b_min, __shfl_xor(b_min, o));
    }
    wc_sync();
    if ((tid & 63) == 0) { red[0][tid >> 6] = a_max; red[1][tid >> 6] = a_min; red[2][tid >> 6] = b_max; red[3][tid >> 6] = b_min; }
    wc_sync();
    a_max = fmax(fmax(red[0][0], red[0][1]), fmax(red[0][2], red[0][3]));
    a_min = fmin(fmin(red[1][0], red[1][1]), fmin(red[1][2], red[1][3]));
    b_max = fmax(fmax(red[2][0], red[2][1]), fmax(red[2][2], red[2][3]));
    b_min = fmin(fmin(red[3][0], red[3][1]), fmin(red[3][2], red[3][3]));
}

// the job's slice of the block tables into LDS; false when it does not fit (the job is scanned in full)
__device__ inline bool stage_block_tables(const long long base, const int L, const double *__restrict__ tmin,
                                          const double *__restrict__ tmax, double *s_tmx, double *s_tmn,
                                          long long &k_base, long long &k_last, int tid,
                                          const double *__restrict__ tmin2 = nullptr,
                                          const double *__restrict__ tmax2 = nullptr, double *s_tmx2 = nullptr,
                                          double *s_tmn2 = nullptr) {
    k_last = (base + L) / QB;
    k_base = base / QB;
    if (k_last - k_base >= Q_BLOCKS) return false;
    for (int i = tid; i <= (int)(k_last - k_base); i += 256) {
        s_tmx[i] = tmax[k_base + i];
        s_tmn[i] = tmin[k_base + i];
    }
    if (tmin2) {      // the second-level blocks that overlap the slice
        const long long k2_base = k_base / 4, k2_last = (base + L) / QB2;
        for (int i = tid; i <= (int)(k2_last - k2_base); i += 256) {
            s_tmx2[i] = tmax2[k2_base + i];
            s_tmn2[i] = tmin2[k2_base + i];
        }
    }
    return true;
}

// The job's starting cuts from the block tables alone (see the header of this section).  cuts[2 j] = cut_hi,
// cuts[2 j + 1] = -cut_lo as ordered bit patterns (k_seg_bound raises them with atomicMax).
__global__ __launch_bounds__(256) void k_seg_seed(const Job *__restrict__ jobs, int n_jobs,
                                                  const Region *__restrict__ regions, const double *__restrict__ rs,
                                                  const double *__restrict__ reg_abs, const int *__restrict__ reg_flag,
                                                  double thr, const double *__restrict__ tmin,
                                                  const double *__restrict__ tmax,
                                                  unsigned long long *__restrict__ cuts) {
    __shared__ double s_tmx[Q_BLOCKS], s_tmn[Q_BLOCKS];
    const int j = blockIdx.x, tid = threadIdx.x;
    if (j >= n_jobs) return;
    const Job job = jobs[j];
    const int L = job.hi - job.lo;
    const double eps = window_eps(regions[job.region].n, reg_abs[job.region]);
    const double T = thr - eps;
    double hi = T, lo = -T, d0 = -INFINITY, d1 = INFINITY;
    if (L > 0 && reg_flag[job.region]) {
        const long long base = regions[job.region].off + job.region + job.lo;
        long long k_base, k_last;
        if (stage_block_tables(base, L, tmin, tmax, s_tmx, s_tmn, k_base, k_last, tid)) {
            wc_sync();
            // blocks that lie wholly inside the job's prefix slice [base, base + L]
            const int b_first = (int)((base + QB - 1) / QB - k_base);
            const int b_last = (int)((base + L + 1) / QB - 1 - k_base);
            const int nb = b_last - b_first + 1;
            if (nb >= 2) {
                // pair (a, b), a < b: thread t takes the pairs with distance d = b - a = 1 + t, 1 + t + 256, ...
                for (int d = 1 + tid; d < nb; d += 256) {
                    const double r = rs[(d + 1) * QB];
                    for (int a = b_first; a + d <= b_last; ++a) {
                        const double up = s_tmx[a + d] - s_tmn[a], dn = s_tmn[a + d] - s_tmx[a];
                        if (up > 0.0) hi = fmax(hi, up * r);
                        if (dn < 0.0) lo = fmin(lo, dn * r);
                    }
                }
            }
        }
    }
    block_minmax4(hi, lo, d0, d1, tid);
    if (tid == 0) {
        cuts[2 * j] = wc::f64_ordered(hi);
        cuts[2 * j + 1] = wc::f64_ordered(-lo);
    }
}

__global__ __launch_bounds__(256) void k_seg_bound(const Job *__restrict__ jobs, int n_jobs,
                                                   const Region *__restrict__ regions,
                                                   const double *__restrict__ prefix, const double *__restrict__ rs,
                                                   const int *__restrict__ reg_flag,
                                                   const double *__restrict__ tmin, const double *__restrict__ tmax,
                                                   const double *__restrict__ tmin2, const double *__restrict__ tmax2,
                                                   int max_chunks, Extreme *__restrict__ partial,
                                                   ChunkBound *__restrict__ cbound, unsigned long long *__restrict__ cuts,
                                                   int *__restrict__ counters, int *__restrict__ next_count,
                                                   unsigned long long *__restrict__ work) {
    __shared__ double s_tmx[Q_BLOCKS], s_tmn[Q_BLOCKS], s_tmx2[Q_BLOCKS / 4 + 2], s_tmn2[Q_BLOCKS / 4 + 2];
    const int j = blockIdx.y, tid = threadIdx.x;
    // first kernel of a round that touches the counters: next-jobs / hot / brute counts start at zero
    if (j == 0 && blockIdx.x == 0 && tid == 0) { *next_count = 0; counters[2] = 0; counters[3] = 0; }
    if (j >= n_jobs) return;
    const Job job = jobs[j];
    const int L = job.hi - job.lo, half = (L + 1) / 2;
    if (L <= 0 || (int)blockIdx.x * ROWS_HALF >= half) return;
    if (!reg_flag[job.region]) return;            // non-finite region: classify sends it to the brute path
    const long long base = regions[job.region].off + job.region + job.lo;    // absolute index of the job's P[0]
    long long k_base, k_last;
    const bool table_ok = stage_block_tables(base, L, tmin, tmax, s_tmx, s_tmn, k_base, k_last, tid, tmin2, tmax2, s_tmx2, s_tmn2);
    for (int chunk = blockIdx.x; chunk * ROWS_HALF < half; chunk += gridDim.x) {
        // the job's cuts as every workgroup has raised them so far (a stale value only costs evaluations)
        double cut_hi = wc::f64_from_ordered(__hip_atomic_load(&cuts[2 * j], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT));
        double cut_lo = -wc::f64_from_ordered(__hip_atomic_load(&cuts[2 * j + 1], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT));
        const double in_hi = cut_hi, in_lo = cut_lo;
        double vmax = -INFINITY, vmin = INFINITY, ubmax = -INFINITY, lbmin = INFINITY;
        bscan_chunk<0>(job, chunk, base, prefix, rs, cut_hi, cut_lo, s_tmx, s_tmn, k_base, k_last, s_tmx2, s_tmn2, table_ok, vmax, vmin, ubmax,
                       lbmin, [](double, int, int) {}, work);
        block_minmax4(vmax, vmin, ubmax, lbmin, tid);
        if (tid == 0) {
            Extreme e;
            e.maxv = vmax; e.minv = vmin;
            e.max_x = e.max_y = e.min_x = e.min_y = -1;
            partial[(int64_t)j * max_chunks + chunk] = e;
            // upper bound of every window of the block: the bounds of what was not evaluated and the values of what was
            ChunkBound cb;
            cb.ubmax = fmax(ubmax, vmax);
            cb.lbmin = fmin(lbmin, vmin);
            cbound[(int64_t)j * max_chunks + chunk] = cb;
            // values of windows that exist (and lower bounds of such) raise the job's cuts for everybody
            cut_hi = fmax(cut_hi, vmax);
            cut_lo = fmin(cut_lo, vmin);
            if (cut_hi > in_hi) atomicMax(&cuts[2 * j], wc::f64_ordered(cut_hi));
            if (cut_lo < in_lo) atomicMax(&cuts[2 * j + 1], wc::f64_ordered(-cut_lo));
        }
    }
}

// k_seg_collect's job on the bound tables: the windows within 2 eps of a hot job's extremes, listed for exact
// scoring -- only for a side that can hold a call (the other side's "extreme" is merely a value below thr - eps:
// it cannot win against a call, and an empty list is what k_seg_decide needs to see then).
__global__ __launch_bounds__(256) void k_seg_bcollect(const Job *__restrict__ jobs, const int *__restrict__ hot,
                                                      const int *__restrict__ counters,
                                                      const Region *__restrict__ regions,
                                                      const double *__restrict__ prefix, const double *__restrict__ rs,
                                                      const double *__restrict__ reg_abs, double thr,
                                                      const Extreme *__restrict__ job_res,
                                                      const double *__restrict__ tmin, const double *__restrict__ tmax,
                                                      const double *__restrict__ tmin2, const double *__restrict__ tmax2,
                                                      int max_chunks, const ChunkBound *__restrict__ cbound,
                                                      int2 *__restrict__ cand, int *__restrict__ cand_cnt,
                                                      unsigned long long *__restrict__ work) {
    __shared__ double s_tmx[Q_BLOCKS], s_tmn[Q_BLOCKS], s_tmx2[Q_BLOCKS / 4 + 2], s_tmn2[Q_BLOCKS / 4 + 2];
    const int h = blockIdx.y, tid = threadIdx.x;
    if (h >= counters[2]) return;
    const int j = hot[h];
    const Job job = jobs[j];
    const int L = job.hi - job.lo, half = (L + 1) / 2;
    if ((int)blockIdx.x * ROWS_HALF >= half) return;
    const int nch = (half + ROWS_HALF - 1) / ROWS_HALF;
    const ChunkBound *cj = cbound + (int64_t)j * max_chunks;
    const double eps = window_eps(regions[job.region].n, reg_abs[job.region]);
    const Extreme e = job_res[j];
    double hi_cut = !(e.maxv + eps < thr) ? e.maxv - 2.0 * eps : INFINITY;
    double lo_cut = !(-e.minv + eps < thr) ? e.minv + 2.0 * eps : -INFINITY;
    int any = 0;
    for (int chunk = blockIdx.x; chunk < nch; chunk += gridDim.x) any |= (cj[chunk].ubmax >= hi_cut) | (cj[chunk].lbmin <= lo_cut);
    if (!any) return;
    const long long base = regions[job.region].off + job.region + job.lo;
    long long k_base, k_last;
    const bool table_ok = stage_block_tables(base, L, tmin, tmax, s_tmx, s_tmn, k_base, k_last, tid, tmin2, tmax2, s_tmx2, s_tmn2);
    for (int chunk = blockIdx.x; chunk < nch; chunk += gridDim.x) {
        if (!(cj[chunk].ubmax >= hi_cut || cj[chunk].lbmin <= lo_cut)) continue;
        double d0 = -INFINITY, d1 = INFINITY, d2 = -INFINITY, d3 = INFINITY;
        bscan_chunk<2>(job, chunk, base, prefix, rs, hi_cut, lo_cut, s_tmx, s_tmn, k_base, k_last, s_tmx2, s_tmn2, table_ok, d0, d1, d2, d3,
                       [&](double v, int x, int y) {
                           if (v >= hi_cut) {
                               const int at = atomicAdd(&cand_cnt[2 * h], 1);
                               if (at < CAND_CAP) cand[((int64_t)2 * h) * CAND_CAP + at] = make_int2(x, y);
                           }
                           if (v <= lo_cut) {
                               const int at = atomicAdd(&cand_cnt[2 * h + 1], 1);
                               if (at < CAND_CAP) cand[((int64_t)2 * h + 1) * CAND_CAP + at] = make_int2(x, y);
                           }
                       }, work);
    }
}

// ------------------------------------------------------------ cell-driven search ----
// The bound-driven search above gives every block of 64 + 64 rows to a workgroup and sweeps, per ROW, one bound
// per end block: ~60 evaluations per row and, per row block, a chain of staging, sweep, queue pass and reduction
// (phase clocks of round 5: 21 k ticks per row block, no phase above a quarter).  The same bounds hold for whole
// CELLS -- a block of rows against a block of ends, both read from block tables:
//     ub(A, K) = max(max P over K - min P over A, 0) * rs_above(shortest window of the cell)  >=  every window of it
// -- so 128 x 128, 32 x 32 and 8 x 8 windows are decided at a time and only the cells that reach the cut are
// taken apart: cell -> rows (one bound each) -> windows.  A job's windows are tiled exactly once by
//   * a row against its own 8-entry block (the entries after it) and against the next one          one bound each,
//   * 8 x 8 cells two or more 8-blocks on, up to the end of the row's next 32-block (below 64 bins),
//   * 32 x 32 cells (A, K), K >= A + 2, whose 128-blocks are the same or adjacent (below 256 bins),
//   * 128 x 128 cells (A2, K2), K2 >= A2 + 2,
// blocks aligned to the concatenated prefix array like the tables (a block that straddles the job's ends bounds a
// superset).  The cuts rise exactly as in bscan_chunk -- only to values of windows that exist or lower bounds of
// such, so whatever is skipped lies below the job's final extreme or below thr - eps -- and every test keeps a
// margin of 2 eps: a window within 2 eps of the FINAL extreme is therefore always evaluated, and it is recorded
// when it is (it lies within 2 eps of the cut of that moment), so the candidate list of a job that may hold a call
// (k_seg_bcollect's product) falls out of the same pass; a second pass with fixed cuts (MODE 2) only runs when
// that record overflows.
// A job is worked on by parts(L) workgroups (one per 1024 rows): part p takes the near windows of trip p and every
// parts-th row block of the cell sweeps; the parts share the job's cuts through global atomics, leave their
// extremes and records in the job's state and the last one to arrive (a ticket) classifies the job and writes the
// list.  All indexes inside a job are 32-bit, relative to the start of its first 128-block.
constexpr int CJ_MAXLEN = 8192;                 // longest job of this path
constexpr int CJ_T1 = (CJ_MAXLEN + 128) / QB + 2;
constexpr int CJ_T2 = (CJ_MAXLEN + 128) / QB2 + 2;
constexpr int CJ_Q2 = 512;                      // loud 128 x 128 cells of a part (a part owns at most ~300)
constexpr int CJ_Q1 = 320;                      // loud 32 x 32 cells of a part's band next to the near windows (at most ~260)
constexpr int CJ_ITEMQ = 1024;                  // loud (row, block) pairs: near sweep (more: the pusher evaluates the block itself),
                                                // and the rows of 32 cells at a time
constexpr int CJ_RPT = 4;                       // rows per thread and near-sweep trip
constexpr int CJ_TRIP = 256 * CJ_RPT;           // rows per trip = rows per part
constexpr int CJ_MAXPARTS = CJ_MAXLEN / CJ_TRIP;
constexpr int CJ_PN = CJ_TRIP + 64;             // the trip's rows and the 64 entries after them
constexpr int CJ_NB8 = CJ_PN / 8 + 2;           // 8-entry blocks that overlap them
constexpr int CJ_LOADS = (CJ_PN + 255) / 256;
constexpr int CJ_REC = 64;                      // near-extreme windows a part records per side
constexpr int CJ_GREC = 128;                    // ... and a job keeps per side (more: the second pass)
__host__ __device__ inline int cell_parts(int L) {
    const int p = (L + CJ_TRIP - 1) / CJ_TRIP;
    return p < 1 ? 1 : (p > CJ_MAXPARTS ? CJ_MAXPARTS : p);
}
struct CellRec { int x, y; double v; };
struct CellJobState {                           // per job and round, zeroed by the host before the launch
    unsigned long long cut[2];                  // ordered bit patterns: [0] cut_hi, [1] -cut_lo (0 = below everything)
    unsigned long long ext[2];                  // ... of the largest value seen, of minus the smallest
    int ticket, n_rec[2], overflow;
};
// Development aid (tools/cell_clocks_variant.py defines WC_CELL_CLOCKS): thread 0 of every k_seg_job workgroup books the
// clock ticks of its phases in LDS and adds them to g_dbg[phase] when it leaves.
// WC_CELL_CLOCKS_SWITCH
#ifdef WC_CELL_CLOCKS
#define CJ_CLK(n) do { if (tid == 0) { const unsigned long long t_ = clock64(); sh.clk[n] += t_ - sh.t_prev; sh.t_prev = t_; } } while (0)
#else
#define CJ_CLK(n) do { } while (0)
#endif
struct CellShared {
#ifdef WC_CELL_CLOCKS
    unsigned long long clk[32], t_prev;
#endif
    double tmx[CJ_T1], tmn[CJ_T1], tmx2[CJ_T2], tmn2[CJ_T2];
    double pn[CJ_PN];
    double b8x[CJ_NB8], b8n[CJ_NB8];            // maximum / minimum of the staged entries per 8-entry block
    unsigned int q2[CJ_Q2], q1[CJ_Q1], l1[256], itemq[CJ_ITEMQ];
    // (k_seg_walk's exact evaluation of the candidates reuses pn .. itemq as its four waves' scratch)
    double rsn[64];                             // rs[0 .. 63]
    unsigned long long cut[2];                  // this workgroup's view of the job's cuts
    CellRec rec[2][CJ_REC];
    int n_rec[2];
    int n_q2, n_q1, n_l1, n_items, slot, last;
    int lost;                                   // MODE 0: a cell queue overflowed, loud cells were dropped: the extremes are NOT
                                                // proven -- k_seg_job / k_seg_walk hand the job to a path that needs no bound
};

// the job's geometry in relative indexes (origin: the start of the 128-block that holds the job's first prefix entry)
struct CellGeom {
    const double *P;        // prefix + origin
    int rb, rhi;            // the job's first and last prefix entry (rows: rb .. rhi - 1)
    int job_lo;             // region coordinate of row rb
};

// MODE 0: extremes with rising cuts, near-extreme windows recorded; MODE 2: windows at or beyond fixed cuts
// recorded (the whole job by the calling workgroup).  part / parts: this workgroup's share (MODE 0).
// A near-sweep trip's rows into LDS: thread t < CJ_NB8 owns the t-th 8-entry block that overlaps the trip (aligned to
// the concatenated array, so the first may start before the trip), loads its eight entries -- trip_load, registers
// only, so that a caller can have these loads in flight with others -- and leaves them in pn together with the
// block's extremes over the job's own entries (trip_store; entries past the job's end are staged as 0 and take no
// part in the extremes).  The queue of the trip is emptied.  A barrier before pn / b8x / b8n / n_items are read.
struct TripRegs { double v[8]; };
__device__ inline void trip_load(TripRegs &r, const CellGeom &g, const int a0, const int tid) {
    const int i0 = (((a0 >> 3) + tid) << 3) - a0;
#pragma unroll
    for (int e = 0; e < 8; ++e) {
        const int i = i0 + e;
        r.v[e] = (tid < CJ_NB8 && i >= 0 && i < CJ_PN && a0 + i <= g.rhi) ? g.P[a0 + i] : 0.0;
    }
}
__device__ inline void trip_store(CellShared &sh, const TripRegs &r, const CellGeom &g, const int a0, const int tid) {
    if (tid < CJ_NB8) {
        const int i0 = (((a0 >> 3) + tid) << 3) - a0;
        double mx = -INFINITY, mn = INFINITY;
#pragma unroll
        for (int e = 0; e < 8; ++e) {
            const int i = i0 + e;
            if (i >= 0 && i < CJ_PN) {
                sh.pn[i] = r.v[e];
                if (a0 + i <= g.rhi) { mx = fmax(mx, r.v[e]); mn = fmin(mn, r.v[e]); }
            }
        }
        sh.b8x[tid] = mx;
        sh.b8n[tid] = mn;
    }
    if (tid == 0) sh.n_items = 0;
}

// prestaged: the caller's cell_setup has staged this workgroup's first trip already (and a barrier has passed)
template <int MODE>
__device__ inline void cell_search(CellShared &sh, const CellGeom g, const double *__restrict__ rs, const double eps2,
                                   const double hi_cut, const double lo_cut, unsigned long long *__restrict__ gcut,
                                   const int part, const int parts, double &vmax, double &vmin, int &wins, int &evals,
                                   const int tid, const bool prestaged = false) {
    const double *__restrict__ P = g.P;
    const int rb = g.rb, rhi = g.rhi;
    const int A1f = rb >> 5, A1l = (rhi - 1) >> 5, K1l = rhi >> 5;
    const int A2l = (rhi - 1) >> 7, K2l = rhi >> 7;          // (the first 128-block is block 0)
    const int lane = tid & 63, w = tid >> 6;
    double chi = hi_cut, clo = lo_cut;                       // the cuts less / plus the 2 eps margin
    auto cuts = [&]() {
        if (MODE == 0) {
            chi = wc::f64_from_ordered(__hip_atomic_load(&sh.cut[0], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP)) - eps2;
            clo = -wc::f64_from_ordered(__hip_atomic_load(&sh.cut[1], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP)) + eps2;
        }
    };
    // the job's cuts as the other parts have raised them (thread 0, between barriers)
    auto share_cuts = [&]() {
        if (MODE == 0 && parts > 1 && tid == 0) {
#pragma unroll
            for (int q = 0; q < 2; ++q) {
                const unsigned long long mine = sh.cut[q];
                const unsigned long long theirs = __hip_atomic_load(&gcut[q], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
                if (theirs > mine) sh.cut[q] = theirs;
                else if (mine > theirs) atomicMax(&gcut[q], mine);
            }
        }
    };
    auto record = [&](const int side, const double v, const int ax, const int ay) {
        const int at = atomicAdd(&sh.n_rec[side], 1);
        if (at < CJ_REC) {
            CellRec r;
            r.x = g.job_lo + (ax - rb);
            r.y = r.x + (ay - ax) - 1;
            r.v = v;
            sh.rec[side][at] = r;
        }
    };
    // a window's value: prefix entries ax (row) and ay (end), ay - ax bins
    auto see = [&](const double v, const int ax, const int ay) {
        ++wins;
        if (MODE == 0) {
            vmax = fmax(vmax, v);
            vmin = fmin(vmin, v);
            if (v >= chi || v <= clo) {
                cuts();                                      // (rare: against the cuts as they stand now)
                if (v >= chi) {
                    record(0, v, ax, ay);
                    if (v - eps2 > chi) { atomicMax(&sh.cut[0], wc::f64_ordered(v)); chi = v - eps2; }
                }
                if (v <= clo) {
                    record(1, v, ax, ay);
                    if (v + eps2 < clo) { atomicMax(&sh.cut[1], wc::f64_ordered(-v)); clo = v + eps2; }
                }
            }
        } else {
            if (v >= hi_cut) record(0, v, ax, ay);
            if (v <= lo_cut) record(1, v, ax, ay);
        }
    };
    // ---- near windows (ends before the row's first 32 x 32 cell, i.e. fewer than 64 bins): CJ_TRIP rows per trip,
    // everything of a trip in LDS.  Nothing is evaluated by default: an 8-row block against an 8-end block two or
    // more blocks on is ONE bound from the 8-entry block extremes, a row against its own and the next 8-block one
    // bound each; what reaches the cut is queued as (row, 8-block) pairs and evaluated by eight lanes each.
    const int L = rhi - rb;
    bool staged = MODE == 0 && prestaged;
    for (int r0 = (MODE == 0 ? part : 0) * CJ_TRIP; r0 < L; r0 += (MODE == 0 ? parts : 1) * CJ_TRIP) {
        const int a0 = rb + r0;                              // the trip's first row
        if (!staged) {
            wc_sync();                                       // the previous trip's rows and queue are done with
            TripRegs tr;
            trip_load(tr, g, a0, tid);
            trip_store(sh, tr, g, a0, tid);
            share_cuts();
            wc_sync();
        }
        staged = false;
        CJ_CLK(20);
        const int k8_0 = a0 >> 3;
        cuts();
        const int rows_here = L - r0 < CJ_TRIP ? L - r0 : CJ_TRIP;           // rows t = 0 .. rows_here - 1 of this trip
        auto y_near_of = [&](const int ax) {                 // the last end before the row's first 32 x 32 cell
            const int y = (((ax >> 5) + 2) << 5) - 1;
            return y > rhi ? rhi : y;
        };
        // rows t_lo .. t_hi of the trip against the staged 8-block j8: ONE reservation for all of them
        auto push_rows = [&](const int t_lo, const int t_hi, const int j8) {
            const int at = atomicAdd(&sh.n_items, t_hi - t_lo + 1);
            for (int t = t_lo; t <= t_hi; ++t) {
                const int slot = at + (t - t_lo);
                if (slot < CJ_ITEMQ) {
                    sh.itemq[slot] = ((unsigned int)t << 8) | (unsigned int)j8;
                } else {                                     // queue full: the pusher walks the block itself
                    const int ax = a0 + t, y_near = y_near_of(ax);
                    for (int e = 0; e < 8; ++e) {
                        const int ay = ((k8_0 + j8) << 3) + e;
                        if (ay > ax && ay <= y_near) see((sh.pn[ay - a0] - sh.pn[t]) * sh.rsn[ay - ax], ax, ay);
                    }
                }
            }
        };
        // The bounds first, every thread's share unrolled (their LDS reads in flight together), as a mask of what
        // reaches the cut; the pushes -- rare -- afterwards in one loop.
        // bits 0..3: 8 x 8 cells -- bits 0..2: the 8-row block tid / 2 against its end blocks 2 + 3 (tid & 1) + u blocks
        // on (two threads share a row block's six), bit 3: the 129th row block of a trip that does not start on a
        // multiple of eight, its six end blocks by threads 0..5;
        // bits 4..7: row tid + 256 q against its own 8-block (the entries after it), bits 8..11: against the next one
        static_assert(CJ_TRIP == 1024 && CJ_RPT <= 4, "the mask of loud bounds: 128 + 1 row blocks, four bits per kind");
        constexpr int PAIR_TRIPS = 4;
        auto pair_of = [&](const int u, int &jb, int &j8, int &t_lo, int &t_hi) {
            jb = u < 3 ? tid >> 1 : 128;
            j8 = jb + 2 + (u < 3 ? 3 * (tid & 1) + u : tid);
            t_lo = ((k8_0 + jb) << 3) - a0;
            t_hi = t_lo + 7;                                  // the block's rows within the trip
            if (t_lo < 0) t_lo = 0;
            if (t_hi >= rows_here) t_hi = rows_here - 1;
            return (u < 3 || tid < 6) && t_lo <= t_hi && ((k8_0 + j8) << 3) <= y_near_of((k8_0 + jb) << 3);   // (the same y_near for the eight rows)
        };
        unsigned int loud = 0u;
#pragma unroll 1                                             // (unrolled: 13 registers of k_seg_walk spilled, and no faster)
        for (int u = 0; u < PAIR_TRIPS; ++u) {
            int jb, j8, t_lo, t_hi;
            if (!pair_of(u, jb, j8, t_lo, t_hi)) continue;
            // (the block's extremes as staged: rows of the neighbouring trip that share it only widen the bound)
            const int minlen = ((j8 - jb - 1) << 3) + 1;
            const double r = rs_above(minlen);
            const double ub = fmax(sh.b8x[j8] - sh.b8n[jb], 0.0) * r, lb = fmin(sh.b8n[j8] - sh.b8x[jb], 0.0) * r;
            ++evals;
            if (ub >= chi || lb <= clo) loud |= 1u << u;
        }
        __builtin_amdgcn_sched_barrier(0);                   // (both kinds' operands in flight at once cost k_seg_walk its registers)
#pragma unroll
        for (int q = 0; q < CJ_RPT; ++q) {
            const int t = tid + 256 * q;
            if (t >= rows_here) continue;
            const int ax = a0 + t;
            const double px = sh.pn[t];
            const int kb = (ax >> 3) - k8_0;
            // own block: windows from 1 bin on (the block's extremes include the entries up to the row: a superset)
            {
                const double ub = fmax(sh.b8x[kb] - px, 0.0), lb = fmin(sh.b8n[kb] - px, 0.0);
                ++evals;
                if ((ax & 7) != 7 && (ub >= chi || lb <= clo)) loud |= 16u << q;
            }
            if (((k8_0 + kb + 1) << 3) <= y_near_of(ax)) {
                const int minlen = ((k8_0 + kb + 1) << 3) - ax;              // 1..8
                const double r = sh.rsn[minlen];
                const double ub = fmax(sh.b8x[kb + 1] - px, 0.0) * r, lb = fmin(sh.b8n[kb + 1] - px, 0.0) * r;
                ++evals;
                if (ub >= chi || lb <= clo) loud |= 256u << q;
            }
        }
        while (loud) {
            const int bit = __ffs((int)loud) - 1;
            loud &= loud - 1u;
            if (bit < 4) {
                int jb, j8, t_lo, t_hi;
                pair_of(bit, jb, j8, t_lo, t_hi);
                push_rows(t_lo, t_hi, j8);
            } else {
                const int t = tid + 256 * (bit & 3);
                push_rows(t, t, ((a0 + t) >> 3) - k8_0 + (bit >> 3));
            }
        }
        wc_sync();
        CJ_CLK(22);
        // queued (row, 8-block) pairs: eight lanes per pair
        const int n_items = sh.n_items < CJ_ITEMQ ? sh.n_items : CJ_ITEMQ;
        for (int i = tid >> 3; i < n_items; i += 32) {
            const unsigned int it = sh.itemq[i];
            const int t = (int)(it >> 8);
            const int ax = a0 + t, ay = ((k8_0 + (int)(it & 255u)) << 3) + (tid & 7);
            if (ay > ax && ay <= y_near_of(ax)) see((sh.pn[ay - a0] - sh.pn[t]) * sh.rsn[ay - ax], ax, ay);
        }
        CJ_CLK(23);
    }
    wc_sync();
    CJ_CLK(MODE * 4 + 2);
    if (tid == 0) { sh.n_q2 = 0; sh.n_q1 = 0; sh.n_items = 0; }
    share_cuts();
    wc_sync();
    cuts();
    // ---- far windows: cells.  What reaches the cut is queued: 128 x 128 cells and the 32 x 32 cells of the band
    // between them and the near windows.  A part of cell_parts(L) owns a few hundred of each at most; ONE workgroup with
    // a whole CJ_MAXLEN job (k_seg_walk, k_seg_job with max_parts = 1) owns ~2 000 and ~1 150: a queue that overflows
    // sets sh.lost and the caller drops the result of this search.
    auto loud1 = [&](const int A, const int K) {
        const int minlen = ((K - A - 1) << 5) + 1;
        const double r = rs_above(minlen);
        const double ub = fmax(sh.tmx[K] - sh.tmn[A], 0.0) * r;
        const double lb = fmin(sh.tmn[K] - sh.tmx[A], 0.0) * r;
        ++evals;
        return ub >= chi || lb <= clo;
    };
    const int stride = MODE == 0 ? parts : 1, first = MODE == 0 ? part : 0;
    // 128 x 128 cells: a wave per row block, a lane per end block
    for (int A2 = first + stride * w; A2 <= A2l; A2 += 4 * stride)
        for (int K2 = A2 + 2 + lane; K2 <= K2l; K2 += 64) {
            const int minlen = ((K2 - A2 - 1) << 7) + 1;
            const double r = rs_above(minlen);
            const double ub = fmax(sh.tmx2[K2] - sh.tmn2[A2], 0.0) * r;
            const double lb = fmin(sh.tmn2[K2] - sh.tmx2[A2], 0.0) * r;
            ++evals;
            if (ub >= chi || lb <= clo) {
                const int at = atomicAdd(&sh.n_q2, 1);
                if (at < CJ_Q2) sh.q2[at] = ((unsigned int)A2 << 16) | (unsigned int)K2;
            }
        }
    // 32 x 32 cells between the near windows and the 128 x 128 cells: a thread per row block
    for (int A = A1f + first + stride * tid; A <= A1l; A += 256 * stride) {
        int k_end = (((A >> 2) + 2) << 2) - 1;
        if (k_end > K1l) k_end = K1l;
        for (int K = A + 2; K <= k_end; ++K)
            if (loud1(A, K)) {
                const int at = atomicAdd(&sh.n_q1, 1);
                if (at < CJ_Q1) sh.q1[at] = ((unsigned int)A << 16) | (unsigned int)K;
            }
    }
    wc_sync();
    CJ_CLK(MODE * 4 + 3);
    // ---- the loud cells, in bounded steps (a job that is one long aberration ends up evaluating everything, but
    // nothing overflows): sixteen 128 x 128 cells -> their 256 32 x 32 cells, a thread each; 32 x 32 cells 32 at a
    // time -> their 1024 rows, four per thread, one bound each; the rows that reach the cut are queued with their
    // end block and evaluated by half a wave each.
    const int hl = tid & 31;
    auto rows_of = [&](const unsigned int *list, const int n) {        // n <= 256 cells of `list` (LDS)
        for (int c0 = 0; c0 < n; c0 += 32) {
            wc_sync();                                          // the previous chunk's queue is drained
            if (tid == 0) sh.n_items = 0;
            share_cuts();
            wc_sync();
            cuts();
            const int nc = n - c0 < 32 ? n - c0 : 32;
            for (int i = tid; i < nc * 32; i += 256) {
                const unsigned int e = list[c0 + (i >> 5)];
                const int K = (int)(e & 0xFFFFu), ax = ((int)(e >> 16) << 5) + (i & 31);
                if (ax < rb || ax >= rhi) continue;
                const double px = P[ax];
                const int minlen = (K << 5) - ax;                    // >= 33
                const double r = rs_above(minlen);
                const double up = sh.tmx[K] - px, dn = sh.tmn[K] - px;
                const double ub = fmax(up, 0.0) * r, lb = fmin(dn, 0.0) * r;
                ++evals;
                if (ub >= chi || lb <= clo) {
                    if (MODE == 0 && (K << 5) + 31 <= rhi) {
                        // the end block lies wholly inside the job: the window that ends on its extreme is at most 31
                        // bins longer than the shortest one -- a lower bound of this row's best window, i.e. a value
                        // the cut may take
                        const double r2 = rs_below(minlen + 31);
                        if (ub >= chi && up > 0.0 && up * r2 - eps2 > chi) atomicMax(&sh.cut[0], wc::f64_ordered(up * r2));
                        if (lb <= clo && dn < 0.0 && dn * r2 + eps2 < clo) atomicMax(&sh.cut[1], wc::f64_ordered(-(dn * r2)));
                    }
                    sh.itemq[atomicAdd(&sh.n_items, 1)] = ((unsigned int)ax << 12) | (unsigned int)K;   // (at most 1024)
                }
            }
            wc_sync();
            const int n_items = sh.n_items;
            for (int i = tid >> 5; i < n_items; i += 8) {
                const unsigned int it = sh.itemq[i];
                const int ax = (int)(it >> 12), ay = ((int)(it & 4095u) << 5) + hl;
                if (ay <= rhi) see((P[ay] - P[ax]) * rs[ay - ax], ax, ay);
            }
        }
    };
    {
        // (the second pass walks the whole job in one workgroup: should a queue overflow there -- only under massive
        // ties -- the record is marked overflowing, which sends the job to the exact scan)
        if (MODE == 2 && tid == 0 && (sh.n_q2 > CJ_Q2 || sh.n_q1 > CJ_Q1)) sh.n_rec[0] = CJ_REC + 1;
        if (MODE == 0 && tid == 0 && (sh.n_q2 > CJ_Q2 || sh.n_q1 > CJ_Q1)) sh.lost = 1;     // (read after the next barrier)
        const int n2 = sh.n_q2 < CJ_Q2 ? sh.n_q2 : CJ_Q2, n1 = sh.n_q1 < CJ_Q1 ? sh.n_q1 : CJ_Q1;
        for (int c2 = 0; c2 < n2; c2 += 16) {
            wc_sync();
            if (tid == 0) sh.n_l1 = 0;
            wc_sync();
            cuts();
            if (c2 + (tid >> 4) < n2) {
                const unsigned int e = sh.q2[c2 + (tid >> 4)];
                const int A = ((int)(e >> 16) << 2) + ((tid >> 2) & 3), K = ((int)(e & 0xFFFFu) << 2) + (tid & 3);
                if (A >= A1f && A <= A1l && K <= K1l && loud1(A, K))
                    sh.l1[atomicAdd(&sh.n_l1, 1)] = ((unsigned int)A << 16) | (unsigned int)K;
            }
            wc_sync();
            rows_of(sh.l1, sh.n_l1);
        }
        for (int c1 = 0; c1 < n1; c1 += 256) rows_of(sh.q1 + c1, n1 - c1 < 256 ? n1 - c1 : 256);
    }
#ifdef WC_CELL_CLOCKS
    wc_sync();
    CJ_CLK(MODE * 4 + 4);
    if (tid == 0) { sh.clk[MODE * 4 + 5] += (unsigned long long)sh.n_q2; sh.clk[MODE * 4 + 6] += (unsigned long long)sh.n_q1; }
#endif
}

// What both kernels of a job need first: geometry, both table levels of the job in LDS (block 0 = the origin's) with
// exact edge blocks, the first 64 table factors, empty records -- and, with stage_r0 >= 0, the near sweep's trip that
// starts at row stage_r0 (cell_search's `prestaged`).  Every global load of all that is requested before anything is
// waited for: ONE memory round trip (tables, then edge entries, then the first trip's rows were three).
// Ends with ONE barrier; the edge blocks are corrected after it, for readers that are at least one more barrier away
// (the cell sweeps; cell_seed only reads blocks that lie wholly inside the job, which need no correction).
__device__ inline CellGeom cell_setup(CellShared &sh, const Job job, const Region rg, const int region,
                                      const double *__restrict__ prefix, const double *__restrict__ rs,
                                      const double *__restrict__ tmin, const double *__restrict__ tmax,
                                      const double *__restrict__ tmin2, const double *__restrict__ tmax2, const int tid,
                                      const int stage_r0 = -1) {
    const long long base = rg.off + region + job.lo;         // absolute index of the job's first prefix entry
    const long long org = (base >> 7) << 7;
    CellGeom g;
    g.P = prefix + org;
    g.rb = (int)(base - org);
    g.rhi = g.rb + (job.hi - job.lo);
    g.job_lo = job.lo;
    const int n1 = (g.rhi >> 5) + 1, n2 = (g.rhi >> 7) + 1;
    const long long k1 = org >> 5, k2 = org >> 7;
    static_assert(CJ_T1 <= 512 && CJ_T2 <= 256, "two and one table entries per thread");
    // ---- loads
    double t1x[2], t1n[2];
#pragma unroll
    for (int u = 0; u < 2; ++u) {
        const int i = tid + 256 * u;
        t1x[u] = i < n1 ? tmax[k1 + i] : 0.0;
        t1n[u] = i < n1 ? tmin[k1 + i] : 0.0;
    }
    const double t2x = tid < n2 ? tmax2[k2 + tid] : 0.0, t2n = tid < n2 ? tmin2[k2 + tid] : 0.0;
    const double rs_mine = tid < 64 ? rs[tid] : 0.0;
    // The tables are aligned to the concatenated array: the job's first and last block of either level also cover
    // entries of its neighbours (another region's prefix sums: a jump).  Their extremes over the job's own
    // entries, a wave each -- otherwise every cell on the job's edges reaches the cut.
    const int w = tid >> 6, lane = tid & 63;
    const int lvl = w >> 1, last = w & 1;                    // waves 0 / 1: 32-blocks, 2 / 3: 128-blocks; first / last block
    const int shift = lvl ? 7 : 5;
    const int blk = last ? g.rhi >> shift : g.rb >> shift;
    double ev[2];
#pragma unroll
    for (int u = 0; u < 2; ++u) {
        const int e = lane + 64 * u, a = (blk << shift) + e;
        ev[u] = (e < (1 << shift) && a >= g.rb && a <= g.rhi) ? g.P[a] : NAN;
    }
    TripRegs tr;
    if (stage_r0 >= 0) trip_load(tr, g, g.rb + stage_r0, tid);
    // ---- LDS
#pragma unroll
    for (int u = 0; u < 2; ++u) {
        const int i = tid + 256 * u;
        if (i < n1) { sh.tmx[i] = t1x[u]; sh.tmn[i] = t1n[u]; }
    }
    if (tid < n2) { sh.tmx2[tid] = t2x; sh.tmn2[tid] = t2n; }
    if (tid < 64) sh.rsn[tid] = rs_mine;
    if (tid == 0) { sh.n_rec[0] = 0; sh.n_rec[1] = 0; sh.lost = 0; }
    if (stage_r0 >= 0) trip_store(sh, tr, g, g.rb + stage_r0, tid);
    double mx = -INFINITY, mn = INFINITY;
#pragma unroll
    for (int u = 0; u < 2; ++u)
        if (ev[u] == ev[u]) { mx = fmax(mx, ev[u]); mn = fmin(mn, ev[u]); }
    for (int o = 32; o > 0; o >>= 1) { mx = fmax(mx, __shfl_xor(mx, o)); mn = fmin(mn, __shfl_xor(mn, o)); }
    wc_sync();
    // (a block that lies wholly inside the job keeps the table's entry -- the same numbers, and cell_seed may be reading it)
    const bool partial = last ? ((g.rhi + 1) & ((1 << shift) - 1)) != 0 : (g.rb & ((1 << shift) - 1)) != 0;
    if (lane == 0 && partial) {
        if (lvl) { sh.tmx2[blk] = mx; sh.tmn2[blk] = mn; }
        else { sh.tmx[blk] = mx; sh.tmn[blk] = mn; }
    }
    return g;
}

// The starting cuts (k_seg_seed's idea): the window from the minimum of block a to the maximum of block b, both wholly
// inside the job, is worth at least (max_b - min_a) * rs[(b - a + 1) blocks] -- 32-blocks up to eight apart (short
// aberrations), 128-blocks at any distance (long ones).  Needs a barrier before the cuts are read.
__device__ inline void cell_seed(CellShared &sh, const CellGeom g, const double T, const int tid) {
    double hi = T, lo = -T, d0 = -INFINITY, d1 = INFINITY;
    const int b_first = (g.rb + QB - 1) >> 5, b_last = ((g.rhi + 1) >> 5) - 1;
    for (int i = tid; i < (b_last - b_first) * 8; i += 256) {
        const int a = b_first + (i >> 3), b = a + 1 + (i & 7);
        if (b <= b_last) {
            const double r = rs_below((b - a + 1) * QB);
            const double up = sh.tmx[b] - sh.tmn[a], dn = sh.tmn[b] - sh.tmx[a];
            if (up > 0.0) hi = fmax(hi, up * r);
            if (dn < 0.0) lo = fmin(lo, dn * r);
        }
    }
    const int c_first = (g.rb + QB2 - 1) >> 7, c_last = ((g.rhi + 1) >> 7) - 1;
    for (int a = c_first + (tid >> 6); a < c_last; a += 4) {
        const double mn_a = sh.tmn2[a], mx_a = sh.tmx2[a];
        for (int b = a + 1 + (tid & 63); b <= c_last; b += 64) {
            const double r = rs_below((b - a + 1) * QB2);
            const double up = sh.tmx2[b] - mn_a, dn = sh.tmn2[b] - mx_a;
            if (up > 0.0) hi = fmax(hi, up * r);
            if (dn < 0.0) lo = fmin(lo, dn * r);
        }
    }
    block_minmax4(hi, lo, d0, d1, tid);
    if (tid == 0) { sh.cut[0] = wc::f64_ordered(hi); sh.cut[1] = wc::f64_ordered(-lo); }
}

// The search of one job by parts(L) workgroups (k_seg_seed + k_seg_bound of the bound-driven rounds).  A part leaves
// its extremes and its record of near-extreme windows in the job's state -- relaxed atomics and plain stores, no
// fence: k_seg_merge reads them after the kernel boundary (an agent-scope fence per workgroup writes back and
// invalidates the XCD's L2: measured 1.6 ms instead of 0.15 for round 1 of a 125 x 50 kb batch).  The cuts the parts
// share while they run are hints only.  state[] is zeroed by the host before the launch.
__global__ __launch_bounds__(256) void k_seg_job(const Job *__restrict__ jobs, int n_jobs,
                                                 const Region *__restrict__ regions,
                                                 const double *__restrict__ prefix, const double *__restrict__ rs,
                                                 const double *__restrict__ reg_abs, const int *__restrict__ reg_flag,
                                                 double thr, const double *__restrict__ tmin,
                                                 const double *__restrict__ tmax, const double *__restrict__ tmin2,
                                                 const double *__restrict__ tmax2, CellJobState *__restrict__ state,
                                                 CellRec *__restrict__ grec, unsigned long long *__restrict__ work) {
    __shared__ CellShared sh;
    const int j = blockIdx.y, part = blockIdx.x, tid = threadIdx.x;
    if (j >= n_jobs) return;
    const Job job = jobs[j];
    const int L = job.hi - job.lo;
    if (L <= 0) return;
    const int parts = cell_parts(L) < (int)gridDim.x ? cell_parts(L) : (int)gridDim.x;
    if (part >= parts) return;
    if (!reg_flag[job.region]) return;            // non-finite region: k_seg_merge hands it to the exact scan
    const Region rg = regions[job.region];
    const double eps = window_eps(rg.n, reg_abs[job.region]);
    const double eps2 = 2.0 * eps, T = thr - eps;
    CellJobState *js = state + j;
#ifdef WC_CELL_CLOCKS
    if (tid < 32) sh.clk[tid] = 0ull;
    if (tid == 0) sh.t_prev = clock64();
    const unsigned long long t_begin = clock64();
    wc_sync();
#endif
    const bool pre = part * CJ_TRIP < L;          // this part's first near-sweep trip rides on the set-up's loads
    const CellGeom g = cell_setup(sh, job, rg, job.region, prefix, rs, tmin, tmax, tmin2, tmax2, tid, pre ? part * CJ_TRIP : -1);
    cell_seed(sh, g, T, tid);
    wc_sync();
    CJ_CLK(1);
    double vmax = -INFINITY, vmin = INFINITY, d2 = -INFINITY, d3 = INFINITY;
    int wins = 0, evals = 0;
    cell_search<0>(sh, g, rs, eps2, INFINITY, -INFINITY, js->cut, part, parts, vmax, vmin, wins, evals, tid, pre);
    block_minmax4(vmax, vmin, d2, d3, tid);
    // this part's extremes and record into the job's state
    CellRec *jrec = grec + (int64_t)j * 2 * CJ_GREC;
    if (tid == 0) {
        atomicMax(&js->ext[0], wc::f64_ordered(vmax));
        atomicMax(&js->ext[1], wc::f64_ordered(-vmin));
        if (sh.n_rec[0] > CJ_REC || sh.n_rec[1] > CJ_REC) atomicOr(&js->overflow, 1);
        if (sh.lost) atomicOr(&js->overflow, 2);             // dropped cells: k_seg_merge sends the job to the exact scan
    }
    for (int side = 0; side < 2; ++side) {
        const int n = sh.n_rec[side] < CJ_REC ? sh.n_rec[side] : CJ_REC;
        // (a record that is no longer near this part's cut cannot be near the job's)
        const double cut = side == 0 ? wc::f64_from_ordered(sh.cut[0]) - eps2 : -wc::f64_from_ordered(sh.cut[1]) + eps2;
        for (int i = tid; i < n; i += 256) {
            const CellRec r = sh.rec[side][i];
            if (side == 0 ? r.v >= cut : r.v <= cut) {
                const int at = atomicAdd(&js->n_rec[side], 1);
                if (at < CJ_GREC) jrec[side * CJ_GREC + at] = r;
            }
        }
    }
#ifdef WC_CELL_CLOCKS
    wc_sync();
    CJ_CLK(15);
    if (tid == 0) {
        sh.clk[16] = 1ull;
        atomicMax(&g_dbg[48], clock64() - t_begin);          // the longest-lived workgroup
    }
    wc_sync();
    if (tid < 32) atomicAdd(&g_dbg[tid], sh.clk[tid]);
#endif
    if (work) {
        for (int o = 32; o > 0; o >>= 1) { evals += __shfl_xor(evals, o); wins += __shfl_xor(wins, o); }
        const int slot = (int)((blockIdx.y * 7u + blockIdx.x * 3u + (unsigned)(tid >> 6)) & 63u);
        if ((tid & 63) == 0) {
            atomicAdd(work + 2 * slot, (unsigned long long)wins);
            atomicAdd(work + 2 * slot + 1, (unsigned long long)evals);
        }
    }
}

// After the search: classification (k_seg_classify's test on the job's extremes) and, for a job that may hold a
// call, the candidate list from the parts' records (k_seg_bcollect's product), in the form k_seg_decide reads.  A job
// whose records overflowed (ties) is walked once more with the final cuts (MODE 2).  One workgroup per job, most
// leave at once.  counters[2] / [3] and next_count are zeroed by the host before the launch.
__global__ __launch_bounds__(256) void k_seg_merge(const Job *__restrict__ jobs, int n_jobs,
                                                   const Region *__restrict__ regions,
                                                   const double *__restrict__ prefix, const double *__restrict__ rs,
                                                   const double *__restrict__ reg_abs, const int *__restrict__ reg_flag,
                                                   double thr, const double *__restrict__ tmin,
                                                   const double *__restrict__ tmax, const double *__restrict__ tmin2,
                                                   const double *__restrict__ tmax2,
                                                   const CellJobState *__restrict__ state,
                                                   const CellRec *__restrict__ grec, int *__restrict__ hot,
                                                   int *__restrict__ brute, int *__restrict__ counters,
                                                   int2 *__restrict__ cand, int *__restrict__ cand_cnt,
                                                   unsigned long long *__restrict__ work) {
    __shared__ CellShared sh;
    const int j = blockIdx.x, tid = threadIdx.x;
    if (j >= n_jobs) return;
    const Job job = jobs[j];
    const int L = job.hi - job.lo;
    if (L <= 0) return;
    if (!reg_flag[job.region]) {                  // non-finite region: the exact scan
        if (tid == 0) brute[atomicAdd(&counters[3], 1)] = j;
        return;
    }
    const CellJobState js = state[j];
    if (js.overflow & 2) {                        // a part dropped loud cells: its extremes prove nothing -- the exact scan
        if (tid == 0) brute[atomicAdd(&counters[3], 1)] = j;
        return;
    }
    const double vmax = wc::f64_from_ordered(js.ext[0]), vmin = -wc::f64_from_ordered(js.ext[1]);
    const Region rg = regions[job.region];
    const double eps = window_eps(rg.n, reg_abs[job.region]);
    if (fmax(fabs(vmax), fabs(vmin)) + eps < thr) return;                    // no call in this job
    // the windows within 2 eps of the extremes, only for a side that can hold a call
    const double eps2 = 2.0 * eps;
    const double hi_cut = !(vmax + eps < thr) ? vmax - eps2 : INFINITY;
    const double lo_cut = !(-vmin + eps < thr) ? vmin + eps2 : -INFINITY;
    const CellRec *jrec = grec + (int64_t)j * 2 * CJ_GREC;
    int wins = 0, evals = 0;
    if (tid == 0) { sh.n_rec[0] = 0; sh.n_rec[1] = 0; }
    wc_sync();
    if (js.overflow || js.n_rec[0] > CJ_GREC || js.n_rec[1] > CJ_GREC) {
        // more near-extreme windows than the records hold (ties): the whole job once more with the final cuts
        const CellGeom g = cell_setup(sh, job, rg, job.region, prefix, rs, tmin, tmax, tmin2, tmax2, tid);
        double e0 = -INFINITY, e1 = INFINITY;
        cell_search<2>(sh, g, rs, eps2, hi_cut, lo_cut, nullptr, 0, 1, e0, e1, wins, evals, tid);
    } else {
        for (int i = tid; i < js.n_rec[0]; i += 256) {
            const CellRec r = jrec[i];
            if (r.v >= hi_cut) { const int at = atomicAdd(&sh.n_rec[0], 1); if (at < CJ_REC) sh.rec[0][at] = r; }
        }
        for (int i = tid; i < js.n_rec[1]; i += 256) {
            const CellRec r = jrec[CJ_GREC + i];
            if (r.v <= lo_cut) { const int at = atomicAdd(&sh.n_rec[1], 1); if (at < CJ_REC) sh.rec[1][at] = r; }
        }
    }
    wc_sync();
    if (tid == 0) sh.slot = atomicAdd(&counters[2], 1);
    wc_sync();
    const int h = sh.slot;
    if (tid < 2) {
        const int side = tid, n = sh.n_rec[side];
        // (a list beyond CAND_CAP makes k_seg_decide send the job to the exact scan)
        for (int i = 0; i < n && i < CAND_CAP; ++i) cand[((int64_t)2 * h + side) * CAND_CAP + i] = make_int2(sh.rec[side][i].x, sh.rec[side][i].y);
        cand_cnt[2 * h + side] = n > CJ_REC ? CAND_CAP + 1 : n;
    }
    if (tid == 0) hot[h] = j;
    if (work && (wins | evals)) {
        for (int o = 32; o > 0; o >>= 1) { evals += __shfl_xor(evals, o); wins += __shfl_xor(wins, o); }
        if ((tid & 63) == 0) {
            atomicAdd(work + 2 * (tid >> 6), (unsigned long long)wins);
            atomicAdd(work + 2 * (tid >> 6) + 1, (unsigned long long)evals);
        }
    }
}

// v_max_f64 / v_min_f64 on finite values, without the canonicalisation fmax() / fmin() imply
__device__ inline double raw_max(double a, double b) {
    double r;
    asm("v_max_f64 %0, %1, %2" : "=v"(r) : "v"(a), "v"(b));
    return r;
}
__device__ inline double raw_min(double a, double b) {
    double r;
    asm("v_min_f64 %0, %1, %2" : "=v"(r) : "v"(a), "v"(b));
    return r;
}

// Extreme Stouffer values (prefix-sum estimates) of one block of window rows of a job.  Only
// the two VALUES leave this kernel: the positions of near-extreme windows are re-derived by
// k_seg_collect and ranked by their exact values, so nothing here tracks where the extreme
// sits (4 float64 operations per window: subtract, scale, max, min).
//   PLDS   the job's prefix slice is staged in LDS (one coalesced read per workgroup)
//   MASKED -mineffectsize: windows whose validity bit is clear count as 0
// A wave takes four consecutive window lengths per trip: their 1/sqrt(len) factors are one
// scalar load, the four prefix values of a lane are adjacent.
// Extremes (prefix-sum estimates) of the windows that start in one side of one block of rows of a
// job: the wave-reduced maximum and minimum over the window lengths this wave owns (1 + 4 w .. 4 + 4 w,
// then every 4 NW-th group).  P: the job's prefix slice (P[0..L]); job_lo only matters when MASKED.
template <bool MASKED, int NW>
__device__ inline void search_side(const double *P, int L, int half, int chunk, int side, const double *__restrict__ rs,
                                   int lane, int w, int job_lo, const WindowMask &wm, double &smax_out,
                                   double &smin_out, int &n_windows) {
    struct { int lo; } job{job_lo};
        int xr = chunk * ROWS_HALF + lane;
        bool live;
        if (side == 0) {
            live = xr < half;
        } else {
            xr = L - 1 - xr;
            live = xr >= half;
        }
        // longest window among the rows of this block (lane 0 has the smallest / largest start)
        const int xr_min = side == 0 ? chunk * ROWS_HALF : L - 1 - (chunk * ROWS_HALF + 63);
        const int max_len = L - (xr_min < 0 ? 0 : xr_min);
        const int xl = live ? xr : 0;
        const double px = P[xl];
        const int room = live ? L - xl : 0;            // windows [xl, xl + len - 1] with len <= room
        n_windows += room;
        double smax = -INFINITY, smin = INFINITY;      // this side, this wave
        int base = 1 + 4 * w;
        if (!MASKED) {
            // Lengths every live lane of the wave still has room for take no per-window test at
            // all: subtract, scale, max, min (the two raw instructions: fmax() / fmin() would each
            // add a canonicalising v_max_f64, and the values are finite here -- non-finite regions
            // never reach this kernel).  Lanes without a row read row 0 and are dropped afterwards.
            const int last_live = side == 0 ? (chunk * ROWS_HALF + 63 < half - 1 ? chunk * ROWS_HALF + 63 : half - 1)
                                            : L - 1 - chunk * ROWS_HALF;
            const int min_room = side == 0 ? L - last_live : L - last_live;   // smallest room among the live lanes
            const double *pb = P + xl;
            // two of the wave's four-length groups per trip (one wait for eight loads); the
            // lengths stay dealt to the waves as k_seg_collect expects: group g goes to wave g % NW
            for (; base + 4 * NW + 3 <= min_room; base += 8 * NW) {
                double r[8], pv[8];
#pragma unroll
                for (int u = 0; u < 4; ++u) {
                    r[u] = rs[base + u];
                    pv[u] = pb[base + u];
                    r[4 + u] = rs[base + 4 * NW + u];
                    pv[4 + u] = pb[base + 4 * NW + u];
                }
#pragma unroll
                for (int u = 0; u < 8; ++u) {
                    const double v = (pv[u] - px) * r[u];
                    smax = raw_max(smax, v);
                    smin = raw_min(smin, v);
                }
            }
            for (; base + 3 <= min_room; base += 4 * NW) {
                const double r0 = rs[base], r1 = rs[base + 1], r2 = rs[base + 2], r3 = rs[base + 3];
                const double p0 = pb[base], p1 = pb[base + 1], p2 = pb[base + 2], p3 = pb[base + 3];
                const double v0 = (p0 - px) * r0, v1 = (p1 - px) * r1, v2 = (p2 - px) * r2, v3 = (p3 - px) * r3;
                smax = raw_max(smax, v0); smin = raw_min(smin, v0);
                smax = raw_max(smax, v1); smin = raw_min(smin, v1);
                smax = raw_max(smax, v2); smin = raw_min(smin, v2);
                smax = raw_max(smax, v3); smin = raw_min(smin, v3);
            }
            if (!live) { smax = -INFINITY; smin = INFINITY; }
        }
        for (; base <= max_len; base += 4 * NW) {
            double r[4], pv[4];
#pragma unroll
            for (int u = 0; u < 4; ++u) {
                r[u] = rs[base + u];                                   // wave-uniform (table padded by 4)
                const int len = base + u;
                pv[u] = P[xl + (len < room ? len : room)];             // always inside the slice
            }
#pragma unroll
            for (int u = 0; u < 4; ++u) {
                const int len = base + u;
                if (len <= room) {
                    double v = (pv[u] - px) * r[u];
                    if (MASKED && !wm.valid(job.lo + xl, job.lo + xl + len - 1)) v = 0.0;
                    smax = fmax(smax, v);
                    smin = fmin(smin, v);
                }
            }
        }
        for (int o = 32; o > 0; o >>= 1) {
            smax = fmax(smax, __shfl_xor(smax, o));
            smin = fmin(smin, __shfl_xor(smin, o));
        }
        smax_out = smax;
        smin_out = smin;
}

template <bool MASKED, bool PLDS, int NW>   // NW waves per block: 4, or 16 for rounds with few blocks (latency bound)
__global__ __launch_bounds__(64 * NW) void k_seg_search(const Job *__restrict__ jobs, int n_jobs,
                                                    const Region *__restrict__ regions,
                                                    const double *__restrict__ prefix, const double *__restrict__ rs,
                                                    const int *__restrict__ reg_flag, int max_chunks,
                                                    const unsigned int *__restrict__ bits,
                                                    const long long *__restrict__ bit_off,
                                                    Extreme *__restrict__ partial, int *__restrict__ counters,
                                                    int certified, double2 *__restrict__ sub,
                                                    unsigned long long *__restrict__ work,
                                                    const int *__restrict__ n_jobs_dev, int *__restrict__ next_count) {
    extern __shared__ double pl[];
    __shared__ double red_max[NW], red_min[NW];
    __shared__ double side_max[2][NW], side_min[2][NW];
    const int j = blockIdx.y, chunk = blockIdx.x, tid = threadIdx.x;
    // first kernel of a round: next-jobs / hot / brute counts start at zero (classify runs after)
    if (j == 0 && chunk == 0 && tid == 0) { *next_count = 0; counters[2] = 0; counters[3] = 0; }
    if (n_jobs_dev) n_jobs = *n_jobs_dev < n_jobs ? *n_jobs_dev : n_jobs;
    if (j >= n_jobs) return;
    const Job job = jobs[j];
    const int L = job.hi - job.lo, half = (L + 1) / 2;
    if (L <= 0 || chunk * ROWS_HALF >= half) return;
    if (certified && !job.pad) return;  // certified quiet by k_seg_quiet (most workgroups of a batch: before the next load)
    if (!reg_flag[job.region]) return;  // non-finite region: exact brute-force path
    const double *Pg = prefix + regions[job.region].off + job.region + job.lo;   // Pg[0..L]
    const double *P = Pg;
    if (PLDS) {
        for (int i = tid; i <= L; i += 64 * NW) pl[i] = Pg[i];
        wc_sync();
        P = pl;
    }
    const WindowMask wm{bits, MASKED ? bit_off[job.region] : 0, regions[job.region].n};
    const int lane = tid & 63;
    const int w = __builtin_amdgcn_readfirstlane(tid >> 6);
    double bmax = -INFINITY, bmin = INFINITY;      // over both sides
    int n_windows = 0;                             // profiling only: windows of this lane's rows
    double2 *sub_blk = sub + ((int64_t)j * max_chunks + chunk) * 8;   // [side][search wave]: {max, min}
    for (int side = 0; side < 2; ++side) {
        double smax, smin;
        search_side<MASKED, NW>(P, L, half, chunk, side, rs, lane, w, job.lo, wm, smax, smin, n_windows);
        // the collect pass re-scans a (side, wave) slice only if its own extremes reach the cut
        if (NW == 4 && lane == 0) sub_blk[side * 4 + w] = make_double2(smax, smin);
        if (NW != 4 && lane == 0) { side_max[side][w] = smax; side_min[side][w] = smin; }
        bmax = fmax(bmax, smax);
        bmin = fmin(bmin, smin);
    }
    if (lane == 0) { red_max[w] = bmax; red_min[w] = bmin; }
    if (work && w == 0) {       // the waves split the window lengths of the same rows: count them once
        for (int o = 32; o > 0; o >>= 1) n_windows += __shfl_xor(n_windows, o);
        if (lane == 0) atomicAdd(work, (unsigned long long)n_windows);
    }
    wc_sync();
    if (tid == 0) {
        Extreme e;
        e.maxv = red_max[0];
        e.minv = red_min[0];
        for (int q = 1; q < NW; ++q) { e.maxv = fmax(e.maxv, red_max[q]); e.minv = fmin(e.minv, red_min[q]); }
        e.max_x = e.max_y = e.min_x = e.min_y = -1;
        partial[(int64_t)j * max_chunks + chunk] = e;
    }
    if (NW != 4 && tid < 8) {
        // sixteen search waves: the collect pass's wave w16 scans the lengths with (len - 1) % 16 == w16,
        // which the search waves (w16 >> 2) + 4 m covered (groups of four lengths dealt round robin):
        // slot [side][g] holds the extremes over the search waves w with w % 4 == g
        const int side = tid >> 2, g = tid & 3;
        double mx = -INFINITY, mn = INFINITY;
        for (int q = g; q < NW; q += 4) { mx = fmax(mx, side_max[side][q]); mn = fmin(mn, side_min[side][q]); }
        sub_blk[side * 4 + g] = make_double2(mx, mn);
    }
}

// One wave per job: merge chunk results, then classify: quiet (no call possible),
// hot (exact evaluation of the near-extreme windows) or brute (non-finite region).
__global__ __launch_bounds__(64) void k_seg_classify(const Job *__restrict__ jobs, int n_jobs,
                                                     const Region *__restrict__ regions,
                                                     const double *__restrict__ reg_abs, const int *__restrict__ reg_flag,
                                                     const Extreme *__restrict__ partial, int max_chunks, double thr,
                                                     Extreme *__restrict__ job_res, int *__restrict__ hot,
                                                     int *__restrict__ brute, int *__restrict__ counters,
                                                     int *__restrict__ cand_cnt, int certified,
                                                     const int *__restrict__ n_jobs_dev) {
    const int j = blockIdx.x, lane = threadIdx.x;
    if (n_jobs_dev) {
        // device-side round loop: more jobs than this round's grid holds -> the caller re-runs the
        // segmentation with the host-driven loop
        if (j == 0 && lane == 0 && *n_jobs_dev > n_jobs) counters[6] = 1;
        n_jobs = *n_jobs_dev < n_jobs ? *n_jobs_dev : n_jobs;
    }
    if (j >= n_jobs) return;
    const Job job = jobs[j];
    const int L = job.hi - job.lo;
    if (L <= 0) return;
    if (!reg_flag[job.region]) {
        if (lane == 0) brute[atomicAdd(&counters[3], 1)] = j;
        return;
    }
    if (certified && !job.pad) return;  // certified quiet: no call, no children
    const int nch = (int)(((L + 1) / 2 + ROWS_HALF - 1) / ROWS_HALF);
    Extreme e;
    e.maxv = -INFINITY; e.minv = INFINITY; e.max_x = e.max_y = e.min_x = e.min_y = -1;
    for (int ch = lane; ch < nch; ch += 64) {
        Extreme p = partial[(int64_t)j * max_chunks + ch];
        if (p.maxv > e.maxv) { e.maxv = p.maxv; e.max_x = p.max_x; e.max_y = p.max_y; }
        if (p.minv < e.minv) { e.minv = p.minv; e.min_x = p.min_x; e.min_y = p.min_y; }
    }
    for (int o = 32; o > 0; o >>= 1) {
        double ov = __shfl_xor(e.maxv, o);
        int ox = __shfl_xor(e.max_x, o), oy = __shfl_xor(e.max_y, o);
        if (ov > e.maxv) { e.maxv = ov; e.max_x = ox; e.max_y = oy; }
        double uv = __shfl_xor(e.minv, o);
        int ux = __shfl_xor(e.min_x, o), uy = __shfl_xor(e.min_y, o);
        if (uv < e.minv) { e.minv = uv; e.min_x = ux; e.min_y = uy; }
    }
    if (lane == 0) {
        job_res[j] = e;
        double eps = window_eps(regions[job.region].n, reg_abs[job.region]);
        double big = fmax(fabs(e.maxv), fabs(e.minv));
        if (!(big + eps < thr)) {
            const int h = atomicAdd(&counters[2], 1);
            hot[h] = j;
            cand_cnt[2 * h] = 0;          // candidate lists of hot job h start empty
            cand_cnt[2 * h + 1] = 0;
        }
    }
}

// Second scan of the hot jobs: every window within 2 eps of the approximate
// maximum (minimum) could be numpy's argmax (argmin); list them for exact scoring.
__global__ __launch_bounds__(1024) void k_seg_collect(const Job *__restrict__ jobs, const int *__restrict__ hot,
                                                     const int *__restrict__ counters,
                                                     const Region *__restrict__ regions,
                                                     const double *__restrict__ prefix, const double *__restrict__ rs,
                                                     const double *__restrict__ reg_abs,
                                                     const Extreme *__restrict__ job_res,
                                                     const Extreme *__restrict__ partial, int max_chunks,
                                                     const double2 *__restrict__ sub,
                                                     const unsigned int *__restrict__ bits,
                                                     const long long *__restrict__ bit_off, int2 *__restrict__ cand,
                                                     int *__restrict__ cand_cnt) {
    const int h = blockIdx.y, chunk = blockIdx.x, tid = threadIdx.x;
    if (h >= counters[2]) return;
    const int j = hot[h];
    const Job job = jobs[j];
    ScanCtx c;
    c.lo = job.lo; c.hi = job.hi; c.L = job.hi - job.lo; c.half = (c.L + 1) / 2; c.chunk = chunk;
    if (chunk * ROWS_HALF >= c.half) return;
    c.P = prefix + regions[job.region].off + job.region;
    c.wm = WindowMask{bits, bits ? bit_off[job.region] : 0, regions[job.region].n};
    const double eps2 = 2.0 * window_eps(regions[job.region].n, reg_abs[job.region]);
    const Extreme e = job_res[j];
    const double hi_cut = e.maxv - eps2, lo_cut = e.minv + eps2;
    {   // the search left this block's own extremes behind: most blocks hold no near-extreme window
        const Extreme p = partial[(int64_t)j * max_chunks + chunk];
        if (p.maxv < hi_cut && p.minv > lo_cut) return;
    }
    // ... and within a block, most (side, search wave) slices: this kernel's wave w16 scans the
    // window lengths 1 + w16 (mod 16), which the search's wave w16 / 4 covered
    unsigned int skip = 0u;
    {
        const double2 *sb = sub + ((int64_t)j * max_chunks + chunk) * 8;
        for (int side = 0; side < 2; ++side)
            for (int w16 = 0; w16 < 16; ++w16) {
                const double2 e2 = sb[side * 4 + (w16 >> 2)];
                if (e2.x < hi_cut && e2.y > lo_cut) skip |= 1u << (side * 16 + w16);
            }
    }
    scan_chunk(c, rs, tid, [&](double v, int x, int y) {
        if (v >= hi_cut) {
            int at = atomicAdd(&cand_cnt[2 * h], 1);
            if (at < CAND_CAP) cand[((int64_t)2 * h) * CAND_CAP + at] = make_int2(x, y);
        }
        if (v <= lo_cut) {
            int at = atomicAdd(&cand_cnt[2 * h + 1], 1);
            if (at < CAND_CAP) cand[((int64_t)2 * h + 1) * CAND_CAP + at] = make_int2(x, y);
        }
    }, skip);
}

// The decision of TriArr.segmentTri (triarray.py:59-84) given the exact extremes.
__device__ inline void decide_emit(const Job &job, double maxv, int mx, int my, double minv, int nx, int ny,
                                   double thr, int min_search, Seg *segs, int seg_cap, Job *next, int job_cap,
                                   int *counters, int *next_count) {
    double champ = maxv;
    int cx = mx, cy = my;
    if (fabs(minv) > champ) { champ = minv; cx = nx; cy = ny; }
    if (fabs(champ) < thr) return;
    int at = atomicAdd(&counters[4], 1);
    if (at < seg_cap) {
        Seg s;
        s.val = champ; s.region = job.region; s.x = cx; s.y = cy; s.pad = 0;
        segs[at] = s;
    }
    const int xr = cx - job.lo, yr = cy - job.lo, edge = job.hi - job.lo;
    if (xr > min_search) {
        int p = atomicAdd(next_count, 1);
        if (p < job_cap) { Job n; n.region = job.region; n.lo = job.lo; n.hi = cx; n.pad = 0; next[p] = n; }
    }
    if (yr + 1 < edge - min_search) {
        int p = atomicAdd(next_count, 1);
        if (p < job_cap) { Job n; n.region = job.region; n.lo = cy + 1; n.hi = job.hi; n.pad = 0; next[p] = n; }
    }
}

// numpy argmax/argmin order: the first NaN wins, otherwise the extreme value with
// the lowest linear (row-major by x, then y) position.
__device__ inline bool better_max(double v, int x, int y, double bv, int bx, int by) {
    if (bx < 0) return true;
    bool vn = v != v, bn = bv != bv;
    if (bn) return vn && (x < bx || (x == bx && y < by));
    if (vn) return true;
    if (v > bv) return true;
    return v == bv && (x < bx || (x == bx && y < by));
}
__device__ inline bool better_min(double v, int x, int y, double bv, int bx, int by) {
    if (bx < 0) return true;
    bool vn = v != v, bn = bv != bv;
    if (bn) return vn && (x < bx || (x == bx && y < by));
    if (vn) return true;
    if (v < bv) return true;
    return v == bv && (x < bx || (x == bx && y < by));
}

struct BestPair {
    double maxv, minv;
    int mx, my, nx, ny;
};

__device__ inline void block_best(BestPair &b, int tid) {
    __shared__ double smax[256], smin[256];
    __shared__ int sx[256], sy[256], tx[256], ty[256];
    smax[tid] = b.maxv; smin[tid] = b.minv; sx[tid] = b.mx; sy[tid] = b.my; tx[tid] = b.nx; ty[tid] = b.ny;
    wc_sync();
    for (int o = 128; o > 0; o >>= 1) {
        if (tid < o) {
            if (sx[tid + o] >= 0 && better_max(smax[tid + o], sx[tid + o], sy[tid + o], smax[tid], sx[tid], sy[tid])) {
                smax[tid] = smax[tid + o]; sx[tid] = sx[tid + o]; sy[tid] = sy[tid + o];
            }
            if (tx[tid + o] >= 0 && better_min(smin[tid + o], tx[tid + o], ty[tid + o], smin[tid], tx[tid], ty[tid])) {
                smin[tid] = smin[tid + o]; tx[tid] = tx[tid + o]; ty[tid] = ty[tid + o];
            }
        }
        wc_sync();
    }
    b.maxv = smax[0]; b.minv = smin[0]; b.mx = sx[0]; b.my = sy[0]; b.nx = tx[0]; b.ny = ty[0];
    wc_sync();
}

__global__ __launch_bounds__(256) void k_seg_decide(const Job *__restrict__ jobs, const int *__restrict__ hot,
                                                    int *__restrict__ counters, const Region *__restrict__ regions,
                                                    const double *__restrict__ z, const int2 *__restrict__ cand,
                                                    const int *__restrict__ cand_cnt, double thr, int min_search,
                                                    const unsigned int *__restrict__ bits,
                                                    const long long *__restrict__ bit_off, Seg *__restrict__ segs,
                                                    int seg_cap, Job *__restrict__ next, int job_cap,
                                                    int *__restrict__ brute, int *__restrict__ next_count) {
    const int h = blockIdx.x, tid = threadIdx.x;
    if (h >= counters[2]) return;
    const int j = hot[h];
    const Job job = jobs[j];
    const int n_hi = cand_cnt[2 * h], n_lo = cand_cnt[2 * h + 1];
    if (n_hi > CAND_CAP || n_lo > CAND_CAP) {  // massive ties: evaluate everything exactly
        if (tid == 0) brute[atomicAdd(&counters[3], 1)] = j;
        return;
    }
    const double *zz = z + regions[job.region].off;
    const WindowMask wm{bits, bits ? bit_off[job.region] : 0, regions[job.region].n};
    BestPair b;
    b.maxv = 0.0; b.minv = 0.0; b.mx = b.my = b.nx = b.ny = -1;
    // a wave per candidate: waves 0-1 evaluate candidates for the maximum, waves 2-3 for the
    // minimum, two of each per trip; the trip count follows the longer list (usually one trip)
    __shared__ wc::PwWaveScratch sc[4];
    const int w = tid >> 6, lane = tid & 63;
    const int which = w >> 1, slot = w & 1;
    const int n_mine = which == 0 ? n_hi : n_lo, n_most = n_hi > n_lo ? n_hi : n_lo;
    for (int base = 0; base < n_most; base += 2) {
        const int t = base + slot;
        if (t >= n_mine) continue;                                  // wave-uniform
        const int2 cw = cand[((int64_t)2 * h + which) * CAND_CAP + t];
        const double v = window_exact_wave(zz, cw.x, cw.y, lane, wm, sc[w]);
        if (lane == 0) {
            if (which == 0) {
                if (better_max(v, cw.x, cw.y, b.maxv, b.mx, b.my)) { b.maxv = v; b.mx = cw.x; b.my = cw.y; }
            } else {
                if (better_min(v, cw.x, cw.y, b.minv, b.nx, b.ny)) { b.minv = v; b.nx = cw.x; b.ny = cw.y; }
            }
        }
    }
    block_best(b, tid);
    if (tid == 0)
        decide_emit(job, b.maxv, b.mx, b.my, b.minv, b.nx, b.ny, thr, min_search, segs, seg_cap, next, job_cap, counters,
                    next_count);
}

// Exact evaluation of every window of a job (non-finite input or tie overflow).
__global__ __launch_bounds__(256) void k_seg_brute(const Job *__restrict__ jobs, const int *__restrict__ brute,
                                                   int *__restrict__ counters, const Region *__restrict__ regions,
                                                   const double *__restrict__ z, double thr, int min_search,
                                                   const unsigned int *__restrict__ bits,
                                                   const long long *__restrict__ bit_off, Seg *__restrict__ segs,
                                                   int seg_cap, Job *__restrict__ next, int job_cap,
                                                   int *__restrict__ next_count) {
    const int q = blockIdx.x, tid = threadIdx.x;
    if (q >= counters[3]) return;
    const Job job = jobs[brute[q]];
    const double *zz = z + regions[job.region].off;
    const WindowMask wm{bits, bits ? bit_off[job.region] : 0, regions[job.region].n};
    BestPair b;
    b.maxv = 0.0; b.minv = 0.0; b.mx = b.my = b.nx = b.ny = -1;
    for (int x = job.lo; x < job.hi; ++x)
        for (int y = x + tid; y < job.hi; y += 256) {
            double v = window_exact<false>(zz, x, y, 0, wm);
            if (better_max(v, x, y, b.maxv, b.mx, b.my)) { b.maxv = v; b.mx = x; b.my = y; }
            if (better_min(v, x, y, b.minv, b.nx, b.ny)) { b.minv = v; b.nx = x; b.ny = y; }
        }
    block_best(b, tid);
    if (tid == 0 && b.mx >= 0)
        decide_emit(job, b.maxv, b.mx, b.my, b.minv, b.nx, b.ny, thr, min_search, segs, seg_cap, next, job_cap, counters,
                    next_count);
}

// numpy's pairwise sum of v[0..n) by a whole workgroup (n small enough that DEPTH halvings reach
// leaves of <= 128 elements: a half has at most n / 2 + 7): the leaves of numpy's
// tree (<= 128 elements each) are summed at the same time, one per group of eight lanes, then
// folded in the tree's order.  The tree is walked by compile-time recursion (no stack in memory).
// All threads must call; leaf_sum: LDS scratch for the leaf sums (>= (1 << DEPTH) doubles);
// every thread returns the sum.
template <int DEPTH>
struct PwBlock {
    template <class V>
    static __device__ inline void leaves(V v, int off, int n, int &next_leaf, int group, int sub, double *leaf_sum) {
        if (n <= WC_PW_BLOCK) {
            if (next_leaf == group) {
                const double s = wc::pw_leaf_group8([&](int64_t i) { return v[i]; }, off, n, sub);
                if (sub == 0) leaf_sum[next_leaf] = s;
            }
            ++next_leaf;
        } else {
            int n2 = n / 2;
            n2 -= n2 % 8;
            PwBlock<DEPTH - 1>::leaves(v, off, n2, next_leaf, group, sub, leaf_sum);
            PwBlock<DEPTH - 1>::leaves(v, off + n2, n - n2, next_leaf, group, sub, leaf_sum);
        }
    }
    static __device__ inline double fold(int n, int &next_leaf, const double *leaf_sum) {
        if (n <= WC_PW_BLOCK) return leaf_sum[next_leaf++];
        int n2 = n / 2;
        n2 -= n2 % 8;
        const double l = PwBlock<DEPTH - 1>::fold(n2, next_leaf, leaf_sum);
        const double r = PwBlock<DEPTH - 1>::fold(n - n2, next_leaf, leaf_sum);
        return l + r;
    }
};
template <>
struct PwBlock<0> {
    template <class V>
    static __device__ inline void leaves(V v, int off, int n, int &next_leaf, int group, int sub, double *leaf_sum) {
        if (next_leaf == group) {
            const double s = wc::pw_leaf_group8([&](int64_t i) { return v[i]; }, off, n, sub);
            if (sub == 0) leaf_sum[next_leaf] = s;
        }
        ++next_leaf;
    }
    static __device__ inline double fold(int, int &next_leaf, const double *leaf_sum) { return leaf_sum[next_leaf++]; }
};

// Latency mode: minrefbins cleaning (k_clean), prefix sums / sum |z| / finiteness (k_region_prefix),
// the whole-region Stouffer value (k_region_whole) and the root job (k_init_jobs) of one region by one
// workgroup of 256 threads -- four launches of one-wave-per-region kernels (sixteen dependent
// 64-bin trips each) as one with four times fewer trips.  The prefix sums are the approximate
// side of the search (any summation order satisfies window_eps' bound); everything decided later
// is re-evaluated exactly, so the outputs equal the general path's bit for bit.
template <int NT>     // workgroup size: 1024 for a handful of regions (latency mode), 256 for a batch's thousands
__global__ __launch_bounds__(NT) void k_lat_setup(const double *__restrict__ zsrc, const double *__restrict__ rsrc,
                                                   const double *__restrict__ nsrc, int64_t str_i, int64_t str_b,
                                                   int64_t B, const int64_t *__restrict__ moff,
                                                   const int64_t *__restrict__ goff, const int *__restrict__ m2g,
                                                   const int *__restrict__ sel, int n_sel, double minref,
                                                   double *__restrict__ zc, double *__restrict__ rc,
                                                   int *__restrict__ gpos, Region *__restrict__ regions,
                                                   double *__restrict__ prefix, double *__restrict__ reg_abs,
                                                   int *__restrict__ reg_flag, double *__restrict__ whole,
                                                   double *__restrict__ whole2, Job *__restrict__ jobs,
                                                   int *__restrict__ counters, int *__restrict__ out_n,
                                                   int *__restrict__ misc, int64_t n_regions,
                                                   const int *__restrict__ tail_flag = nullptr) {
    extern __shared__ double zl[];                // the region's kept z values (the prefix pass and the exact sum read them here)
    constexpr int NW = NT / 64;
    __shared__ int s_cnt[NW];
    __shared__ double s_sum[NW], s_abs[NW];
    __shared__ int s_fin[NW];
    __shared__ double s_leaf[32];
    __shared__ wc::PwWaveScratch sc;
    const int tid = threadIdx.x, lane = tid & 63, w = tid >> 6;
    const int64_t r = blockIdx.x;
    if (r == 0 && tid < 8) counters[tid] = tid == 1 ? (int)n_regions : 0;
    if (r == 0 && tid >= 8 && tid < 12 && misc) misc[tid - 8] = (tid == 9 && tail_flag) ? *tail_flag : 0;   // ([1]: the late repeats' overflow word, see run_repeat)
    WC_STAMP(0);
    const int64_t i = r / n_sel;
    const int si = (int)(r - i * n_sel);
    const int c = sel[si];
    const int64_t cs = moff[c], ce = moff[c + 1];
    const int64_t off = i * B + cs;
    // 1. keep bins with enough reference bins, in order (wisecondor.py:215-222)
    int count = 0;
    for (int64_t base = cs; base < ce; base += NT) {
        const int64_t b = base + tid;
        bool keep = false;
        if (b < ce) keep = nsrc[i * str_i + b * str_b] >= minref;
        const unsigned long long mask = __ballot(keep);
        if (lane == 0) s_cnt[w] = __popcll(mask);
        wc_sync();
        int before = count;
        for (int q = 0; q < w; ++q) before += s_cnt[q];
        if (keep) {
            const int at = before + __popcll(mask & ((1ull << lane) - 1ull));
            const double zv = zsrc[i * str_i + b * str_b];
            zc[off + at] = zv;
            zl[at] = zv;
            rc[off + at] = rsrc[i * str_i + b * str_b];
            gpos[off + at] = (int)(m2g[b] - goff[c]);
        }
        for (int q = 0; q < NW; ++q) count += s_cnt[q];
        wc_sync();
    }
    const int n = count;
    if (tid == 0) {
        Region rg;
        rg.off = off; rg.n = n; rg.pad = c;
        regions[r] = rg;
        out_n[r] = 0;
        Job j;
        j.region = (int)r; j.lo = 0; j.hi = n; j.pad = 0;
        jobs[r] = j;
    }
    wc_sync();
    WC_STAMP(1);
    // 2. prefix sums, sum |z|, finiteness
    const double *zz = zl;
    double *P = prefix + off + r;
    if (tid == 0) P[0] = 0.0;
    double run = 0.0, a = 0.0;
    int finite = 1;
    for (int t0 = 0; t0 < n; t0 += NT) {
        const int t = t0 + tid;
        const double v = t < n ? zz[t] : 0.0;
        if (!isfinite(v)) finite = 0;
        a += fabs(v);
        double incl = v;
        for (int o = 1; o < 64; o <<= 1) {
            const double up = __shfl_up(incl, o);
            if (lane >= o) incl += up;
        }
        if (lane == 63) s_sum[w] = incl;
        wc_sync();
        double before = run;
        for (int q = 0; q < w; ++q) before += s_sum[q];
        if (t < n) P[t + 1] = before + incl;
        for (int q = 0; q < NW; ++q) run += s_sum[q];
        wc_sync();
    }
    for (int o = 32; o > 0; o >>= 1) {
        a += __shfl_xor(a, o);
        finite &= __shfl_xor(finite, o);
    }
    if (lane == 0) { s_abs[w] = a; s_fin[w] = finite; }
    wc_sync();
    if (tid == 0) {
        double aa = 0.0;
        int ff = 1;
        for (int q = 0; q < NW; ++q) { aa += s_abs[q]; ff &= s_fin[q]; }
        reg_abs[r] = aa;
        reg_flag[r] = ff;
    }
    WC_STAMP(2);
    // 3. getValue(0, n - 1): numpy's own sum of the whole region / sqrt(n) (wisecondor.py:237): the
    // leaves of numpy's pairwise tree at once, one per group of eight lanes, then the fold
    {
        double v = NAN;
        if (n > 0 && n <= 2048) {        // five halvings of <= 2048 (each half at most n / 2 + 7) end at <= 128 elements
            int next = 0;
            PwBlock<5>::leaves(zz, 0, n, next, tid >> 3, tid & 7, s_leaf);
            wc_sync();
            next = 0;
            v = (PwBlock<5>::fold(n, next, s_leaf) + 0.0) / sqrt((double)n);      // (+ 0.0: np.sum starts from its identity)
        } else if (n > 0 && w == 0) {
            const WindowMask wm{nullptr, 0, n};
            v = window_exact_wave(zz, 0, n - 1, lane, wm, sc);
        }
        if (tid == 0) {
            whole[r] = v;
            if (whole2) whole2[r] = v;
        }
    }
    WC_STAMP(3);
}

// Latency mode: the whole segmentation of a region AND its calls in ONE launch, one workgroup of
// 1024 threads per region: for the region and then for every child range a block-local value search
// (the general path's inner loops, sixteen waves), the candidates within 2 eps of the extremes, their exact
// values and the reference's champion rule (triarray.py:59-84) -- the region's prefix array and the
// job stack live in LDS; then the region's segments are put in order and turned into call rows
// (coordinates with the reference's end quirk, median effect; wisecondor.py:239-257).  Regions are
// independent, so no workgroup waits for another and the host never sees a round.  Anything this
// kernel is not built for (non-finite values, more than CAND_CAP tied candidates, a deeper stack
// than TREE_STACK, more than TREE_SEGS segments) raises counters[6]: the caller repeats the call
// on the general path.
constexpr int TREE_STACK = 64;
constexpr int TREE_CHUNKS = 17;     // row blocks of the longest region the tree kernel takes (2048 bins)
constexpr int TREE_SEGS = 128;
constexpr int TREE_MAXLEN = 2048;
__device__ inline void seg_tree_region(const int region, int *__restrict__ counters, const Region *__restrict__ regions,
                                       int64_t n_regions, const int *__restrict__ reg_flag,
                                       const double *__restrict__ prefix, const double *__restrict__ rs,
                                       const double *__restrict__ reg_abs, const double *__restrict__ z,
                                       const double *__restrict__ ratio, const int *__restrict__ gpos,
                                       double thr, int min_search, int max_calls,
                                       double *__restrict__ reg_calls, int *__restrict__ out_n,
                                       const Extreme *__restrict__ partial,
                                       const double2 *__restrict__ sub, int max_chunks) {
    extern __shared__ double pl[];                 // the region's prefix array, n + 1 doubles (later: median scratch)
    __shared__ Job stack[TREE_STACK];
    __shared__ int s_sp, s_nhi, s_nlo, s_nseg, s_root;
    __shared__ int2 c_hi[CAND_CAP], c_lo[CAND_CAP];
    __shared__ double red_max[16], red_min[16];
    __shared__ double ext_max[TREE_CHUNKS][2][16], ext_min[TREE_CHUNKS][2][16];   // per (row block, side, wave)
    __shared__ wc::PwWaveScratch sc[4];
    __shared__ BestPair s_best[4];
    __shared__ double seg_val[TREE_SEGS];
    __shared__ int seg_x[TREE_SEGS], seg_y[TREE_SEGS];
    __shared__ double s_mid[2];
    __shared__ int s_nan;
    __shared__ unsigned int s_reach[TREE_CHUNKS];
    __shared__ unsigned int s_loud[TREE_CHUNKS];       // slices of the range [s_ext_lo, s_ext_hi) that may hold a call
    __shared__ int s_ext_lo, s_ext_hi, s_found;
    const int tid = threadIdx.x;
    if (region >= n_regions) return;
    const Region rg = regions[region];
    if (rg.n <= 0) return;
    if (!reg_flag[region] || rg.n > TREE_MAXLEN) {     // non-finite values: the general path's exact scan
        if (tid == 0) counters[6] = 1;
        return;
    }
    WC_STAMP_AT(8, region);
    const long long a0 = rg.off + region;              // absolute index of the region's P[0]
    const double *Pg = prefix + a0;
    double *zl = pl + (rg.n + 1);                      // the region's z values, for the exact sums
    for (int i = tid; i <= rg.n; i += 1024) pl[i] = Pg[i];
    for (int i = tid; i < rg.n; i += 1024) zl[i] = z[rg.off + i];
    if (tid == 0) { s_sp = 1; s_nseg = 0; s_root = 1; s_ext_lo = 0; s_ext_hi = -1; }
    wc_sync();
    const double eps = window_eps(rg.n, reg_abs[region]);
    const double *zz = zl;
    const WindowMask wm{nullptr, 0, rg.n};
    const int lane = tid & 63, w = tid >> 6;
    if (tid == 0) {
        Job root;
        root.region = region; root.lo = 0; root.hi = rg.n; root.pad = 0;
        stack[0] = root;
    }
    wc_sync();
    // which slices of the job whose extremes sit in ext_max / ext_min may hold a call (same test as a whole
    // range's: the larger magnitude + eps against the threshold); bit side * 16 + wave of s_loud[row block]
    auto loud_slices = [&](int nch_job) {
        for (int t = tid; t < nch_job; t += 1024) s_loud[t] = 0u;
        wc_sync();
        for (int t = tid; t < nch_job * 32; t += 1024) {
            const double mx = (&ext_max[0][0][0])[t], mn = (&ext_min[0][0][0])[t];
            const bool empty = mx == -INFINITY && mn == INFINITY;       // a slice without a window
            if (!empty && !(fmax(fabs(mx), fabs(mn)) + eps < thr)) atomicOr(&s_loud[t >> 5], 1u << (t & 31));
        }
        wc_sync();
    };
    while (true) {
        wc_sync();
        if (s_sp == 0) break;
        const Job job = stack[s_sp - 1];
        const bool is_root = s_root != 0;
        wc_sync();
        if (tid == 0) { --s_sp; s_nhi = 0; s_nlo = 0; s_root = 0; }
        wc_sync();
        ScanCtx c;
        c.P = pl; c.lo = job.lo; c.hi = job.hi; c.L = job.hi - job.lo; c.half = (c.L + 1) / 2; c.chunk = 0;
        c.wm = wm;
        if (c.L <= 0) continue;
        const int nch = (c.half + ROWS_HALF - 1) / ROWS_HALF;
        // (no quiet certificate here: with sixteen waves on a range its full value search takes a
        // few microseconds, less than the certificate's barrier rounds)
        double emax, emin;
        if (is_root) {
            // the whole region was searched by k_seg_search, all row blocks in parallel on as many
            // compute units: its per-block extremes stand in for the block-local search
            emax = -INFINITY;
            emin = INFINITY;
            for (int ch = 0; ch < nch; ++ch) {
                const Extreme pe = partial[(int64_t)region * max_chunks + ch];
                emax = fmax(emax, pe.maxv);
                emin = fmin(emin, pe.minv);
            }
            for (int t = tid; t < nch * 32; t += 1024) {
                const int ch = t >> 5, side = (t >> 4) & 1, q = t & 15;
                const double2 e2 = sub[((int64_t)region * max_chunks + ch) * 8 + side * 4 + (q >> 2)];
                ext_max[ch][side][q] = e2.x;
                ext_min[ch][side][q] = e2.y;
            }
            wc_sync();
            if (fmax(fabs(emax), fabs(emin)) + eps < thr) continue;
            loud_slices(nch);
            if (tid == 0) { s_ext_lo = job.lo; s_ext_hi = job.hi; }
        } else {
            // A range inside the range whose slice extremes are still at hand (a child of the job just
            // decided, or of the root): every window of it lies in one of those slices, and a slice
            // whose own extremes stay below the threshold cannot hold a call.  Only the windows of the few
            // loud slices that fall inside this range are evaluated -- a few thousand instead of the
            // range's L^2 / 2 -- and if none reaches the threshold the range is done.
            if (s_ext_lo <= job.lo && job.hi <= s_ext_hi) {
                ScanCtx pc;
                pc.P = pl; pc.lo = s_ext_lo; pc.hi = s_ext_hi; pc.L = pc.hi - pc.lo; pc.half = (pc.L + 1) / 2;
                pc.wm = wm;
                pc.clip_lo = job.lo; pc.clip_hi = job.hi;
                const int pnch = (pc.half + ROWS_HALF - 1) / ROWS_HALF;
                if (tid == 0) s_found = 0;
                wc_sync();
                bool found = false;
                for (int ch = 0; ch < pnch; ++ch) {
                    const unsigned int loud = s_loud[ch];
                    if (loud == 0u) continue;
                    pc.chunk = ch;
                    scan_chunk(pc, rs, tid, [&](double v, int, int) {
                        if (!(fabs(v) + eps < thr)) found = true;
                    }, ~loud);
                }
                if (found) s_found = 1;
                wc_sync();
                const bool quiet = s_found == 0;
                wc_sync();
                if (quiet) continue;
            }
            // value search with the general path's inner loops (four float64 operations per window);
            // every (row block, side, wave) leaves its extremes behind for the candidate pass
            // Four (row block, side) pairs at a time, four waves each: a pair costs a fixed round of
            // loads and reductions whatever its size, and a child range has up to 2 x 17 of them.
            const double *P = pl + job.lo;
            double bmax = -INFINITY, bmin = INFINITY;
            int dummy = 0;
            const int grp = w >> 2, w4 = w & 3;
            for (int cs = grp; cs < 2 * nch; cs += 4) {
                const int ch = cs >> 1, side = cs & 1;
                double smax, smin;
                search_side<false, 4>(P, c.L, c.half, ch, side, rs, lane, w4, job.lo, wm, smax, smin, dummy);
                // slots 4 w4 .. 4 w4 + 3: the candidate pass's waves that scan this search wave's lengths
                if (lane < 4) { ext_max[ch][side][4 * w4 + lane] = smax; ext_min[ch][side][4 * w4 + lane] = smin; }
                bmax = fmax(bmax, smax);
                bmin = fmin(bmin, smin);
            }
            if (lane == 0) { red_max[w] = bmax; red_min[w] = bmin; }
            wc_sync();
            emax = red_max[0];
            emin = red_min[0];
            for (int q = 1; q < 16; ++q) { emax = fmax(emax, red_max[q]); emin = fmin(emin, red_min[q]); }
            wc_sync();
            if (fmax(fabs(emax), fabs(emin)) + eps < thr) {              // no call in this range (k_seg_classify's test)
                if (tid == 0) s_ext_hi = -1;                             // the slice extremes at hand are this range's now: drop them
                continue;
            }
            loud_slices(nch);
            if (tid == 0) { s_ext_lo = job.lo; s_ext_hi = job.hi; }
        }
        // windows within 2 eps of the extremes: one of them is numpy's argmax / argmin.  Only the
        // (row block, side) pairs whose own extremes reach a cut are scanned again.
        WC_STAMP_AT(is_root ? 9 : 13, region);
        const double hi_cut = emax - 2.0 * eps, lo_cut = emin + 2.0 * eps;
        // which (row block, side, wave) slices reach a cut: one thread per stored extreme, one LDS word per block
        // (every thread walking the 32 extremes of every block itself, with short-circuit tests, was a
        // chain of ~250 dependent LDS reads: 11 us at the root of a 900-bin region)
        for (int t = tid; t < nch; t += 1024) s_reach[t] = 0u;
        wc_sync();
        for (int t = tid; t < nch * 32; t += 1024) {
            const bool reach = !((&ext_max[0][0][0])[t] < hi_cut && (&ext_min[0][0][0])[t] > lo_cut);
            if (reach) atomicOr(&s_reach[t >> 5], 1u << (t & 31));      // bit side * 16 + candidate-pass wave
        }
        wc_sync();
        for (int ch = 0; ch < nch; ++ch) {
            const unsigned int reach = s_reach[ch];
            if (reach == 0u) continue;
            const unsigned int skip = ~reach;
            c.chunk = ch;
            scan_chunk(c, rs, tid, [&](double v, int x, int y) {
                if (v >= hi_cut) {
                    const int at = atomicAdd(&s_nhi, 1);
                    if (at < CAND_CAP) c_hi[at] = make_int2(x, y);
                }
                if (v <= lo_cut) {
                    const int at = atomicAdd(&s_nlo, 1);
                    if (at < CAND_CAP) c_lo[at] = make_int2(x, y);
                }
            }, skip);
        }
        wc_sync();
        WC_STAMP_AT(is_root ? 10 : 14, region);
        const int n_hi = s_nhi, n_lo = s_nlo;
        if (n_hi > CAND_CAP || n_lo > CAND_CAP) {     // massive ties: the general path evaluates everything exactly
            if (tid == 0) counters[6] = 1;
            return;
        }
        // exact values of the candidates (numpy pairwise sum / sqrt) by the first four waves: waves
        // 0-1 the candidates for the maximum, 2-3 for the minimum
        if (w < 4) {
            BestPair b;
            b.maxv = 0.0; b.minv = 0.0; b.mx = b.my = b.nx = b.ny = -1;
            const int which = w >> 1, slot = w & 1;
            const int n_mine = which == 0 ? n_hi : n_lo;
            for (int t = slot; t < n_mine; t += 2) {
                const int2 cw = which == 0 ? c_hi[t] : c_lo[t];
                const double v = window_exact_wave(zz, cw.x, cw.y, lane, wm, sc[w]);
                if (which == 0) {
                    if (better_max(v, cw.x, cw.y, b.maxv, b.mx, b.my)) { b.maxv = v; b.mx = cw.x; b.my = cw.y; }
                } else {
                    if (better_min(v, cw.x, cw.y, b.minv, b.nx, b.ny)) { b.minv = v; b.nx = cw.x; b.ny = cw.y; }
                }
            }
            if (lane == 0) s_best[w] = b;
        }
        wc_sync();
        WC_STAMP_AT(is_root ? 11 : 15, region);
        if (tid == 0) {
            BestPair b = s_best[0];
            if (s_best[1].mx >= 0 && better_max(s_best[1].maxv, s_best[1].mx, s_best[1].my, b.maxv, b.mx, b.my)) {
                b.maxv = s_best[1].maxv; b.mx = s_best[1].mx; b.my = s_best[1].my;
            }
            b.minv = s_best[2].minv; b.nx = s_best[2].nx; b.ny = s_best[2].ny;
            if (s_best[3].nx >= 0 && better_min(s_best[3].minv, s_best[3].nx, s_best[3].ny, b.minv, b.nx, b.ny)) {
                b.minv = s_best[3].minv; b.nx = s_best[3].nx; b.ny = s_best[3].ny;
            }
            double champ = b.maxv;
            int cx = b.mx, cy = b.my;
            if (fabs(b.minv) > champ) { champ = b.minv; cx = b.nx; cy = b.ny; }
            if (!(fabs(champ) < thr)) {
                if (s_nseg < TREE_SEGS) { seg_val[s_nseg] = champ; seg_x[s_nseg] = cx; seg_y[s_nseg] = cy; ++s_nseg; }
                else counters[6] = 1;
                const int xr = cx - job.lo, yr = cy - job.lo, edge = job.hi - job.lo;
                const bool left = xr > min_search, right = yr + 1 < edge - min_search;
                if (s_sp + (left ? 1 : 0) + (right ? 1 : 0) > TREE_STACK) {
                    counters[6] = 1;
                } else {
                    if (right) { Job n; n.region = job.region; n.lo = cy + 1; n.hi = job.hi; n.pad = 0; stack[s_sp++] = n; }
                    if (left) { Job n; n.region = job.region; n.lo = job.lo; n.hi = cx; n.pad = 0; stack[s_sp++] = n; }
                }
            }
        }
    }
    // ---- the region's calls, in position order (k_seg_gather + k_call_post of the general path)
    wc_sync();
    WC_STAMP_AT(16, region);
    const int nseg = s_nseg;
    if (tid == 0) {
        out_n[region] = nseg;
        atomicAdd(&counters[4], nseg);
    }
    const double *rr = ratio + rg.off;
    for (int sidx = 0; sidx < nseg; ++sidx) {
        const int x = seg_x[sidx], y = seg_y[sidx], L = y - x + 1;
        int rank = 0;
        for (int u = 0; u < nseg; ++u) rank += seg_x[u] < x;
        if (rank >= max_calls) continue;                   // k_assemble_calls reports the overflow from out_n
        wc_sync();
        if (tid == 0) s_nan = 0;
        wc_sync();
        double *sv = pl;                                   // the prefix array is not needed any more
        for (int e = tid; e < L; e += 1024) {
            const double v = rr[x + e];
            sv[e] = v;
            if (v != v) s_nan = 1;
        }
        wc_sync();
        const bool has_nan = s_nan != 0;
        if (!has_nan) {
            // every value counts the values below / at-or-below it; the value whose interval covers
            // a middle rank is that order statistic (np.median: mean of the middle pair)
            const int k_lo = (L - 1) / 2, k_hi = L / 2;
            for (int e = tid; e < L; e += 1024) {
                const double xv = sv[e];
                int lt = 0, le = 0;
                for (int u = 0; u < L; ++u) {
                    const double yv = sv[u];
                    lt += yv < xv;
                    le += yv <= xv;
                }
                if (lt <= k_lo && k_lo < le) s_mid[0] = xv;      // equal values write the same number
                if (lt <= k_hi && k_hi < le) s_mid[1] = xv;
            }
        }
        wc_sync();
        if (tid == 0) {
            double med = (L & 1) ? s_mid[0] : (s_mid[0] + s_mid[1]) / 2.0;
            if (has_nan) med = NAN;
            // the end walk of the reference restarts at `start` and re-counts it:
            // end = position(survivor y-1) + 1, or start itself when y == x
            const int start = gpos[rg.off + x];
            const int end = (y > x) ? gpos[rg.off + y - 1] + 1 : start;
            double *o = reg_calls + ((int64_t)region * max_calls + rank) * 5;
            o[0] = (double)(rg.pad + 1);
            o[1] = (double)start;
            o[2] = (double)end;
            o[3] = seg_val[sidx];
            o[4] = med - 1.0;
        }
    }
    WC_STAMP_AT(17, region);
}

// Order each region's segments by position (the reference's in-order recursion).
__global__ __launch_bounds__(256) void k_seg_gather(const Seg *__restrict__ segs, int n_segs, int max_calls,
                                                    double *__restrict__ out_val, int *__restrict__ out_x,
                                                    int *__restrict__ out_y, int *__restrict__ out_n,
                                                    const int *__restrict__ n_segs_dev) {
    if (n_segs_dev) n_segs = *n_segs_dev < n_segs ? *n_segs_dev : n_segs;      // n_segs: the grid's bound
    // rank of a segment among the segments of its region (by start bin); the segment list is
    // streamed through LDS in tiles of 256
    __shared__ int t_region[256], t_x[256];
    const int s = blockIdx.x * 256 + threadIdx.x;
    const bool live = s < n_segs;
    Seg me;
    me.val = 0.0; me.region = -1; me.x = 0; me.y = 0; me.pad = 0;
    if (live) me = segs[s];
    int rank = 0;
    for (int t0 = 0; t0 < n_segs; t0 += 256) {
        wc_sync();
        const int t = t0 + threadIdx.x;
        t_region[threadIdx.x] = t < n_segs ? segs[t].region : -2;
        t_x[threadIdx.x] = t < n_segs ? segs[t].x : 0;
        wc_sync();
        const int m = n_segs - t0 < 256 ? n_segs - t0 : 256;
        for (int u = 0; u < m; ++u) rank += (t_region[u] == me.region) & (t_x[u] < me.x);
    }
    if (!live) return;
    atomicAdd(&out_n[me.region], 1);
    if (rank < max_calls) {
        int64_t at = (int64_t)me.region * max_calls + rank;
        out_val[at] = me.val;
        out_x[at] = me.x;
        out_y[at] = me.y;
    }
}

constexpr int CP_THREADS = 1024;     // threads of a k_call_post workgroup (one per segment)
// k-th smallest (0-based) of v[0..L) by radix selection on the ordered 64-bit image;
// all CP_THREADS threads of the workgroup take part.
template <int NT, class F>          // NT >= 256 threads; at(e): the e-th value (a lambda over an LDS array reads it as LDS)
__device__ inline double block_select_of(F at, int L, int k, int tid) {
    // NT threads; the 256 digit buckets are scanned by the first four waves
    __shared__ unsigned int hist[256];
    __shared__ unsigned int s_wsum[4];
    __shared__ unsigned long long s_prefix;
    __shared__ int s_k;
    unsigned long long prefix = 0ull, mask = 0ull;
    for (int shift = 56; shift >= 0; shift -= 8) {
        if (tid < 256) hist[tid] = 0;
        wc_sync();
        for (int e0 = tid; e0 < L; e0 += 4 * NT) {            // four values per trip: their loads are in flight together
            double x4[4];
#pragma unroll
            for (int u = 0; u < 4; ++u) x4[u] = e0 + u * NT < L ? at(e0 + u * NT) : 0.0;
#pragma unroll
            for (int u = 0; u < 4; ++u) {
                const unsigned long long key = wc::f64_ordered(x4[u]);
                if (e0 + u * NT < L && (key & mask) == prefix) atomicAdd(&hist[(unsigned)(key >> shift) & 255u], 1u);
            }
        }
        wc_sync();
        unsigned int h = 0, incl = 0;
        const int lane = tid & 63, wv = tid >> 6;
        if (tid < 256) {
            // the digit whose bucket holds rank k: prefix sums of the 256 counts (a walk by one thread is
            // 255 dependent LDS reads per digit)
            h = hist[tid];
            incl = h;
            for (int o = 1; o < 64; o <<= 1) {
                const unsigned int up = __shfl_up(incl, o);
                if (lane >= o) incl += up;
            }
            if (lane == 63) s_wsum[wv] = incl;
        }
        wc_sync();
        if (tid < 256) {
            for (int q = 0; q < wv; ++q) incl += s_wsum[q];
            const unsigned int excl = incl - h;
            if ((unsigned int)k >= excl && (unsigned int)k < incl) {     // exactly one bucket (k < number of values)
                s_k = k - (int)excl;
                s_prefix = prefix | ((unsigned long long)tid << shift);
            }
        }
        wc_sync();
        k = s_k;
        prefix = s_prefix;
        mask |= 0xFFull << shift;
        wc_sync();
    }
    return wc::f64_from_ordered(prefix);
}
template <int NT = CP_THREADS>
__device__ inline double block_select(const double *__restrict__ v, int L, int k, int tid) {
    return block_select_of<NT>([&](int e) { return v[e]; }, L, k, tid);
}
// The two middle order statistics np.median needs, ranks k and k2 (k2 == k or k + 1): the radix selection once, then
// ONE pass for the successor -- the smallest value above the k-th unless the k-th value occurs often enough to hold
// rank k2 as well -- instead of a second selection.
template <int NT, class F>
__device__ inline void block_select_pair(F at, int L, int k, int k2, int tid, double &lo, double &hi) {
    __shared__ unsigned long long s_next;
    __shared__ int s_le;
    lo = block_select_of<NT>(at, L, k, tid);
    hi = lo;
    if (k2 == k) return;
    const unsigned long long klo = wc::f64_ordered(lo);
    if (tid == 0) { s_next = ~0ull; s_le = 0; }
    wc_sync();
    unsigned long long mine = ~0ull;
    int le = 0;
    for (int e = tid; e < L; e += NT) {
        const unsigned long long key = wc::f64_ordered(at(e));
        if (key <= klo) ++le;
        else if (key < mine) mine = key;
    }
    for (int o = 32; o > 0; o >>= 1) {
        le += __shfl_xor(le, o);
        const unsigned long long other = (unsigned long long)__shfl_xor((long long)mine, o);
        mine = other < mine ? other : mine;
    }
    if ((tid & 63) == 0) { atomicAdd(&s_le, le); atomicMin(&s_next, mine); }
    wc_sync();
    if (s_le <= k2) hi = wc::f64_from_ordered(s_next);     // (ranks 0 .. s_le - 1 hold values <= lo)
    wc_sync();
}
__global__ __launch_bounds__(CP_THREADS) void k_call_post(const Seg *__restrict__ segs, int n_segs,
                                                   const Region *__restrict__ regions, const double *__restrict__ rc,
                                                   const int *__restrict__ gpos, int max_calls,
                                                   double *__restrict__ reg_calls,
                                                   const int *__restrict__ n_segs_dev) {
    __shared__ int s_flag[2];
    const int tid = threadIdx.x;
    if (n_segs_dev) n_segs = *n_segs_dev < n_segs ? *n_segs_dev : n_segs;
    if ((int)blockIdx.x >= n_segs) return;
    const Seg me = segs[blockIdx.x];
    if (tid == 0) { s_flag[0] = 0; s_flag[1] = 0; }
    wc_sync();
    int rank = 0;
    for (int t = tid; t < n_segs; t += CP_THREADS) {
        const Seg o = segs[t];
        rank += (o.region == me.region && o.x < me.x);
    }
    if (rank) atomicAdd(&s_flag[0], rank);
    const Region rg = regions[me.region];
    const int x = me.x, y = me.y, L = y - x + 1;
    const double *v = rc + rg.off + x;
    for (int e = tid; e < L; e += CP_THREADS)
        if (v[e] != v[e]) s_flag[1] = 1;
    wc_sync();
    rank = s_flag[0];
    const bool has_nan = s_flag[1] != 0;
    if (rank >= max_calls) return;
    double lo = 0.0, hi = 0.0;
    if (!has_nan && L <= SHORT_SEG) {
        // short segment: each value counts how many others lie below / at-or-below it; the
        // value whose count interval covers a middle rank is that order statistic
        __shared__ double sv[SHORT_SEG];
        __shared__ double s_mid[2];
        for (int e = tid; e < L; e += CP_THREADS) sv[e] = v[e];
        wc_sync();
        const int k_lo = (L - 1) / 2, k_hi = L / 2;
        for (int e = tid; e < L; e += CP_THREADS) {
            const double xv = sv[e];
            int lt = 0, le = 0;
            for (int u = 0; u < L; ++u) {
                const double yv = sv[u];
                lt += yv < xv;
                le += yv <= xv;
            }
            if (lt <= k_lo && k_lo < le) s_mid[0] = xv;      // equal values write the same number
            if (lt <= k_hi && k_hi < le) s_mid[1] = xv;
        }
        wc_sync();
        lo = s_mid[0];
        hi = s_mid[1];
    } else if (!has_nan) {
        lo = block_select(v, L, (L - 1) / 2, tid);
        hi = (L & 1) ? lo : block_select(v, L, L / 2, tid);
    }
    if (tid == 0) {
        double med = (L & 1) ? lo : (lo + hi) / 2.0;  // np.median: mean of the middle pair
        if (has_nan) med = NAN;
        // the end walk of the reference restarts at `start` and re-counts it:
        // end = position(survivor y-1) + 1, or start itself when y == x
        int start = gpos[rg.off + x];
        int end = (y > x) ? gpos[rg.off + y - 1] + 1 : start;
        double *o = reg_calls + ((int64_t)me.region * max_calls + rank) * 5;
        o[0] = (double)(rg.pad + 1);
        o[1] = (double)start;
        o[2] = (double)end;
        o[3] = me.val;
        o[4] = med - 1.0;
    }
}

__device__ inline void assemble_one(const double *__restrict__ reg_calls, const int *__restrict__ out_n, int n_sel,
                                    int max_calls, int64_t Ns, double *__restrict__ calls, int *__restrict__ n_calls,
                                    int *__restrict__ overflow, int64_t i) {
    if (i >= Ns) return;
    int total = 0;
    for (int s = 0; s < n_sel; ++s) {
        int64_t r = i * n_sel + s;
        int n = out_n[r];
        if (n > max_calls) { *overflow = 1; n = max_calls; }
        for (int c = 0; c < n; ++c) {
            if (total < max_calls) {
                for (int f = 0; f < 5; ++f)
                    calls[(i * max_calls + total) * 5 + f] = reg_calls[(r * max_calls + c) * 5 + f];
            } else {
                *overflow = 1;
            }
            ++total;
        }
    }
    n_calls[i] = total < max_calls ? total : max_calls;
}

// (host_status, latency mode, one workgroup: the status words for the host go straight to pinned
// memory -- [16] call overflow, [24..31] the segmentation counters, [32] k_lat_repeats' overflow flag,
// [33] stdDevAvg left to the serial kernel)
__global__ void k_assemble_calls(const double *__restrict__ reg_calls, const int *__restrict__ out_n, int n_sel,
                                 int max_calls, int64_t Ns, double *__restrict__ calls, int *__restrict__ n_calls,
                                 int *__restrict__ overflow, int *__restrict__ host_status,
                                 const int *__restrict__ counters, const int *__restrict__ rep_overflow,
                                 const int *__restrict__ sd_fail) {
    int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (host_status) {
        assemble_one(reg_calls, out_n, n_sel, max_calls, Ns, calls, n_calls, overflow, i);
        wc_sync();
        const int t = threadIdx.x;
        if (t == 0) host_status[16] = overflow[0];
        if (t < 8) host_status[24 + t] = counters[t];
        if (t == 8) host_status[32] = rep_overflow[0];
        if (t == 9) {
            int f = 0;
            for (int64_t q = 0; q < Ns; ++q) f |= sd_fail[q];
            host_status[33] = f;
        }
        __threadfence_system();
        return;
    }
    assemble_one(reg_calls, out_n, n_sel, max_calls, Ns, calls, n_calls, overflow, i);
}


// The same for a batch: one wave per sample, lane s = the sample's s-th region (n_sel <= 64): the regions' call
// counts become offsets by a wave scan and every lane copies its own region's rows (one thread per sample walked
// 22 regions x their rows one after the other: two workgroups, 22 us per 125-sample batch).
__global__ __launch_bounds__(64) void k_assemble_batch(const double *__restrict__ reg_calls, const int *__restrict__ out_n,
                                                       int n_sel, int max_calls, int64_t Ns, double *__restrict__ calls,
                                                       int *__restrict__ n_calls, int *__restrict__ overflow) {
    const int64_t i = blockIdx.x;
    const int s = threadIdx.x;
    if (i >= Ns) return;
    const int64_t r = i * n_sel + s;
    int n = s < n_sel ? out_n[r] : 0;
    bool over = n > max_calls;
    if (over) n = max_calls;
    int incl = n;
    for (int o = 1; o < 64; o <<= 1) {
        const int up = __shfl_up(incl, o);
        if (s >= o) incl += up;
    }
    const int first = incl - n, total = __shfl(incl, 63);
    for (int c = 0; c < n; ++c) {
        if (first + c < max_calls) {
#pragma unroll
            for (int f = 0; f < 5; ++f)
                calls[(i * max_calls + first + c) * 5 + f] = reg_calls[(r * max_calls + c) * 5 + f];
        } else {
            over = true;
        }
    }
    if (over) *overflow = 1;
    if (s == 0) n_calls[i] = total < max_calls ? total : max_calls;
}

// Latency mode's last launch.  One 1024-thread workgroup per region walks the region's recursion
// (seg_tree_region); the same grid carries three riders that would otherwise be launches (or a
// forked graph branch) of their own:
//   * the first sd.blocks workgroups compute stdDevAvg (k_sd_fast's body, one per sample) -- a second
//     stream for that one kernel cost ~20 us of fork and join bubbles per call;
//   * the next inf.blocks workgroups write results_z / results_r (k_inflate's arithmetic);
//   * the LAST workgroup to finish (a counter in counters[7]) gathers the calls of every sample and
//     publishes the status words to the host (k_assemble_calls' job).
struct InflateRider {
    int blocks;
    const double *zs, *rs, *ns;
    int64_t B, Btot, Ns, si, sb;
    const int *g2m;
    double minref;
    double *res_z, *res_r;
};
__global__ __launch_bounds__(1024) void k_seg_tree(int *__restrict__ counters, const Region *__restrict__ regions,
                                                   int64_t n_regions, const int *__restrict__ reg_flag,
                                                   const double *__restrict__ prefix, const double *__restrict__ rs,
                                                   const double *__restrict__ reg_abs, const double *__restrict__ z,
                                                   const double *__restrict__ ratio, const int *__restrict__ gpos,
                                                   double thr, int min_search, int max_calls,
                                                   double *__restrict__ reg_calls, int *__restrict__ out_n,
                                                   const Extreme *__restrict__ partial,
                                                   const double2 *__restrict__ sub, int max_chunks, SdRider sd,
                                                   InflateRider inf,
                                                   const int *__restrict__ hot_list, const int *__restrict__ hot_count) {
    const int blk = (int)blockIdx.x;
    if (blk < sd.blocks) {
        extern __shared__ double pl[];                 // the launch's dynamic LDS: at least sizeof(SdShared) with a rider
        SdShared *sm = reinterpret_cast<SdShared *>(pl);
        const int64_t per = (sd.B + 1023) / 1024;
        if (per <= 12) sd_fast_block<12>(blk, sm, sd.sdT, sd.B, sd.Ns, sd.out, sd.fail, sd.out2, sd.sb, sd.si);
        else if (per <= 24) sd_fast_block<24>(blk, sm, sd.sdT, sd.B, sd.Ns, sd.out, sd.fail, sd.out2, sd.sb, sd.si);
        else sd_fast_block<0>(blk, sm, sd.sdT, sd.B, sd.Ns, sd.out, sd.fail, sd.out2, sd.sb, sd.si);
    } else if (blk < sd.blocks + inf.blocks) {
        const int64_t n = inf.Btot * inf.Ns;
        for (int64_t t = (int64_t)(blk - sd.blocks) * 1024 + threadIdx.x; t < n; t += (int64_t)inf.blocks * 1024) {
            const int64_t i = t / inf.Btot, g = t - i * inf.Btot;
            const int m = inf.g2m[g];
            double zv = 0.0, rv = 0.0;
            if (m >= 0 && inf.ns[i * inf.si + m * inf.sb] >= inf.minref) {
                zv = inf.zs[i * inf.si + m * inf.sb];
                rv = inf.rs[i * inf.si + m * inf.sb] - 1.0;
            }
            if (inf.res_z) inf.res_z[t] = zv;
            if (inf.res_r) inf.res_r[t] = rv;
        }
    } else {
        // the general path's first round hands over its hot jobs (job index == region index there):
        // workgroup h walks region hot_list[h]; the grid is an upper bound, surplus workgroups leave
        int region = blk - sd.blocks - inf.blocks;
        if (hot_list) region = region < *hot_count ? hot_list[region] : (int)n_regions;
        seg_tree_region(region, counters, regions, n_regions, reg_flag, prefix, rs, reg_abs, z, ratio,
                        gpos, thr, min_search, max_calls, reg_calls, out_n, partial, sub, max_chunks);
    }
}

// The whole recursion of TriArr.segmentTri (triarray.py:59-84) for one region by one workgroup, every range -- the
// region itself and every child -- searched with the cell bounds (cell_search): root search, classification,
// candidate list, exact decision, children, in ONE launch for the whole batch and without a host round trip
// (k_seg_job / k_seg_merge / k_seg_decide once per recursion level, and k_seg_quiet / k_seg_search / k_seg_classify /
// k_seg_tree of the 250 kb batches, are what it replaces); the region's call rows -- position order, genomic bounds,
// effect size -- at the end of the same workgroup's life (k_seg_gather + k_call_post).  What it is not built for -- non-finite values, more than CAND_CAP tied
// candidates, a recursion deeper than the stack -- sets counters[6] and the caller repeats the call with the
// host-driven rounds.
constexpr int WALK_STACK = 64;
// (Register allocation at the 128-register budget is knife-edge here: the same kernel WITHOUT the two evaluation
//  counters of the profiled run spills two registers, with them none -- profiles/r06_kernel_resources.txt.)
__global__ __launch_bounds__(256, 4) void k_seg_walk(int *__restrict__ counters, const Region *__restrict__ regions,
                                                  int n_regions, const int *__restrict__ reg_flag,
                                                  const double *__restrict__ prefix, const double *__restrict__ rs,
                                                  const double *__restrict__ reg_abs, const double *__restrict__ z,
                                                  double thr, int min_search, const double *__restrict__ tmin,
                                                  const double *__restrict__ tmax, const double *__restrict__ tmin2,
                                                  const double *__restrict__ tmax2, Seg *__restrict__ wsegs,
                                                  int seg_cap, int *__restrict__ out_n,
                                                  unsigned long long *__restrict__ work, int per_sample,
                                                  const WalkHot hot) {
    __shared__ CellShared sh;
    __shared__ Job stack[WALK_STACK];
    __shared__ int s_sp, s_nseg, s_stop;
    __shared__ BestPair s_best[4];
    __shared__ double seg_val[TREE_SEGS];
    __shared__ int seg_x[TREE_SEGS], seg_y[TREE_SEGS];
    const int tid0 = threadIdx.x;
    // The kernel lasts as long as its slowest workgroup, and the slow ones (a region with a long aberration: 150-250 us
    // against a mean of 44) must not be among the last to start: the first hot.cap workgroups of the grid take the
    // regions of the hot list (k_region_prefix), the regions' own workgroups then leave them alone.
    int region;
    if ((int)blockIdx.x < hot.cap) {
        if ((int)blockIdx.x >= *hot.count) return;
        region = hot.list[blockIdx.x];
        if (region < 0 || region >= n_regions || hot.index[region] != (int)blockIdx.x + 1) return;   // (an entry of an earlier batch)
    } else {
        const int wg = (int)blockIdx.x - hot.cap;
        region = wg;
        if (region >= n_regions) return;
        if (per_sample > 1) {                      // workgroup w: chromosome w / samples of sample w % samples
            const int samples = n_regions / per_sample;
            region = (wg % samples) * per_sample + wg / samples;
        }
        if (hot.cap > 0 && hot.index[region] != 0) return;         // (started early)
    }
    const Region rg = regions[region];
    if (rg.n <= 0) return;                         // (out_n was zeroed by the set-up kernel)
    if (rg.n > CJ_MAXLEN || !reg_flag[region]) {
        if (tid0 == 0) atomicOr(&counters[6], rg.n > CJ_MAXLEN ? 1 : 2);      // (the bits say why: tools/gpu_test_scale.py prints them with WC_TEST_VERBOSE)
        return;
    }
    const double eps = window_eps(rg.n, reg_abs[region]);
    const double eps2 = 2.0 * eps, T = thr - eps;
    const double *zz = z + rg.off;
    const WindowMask wm{nullptr, 0, rg.n};
    wc::PwWaveScratch *sc = reinterpret_cast<wc::PwWaveScratch *>(sh.pn);      // four of them fit pn .. itemq
    static_assert(4 * sizeof(wc::PwWaveScratch) <= sizeof(sh.pn) + sizeof(sh.b8x) + sizeof(sh.b8n) + sizeof(sh.q2) +
                                                       sizeof(sh.q1) + sizeof(sh.l1) + sizeof(sh.itemq),
                  "the exact evaluation's scratch does not fit the search's staging area");
    int wins = 0, evals = 0;
    if (tid0 == 0) {
        Job root;
        root.region = region; root.lo = 0; root.hi = rg.n; root.pad = 0;
        stack[0] = root;
        s_sp = 1; s_nseg = 0; s_stop = 0;
    }
    while (true) {
        // (the thread's number as something the compiler cannot see through: everything derived from it -- dozens of lane
        //  offsets and LDS addresses -- is otherwise computed once in front of this loop and kept in registers through all
        //  of it: 18 of them spilled to scratch memory)
        int tid = tid0;
        asm volatile("" : "+v"(tid));
        const int lane = tid & 63, w = tid >> 6;
        wc_sync();
        if (s_sp == 0 || s_stop) break;
        const Job job = stack[s_sp - 1];
        wc_sync();
        if (tid == 0) --s_sp;
        if (job.hi - job.lo <= 0) continue;
        const CellGeom g = cell_setup(sh, job, rg, region, prefix, rs, tmin, tmax, tmin2, tmax2, tid, 0);
        cell_seed(sh, g, T, tid);
        wc_sync();
        double vmax = -INFINITY, vmin = INFINITY, d2 = -INFINITY, d3 = INFINITY;
        cell_search<0>(sh, g, rs, eps2, INFINITY, -INFINITY, nullptr, 0, 1, vmax, vmin, wins, evals, tid, true);
        block_minmax4(vmax, vmin, d2, d3, tid);
        if (sh.lost) {                              // a cell queue overflowed: the general path (its overflow leads to the exact scan)
            if (tid == 0) { atomicOr(&counters[6], 32); s_stop = 1; }
            continue;
        }
        if (fmax(fabs(vmax), fabs(vmin)) + eps < thr) continue;              // no call in this range (k_seg_classify's test)
        // the windows within 2 eps of the extremes, only for a side that can hold a call
        const double hi_cut = !(vmax + eps < thr) ? vmax - eps2 : INFINITY;
        const double lo_cut = !(-vmin + eps < thr) ? vmin + eps2 : -INFINITY;
        if (sh.n_rec[0] > CJ_REC || sh.n_rec[1] > CJ_REC) {
            // more near-extreme windows than the record holds (ties): the range once more with the final cuts
            wc_sync();
            if (tid == 0) { sh.n_rec[0] = 0; sh.n_rec[1] = 0; }
            double e0 = -INFINITY, e1 = INFINITY;
            cell_search<2>(sh, g, rs, eps2, hi_cut, lo_cut, nullptr, 0, 1, e0, e1, wins, evals, tid);
            wc_sync();
            if (sh.n_rec[0] > CJ_REC || sh.n_rec[1] > CJ_REC) {       // massive ties: the general path evaluates everything exactly
                if (tid == 0) { atomicOr(&counters[6], 4); s_stop = 1; }
                continue;
            }
        }
        wc_sync();                            // the search is done with its staging area: the waves' scratch now
        // exact values of the candidates (numpy pairwise sum / sqrt): waves 0-1 the candidates for the maximum,
        // 2-3 for the minimum (the record also holds windows that were near an earlier cut: the final cuts select)
        {
            BestPair b;
            b.maxv = 0.0; b.minv = 0.0; b.mx = b.my = b.nx = b.ny = -1;
            const int which = w >> 1, slot = w & 1;
            const int n_mine = sh.n_rec[which];
            int taken = 0;
            for (int t = 0; t < n_mine; ++t) {
                const CellRec r = sh.rec[which][t];
                if (which == 0 ? !(r.v >= hi_cut) : !(r.v <= lo_cut)) continue;
                if ((taken++ & 1) != slot) continue;
                const double v = window_exact_wave(zz, r.x, r.y, lane, wm, sc[w]);
                if (which == 0) {
                    if (better_max(v, r.x, r.y, b.maxv, b.mx, b.my)) { b.maxv = v; b.mx = r.x; b.my = r.y; }
                } else {
                    if (better_min(v, r.x, r.y, b.minv, b.nx, b.ny)) { b.minv = v; b.nx = r.x; b.ny = r.y; }
                }
            }
            if (lane == 0) s_best[w] = b;
        }
        wc_sync();
        if (tid == 0) {
            BestPair b = s_best[0];
            if (s_best[1].mx >= 0 && better_max(s_best[1].maxv, s_best[1].mx, s_best[1].my, b.maxv, b.mx, b.my)) {
                b.maxv = s_best[1].maxv; b.mx = s_best[1].mx; b.my = s_best[1].my;
            }
            b.minv = s_best[2].minv; b.nx = s_best[2].nx; b.ny = s_best[2].ny;
            if (s_best[3].nx >= 0 && better_min(s_best[3].minv, s_best[3].nx, s_best[3].ny, b.minv, b.nx, b.ny)) {
                b.minv = s_best[3].minv; b.nx = s_best[3].nx; b.ny = s_best[3].ny;
            }
            double champ = b.maxv;
            int cx = b.mx, cy = b.my;
            if (fabs(b.minv) > champ) { champ = b.minv; cx = b.nx; cy = b.ny; }
            if (!(fabs(champ) < thr)) {
                if (s_nseg < TREE_SEGS) { seg_val[s_nseg] = champ; seg_x[s_nseg] = cx; seg_y[s_nseg] = cy; ++s_nseg; }
                else { atomicOr(&counters[6], 8); s_stop = 1; }
                const int xr = cx - job.lo, yr = cy - job.lo, edge = job.hi - job.lo;
                const bool left = xr > min_search, right = yr + 1 < edge - min_search;
                if (s_sp + (left ? 1 : 0) + (right ? 1 : 0) > WALK_STACK) {
                    atomicOr(&counters[6], 16);
                    s_stop = 1;
                } else {
                    if (right) { Job n; n.region = region; n.lo = cy + 1; n.hi = job.hi; n.pad = 0; stack[s_sp++] = n; }
                    if (left) { Job n; n.region = region; n.lo = job.lo; n.hi = cx; n.pad = 0; stack[s_sp++] = n; }
                }
            }
        }
    }
    // ---- the region's segments, appended to the batch's list for k_walk_rows (the call rows: position order, genomic
    // bounds, effect size); pad = the segment's number within the region | the region's count << 16
    wc_sync();
    const int tid = tid0, lane = tid & 63, w = tid >> 6;
    const int nseg = s_nseg;
    if (tid == 0) {
        out_n[region] = nseg;
        s_sp = nseg ? atomicAdd(&counters[4], nseg) : 0;     // (the stack pointer's slot: the walk is over)
    }
    wc_sync();
    const int base = s_sp;
    for (int sidx = tid; sidx < nseg; sidx += 256) {
        Seg sg;
        sg.val = seg_val[sidx]; sg.region = region; sg.x = seg_x[sidx]; sg.y = seg_y[sidx]; sg.pad = sidx | (nseg << 16);
        if (base + sidx < seg_cap) wsegs[base + sidx] = sg;   // (beyond: the caller sees counters[4] > seg_cap)
    }
    if (work) {
        for (int o = 32; o > 0; o >>= 1) { evals += __shfl_xor(evals, o); wins += __shfl_xor(wins, o); }
        const int slot = (int)((blockIdx.x * 7u + (unsigned)w) & 63u);
        if (lane == 0) {
            atomicAdd(work + 2 * slot, (unsigned long long)wins);
            atomicAdd(work + 2 * slot + 1, (unsigned long long)evals);
        }
    }
}

// The call rows of k_seg_walk's segments: position order, genomic start / end, value, effect = median of the
// segment's ratios - 1 (np.median; wisecondor.py:233-257) -- k_call_post's work.  A few hundred resident workgroups
// take the segments of the batch's list with a static stride (workgroup w: items w, w + gridDim.x, ...; a workgroup per
// region or per possible segment would mostly be workgroups with nothing to do, and each still has to be given its LDS).
// A kernel of its own: as the tail of k_seg_walk the radix selection returned wrong medians for about one row in a
// thousand, differently from run to run (the selection alone, tools/micro/select_test.hip, is clean), see
// EXPERIMENTS.md.
// (2 304 workgroups, three times what is resident: a workgroup whose items are short is gone in a few microseconds and
//  its slot goes to the next one, so that the long segments -- ~20 us each -- spread over the slots by themselves; with
//  768 some workgroup always held two of them: 52 -> 38 us at 125 x 50 kb, 256 -> 216 us at 1 000 x 50 kb)
constexpr int WALK_ROWS_GRID = 2304;
__global__ __launch_bounds__(256) void k_walk_rows(const Seg *__restrict__ wsegs, const int *__restrict__ n_segs_dev,
                                                   int seg_cap,
                                                   const Region *__restrict__ regions, const double *__restrict__ ratio,
                                                   const int *__restrict__ gpos, int max_calls,
                                                   double *__restrict__ reg_calls, int *__restrict__ hot_count) {
    if (hot_count && blockIdx.x == 0 && threadIdx.x == 0) *hot_count = 0;    // (the walk is over: the next batch's list starts empty)
    // ratios staged in LDS up to STAGED values (40 KB: three workgroups per CU); the counting median up to COUNTED (beyond
    // that its L x L / 256 dependent LDS reads per thread lose to the selection's eight passes).  An item costs ~20 us of
    // dependent round trips and barriers whatever its length: 2 304 workgroups with 16 KB each were slower (75 / 38 us)
    constexpr int STAGED = 5120, COUNTED = 64;
    __shared__ double sv[STAGED];
    __shared__ double s_mid[2];
    __shared__ int s_nan, s_rank;
    const int tid = threadIdx.x;
    const int total = *n_segs_dev < seg_cap ? *n_segs_dev : seg_cap;
    // ---- segments of at most 64 bins (most calls: a noise spike of a bin or two), a WAVE each, no barriers: a lane per
    // ratio, the counting median by lane broadcasts (64 x readlane + two compares), the rank among the region's
    // segments by a ballot.  A workgroup per item spent ~20 us of dependent round trips and barriers on each of them
    // whatever its length (round 5: 53 us for the 1 283 rows of a 125 x 50 kb batch, 171 us for 9 348 rows).
    {
        const int lane = tid & 63;
        const int n_waves = (int)gridDim.x * 4, wave = (int)blockIdx.x * 4 + (tid >> 6);
        for (int item = wave; item < total; item += n_waves) {
            const Seg me = wsegs[item];
            const int x = me.x, y = me.y, Ls = y - x + 1;
            if (Ls > 64) continue;                             // (wave-uniform)
            const int region = me.region, nseg = me.pad >> 16, first = item - (me.pad & 0xFFFF);
            const Region rg = regions[region];
            int rank = 0;
            for (int t0 = 0; t0 < nseg; t0 += 64) {
                const int t = t0 + lane;
                rank += __popcll(__ballot(t < nseg && first + t < total && wsegs[first + t].x < x));
            }
            if (rank >= max_calls) continue;
            const double v = lane < Ls ? ratio[rg.off + x + lane] : 0.0;
            const bool has_nan = __ballot(lane < Ls && v != v) != 0ull;
            const int k_lo = (Ls - 1) / 2, k_hi = Ls / 2;
            int lt = 0, le = 0;
            for (int u = 0; u < Ls; ++u) {
                const double yv = __shfl(v, u);
                lt += yv < v;
                le += yv <= v;
            }
            // the lanes whose count interval covers a middle rank hold that order statistic (equal values: any of them)
            const unsigned long long m_lo = __ballot(lane < Ls && lt <= k_lo && k_lo < le);
            const unsigned long long m_hi = __ballot(lane < Ls && lt <= k_hi && k_hi < le);
            const double lo = __shfl(v, m_lo ? __ffsll((long long)m_lo) - 1 : 0);
            const double hi = __shfl(v, m_hi ? __ffsll((long long)m_hi) - 1 : 0);
            if (lane == 0) {
                double med = (Ls & 1) ? lo : (lo + hi) / 2.0;  // np.median: mean of the middle pair
                if (has_nan) med = NAN;
                const int start = gpos[rg.off + x];
                const int end = (y > x) ? gpos[rg.off + y - 1] + 1 : start;
                double *o = reg_calls + ((int64_t)region * max_calls + rank) * 5;
                o[0] = (double)(rg.pad + 1);
                o[1] = (double)start;
                o[2] = (double)end;
                o[3] = me.val;
                o[4] = med - 1.0;
            }
        }
    }
    // ---- longer segments, a workgroup each
    // (workgroup w takes items w, w + gridDim.x, ...: they cost about the same, and a cursor would put one more global
    //  round trip in front of every item)
    for (int item = blockIdx.x; item < total; item += gridDim.x) {
        const Seg me = wsegs[item];
        if (me.y - me.x + 1 <= 64) continue;                   // (done above)
        wc_sync();
        if (tid == 0) { s_nan = 0; s_rank = 0; }
        wc_sync();
        const int region = me.region, nseg = me.pad >> 16, first = item - (me.pad & 0xFFFF);
        const Region rg = regions[region];
        const int x = me.x, y = me.y, Ls = y - x + 1;
        // (nseg <= TREE_SEGS = 128; a region whose segments were cut off by seg_cap is not read past the stored ones:
        //  the caller sees counters[4] > seg_cap and repeats the batch with a larger list)
        if (tid < nseg && first + tid < total && wsegs[first + tid].x < x) atomicAdd(&s_rank, 1);
        const double *rr = ratio + rg.off + x;
        const bool staged = Ls <= STAGED;
        for (int e0 = tid; e0 < Ls; e0 += 8 * 256) {           // eight loads in flight per thread and trip
            double t8[8];
#pragma unroll
            for (int u = 0; u < 8; ++u) t8[u] = e0 + u * 256 < Ls ? rr[e0 + u * 256] : 0.0;
#pragma unroll
            for (int u = 0; u < 8; ++u) {
                if (staged && e0 + u * 256 < Ls) sv[e0 + u * 256] = t8[u];
                if (t8[u] != t8[u]) s_nan = 1;
            }
        }
        wc_sync();
        const int rank = s_rank;
        if (rank >= max_calls) continue;                   // k_assemble_calls reports the overflow from out_n
        const bool has_nan = s_nan != 0;
        double lo = 0.0, hi = 0.0;
        if (!has_nan && Ls <= COUNTED) {
            // each value counts how many others lie below / at-or-below it; the value whose count interval covers a
            // middle rank is that order statistic (k_call_post's short path)
            const int k_lo = (Ls - 1) / 2, k_hi = Ls / 2;
            for (int e = tid; e < Ls; e += 256) {
                const double xv = sv[e];
                int lt = 0, le = 0;
                for (int u = 0; u < Ls; ++u) {
                    const double yv = sv[u];
                    lt += yv < xv;
                    le += yv <= xv;
                }
                if (lt <= k_lo && k_lo < le) s_mid[0] = xv;      // equal values write the same number
                if (lt <= k_hi && k_hi < le) s_mid[1] = xv;
            }
            wc_sync();
            lo = s_mid[0];
            hi = s_mid[1];
        } else if (!has_nan) {
            // (radix selection: eight passes over the values)
            if (staged) block_select_pair<256>([&](int e) { return sv[e]; }, Ls, (Ls - 1) / 2, Ls / 2, tid, lo, hi);
            else block_select_pair<256>([&](int e) { return rr[e]; }, Ls, (Ls - 1) / 2, Ls / 2, tid, lo, hi);
        }
        if (tid == 0) {
            double med = (Ls & 1) ? lo : (lo + hi) / 2.0;  // np.median: mean of the middle pair
            if (has_nan) med = NAN;
            // the end walk of the reference restarts at `start` and re-counts it:
            // end = position(survivor y-1) + 1, or start itself when y == x
            const int start = gpos[rg.off + x];
            const int end = (y > x) ? gpos[rg.off + y - 1] + 1 : start;
            double *o = reg_calls + ((int64_t)region * max_calls + rank) * 5;
            o[0] = (double)(rg.pad + 1);
            o[1] = (double)start;
            o[2] = (double)end;
            o[3] = me.val;
            o[4] = med - 1.0;
        }
    }
}

// ------------------------------------------------------------ host drivers ----
int run_prepare(wc_ctx *ctx, const wc_reference *ref, const int *counts_dev, int64_t Ns, hipStream_t stream) {
    TestState &ts = ctx->ts;
    int rc;
    if ((rc = ts.totals.reserve(sizeof(long long) * Ns * TOT_SPLIT))) return rc;
    if ((rc = ts.raw.reserve(sizeof(double) * Ns * ref->B))) return rc;
    if ((rc = ts.proj.reserve(sizeof(double) * Ns * MAX_COMP * PROJ_SPLIT))) return rc;
    if ((rc = ts.data.reserve(sizeof(double) * Ns * ref->B))) return rc;
    hipLaunchKernelGGL(k_sample_totals, dim3((unsigned)Ns, TOT_SPLIT), dim3(256), 0, stream, counts_dev, ref->Btot,
                       ts.totals.as<long long>());
    dim3 g((unsigned)cdiv(ref->B, 256), (unsigned)Ns);
    hipLaunchKernelGGL(k_normalize, g, dim3(256), 0, stream, counts_dev, ref->Btot, (const int *)ref->m2g.as<int>(),
                       ref->B, (const long long *)ts.totals.as<long long>(), ts.raw.as<double>());
    hipLaunchKernelGGL(k_pca_project, dim3((unsigned)Ns, PROJ_SPLIT), dim3(256), 0, stream, (const double *)ts.raw.as<double>(),
                       ref->B, (const double *)ref->pca_mean.as<double>(), (const double *)ref->pca_comp.as<double>(),
                       ref->n_comp, ts.proj.as<double>());
    hipLaunchKernelGGL(k_pca_apply, g, dim3(256), 0, stream, (const double *)ts.raw.as<double>(), ref->B,
                       (const double *)ref->pca_mean.as<double>(), (const double *)ref->pca_comp.as<double>(),
                       ref->n_comp, (const double *)ts.proj.as<double>(), ts.data.as<double>());
    WC_HIP(hipGetLastError());
    return WC_OK;
}

// [R, C] -> [C, R] for three arrays in one launch (blockIdx.z picks the array)
__global__ void k_transpose3(const double *__restrict__ in0, const double *__restrict__ in1,
                             const double *__restrict__ in2, int64_t R, int64_t C, double *__restrict__ out0,
                             double *__restrict__ out1, double *__restrict__ out2) {
    __shared__ double tile[32][33];
    const double *in = blockIdx.z == 0 ? in0 : blockIdx.z == 1 ? in1 : in2;
    double *out = blockIdx.z == 0 ? out0 : blockIdx.z == 1 ? out1 : out2;
    int64_t c0 = (int64_t)blockIdx.x * 32, r0 = (int64_t)blockIdx.y * 32;
    for (int j = threadIdx.y; j < 32; j += 8) {
        int64_t r = r0 + j, c = c0 + threadIdx.x;
        if (r < R && c < C) tile[j][threadIdx.x] = in[r * C + c];
    }
    wc_sync();
    for (int j = threadIdx.y; j < 32; j += 8) {
        int64_t c = c0 + j, r = r0 + threadIdx.x;
        if (r < R && c < C) out[c * R + r] = tile[threadIdx.x][j];
    }
}

void launch_transpose(const double *in, int64_t R, int64_t C, double *out, hipStream_t stream,
                      double *out2 = nullptr, int *zero = nullptr, int64_t n_zero = 0) {
    dim3 g((unsigned)cdiv(C, 32), (unsigned)cdiv(R, 32));
    hipLaunchKernelGGL(k_transpose, g, dim3(32, 8), 0, stream, in, R, C, out, out2, zero, n_zero);
}

// The bin-major working arrays of the repeats, [B, Ns] + one more row (xc's row B holds -1.0, see k_zscore_tiled).  ONE
// place sizes them: the batch body calls this BEFORE its prepare kernels write xt / xc, run_repeat again (a no-op then).
static int reserve_repeat_arrays(TestState &ts, int64_t n, int64_t Ns) {
    int rc;
    for (wc::DevBuf *b : {&ts.xt, &ts.xc, &ts.zt, &ts.rt, &ts.nt, &ts.sdt})
        if ((rc = b->reserve(sizeof(double) * (n + Ns)))) return rc;
    return ts.sd_avg.reserve(sizeof(double) * Ns);
}

// repeatTest on device data [Ns, B]; leaves zt/rt/nt/sdt as [B, Ns] and sd_avg[Ns]
// lat: latency mode -- k_lat_prepare has already written xt / xc and cleared the counters; the first
// repeat runs as usual, repeats 2.. in one launch (k_lat_repeats; its overflow flag is pair_counts[repeats + 1])
// allow_sm: the caller reads z / ratio / counts / sd per SAMPLE and takes them in either layout: when the first repeat
// runs the tiled kernel they are written sample-major [Ns, B] straight away (ts.sm_out tells which it was)
int run_repeat(wc_ctx *ctx, const wc_reference *ref, const double *data_dev, int64_t Ns, double thr, int repeats,
               hipStream_t stream, bool lat = false, double *asdef_out = nullptr, bool xt_ready = false,
               bool allow_sm = false, bool allow_tail = false) {
    TestState &ts = ctx->ts;
    const int64_t n = ref->B * Ns;
    int rc;
    const char *tiled_env = getenv("WC_ZSCORE_TILED");       // "0": the untiled kernel (a wave = one bin x 64 samples)
    const char *sm_env = getenv("WC_ZSCORE_SM");             // "0": bin-major outputs + the transposes (round 5's form)
    // (fewer than eight tiles would leave XCDs idle or without a column of their own: the untiled kernel)
    const bool tiled = !lat && repeats >= 1 && ref->k <= 128 && Ns >= 128 && (Ns & 15) == 0 &&
                       (ref->B + 1) * Ns * 8 < ((int64_t)1 << 32) && ref->B < (1 << 24) && Ns * 8 < (1 << 24) &&
                       !(tiled_env && tiled_env[0] == '0');
    // (a list stride above 100 -- refsize 101 .. 128 -- needs 104 value slots: with the output permutation that form
    //  spills six registers, so those references keep the bin-major outputs and the transposes)
    const bool sm_out = tiled && allow_sm && ref->k <= 100 && !(sm_env && sm_env[0] == '0');
    const int64_t osm = sm_out ? ref->B : 0;
    ts.sm_out = sm_out;
    // a caller whose prepare kernel has ALREADY written xt / xc must have sized them with reserve_repeat_arrays: a
    // reserve that grows a buffer frees it (DevBuf keeps no contents) and the z-scores would read fresh memory
    if (lat || xt_ready)
        WC_CHECK(ts.xt.bytes >= sizeof(double) * (size_t)(n + Ns) && ts.xc.bytes >= sizeof(double) * (size_t)(n + Ns),
                 WC_E_INTERNAL, "repeatTest: the prepared arrays are smaller than the repeats need");
    if ((rc = reserve_repeat_arrays(ts, n, Ns))) return rc;
    WC_CHECK(n < (1ll << 32), WC_E_LIMIT, "repeatTest: more than 2^32 (bin, sample) pairs per call");
    // xt = data^T, xc = its working copy (flags go in there); in the same launch the repeats'
    // pair counts and the `dirty` bitmap (one bit per pair: queued for the next repeat) are cleared
    const int64_t n_words = cdiv(n, 32);
    if ((rc = ts.misc2.reserve(sizeof(int) * (repeats + 2 + n_words)))) return rc;
    if (repeats > 1)
        for (wc::DevBuf *b : {&ts.pairs_a, &ts.pairs_b})
            if ((rc = b->reserve(sizeof(unsigned int) * n))) return rc;
    int *pair_counts = ts.misc2.as<int>();                       // [repeats + 2]: pairs queued for repeat it
    unsigned int *dirty = (unsigned int *)(pair_counts + repeats + 2);
    if (!lat && !xt_ready)       // (xt_ready: the caller's prepare kernel wrote xt / xc and cleared the words)
        launch_transpose(data_dev, Ns, ref->B, ts.xt.as<double>(), stream, ts.xc.as<double>(), pair_counts,
                         repeats > 0 ? repeats + 2 + n_words : 0);
    const unsigned g = (unsigned)cdiv(n, 256);
    if (repeats < 1) {  // the reference would return None; give NaNs
        WC_HIP(hipMemsetAsync(ts.zt.p, 0xFF, sizeof(double) * n, stream));
        WC_HIP(hipMemsetAsync(ts.rt.p, 0xFF, sizeof(double) * n, stream));
        WC_HIP(hipMemsetAsync(ts.nt.p, 0, sizeof(double) * n, stream));
        WC_HIP(hipMemsetAsync(ts.sdt.p, 0xFF, sizeof(double) * n, stream));
    }
    const int *uoff = ref->users_off.as<int>(), *ulst = ref->users.as<int>();
    // batches whose caller can repeat them (allow_tail): repeats 3 .. by one workgroup in one launch
    ts.tail_used = false;
    ts.tail_flag = nullptr;
    const bool tail_ok = allow_tail && !lat && !ts.no_tail && repeats > 2 && ref->k <= 128 &&
                         !(getenv("WC_TEST_TAIL_REPEATS") && getenv("WC_TEST_TAIL_REPEATS")[0] == '0');
    // latency mode, a sample or two: the first repeat's flag pass runs at the head of k_lat_repeats
    const bool lat_flag_inside = lat && repeats > 1 && ref->k <= 128 && n <= 16384;
    for (int it = 0; it < repeats; ++it) {
        if (lat && it == 1) {
            hipLaunchKernelGGL(k_lat_repeats, dim3(1), dim3(1024), 0, stream, ts.pairs_a.as<unsigned int>(),
                               ts.pairs_b.as<unsigned int>(), pair_counts, repeats, dirty,
                               (const double *)ts.xt.as<double>(), ts.xc.as<double>(), (const int *)ref->gidx.as<int>(),
                               (const int *)ref->nref.as<int>(), ref->k, Ns, thr, uoff, ulst, ts.zt.as<double>(),
                               ts.rt.as<double>(), ts.nt.as<double>(), ts.sdt.as<double>(), pair_counts + repeats + 1,
                               lat_flag_inside ? (int)n : 0);
            break;
        }
        // repeat `it` recomputes the pairs queued in its list (all pairs in the first repeat) and
        // queues, for repeat it + 1, the pairs that see one of its new flags
        unsigned int *cur = (it & 1) ? ts.pairs_a.as<unsigned int>() : ts.pairs_b.as<unsigned int>();
        unsigned int *next = it + 1 < repeats ? ((it & 1) ? ts.pairs_b.as<unsigned int>() : ts.pairs_a.as<unsigned int>())
                                              : nullptr;
        if (ref->k > 128) {
            // long reference lists: the generic wave-per-pair kernel for every repeat
            const int64_t np = it == 0 ? n : 0;
            const unsigned gb = (unsigned)std::min<int64_t>(cdiv(n, 4), 16384);
            hipLaunchKernelGGL(k_zscore_big, dim3(gb), dim3(256), 0, stream,
                               it == 0 ? (const unsigned int *)nullptr : (const unsigned int *)cur,
                               (const int *)(pair_counts + it), np, dirty, (const double *)ts.xt.as<double>(),
                               (const double *)ts.xc.as<double>(), (const int *)ref->gidx.as<int>(),
                               (const int *)ref->nref.as<int>(), ref->k, Ns, ts.zt.as<double>(), ts.rt.as<double>(),
                               ts.nt.as<double>(), ts.sdt.as<double>());
            if (it == 0)
                hipLaunchKernelGGL(k_flag, dim3(g), dim3(256), 0, stream, (const double *)ts.zt.as<double>(), thr, n, Ns,
                                   ts.xc.as<double>(), uoff, ulst, dirty, next, pair_counts + it + 1);
            else
                hipLaunchKernelGGL(k_flag_pairs, dim3((unsigned)std::min<int64_t>(g, 2048)), dim3(256), 0, stream,
                                   (const unsigned int *)cur, (const int *)(pair_counts + it),
                                   (const double *)ts.zt.as<double>(), thr, Ns, ts.xc.as<double>(), uoff, ulst, dirty,
                                   next, pair_counts + it + 1, (int64_t)0);
        } else if (it == 0) {
            if (tiled) {
                constexpr int ZT_G = 12;
                constexpr int zwaves = 1;                           // (one wave per workgroup: a slot is refilled as soon as it frees; 2 and 4 measured 1-2 % slower)
                const int64_t zgroups = cdiv(ref->B, 4 * zwaves), ztiles = Ns / 16;
                const int64_t n_wg = 8 * ((ztiles / 8) * zgroups + cdiv((ztiles % 8) * zgroups, 8));
                if (!xt_ready)                                      // (the batch's prepare kernel, k_pca_apply_t, wrote it)
                hipLaunchKernelGGL(k_fill, dim3((unsigned)cdiv(Ns, 256)), dim3(256), 0, stream, ts.xc.as<double>() + ref->B * Ns,
                                   Ns, -1.0);                       // row B of xc: what an index of -1 reads
                // (a list stride of up to 100 -- refsize 100 -- needs 100 value slots, not the 104 of 8 G + 8)
                auto zk = sm_out ? k_zscore_tiled<ZT_G, zwaves, 25, true>
                                 : (ref->k <= 100 ? k_zscore_tiled<ZT_G, zwaves, 25, false> : k_zscore_tiled<ZT_G, zwaves, 26, false>);
                hipLaunchKernelGGL(zk, dim3((unsigned)n_wg), dim3(64 * zwaves), 0, stream,
                                   (const double *)ts.xt.as<double>(), (const double *)ts.xc.as<double>(),
                                   (const int *)ref->gidx.as<int>(), (const int *)ref->nref.as<int>(), ref->k,
                                   (int)ref->B, (int)Ns, ts.zt.as<double>(), ts.rt.as<double>(), ts.nt.as<double>(),
                                   ts.sdt.as<double>(), thr, sm_out && repeats > 1 ? cur : (unsigned int *)nullptr, pair_counts + 0);
            } else if (Ns >= 32) {
                const unsigned n_uni = (unsigned)cdiv(ref->B * cdiv(Ns, 64), 4);
                hipLaunchKernelGGL(k_zscore, dim3(n_uni), dim3(256), 0, stream,
                                   (const double *)ts.xt.as<double>(), (const double *)ts.xc.as<double>(),
                                   (const int *)ref->gidx.as<int>(), (const int *)ref->nref.as<int>(), ref->k, ref->B,
                                   Ns, ts.zt.as<double>(), ts.rt.as<double>(), ts.nt.as<double>(), ts.sdt.as<double>());
            } else {
                hipLaunchKernelGGL(k_zscore_pairs, dim3((unsigned)std::min<int64_t>(cdiv(n, 32), 8192)), dim3(256), 0,
                                   stream, (const unsigned int *)nullptr, (const int *)nullptr, n, dirty,
                                   (const double *)ts.xt.as<double>(), (const double *)ts.xc.as<double>(),
                                   (const int *)ref->gidx.as<int>(), (const int *)ref->nref.as<int>(), ref->k, Ns,
                                   ts.zt.as<double>(), ts.rt.as<double>(), ts.nt.as<double>(), ts.sdt.as<double>(), (int64_t)0);
            }
            if (sm_out && repeats > 1)
                // the tiled kernel listed the pairs that reach the threshold (in `cur`, counted in pair_counts[0])
                hipLaunchKernelGGL(k_flag_hits, dim3(2048), dim3(256), 0, stream,
                                   (const unsigned int *)cur, (const int *)(pair_counts + 0),
                                   (const double *)ts.zt.as<double>(), thr, Ns, ts.xc.as<double>(), uoff, ulst, dirty, next,
                                   pair_counts + it + 1, osm);
            else if (sm_out)
                hipLaunchKernelGGL(k_flag_sm, dim3((unsigned)cdiv(ref->B, 256), (unsigned)Ns), dim3(256), 0, stream,
                                   (const double *)ts.zt.as<double>(), thr, ref->B, Ns, ts.xc.as<double>(), uoff, ulst, dirty,
                                   next, pair_counts + it + 1);
            else if (!lat_flag_inside)
                hipLaunchKernelGGL(k_flag, dim3(g), dim3(256), 0, stream, (const double *)ts.zt.as<double>(), thr, n, Ns,
                                   ts.xc.as<double>(), uoff, ulst, dirty, next, pair_counts + it + 1);
        } else if (it == 2 && tail_ok) {
            // repeats 3 .. in ONE launch of one workgroup (see k_lat_repeats); its overflow word reaches the caller through
            // the segmentation's set-up kernel (ts.tail_flag -> misc[1])
            ts.tail_used = true;
            ts.tail_flag = pair_counts + repeats + 1;
            hipLaunchKernelGGL(k_lat_repeats, dim3(1), dim3(1024), 0, stream, ts.pairs_a.as<unsigned int>(),
                               ts.pairs_b.as<unsigned int>(), pair_counts, repeats, dirty,
                               (const double *)ts.xt.as<double>(), ts.xc.as<double>(), (const int *)ref->gidx.as<int>(),
                               (const int *)ref->nref.as<int>(), ref->k, Ns, thr, uoff, ulst, ts.zt.as<double>(),
                               ts.rt.as<double>(), ts.nt.as<double>(), ts.sdt.as<double>(), pair_counts + repeats + 1,
                               0, 2, osm, getenv("WC_TEST_TAIL_CAP") ? atoi(getenv("WC_TEST_TAIL_CAP")) : TAIL_PAIR_CAP);   // (the test suite forces the overflow with a cap of 0)
            break;
        } else {
            // repeat 2 recomputes the users of the first repeat's flags (10^3 .. 10^5 pairs); repeats 3 .. are a few
            // hundred pairs at most -- their time is the dispatch of workgroups that find nothing to do: an eighth of the
            // grid (both kernels stride over their list, so a long list is only walked in more trips)
            const unsigned gp = (unsigned)std::min<int64_t>(g, it >= 2 ? 256 : 2048);
            hipLaunchKernelGGL(k_zscore_pairs, dim3(gp), dim3(256), 0, stream, (const unsigned int *)cur,
                               (const int *)(pair_counts + it), (int64_t)0, dirty, (const double *)ts.xt.as<double>(),
                               (const double *)ts.xc.as<double>(), (const int *)ref->gidx.as<int>(),
                               (const int *)ref->nref.as<int>(), ref->k, Ns, ts.zt.as<double>(), ts.rt.as<double>(),
                               ts.nt.as<double>(), ts.sdt.as<double>(), osm);
            hipLaunchKernelGGL(k_flag_pairs, dim3(gp), dim3(256), 0, stream, (const unsigned int *)cur,
                               (const int *)(pair_counts + it), (const double *)ts.zt.as<double>(), thr, Ns,
                               ts.xc.as<double>(), uoff, ulst, dirty, next, pair_counts + it + 1, osm);
        }
    }
    // stdDevAvg only feeds the asdef output: it runs on the context's side stream under the
    // segmentation work (latency mode: on the launch stream -- the parallel form takes a few
    // microseconds, a second stream in the captured graph costs more -- straight into `asdef_out`).
    ts.lat_ride = false;
    if (lat && ref->B <= 65536) {
        // latency mode: the parallel form rides in k_seg_tree's grid (run_seg_lat) -- no second stream,
        // whose fork and join cost more than the kernel; a sample it gives up on raises status word 33
        if ((rc = ts.sd_fail.reserve(sizeof(int) * Ns))) return rc;
        ts.lat_ride = true;
        ts.lat_ride_out2 = asdef_out;
        return WC_OK;
    }
    hipStream_t sds = stream;
    {
        if ((rc = ctx->ensure_side_stream())) return rc;
        WC_HIP(hipEventRecord(ctx->ev_fork, stream));
        WC_HIP(hipStreamWaitEvent(ctx->side, ctx->ev_fork, 0));
        sds = ctx->side;
    }
    // the parallel form first (exact, see k_sd_fast); the serial chain only for samples it gave up on
    const int *only = nullptr;
    double *out2 = lat ? asdef_out : nullptr;
    if (ref->B <= 65536) {
        if ((rc = ts.sd_fail.reserve(sizeof(int) * Ns))) return rc;
        if (sm_out) {
            // the first repeat wrote the standard deviations sample-major already
            launch_sd_fast(sds, ts.sdt.as<double>(), ref->B, Ns, ts.sd_avg.as<double>(), ts.sd_fail.as<int>(), out2, 1,
                           ref->B);
        } else if (Ns > 8) {
            // a batch: the sums run over a sample-major copy (a sample's standard deviations contiguous): in
            // the bin-major array every element of a sample sits in a cache line of its own, and 125
            // workgroups walking 55 337 such lines three times kept the side stream busy for 0.4 ms
            if ((rc = ts.sds.reserve(sizeof(double) * n))) return rc;
            launch_transpose((const double *)ts.sdt.as<double>(), ref->B, Ns, ts.sds.as<double>(), sds);
            launch_sd_fast(sds, ts.sds.as<double>(), ref->B, Ns, ts.sd_avg.as<double>(), ts.sd_fail.as<int>(), out2, 1,
                           ref->B);
        } else {
            launch_sd_fast(sds, ts.sdt.as<double>(), ref->B, Ns, ts.sd_avg.as<double>(), ts.sd_fail.as<int>(), out2, Ns, 1);
        }
        only = ts.sd_fail.as<int>();
    }
    if (lat && !only) {        // the status word reads the flags: none raised
        if ((rc = ts.sd_fail.reserve(sizeof(int) * Ns))) return rc;
        WC_HIP(hipMemsetAsync(ts.sd_fail.p, 0, sizeof(int) * Ns, sds));
    }
    if (lat && only) {
        // latency mode: a sample the parallel form gave up on sends the call to the general path
        // (status word 33) instead of costing every call a launch
    } else if (Ns <= 16)
        hipLaunchKernelGGL(k_sd_avg<16>, dim3((unsigned)cdiv(Ns, 16)), dim3(256), 0, sds,
                           (const double *)ts.sdt.as<double>(), ref->B, Ns, ts.sd_avg.as<double>(), only, out2, Ns, (int64_t)1);
    else
        hipLaunchKernelGGL(k_sd_avg<64>, dim3((unsigned)cdiv(Ns, 64)), dim3(256), 0, sds,
                           (const double *)ts.sdt.as<double>(), ref->B, Ns, ts.sd_avg.as<double>(), only, out2,
                           sm_out ? (int64_t)1 : Ns, sm_out ? ref->B : (int64_t)1);
    WC_HIP(hipEventRecord(ctx->ev_join, ctx->side));
    ctx->side_pending = true;
    WC_HIP(hipGetLastError());
    return WC_OK;
}

// Output-only work of a batch (stdDevAvg, the inflated result arrays, the whole-region values) runs on the
// context's side stream under the segmentation: side_begin makes the side stream wait for what `stream` has
// enqueued so far, side_end marks the point join_side waits for.
// second: the SECOND side stream.  In a big batch stdDevAvg (k_sd_fast: a 1 024-thread workgroup per sample) is as long
// as the whole segmentation -- 1.8 ms of a 1 000 x 50 kb batch -- and the inflated outputs and whole-region values
// behind it on one stream ended 0.5 ms after the launch stream (kernel trace, round 6): they get a stream of their own.
static bool side_second(int64_t n_samples) {
    static const int from = getenv("WC_TEST_SIDE2") ? atoi(getenv("WC_TEST_SIDE2")) : 768;   // (0: never; from 256 on: 256 x 50 kb and 512 x 250 kb are 1 % slower with it)
    return from > 0 && n_samples >= from;
}
static int side_begin(wc_ctx *ctx, hipStream_t stream, bool second = false) {
    int rc;
    if ((rc = ctx->ensure_side_stream())) return rc;
    hipStream_t s = second ? ctx->side2 : ctx->side;
    if (ctx->side_fresh) {             // forked a moment ago, nothing enqueued on `stream` since: one fork serves both
        ctx->side_fresh = false;       // (an event record on the launch stream costs it ~10 us: kernel trace, round 6)
        if (second) WC_HIP(hipStreamWaitEvent(s, ctx->ev_fork, 0));
        return WC_OK;
    }
    WC_HIP(hipEventRecord(ctx->ev_fork, stream));
    WC_HIP(hipStreamWaitEvent(s, ctx->ev_fork, 0));
    return WC_OK;
}
static int side_end(wc_ctx *ctx, bool second = false) {
    if (second) {
        WC_HIP(hipEventRecord(ctx->ev_join2, ctx->side2));
        ctx->side2_pending = true;
        return WC_OK;
    }
    WC_HIP(hipEventRecord(ctx->ev_join, ctx->side));
    ctx->side_pending = true;
    return WC_OK;
}

// Make `stream` wait for the side-stream work of run_repeat (before sd_avg is consumed).
int join_side(wc_ctx *ctx, hipStream_t stream) {
    ctx->side_fresh = false;
    if (ctx->side_pending) {
        WC_HIP(hipStreamWaitEvent(stream, ctx->ev_join, 0));
        ctx->side_pending = false;
    }
    if (ctx->side2_pending) {
        WC_HIP(hipStreamWaitEvent(stream, ctx->ev_join2, 0));
        ctx->side2_pending = false;
    }
    return WC_OK;
}

// Segment search over device regions.  Results: ts.out_val/out_x/out_y [n_regions, max_calls],
// ts.out_n [n_regions], ts.whole [n_regions].
// lat_rounds > 0 (latency mode, small batches): no host round trips -- that many search rounds are
// enqueued with grids sized for the most jobs a round can hold (every job has at most two children)
// and the kernels read the real job / segment counts on the device; *lat_incomplete (pinned, valid
// after the stream has been synchronised) tells whether jobs were left over or a bound was
// exceeded, in which case the caller runs the call again with the host-driven loop.
// `tail` (optional; the batched `test` call): where call rows go.  When the regions fit the tree kernel
// (<= TREE_MAXLEN bins, no -mineffectsize mask) the first round's hot regions are walked to the end by
// k_seg_tree, one workgroup per region -- collect, decide, every child range, the order of the
// segments and the call rows -- instead of one host-driven round per recursion level: rounds two and
// later are a handful of short ranges each and cost a launch series and a count read-back apiece.
// ts.tree_done tells the caller that ts.effect / ts.out_n already hold the calls.
struct TreeTail {
    const double *ratio;     // cleaned ratios, as z_dev
    const int *gpos;         // genomic position of every kept bin
    double *reg_calls;       // [n_regions, max_calls, 5]
    bool defer_status;       // the caller reads the tree kernel's status words after its own synchronize
    double *cwz_out;         // where the whole-region values go besides ts.whole (written on the side stream), or NULL
    int per_sample = 0;      // regions per sample (sample-major region list), or 0: k_seg_walk then takes every sample's
                             // first region first -- chromosome 1, the longest -- so that the launch ends on short ones
};
// Batches whose regions fit the fused set-up kernel (<= TREE_MAXLEN bins, no -mineffectsize mask): cleaning,
// prefix sums, whole-region values and the root jobs in ONE launch (k_lat_setup<256>) instead of k_clean +
// k_region_prefix + k_region_whole + k_init_jobs + the copy of the whole-region values.
struct FusedSetup {
    const double *zsrc, *rsrc, *nsrc;
    int64_t str_i, str_b, B;
    const int64_t *moff, *goff;
    const int *m2g, *sel;
    int n_sel;
    double minref;
    double *zc, *rc;
    int *gpos;
    Region *regions;
    double *whole_copy;
};
int run_stouffer(wc_ctx *ctx, const double *z_dev, const Region *regions_dev, int64_t n_regions, int64_t total_len,
                 int64_t max_n, double thr, int min_search, int max_calls, hipStream_t stream,
                 const double *ratio_dev = nullptr, double min_effect = 0.0, int64_t bits_upper = 0,
                 int lat_rounds = 0, double *whole_copy = nullptr, const TreeTail *tail = nullptr,
                 const FusedSetup *fused = nullptr) {
    TestState &ts = ctx->ts;
    int rc;
    ts.last_segs = 0;
    ts.tree_done = false;
    ts.tree_pending = false;
    if (n_regions == 0) return WC_OK;
    const int64_t job_cap = n_regions + total_len / 4 + 64;
    const int64_t seg_cap = n_regions * (int64_t)max_calls + 64;
    const int max_chunks = (int)std::max<int64_t>(1, cdiv((max_n + 1) / 2, ROWS_HALF));
    if ((rc = ts.prefix.reserve(sizeof(double) * (total_len + n_regions + 8)))) return rc;
    if ((rc = ts.reg_abs.reserve(sizeof(double) * n_regions))) return rc;
    if ((rc = ts.reg_flag.reserve(sizeof(int) * n_regions))) return rc;
    const int64_t rs_need = std::max<int64_t>(max_n + 80 + QB2, 2 * QB + 80 + QB2);   // the search reads four lengths at a time (the candidate scan one trip ahead); the bound scan min len + 128
                                                                        // past the longest window, the certificate 1..64
    if (ts.rs_len < rs_need) {
        if ((rc = ts.rs.reserve(sizeof(double) * rs_need))) return rc;
        ts.rs_len = rs_need;
        hipLaunchKernelGGL(k_fill_rs, dim3((unsigned)cdiv(ts.rs_len, 256)), dim3(256), 0, stream, ts.rs.as<double>(),
                           ts.rs_len);
    }
    if ((rc = ts.jobs_a.reserve(sizeof(Job) * job_cap))) return rc;
    if ((rc = ts.jobs_b.reserve(sizeof(Job) * job_cap))) return rc;
    if ((rc = ts.job_cnt.reserve(sizeof(int) * 16))) return rc;
    if ((rc = ts.job_res.reserve(sizeof(Extreme) * job_cap))) return rc;
    if ((rc = ts.hot.reserve(sizeof(int) * 2 * job_cap))) return rc;
    if ((rc = ts.seg.reserve(sizeof(Seg) * seg_cap))) return rc;
    if ((rc = ts.out_val.reserve(sizeof(double) * n_regions * max_calls))) return rc;
    if ((rc = ts.out_x.reserve(sizeof(int) * n_regions * max_calls))) return rc;
    if ((rc = ts.out_y.reserve(sizeof(int) * n_regions * max_calls))) return rc;
    if ((rc = ts.out_n.reserve(sizeof(int) * n_regions))) return rc;
    if ((rc = ts.whole.reserve(sizeof(double) * n_regions))) return rc;

    // -mineffectsize: one validity bit per window (wisetools.py:479-487)
    const unsigned int *bits = nullptr;
    const long long *bit_off = nullptr;
    if (min_effect != 0.0) {
        WC_CHECK(ratio_dev, WC_E_ARG, "segments: mineffectsize needs the ratio vector");
        const int64_t words = bits_upper / 32 + 2;
        if ((rc = ts.win_bits.reserve(sizeof(unsigned int) * words))) return rc;
        if ((rc = ts.bit_off.reserve(sizeof(long long) * n_regions))) return rc;
        WC_HIP(hipMemsetAsync(ts.win_bits.p, 0, sizeof(unsigned int) * words, stream));
        hipLaunchKernelGGL(k_bit_offsets, dim3(1), dim3(1), 0, stream, regions_dev, n_regions, ts.bit_off.as<long long>());
        const char *me = getenv("WC_MINEFFECT");         // "sorted": the O(n) per window sorted-insert kernel
        if (me && strcmp(me, "sorted") == 0) {
            WC_CHECK(max_n * 8 <= 64 * 1024 - 1024, WC_E_LIMIT, "segments: region too long for the sorted-insert median filter");
            hipLaunchKernelGGL(k_window_valid, dim3((unsigned)max_n, (unsigned)n_regions), dim3(64),
                               sizeof(double) * (max_n + 1), stream, ratio_dev, regions_dev,
                               (const long long *)ts.bit_off.as<long long>(), min_effect, ts.win_bits.as<unsigned int>());
        } else {
            // u_hi: the smallest double >= 1 with fabs(u - 1.0) >= min_effect; u_lo: the largest <= 1 -- found
            // with the kernel's own expression by bisection over the ordered bit patterns (monotone on each side)
            auto passes = [&](double u) { return fabs(u - 1.0) >= min_effect; };
            double u_hi = NAN, u_lo = NAN;
            if (passes(INFINITY)) {
                uint64_t lo = wc::f64_ordered(1.0), hi = wc::f64_ordered(INFINITY);     // passes(hi) holds
                if (passes(1.0)) hi = lo;
                while (lo < hi) {
                    const uint64_t mid = lo + ((hi - lo) >> 1);
                    if (passes(wc::f64_from_ordered(mid))) hi = mid; else lo = mid + 1;
                }
                u_hi = wc::f64_from_ordered(hi);
            }
            if (passes(-INFINITY)) {
                uint64_t lo = wc::f64_ordered(-INFINITY), hi = wc::f64_ordered(1.0);     // passes(lo) holds
                if (passes(1.0)) lo = hi;
                while (lo < hi) {
                    const uint64_t mid = lo + ((hi - lo + 1) >> 1);
                    if (passes(wc::f64_from_ordered(mid))) lo = mid; else hi = mid - 1;
                }
                u_lo = wc::f64_from_ordered(lo);
            }
            hipLaunchKernelGGL(k_window_valid_count, dim3((unsigned)cdiv(max_n, 256), (unsigned)n_regions), dim3(256), 0,
                               stream, ratio_dev, regions_dev, (const long long *)ts.bit_off.as<long long>(), min_effect,
                               u_hi, u_lo, ts.win_bits.as<unsigned int>());
        }
        bits = ts.win_bits.as<unsigned int>();
        bit_off = ts.bit_off.as<long long>();
    }
    unsigned long long *work = nullptr;     // profiling: evaluation counters of the search kernels
    if (ts.profile) {
        if ((rc = ts.prof_work.reserve(sizeof(unsigned long long) * 128))) return rc;      // 64 pairs {windows, bounds}
        WC_HIP(hipMemsetAsync(ts.prof_work.p, 0, sizeof(unsigned long long) * 128, stream));
        work = ts.prof_work.as<unsigned long long>();
    }
    int *counters = ts.job_cnt.as<int>();  // [1] next jobs [2] hot [3] brute [4] segments
    int *hot = ts.hot.as<int>();
    int *brute = hot + job_cap;
    if ((rc = ts.job_cnt.reserve(sizeof(int) * 16))) return rc;
    if (fused)
        hipLaunchKernelGGL(k_lat_setup<256>, dim3((unsigned)n_regions), dim3(256), sizeof(double) * (max_n + 1), stream,
                           fused->zsrc, fused->rsrc, fused->nsrc, fused->str_i, fused->str_b, fused->B, fused->moff,
                           fused->goff, fused->m2g, fused->sel, fused->n_sel, fused->minref, fused->zc, fused->rc,
                           fused->gpos, fused->regions, ts.prefix.as<double>(), ts.reg_abs.as<double>(),
                           ts.reg_flag.as<int>(), ts.whole.as<double>(), fused->whole_copy, ts.jobs_a.as<Job>(), counters,
                           ts.out_n.as<int>(), (ts.job_cnt.as<int>() + 8), n_regions, ts.tail_used ? (const int *)ts.tail_flag : (const int *)nullptr);
    // Callers with call rows, regions up to CJ_MAXLEN bins, no -mineffectsize mask: the whole recursion of every region
    // in ONE launch (k_seg_walk), no host round trip.  WC_TEST_WALK=0: the paths it replaces (tree kernel up to
    // TREE_MAXLEN, host-driven rounds beyond).
    const char *walk_env = getenv("WC_TEST_WALK");
    const bool walk_path = tail && !bits && max_n <= CJ_MAXLEN && !ts.no_tree && !(walk_env && walk_env[0] == '0');
    const int64_t total = total_len + n_regions, nblk = cdiv(total, QB);
    if ((rc = ts.tmin.reserve(sizeof(double) * nblk))) return rc;
    if ((rc = ts.tmax.reserve(sizeof(double) * nblk))) return rc;
    const int64_t nblk2 = cdiv(nblk, 4);
    if ((rc = ts.tmin2.reserve(sizeof(double) * nblk2))) return rc;
    if ((rc = ts.tmax2.reserve(sizeof(double) * nblk2))) return rc;
    auto block_tables = [&]() {
        hipLaunchKernelGGL(k_block_minmax, dim3((unsigned)cdiv(total, 1024)), dim3(256), 0, stream,
                           (const double *)ts.prefix.as<double>(), total, ts.tmin.as<double>(), ts.tmax.as<double>(),
                           ts.tmin2.as<double>(), ts.tmax2.as<double>());
    };
    // (measured and not kept: ONE fork of the side stream behind k_clean for stdDevAvg + inflated outputs + whole-region
    //  values -- k_clean is as long without k_sd_fast beside it, k_block_minmax twice as long with it: 1.15 against 1.115 ms)
    // (prefix sums AND block tables by one 256-thread workgroup per region -- a thread per 16 consecutive bins, the tables
    //  from the values it had just written -- measured 125 us against 49 + 42 for these two launches: not kept)
    // (a 256-thread workgroup per region -- 1 024 coalesced bins per trip, four wave scans, one barrier -- measured 53 us
    //  against this kernel's 49 at 125 x 50 kb: what these set-up launches wait for is the half of the chip k_sd_fast's
    //  1 024-thread workgroups hold on the side stream, not their own parallelism)
    // k_seg_walk's early starters (WalkHot): listed by k_region_prefix; WC_TEST_WALK_HOT=0 switches them off
    WalkHot whot{nullptr, nullptr, nullptr, 0, 0.0, ts.tail_used ? ts.tail_flag : nullptr};
    if (walk_path && !fused && !(getenv("WC_TEST_WALK_HOT") && getenv("WC_TEST_WALK_HOT")[0] == '0')) {
        const int cap = n_regions >= 4096 ? 4096 : 512;
        if ((rc = ts.walk_hot.reserve(sizeof(int) * (16 + 4096 + n_regions)))) return rc;
        if (ts.walk_hot.p != ts.walk_hot_clean) {
            WC_HIP(hipMemsetAsync(ts.walk_hot.p, 0, sizeof(int) * 16, stream));
            ts.walk_hot_clean = ts.walk_hot.p;
        }
        whot.count = ts.walk_hot.as<int>();
        whot.list = whot.count + 16;
        whot.index = whot.count + 16 + 4096;
        whot.cap = cap;
        whot.cut = 0.75 * thr;
    }
    if (!fused)
    hipLaunchKernelGGL(k_region_prefix, dim3((unsigned)cdiv(n_regions, 4)), dim3(256), 0, stream, z_dev, regions_dev,
                       n_regions, ts.prefix.as<double>(), ts.reg_abs.as<double>(), ts.reg_flag.as<int>(), counters,
                       ts.out_n.as<int>(), (ts.job_cnt.as<int>() + 8), whot);
    // The block tables: the walker's cell search, the quiet-job certificate (k_seg_quiet; measured -7 % per 250 kb batch
    // and -17 % per 50 kb batch on data where 10-40 % of the regions hold a call) and the bound-driven rounds read them
    block_tables();
    const bool certify = true;         // (the certificate runs before every search round of the tree / masked paths)
    // whole-region values (an output) and the root jobs of the host-driven rounds.  The walker needs no job list:
    // k_init_jobs stays off its path.  (The whole-region values by the walker's own workgroups -- its last wave before
    // the walk starts -- were measured: k_seg_walk 265 -> 385 us at 125 x 50 kb, a 4 700-bin pairwise tree walked by
    // one wave costs ~40 us; they stay a side-stream launch.)
    bool whole_second = false;
    auto whole_and_jobs = [&](const bool jobs_too) -> int {
        // the whole-region values are an output only -> side stream (one wave per region walks
        // the region in numpy's order: 0.1 ms at 50 kb that the search does not have to wait for)
        hipStream_t ws = stream;
        double *wcopy = whole_copy;
        int rc2;
        if (tail && tail->cwz_out && !bits) {
            // (a second side stream for this launch alone was measured: it then runs beside the walk -- 180 us instead
            //  of 105, the walk 277 instead of 267, 1 000 x 50 kb 7.69 instead of 7.34 ms; one side stream)
            //  (big batches: the second side stream, see side_second)
            const bool second = side_second(tail->per_sample > 0 ? n_regions / tail->per_sample : 0);
            if ((rc2 = side_begin(ctx, stream, second))) return rc2;
            ws = second ? ctx->side2 : ctx->side;
            wcopy = tail->cwz_out;
            whole_second = second;
        }
        hipLaunchKernelGGL(k_region_whole, dim3((unsigned)cdiv(n_regions, 4)), dim3(256), 0, ws, z_dev, regions_dev,
                           n_regions, bits, bit_off, ts.whole.as<double>(), wcopy);
        if (ws != stream && (rc2 = side_end(ctx, whole_second))) return rc2;
        if (jobs_too)
            hipLaunchKernelGGL(k_init_jobs, dim3((unsigned)cdiv(n_regions, 256)), dim3(256), 0, stream, regions_dev, n_regions,
                               ts.jobs_a.as<Job>(), counters);
        return WC_OK;
    };
    const bool walker_owns_setup = walk_path && !fused;
    if (!fused && (rc = whole_and_jobs(!walker_owns_setup))) return rc;
    Job *cur = ts.jobs_a.as<Job>(), *next = ts.jobs_b.as<Job>();
    int64_t n_jobs = n_regions;
    int guard = 0;
    // Rounds that the tree kernel does not take over (regions beyond TREE_MAXLEN, callers without call rows) and
    // that carry no -mineffectsize mask locate the extremes from the block bounds (k_seg_bound / k_seg_refine /
    // k_seg_bcollect) instead of evaluating every window of every job that may hold a call.
    const char *tree_env = getenv("WC_TEST_TREE_TAIL");        // "0": host-driven rounds only
    const bool tree_ok0 = tail && !bits && max_n <= TREE_MAXLEN && !ts.no_tree && !(tree_env && tree_env[0] == '0');
    const char *cells_env = getenv("WC_TEST_CELLS");
    if ((rc = ctx->ensure_pinned(256))) return rc;
    int *h = (int *)ctx->pinned;          // counter read-backs land in pinned memory
    h[4] = 0;
    if (walk_path) {
        ts.mark(10, stream);
        hipLaunchKernelGGL(k_seg_walk, dim3((unsigned)(n_regions + whot.cap)), dim3(256), 0, stream, counters, regions_dev,
                           (int)n_regions, (const int *)ts.reg_flag.as<int>(), (const double *)ts.prefix.as<double>(),
                           (const double *)ts.rs.as<double>(), (const double *)ts.reg_abs.as<double>(), z_dev, thr,
                           min_search, (const double *)ts.tmin.as<double>(), (const double *)ts.tmax.as<double>(),
                           (const double *)ts.tmin2.as<double>(), (const double *)ts.tmax2.as<double>(), ts.seg.as<Seg>(),
                           (int)seg_cap, ts.out_n.as<int>(), work,
                           (tail->per_sample > 1 && n_regions % tail->per_sample == 0) ? tail->per_sample : 0,   // (125 x 50 kb: 287 -> 255 us)
                           whot);
        hipLaunchKernelGGL(k_walk_rows, dim3(WALK_ROWS_GRID), dim3(256), 0, stream, (const Seg *)ts.seg.as<Seg>(),
                           (const int *)(counters + 4), (int)seg_cap, regions_dev, tail->ratio, tail->gpos,
                           max_calls, tail->reg_calls, whot.count);
        ts.mark(11, stream);
        const int64_t bound = seg_cap;
        if (!tail->defer_status) WC_HIP(hipMemcpyAsync(h, counters, sizeof(int) * 8, hipMemcpyDeviceToHost, stream));
        if (tail->defer_status) {
            // (the caller's ONE copy at the end of the batch brings the counters and the flag words behind them)
            // the caller looks at the counters after ITS synchronize: h[6] non-zero = the walk gave up on some
            // region, h[4] beyond the bound = more segments than k_call_post's grid covers; it then repeats the
            // batch with the host-driven rounds
            ts.tree_done = true;
            ts.tree_pending = true;
            ts.tree_seg_cap = bound;
            WC_HIP(hipGetLastError());
            return WC_OK;
        }
        WC_HIP(hipStreamSynchronize(stream));
        if (h[6] == 0 && h[4] <= bound) {
            ts.tree_done = true;
            WC_HIP(hipGetLastError());
            return WC_OK;
        }
        // rare: again on the host-driven path
        if (getenv("WC_TEST_VERBOSE"))
            fprintf(stderr, "wisecondor_amd: segmentation repeated on the host-driven rounds (walk status %d: 1 = region beyond %d "
                            "bins, 2 = non-finite z, 4 = ties beyond the record, 8 = more than %d segments, 16 = stack; %d segments)\n",
                    h[6], CJ_MAXLEN, TREE_SEGS, h[4]);
        WC_HIP(hipMemsetAsync(ts.out_n.p, 0, sizeof(int) * n_regions, stream));
        WC_HIP(hipMemsetAsync(counters + 4, 0, sizeof(int), stream));
        WC_HIP(hipMemsetAsync(counters + 6, 0, sizeof(int), stream));
        h[4] = 0;
        if (walker_owns_setup)             // the root jobs the walker did not need
            hipLaunchKernelGGL(k_init_jobs, dim3((unsigned)cdiv(n_regions, 256)), dim3(256), 0, stream, regions_dev, n_regions,
                               ts.jobs_a.as<Job>(), counters);
    }
    const bool tree_ok = tree_ok0 && (walk_env && walk_env[0] == '0');
    const bool bound_path = !bits && !tree_ok;
    while (n_jobs > 0) {
        WC_CHECK(++guard < 100000, WC_E_INTERNAL, "stouffer: recursion did not terminate");
        // per-round scratch is sized by the jobs of this round, not by the worst case
        if ((rc = ts.partial.reserve(sizeof(Extreme) * n_jobs * max_chunks))) return rc;
        if ((rc = ts.sub.reserve(sizeof(double2) * 8 * n_jobs * max_chunks))) return rc;
        if ((rc = ts.cand.reserve(sizeof(int2) * 2 * CAND_CAP * n_jobs))) return rc;
        if ((rc = ts.cand_cnt.reserve(sizeof(int) * 2 * n_jobs))) return rc;
        // round counters are reset by the search kernel, candidate counts by classify
        dim3 sg((unsigned)max_chunks, (unsigned)n_jobs);
        ts.mark(10, stream);
        // about 16 384 workgroups in all: one per job when there are many jobs, every row block of a job in
        // parallel when there are few
        const unsigned per_job = (unsigned)std::min<int64_t>(max_chunks, std::max<int64_t>(1, 16384 / n_jobs));
        // (the certificate stages a job's block table per workgroup: half as many, each with two row blocks, measured
        // 75 -> 61 us at 128 x 250 kb; the bound sweep loses with fewer)
        const unsigned per_job_quiet = (unsigned)std::min<int64_t>(max_chunks, std::max<int64_t>(1, 8192 / n_jobs));
        // jobs up to CJ_MAXLEN bins: one workgroup per job finds the extremes from cell bounds and lists the candidates
        // (k_seg_job); WC_TEST_CELLS=0 keeps the row-block kernels (k_seg_seed / k_seg_bound / k_seg_bcollect)
        const bool cell_path = bound_path && max_n <= CJ_MAXLEN && !(cells_env && cells_env[0] == '0');
        if (cell_path) {
            WC_HIP(hipMemsetAsync(counters + 1, 0, sizeof(int) * 3, stream));      // next jobs, hot, brute
            if ((rc = ts.cell_state.reserve(sizeof(CellJobState) * n_jobs))) return rc;
            if ((rc = ts.cell_rec.reserve(sizeof(CellRec) * 2 * CJ_GREC * n_jobs))) return rc;
            WC_HIP(hipMemsetAsync(ts.cell_state.p, 0, sizeof(CellJobState) * n_jobs, stream));
            // a round of many jobs fills the chip with one workgroup per job (parts repeat the set-up and the seed);
            // a round of few jobs -- the later rounds -- is as long as its longest job: several workgroups per job
            const char *parts_env = getenv("WC_CELL_PARTS");               // experiments: at most this many workgroups per job
            const int max_parts = std::max(1, std::min(cell_parts((int)max_n), parts_env ? atoi(parts_env) : (n_jobs >= 2048 ? 1 : CJ_MAXPARTS)));
            hipLaunchKernelGGL(k_seg_job, dim3((unsigned)max_parts, (unsigned)n_jobs), dim3(256), 0, stream,
                               (const Job *)cur, (int)n_jobs,
                               regions_dev, (const double *)ts.prefix.as<double>(), (const double *)ts.rs.as<double>(),
                               (const double *)ts.reg_abs.as<double>(), (const int *)ts.reg_flag.as<int>(), thr,
                               (const double *)ts.tmin.as<double>(), (const double *)ts.tmax.as<double>(),
                               (const double *)ts.tmin2.as<double>(), (const double *)ts.tmax2.as<double>(),
                               ts.cell_state.as<CellJobState>(), ts.cell_rec.as<CellRec>(), work);
            hipLaunchKernelGGL(k_seg_merge, dim3((unsigned)n_jobs), dim3(256), 0, stream, (const Job *)cur, (int)n_jobs,
                               regions_dev, (const double *)ts.prefix.as<double>(), (const double *)ts.rs.as<double>(),
                               (const double *)ts.reg_abs.as<double>(), (const int *)ts.reg_flag.as<int>(), thr,
                               (const double *)ts.tmin.as<double>(), (const double *)ts.tmax.as<double>(),
                               (const double *)ts.tmin2.as<double>(), (const double *)ts.tmax2.as<double>(),
                               (const CellJobState *)ts.cell_state.as<CellJobState>(),
                               (const CellRec *)ts.cell_rec.as<CellRec>(), hot, brute, counters, ts.cand.as<int2>(),
                               ts.cand_cnt.as<int>(), work);
            ts.mark(11, stream);
            hipLaunchKernelGGL(k_seg_decide, dim3((unsigned)n_jobs), dim3(256), 0, stream, (const Job *)cur,
                               (const int *)hot, counters, regions_dev, z_dev, (const int2 *)ts.cand.as<int2>(),
                               (const int *)ts.cand_cnt.as<int>(), thr, min_search, bits, bit_off, ts.seg.as<Seg>(),
                               (int)seg_cap, next, (int)job_cap, brute, counters + 1);
            hipLaunchKernelGGL(k_seg_brute, dim3((unsigned)n_jobs), dim3(256), 0, stream, (const Job *)cur,
                               (const int *)brute, counters, regions_dev, z_dev, thr, min_search, bits, bit_off,
                               ts.seg.as<Seg>(), (int)seg_cap, next, (int)job_cap, counters + 1);
            WC_HIP(hipMemcpyAsync(h, counters, sizeof(int) * 8, hipMemcpyDeviceToHost, stream));
            WC_HIP(hipStreamSynchronize(stream));
            WC_CHECK(h[1] <= job_cap, WC_E_INTERNAL, "stouffer: job list overflow");
            WC_CHECK(h[4] <= seg_cap, WC_E_LIMIT, "stouffer: more than max_calls=%d segments per region", max_calls);
            n_jobs = h[1];
            std::swap(cur, next);
            continue;
        }
        if (bound_path) {
            if ((rc = ts.cbound.reserve(sizeof(ChunkBound) * n_jobs * max_chunks))) return rc;
            if ((rc = ts.cuts.reserve(sizeof(unsigned long long) * 2 * n_jobs))) return rc;
            hipLaunchKernelGGL(k_seg_seed, dim3((unsigned)n_jobs), dim3(256), 0, stream, (const Job *)cur, (int)n_jobs,
                               regions_dev, (const double *)ts.rs.as<double>(), (const double *)ts.reg_abs.as<double>(),
                               (const int *)ts.reg_flag.as<int>(), thr, (const double *)ts.tmin.as<double>(),
                               (const double *)ts.tmax.as<double>(), ts.cuts.as<unsigned long long>());
            hipLaunchKernelGGL(k_seg_bound, dim3(per_job, (unsigned)n_jobs), dim3(256), 0, stream, (const Job *)cur, (int)n_jobs,
                               regions_dev, (const double *)ts.prefix.as<double>(), (const double *)ts.rs.as<double>(),
                               (const int *)ts.reg_flag.as<int>(), (const double *)ts.tmin.as<double>(),
                               (const double *)ts.tmax.as<double>(), (const double *)ts.tmin2.as<double>(),
                               (const double *)ts.tmax2.as<double>(), max_chunks, ts.partial.as<Extreme>(),
                               ts.cbound.as<ChunkBound>(), ts.cuts.as<unsigned long long>(), counters, counters + 1, work);
        } else {
        if (certify) {
            // about 16 384 workgroups in all: one per job when there are many jobs, every row block
            // of a job in parallel when there are few
            hipLaunchKernelGGL(k_seg_quiet, dim3(per_job_quiet, (unsigned)n_jobs), dim3(256), 0, stream, cur, (int)n_jobs, regions_dev,
                               (const double *)ts.prefix.as<double>(), (const double *)ts.rs.as<double>(),
                               (const double *)ts.reg_abs.as<double>(), (const int *)ts.reg_flag.as<int>(), thr,
                               (const double *)ts.tmin.as<double>(), (const double *)ts.tmax.as<double>(), work,
                               (const int *)nullptr, (int)n_regions);
        }
        {
            const bool plds = max_n + 1 <= 6144;      // the longest region's prefix slice fits 48 KB of LDS
            const size_t dyn = plds ? sizeof(double) * (max_n + 1) : 0;
#define WC_SEARCH(M, P_, NW_)                                                                                     \
    hipLaunchKernelGGL((k_seg_search<M, P_, NW_>), sg, dim3(64 * NW_), dyn, stream, (const Job *)cur, (int)n_jobs,   \
                       regions_dev, (const double *)ts.prefix.as<double>(), (const double *)ts.rs.as<double>(),     \
                       (const int *)ts.reg_flag.as<int>(), max_chunks, bits, bit_off, ts.partial.as<Extreme>(), \
                       counters, (int)certify, ts.sub.as<double2>(), work, (const int *)nullptr, counters + 1)
#define WC_SEARCH_NW(M, P_) do { if (wide) WC_SEARCH(M, P_, 16); else WC_SEARCH(M, P_, 4); } while (0)
            const bool wide = n_jobs * max_chunks <= 2048;     // few blocks: sixteen waves each
            if (bits) { if (plds) WC_SEARCH_NW(true, true); else WC_SEARCH_NW(true, false); }
            else {
                // unmasked rounds get here only on the tree path (regions <= TREE_MAXLEN): the slice always fits
                WC_CHECK(plds, WC_E_INTERNAL, "stouffer: unmasked value search beyond the LDS-staged sizes");
                WC_SEARCH_NW(false, true);
            }
#undef WC_SEARCH_NW
#undef WC_SEARCH
        }
        }   // !bound_path
        ts.mark(11, stream);
        hipLaunchKernelGGL(k_seg_classify, dim3((unsigned)n_jobs), dim3(64), 0, stream, (const Job *)cur, (int)n_jobs,
                           regions_dev, (const double *)ts.reg_abs.as<double>(), (const int *)ts.reg_flag.as<int>(),
                           (const Extreme *)ts.partial.as<Extreme>(), max_chunks, thr, ts.job_res.as<Extreme>(), hot,
                           brute, counters, ts.cand_cnt.as<int>(), (int)(certify && !bound_path), (const int *)nullptr);
        // The number of hot jobs lives on the device.  Small rounds (latency mode, child
        // ranges) launch the follow-up kernels for the upper bound n_jobs and let surplus
        // workgroups exit, which saves a host round trip; big rounds read the count back.
        int n_hot = (int)n_jobs;
        if (n_jobs * max_chunks > 65536) {
            WC_HIP(hipMemcpyAsync(h, counters, sizeof(int) * 8, hipMemcpyDeviceToHost, stream));
            WC_HIP(hipStreamSynchronize(stream));
            n_hot = h[2];
        }
        if (guard == 1 && tree_ok && n_hot > 0) {
            SdRider no_sd{};
            InflateRider no_inf{};
            hipLaunchKernelGGL(k_seg_tree, dim3((unsigned)n_hot), dim3(1024), sizeof(double) * (2 * max_n + 2), stream,
                               counters, regions_dev, n_regions, (const int *)ts.reg_flag.as<int>(),
                               (const double *)ts.prefix.as<double>(), (const double *)ts.rs.as<double>(),
                               (const double *)ts.reg_abs.as<double>(), z_dev, tail->ratio, tail->gpos, thr, min_search,
                               max_calls, tail->reg_calls, ts.out_n.as<int>(), (const Extreme *)ts.partial.as<Extreme>(),
                               (const double2 *)ts.sub.as<double2>(), max_chunks, no_sd, no_inf,
                               (const int *)hot, (const int *)(counters + 2));
            WC_HIP(hipMemcpyAsync(h, counters, sizeof(int) * 8, hipMemcpyDeviceToHost, stream));
            if (tail->defer_status) {
                // the caller looks at the counters after ITS synchronize (one host round trip per batch
                // instead of two): h[3] / h[6] non-zero = the tree kernel passed on some region, and the
                // caller repeats the batch with host-driven rounds
                ts.tree_done = true;
                ts.tree_pending = true;
                ts.tree_seg_cap = seg_cap;
                WC_HIP(hipGetLastError());
                return WC_OK;
            }
            WC_HIP(hipStreamSynchronize(stream));
            if (h[3] == 0 && h[6] == 0) {          // no job for the exact scan, nothing the tree kernel gave up on
                WC_CHECK(h[4] <= seg_cap, WC_E_LIMIT, "stouffer: more than max_calls=%d segments per region", max_calls);
                ts.tree_done = true;
                WC_HIP(hipGetLastError());
                return WC_OK;
            }
            // rare: start the round's second half again on the host-driven path
            WC_HIP(hipMemsetAsync(ts.out_n.p, 0, sizeof(int) * n_regions, stream));
            WC_HIP(hipMemsetAsync(counters + 4, 0, sizeof(int), stream));
            WC_HIP(hipMemsetAsync(counters + 6, 0, sizeof(int), stream));
        }
        if (n_hot > 0 && bound_path) {
            const unsigned per_hot = (unsigned)std::min<int64_t>(max_chunks, std::max<int64_t>(1, 16384 / n_hot));
            hipLaunchKernelGGL(k_seg_bcollect, dim3(per_hot, (unsigned)n_hot), dim3(256), 0, stream, (const Job *)cur,
                               (const int *)hot, (const int *)counters, regions_dev, (const double *)ts.prefix.as<double>(),
                               (const double *)ts.rs.as<double>(), (const double *)ts.reg_abs.as<double>(), thr,
                               (const Extreme *)ts.job_res.as<Extreme>(), (const double *)ts.tmin.as<double>(),
                               (const double *)ts.tmax.as<double>(), (const double *)ts.tmin2.as<double>(),
                               (const double *)ts.tmax2.as<double>(), max_chunks, (const ChunkBound *)ts.cbound.as<ChunkBound>(),
                               ts.cand.as<int2>(), ts.cand_cnt.as<int>(), work);
        } else if (n_hot > 0) {
            // only a few blocks survive the pruning; sixteen waves each keep their scan short
            hipLaunchKernelGGL(k_seg_collect, dim3((unsigned)max_chunks, (unsigned)n_hot), dim3(1024), 0, stream,
                               (const Job *)cur, (const int *)hot, (const int *)counters, regions_dev,
                               (const double *)ts.prefix.as<double>(), (const double *)ts.rs.as<double>(),
                               (const double *)ts.reg_abs.as<double>(), (const Extreme *)ts.job_res.as<Extreme>(),
                               (const Extreme *)ts.partial.as<Extreme>(), max_chunks,
                               (const double2 *)ts.sub.as<double2>(), bits, bit_off, ts.cand.as<int2>(),
                               ts.cand_cnt.as<int>());
        }
        if (n_hot > 0) {
            hipLaunchKernelGGL(k_seg_decide, dim3((unsigned)n_hot), dim3(256), 0, stream, (const Job *)cur,
                               (const int *)hot, counters, regions_dev, z_dev, (const int2 *)ts.cand.as<int2>(),
                               (const int *)ts.cand_cnt.as<int>(), thr, min_search, bits, bit_off, ts.seg.as<Seg>(),
                               (int)seg_cap, next, (int)job_cap, brute, counters + 1);
        }
        // brute list may have grown in decide; its length is only known on the device
        hipLaunchKernelGGL(k_seg_brute, dim3((unsigned)n_jobs), dim3(256), 0, stream, (const Job *)cur,
                           (const int *)brute, counters, regions_dev, z_dev, thr, min_search, bits, bit_off,
                           ts.seg.as<Seg>(), (int)seg_cap, next, (int)job_cap, counters + 1);
        WC_HIP(hipMemcpyAsync(h, counters, sizeof(int) * 8, hipMemcpyDeviceToHost, stream));
        WC_HIP(hipStreamSynchronize(stream));
        WC_CHECK(h[1] <= job_cap, WC_E_INTERNAL, "stouffer: job list overflow");
        WC_CHECK(h[4] <= seg_cap, WC_E_LIMIT, "stouffer: more than max_calls=%d segments per region", max_calls);
        n_jobs = h[1];
        std::swap(cur, next);
    }
    ts.last_segs = h[4];                  // from the last round's read-back
    if (h[4] > 0)
        hipLaunchKernelGGL(k_seg_gather, dim3((unsigned)cdiv(h[4], 256)), dim3(256), 0, stream,
                           (const Seg *)ts.seg.as<Seg>(), h[4], max_calls, ts.out_val.as<double>(), ts.out_x.as<int>(),
                           ts.out_y.as<int>(), ts.out_n.as<int>(), (const int *)nullptr);
    WC_HIP(hipGetLastError());
    return WC_OK;
}

// Latency mode's cleaning + segmentation: k_lat_setup, k_seg_search (root windows of every region, all
// row blocks in parallel), k_seg_tree (one workgroup per region walks the recursion and writes the call
// rows; stdDevAvg and the result inflation ride in its grid) -- no host round trip; counters[4]
// (segments) and [6] (give-up flag) are looked at by the caller after it synchronised.  The calls are
// left in ts.effect / ts.out_n, the whole-chromosome values in ts.whole.
int run_seg_lat(wc_ctx *ctx, const wc_reference *ref, const double *zsrc, const double *rsrc, const double *nsrc,
                int64_t str_i, int64_t str_b, int64_t Ns, int n_sel, int64_t max_n, double thr, int min_ref_bins,
                int max_calls, hipStream_t stream, double *whole_copy, InflateRider inf) {
    TestState &ts = ctx->ts;
    const int64_t B = ref->B, n_regions = Ns * n_sel, total_len = Ns * B;
    int rc;
    const int64_t seg_cap = n_regions * (int64_t)max_calls + 64;
    if ((rc = ts.prefix.reserve(sizeof(double) * (total_len + n_regions + 8)))) return rc;
    if ((rc = ts.reg_abs.reserve(sizeof(double) * n_regions))) return rc;
    if ((rc = ts.reg_flag.reserve(sizeof(int) * n_regions))) return rc;
    const int64_t rs_need = std::max<int64_t>(max_n + 80, 2 * QB + 80);
    if (ts.rs_len < rs_need) {
        if ((rc = ts.rs.reserve(sizeof(double) * rs_need))) return rc;
        ts.rs_len = rs_need;
        hipLaunchKernelGGL(k_fill_rs, dim3((unsigned)cdiv(ts.rs_len, 256)), dim3(256), 0, stream, ts.rs.as<double>(),
                           ts.rs_len);
    }
    if ((rc = ts.jobs_a.reserve(sizeof(Job) * (n_regions + 64)))) return rc;
    if ((rc = ts.job_cnt.reserve(sizeof(int) * 16))) return rc;
    if ((rc = ts.seg.reserve(sizeof(Seg) * seg_cap))) return rc;
    if ((rc = ts.out_val.reserve(sizeof(double) * n_regions * max_calls))) return rc;
    if ((rc = ts.out_x.reserve(sizeof(int) * n_regions * max_calls))) return rc;
    if ((rc = ts.out_y.reserve(sizeof(int) * n_regions * max_calls))) return rc;
    if ((rc = ts.out_n.reserve(sizeof(int) * n_regions))) return rc;
    if ((rc = ts.whole.reserve(sizeof(double) * n_regions))) return rc;
    if ((rc = ts.job_cnt.reserve(sizeof(int) * 16))) return rc;
    const int64_t total = total_len + n_regions, nblk = cdiv(total, QB);
    if ((rc = ts.tmin.reserve(sizeof(double) * nblk))) return rc;
    if ((rc = ts.tmax.reserve(sizeof(double) * nblk))) return rc;
    int *counters = ts.job_cnt.as<int>();
    hipLaunchKernelGGL(k_lat_setup<1024>, dim3((unsigned)n_regions), dim3(1024), sizeof(double) * (max_n + 1), stream, zsrc, rsrc, nsrc, str_i, str_b, B,
                       (const int64_t *)ref->moff_dev.as<int64_t>(), (const int64_t *)ref->goff_dev.as<int64_t>(),
                       (const int *)ref->m2g.as<int>(), (const int *)ts.sel.as<int>(), n_sel, (double)min_ref_bins,
                       ts.zc.as<double>(), ts.rc.as<double>(), ts.gpos.as<int>(), ts.regions.as<Region>(),
                       ts.prefix.as<double>(), ts.reg_abs.as<double>(), ts.reg_flag.as<int>(), ts.whole.as<double>(),
                       whole_copy, ts.jobs_a.as<Job>(), counters, ts.out_n.as<int>(), (ts.job_cnt.as<int>() + 8), n_regions);
    // the regions' value search with all row blocks in parallel (the general path's kernel, no
    // certificate); the tree kernel starts from its per-block extremes
    const int max_chunks = (int)std::max<int64_t>(1, cdiv((max_n + 1) / 2, ROWS_HALF));
    if ((rc = ts.partial.reserve(sizeof(Extreme) * n_regions * max_chunks))) return rc;
    if ((rc = ts.sub.reserve(sizeof(double2) * 8 * n_regions * max_chunks))) return rc;
    {
        const dim3 sg((unsigned)max_chunks, (unsigned)n_regions);
        const size_t dyn = sizeof(double) * (max_n + 1);
#define WC_LSEARCH(NW_)                                                                                              \
    hipLaunchKernelGGL((k_seg_search<false, true, NW_>), sg, dim3(64 * NW_), dyn, stream,                            \
                       (const Job *)ts.jobs_a.as<Job>(), (int)n_regions, (const Region *)ts.regions.as<Region>(),    \
                       (const double *)ts.prefix.as<double>(), (const double *)ts.rs.as<double>(),                   \
                       (const int *)ts.reg_flag.as<int>(), max_chunks, (const unsigned int *)nullptr,                \
                       (const long long *)nullptr, ts.partial.as<Extreme>(), counters, 0, ts.sub.as<double2>(),      \
                       (unsigned long long *)nullptr, (const int *)nullptr, counters + 5)
        if (n_regions * max_chunks <= 2048) WC_LSEARCH(16); else WC_LSEARCH(4);
#undef WC_LSEARCH
    }
    SdRider rider{};
    if (ts.lat_ride) {
        rider.blocks = (int)Ns;
        rider.sdT = ts.sdt.as<double>();
        rider.B = B;
        rider.Ns = Ns;
        rider.out = ts.sd_avg.as<double>();
        rider.out2 = ts.lat_ride_out2;
        rider.fail = ts.sd_fail.as<int>();
        rider.sb = Ns;
        rider.si = 1;
    }
    const size_t tree_lds = std::max<size_t>(sizeof(double) * (2 * max_n + 2), rider.blocks ? sizeof(SdShared) : 0);
    hipLaunchKernelGGL(k_seg_tree, dim3((unsigned)(n_regions + rider.blocks + inf.blocks)), dim3(1024), tree_lds, stream, counters,
                       (const Region *)ts.regions.as<Region>(), n_regions, (const int *)ts.reg_flag.as<int>(),
                       (const double *)ts.prefix.as<double>(), (const double *)ts.rs.as<double>(),
                       (const double *)ts.reg_abs.as<double>(), (const double *)ts.zc.as<double>(),
                       (const double *)ts.rc.as<double>(), (const int *)ts.gpos.as<int>(), thr, 3, max_calls,
                       ts.effect.as<double>(), ts.out_n.as<int>(), (const Extreme *)ts.partial.as<Extreme>(),
                       (const double2 *)ts.sub.as<double2>(), max_chunks, rider, inf, (const int *)nullptr,
                       (const int *)nullptr);
    ts.last_segs = 0;                     // the calls are already in ts.effect / ts.out_n
    WC_HIP(hipGetLastError());
    return WC_OK;
}

}  // namespace

extern "C" {

int wc_optimal_cutoff(wc_ctx *ctx, const double *distances, int64_t count, int repeats, double *cutoff) {
    WC_CHECK(ctx && distances && cutoff && count > 0, WC_E_ARG, "getOptimalCutoff: bad argument");
    WC_HIP(hipSetDevice(ctx->device));
    int rc;
    if ((rc = ctx->tmp_a.reserve(sizeof(double) * count))) return rc;
    WC_HIP(hipMemcpy(ctx->tmp_a.p, distances, sizeof(double) * count, hipMemcpyHostToDevice));
    return device_cutoff(ctx, ctx->tmp_a.as<double>(), count, repeats, nullptr, cutoff);
}

int wc_optimal_cutoff_mask(wc_ctx *ctx, const double *distances, int64_t count, int repeats, double *cutoff,
                           uint8_t *mask) {
    WC_CHECK(ctx && distances && cutoff && mask && count > 0 && repeats > 0, WC_E_ARG,
             "getOptimalCutoff: bad argument");
    WC_HIP(hipSetDevice(ctx->device));
    int rc;
    if ((rc = ctx->tmp_a.reserve(sizeof(double) * count))) return rc;
    if ((rc = ctx->tmp_b.reserve(count))) return rc;
    WC_HIP(hipMemcpy(ctx->tmp_a.p, distances, sizeof(double) * count, hipMemcpyHostToDevice));
    double previous = INFINITY;
    if ((rc = device_cutoff(ctx, ctx->tmp_a.as<double>(), count, repeats, nullptr, cutoff, 1, &previous))) return rc;
    hipLaunchKernelGGL(k_cut_mask, dim3((unsigned)cdiv(count, 256)), dim3(256), 0, nullptr,
                       (const double *)ctx->tmp_a.as<double>(), count, previous, ctx->tmp_b.as<uint8_t>());
    WC_HIP(hipGetLastError());
    WC_HIP(hipMemcpy(mask, ctx->tmp_b.p, count, hipMemcpyDeviceToHost));
    return WC_OK;
}

wc_reference *wc_reference_create(wc_ctx *ctx, const int32_t *indexes, const double *distances, int64_t n_bins,
                                  int k, const int64_t *chromosome_sizes, const int64_t *masked_sizes, int n_chrom,
                                  const uint8_t *mask, const double *pca_mean, const double *pca_components,
                                  int n_comp, int cutoff_repeats, const double *cutoff_override) {
    auto fail = [&](wc_reference *r) -> wc_reference * {
        if (r) wc_reference_destroy(r);
        return nullptr;
    };
    if (!ctx || !indexes || !distances || !chromosome_sizes || !masked_sizes || !mask || !pca_mean ||
        !pca_components) {
        wc::set_error("reference: NULL argument");
        return nullptr;
    }
    if (n_bins <= 0 || k <= 0 || k > BIG_K || n_chrom <= 0 || n_chrom > WC_MAX_CHROM || n_comp < 0 ||
        n_comp > MAX_COMP) {
        wc::set_error("reference: unsupported shape (bins %lld, refsize %d (max 1024), chromosomes %d, components %d)",
                      (long long)n_bins, k, n_chrom, n_comp);
        return nullptr;
    }
    if (hipSetDevice(ctx->device) != hipSuccess) {
        wc::set_error("reference: hipSetDevice failed");
        return nullptr;
    }
    wc_reference *ref = new wc_reference();
    static std::atomic<unsigned long long> next_serial{0};
    ref->serial = ++next_serial;
    ref->ctx = ctx;
    ref->B = n_bins;
    ref->k = k;
    ref->n_chrom = n_chrom;
    ref->n_comp = n_comp;
    for (int c = 0; c < n_chrom; ++c) {
        ref->moff[c + 1] = ref->moff[c] + masked_sizes[c];
        ref->goff[c + 1] = ref->goff[c] + chromosome_sizes[c];
    }
    ref->Btot = ref->goff[n_chrom];
    if (ref->moff[n_chrom] != n_bins) {
        wc::set_error("reference: masked_sizes sum to %lld, indexes have %lld rows", (long long)ref->moff[n_chrom],
                      (long long)n_bins);
        return fail(ref);
    }
    std::vector<int> m2g(n_bins), g2m(ref->Btot, -1);
    int64_t at = 0;
    for (int c = 0; c < n_chrom; ++c) {
        int64_t in_chrom = 0;
        for (int64_t g = ref->goff[c]; g < ref->goff[c + 1]; ++g)
            if (mask[g]) {
                if (at < n_bins) { m2g[at] = (int)g; g2m[g] = (int)at; }
                ++at;
                ++in_chrom;
            }
        if (in_chrom != masked_sizes[c]) {
            wc::set_error("reference: mask has %lld bins on chromosome %d, masked_sizes says %lld", (long long)in_chrom,
                          c + 1, (long long)masked_sizes[c]);
            return fail(ref);
        }
    }
    const int64_t nk = n_bins * k;
    // (+ 448 bytes: k_zscore reads a bin's list by seven unconditional 16-index loads -- 112 indexes from the list's
    // start whatever refsize is; only their USE is guarded -- so the last bin's loads reach up to 112 - k indexes
    // beyond the array)
    bool ok = ref->gidx.reserve(sizeof(int) * nk + 512) == 0 && ref->nref.reserve(sizeof(int) * n_bins) == 0 &&
              ref->pca_mean.reserve(sizeof(double) * n_bins) == 0 &&
              ref->pca_comp.reserve(sizeof(double) * std::max<int64_t>(1, (int64_t)n_comp * n_bins)) == 0 &&
              ref->m2g.reserve(sizeof(int) * n_bins) == 0 && ref->g2m.reserve(sizeof(int) * ref->Btot) == 0 &&
              ref->moff_dev.reserve(sizeof(int64_t) * (WC_MAX_CHROM + 1)) == 0 &&
              ref->goff_dev.reserve(sizeof(int64_t) * (WC_MAX_CHROM + 1)) == 0 &&
              ctx->tmp_a.reserve(sizeof(double) * nk) == 0 && ctx->tmp_b.reserve(sizeof(int) * nk) == 0;
    if (!ok) return fail(ref);
    bool cp = hipMemcpy(ctx->tmp_a.p, distances, sizeof(double) * nk, hipMemcpyHostToDevice) == hipSuccess &&
              hipMemcpy(ctx->tmp_b.p, indexes, sizeof(int) * nk, hipMemcpyHostToDevice) == hipSuccess &&
              hipMemcpy(ref->pca_mean.p, pca_mean, sizeof(double) * n_bins, hipMemcpyHostToDevice) == hipSuccess &&
              (n_comp == 0 || hipMemcpy(ref->pca_comp.p, pca_components, sizeof(double) * n_comp * n_bins,
                                        hipMemcpyHostToDevice) == hipSuccess) &&
              hipMemcpy(ref->m2g.p, m2g.data(), sizeof(int) * n_bins, hipMemcpyHostToDevice) == hipSuccess &&
              hipMemcpy(ref->g2m.p, g2m.data(), sizeof(int) * ref->Btot, hipMemcpyHostToDevice) == hipSuccess &&
              hipMemcpy(ref->moff_dev.p, ref->moff, sizeof(int64_t) * (n_chrom + 1), hipMemcpyHostToDevice) == hipSuccess &&
              hipMemcpy(ref->goff_dev.p, ref->goff, sizeof(int64_t) * (n_chrom + 1), hipMemcpyHostToDevice) == hipSuccess;
    if (!cp) {
        wc::set_error("reference: upload failed: %s", hipGetErrorString(hipGetLastError()));
        return fail(ref);
    }
    if (cutoff_override) {
        ref->cutoff = *cutoff_override;
    } else if (device_cutoff(ctx, ctx->tmp_a.as<double>(), nk, cutoff_repeats, nullptr, &ref->cutoff, k) != WC_OK) {
        return fail(ref);
    }
    hipLaunchKernelGGL(k_ref_lists, dim3((unsigned)cdiv(n_bins, 128)), dim3(128), 0, nullptr,
                       (const int *)ctx->tmp_b.as<int>(), (const double *)ctx->tmp_a.as<double>(), n_bins, k,
                       (const int64_t *)ref->moff_dev.as<int64_t>(), n_chrom, ref->cutoff, ref->gidx.as<int>(),
                       ref->nref.as<int>());
    if (hipDeviceSynchronize() != hipSuccess) {
        wc::set_error("reference: list kernel failed: %s", hipGetErrorString(hipGetLastError()));
        return fail(ref);
    }
    {
        // reverse lists (users of every bin), CSR: counts on the device, offsets on the host
        // (one-off set-up), then the fill with the counts array recycled as cursors
        const unsigned gb = (unsigned)cdiv(n_bins * k, 256);
        if (ref->users_off.reserve(sizeof(int) * (n_bins + 1)) || ctx->tmp_c.reserve(sizeof(int) * n_bins))
            return fail(ref);
        int *cnt = ctx->tmp_c.as<int>();
        std::vector<int> host(n_bins + 1, 0);
        bool ok = hipMemset(cnt, 0, sizeof(int) * n_bins) == hipSuccess;
        if (ok) {
            hipLaunchKernelGGL(k_count_users, dim3(gb), dim3(256), 0, nullptr, (const int *)ref->gidx.as<int>(),
                               (const int *)ref->nref.as<int>(), n_bins, k, cnt);
            ok = hipMemcpy(host.data() + 1, cnt, sizeof(int) * n_bins, hipMemcpyDeviceToHost) == hipSuccess;
        }
        if (ok) {
            host[0] = 0;
            for (int64_t g = 0; g < n_bins; ++g) host[g + 1] += host[g];
            ok = ref->users.reserve(sizeof(int) * std::max<int64_t>(host[n_bins], 1)) == WC_OK &&
                 hipMemcpy(ref->users_off.p, host.data(), sizeof(int) * (n_bins + 1), hipMemcpyHostToDevice) == hipSuccess &&
                 hipMemset(cnt, 0, sizeof(int) * n_bins) == hipSuccess;
        }
        if (ok) {
            hipLaunchKernelGGL(k_fill_users, dim3(gb), dim3(256), 0, nullptr, (const int *)ref->gidx.as<int>(),
                               (const int *)ref->nref.as<int>(), n_bins, k, (const int *)ref->users_off.as<int>(), cnt,
                               ref->users.as<int>());
            ok = hipDeviceSynchronize() == hipSuccess;
        }
        if (!ok) {
            wc::set_error("reference: reverse lists failed: %s", hipGetErrorString(hipGetLastError()));
            return fail(ref);
        }
    }
    return ref;
}

void wc_reference_destroy(wc_reference *ref) {
    if (!ref) return;
    if (ref->ctx) (void)hipSetDevice(ref->ctx->device);
    (void)hipDeviceSynchronize();
    for (wc::DevBuf *b : {&ref->gidx, &ref->nref, &ref->pca_mean, &ref->pca_comp, &ref->m2g, &ref->g2m,
                          &ref->moff_dev, &ref->goff_dev, &ref->users_off, &ref->users})
        b->release();
    delete ref;
}

double wc_reference_cutoff(const wc_reference *ref) { return ref ? ref->cutoff : NAN; }

int wc_prepare_samples(wc_ctx *ctx, const wc_reference *ref, const int32_t *counts, int64_t n_samples, double *out,
                       double *raw) {
    WC_CHECK(ctx && ref && counts && out && n_samples > 0, WC_E_ARG, "prepare: bad argument");
    WC_HIP(hipSetDevice(ctx->device));
    int rc;
    if ((rc = ctx->ts.counts.reserve(sizeof(int) * n_samples * ref->Btot))) return rc;
    WC_HIP(hipMemcpy(ctx->ts.counts.p, counts, sizeof(int) * n_samples * ref->Btot, hipMemcpyHostToDevice));
    if ((rc = run_prepare(ctx, ref, ctx->ts.counts.as<int>(), n_samples, nullptr))) return rc;
    WC_HIP(hipDeviceSynchronize());
    WC_HIP(hipMemcpy(out, ctx->ts.data.p, sizeof(double) * n_samples * ref->B, hipMemcpyDeviceToHost));
    if (raw) WC_HIP(hipMemcpy(raw, ctx->ts.raw.p, sizeof(double) * n_samples * ref->B, hipMemcpyDeviceToHost));
    return WC_OK;
}

int wc_apply_pca(wc_ctx *ctx, const double *samples, int64_t n_samples, int64_t n_bins, const double *pca_mean,
                 const double *pca_components, int n_comp, double *out) {
    WC_CHECK(ctx && samples && pca_mean && out && n_samples > 0 && n_bins > 0, WC_E_ARG, "applyPCA: bad argument");
    WC_CHECK(n_samples <= 60000, WC_E_LIMIT, "applyPCA: more than 60000 samples per call");
    WC_CHECK(n_comp >= 0 && n_comp <= MAX_COMP && (n_comp == 0 || pca_components), WC_E_ARG,
             "applyPCA: 0..%d components supported", MAX_COMP);
    WC_HIP(hipSetDevice(ctx->device));
    TestState &ts = ctx->ts;
    int rc;
    if ((rc = ts.raw.reserve(sizeof(double) * n_samples * n_bins))) return rc;
    if ((rc = ts.data.reserve(sizeof(double) * n_samples * n_bins))) return rc;
    if ((rc = ts.proj.reserve(sizeof(double) * n_samples * MAX_COMP * PROJ_SPLIT))) return rc;
    if ((rc = ctx->tmp_a.reserve(sizeof(double) * n_bins))) return rc;
    if ((rc = ctx->tmp_c.reserve(sizeof(double) * std::max<int64_t>(1, (int64_t)n_comp * n_bins)))) return rc;
    WC_HIP(hipMemcpy(ts.raw.p, samples, sizeof(double) * n_samples * n_bins, hipMemcpyHostToDevice));
    WC_HIP(hipMemcpy(ctx->tmp_a.p, pca_mean, sizeof(double) * n_bins, hipMemcpyHostToDevice));
    if (n_comp) WC_HIP(hipMemcpy(ctx->tmp_c.p, pca_components, sizeof(double) * n_comp * n_bins, hipMemcpyHostToDevice));
    hipLaunchKernelGGL(k_pca_project, dim3((unsigned)n_samples, PROJ_SPLIT), dim3(256), 0, nullptr,
                       (const double *)ts.raw.as<double>(), n_bins, (const double *)ctx->tmp_a.as<double>(),
                       (const double *)ctx->tmp_c.as<double>(), n_comp, ts.proj.as<double>());
    dim3 g((unsigned)cdiv(n_bins, 256), (unsigned)n_samples);
    hipLaunchKernelGGL(k_pca_apply, g, dim3(256), 0, nullptr, (const double *)ts.raw.as<double>(), n_bins,
                       (const double *)ctx->tmp_a.as<double>(), (const double *)ctx->tmp_c.as<double>(), n_comp,
                       (const double *)ts.proj.as<double>(), ts.data.as<double>());
    WC_HIP(hipDeviceSynchronize());
    WC_HIP(hipMemcpy(out, ts.data.p, sizeof(double) * n_samples * n_bins, hipMemcpyDeviceToHost));
    return WC_OK;
}

int wc_repeat_test(wc_ctx *ctx, const wc_reference *ref, const double *data, int64_t n_samples, double threshold,
                   int repeats, double *z, double *r, double *ref_sizes, double *sd_avg) {
    WC_CHECK(ctx && ref && data && n_samples > 0, WC_E_ARG, "repeatTest: bad argument");
    WC_HIP(hipSetDevice(ctx->device));
    TestState &ts = ctx->ts;
    const int64_t n = n_samples * ref->B;
    int rc;
    if ((rc = ts.data.reserve(sizeof(double) * n))) return rc;
    if ((rc = ts.z.reserve(sizeof(double) * n))) return rc;
    WC_HIP(hipMemcpy(ts.data.p, data, sizeof(double) * n, hipMemcpyHostToDevice));
    if ((rc = run_repeat(ctx, ref, ts.data.as<double>(), n_samples, threshold, repeats, nullptr))) return rc;
    if ((rc = join_side(ctx, nullptr))) return rc;
    struct { wc::DevBuf *src; double *dst; } outs[] = {{&ts.zt, z}, {&ts.rt, r}, {&ts.nt, ref_sizes}};
    for (auto &o : outs) {
        if (!o.dst) continue;
        launch_transpose(o.src->as<double>(), ref->B, n_samples, ts.z.as<double>(), nullptr);
        WC_HIP(hipDeviceSynchronize());
        WC_HIP(hipMemcpy(o.dst, ts.z.p, sizeof(double) * n, hipMemcpyDeviceToHost));
    }
    WC_HIP(hipDeviceSynchronize());
    if (sd_avg) WC_HIP(hipMemcpy(sd_avg, ts.sd_avg.p, sizeof(double) * n_samples, hipMemcpyDeviceToHost));
    return WC_OK;
}

int wc_std_dev_avg(wc_ctx *ctx, const double *sd, int64_t n_samples, int64_t n_bins, double *out,
                   int32_t *serial_samples) {
    WC_CHECK(ctx && sd && out && n_samples > 0 && n_bins > 0, WC_E_ARG, "stdDevAvg: bad argument");
    WC_HIP(hipSetDevice(ctx->device));
    TestState &ts = ctx->ts;
    const int64_t n = n_samples * n_bins;
    int rc;
    if ((rc = ts.data.reserve(sizeof(double) * n))) return rc;
    if ((rc = ts.sdt.reserve(sizeof(double) * n))) return rc;
    if ((rc = ts.sd_avg.reserve(sizeof(double) * n_samples))) return rc;
    if ((rc = ts.sd_fail.reserve(sizeof(int) * n_samples))) return rc;
    WC_HIP(hipMemcpy(ts.data.p, sd, sizeof(double) * n, hipMemcpyHostToDevice));
    launch_transpose(ts.data.as<double>(), n_samples, n_bins, ts.sdt.as<double>(), nullptr);     // bin-major like the repeats
    WC_HIP(hipMemset(ts.sd_fail.p, 0, sizeof(int) * n_samples));
    const int *only = nullptr;
    if (n_bins <= 65536) {
        launch_sd_fast(nullptr, ts.data.as<double>(), n_bins, n_samples, ts.sd_avg.as<double>(), ts.sd_fail.as<int>(),
                       nullptr, 1, n_bins);       // the caller's layout: a sample's values contiguous
        only = ts.sd_fail.as<int>();
    }
    hipLaunchKernelGGL(k_sd_avg<64>, dim3((unsigned)cdiv(n_samples, 64)), dim3(256), 0, nullptr,
                       (const double *)ts.sdt.as<double>(), n_bins, n_samples, ts.sd_avg.as<double>(), only,
                       (double *)nullptr, n_samples, (int64_t)1);
    WC_HIP(hipDeviceSynchronize());
    WC_HIP(hipMemcpy(out, ts.sd_avg.p, sizeof(double) * n_samples, hipMemcpyDeviceToHost));
    if (serial_samples) {
        std::vector<int> f(n_samples, 1);
        if (only) WC_HIP(hipMemcpy(f.data(), ts.sd_fail.p, sizeof(int) * n_samples, hipMemcpyDeviceToHost));
        int c = 0;
        for (int v : f) c += v != 0;
        *serial_samples = c;
    }
    return WC_OK;
}

int wc_stouffer_segments(wc_ctx *ctx, const double *z, const double *ratio, double min_effect,
                         const int64_t *region_offsets, int64_t n_regions, double threshold, int min_search,
                         int max_calls, double *region_z, int32_t *n_calls, double *call_value, int32_t *call_x,
                         int32_t *call_y) {
    WC_CHECK(ctx && z && region_offsets && n_regions >= 0 && max_calls > 0, WC_E_ARG, "segments: bad argument");
    if (n_regions == 0) return WC_OK;
    WC_CHECK(n_regions <= 60000, WC_E_LIMIT, "segments: more than 60000 regions per call");
    WC_HIP(hipSetDevice(ctx->device));
    TestState &ts = ctx->ts;
    const int64_t total = region_offsets[n_regions];
    std::vector<Region> regs(n_regions);
    int64_t max_n = 0;
    for (int64_t r = 0; r < n_regions; ++r) {
        int64_t n = region_offsets[r + 1] - region_offsets[r];
        WC_CHECK(n >= 0 && n < (1ll << 30), WC_E_ARG, "segments: bad region offsets");
        regs[r].off = region_offsets[r];
        regs[r].n = (int)n;
        regs[r].pad = 0;
        max_n = std::max(max_n, n);
    }
    int rc;
    if ((rc = ts.zc.reserve(sizeof(double) * std::max<int64_t>(total, 1)))) return rc;
    if ((rc = ts.regions.reserve(sizeof(Region) * n_regions))) return rc;
    WC_HIP(hipMemcpy(ts.zc.p, z, sizeof(double) * total, hipMemcpyHostToDevice));
    WC_HIP(hipMemcpy(ts.regions.p, regs.data(), sizeof(Region) * n_regions, hipMemcpyHostToDevice));
    int64_t bits_upper = 0;
    if (min_effect != 0.0) {
        WC_CHECK(ratio, WC_E_ARG, "segments: mineffectsize needs the ratio vector");
        if ((rc = ts.rc.reserve(sizeof(double) * std::max<int64_t>(total, 1)))) return rc;
        WC_HIP(hipMemcpy(ts.rc.p, ratio, sizeof(double) * total, hipMemcpyHostToDevice));
        for (int64_t r = 0; r < n_regions; ++r) bits_upper += (int64_t)regs[r].n * (regs[r].n + 1) / 2;
    }
    if ((rc = run_stouffer(ctx, ts.zc.as<double>(), ts.regions.as<Region>(), n_regions, total, max_n, threshold,
                           min_search, max_calls, nullptr, ts.rc.as<double>(), min_effect, bits_upper)))
        return rc;
    WC_HIP(hipDeviceSynchronize());
    if (region_z) WC_HIP(hipMemcpy(region_z, ts.whole.p, sizeof(double) * n_regions, hipMemcpyDeviceToHost));
    std::vector<int> hn(n_regions);
    WC_HIP(hipMemcpy(hn.data(), ts.out_n.p, sizeof(int) * n_regions, hipMemcpyDeviceToHost));
    for (int64_t r = 0; r < n_regions; ++r) {
        WC_CHECK(hn[r] <= max_calls, WC_E_LIMIT, "segments: region %lld has %d calls, max_calls is %d", (long long)r,
                 hn[r], max_calls);
        if (n_calls) n_calls[r] = hn[r];
    }
    const int64_t m = n_regions * max_calls;
    if (call_value) WC_HIP(hipMemcpy(call_value, ts.out_val.p, sizeof(double) * m, hipMemcpyDeviceToHost));
    if (call_x) WC_HIP(hipMemcpy(call_x, ts.out_x.p, sizeof(int) * m, hipMemcpyDeviceToHost));
    if (call_y) WC_HIP(hipMemcpy(call_y, ts.out_y.p, sizeof(int) * m, hipMemcpyDeviceToHost));
    return WC_OK;
}

// Everything wc_test_batch_dev enqueues.  lat_rounds == 0: the general path (host-driven segmentation
// rounds, synchronises per round and at the end).  lat_rounds > 0: latency mode -- no host round
// trip at all (capturable in a hipGraph); the read-backs the caller has to look at after it
// synchronised the stream land in ctx->pinned: int[16] = call overflow flag, int[24..31] = the
// segmentation counters ([6] a round bound was exceeded, [lat_left] jobs left after the last round).
static int test_batch_body(wc_ctx *ctx, hipStream_t stream, const wc_reference *ref, const int32_t *counts, int64_t Ns,
                           double threshold, int min_ref_bins, int repeats, double min_effect,
                           const std::vector<int> &sel, int64_t max_n, int max_calls, double *results_z,
                           double *results_r, double *results_cwz, double *calls, int32_t *n_calls, double *asdef,
                           int lat_rounds) {
    TestState &ts = ctx->ts;
    const int64_t B = ref->B;
    const int n_sel = (int)sel.size();
    int rc;
    ts.prof_tag.clear();
    ts.mark(0, stream);
    const bool lat = lat_rounds > 0;
    // The z-score stage of a batch works on bin-major arrays [bins, Np], Np = the sample count padded to a multiple of
    // 16: a reference bin's row then starts on a 128-byte line and the tiled z-score kernel's 16-sample columns are
    // whole lines; the extra samples are copies of sample 0 whose results nobody reads.
    const int64_t Np = (!lat && Ns >= 32) ? ((Ns + 15) & ~(int64_t)15) : Ns;
    if (lat) {
        // totals, normalisation, PCA and the repeats' working arrays in one launch
        const int64_t n = B * Ns, n_words = cdiv(n, 32);
        if ((rc = ts.totals.reserve(sizeof(double) * Ns))) return rc;
        if ((rc = ts.raw.reserve(sizeof(double) * n))) return rc;
        if ((rc = ts.data.reserve(sizeof(double) * n))) return rc;
        if ((rc = reserve_repeat_arrays(ts, n, Np))) return rc;
        if ((rc = ts.misc2.reserve(sizeof(int) * (repeats + 2 + n_words)))) return rc;
        if ((rc = ts.proj.reserve(sizeof(double) * Ns * MAX_COMP * PROJ_SPLIT))) return rc;
        hipLaunchKernelGGL(k_lat_project, dim3((unsigned)Ns, PROJ_SPLIT), dim3(256), 0, stream, counts, ref->Btot,
                           (const int *)ref->m2g.as<int>(), B, (const double *)ref->pca_mean.as<double>(),
                           (const double *)ref->pca_comp.as<double>(), ref->n_comp, ts.totals.as<double>(),
                           ts.proj.as<double>());
        hipLaunchKernelGGL(k_lat_apply, dim3((unsigned)cdiv(B, 256), (unsigned)Ns), dim3(256), 0, stream, counts,
                           ref->Btot, (const int *)ref->m2g.as<int>(), B, Ns, (const double *)ts.totals.as<double>(),
                           (const double *)ref->pca_mean.as<double>(), (const double *)ref->pca_comp.as<double>(),
                           ref->n_comp, (const double *)ts.proj.as<double>(), ts.data.as<double>(), ts.xt.as<double>(),
                           ts.xc.as<double>(), ts.misc2.as<int>(), repeats + 2 + n_words);
    } else {
        // totals -> projection partial sums (normalised values on the fly) -> corrected values straight into the
        // repeats' bin-major arrays: three launches, no sample-major intermediate
        const int64_t n = B * Np, n_words = cdiv(n, 32);
        if ((rc = ts.totals.reserve(sizeof(long long) * Ns * TOT_SPLIT))) return rc;
        if ((rc = ts.proj.reserve(sizeof(double) * Ns * MAX_COMP * PROJ_SPLIT))) return rc;
        if ((rc = reserve_repeat_arrays(ts, n, Np))) return rc;
        if ((rc = ts.misc2.reserve(sizeof(int) * (repeats + 2 + n_words)))) return rc;
        hipLaunchKernelGGL(k_sample_totals, dim3((unsigned)Ns, TOT_SPLIT), dim3(256), 0, stream, counts, ref->Btot,
                           ts.totals.as<long long>());
        hipLaunchKernelGGL(k_pca_project, dim3((unsigned)Ns, PROJ_SPLIT), dim3(256), 0, stream, (const double *)nullptr, B,
                           (const double *)ref->pca_mean.as<double>(), (const double *)ref->pca_comp.as<double>(),
                           ref->n_comp, ts.proj.as<double>(), counts, ref->Btot, (const int *)ref->m2g.as<int>(),
                           (const long long *)ts.totals.as<long long>());
        hipLaunchKernelGGL(k_pca_apply_t, dim3((unsigned)cdiv(B, 32), (unsigned)cdiv(Np, 32)), dim3(32, 8), 0, stream, counts,
                           ref->Btot, (const int *)ref->m2g.as<int>(), B, Ns, Np, (const long long *)ts.totals.as<long long>(),
                           (const double *)ref->pca_mean.as<double>(), (const double *)ref->pca_comp.as<double>(),
                           ref->n_comp, (const double *)ts.proj.as<double>(), ts.xt.as<double>(), ts.xc.as<double>(),
                           ts.misc2.as<int>(), repeats > 0 ? repeats + 2 + n_words : 0);
    }
    ts.mark(1, stream);
    if ((rc = run_repeat(ctx, ref, ts.data.as<double>(), Np, threshold, repeats, stream, lat, asdef, !lat, !lat,
                         !lat && calls && n_calls)))      // (the late repeats as one launch: only where the status check below runs)
        return rc;
    // run_repeat forked the side stream for stdDevAvg at its very end; with sample-major outputs nothing is enqueued on
    // the launch stream before the inflated outputs go to the side stream too: they ride on the same fork
    ctx->side_fresh = !lat && ts.sm_out && ctx->side_pending && !ts.profile;
    ts.mark(2, stream);
    struct Joiner {   // asdef is copied out once the side stream's sum is done, on every exit path
        wc_ctx *c; hipStream_t s; double *dst; int64_t n; bool on;
        void now() {
            if (on && join_side(c, s) == WC_OK && dst)
                (void)hipMemcpyAsync(dst, c->ts.sd_avg.p, sizeof(double) * n, hipMemcpyDeviceToDevice, s);
            on = false;
        }
        ~Joiner() { now(); }
    } joiner{ctx, stream, asdef, Ns, !lat};
    struct LatJoin {   // latency mode: asdef is written by the kernel itself; the side branch still has to rejoin
        wc_ctx *c; hipStream_t s; bool on;
        ~LatJoin() { if (on) (void)join_side(c, s); }
    } lat_join{ctx, stream, lat};
    // z, ratio and reference counts back to sample-major [Ns, B] for the per-sample consumers
    // (latency mode: the few samples are read straight from the bin-major arrays)
    const double *zsrc = ts.zt.as<double>(), *rsrc = ts.rt.as<double>(), *nsrc = ts.nt.as<double>();
    int64_t str_i = 1, str_b = Ns;
    if (!lat && ts.sm_out) {
        // the tiled first repeat (and the later repeats behind it) wrote them sample-major: nothing to transpose
        str_i = B; str_b = 1;
    } else if (!lat) {
        for (wc::DevBuf *b : {&ts.zs, &ts.rs2, &ts.ns2})
            if ((rc = b->reserve(sizeof(double) * Np * B))) return rc;
        dim3 g3((unsigned)cdiv(Np, 32), (unsigned)cdiv(B, 32), 3);
        hipLaunchKernelGGL(k_transpose3, g3, dim3(32, 8), 0, stream, (const double *)ts.zt.as<double>(),
                           (const double *)ts.rt.as<double>(), (const double *)ts.nt.as<double>(), B, Np,
                           ts.zs.as<double>(), ts.rs2.as<double>(), ts.ns2.as<double>());
        zsrc = ts.zs.as<double>(); rsrc = ts.rs2.as<double>(); nsrc = ts.ns2.as<double>();
        str_i = B; str_b = 1;
    }
    // latency mode with something to segment: results_z / results_r are written by rider workgroups of the
    // last launch (k_seg_tree) instead of a launch of their own on the critical path
    // (k_assemble_calls as a rider too measured 8 us slower than its own launch: device-scope fences; EXPERIMENTS.md)
    const bool ride_inf = lat && n_sel > 0 && calls && n_calls;
    if ((results_z || results_r) && !ride_inf) {
        dim3 g((unsigned)cdiv(ref->Btot, 256), (unsigned)Ns);
        // a batch: the inflated outputs feed nothing downstream -> side stream, under the segmentation
        hipStream_t is = stream;
        const bool second = side_second(Ns);
        if (!lat && Ns > 8) {
            if ((rc = side_begin(ctx, stream, second))) return rc;
            is = second ? ctx->side2 : ctx->side;
        }
        hipLaunchKernelGGL(k_inflate, g, dim3(256), 0, is, zsrc, rsrc, nsrc, B, ref->Btot,
                           (const int *)ref->g2m.as<int>(), (double)min_ref_bins, results_z, results_r, str_i, str_b);
        if (is != stream && (rc = side_end(ctx, second))) return rc;
    }
    ctx->side_fresh = false;      // (whatever follows on the launch stream is not covered by that fork)
    if (n_sel == 0) {      // nothing to segment: no calls (otherwise k_assemble_calls writes every n_calls)
        if (n_calls) WC_HIP(hipMemsetAsync(n_calls, 0, sizeof(int) * Ns, stream));
        return WC_OK;
    }
    const int64_t n_regions = Ns * n_sel;
    bool cwz_done = false;        // the fused set-up kernel wrote results_cwz itself
    if ((rc = ts.zc.reserve(sizeof(double) * Ns * B))) return rc;
    if ((rc = ts.rc.reserve(sizeof(double) * Ns * B))) return rc;
    if ((rc = ts.gpos.reserve(sizeof(int) * Ns * B))) return rc;
    if ((rc = ts.regions.reserve(sizeof(Region) * n_regions))) return rc;
    if ((rc = ts.effect.reserve(sizeof(double) * n_regions * max_calls * 5))) return rc;
    if ((rc = ts.job_cnt.reserve(sizeof(int) * 16))) return rc;
    if (lat) {
        if ((rc = ctx->ensure_pinned(256))) return rc;
        InflateRider inf{};
        if (ride_inf && (results_z || results_r)) {
            {
                inf.blocks = (int)std::min<int64_t>(16, cdiv(ref->Btot * Ns, 1024));
                inf.zs = zsrc; inf.rs = rsrc; inf.ns = nsrc;
                inf.B = B; inf.Btot = ref->Btot; inf.Ns = Ns; inf.si = str_i; inf.sb = str_b;
                inf.g2m = ref->g2m.as<int>();
                inf.minref = (double)min_ref_bins;
                inf.res_z = results_z; inf.res_r = results_r;
            }
        }
        if ((rc = run_seg_lat(ctx, ref, zsrc, rsrc, nsrc, str_i, str_b, Ns, n_sel, max_n, threshold, min_ref_bins,
                              max_calls, stream, results_cwz, inf)))
            return rc;
    } else {
        const bool fuse = min_effect == 0.0 && max_n <= TREE_MAXLEN;       // regions that fit the fused set-up kernel
        // (256 threads = 1 024 bins per trip: 1 000 x 50 kb 1.55 ms with 1 024 threads, 0.92 with 512, 0.56 with 256, 0.68
        //  with 128 -- eight small workgroups per CU cover each other's barriers and tails; 125 x 50 kb 124 -> < 95 us)
        if (!fuse)
        hipLaunchKernelGGL(k_clean, dim3((unsigned)n_sel, (unsigned)Ns),
                           dim3(max_n <= 8192 ? 256u : 1024u), 0, stream, zsrc, rsrc, nsrc, B, Ns,
                           (const int64_t *)ref->moff_dev.as<int64_t>(), (const int64_t *)ref->goff_dev.as<int64_t>(),
                           (const int *)ref->m2g.as<int>(), (const int *)ts.sel.as<int>(), n_sel, (double)min_ref_bins,
                           ts.zc.as<double>(), ts.rc.as<double>(), ts.gpos.as<int>(), ts.regions.as<Region>(), str_i,
                           str_b);
        int64_t bits_upper = 0;
        if (min_effect != 0.0)
            for (int s2 = 0; s2 < n_sel; ++s2) {
                int64_t n = ref->moff[sel[s2] + 1] - ref->moff[sel[s2]];
                bits_upper += Ns * (n * (n + 1) / 2);
            }
        ts.mark(3, stream);
        const TreeTail tail{ts.rc.as<double>(), ts.gpos.as<int>(), ts.effect.as<double>(), calls && n_calls && !ts.profile,
                            (!fuse && min_effect == 0.0) ? results_cwz : nullptr, n_sel};
        const FusedSetup fsu{zsrc, rsrc, nsrc, str_i, str_b, B, ref->moff_dev.as<int64_t>(), ref->goff_dev.as<int64_t>(),
                             ref->m2g.as<int>(), ts.sel.as<int>(), n_sel, (double)min_ref_bins, ts.zc.as<double>(),
                             ts.rc.as<double>(), ts.gpos.as<int>(), ts.regions.as<Region>(), results_cwz};
        cwz_done = results_cwz && (fuse || (min_effect == 0.0 && calls && n_calls));
        if ((rc = run_stouffer(ctx, ts.zc.as<double>(), ts.regions.as<Region>(), n_regions, Ns * B, max_n, threshold, 3,
                               max_calls, stream, ts.rc.as<double>(), min_effect, bits_upper, 0, nullptr,
                               calls && n_calls ? &tail : nullptr, fuse ? &fsu : nullptr)))
            return rc;
    }
    ts.mark(4, stream);
    if (results_cwz && !lat && !cwz_done)
        WC_HIP(hipMemcpyAsync(results_cwz, ts.whole.p, sizeof(double) * n_regions, hipMemcpyDeviceToDevice, stream));
    if ((rc = ctx->ensure_pinned(256))) return rc;
    if (calls && n_calls) {
        // the overflow flag (counters + 8) was cleared by k_region_prefix
        if (ts.last_segs > 0)
            hipLaunchKernelGGL(k_call_post, dim3((unsigned)ts.last_segs), dim3(CP_THREADS), 0, stream,
                               (const Seg *)ts.seg.as<Seg>(), (int)ts.last_segs,
                               (const Region *)ts.regions.as<Region>(), (const double *)ts.rc.as<double>(),
                               (const int *)ts.gpos.as<int>(), max_calls, ts.effect.as<double>(), (const int *)nullptr);
        if (lat && (rc = join_side(ctx, stream))) return rc;      // the status words read k_sd_fast's flags
        if (!lat && n_sel <= 64)
            hipLaunchKernelGGL(k_assemble_batch, dim3((unsigned)Ns), dim3(64), 0, stream, (const double *)ts.effect.as<double>(),
                               (const int *)ts.out_n.as<int>(), n_sel, max_calls, Ns, calls, n_calls, (ts.job_cnt.as<int>() + 8));
        else
        hipLaunchKernelGGL(k_assemble_calls, dim3((unsigned)cdiv(Ns, 64)), dim3(64), 0, stream,   // latency mode: Ns <= 8, one workgroup
                           (const double *)ts.effect.as<double>(), (const int *)ts.out_n.as<int>(), n_sel, max_calls, Ns,
                           calls, n_calls, (ts.job_cnt.as<int>() + 8), lat ? (int *)ctx->pinned : (int *)nullptr,
                           (const int *)ts.job_cnt.as<int>(), (const int *)(ts.misc2.as<int>() + repeats + 1),
                           (const int *)ts.sd_fail.as<int>());
        // batches: the segmentation's counters [0..7] and the flag words behind them ([8] a sample with more than max_calls
        // calls, [9] the late repeats' overflow) come back in ONE copy
        int *overflow = (int *)ctx->pinned + (lat ? 16 : 8);
        if (!lat) {
            // the side stream rejoins and asdef is copied out IN FRONT of the synchronize: enqueued behind it (the
            // destructor's place) they were one more host round trip -- 25 us of idle GPU -- at the end of every batch
            joiner.now();
            WC_HIP(hipMemcpyAsync(ctx->pinned, ts.job_cnt.p, 12 * sizeof(int), hipMemcpyDeviceToHost, stream));
            WC_HIP(hipStreamSynchronize(stream));
            if (ts.tail_used && overflow[1] != 0) {
                // a late repeat had more pairs queued than the one-workgroup form takes: the batch again, a launch pair per repeat
                if (getenv("WC_TEST_VERBOSE")) fprintf(stderr, "wisecondor_amd: batch repeated with a launch pair per late repeat\n");
                ts.tree_pending = false;
                ts.no_tail = true;
                rc = test_batch_body(ctx, stream, ref, counts, Ns, threshold, min_ref_bins, repeats, min_effect, sel,
                                     max_n, max_calls, results_z, results_r, results_cwz, calls, n_calls, asdef, lat_rounds);
                ts.no_tail = false;
                return rc;
            }
            if (ts.tree_pending) {
                // the tree kernel's status words arrived with this synchronize (run_stouffer queued the copy)
                ts.tree_pending = false;
                const int *h = (const int *)ctx->pinned;
                if (h[3] != 0 || h[6] != 0) {
                    if (getenv("WC_TEST_VERBOSE"))
                        fprintf(stderr, "wisecondor_amd: batch repeated on the host-driven rounds (tree status %d, walk status %d: 1 = region "
                                        "beyond %d bins, 2 = non-finite z, 4 = ties beyond the record, 8 = more than %d segments, 16 = stack)\n",
                                h[3], h[6], CJ_MAXLEN, TREE_SEGS);
                    // rare (non-finite region, tie overflow, deep recursion): the whole batch again with
                    // host-driven rounds -- the same results by construction, one batch time lost
                    ts.no_tree = true;
                    rc = test_batch_body(ctx, stream, ref, counts, Ns, threshold, min_ref_bins, repeats, min_effect, sel,
                                         max_n, max_calls, results_z, results_r, results_cwz, calls, n_calls, asdef, lat_rounds);
                    ts.no_tree = false;
                    return rc;
                }
                WC_CHECK(h[4] <= ts.tree_seg_cap, WC_E_LIMIT, "stouffer: more than max_calls=%d segments per region", max_calls);
            }
            WC_CHECK(!*overflow, WC_E_LIMIT, "test: a sample has more than max_calls=%d calls", max_calls);
            if (getenv("WC_TEST_VERBOSE") && ts.misc2.p && repeats > 0 && repeats < 16) {
                // the (bin, sample) pairs every repeat had queued: [0] the first repeat's threshold hits (tiled kernel), [it] the pairs repeat it + 1 recomputed
                int pc[18];
                if (hipMemcpy(pc, ts.misc2.p, sizeof(int) * (repeats + 2), hipMemcpyDeviceToHost) == hipSuccess) {
                    fprintf(stderr, "wisecondor_amd: pairs queued per repeat:");
                    for (int q = 0; q <= repeats; ++q) fprintf(stderr, " %d", pc[q]);
                    fprintf(stderr, "\n");
                }
            }
            if (getenv("WC_TEST_VERBOSE") && ts.sd_fail.p && ref->B <= 65536 && ctx->side) {
                // how many samples the parallel stdDevAvg gave up on (the serial kernel computed them)
                (void)hipStreamSynchronize(ctx->side);
                std::vector<int> f((size_t)Ns);
                if (hipMemcpy(f.data(), ts.sd_fail.p, sizeof(int) * Ns, hipMemcpyDeviceToHost) == hipSuccess) {
                    int64_t nf = 0;
                    for (int64_t q = 0; q < Ns; ++q) nf += f[q] != 0;
                    fprintf(stderr, "wisecondor_amd: stdDevAvg: %lld of %lld samples took the serial kernel\n", (long long)nf,
                            (long long)Ns);
                }
            }
        }
    }
    ts.mark(5, stream);
    WC_HIP(hipGetLastError());
    return WC_OK;
}

int wc_test_batch_dev(wc_ctx *ctx, void *stream_, const wc_reference *ref, const int32_t *counts, int64_t n_samples,
                      double threshold, int min_ref_bins, int repeats, double min_effect,
                      const int32_t *chromosomes_host, int n_sel, int max_calls, double *results_z,
                      double *results_r, double *results_cwz, double *calls, int32_t *n_calls, double *asdef) {
    WC_CHECK(ctx && ref && counts && n_samples > 0, WC_E_ARG, "test: bad argument");
    WC_CHECK(n_samples <= 60000, WC_E_LIMIT, "test: more than 60000 samples per call; split the batch");
    WC_CHECK(n_sel >= 0 && n_sel <= WC_MAX_CHROM && max_calls > 0, WC_E_ARG, "test: bad chromosome selection");
    WC_CHECK(n_sel == 0 || chromosomes_host, WC_E_ARG, "test: NULL chromosome list");
    hipStream_t stream = (hipStream_t)stream_;
    WC_HIP(hipSetDevice(ctx->device));
    TestState &ts = ctx->ts;
    const int64_t Ns = n_samples;
    int rc;
    std::vector<int> sel(n_sel);
    int64_t max_n = 0;
    for (int s = 0; s < n_sel; ++s) {
        int c = chromosomes_host[s] - 1;
        WC_CHECK(c >= 0 && c < ref->n_chrom, WC_E_ARG, "test: chromosome %d out of range", chromosomes_host[s]);
        sel[s] = c;
        max_n = std::max(max_n, ref->moff[c + 1] - ref->moff[c]);
    }
    WC_CHECK(Ns * n_sel <= 60000, WC_E_LIMIT, "test: samples x chromosomes = %lld exceeds 60000 per call; split the batch",
             (long long)(Ns * n_sel));
    if (n_sel > 0) {
        if ((rc = ts.sel.reserve(sizeof(int) * n_sel))) return rc;
        if (sel != ts.sel_host) {      // the device copy of the chromosome selection is reused across calls
            WC_HIP(hipMemcpyAsync(ts.sel.p, sel.data(), sizeof(int) * n_sel, hipMemcpyHostToDevice, stream));
            WC_HIP(hipStreamSynchronize(stream));
            ts.sel_host = sel;
        }
    }
    // Latency mode (BASELINE config 3: one sample per call): the whole call is one hipGraph replay of
    // eight launches without a host round trip; what the fixed-shape kernels cannot hold (deep or wide
    // recursions, too many queued pairs, a failed stdDevAvg assumption) is detected afterwards through
    // status words and the call is repeated on the general path.  The first call of a shape runs
    // eagerly (it sizes every workspace), the second one is captured, later ones replay.
    constexpr int LAT_MAX_SAMPLES = 8, LAT_ROUNDS = 1;
    const char *lat_env = getenv("WC_TEST_LATENCY_MODE");          // "0": general path for every call
    const bool lat = Ns <= LAT_MAX_SAMPLES && min_effect == 0.0 && n_sel > 0 && calls && n_calls && !ts.profile &&
                     ref->k <= 128 && repeats >= 1 && max_n <= 2048 && ref->B * Ns < (1ll << 31) &&
                     !(lat_env && lat_env[0] == '0');
    if (!lat)
        return test_batch_body(ctx, stream, ref, counts, Ns, threshold, min_ref_bins, repeats, min_effect, sel, max_n,
                               max_calls, results_z, results_r, results_cwz, calls, n_calls, asdef, 0);
    std::vector<int64_t> key = {(int64_t)ref->serial, (int64_t)(intptr_t)counts, Ns, min_ref_bins, repeats, max_calls,
                                (int64_t)(intptr_t)results_z, (int64_t)(intptr_t)results_r,
                                (int64_t)(intptr_t)results_cwz, (int64_t)(intptr_t)calls, (int64_t)(intptr_t)n_calls,
                                (int64_t)(intptr_t)asdef};
    {
        int64_t tbits;
        memcpy(&tbits, &threshold, 8);
        key.push_back(tbits);
        for (int c : sel) key.push_back(c);
    }
    auto fall_back = [&]() {
        return test_batch_body(ctx, stream, ref, counts, Ns, threshold, min_ref_bins, repeats, min_effect, sel, max_n,
                               max_calls, results_z, results_r, results_cwz, calls, n_calls, asdef, 0);
    };
    if (ts.lat_exec && ts.lat_epoch != wc::realloc_epoch()) ts.lat_key.clear();   // a workspace moved since the capture
    if (key != ts.lat_key) {
        // new shape: drop the old graph, run eagerly once (reserves every buffer), capture next time
        if (ts.lat_exec) { (void)hipGraphExecDestroy(ts.lat_exec); ts.lat_exec = nullptr; }
        ts.lat_key = key;
        ts.lat_warm = false;
    }
    // the call runs on the context's own stream (the caller's may be the NULL stream, which cannot
    // be captured), ordered after everything the caller has enqueued so far
    if (!ctx->lat_stream) {
        WC_HIP(hipStreamCreateWithFlags(&ctx->lat_stream, hipStreamNonBlocking));
        WC_HIP(hipEventCreateWithFlags(&ctx->ev_lat_in, hipEventDisableTiming));
        WC_HIP(hipEventCreateWithFlags(&ctx->ev_lat_out, hipEventDisableTiming));
    }
    hipStream_t ls = ctx->lat_stream;
    if (stream != nullptr) {
        ls = stream;                      // a real stream: launch (and capture) on it directly
    } else {
        WC_HIP(hipEventRecord(ctx->ev_lat_in, stream));
        WC_HIP(hipStreamWaitEvent(ls, ctx->ev_lat_in, 0));
    }
    // "2": the latency kernels, launched one by one; a shape whose capture failed once stays there
    const bool eager = (lat_env && lat_env[0] == '2') || key == ts.lat_fail_key;
    if (!ts.lat_exec && ts.lat_warm && ts.lat_epoch != wc::realloc_epoch()) ts.lat_warm = false;   // workspaces moved: size them again
    if (!ts.lat_exec && ts.lat_warm && !eager) {
        hipGraph_t graph = nullptr;
        if (hipStreamBeginCapture(ls, hipStreamCaptureModeThreadLocal) != hipSuccess) {
            (void)hipGetLastError();
            ts.lat_fail_key = key;
            return fall_back();
        }
        rc = test_batch_body(ctx, ls, ref, counts, Ns, threshold, min_ref_bins, repeats, min_effect, sel, max_n,
                             max_calls, results_z, results_r, results_cwz, calls, n_calls, asdef, LAT_ROUNDS);
        const hipError_t e = hipStreamEndCapture(ls, &graph);
        if (rc != WC_OK || e != hipSuccess || !graph) {
            if (graph) (void)hipGraphDestroy(graph);
            (void)hipGetLastError();
            ts.lat_fail_key = key;
            return fall_back();
        }
        const hipError_t ei = hipGraphInstantiate(&ts.lat_exec, graph, nullptr, nullptr, 0);
        (void)hipGraphDestroy(graph);
        ts.lat_epoch = wc::realloc_epoch();
        if (ei != hipSuccess) {
            ts.lat_exec = nullptr;
            ts.lat_fail_key = key;
            (void)hipGetLastError();
            return fall_back();
        }
    }
    if (ts.lat_exec) {
        WC_HIP(hipGraphLaunch(ts.lat_exec, ls));
    } else {
        if ((rc = test_batch_body(ctx, ls, ref, counts, Ns, threshold, min_ref_bins, repeats, min_effect, sel, max_n,
                                  max_calls, results_z, results_r, results_cwz, calls, n_calls, asdef, LAT_ROUNDS)))
            return rc;
        ts.lat_warm = true;
        ts.lat_epoch = wc::realloc_epoch();
    }
    WC_HIP(hipStreamSynchronize(ls));
    const int *overflow = (const int *)ctx->pinned + 16, *cnt = (const int *)ctx->pinned + 24;
    // anything the latency kernels are not built for -> the general path computes the call again
    // (non-finite region, tie overflow, deep stack: cnt[6]; many segments; many queued pairs)
    if (cnt[6] || ((const int *)ctx->pinned)[32] || ((const int *)ctx->pinned)[33]) return fall_back();
    WC_CHECK(!*overflow, WC_E_LIMIT, "test: a sample has more than max_calls=%d calls", max_calls);
    return WC_OK;
}

int wc_debug_times(wc_ctx *ctx, int block_plus_one, unsigned long long *out64) {
    WC_CHECK(ctx, WC_E_ARG, "debug: NULL context");
    WC_HIP(hipSetDevice(ctx->device));
    WC_HIP(hipDeviceSynchronize());
    if (out64) WC_HIP(hipMemcpyFromSymbol(out64, HIP_SYMBOL(g_dbg), sizeof(unsigned long long) * 64));
    WC_HIP(hipMemcpyToSymbol(HIP_SYMBOL(g_dbg_on), &block_plus_one, sizeof(int)));
    return WC_OK;
}

int wc_test_profile(wc_ctx *ctx, int enable) {
    WC_CHECK(ctx, WC_E_ARG, "profile: NULL context");
    ctx->ts.profile = enable != 0;
    ctx->ts.prof_tag.clear();
    return WC_OK;
}

int wc_test_profile_read(wc_ctx *ctx, double out[8]) {
    WC_CHECK(ctx && out, WC_E_ARG, "profile: NULL argument");
    for (int i = 0; i < 8; ++i) out[i] = 0.0;
    TestState &ts = ctx->ts;
    if (ts.prof_tag.empty()) return WC_OK;
    WC_HIP(hipSetDevice(ctx->device));
    WC_HIP(hipDeviceSynchronize());
    auto ms = [&](size_t a, size_t b) {
        float t = 0.f;
        return hipEventElapsedTime(&t, ts.prof_ev[a], ts.prof_ev[b]) == hipSuccess ? (double)t : 0.0;
    };
    // stage boundaries 0..5 in order; pairs (10, 11) bracket the certificate + search launches of a round
    size_t last_stage = 0, open_search = 0;
    bool have_stage = false, have_open = false;
    for (size_t i = 0; i < ts.prof_tag.size(); ++i) {
        const int tag = ts.prof_tag[i];
        if (tag >= 0 && tag <= 5) {
            if (have_stage && tag >= 1) out[tag - 1] += ms(last_stage, i);
            last_stage = i;
            have_stage = true;
        } else if (tag == 10) {
            open_search = i;
            have_open = true;
        } else if (tag == 11 && have_open) {
            out[5] += ms(open_search, i);
            have_open = false;
        }
    }
    if (ts.prof_work.p) {
        unsigned long long w[128];
        WC_HIP(hipMemcpy(w, ts.prof_work.p, sizeof(w), hipMemcpyDeviceToHost));
        out[6] = out[7] = 0.0;
        for (int q = 0; q < 64; ++q) { out[6] += (double)w[2 * q]; out[7] += (double)w[2 * q + 1]; }
    }
    return WC_OK;
}

int wc_test_batch(wc_ctx *ctx, const wc_reference *ref, const int32_t *counts, int64_t n_samples, double threshold,
                  int min_ref_bins, int repeats, double min_effect, const int32_t *chromosomes, int n_sel,
                  int max_calls, double *results_z, double *results_r, double *results_cwz, double *calls,
                  int32_t *n_calls, double *asdef) {
    WC_CHECK(ctx && ref && counts && n_samples > 0, WC_E_ARG, "test: bad argument");
    WC_HIP(hipSetDevice(ctx->device));
    TestState &ts = ctx->ts;
    const int64_t Ns = n_samples;
    int rc;
    if ((rc = ts.counts.reserve(sizeof(int) * Ns * ref->Btot))) return rc;
    if ((rc = ts.res_z.reserve(sizeof(double) * Ns * ref->Btot))) return rc;
    if ((rc = ts.res_r.reserve(sizeof(double) * Ns * ref->Btot))) return rc;
    if ((rc = ts.cwz.reserve(sizeof(double) * Ns * std::max(n_sel, 1)))) return rc;
    if ((rc = ts.calls.reserve(sizeof(double) * Ns * max_calls * 5))) return rc;
    if ((rc = ts.n_calls.reserve(sizeof(int) * Ns))) return rc;
    if ((rc = ts.n.reserve(sizeof(double) * Ns))) return rc;
    WC_HIP(hipMemcpy(ts.counts.p, counts, sizeof(int) * Ns * ref->Btot, hipMemcpyHostToDevice));
    rc = wc_test_batch_dev(ctx, nullptr, ref, ts.counts.as<int>(), Ns, threshold, min_ref_bins, repeats, min_effect,
                           chromosomes, n_sel, max_calls, ts.res_z.as<double>(), ts.res_r.as<double>(), ts.cwz.as<double>(),
                           ts.calls.as<double>(), ts.n_calls.as<int>(), ts.n.as<double>());
    if (rc) return rc;
    WC_HIP(hipDeviceSynchronize());
    if (results_z) WC_HIP(hipMemcpy(results_z, ts.res_z.p, sizeof(double) * Ns * ref->Btot, hipMemcpyDeviceToHost));
    if (results_r) WC_HIP(hipMemcpy(results_r, ts.res_r.p, sizeof(double) * Ns * ref->Btot, hipMemcpyDeviceToHost));
    if (results_cwz && n_sel > 0)
        WC_HIP(hipMemcpy(results_cwz, ts.cwz.p, sizeof(double) * Ns * n_sel, hipMemcpyDeviceToHost));
    if (calls) WC_HIP(hipMemcpy(calls, ts.calls.p, sizeof(double) * Ns * max_calls * 5, hipMemcpyDeviceToHost));
    if (n_calls) WC_HIP(hipMemcpy(n_calls, ts.n_calls.p, sizeof(int) * Ns, hipMemcpyDeviceToHost));
    if (asdef) WC_HIP(hipMemcpy(asdef, ts.n.p, sizeof(double) * Ns, hipMemcpyDeviceToHost));
    return WC_OK;
}

}  // extern "C"
