// newref prep on gfx950: toNumpyArray's normalisation and
// zero mask (wisetools.py:240-264) and trainPCA (wisetools.py:89-101) as an exact,
// deterministic rank-n PCA: float64 Gram matrix of the centred [samples, bins] data on the
// GPU, its small [samples, samples] eigenproblem by the direct solver of eigh.hip (or by the caller:
// the Gram matrix can be fetched and the pairs handed back), the components, projection,
// reconstruction and the corrected matrix on the GPU.
#include "ctx.h"

#include <algorithm>

namespace {

inline int64_t cdiv(int64_t a, int64_t b) { return (a + b - 1) / b; }
inline int64_t round_up(int64_t a, int64_t b) { return cdiv(a, b) * b; }

// per-sample totals over all bins: integer sums are exact in any order (wisetools.py:255)
__global__ __launch_bounds__(256) void k_prep_totals(const int *__restrict__ counts, int64_t Btot,
                                                     double *__restrict__ totals) {
    __shared__ long long sh[256];
    const int *row = counts + (int64_t)blockIdx.x * Btot;
    long long s = 0;
    for (int64_t g = threadIdx.x; g < Btot; g += 256) s += row[g];
    sh[threadIdx.x] = s;
    wc_sync();
    for (int o = 128; o > 0; o >>= 1) {
        if ((int)threadIdx.x < o) sh[threadIdx.x] += sh[threadIdx.x + o];
        wc_sync();
    }
    if (threadIdx.x == 0) totals[blockIdx.x] = (double)sh[0];
}

// mask[g] = any sample has a positive normalised value in bin g (sum over samples > 0, wisetools.py:259-260)
__global__ void k_prep_mask(const int *__restrict__ counts, int64_t S, int64_t Btot, const double *__restrict__ totals,
                            unsigned char *__restrict__ mask) {
    int64_t g = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (g >= Btot) return;
    double sum = 0.0;
    for (int64_t s = 0; s < S; ++s) sum = sum + (double)counts[s * Btot + g] / totals[s];
    mask[g] = sum > 0.0;
}

// maskedData[b, s] = counts[s, m2g[b]] / total[s], tdata[s, b] the same transposed, the per-bin mean
// and the centred matrix xc[s, b], in ONE pass: a workgroup takes NB neighbouring bins for all
// samples through LDS (row stride S | 1: the transposed reads are conflict free), so every
// global access is a contiguous run -- whole rows of maskedData, NB x 8 bytes of the three
// [samples, bins] arrays -- where the two per-element kernels before it wrote one double per
// 4.8 KB stride (0.97 -> 0.3 ms at 600 x 50 kb).  The mean is numpy's: tData is the transposed
// VIEW of maskedData (wisetools.py:90), so the axis-0 mean runs along the contiguous sample axis
// of maskedData[b, :], a pairwise sum per bin -- walked by eight lanes per bin.
__global__ __launch_bounds__(256, 2) void k_prep_norm_centre(const int *__restrict__ counts, int64_t S, int64_t Btot,
                                                          const int *__restrict__ m2g, int64_t B,
                                                          const double *__restrict__ totals, int NB,
                                                          double *__restrict__ masked, double *__restrict__ tdata,
                                                          double *__restrict__ mean, double *__restrict__ xc) {
    extern __shared__ double tile[];
    __shared__ double sh_mean[32];
    const int64_t b0 = (int64_t)blockIdx.x * NB;
    const int nb = (int)(B - b0 < NB ? B - b0 : NB);
    const int64_t ld = S | 1;
    const int tid = threadIdx.x;
    const int i = tid % NB, s0 = tid / NB, sstep = 256 / NB;
    if (i < nb) {
        const int64_t g = m2g[b0 + i];
        int64_t s = s0;
        for (; s + 3 * sstep < S; s += 4 * sstep) {          // four rows' loads in flight per thread
            int c[4];
            double tt[4];
#pragma unroll
            for (int u = 0; u < 4; ++u) { c[u] = counts[(s + u * sstep) * Btot + g]; tt[u] = totals[s + u * sstep]; }
#pragma unroll
            for (int u = 0; u < 4; ++u) tile[i * ld + s + u * sstep] = (double)c[u] / tt[u];
        }
        for (; s < S; s += sstep) tile[i * ld + s] = (double)counts[s * Btot + g] / totals[s];
    }
    wc_sync();
    for (int r = 0; r < nb; ++r)
        for (int64_t s = tid; s < S; s += 256) masked[(b0 + r) * S + s] = tile[r * ld + s];
    {
        const int grp = tid >> 3, sub = tid & 7;
        const int r = grp < nb ? grp : 0;         // (whole waves walk the tree together)
        const double sum = wc::pairwise_sum<true>([&](int64_t s) { return tile[r * ld + s]; }, S, sub);
        if (grp < nb && sub == 0) {
            const double m = sum / (double)S;
            sh_mean[grp] = m;
            mean[b0 + grp] = m;
        }
    }
    wc_sync();
    if (i < nb) {
        const double m = sh_mean[i];
        for (int64_t s = s0; s < S; s += sstep) {
            const double v = tile[i * ld + s];
            tdata[s * B + b0 + i] = v;
            xc[s * B + b0 + i] = v - m;
        }
    }
}

// Gram matrix G[s, t] = sum_b xc[s, b] * xc[t, b] on the float64 matrix cores
// (v_mfma_f64_16x16x4_f64), as a split-K SYRK: one workgroup = one 64 x 64 tile of the lower
// triangle x one slice of the bins; four waves, each a 32 x 32 quadrant as 2 x 2 MFMA blocks.
// 32-bin panels of both operands go through LDS (row stride 34 doubles: the sixteen rows of a
// fragment read land on distinct banks).  Slice results are written as partial tiles and summed
// in slice order by k_prep_gram_reduce: deterministic, no atomics.
// Operand maps (cdna_hip_programming.md, f64 form): A[i = lane & 15][k = lane >> 4],
// B[k = lane >> 4][j = lane & 15], C/D col = lane & 15, row = (lane >> 4) + 4 * reg.
using f64x4 = __attribute__((ext_vector_type(4))) double;
using f64x2p = __attribute__((ext_vector_type(2))) double;
constexpr int SY_T = 64, SY_KB = 32, SY_LD = SY_KB + 2;
__global__ __launch_bounds__(256) void k_prep_syrk(const double *__restrict__ xc, int64_t S, int64_t B,
                                                   const int2 *__restrict__ tiles, int64_t bins_per_slice,
                                                   double *__restrict__ partial) {
    // two LDS stages: the next panel's global loads are issued before the current panel's 32 MFMAs
    // per wave and land in the other stage after them -- one barrier per panel, the memory latency
    // under the matrix-core time (single-staged: load, barrier, store, barrier, multiply; 0.30 of
    // the float64 matrix-core time the tiles need)
    __shared__ __attribute__((aligned(16))) double As[2][SY_T * SY_LD], Bs[2][SY_T * SY_LD];
    const int tid = threadIdx.x, lane = tid & 63, w = tid >> 6;
    const int2 tile = tiles[blockIdx.x];
    const int64_t s0 = (int64_t)tile.x * SY_T, t0 = (int64_t)tile.y * SY_T;
    const int64_t b_lo = (int64_t)blockIdx.y * bins_per_slice;
    const int64_t b_hi = b_lo + bins_per_slice < B ? b_lo + bins_per_slice : B;
    const int wr = (w >> 1) * 32, wc = (w & 1) * 32;
    const int fi = lane & 15, fk = lane >> 4;
    f64x4 acc[2][2];
#pragma unroll
    for (int m = 0; m < 2; ++m)
#pragma unroll
        for (int n = 0; n < 2; ++n) acc[m][n] = f64x4{0.0, 0.0, 0.0, 0.0};
    const int lr = tid >> 2, lc = (tid & 3) * 8;          // this thread's row and first column of a panel
    const bool a_row = s0 + lr < S, b_row = t0 + lr < S;
    const double *ga = xc + (a_row ? s0 + lr : 0) * B, *gb = xc + (b_row ? t0 + lr : 0) * B;
    double va[8], vb[8];
    auto fetch = [&](int64_t b0) {
        if (b0 + lc + 8 <= b_hi && (((b0 + lc) & 1) == 0) && ((B & 1) == 0)) {
#pragma unroll
            for (int e = 0; e < 8; e += 2) {
                const f64x2p pa = *(const f64x2p *)(ga + b0 + lc + e), pb = *(const f64x2p *)(gb + b0 + lc + e);
                va[e] = pa.x; va[e + 1] = pa.y;
                vb[e] = pb.x; vb[e + 1] = pb.y;
            }
        } else {
#pragma unroll
            for (int e = 0; e < 8; ++e) {
                const bool in = b0 + lc + e < b_hi;
                va[e] = in ? ga[b0 + lc + e] : 0.0;
                vb[e] = in ? gb[b0 + lc + e] : 0.0;
            }
        }
    };
    auto stash = [&](int stage) {
#pragma unroll
        for (int e = 0; e < 8; ++e) {
            As[stage][lr * SY_LD + lc + e] = a_row ? va[e] : 0.0;
            Bs[stage][lr * SY_LD + lc + e] = b_row ? vb[e] : 0.0;
        }
    };
    if (b_lo < b_hi) {
        fetch(b_lo);
        stash(0);
    }
    wc_sync();
    int stage = 0;
    for (int64_t b0 = b_lo; b0 < b_hi; b0 += SY_KB, stage ^= 1) {
        const bool more = b0 + SY_KB < b_hi;
        if (more) fetch(b0 + SY_KB);
#pragma unroll
        for (int kk = 0; kk < SY_KB; kk += 4) {
            double fa[2], fb[2];
#pragma unroll
            for (int m = 0; m < 2; ++m) fa[m] = As[stage][(wr + 16 * m + fi) * SY_LD + kk + fk];
#pragma unroll
            for (int n = 0; n < 2; ++n) fb[n] = Bs[stage][(wc + 16 * n + fi) * SY_LD + kk + fk];
#pragma unroll
            for (int m = 0; m < 2; ++m)
#pragma unroll
                for (int n = 0; n < 2; ++n)
                    acc[m][n] = __builtin_amdgcn_mfma_f64_16x16x4f64(fa[m], fb[n], acc[m][n], 0, 0, 0);
        }
        if (more) stash(stage ^ 1);
        wc_sync();
    }
    double *out = partial + ((int64_t)blockIdx.y * gridDim.x + blockIdx.x) * (SY_T * SY_T);
#pragma unroll
    for (int m = 0; m < 2; ++m)
#pragma unroll
        for (int n = 0; n < 2; ++n)
#pragma unroll
            for (int r = 0; r < 4; ++r)
                out[(wr + 16 * m + fk + 4 * r) * SY_T + wc + 16 * n + fi] = acc[m][n][r];
}

// G[s, t] = G[t, s] = sum over the slices, in slice order, of the partial tile entries
__global__ __launch_bounds__(256) void k_prep_gram_reduce(const double *__restrict__ partial, const int2 *__restrict__ tiles,
                                                          int n_tiles, int n_slices, int64_t S, double *__restrict__ G) {
    // grid (tiles, 16): one entry per thread -- the slice loop is a chain of adds, the parallelism has to
    // come from the entries
    const int2 tile = tiles[blockIdx.x];
    const int e = (int)blockIdx.y * 256 + (int)threadIdx.x;
    const int64_t s = (int64_t)tile.x * SY_T + (e >> 6), t = (int64_t)tile.y * SY_T + (e & 63);
    if (s >= S || t >= S || t > s) return;
    double sum = 0.0;
    for (int q = 0; q < n_slices; ++q) sum = sum + partial[((int64_t)q * n_tiles + blockIdx.x) * (SY_T * SY_T) + e];
    G[s * S + t] = sum;
    G[t * S + s] = sum;
}

// comp[c, b] = sum_s w[c, s] * xc[s, b]: the principal axes from the sample-space eigenvectors
// (w = eigenvector / sigma).  64 bins x 16 sample groups per workgroup: every group walks a
// sixteenth of the samples (one read of its part of the column for all components), the sixteen
// partial sums of a bin are added in group order.  (One thread per bin walking all samples kept
// 0.9 waves per SIMD busy: 0.32 ms for one pass over 277 MB at 600 x 50 kb; this form 0.06.)
__global__ __launch_bounds__(1024) void k_prep_components(const double *__restrict__ xc, int64_t S, int64_t B,
                                                          const double *__restrict__ w, int n_comp,
                                                          double *__restrict__ comp) {
    __shared__ double sh[16][64];
    const int x = threadIdx.x & 63, g = threadIdx.x >> 6;
    const int64_t b = (int64_t)blockIdx.x * 64 + x;
    double acc[8];
#pragma unroll
    for (int c = 0; c < 8; ++c) acc[c] = 0.0;
    if (b < B)
        for (int64_t s = g; s < S; s += 16) {
            const double v = xc[s * B + b];
#pragma unroll
            for (int c = 0; c < 8; ++c)
                if (c < n_comp) acc[c] += w[(int64_t)c * S + s] * v;
        }
#pragma unroll
    for (int c = 0; c < 8; ++c) {
        if (c >= n_comp) break;
        sh[g][x] = acc[c];
        wc_sync();
        if (g == 0 && b < B) {
            double sum = 0.0;
            for (int q = 0; q < 16; ++q) sum += sh[q][x];
            comp[(int64_t)c * B + b] = sum;
        }
        wc_sync();
    }
}

// scikit-learn's svd_flip (v based): the entry of largest magnitude of every component becomes
// positive (the first one on ties, like numpy's argmax).  One workgroup per component.
__global__ __launch_bounds__(1024) void k_prep_sign(double *__restrict__ comp, int64_t B) {
    __shared__ double s_abs[16];
    __shared__ long long s_idx[16];
    __shared__ int s_flip;
    double *row = comp + (int64_t)blockIdx.x * B;
    double best = -1.0;
    long long at = 0;
    for (int64_t b = threadIdx.x; b < B; b += 1024) {
        const double v = fabs(row[b]);
        if (v > best) { best = v; at = b; }              // ascending b per thread: the first maximum stays
    }
    for (int o = 32; o > 0; o >>= 1) {
        const double ob = __shfl_xor(best, o);
        const long long oa = __shfl_xor(at, o);
        if (ob > best || (ob == best && oa < at)) { best = ob; at = oa; }
    }
    if ((threadIdx.x & 63) == 0) { s_abs[threadIdx.x >> 6] = best; s_idx[threadIdx.x >> 6] = at; }
    wc_sync();
    if (threadIdx.x == 0) {
        for (int q = 1; q < 16; ++q)
            if (s_abs[q] > best || (s_abs[q] == best && s_idx[q] < at)) { best = s_abs[q]; at = s_idx[q]; }
        s_flip = row[at] < 0.0;
    }
    wc_sync();
    if (s_flip)
        for (int64_t b = threadIdx.x; b < B; b += 1024) row[b] = -row[b];
}

// transformed[s, c] = sum_b xc[s, b] * comp[c, b]  (pca.transform, wisetools.py:94); one workgroup of
// sixteen waves per sample
__global__ __launch_bounds__(1024) void k_prep_transform(const double *__restrict__ xc, int64_t B,
                                                         const double *__restrict__ comp, int n_comp,
                                                         double *__restrict__ tr) {
    __shared__ double sh[8][16];
    const double *x = xc + (int64_t)blockIdx.x * B;
    double acc[8];
#pragma unroll
    for (int c = 0; c < 8; ++c) acc[c] = 0.0;
    for (int64_t b = threadIdx.x; b < B; b += 1024) {
        double v = x[b];
#pragma unroll
        for (int c = 0; c < 8; ++c)
            if (c < n_comp) acc[c] += v * comp[(int64_t)c * B + b];
    }
#pragma unroll
    for (int c = 0; c < 8; ++c) {
        double a = acc[c];
        for (int o = 32; o > 0; o >>= 1) a += __shfl_xor(a, o);
        if ((threadIdx.x & 63) == 0) sh[c][threadIdx.x >> 6] = a;
    }
    wc_sync();
    if ((int)threadIdx.x < n_comp) {
        double sum = 0.0;
        for (int q = 0; q < 16; ++q) sum += sh[threadIdx.x][q];
        tr[(int64_t)blockIdx.x * 8 + threadIdx.x] = sum;
    }
}

// corrected_t[s, b] = tdata[s, b] / (transformed[s, :] . comp[:, b] + mean[b])  (wisetools.py:95-96)
__global__ void k_prep_correct(const double *__restrict__ tdata, int64_t B, const double *__restrict__ mean,
                               const double *__restrict__ comp, int n_comp, const double *__restrict__ tr,
                               double *__restrict__ corrected_t) {
    int64_t b = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    int64_t s = blockIdx.y;
    if (b >= B) return;
    double inv = 0.0;
    for (int c = 0; c < n_comp; ++c) inv += tr[s * 8 + c] * comp[(int64_t)c * B + b];
    inv += mean[b];
    corrected_t[s * B + b] = tdata[s * B + b] / inv;
}

// The same values in the layout newref takes, corrected[b, s] (row-major [bins, samples]):
// 32 x 32 tiles through LDS so that both the reads (along b) and the writes (along s) are coalesced
__global__ __launch_bounds__(256) void k_prep_correct_bs(const double *__restrict__ tdata, int64_t S, int64_t B,
                                                         const double *__restrict__ mean, const double *__restrict__ comp,
                                                         int n_comp, const double *__restrict__ tr,
                                                         double *__restrict__ corrected_bs) {
    __shared__ double tile[32][33];
    const int tx = threadIdx.x & 31, ty = threadIdx.x >> 5;
    const int64_t b0 = (int64_t)blockIdx.x * 32, s0 = (int64_t)blockIdx.y * 32;
    for (int j = ty; j < 32; j += 8) {
        const int64_t s = s0 + j, b = b0 + tx;
        double v = 0.0;
        if (s < S && b < B) {
            double inv = 0.0;
            for (int c = 0; c < n_comp; ++c) inv += tr[s * 8 + c] * comp[(int64_t)c * B + b];
            inv += mean[b];
            v = tdata[s * B + b] / inv;
        }
        tile[j][tx] = v;
    }
    wc_sync();
    for (int j = ty; j < 32; j += 8) {
        const int64_t b = b0 + j, s = s0 + tx;
        if (b < B && s < S) corrected_bs[b * S + s] = tile[tx][j];
    }
}

// Cyclic Jacobi eigen-decomposition of a small symmetric matrix (host, float64).
// Returns eigenvalues (descending) and eigenvectors as rows of `vec`.
void jacobi_eigh(std::vector<double> a, int n, std::vector<double> &val, std::vector<double> &vec) {
    std::vector<double> v((size_t)n * n, 0.0);
    for (int i = 0; i < n; ++i) v[(size_t)i * n + i] = 1.0;
    for (int sweep = 0; sweep < 60; ++sweep) {
        double off = 0.0, diag = 0.0;
        for (int i = 0; i < n; ++i) {
            diag += a[(size_t)i * n + i] * a[(size_t)i * n + i];
            for (int j = i + 1; j < n; ++j) off += a[(size_t)i * n + j] * a[(size_t)i * n + j];
        }
        if (off <= 1e-30 * diag || off == 0.0) break;
        for (int p = 0; p < n - 1; ++p)
            for (int q = p + 1; q < n; ++q) {
                double apq = a[(size_t)p * n + q];
                if (apq == 0.0) continue;
                double app = a[(size_t)p * n + p], aqq = a[(size_t)q * n + q];
                double tau = (aqq - app) / (2.0 * apq);
                double t = (tau >= 0 ? 1.0 : -1.0) / (fabs(tau) + sqrt(1.0 + tau * tau));
                double c = 1.0 / sqrt(1.0 + t * t), s = t * c;
                for (int k = 0; k < n; ++k) {
                    double akp = a[(size_t)k * n + p], akq = a[(size_t)k * n + q];
                    a[(size_t)k * n + p] = c * akp - s * akq;
                    a[(size_t)k * n + q] = s * akp + c * akq;
                }
                for (int k = 0; k < n; ++k) {
                    double apk = a[(size_t)p * n + k], aqk = a[(size_t)q * n + k];
                    a[(size_t)p * n + k] = c * apk - s * aqk;
                    a[(size_t)q * n + k] = s * apk + c * aqk;
                }
                for (int k = 0; k < n; ++k) {
                    double vkp = v[(size_t)k * n + p], vkq = v[(size_t)k * n + q];
                    v[(size_t)k * n + p] = c * vkp - s * vkq;
                    v[(size_t)k * n + q] = s * vkp + c * vkq;
                }
            }
    }
    std::vector<int> order(n);
    for (int i = 0; i < n; ++i) order[i] = i;
    std::sort(order.begin(), order.end(), [&](int x, int y) { return a[(size_t)x * n + x] > a[(size_t)y * n + y]; });
    val.resize(n);
    vec.assign((size_t)n * n, 0.0);
    for (int r = 0; r < n; ++r) {
        val[r] = a[(size_t)order[r] * n + order[r]];
        for (int k = 0; k < n; ++k) vec[(size_t)r * n + k] = v[(size_t)k * n + order[r]];
    }
}

}  // namespace

extern "C" {

int wc_newref_prep_gram(wc_ctx *ctx, const int32_t *counts, int64_t n_samples, int64_t n_total_bins,
                        const int64_t *chromosome_bins, int n_chrom, uint8_t *mask_out,
                        int64_t *masked_chrom_bins_out, int64_t *n_masked_out, double *gram_out) {
    WC_CHECK(ctx && counts && chromosome_bins && mask_out && masked_chrom_bins_out && n_masked_out,
             WC_E_ARG, "prep: NULL argument");           // gram_out may be NULL: the matrix stays in HBM for wc_newref_prep_eig
    WC_CHECK(n_samples > 0 && n_total_bins > 0 && n_chrom > 0 && n_chrom <= WC_MAX_CHROM, WC_E_ARG, "prep: bad shape");
    WC_CHECK(n_samples <= 4096, WC_E_LIMIT, "prep: more than 4096 samples not supported");
    WC_HIP(hipSetDevice(ctx->device));
    PrepState &g_prep = ctx->prep;
    g_prep.ready = false;
    const int64_t S = n_samples, Btot = n_total_bins;
    TestState &ts = ctx->ts;
    int rc;
    if ((rc = ts.counts.reserve(sizeof(int) * S * Btot))) return rc;
    if ((rc = ts.totals.reserve(sizeof(double) * S))) return rc;
    if ((rc = ctx->tmp_d.reserve(Btot))) return rc;
    WC_HIP(hipMemcpy(ts.counts.p, counts, sizeof(int) * S * Btot, hipMemcpyHostToDevice));
    hipLaunchKernelGGL(k_prep_totals, dim3((unsigned)S), dim3(256), 0, nullptr, (const int *)ts.counts.as<int>(), Btot,
                       ts.totals.as<double>());
    hipLaunchKernelGGL(k_prep_mask, dim3((unsigned)cdiv(Btot, 256)), dim3(256), 0, nullptr,
                       (const int *)ts.counts.as<int>(), S, Btot, (const double *)ts.totals.as<double>(),
                       ctx->tmp_d.as<unsigned char>());
    WC_HIP(hipDeviceSynchronize());
    WC_HIP(hipMemcpy(mask_out, ctx->tmp_d.p, Btot, hipMemcpyDeviceToHost));
    std::vector<int> m2g;
    int64_t at = 0;
    for (int c = 0; c < n_chrom; ++c) {
        int64_t in_chrom = 0;
        WC_CHECK(at + chromosome_bins[c] <= Btot, WC_E_ARG, "prep: chromosome bins exceed the count matrix");
        for (int64_t g = at; g < at + chromosome_bins[c]; ++g)
            if (mask_out[g]) { m2g.push_back((int)g); ++in_chrom; }
        masked_chrom_bins_out[c] = in_chrom;
        at += chromosome_bins[c];
    }
    WC_CHECK(at == Btot, WC_E_ARG, "prep: chromosome bins sum to %lld, expected %lld", (long long)at, (long long)Btot);
    const int64_t B = (int64_t)m2g.size();
    *n_masked_out = B;
    WC_CHECK(B > 0, WC_E_ARG, "prep: every bin is empty");
    if ((rc = ts.sel.reserve(sizeof(int) * B))) return rc;
    if ((rc = ts.raw.reserve(sizeof(double) * B * S))) return rc;     // maskedData [B, S]
    if ((rc = ts.data.reserve(sizeof(double) * B * S))) return rc;    // tdata [S, B]
    if ((rc = ts.xt.reserve(sizeof(double) * B * S))) return rc;      // centred [S, B]
    if ((rc = ts.xc.reserve(sizeof(double) * B * S))) return rc;      // corrected_t [S, B]
    if ((rc = ts.z.reserve(sizeof(double) * (S * S + 8 * S + 8 * B + B)))) return rc;
    if ((rc = ts.proj.reserve(sizeof(double) * 8 * S))) return rc;
    double *G = ts.z.as<double>(), *mean = G + S * S + 8 * S + 8 * B;
    ts.sel_host.clear();      // the test path caches its chromosome selection in this buffer: not valid any more
    WC_HIP(hipMemcpy(ts.sel.p, m2g.data(), sizeof(int) * B, hipMemcpyHostToDevice));
    {
        int nbins = 32;                             // bins per workgroup: a power of two, tile <= 96 KB of LDS
        while (nbins > 1 && (int64_t)nbins * (S | 1) * 8 > 96 * 1024) nbins >>= 1;
        const size_t lds = sizeof(double) * (size_t)nbins * (size_t)(S | 1);
        if (lds > 48 * 1024)
            WC_HIP(hipFuncSetAttribute(reinterpret_cast<const void *>(k_prep_norm_centre),
                                       hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds));
        hipLaunchKernelGGL(k_prep_norm_centre, dim3((unsigned)cdiv(B, nbins)), dim3(256), lds, nullptr,
                           (const int *)ts.counts.as<int>(), S, Btot, (const int *)ts.sel.as<int>(), B,
                           (const double *)ts.totals.as<double>(), nbins, ts.raw.as<double>(), ts.data.as<double>(), mean,
                           ts.xt.as<double>());
    }
    {
        // lower triangle of 64 x 64 tiles x bin slices: about 2 000 workgroups, slices of whole panels
        std::vector<int2> tiles;
        const int nt = (int)cdiv(S, SY_T);
        for (int a = 0; a < nt; ++a)
            for (int b = 0; b <= a; ++b) tiles.push_back(make_int2(a, b));
        const int n_tiles = (int)tiles.size();
        int64_t n_slices = std::max<int64_t>(1, std::min<int64_t>(256, 2048 / n_tiles));
        int64_t per = round_up(cdiv(B, n_slices), SY_KB);
        per = std::max<int64_t>(per, 4 * SY_KB);
        n_slices = cdiv(B, per);
        if ((rc = ts.misc.reserve(sizeof(int2) * n_tiles + 16))) return rc;
        if ((rc = ts.zt.reserve(sizeof(double) * n_slices * n_tiles * SY_T * SY_T))) return rc;
        WC_HIP(hipMemcpy(ts.misc.p, tiles.data(), sizeof(int2) * n_tiles, hipMemcpyHostToDevice));
        hipLaunchKernelGGL(k_prep_syrk, dim3((unsigned)n_tiles, (unsigned)n_slices), dim3(256), 0, nullptr,
                           (const double *)ts.xt.as<double>(), S, B, (const int2 *)ts.misc.as<int2>(), per,
                           ts.zt.as<double>());
        hipLaunchKernelGGL(k_prep_gram_reduce, dim3((unsigned)n_tiles, SY_T * SY_T / 256), dim3(256), 0, nullptr,
                           (const double *)ts.zt.as<double>(), (const int2 *)ts.misc.as<int2>(), n_tiles,
                           (int)n_slices, S, G);
    }
    WC_HIP(hipDeviceSynchronize());
    if (gram_out) WC_HIP(hipMemcpy(gram_out, G, sizeof(double) * S * S, hipMemcpyDeviceToHost));
    g_prep.S = S; g_prep.Btot = Btot; g_prep.B = B; g_prep.ready = true;
    return WC_OK;
}

// The leading eigenpairs of the Gram matrix wc_newref_prep_gram left in HBM (eigh.hip): eigenvalues
// descending, unit eigenvectors as rows -- what wc_newref_prep_finish* take.
int wc_newref_prep_eig(wc_ctx *ctx, int n_pairs, double *eigvals_out, double *eigvecs_out) {
    WC_CHECK(ctx && eigvals_out && eigvecs_out, WC_E_ARG, "prep: NULL argument");
    WC_CHECK(ctx->prep.ready, WC_E_ARG, "prep: wc_newref_prep_gram has not run");
    WC_HIP(hipSetDevice(ctx->device));
    return wc::sym_eigh_leading(ctx, ctx->ts.z.as<double>(), ctx->prep.S, n_pairs, eigvals_out, eigvecs_out);
}

// Device part of the finish step: components (sign fixed), projection, corrected_t [S, B] in ts.xc.
// Leaves maskedData in ts.raw, components / mean in ts.z; *hc points at the host copy of the components in the
// context's pinned block (valid once the caller has synchronised; owned by the context, so an error return
// between the copy's launch and that synchronisation leaves nothing dangling).
static int prep_finish_body(wc_ctx *ctx, int n_comp, const double *eigvecs, const double *eigvals,
                            const double **hc, bool want_t) {
    PrepState &g_prep = ctx->prep;
    WC_CHECK(g_prep.ready, WC_E_ARG, "prep: wc_newref_prep_gram has not run");
    const int64_t S = g_prep.S, B = g_prep.B;
    WC_CHECK(n_comp >= 1 && n_comp <= 8 && n_comp <= S, WC_E_ARG, "prep: 1..8 components supported");
    WC_HIP(hipSetDevice(ctx->device));
    TestState &ts = ctx->ts;
    double *G = ts.z.as<double>(), *w = G + S * S, *comp = w + 8 * S, *mean = comp + 8 * B;
    std::vector<double> hw((size_t)8 * S, 0.0);
    for (int c = 0; c < n_comp; ++c) {
        WC_CHECK(eigvals[c] > 0.0, WC_E_ARG, "prep: data has rank below %d", n_comp);
        double inv_sigma = 1.0 / sqrt(eigvals[c]);
        for (int64_t s = 0; s < S; ++s) hw[(size_t)c * S + s] = eigvecs[(size_t)c * S + s] * inv_sigma;
    }
    WC_HIP(hipMemcpy(w, hw.data(), sizeof(double) * 8 * S, hipMemcpyHostToDevice));
    hipLaunchKernelGGL(k_prep_components, dim3((unsigned)cdiv(B, 64)), dim3(1024), 0, nullptr,
                       (const double *)ts.xt.as<double>(), S, B, (const double *)w, n_comp, comp);
    hipLaunchKernelGGL(k_prep_sign, dim3((unsigned)n_comp), dim3(1024), 0, nullptr, comp, B);
    hipLaunchKernelGGL(k_prep_transform, dim3((unsigned)S), dim3(1024), 0, nullptr, (const double *)ts.xt.as<double>(),
                       B, (const double *)comp, n_comp, ts.proj.as<double>());
    // the host copy of the components travels while the rest runs (the callers synchronise): pinned memory, so the
    // copy really is asynchronous
    {
        int rc = ctx->ensure_pinned(sizeof(double) * (size_t)n_comp * B);
        if (rc) return rc;
    }
    *hc = (const double *)ctx->pinned;
    WC_HIP(hipMemcpyAsync(ctx->pinned, comp, sizeof(double) * n_comp * B, hipMemcpyDeviceToHost, nullptr));
    if (want_t) {
        dim3 gb((unsigned)cdiv(B, 256), (unsigned)S);
        hipLaunchKernelGGL(k_prep_correct, gb, dim3(256), 0, nullptr, (const double *)ts.data.as<double>(), B,
                           (const double *)mean, (const double *)comp, n_comp, (const double *)ts.proj.as<double>(),
                           ts.xc.as<double>());
    }
    WC_HIP(hipGetLastError());
    return WC_OK;
}

int wc_newref_prep_finish(wc_ctx *ctx, int n_comp, const double *eigvecs, const double *eigvals,
                          double *masked_data_out, double *corrected_t_out, double *pca_components_out,
                          double *pca_mean_out) {
    WC_CHECK(ctx && eigvecs && eigvals && masked_data_out && corrected_t_out && pca_components_out && pca_mean_out,
             WC_E_ARG, "prep: NULL argument");
    const double *hc = nullptr;
    int rc = prep_finish_body(ctx, n_comp, eigvecs, eigvals, &hc, true);
    if (rc) return rc;
    TestState &ts = ctx->ts;
    const int64_t S = ctx->prep.S, B = ctx->prep.B;
    const double *mean = ts.z.as<double>() + S * S + 8 * S + 8 * B;
    WC_HIP(hipDeviceSynchronize());
    WC_HIP(hipMemcpy(masked_data_out, ts.raw.p, sizeof(double) * B * S, hipMemcpyDeviceToHost));
    WC_HIP(hipMemcpy(corrected_t_out, ts.xc.p, sizeof(double) * B * S, hipMemcpyDeviceToHost));
    memcpy(pca_components_out, hc, sizeof(double) * n_comp * B);
    WC_HIP(hipMemcpy(pca_mean_out, mean, sizeof(double) * B, hipMemcpyDeviceToHost));
    return WC_OK;
}

// Device-resident variant: the bins-sized results stay in HBM.  corrected_bs_dev [B, S] row-major is
// the layout wc_newref_*_dev take (the values of the reference's Fortran-ordered correctedData:
// pass WC_SUM_SEQUENTIAL); masked_dev [B, S]; components [n_comp, B] and mean [B] on the host
// (small).  Every output may be NULL.
int wc_newref_prep_finish_dev(wc_ctx *ctx, int n_comp, const double *eigvecs, const double *eigvals,
                              double *masked_dev, double *corrected_bs_dev, double *pca_components_out,
                              double *pca_mean_out) {
    WC_CHECK(ctx && eigvecs && eigvals, WC_E_ARG, "prep: NULL argument");
    const double *hc = nullptr;
    int rc = prep_finish_body(ctx, n_comp, eigvecs, eigvals, &hc, false);
    if (rc) return rc;
    TestState &ts = ctx->ts;
    const int64_t S = ctx->prep.S, B = ctx->prep.B;
    const double *comp = ts.z.as<double>() + S * S + 8 * S, *mean = comp + 8 * B;
    if (corrected_bs_dev) {
        dim3 g((unsigned)cdiv(B, 32), (unsigned)cdiv(S, 32));
        hipLaunchKernelGGL(k_prep_correct_bs, g, dim3(256), 0, nullptr, (const double *)ts.data.as<double>(), S, B,
                           mean, comp, n_comp, (const double *)ts.proj.as<double>(), corrected_bs_dev);
    }
    if (masked_dev)
        WC_HIP(hipMemcpyAsync(masked_dev, ts.raw.p, sizeof(double) * B * S, hipMemcpyDeviceToDevice, nullptr));
    if (pca_mean_out) WC_HIP(hipMemcpy(pca_mean_out, mean, sizeof(double) * B, hipMemcpyDeviceToHost));
    WC_HIP(hipDeviceSynchronize());
    if (pca_components_out) memcpy(pca_components_out, hc, sizeof(double) * n_comp * B);   // (its copy has landed)
    return WC_OK;
}

// One-call variant: the [samples, samples] eigenproblem by eigh.hip (one or two samples: the
// host Jacobi below, there is nothing to tridiagonalise).
int wc_newref_prep(wc_ctx *ctx, const int32_t *counts, int64_t n_samples, int64_t n_total_bins,
                   const int64_t *chromosome_bins, int n_chrom, int n_comp, uint8_t *mask_out,
                   int64_t *masked_chrom_bins_out, int64_t *n_masked_out, double *masked_data_out,
                   double *corrected_t_out, double *pca_components_out, double *pca_mean_out) {
    std::vector<double> gram((size_t)std::max<int64_t>(n_samples, 1) * std::max<int64_t>(n_samples, 1));
    int rc = wc_newref_prep_gram(ctx, counts, n_samples, n_total_bins, chromosome_bins, n_chrom, mask_out,
                                 masked_chrom_bins_out, n_masked_out, gram.data());
    if (rc) return rc;
    if (!masked_data_out || !corrected_t_out || !pca_components_out || !pca_mean_out) return WC_OK;  // size query
    std::vector<double> val, vec;
    if (n_samples >= 3) {
        WC_CHECK(n_comp >= 1 && n_comp <= 8 && n_comp <= n_samples, WC_E_ARG, "prep: 1..8 components supported");
        val.resize(8);
        vec.resize((size_t)8 * n_samples);
        if ((rc = wc_newref_prep_eig(ctx, n_comp, val.data(), vec.data()))) return rc;
    } else {
        jacobi_eigh(gram, (int)n_samples, val, vec);
    }
    return wc_newref_prep_finish(ctx, n_comp, vec.data(), val.data(), masked_data_out, corrected_t_out,
                                 pca_components_out, pca_mean_out);
}

}  // extern "C"
