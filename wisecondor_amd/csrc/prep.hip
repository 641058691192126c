// newref prep on gfx950: toNumpyArray's normalisation and
// zero mask (wisetools.py:240-264) and trainPCA (wisetools.py:89-101) as an exact,
// deterministic rank-n PCA: float64 Gram matrix of the centred [samples, bins] data on the
// GPU, its small [samples, samples] eigenproblem on the host (LAPACK-free Jacobi), the
// components, projection, reconstruction and the corrected matrix on the GPU.
#include "ctx.h"

#include <algorithm>

namespace {

inline int64_t cdiv(int64_t a, int64_t b) { return (a + b - 1) / b; }

// per-sample totals over all bins: integer sums are exact in any order (wisetools.py:255)
__global__ __launch_bounds__(256) void k_prep_totals(const int *__restrict__ counts, int64_t Btot,
                                                     double *__restrict__ totals) {
    __shared__ long long sh[256];
    const int *row = counts + (int64_t)blockIdx.x * Btot;
    long long s = 0;
    for (int64_t g = threadIdx.x; g < Btot; g += 256) s += row[g];
    sh[threadIdx.x] = s;
    __syncthreads();
    for (int o = 128; o > 0; o >>= 1) {
        if ((int)threadIdx.x < o) sh[threadIdx.x] += sh[threadIdx.x + o];
        __syncthreads();
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

// maskedData[b, s] = counts[s, m2g[b]] / total[s]; tdata[s, b] the same transposed
__global__ void k_prep_normalize(const int *__restrict__ counts, int64_t S, int64_t Btot, const int *__restrict__ m2g,
                                 int64_t B, const double *__restrict__ totals, double *__restrict__ masked,
                                 double *__restrict__ tdata) {
    int64_t b = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    int64_t s = blockIdx.y;
    if (b >= B) return;
    double v = (double)counts[s * Btot + m2g[b]] / totals[s];
    masked[b * S + s] = v;
    tdata[s * B + b] = v;
}

// mean over samples: tData is the transposed VIEW of maskedData (wisetools.py:90), so
// numpy's axis-0 mean runs along the contiguous sample axis of maskedData[b, :], i.e. a
// pairwise sum per bin; then centre
__global__ void k_prep_centre(const double *__restrict__ masked, const double *__restrict__ tdata, int64_t S,
                              int64_t B, double *__restrict__ mean, double *__restrict__ xc) {
    int64_t b = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (b >= B) return;
    const double *row = masked + b * S;
    const double sum = wc::pairwise_sum<false>([&](int64_t s) { return row[s]; }, S, 0);
    const double m = sum / (double)S;
    mean[b] = m;
    for (int64_t s = 0; s < S; ++s) xc[s * B + b] = tdata[s * B + b] - m;
}

// Gram matrix G[s, t] = sum_b xc[s, b] * xc[t, b]: 16x16 output tile per block, bins in
// chunks of 64 through LDS, fixed summation order (deterministic)
__global__ __launch_bounds__(256) void k_prep_gram(const double *__restrict__ xc, int64_t S, int64_t B,
                                                   double *__restrict__ G) {
    __shared__ double As[16][65], Bs[16][65];
    const int tx = threadIdx.x & 15, ty = threadIdx.x >> 4;
    const int64_t s0 = (int64_t)blockIdx.y * 16, t0 = (int64_t)blockIdx.x * 16;
    if (t0 > s0) return;   // symmetric: lower triangle of tiles only
    double acc = 0.0;
    for (int64_t b0 = 0; b0 < B; b0 += 64) {
        for (int e = threadIdx.x; e < 16 * 64; e += 256) {
            int r = e >> 6, c = e & 63;
            int64_t b = b0 + c;
            As[r][c] = (s0 + r < S && b < B) ? xc[(s0 + r) * B + b] : 0.0;
            Bs[r][c] = (t0 + r < S && b < B) ? xc[(t0 + r) * B + b] : 0.0;
        }
        __syncthreads();
#pragma unroll 8
        for (int c = 0; c < 64; ++c) acc += As[ty][c] * Bs[tx][c];
        __syncthreads();
    }
    if (s0 + ty < S && t0 + tx < S) {
        G[(s0 + ty) * S + t0 + tx] = acc;
        G[(t0 + tx) * S + s0 + ty] = acc;
    }
}

// components[c, b] = sum_s w[c, s] * xc[s, b]   (w = eigenvector / singular value)
__global__ void k_prep_components(const double *__restrict__ xc, int64_t S, int64_t B, const double *__restrict__ w,
                                  int n_comp, double *__restrict__ comp) {
    int64_t b = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (b >= B) return;
    for (int c = 0; c < n_comp; ++c) {
        double acc = 0.0;
        for (int64_t s = 0; s < S; ++s) acc += w[(int64_t)c * S + s] * xc[s * B + b];
        comp[(int64_t)c * B + b] = acc;
    }
}

// transformed[s, c] = sum_b xc[s, b] * comp[c, b]  (pca.transform, wisetools.py:94)
__global__ __launch_bounds__(256) void k_prep_transform(const double *__restrict__ xc, int64_t B,
                                                        const double *__restrict__ comp, int n_comp,
                                                        double *__restrict__ tr) {
    __shared__ double sh[8][256];
    const double *x = xc + (int64_t)blockIdx.x * B;
    double acc[8];
#pragma unroll
    for (int c = 0; c < 8; ++c) acc[c] = 0.0;
    for (int64_t b = threadIdx.x; b < B; b += 256) {
        double v = x[b];
#pragma unroll
        for (int c = 0; c < 8; ++c)
            if (c < n_comp) acc[c] += v * comp[(int64_t)c * B + b];
    }
#pragma unroll
    for (int c = 0; c < 8; ++c) sh[c][threadIdx.x] = acc[c];
    __syncthreads();
    for (int o = 128; o > 0; o >>= 1) {
        if ((int)threadIdx.x < o)
#pragma unroll
            for (int c = 0; c < 8; ++c) sh[c][threadIdx.x] += sh[c][threadIdx.x + o];
        __syncthreads();
    }
    if ((int)threadIdx.x < n_comp) tr[(int64_t)blockIdx.x * 8 + threadIdx.x] = sh[threadIdx.x][0];
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
    WC_CHECK(ctx && counts && chromosome_bins && mask_out && masked_chrom_bins_out && n_masked_out && gram_out,
             WC_E_ARG, "prep: NULL argument");
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
    WC_HIP(hipMemcpy(ts.sel.p, m2g.data(), sizeof(int) * B, hipMemcpyHostToDevice));
    dim3 gb((unsigned)cdiv(B, 256), (unsigned)S);
    hipLaunchKernelGGL(k_prep_normalize, gb, dim3(256), 0, nullptr, (const int *)ts.counts.as<int>(), S, Btot,
                       (const int *)ts.sel.as<int>(), B, (const double *)ts.totals.as<double>(), ts.raw.as<double>(),
                       ts.data.as<double>());
    hipLaunchKernelGGL(k_prep_centre, dim3((unsigned)cdiv(B, 256)), dim3(256), 0, nullptr,
                       (const double *)ts.raw.as<double>(), (const double *)ts.data.as<double>(), S, B, mean,
                       ts.xt.as<double>());
    dim3 gg((unsigned)cdiv(S, 16), (unsigned)cdiv(S, 16));
    hipLaunchKernelGGL(k_prep_gram, gg, dim3(256), 0, nullptr, (const double *)ts.xt.as<double>(), S, B, G);
    WC_HIP(hipDeviceSynchronize());
    WC_HIP(hipMemcpy(gram_out, G, sizeof(double) * S * S, hipMemcpyDeviceToHost));
    g_prep.S = S; g_prep.Btot = Btot; g_prep.B = B; g_prep.ready = true;
    return WC_OK;
}

int wc_newref_prep_finish(wc_ctx *ctx, int n_comp, const double *eigvecs, const double *eigvals,
                          double *masked_data_out, double *corrected_t_out, double *pca_components_out,
                          double *pca_mean_out) {
    WC_CHECK(ctx && eigvecs && eigvals && masked_data_out && corrected_t_out && pca_components_out && pca_mean_out,
             WC_E_ARG, "prep: NULL argument");
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
    hipLaunchKernelGGL(k_prep_components, dim3((unsigned)cdiv(B, 256)), dim3(256), 0, nullptr,
                       (const double *)ts.xt.as<double>(), S, B, (const double *)w, n_comp, comp);
    // scikit-learn's svd_flip (v based): the largest |entry| of every component is positive
    std::vector<double> hc((size_t)n_comp * B);
    WC_HIP(hipMemcpy(hc.data(), comp, sizeof(double) * n_comp * B, hipMemcpyDeviceToHost));
    for (int c = 0; c < n_comp; ++c) {
        double *row = hc.data() + (size_t)c * B;
        int64_t arg = 0;
        for (int64_t b = 1; b < B; ++b)
            if (fabs(row[b]) > fabs(row[arg])) arg = b;
        if (row[arg] < 0.0)
            for (int64_t b = 0; b < B; ++b) row[b] = -row[b];
    }
    WC_HIP(hipMemcpy(comp, hc.data(), sizeof(double) * n_comp * B, hipMemcpyHostToDevice));
    hipLaunchKernelGGL(k_prep_transform, dim3((unsigned)S), dim3(256), 0, nullptr, (const double *)ts.xt.as<double>(),
                       B, (const double *)comp, n_comp, ts.proj.as<double>());
    dim3 gb((unsigned)cdiv(B, 256), (unsigned)S);
    hipLaunchKernelGGL(k_prep_correct, gb, dim3(256), 0, nullptr, (const double *)ts.data.as<double>(), B,
                       (const double *)mean, (const double *)comp, n_comp, (const double *)ts.proj.as<double>(),
                       ts.xc.as<double>());
    WC_HIP(hipDeviceSynchronize());
    WC_HIP(hipMemcpy(masked_data_out, ts.raw.p, sizeof(double) * B * S, hipMemcpyDeviceToHost));
    WC_HIP(hipMemcpy(corrected_t_out, ts.xc.p, sizeof(double) * B * S, hipMemcpyDeviceToHost));
    memcpy(pca_components_out, hc.data(), sizeof(double) * n_comp * B);
    WC_HIP(hipMemcpy(pca_mean_out, mean, sizeof(double) * B, hipMemcpyDeviceToHost));
    return WC_OK;
}

// One-call variant for callers without a LAPACK: the [samples, samples] eigenproblem is
// solved by cyclic Jacobi on the host (fine up to a few hundred samples).
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
    jacobi_eigh(gram, (int)n_samples, val, vec);
    return wc_newref_prep_finish(ctx, n_comp, vec.data(), val.data(), masked_data_out, corrected_t_out,
                                 pca_components_out, pca_mean_out);
}

}  // extern "C"
