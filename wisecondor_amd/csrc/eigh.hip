// The leading eigenpairs of a dense symmetric float64 matrix on gfx950: the [samples, samples]
// eigenproblem of trainPCA (wisetools.py:89-101 hands it to scikit-learn; SURVEY.md section 8f
// rank 1).  A DIRECT method, because the spectrum of this data leaves no choice: one systematic
// component, then the noise bulk with relative gaps of 1e-3 and less between the second, third and
// fourth eigenvalue -- anything that converges like a ratio of eigenvalues needs thousands of
// rounds for the 1e-9 the tests hold.
//
//   1. Householder tridiagonalisation, one launch per column (k_tri_step): the launch boundary
//      is the only grid-wide synchronisation, so nothing ever spins on another workgroup (ranks
//      sharing a GPU cannot deadlock each other).  The rank-2 update of step k-1 is applied
//      lazily inside the pass of step k that multiplies the trailing matrix with the new
//      reflector: ONE read-modify-write sweep over the trailing matrix per column.
//   2. The wanted eigenvalues of the tridiagonal matrix by multisection on Sturm counts
//      (k_tri_eigvals: 256 shifts per round, one workgroup per eigenvalue).
//   3. Their eigenvectors by inverse iteration with partial pivoting, re-orthogonalised among the
//      wanted vectors (k_tri_eigvecs: the recurrences are sequential, one wave per vector).
//   4. Back-transformation through the stored reflectors (k_tri_back).
// Everything is deterministic (fixed reduction orders, no atomics).
#include "ctx.h"

#include <algorithm>
#include <cfloat>
#include <cstdlib>
#include <vector>

namespace {

constexpr double EPS = 2.220446049250313e-16;

// Sum over the 64 lanes of a wave through the DPP network (row shifts, then the two row
// broadcasts): six short adds instead of six trips through the LDS crossbar -- these kernels are
// chains of dependent reductions, the latency is what they cost.  Every lane gets the total.
template <int CTRL, int ROW_MASK>
__device__ inline double dpp_add(double v) {
    const int lo = __builtin_amdgcn_update_dpp(0, __double2loint(v), CTRL, ROW_MASK, 0xF, false);
    const int hi = __builtin_amdgcn_update_dpp(0, __double2hiint(v), CTRL, ROW_MASK, 0xF, false);
    return v + __hiloint2double(hi, lo);
}
__device__ inline double wave_sum(double v) {
    v = dpp_add<0x111, 0xF>(v);      // row_shr:1, 2, 4, 8: lane 15 of every row holds the row's sum
    v = dpp_add<0x112, 0xF>(v);
    v = dpp_add<0x114, 0xF>(v);
    v = dpp_add<0x118, 0xF>(v);
    v = dpp_add<0x142, 0xA>(v);      // row_bcast:15 into rows 1 and 3
    v = dpp_add<0x143, 0xC>(v);      // row_bcast:31 into rows 2 and 3: lane 63 holds the total
    return __hiloint2double(__builtin_amdgcn_readlane(__double2hiint(v), 63),
                            __builtin_amdgcn_readlane(__double2loint(v), 63));
}

// Sum over the 256 threads of a workgroup, the same order everywhere.  `red` holds 4 doubles.
__device__ inline double block_sum256(double v, double *red) {
    v = wave_sum(v);
    wc_sync();
    if ((threadIdx.x & 63) == 0) red[threadIdx.x >> 6] = v;
    wc_sync();
    return (red[0] + red[1]) + (red[2] + red[3]);
}

// Working copy of the matrix, scaled by a power of two so that its largest entry lands in [1, 2):
// the squares of the reflector norms can neither underflow nor overflow whatever the units of the
// input (normalised counts give Gram entries of 1e-10; the products are exact, so are the
// eigenvalues scaled back).  Two launches: per-workgroup maxima, then the scaled copy.
__global__ __launch_bounds__(256) void k_eig_absmax(const double *__restrict__ M, size_t count, double *__restrict__ part) {
    __shared__ double red[4];
    double m = 0.0;
    for (size_t i = (size_t)blockIdx.x * 256 + threadIdx.x; i < count; i += (size_t)gridDim.x * 256) {
        const double v = fabs(M[i]);
        m = (v > m || v != v) ? v : m;            // a NaN sticks
    }
    for (int o = 32; o > 0; o >>= 1) {
        const double other = __shfl_xor(m, o);
        m = (other > m || other != other) ? other : m;
    }
    if ((threadIdx.x & 63) == 0) red[threadIdx.x >> 6] = m;
    wc_sync();
    if (threadIdx.x == 0) {
        for (int w = 1; w < 4; ++w) m = (red[w] > m || red[w] != red[w]) ? red[w] : m;
        part[blockIdx.x] = m;
    }
}

__global__ __launch_bounds__(256) void k_eig_scaled_copy(const double *__restrict__ M, size_t count,
                                                         const double *__restrict__ part, int nparts,
                                                         double *__restrict__ A, double *__restrict__ scale_out) {
    double m = 0.0;
    for (int w = 0; w < nparts; ++w) { const double v = part[w]; m = (v > m || v != v) ? v : m; }
    double s = 1.0;
    if (m > 0.0 && isfinite(m)) s = ldexp(1.0, -ilogb(m));
    if (blockIdx.x == 0 && threadIdx.x == 0) *scale_out = s;
    for (size_t i = (size_t)blockIdx.x * 256 + threadIdx.x; i < count; i += (size_t)gridDim.x * 256) A[i] = M[i] * s;
}

// w_{k-1} = p - (tau/2 p.v) v of the step before `k` (its p came from the previous launch), and
// v_{k-1} itself, into LDS.  Every workgroup derives the same values.
__device__ inline void previous_reflector(const double *__restrict__ V, const double *__restrict__ P,
                                          const double *__restrict__ tau, int n, int k, double *vp, double *wp,
                                          double *red) {
    const int tid = threadIdx.x;
    if (k == 0) {
        for (int j = tid; j < n; j += 256) { vp[j] = 0.0; wp[j] = 0.0; }
        wc_sync();
        return;
    }
    const double tp = tau[k - 1];
    const double *pv = P + (size_t)((k - 1) & 1) * n;
    const double *vr = V + (size_t)(k - 1) * n;
    double acc = 0.0;
    for (int j = tid; j < n; j += 256) {
        const double v = vr[j];
        const double p = j >= k ? pv[j] : 0.0;      // rows below k were not part of that product
        vp[j] = v;
        wp[j] = p;
        acc = fma(p, v, acc);
    }
    const double K = 0.5 * tp * block_sum256(acc, red);
    for (int j = tid; j < n; j += 256) wp[j] = wp[j] - K * vp[j];
    wc_sync();
}

// One column of the tridiagonalisation.  A is the full symmetric matrix (row k serves as column k),
// rows i > k are owned by waves (i - k - 1) mod (4 gridDim.x).
__global__ __launch_bounds__(256) void k_tri_step(double *__restrict__ A, double *__restrict__ V,
                                                  double *__restrict__ P, double *__restrict__ d,
                                                  double *__restrict__ e, double *__restrict__ tau, int n, int k) {
    extern __shared__ double lds[];
    __shared__ double red[4];
    double *vp = lds, *wp = lds + n, *vc = lds + 2 * (size_t)n;
    const int tid = threadIdx.x;
    previous_reflector(V, P, tau, n, k, vp, wp, red);
    // row k with the pending update applied: the new column
    const double vpk = vp[k], wpk = wp[k];
    const double *rowk = A + (size_t)k * n;
    double ss = 0.0;
    for (int j = tid; j < n; j += 256) {
        double r = 0.0;
        if (j >= k) r = rowk[j] - (vpk * wp[j] + wpk * vp[j]);
        if (j == k && blockIdx.x == 0) d[k] = r;
        r = j > k ? r : 0.0;
        vc[j] = r;
        ss = fma(r, r, ss);
    }
    ss = block_sum256(ss, red);
    const double x0 = vc[k + 1];
    double alpha = 0.0, t = 0.0, v0 = x0;
    if (ss > 0.0 && isfinite(ss)) {
        const double nrm = sqrt(ss);
        alpha = -copysign(nrm, x0);
        v0 = x0 - alpha;
        t = 2.0 / ((ss - x0 * x0) + v0 * v0);
    }
    wc_sync();
    if (tid == 0) vc[k + 1] = v0;
    wc_sync();
    if (blockIdx.x == 0) {
        double *vrow = V + (size_t)k * n;
        for (int j = tid; j < n; j += 256) vrow[j] = vc[j];
        if (tid == 0) { e[k] = alpha; tau[k] = t; }
    }
    // trailing rows: pending update, then the product with the new reflector
    const int wave = tid >> 6, lane = tid & 63;
    const int nw = (int)gridDim.x * 4;
    double *Pc = P + (size_t)(k & 1) * n;
    for (int i = k + 1 + (int)blockIdx.x * 4 + wave; i < n; i += nw) {
        const double vpi = vp[i], wpi = wp[i];
        double *row = A + (size_t)i * n;
        double acc = 0.0;
        for (int j = k + 1 + lane; j < n; j += 64) {
            const double a = row[j] - (vpi * wp[j] + wpi * vp[j]);
            row[j] = a;
            acc = fma(a, vc[j], acc);
        }
        acc = wave_sum(acc);
        if (lane == 0) Pc[i] = t * acc;
    }
}

// The same column step for n <= 1024 with every global load of the launch issued up front: the
// previous reflector and its product, row k, and the (at most two) trailing rows of each wave sit
// in registers before the first reduction starts, so the step is ONE memory round trip followed by
// arithmetic instead of three round trips in a row (8.7 -> ~4 us per column at 600 samples).
// The host launches ceil((n-k-1)/8) workgroups: two rows per wave at most.
__global__ __launch_bounds__(256) void k_tri_step_reg(double *__restrict__ A, double *__restrict__ V,
                                                      double *__restrict__ P, double *__restrict__ d,
                                                      double *__restrict__ e, double *__restrict__ tau, int n, int k) {
    constexpr int NQ = 4, PER = 16;
    extern __shared__ double lds[];
    __shared__ double red[4];
    double *vp = lds, *wp = lds + n, *vc = lds + 2 * (size_t)n;
    const int tid = threadIdx.x, wave = tid >> 6, lane = tid & 63;
    const int nw = (int)gridDim.x * 4;
    const int i0 = k + 1 + (int)blockIdx.x * 4 + wave, i1 = i0 + nw;
    double pj[NQ], vj[NQ], rk[NQ], r0[PER], r1[PER];
    const double tp = k > 0 ? tau[k - 1] : 0.0;
    const double *pv = P + (size_t)((k + 1) & 1) * n;        // (k-1) & 1
    const double *vr = V + (size_t)(k > 0 ? k - 1 : 0) * n;
    const double *rowk = A + (size_t)k * n;
#pragma unroll
    for (int q = 0; q < NQ; ++q) {
        const int j = tid + 256 * q;
        vj[q] = (k > 0 && j < n) ? vr[j] : 0.0;
        pj[q] = (k > 0 && j >= k && j < n) ? pv[j] : 0.0;
        rk[q] = (j >= k && j < n) ? rowk[j] : 0.0;
    }
    double *row0 = A + (size_t)(i0 < n ? i0 : 0) * n, *row1 = A + (size_t)(i1 < n ? i1 : 0) * n;
#pragma unroll
    for (int q = 0; q < PER; ++q) {
        const int j = k + 1 + lane + 64 * q;
        r0[q] = (i0 < n && j < n) ? row0[j] : 0.0;
        r1[q] = (i1 < n && j < n) ? row1[j] : 0.0;
    }
    __builtin_amdgcn_sched_barrier(0);      // (the loads stay in front of the reductions)
    // previous reflector: w = p - (tau/2 p.v) v
    double acc = 0.0;
#pragma unroll
    for (int q = 0; q < NQ; ++q) acc = fma(pj[q], vj[q], acc);
    const double K = 0.5 * tp * block_sum256(acc, red);
#pragma unroll
    for (int q = 0; q < NQ; ++q) {
        const int j = tid + 256 * q;
        pj[q] = pj[q] - K * vj[q];          // now w
        if (j < n) { vp[j] = vj[q]; wp[j] = pj[q]; }
    }
    wc_sync();
    // row k with the pending update applied: the new column
    const double vpk = vp[k], wpk = wp[k];
    double ss = 0.0;
#pragma unroll
    for (int q = 0; q < NQ; ++q) {
        const int j = tid + 256 * q;
        double r = j >= k ? fma(-wpk, vj[q], fma(-vpk, pj[q], rk[q])) : 0.0;
        if (j == k && blockIdx.x == 0) d[k] = r;
        r = j > k ? r : 0.0;
        if (j < n) vc[j] = r;
        ss = fma(r, r, ss);
    }
    ss = block_sum256(ss, red);
    const double x0 = vc[k + 1];
    double alpha = 0.0, t = 0.0, v0 = x0;
    if (ss > 0.0 && isfinite(ss)) {
        const double nrm = sqrt(ss);
        alpha = -copysign(nrm, x0);
        v0 = x0 - alpha;
        t = 2.0 / ((ss - x0 * x0) + v0 * v0);
    }
    wc_sync();
    if (tid == 0) vc[k + 1] = v0;
    wc_sync();
    if (blockIdx.x == 0) {
        double *vrow = V + (size_t)k * n;
        for (int j = tid; j < n; j += 256) vrow[j] = vc[j];
        if (tid == 0) { e[k] = alpha; tau[k] = t; }
    }
    // both rows of the wave together, four partial dot products each: the chains of dependent
    // float64 operations are what this part costs
    double *Pc = P + (size_t)(k & 1) * n;
    const double vp0 = i0 < n ? vp[i0] : 0.0, wp0 = i0 < n ? wp[i0] : 0.0;
    const double vp1 = i1 < n ? vp[i1] : 0.0, wp1 = i1 < n ? wp[i1] : 0.0;
    double dot0[4] = {0.0, 0.0, 0.0, 0.0}, dot1[4] = {0.0, 0.0, 0.0, 0.0};
    // float64 instructions are what this part costs (a wave issues one every ~8 ns here): the
    // update is two fused multiply-adds per element, and the 64-column groups beyond the end of
    // the row are skipped by a wave-uniform branch instead of being issued under an empty mask
    const int nq = (n - k - 1 + 63) >> 6;
#pragma unroll
    for (int q = 0; q < PER; ++q) {
        if (q < nq) {
            const int j = k + 1 + lane + 64 * q;
            if (j < n) {
                const double wj = wp[j], vj2 = vp[j], cj = vc[j];
                const double a0 = fma(-wp0, vj2, fma(-vp0, wj, r0[q]));
                const double a1 = fma(-wp1, vj2, fma(-vp1, wj, r1[q]));
                if (i0 < n) row0[j] = a0;
                if (i1 < n) row1[j] = a1;
                dot0[q & 3] = fma(a0, cj, dot0[q & 3]);
                dot1[q & 3] = fma(a1, cj, dot1[q & 3]);
            }
        }
    }
    const double s0 = wave_sum((dot0[0] + dot0[1]) + (dot0[2] + dot0[3]));
    const double s1 = wave_sum((dot1[0] + dot1[1]) + (dot1[2] + dot1[3]));
    if (lane == 0) {
        if (i0 < n) Pc[i0] = t * s0;
        if (i1 < n) Pc[i1] = t * s1;
    }
}

// The last TAIL_MAX columns in ONE workgroup: once the trailing matrix fits into LDS (128 x 128
// float64 = 132 KB with the padding), a column costs a handful of workgroup barriers (~1.3 us)
// instead of a launch (5.8 us).  The pending update of the last multi-workgroup column is applied
// while the block is loaded; from there the block is kept up to date (product, then rank-2 update,
// both with eight threads per row).  Also writes the reflector rows, d, e, tau of its columns and
// the final 2 x 2 block.  Matrices of up to TAIL_MAX rows never see another kernel.
constexpr int TAIL_MAX = 128, TAIL_LD = TAIL_MAX + 1, TAIL_THREADS = 1024;

__device__ inline double block_sum1024(double v, double *red) {
    v = wave_sum(v);
    wc_sync();
    if ((threadIdx.x & 63) == 0) red[threadIdx.x >> 6] = v;
    wc_sync();
    double s = 0.0;
#pragma unroll
    for (int w = 0; w < TAIL_THREADS / 64; ++w) s += red[w];
    return s;
}

__global__ __launch_bounds__(TAIL_THREADS) void k_tri_tail(const double *__restrict__ A, double *__restrict__ V,
                                                           const double *__restrict__ P, double *__restrict__ d,
                                                           double *__restrict__ e, double *__restrict__ tau, int n,
                                                           int k0) {
    extern __shared__ double lds[];
    __shared__ double red[TAIL_THREADS / 64];
    double *M = lds, *v = M + TAIL_MAX * TAIL_LD, *w = v + TAIL_MAX, *pl = w + TAIL_MAX;
    double *es = pl + TAIL_MAX, *ts = es + TAIL_MAX;
    const int tid = threadIdx.x;
    const int mt = n - k0;                       // <= TAIL_MAX
    // the previous column's reflector and w = p - (tau/2 p.v) v, local indices
    {
        double vv = 0.0, pp = 0.0;
        if (k0 > 0 && tid < mt) {
            vv = V[(size_t)(k0 - 1) * n + k0 + tid];
            pp = P[(size_t)((k0 - 1) & 1) * n + k0 + tid];
        }
        const double tp = k0 > 0 ? tau[k0 - 1] : 0.0;
        const double K = 0.5 * tp * block_sum1024(pp * vv, red);
        if (tid < TAIL_MAX) { v[tid] = vv; w[tid] = pp - K * vv; }
        wc_sync();
    }
    for (int idx = tid; idx < mt * mt; idx += TAIL_THREADS) {
        const int r = idx / mt, c = idx - r * mt;
        M[r * TAIL_LD + c] = A[(size_t)(k0 + r) * n + k0 + c] - (v[r] * w[c] + w[r] * v[c]);
    }
    wc_sync();
    const int row = tid >> 3, part = tid & 7;    // eight threads per row of the block
    const int lane = tid & 63;
    for (int c = 0; c < mt - 2; ++c) {
        // the column: row c of the block right of the diagonal.  Every wave reduces it for itself
        // (two elements per lane): no workgroup barrier for a number all of them need
        const double xa = (lane > c && lane < mt) ? M[c * TAIL_LD + lane] : 0.0;
        const double xb = (lane + 64 > c && lane + 64 < mt) ? M[c * TAIL_LD + lane + 64] : 0.0;
        const double ss = wave_sum(xa * xa + xb * xb);
        const double x0 = M[c * TAIL_LD + c + 1];
        double alpha = 0.0, t = 0.0, v0 = x0;
        if (ss > 0.0 && isfinite(ss)) {
            const double nrm = sqrt(ss);
            alpha = -copysign(nrm, x0);
            v0 = x0 - alpha;
            t = 2.0 / ((ss - x0 * x0) + v0 * v0);
        }
        if (tid < 64) {
            v[lane] = lane == c + 1 ? v0 : xa;
            v[lane + 64] = lane + 64 == c + 1 ? v0 : xb;
        }
        if (tid == 0) { es[c] = alpha; ts[c] = t; }
        wc_sync();
        // the reflector replaces the row it came from (nobody reads row c any more); global memory
        // sees it after the loop -- a store inside the loop would sit in front of every barrier
        if (tid > c && tid < mt) M[c * TAIL_LD + tid] = v[tid];
        // p = t M v, eight threads per row, four partial sums each; columns left of the diagonal
        // hold zeros of v and are skipped in whole groups of eight
        const int q0 = (c + 1) >> 3;
        double acc[4] = {0.0, 0.0, 0.0, 0.0};
        if (row > c && row < mt) {
#pragma unroll
            for (int q = 0; q < TAIL_MAX / 8; ++q) {
                const int j = part + 8 * q;
                if (q >= q0 && j < mt) acc[q & 3] = fma(M[row * TAIL_LD + j], v[j], acc[q & 3]);
            }
        }
        double s = (acc[0] + acc[1]) + (acc[2] + acc[3]);
        s = dpp_add<0x111, 0xF>(s);
        s = dpp_add<0x112, 0xF>(s);
        s = dpp_add<0x114, 0xF>(s);               // lane 8 g + 7 holds the sum of its group of eight
        if (part == 7) pl[row] = (row > c && row < mt) ? t * s : 0.0;
        wc_sync();
        // K = t/2 p.v per wave again; w = p - K v by the first two waves' worth of threads
        const double K = 0.5 * t * wave_sum(pl[lane] * v[lane] + pl[lane + 64] * v[lane + 64]);
        if (tid < TAIL_MAX) w[tid] = fma(-K, v[tid], pl[tid]);
        wc_sync();
        if (row > c && row < mt) {
            const double vi = v[row], wi = w[row];
#pragma unroll
            for (int q = 0; q < TAIL_MAX / 8; ++q) {
                const int j = part + 8 * q;
                if (q >= q0 && j > c && j < mt)
                    M[row * TAIL_LD + j] = fma(-wi, v[j], fma(-vi, w[j], M[row * TAIL_LD + j]));
            }
        }
        wc_sync();
    }
    if (tid < mt) d[k0 + tid] = M[tid * TAIL_LD + tid];
    if (tid < mt - 2) { e[k0 + tid] = es[tid]; tau[k0 + tid] = ts[tid]; }
    if (tid == 0) e[k0 + mt - 2] = M[(mt - 1) * TAIL_LD + mt - 2];
    for (int c = 0; c < mt - 2; ++c)
        for (int j = tid; j < n; j += TAIL_THREADS)
            V[(size_t)(k0 + c) * n + j] = j > k0 + c ? M[c * TAIL_LD + j - k0] : 0.0;
}


// Eigenvalue number (n-1-blockIdx.x) in ascending order -- the blockIdx.x-th largest -- of the
// symmetric tridiagonal (d, e): multisection on the Sturm count (number of eigenvalues below x; the
// recurrence and its pivmin guard are dstebz's).  256 threads x 2 shifts = 512 points per round (one
// wave per SIMD: the recurrence is a chain of dependent float64 operations, more waves only
// queue for the same divider; the two recurrences of a thread overlap their latencies), until the
// bracket is 4 eps |T| wide: six rounds of n dependent steps.  The quotient e^2 / q is e^2 times a
// reciprocal refined by two Newton steps (a few ulps, which the count tolerates like any rounding
// of the recurrence); |q| >= pivmin keeps it finite.
constexpr int EV_THREADS = 256, EV_POINTS = 2 * EV_THREADS;
__device__ inline double recip2(double q) {
    double r = __builtin_amdgcn_rcp(q);
    r = fma(fma(-q, r, 1.0), r, r);
    r = fma(fma(-q, r, 1.0), r, r);
    return r;
}
__global__ __launch_bounds__(EV_THREADS) void k_tri_eigvals(const double *__restrict__ d, const double *__restrict__ e,
                                                            int n, double *__restrict__ evals,
                                                            double *__restrict__ tnorm_out) {
    extern __shared__ double lds[];
    __shared__ double red[EV_THREADS / 64], red2[EV_THREADS / 64], red3[EV_THREADS / 64];
    __shared__ int best[EV_THREADS / 64];
    double *sd = lds, *se2 = lds + n;
    const int tid = threadIdx.x;
    double gl = INFINITY, gu = -INFINITY, emax = 0.0;
    for (int i = tid; i < n; i += EV_THREADS) {
        const double di = d[i];
        const double el = i ? fabs(e[i - 1]) : 0.0, er = i < n - 1 ? fabs(e[i]) : 0.0;
        const double e2 = i < n - 1 ? e[i] * e[i] : 0.0;
        sd[i] = di;
        se2[i] = e2;
        gl = fmin(gl, di - el - er);
        gu = fmax(gu, di + el + er);
        emax = fmax(emax, e2);
    }
    for (int o = 32; o > 0; o >>= 1) {
        gl = fmin(gl, __shfl_xor(gl, o));
        gu = fmax(gu, __shfl_xor(gu, o));
        emax = fmax(emax, __shfl_xor(emax, o));
    }
    if ((tid & 63) == 0) { red[tid >> 6] = gl; red2[tid >> 6] = gu; red3[tid >> 6] = emax; }
    wc_sync();
    for (int w = 0; w < EV_THREADS / 64; ++w) {
        gl = fmin(gl, red[w]);
        gu = fmax(gu, red2[w]);
        emax = fmax(emax, red3[w]);
    }
    const double tn = fmax(fabs(gl), fabs(gu));
    const double pivmin = DBL_MIN * fmax(1.0, emax);
    double lo = gl - (2.0 * tn * EPS * n + 2.0 * pivmin), hi = gu + (2.0 * tn * EPS * n + 2.0 * pivmin);
    if (blockIdx.x == 0 && tid == 0) *tnorm_out = tn;
    const int target = n - 1 - (int)blockIdx.x;
    for (int round = 0; round < 12; ++round) {
        const double width = hi - lo;
        const double xa = lo + width * ((double)(2 * tid + 1) / (double)(EV_POINTS + 1));
        const double xb = lo + width * ((double)(2 * tid + 2) / (double)(EV_POINTS + 1));
        double qa = sd[0] - xa, qb = sd[0] - xb;
        if (fabs(qa) < pivmin) qa = -pivmin;
        if (fabs(qb) < pivmin) qb = -pivmin;
        int ca = qa < 0.0 ? 1 : 0, cb = qb < 0.0 ? 1 : 0;
        for (int i = 1; i < n; ++i) {
            const double di = sd[i], e2 = se2[i - 1];
            qa = di - xa - e2 * recip2(qa);
            qb = di - xb - e2 * recip2(qb);
            if (fabs(qa) < pivmin) qa = -pivmin;
            if (fabs(qb) < pivmin) qb = -pivmin;
            ca += qa < 0.0 ? 1 : 0;
            cb += qb < 0.0 ? 1 : 0;
        }
        // the last point whose count does not exceed the target: the eigenvalue is at or above it
        int mine = cb <= target ? 2 * tid + 1 : (ca <= target ? 2 * tid : -1);
        for (int o = 32; o > 0; o >>= 1) mine = max(mine, __shfl_xor(mine, o));
        wc_sync();
        if ((tid & 63) == 0) best[tid >> 6] = mine;
        wc_sync();
        int a = -1;
        for (int w = 0; w < EV_THREADS / 64; ++w) a = max(a, best[w]);
        const double nlo = a >= 0 ? lo + width * ((double)(a + 1) / (double)(EV_POINTS + 1)) : lo;
        const double nhi = a + 1 < EV_POINTS ? lo + width * ((double)(a + 2) / (double)(EV_POINTS + 1)) : hi;
        lo = nlo;
        hi = nhi;
        if (!(hi - lo > fmax(4.0 * EPS * tn, 2.0 * EPS * fmax(fabs(lo), fabs(hi))))) break;
    }
    if (tid == 0) evals[blockIdx.x] = 0.5 * (lo + hi);
}

// The sequential recurrences of inverse iteration, run by one lane.  The operands of CH steps are
// fetched together before the dependent chain of those steps starts: the chain then waits for
// arithmetic only (a load per step in the chain cost 0.6 ms at 600 samples, this form 0.1 ms).
constexpr int CH = 8;

// T - lam I = P L U with partial pivoting (U: diagonal u0 and two superdiagonals u1, u2; L: the
// multipliers l; sw: rows i and i+1 were exchanged).  A zero pivot becomes `tiny`.
__device__ inline void tri_factor(const double *dd, const double *ee, int n, double lam, double tiny, double *u0,
                                  double *u1, double *u2, double *l, double *sw) {
    double ai = dd[0] - lam, bi = n > 1 ? ee[0] : 0.0;
    for (int i0 = 0; i0 < n - 1; i0 += CH) {
        double C[CH], AN[CH], BN[CH];
#pragma unroll
        for (int q = 0; q < CH; ++q) {
            const int i = i0 + q;
            C[q] = i < n - 1 ? ee[i] : 0.0;
            AN[q] = i < n - 1 ? dd[i + 1] - lam : 0.0;
            BN[q] = i + 1 < n - 1 ? ee[i + 1] : 0.0;
        }
#pragma unroll
        for (int q = 0; q < CH; ++q) {
            const int i = i0 + q;
            if (i >= n - 1) break;
            if (fabs(ai) >= fabs(C[q])) {
                if (ai == 0.0) ai = tiny;
                const double m = C[q] / ai;
                u0[i] = ai; u1[i] = bi; u2[i] = 0.0; l[i] = m; sw[i] = 0.0;
                ai = AN[q] - m * bi;
                bi = BN[q];
            } else {
                const double m = ai / C[q];
                u0[i] = C[q]; u1[i] = AN[q]; u2[i] = BN[q]; l[i] = m; sw[i] = 1.0;
                ai = bi - m * AN[q];
                bi = -m * BN[q];
            }
        }
    }
    if (fabs(ai) < tiny) ai = copysign(tiny, ai);
    u0[n - 1] = ai;
}

// y = L^-1 P z
__device__ inline void tri_lsolve(int n, const double *l, const double *sw, const double *z, double *y) {
    double yc = z[0];
    for (int i0 = 0; i0 < n - 1; i0 += CH) {
        double L[CH], S[CH], ZN[CH];
#pragma unroll
        for (int q = 0; q < CH; ++q) {
            const int i = i0 + q;
            L[q] = i < n - 1 ? l[i] : 0.0;
            S[q] = i < n - 1 ? sw[i] : 0.0;
            ZN[q] = i < n - 1 ? z[i + 1] : 0.0;
        }
#pragma unroll
        for (int q = 0; q < CH; ++q) {
            const int i = i0 + q;
            if (i >= n - 1) break;
            double nxt = ZN[q];
            if (S[q] != 0.0) { const double t = yc; yc = nxt; nxt = t; }
            nxt -= L[q] * yc;
            y[i] = yc;
            yc = nxt;
        }
    }
    y[n - 1] = yc;
}

// z = U^-1 y
__device__ inline void tri_usolve(int n, const double *u0, const double *u1, const double *u2, const double *y,
                                  double *z) {
    double z1 = y[n - 1] / u0[n - 1], z2 = 0.0;
    z[n - 1] = z1;
    for (int i0 = n - 2; i0 >= 0; i0 -= CH) {
        double Yv[CH], U1[CH], U2[CH], R[CH];
#pragma unroll
        for (int q = 0; q < CH; ++q) {
            const int i = i0 - q;
            Yv[q] = i >= 0 ? y[i] : 0.0;
            U1[q] = i >= 0 ? u1[i] : 0.0;
            U2[q] = i >= 0 ? u2[i] : 0.0;
            R[q] = i >= 0 ? 1.0 / u0[i] : 0.0;        // (off the dependent chain)
        }
#pragma unroll
        for (int q = 0; q < CH; ++q) {
            const int i = i0 - q;
            if (i < 0) break;
            const double zi = (Yv[q] - U1[q] * z1 - U2[q] * z2) * R[q];
            z[i] = zi;
            z2 = z1;
            z1 = zi;
        }
    }
}

// Eigenvectors of the tridiagonal matrix for the eigenvalues of k_tri_eigvals: wave c factors
// T - lambda_c I and runs inverse iteration from a fixed pseudo-random vector: one U-solve, then
// two full solves, each followed by Gram-Schmidt against the vectors before it (the wanted
// eigenvalues may sit 1e-5 |T| apart) and normalisation.  Work arrays: 7 n doubles per vector plus
// a copy of (d, e) -- in LDS when they fit (IN_LDS; a template parameter so that the accesses are
// LDS instructions and not flat ones, whose trip through the vector-memory front end is several
// times slower), else in `work_global` / the inputs themselves.
template <bool IN_LDS>
__global__ __launch_bounds__(512) void k_tri_eigvecs(const double *__restrict__ d, const double *__restrict__ e, int n,
                                                     int ncomp, const double *__restrict__ evals,
                                                     const double *__restrict__ tnorm, double *__restrict__ work_global,
                                                     double *__restrict__ Y) {
    extern __shared__ double lds[];
    const int c = threadIdx.x >> 6, lane = threadIdx.x & 63;
    double *base = work_global;
    const double *dd = d, *ee = e;
    if constexpr (IN_LDS) {
        for (int i = threadIdx.x; i < n; i += blockDim.x) { lds[i] = d[i]; lds[n + i] = i < n - 1 ? e[i] : 0.0; }
        dd = lds;
        ee = lds + n;
        base = lds + 2 * (size_t)n;
    }
    double *u0 = base + (size_t)c * 7 * n, *u1 = u0 + n, *u2 = u1 + n, *l = u2 + n, *sw = l + n, *z = sw + n, *y = z + n;
    const double lam = evals[c];
    const double tiny = fmax(EPS * tnorm[0], DBL_MIN);
    for (int i = lane; i < n; i += 64) {
        // a fixed pseudo-random start in (-1, 1): the same for every run
        unsigned int h = (unsigned int)i * 2654435761u + (unsigned int)c * 40503u + 12345u;
        h ^= h >> 15; h *= 2246822519u; h ^= h >> 13; h *= 3266489917u; h ^= h >> 16;
        y[i] = (double)h * (2.0 / 4294967296.0) - 1.0;
    }
    wc_sync();
    if (lane == 0) tri_factor(dd, ee, n, lam, tiny, u0, u1, u2, l, sw);
    for (int it = 0; it < 3; ++it) {
        if (lane == 0) {
            if (it > 0) tri_lsolve(n, l, sw, z, y);
            tri_usolve(n, u0, u1, u2, y, z);
        }
        // Gram-Schmidt in the order of the eigenvalues, then unit length
        for (int j = 0; j < ncomp; ++j) {
            wc_sync();
            if (c != j) continue;
            for (int p = 0; p < j; ++p) {
                const double *zp = base + (size_t)p * 7 * n + 5 * (size_t)n;
                double dot = 0.0;
                for (int i = lane; i < n; i += 64) dot = fma(zp[i], z[i], dot);
                dot = wave_sum(dot);
                for (int i = lane; i < n; i += 64) z[i] -= dot * zp[i];
            }
            double big = 0.0;
            for (int i = lane; i < n; i += 64) big = fmax(big, fabs(z[i]));
            for (int o = 32; o > 0; o >>= 1) big = fmax(big, __shfl_xor(big, o));
            const double sc = big > 0.0 && isfinite(big) ? 1.0 / big : 1.0;   // (no overflow in the squares)
            double nn = 0.0;
            for (int i = lane; i < n; i += 64) { const double v = z[i] * sc; nn = fma(v, v, nn); }
            nn = wave_sum(nn);
            const double inv = nn > 0.0 ? sc / sqrt(nn) : 0.0;
            for (int i = lane; i < n; i += 64) z[i] *= inv;
        }
        wc_sync();
    }
    for (int i = lane; i < n; i += 64) Y[(size_t)c * n + i] = z[i];
}

// z = H_0 H_1 ... H_{n-3} y for one vector per workgroup: NT threads hold PER elements of z each,
// the reflector rows stream through registers one step ahead of the dot product that needs them.
template <int NT, int PER>
__global__ __launch_bounds__(NT) void k_tri_back(const double *__restrict__ V, const double *__restrict__ tau,
                                                 const double *__restrict__ Y, int n, double *__restrict__ out) {
    __shared__ double part[2][NT / 64];
    const int tid = threadIdx.x, c = blockIdx.x;
    double z[PER], v[PER], vn[PER];
#pragma unroll
    for (int m = 0; m < PER; ++m) {
        const int j = tid + m * NT;
        z[m] = j < n ? Y[(size_t)c * n + j] : 0.0;
        v[m] = (j < n && n >= 3) ? V[(size_t)(n - 3) * n + j] : 0.0;
        vn[m] = 0.0;
    }
    double tk = n >= 3 ? tau[n - 3] : 0.0;
    for (int k = n - 3; k >= 0; --k) {
        double tn = 0.0;
        if (k > 0) {
            tn = tau[k - 1];
#pragma unroll
            for (int m = 0; m < PER; ++m) {
                const int j = tid + m * NT;
                vn[m] = j < n ? V[(size_t)(k - 1) * n + j] : 0.0;
            }
        }
        double s = 0.0;
#pragma unroll
        for (int m = 0; m < PER; ++m) s = fma(v[m], z[m], s);
        s = wave_sum(s);
        if ((tid & 63) == 0) part[k & 1][tid >> 6] = s;
        wc_sync();
        double tot = 0.0;
#pragma unroll
        for (int w = 0; w < NT / 64; ++w) tot += part[k & 1][w];
        tot *= tk;
#pragma unroll
        for (int m = 0; m < PER; ++m) {
            z[m] -= tot * v[m];
            v[m] = vn[m];
        }
        tk = tn;
    }
#pragma unroll
    for (int m = 0; m < PER; ++m) {
        const int j = tid + m * NT;
        if (j < n) out[(size_t)c * n + j] = z[m];
    }
}

template <class K>
int allow_lds(K kernel, size_t bytes) {
    if (bytes <= 48 * 1024) return WC_OK;
    WC_HIP(hipFuncSetAttribute(reinterpret_cast<const void *>(kernel), hipFuncAttributeMaxDynamicSharedMemorySize,
                               (int)bytes));
    return WC_OK;
}

}  // namespace

namespace wc {

// Leading n_pairs eigenpairs (eigenvalues descending, unit eigenvectors as rows) of the symmetric
// n x n matrix at `matrix_dev` (left untouched).  Outputs on the host.  3 <= n <= 4096, n_pairs <= 8.
int sym_eigh_leading(wc_ctx *ctx, const double *matrix_dev, int64_t n64, int n_pairs, double *eigvals_out,
                     double *eigvecs_out) {
    WC_CHECK(n64 >= 3 && n64 <= 4096, WC_E_LIMIT, "eigh: order %lld outside 3..4096", (long long)n64);
    WC_CHECK(n_pairs >= 1 && n_pairs <= 8 && n_pairs <= n64, WC_E_ARG, "eigh: 1..8 pairs supported");
    const int n = (int)n64;
    const size_t nn = (size_t)n * n;
    wc::DevBuf &ws = ctx->prep.eig_ws;
    int rc;
    if ((rc = ws.reserve(sizeof(double) * (2 * nn + 5 * (size_t)n + 16 + 16 * (size_t)n + 8 * 7 * (size_t)n)))) return rc;
    double *A = ws.as<double>(), *V = A + nn, *P = V + nn, *d = P + 2 * (size_t)n, *e = d + n, *tau = e + n;
    double *ev = tau + n, *tnorm = ev + 8, *Y = ev + 16, *Z = Y + 8 * (size_t)n, *W = Z + 8 * (size_t)n;
    hipStream_t stream = nullptr;
    {
        const int nparts = (int)std::min<size_t>(256, (nn + 4095) / 4096);
        hipLaunchKernelGGL(k_eig_absmax, dim3((unsigned)nparts), dim3(256), 0, stream, matrix_dev, nn, W);
        hipLaunchKernelGGL(k_eig_scaled_copy, dim3((unsigned)std::min<size_t>(1024, (nn + 255) / 256)), dim3(256), 0, stream,
                           matrix_dev, nn, (const double *)W, nparts, A, tnorm + 1);
    }
    const size_t lds_step = sizeof(double) * 3 * (size_t)n;
    if ((rc = allow_lds(k_tri_step, lds_step))) return rc;
    if ((rc = allow_lds(k_tri_step_reg, lds_step))) return rc;
    // the last TAIL_MAX rows in one workgroup
    const int k_split = std::max(0, n - TAIL_MAX);
    for (int k = 0; k < k_split; ++k) {
        const int m = n - k - 1;
        const unsigned grid = (unsigned)std::min(256, std::max(1, (m + 7) / 8));
        if (n <= 1024)
            hipLaunchKernelGGL(k_tri_step_reg, dim3(grid), dim3(256), lds_step, stream, A, V, P, d, e, tau, n, k);
        else
            hipLaunchKernelGGL(k_tri_step, dim3(grid), dim3(256), lds_step, stream, A, V, P, d, e, tau, n, k);
    }
    {
        const size_t lds_tail = sizeof(double) * ((size_t)TAIL_MAX * TAIL_LD + 5 * TAIL_MAX);
        if ((rc = allow_lds(k_tri_tail, lds_tail))) return rc;
        hipLaunchKernelGGL(k_tri_tail, dim3(1), dim3(TAIL_THREADS), lds_tail, stream, (const double *)A, V,
                           (const double *)P, d, e, tau, n, k_split);
    }
    const size_t lds_vals = sizeof(double) * 2 * (size_t)n;
    if ((rc = allow_lds(k_tri_eigvals, lds_vals))) return rc;
    hipLaunchKernelGGL(k_tri_eigvals, dim3((unsigned)n_pairs), dim3(EV_THREADS), lds_vals, stream, (const double *)d,
                       (const double *)e, n, ev, tnorm);
    const size_t lds_vecs = sizeof(double) * (7 * (size_t)n * n_pairs + 2 * (size_t)n);
    const int use_lds = lds_vecs <= 150 * 1024;
    if (use_lds) {
        if ((rc = allow_lds(k_tri_eigvecs<true>, lds_vecs))) return rc;
        hipLaunchKernelGGL(k_tri_eigvecs<true>, dim3(1), dim3(64 * (unsigned)n_pairs), lds_vecs, stream, (const double *)d,
                           (const double *)e, n, n_pairs, (const double *)ev, (const double *)tnorm, W, Y);
    } else {
        hipLaunchKernelGGL(k_tri_eigvecs<false>, dim3(1), dim3(64 * (unsigned)n_pairs), 0, stream, (const double *)d,
                           (const double *)e, n, n_pairs, (const double *)ev, (const double *)tnorm, W, Y);
    }
    if (n <= 1024)
        hipLaunchKernelGGL((k_tri_back<256, 4>), dim3((unsigned)n_pairs), dim3(256), 0, stream, (const double *)V,
                           (const double *)tau, (const double *)Y, n, Z);
    else
        hipLaunchKernelGGL((k_tri_back<1024, 4>), dim3((unsigned)n_pairs), dim3(1024), 0, stream, (const double *)V,
                           (const double *)tau, (const double *)Y, n, Z);
    WC_HIP(hipGetLastError());
    double head[16];
    WC_HIP(hipMemcpyAsync(head, ev, sizeof(double) * 16, hipMemcpyDeviceToHost, stream));
    WC_HIP(hipMemcpyAsync(eigvecs_out, Z, sizeof(double) * n_pairs * (size_t)n, hipMemcpyDeviceToHost, stream));
    WC_HIP(hipStreamSynchronize(stream));
    for (int c = 0; c < n_pairs; ++c) eigvals_out[c] = head[c] / head[9];       // (the scale is a power of two)
    for (int c = 0; c < n_pairs; ++c) {
        WC_CHECK(std::isfinite(eigvals_out[c]), WC_E_ARG, "eigh: the matrix holds non-finite values");
        double nrm = 0.0;
        for (int i = 0; i < n; ++i) nrm += eigvecs_out[(size_t)c * n + i] * eigvecs_out[(size_t)c * n + i];
        WC_CHECK(std::isfinite(nrm) && fabs(nrm - 1.0) < 1e-6, WC_E_ARG,
                 "eigh: eigenvector %d did not come out with unit length (%g): non-finite input?", c, nrm);
    }
    return WC_OK;
}

}  // namespace wc

extern "C" {

int wc_sym_eigh_leading_dev(wc_ctx *ctx, const double *matrix_dev, int64_t n, int n_pairs, double *eigvals_out,
                            double *eigvecs_out) {
    WC_CHECK(ctx && matrix_dev && eigvals_out && eigvecs_out, WC_E_ARG, "eigh: NULL argument");
    WC_HIP(hipSetDevice(ctx->device));
    return wc::sym_eigh_leading(ctx, matrix_dev, n, n_pairs, eigvals_out, eigvecs_out);
}

}  // extern "C"
