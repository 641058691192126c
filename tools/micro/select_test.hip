// Stand-alone stress of block_select<256> (copied from testpath.hip by tools; not part of the library):
// every workgroup selects medians of many small arrays and checks them by counting.
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdint>
#include <vector>
namespace wc {
__device__ inline unsigned long long f64_ordered(double v) {
    unsigned long long b = (unsigned long long)__double_as_longlong(v);
    if (v != v) return ~0ull;
    return (b >> 63) ? ~b : (b | 0x8000000000000000ull);
}
__device__ inline double f64_from_ordered(unsigned long long k) {
    unsigned long long b = (k >> 63) ? (k & 0x7FFFFFFFFFFFFFFFull) : ~k;
    return __longlong_as_double((long long)b);
}
}
constexpr int CP_THREADS = 1024;
template <int NT = CP_THREADS>      // NT >= 256 threads
__device__ inline double block_select(const double *__restrict__ v, int L, int k, int tid) {
    // NT threads; the 256 digit buckets are scanned by the first four waves
    __shared__ unsigned int hist[256];
    __shared__ unsigned int s_wsum[4];
    __shared__ unsigned long long s_prefix;
    __shared__ int s_k;
    unsigned long long prefix = 0ull, mask = 0ull;
    for (int shift = 56; shift >= 0; shift -= 8) {
        if (tid < 256) hist[tid] = 0;
        __syncthreads();
        for (int e0 = tid & ~63; e0 < L; e0 += NT) {         // (wave-uniform trip count: the ballot below needs every lane)
            const int e = e0 + (tid & 63);
            const unsigned long long key = e < L ? wc::f64_ordered(v[e]) : 0ull;
            const bool in = e < L && (key & mask) == prefix;
            const unsigned int digit = (unsigned)(key >> shift) & 255u;
            // ratios share their leading bytes: a wave's lanes that hold the first lane's digit add their count at once
            // (4 000 single increments of ONE bucket were most of a long segment's selection)
            const unsigned long long live = __ballot(in);
            if (live) {
                const unsigned int d0 = (unsigned)__shfl((int)digit, __ffsll((long long)live) - 1);
                const unsigned long long same = __ballot(in && digit == d0);
                if (in && digit == d0) {
                    if ((tid & 63) == __ffsll((long long)same) - 1) atomicAdd(&hist[d0], (unsigned)__popcll(same));
                } else if (in) {
                    atomicAdd(&hist[digit], 1u);
                }
            }
        }
        __syncthreads();
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
        __syncthreads();
        if (tid < 256) {
            for (int q = 0; q < wv; ++q) incl += s_wsum[q];
            const unsigned int excl = incl - h;
            if ((unsigned int)k >= excl && (unsigned int)k < incl) {     // exactly one bucket (k < number of values)
                s_k = k - (int)excl;
                s_prefix = prefix | ((unsigned long long)tid << shift);
            }
        }
        __syncthreads();
        k = s_k;
        prefix = s_prefix;
        mask |= 0xFFull << shift;
        __syncthreads();
    }
    return wc::f64_from_ordered(prefix);
}

__global__ __launch_bounds__(256, 4) void k_test(const double *__restrict__ data, int n_arrays, int maxlen, int *__restrict__ bad, int reps) {
    __shared__ double sv[2048];
    __shared__ double pad[8];          // make the LDS footprint similar to the walker's
    const int tid = threadIdx.x;
    if (tid == 0) pad[0] = 0;
    for (int r = 0; r < reps; ++r)
        for (int a = blockIdx.x; a < n_arrays; a += gridDim.x) {
            const int L = 1 + (a * 7 + r) % maxlen;
            const double *v = data + (int64_t)a * maxlen;
            __syncthreads();
            for (int e = tid; e < L; e += 256) sv[e] = v[e];
            __syncthreads();
            const double lo = block_select<256>(sv, L, (L - 1) / 2, tid);
            const double hi = (L & 1) ? lo : block_select<256>(sv, L, L / 2, tid);
            if (tid == 0) {
                int lt = 0, le = 0, lt2 = 0, le2 = 0;
                for (int u = 0; u < L; ++u) { lt += sv[u] < lo; le += sv[u] <= lo; lt2 += sv[u] < hi; le2 += sv[u] <= hi; }
                const int k1 = (L - 1) / 2, k2 = L / 2;
                if (!(lt <= k1 && k1 < le) || !(lt2 <= k2 && k2 < le2)) atomicAdd(bad, 1);
            }
        }
}
int main() {
    const int n_arrays = 4000, maxlen = 1500, reps = 4;
    std::vector<double> h((size_t)n_arrays * maxlen);
    unsigned long long s = 88172645463325252ull;
    for (auto &x : h) { s ^= s << 13; s ^= s >> 7; s ^= s << 17; x = 0.6 + (double)(s >> 11) / 9007199254740992.0 * 0.8; }
    double *d; int *bad;
    hipMalloc(&d, h.size() * 8); hipMalloc(&bad, 4); hipMemset(bad, 0, 4);
    hipMemcpy(d, h.data(), h.size() * 8, hipMemcpyHostToDevice);
    for (int grid : {256, 1024, 4096}) {
        hipMemset(bad, 0, 4);
        hipLaunchKernelGGL(k_test, dim3(grid), dim3(256), 0, 0, d, n_arrays, maxlen, bad, reps);
        int hb = -1; hipMemcpy(&hb, bad, 4, hipMemcpyDeviceToHost);
        printf("grid %d: %d wrong medians of %d (%s)\n", grid, hb, n_arrays * reps, hipGetErrorString(hipGetLastError()));
    }
    return 0;
}
