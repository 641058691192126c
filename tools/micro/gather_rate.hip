// Micro-benchmark: how fast can a CU gather whole rows of a [rows, row_bytes] matrix that sits in L2 / Infinity Cache?
// One wave = one "bin": it reads n_ref rows (indexes from a list, wave-uniform) and adds them up.
//   mode 0: 8 B per lane, 64 lanes = 512 B of the row (k_zscore's access), 16 loads in flight
//   mode 1: 16 B per lane, 64 lanes = 1 KB of the row
//   mode 2: 16 B per lane through LDS-DMA (global_load_lds_dwordx4), then read back from LDS
//   mode 3: sample tiles dealt to XCDs -- workgroup w works on tile w % 8 (16 samples = one 128-byte line of every
//           row), a wave = 4 bins x 16 samples: each XCD only ever touches its own 128-byte column of the matrix
// build: hipcc --offload-arch=gfx950 -O3 -shared -fPIC -o wisecondor_amd/ab/libgather.so tools/micro/gather_rate.hip
#include <hip/hip_runtime.h>
#include <cstdint>

template <int MODE>
__global__ __launch_bounds__(256) void k_gather(const char *__restrict__ X, int64_t row_bytes, const int *__restrict__ idx,
                                                int n_ref, int n_bins, double *__restrict__ out) {
    const int wave = blockIdx.x * 4 + (threadIdx.x >> 6), lane = threadIdx.x & 63;
    const int b = __builtin_amdgcn_readfirstlane(wave);
    if (b >= n_bins) return;
    const int *lst = idx + (int64_t)b * n_ref;
    double acc = 0.0, acc2 = 0.0;
    if (MODE == 0) {
        for (int r0 = 0; r0 < n_ref; r0 += 16) {
            double v[16];
#pragma unroll
            for (int e = 0; e < 16; ++e) {
                const int g = lst[r0 + e];
                v[e] = *reinterpret_cast<const double *>(X + (int64_t)g * row_bytes + lane * 8);
            }
#pragma unroll
            for (int e = 0; e < 16; ++e) acc += v[e];
        }
    } else if (MODE == 1) {
        for (int r0 = 0; r0 < n_ref; r0 += 16) {
            double2 v[16];
#pragma unroll
            for (int e = 0; e < 16; ++e) {
                const int g = lst[r0 + e];
                v[e] = *reinterpret_cast<const double2 *>(X + (int64_t)g * row_bytes + lane * 16);
            }
#pragma unroll
            for (int e = 0; e < 16; ++e) { acc += v[e].x; acc2 += v[e].y; }
        }
    } else if (MODE == 3) {
        const int tile = blockIdx.x & 7, grp = blockIdx.x >> 3;                // (workgroups go to XCDs round robin)
        const int q = lane >> 4, sm = lane & 15;
        const int bin = (grp * 4 + (threadIdx.x >> 6)) * 4 + q;                // 4 waves x 4 bins per workgroup
        if (bin < n_bins) {
            const int *l4 = idx + (int64_t)bin * n_ref;
            for (int r0 = 0; r0 < n_ref; r0 += 16) {
                double v[16];
#pragma unroll
                for (int e = 0; e < 16; ++e) {
                    const int g = l4[r0 + e];
                    v[e] = *reinterpret_cast<const double *>(X + (int64_t)g * row_bytes + tile * 128 + sm * 8);
                }
#pragma unroll
                for (int e = 0; e < 16; ++e) acc += v[e];
            }
            out[(int64_t)bin * 64 + (tile * 16 + sm) % 64] = acc;
        }
        return;
    } else {
        extern __shared__ __attribute__((aligned(16))) char lds[];
        char *mine = lds + (threadIdx.x >> 6) * 16 * 1024;        // 16 rows of 1 KB per wave
        for (int r0 = 0; r0 < n_ref; r0 += 16) {
#pragma unroll
            for (int e = 0; e < 16; ++e) {
                const int g = lst[r0 + e];
                const char *src = X + (int64_t)g * row_bytes + lane * 16;
                __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void *)src,
                                                 (__attribute__((address_space(3))) void *)(mine + e * 1024), 16, 0, 0);
            }
            __builtin_amdgcn_s_waitcnt(0);                          // vmcnt(0) ...
            __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
#pragma unroll
            for (int e = 0; e < 16; ++e) {
                const double2 t = *reinterpret_cast<const double2 *>(mine + e * 1024 + lane * 16);
                acc += t.x; acc2 += t.y;
            }
        }
    }
    out[(int64_t)b * 64 + lane] = acc + acc2;
}

extern "C" int gather_run(int mode, const void *X, int64_t row_bytes, const int *idx, int n_ref, int n_bins, double *out,
                          void *stream) {
    const dim3 grid((n_bins + 3) / 4), block(256);
    hipStream_t s = (hipStream_t)stream;
    if (mode == 0) hipLaunchKernelGGL(k_gather<0>, grid, block, 0, s, (const char *)X, row_bytes, idx, n_ref, n_bins, out);
    else if (mode == 1) hipLaunchKernelGGL(k_gather<1>, grid, block, 0, s, (const char *)X, row_bytes, idx, n_ref, n_bins, out);
    else if (mode == 3) hipLaunchKernelGGL(k_gather<3>, dim3(8 * ((n_bins + 15) / 16)), block, 0, s, (const char *)X, row_bytes, idx, n_ref, n_bins, out);
    else hipLaunchKernelGGL(k_gather<2>, grid, block, 4 * 16 * 1024, s, (const char *)X, row_bytes, idx, n_ref, n_bins, out);
    return (int)hipGetLastError();
}
