// Shared helpers for the gfx950 WISECONDOR kernels.
#pragma once
#include <hip/hip_runtime.h>
#include <atomic>
#include <stdint.h>
#include <stdio.h>
#include <string.h>
#include <math.h>
#include <float.h>
#include <string>
#include <vector>

#include "../../include/wisecondor_hip.h"

// The workgroup barrier of every kernel here: __syncthreads() with the wait of its release half spelled out.
// A barrier only orders LDS traffic if every wave's own DS operations have COMPLETED when it arrives (s_waitcnt
// lgkmcnt(0) in front of s_barrier); the compiler normally derives that wait from the fence inside __syncthreads(),
// but at the head of k_seg_walk's loop (ROCm 7.2, -O1 and -O3 alike) the listing shows a bare `s_barrier` behind back
// edges that carry thread 0's ds_write of the stack pointer: the other waves could read the OLD pointer after the
// barrier, take a different job and fall out of step with the workgroup's barriers for the rest of the kernel --
// round 5's "wrong medians in one region of 14 000, differently from run to run" (EXPERIMENTS.md, round 6;
// tools/barrier_scan.py checks every listing, tests/test_isa_cpu.py runs it).  The explicit wait costs one issue slot
// where the counter is already zero.  0xC07F = vmcnt(63) expcnt(7) lgkmcnt(0): global loads stay in flight across it.
#if defined(__HIPCC__)
__device__ __forceinline__ void wc_sync() {
    __builtin_amdgcn_s_waitcnt(0xC07F);
    __syncthreads();
}
__device__ __forceinline__ int wc_sync_or(int predicate) {
    __builtin_amdgcn_s_waitcnt(0xC07F);
    return __syncthreads_or(predicate);
}
#endif

namespace wc {

void set_error(const char *fmt, ...);

#define WC_HIP(call)                                                                       \
    do {                                                                                   \
        hipError_t e_ = (call);                                                            \
        if (e_ != hipSuccess) {                                                            \
            wc::set_error("%s failed: %s (%s:%d)", #call, hipGetErrorString(e_), __FILE__, \
                          __LINE__);                                                       \
            return WC_E_HIP;                                                               \
        }                                                                                  \
    } while (0)

#define WC_CHECK(cond, code, ...)      \
    do {                               \
        if (!(cond)) {                 \
            wc::set_error(__VA_ARGS__); \
            return code;               \
        }                              \
    } while (0)

// Counts the (re)allocations of every DevBuf of the process: a captured hipGraph bakes device
// addresses in, so whoever caches one compares this number before replaying it.
inline std::atomic<unsigned long long> &realloc_epoch() {
    static std::atomic<unsigned long long> epoch{0};
    return epoch;
}

// A grow-only device buffer; contexts keep these so repeated calls do not
// hipMalloc/hipFree (sized for 288 GB HBM: never shrinks, never spills).
struct DevBuf {
    void *p = nullptr;
    size_t bytes = 0;
    int reserve(size_t want) {
        if (want <= bytes) return WC_OK;
        ++realloc_epoch();
        if (p) (void)hipFree(p);
        p = nullptr;
        bytes = 0;
        size_t ask = want + (want >> 3) + 256;
        hipError_t e = hipMalloc(&p, ask);
        if (e != hipSuccess) {
            set_error("hipMalloc(%zu) failed: %s", ask, hipGetErrorString(e));
            return WC_E_HIP;
        }
        bytes = ask;
        return WC_OK;
    }
    void release() {
        ++realloc_epoch();
        if (p) (void)hipFree(p);
        p = nullptr;
        bytes = 0;
    }
    template <class T> T *as() const { return reinterpret_cast<T *>(p); }
};

// ---- order-preserving integer images of floats ---------------------------------
// NaN maps to the largest image so that it never wins a "smallest" selection.
__host__ __device__ inline uint32_t f32_ordered(float f) {
    if (f != f) return 0xFFFFFFFFu;
    uint32_t b;
    memcpy(&b, &f, 4);
    return (b & 0x80000000u) ? ~b : (b | 0x80000000u);
}
__host__ __device__ inline float f32_from_ordered(uint32_t u) {
    uint32_t b = (u & 0x80000000u) ? (u & 0x7FFFFFFFu) : ~u;
    float f;
    memcpy(&f, &b, 4);
    return f;
}
__host__ __device__ inline uint64_t f64_ordered(double d) {
    if (d != d) return ~0ull;
    uint64_t b;
    memcpy(&b, &d, 8);
    return (b & 0x8000000000000000ull) ? ~b : (b | 0x8000000000000000ull);
}
__host__ __device__ inline double f64_from_ordered(uint64_t u) {
    uint64_t b = (u & 0x8000000000000000ull) ? (u & 0x7FFFFFFFFFFFFFFFull) : ~u;
    double d;
    memcpy(&d, &b, 8);
    return d;
}

// ---- numpy pairwise summation ---------------------------------------------------
// numpy's add.reduce over a contiguous float64 run of n elements
// (loops_utils.h.src pairwise_sum): n < 8 sequential from 0; n <= 128 eight strided
// accumulators combined as ((r0+r1)+(r2+r3))+((r4+r5)+(r6+r7)) then the n%8 tail
// sequentially; larger n split at (n/2 - (n/2)%8) recursively.  The device code
// walks that tree with an explicit stack (no recursion) and evaluates each leaf
// either in one thread or across an aligned group of 8 lanes (lane j owns the
// elements congruent to j mod 8, which is exactly numpy's accumulator r[j]).
#define WC_PW_BLOCK 128
#define WC_PW_DEPTH 12

// Leaf evaluated by ONE thread; f(i) returns element i.
template <class F> __device__ inline double pw_leaf_serial(F f, int64_t off, int n) {
    if (n < 8) {
        double res = 0.0;
        for (int i = 0; i < n; ++i) res = res + f(off + i);
        return res;
    }
    double r0 = f(off + 0), r1 = f(off + 1), r2 = f(off + 2), r3 = f(off + 3);
    double r4 = f(off + 4), r5 = f(off + 5), r6 = f(off + 6), r7 = f(off + 7);
    int body = n - (n % 8);
    for (int i = 8; i < body; i += 8) {
        r0 = r0 + f(off + i + 0);
        r1 = r1 + f(off + i + 1);
        r2 = r2 + f(off + i + 2);
        r3 = r3 + f(off + i + 3);
        r4 = r4 + f(off + i + 4);
        r5 = r5 + f(off + i + 5);
        r6 = r6 + f(off + i + 6);
        r7 = r7 + f(off + i + 7);
    }
    double res = ((r0 + r1) + (r2 + r3)) + ((r4 + r5) + (r6 + r7));
    for (int i = body; i < n; ++i) res = res + f(off + i);
    return res;
}

// Leaf evaluated by an aligned group of 8 lanes (sub = lane & 7); every lane of the
// group returns the same value.  All 64 lanes of the wave must call this together.
template <class F> __device__ inline double pw_leaf_group8(F f, int64_t off, int n, int sub) {
    if (n < 8) {
        double res = 0.0;
        for (int i = 0; i < n; ++i) res = res + f(off + i);
        return res;
    }
    int body = n - (n % 8);
    double r = f(off + sub);
    int i = 8;
    // four loads in flight per lane; the adds stay in numpy's order
    for (; i + 24 < body; i += 32) {
        double v0 = f(off + i + sub), v1 = f(off + i + 8 + sub);
        double v2 = f(off + i + 16 + sub), v3 = f(off + i + 24 + sub);
        r = r + v0;
        r = r + v1;
        r = r + v2;
        r = r + v3;
    }
    for (; i < body; i += 8) r = r + f(off + i + sub);
    // (r0+r1), (r2+r3), ... : fp addition is commutative, so the xor butterfly
    // yields numpy's tree in every lane.
    r = r + __shfl_xor(r, 1);
    r = r + __shfl_xor(r, 2);
    r = r + __shfl_xor(r, 4);
    for (int i = body; i < n; ++i) r = r + f(off + i);
    return r;
}

// numpy's pairwise sum of a node of at most 128 * 2^D elements by ONE thread, the tree unrolled at compile time (no
// stack: a runtime-indexed private array would live in scratch memory)
template <int D, class F> __device__ inline double pw_node_serial(F f, int64_t off, int n) {
    if (n <= WC_PW_BLOCK) return pw_leaf_serial(f, off, n);
    if constexpr (D == 0) {
        return NAN;                                  // (not reached: the caller bounds n)
    } else {
        int n2 = n / 2;
        n2 -= n2 % 8;
        const double l = pw_node_serial<D - 1>(f, off, n2);
        const double r = pw_node_serial<D - 1>(f, off + n2, n - n2);
        return l + r;
    }
}

// ... and by aligned groups of eight lanes (every lane of the wave calls it with the same n; all return the value)
template <int D, class F> __device__ inline double pw_node_group8(F f, int64_t off, int n, int sub) {
    if (n <= WC_PW_BLOCK) return pw_leaf_group8(f, off, n, sub);
    if constexpr (D == 0) {
        return NAN;                                  // (not reached: the caller bounds n)
    } else {
        int n2 = n / 2;
        n2 -= n2 % 8;
        const double l = pw_node_group8<D - 1>(f, off, n2, sub);
        const double r = pw_node_group8<D - 1>(f, off + n2, n - n2, sub);
        return l + r;
    }
}

// numpy's pairwise sum of f(0 .. n - 1), 128 < n <= 8192, by a wave in which EVERY LANE FINDS ITS OWN NODE: with K the
// first level of numpy's tree whose largest node (the right-most: n - n2 >= n2) holds at most 128 elements, level
// K - 1 has 2^(K-1) <= 64 nodes; lane l descends from the root along the bits of l (most significant first), sums the
// node it arrives at ALONE (at most ~270 elements: a leaf or a split or two, pw_node_serial), and the levels are folded
// by a butterfly that keeps a subtree's value in the subtree's first lane (left + right, numpy's order).  A node that
// is a leaf before level K - 1 (the left-most nodes can be 8 per level smaller than the right-most) belongs to the
// first lane of its subtree; the other lanes of that subtree hold nothing and the fold skips them (adding +0.0 would
// turn a sum of -0.0 into +0.0).  No LDS, no stack: pairwise_tree_wave walks the tree twice through an LDS stack,
// 40 us for a 4 700-bin region; this form takes a few microseconds.  Every lane returns the value.
// GROUPS: the nodes are summed by the wave's eight 8-lane groups, eight nodes per round (pw_node_group8: one accumulator
// and four loads in flight per lane), instead of by their own lanes alone (eight accumulators per lane) -- for callers
// without the registers; ~3 x the time of the per-lane form, a fifth of the walk through the LDS stack.
template <bool GROUPS = false, class F> __device__ inline double pairwise_tree_lanes(F f, int n, int lane) {
    // (a node of level K - 1 holds at most 256 elements -- its right child, at most 128, is at least half of it -- so two
    //  more splits always reach leaves: pw_node_serial<2>)
    int K = 0;
    for (int r = n; r > WC_PW_BLOCK; ++K) {          // the right-most path
        int n2 = r / 2;
        n2 -= n2 % 8;
        r -= n2;
    }
    const int LV = K - 1, nodes = 1 << LV;            // K >= 1 (n > 128); nodes <= 64 for n <= 8192
    int off = 0, nn = n;
    bool valid = lane < nodes;
    for (int l = 0; l < LV; ++l) {
        if (nn <= WC_PW_BLOCK) {                      // a leaf above level LV: the subtree's first lane keeps it
            valid = valid && (lane & ((1 << (LV - l)) - 1)) == 0;
            break;
        }
        int n2 = nn / 2;
        n2 -= n2 % 8;
        if ((lane >> (LV - 1 - l)) & 1) { off += n2; nn -= n2; }
        else nn = n2;
    }
    double val = 0.0;
    if constexpr (!GROUPS) {
        val = valid ? pw_node_serial<2>(f, (int64_t)off, nn) : 0.0;
    } else {
        const int sub = lane & 7, grp = lane >> 3;
        for (int t = 0; 8 * t < nodes; ++t) {         // (nodes is the same in every lane)
            const int j = 8 * t + grp;                // this group's node of the round: lane j knows where it is
            const int joff = __shfl(off, j & 63), jn = __shfl(nn, j & 63);
            const bool jvalid = __shfl((int)valid, j & 63) != 0 && j < nodes;
            // (the shuffles inside stay within a group: groups may take different branches)
            const double v = pw_node_group8<2>(f, (int64_t)(jvalid ? joff : 0), jvalid ? jn : 0, sub);
            const double got = __shfl(v, 8 * (lane & 7));   // the sum of node 8 t + (lane & 7)
            if ((lane >> 3) == t) val = got;
        }
    }
    for (int st = 1; st < nodes; st <<= 1) {          // fold: subtrees of `st` lanes pair up
        const double right = __shfl(val, (lane + st) & 63);
        const bool rvalid = __shfl((int)valid, (lane + st) & 63) != 0;
        if ((lane & (2 * st - 1)) == 0) {
            if (valid && rvalid) val = val + right;
            else if (rvalid) { val = right; valid = true; }
        }
    }
    return __shfl(val, 0);
}

template <bool GROUP8, class F> __device__ inline double pairwise_tree(F f, int64_t n, int sub) {
    if (n <= WC_PW_BLOCK)
        return GROUP8 ? pw_leaf_group8(f, 0, (int)n, sub) : pw_leaf_serial(f, 0, (int)n);
    int64_t s_off[WC_PW_DEPTH], s_n[WC_PW_DEPTH];
    double s_left[WC_PW_DEPTH];
    int s_phase[WC_PW_DEPTH];
    int sp = 0;
    s_off[0] = 0; s_n[0] = n; s_phase[0] = 0; s_left[0] = 0.0; sp = 1;
    double result = 0.0;
    while (sp > 0) {
        int t = sp - 1;
        int64_t nn = s_n[t], off = s_off[t];
        if (nn <= WC_PW_BLOCK) {
            result = GROUP8 ? pw_leaf_group8(f, off, (int)nn, sub) : pw_leaf_serial(f, off, (int)nn);
            sp--;
        } else {
            int64_t n2 = nn / 2;
            n2 -= n2 % 8;
            if (s_phase[t] == 0) {
                s_phase[t] = 1;
                s_off[sp] = off; s_n[sp] = n2; s_phase[sp] = 0; sp++;
            } else if (s_phase[t] == 1) {
                s_left[t] = result;
                s_phase[t] = 2;
                s_off[sp] = off + n2; s_n[sp] = nn - n2; s_phase[sp] = 0; sp++;
            } else {
                result = s_left[t] + result;
                sp--;
            }
        }
    }
    return result;
}

// The same sum by a whole wave: numpy's tree is walked once to list the leaves (<= 128
// elements each), eight leaves at a time are summed by the eight 8-lane groups of the wave,
// and the leaf sums are folded in the tree's order.  `leaf` is LDS scratch of the calling
// wave: 3 * WC_PW_LEAVES ints worth of offsets / lengths plus WC_PW_LEAVES doubles.
// Every lane returns the value.  n up to WC_PW_LEAVES * 64 elements at least (leaves hold
// 65..128 elements); longer inputs fall back to the 8-lane walk.
#define WC_PW_LEAVES 256
struct PwWaveScratch {
    int off[WC_PW_LEAVES];
    int len[WC_PW_LEAVES];
    double sum[WC_PW_LEAVES];
    // the tree walk's stack lives here too: a runtime-indexed private array would go to scratch
    // memory (a global round trip per step); every lane reads and writes the same slots
    int s_off[WC_PW_DEPTH], s_n[WC_PW_DEPTH], s_phase[WC_PW_DEPTH];
    double s_left[WC_PW_DEPTH];
};
template <class F> __device__ inline double pairwise_tree_wave(F f, int64_t n, int lane, PwWaveScratch &sc) {
    const int sub = lane & 7, grp = lane >> 3;
    if (n <= WC_PW_BLOCK) return pw_leaf_group8(f, 0, (int)n, sub);
    if (n > (int64_t)WC_PW_LEAVES * 64) return pairwise_tree<true>(f, n, sub);
    // 1. leaves in tree (depth-first) order; every lane walks the same tree
    int *s_off = sc.s_off, *s_n = sc.s_n, *s_phase = sc.s_phase;
    int sp = 1, n_leaves = 0;
    s_off[0] = 0; s_n[0] = (int)n; s_phase[0] = 0;
    while (sp > 0) {
        const int t = sp - 1;
        const int nn = s_n[t], off = s_off[t];
        if (nn <= WC_PW_BLOCK) {
            if (lane == 0) { sc.off[n_leaves] = off; sc.len[n_leaves] = nn; }
            ++n_leaves;
            --sp;
        } else {
            int n2 = nn / 2;
            n2 -= n2 % 8;
            if (s_phase[t] == 0) {
                s_phase[t] = 1;
                s_off[sp] = off; s_n[sp] = n2; s_phase[sp] = 0; ++sp;
            } else if (s_phase[t] == 1) {
                s_phase[t] = 2;
                s_off[sp] = off + n2; s_n[sp] = nn - n2; s_phase[sp] = 0; ++sp;
            } else {
                --sp;
            }
        }
    }
    __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
    __builtin_amdgcn_wave_barrier();
    // 2. eight leaves per trip, one per 8-lane group
    for (int base = 0; base < n_leaves; base += 8) {
        const int l = base + grp;
        const bool mine = l < n_leaves;
        const double v = pw_leaf_group8(f, mine ? sc.off[l] : 0, mine ? sc.len[l] : 0, sub);
        if (mine && sub == 0) sc.sum[l] = v;
    }
    __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
    __builtin_amdgcn_wave_barrier();
    // 3. fold: the walk again, leaf values from the table
    double *s_left = sc.s_left;
    double result = 0.0;
    int next_leaf = 0;
    sp = 1;
    s_off[0] = 0; s_n[0] = (int)n; s_phase[0] = 0;
    while (sp > 0) {
        const int t = sp - 1;
        const int nn = s_n[t], off = s_off[t];
        if (nn <= WC_PW_BLOCK) {
            result = sc.sum[next_leaf++];
            --sp;
        } else {
            int n2 = nn / 2;
            n2 -= n2 % 8;
            if (s_phase[t] == 0) {
                s_phase[t] = 1;
                s_off[sp] = off; s_n[sp] = n2; s_phase[sp] = 0; ++sp;
            } else if (s_phase[t] == 1) {
                s_left[t] = result;
                s_phase[t] = 2;
                s_off[sp] = off + n2; s_n[sp] = nn - n2; s_phase[sp] = 0; ++sp;
            } else {
                result = s_left[t] + result;
                --sp;
            }
        }
    }
    __builtin_amdgcn_wave_barrier();          // the scratch may be reused by the caller's next sum
    return result;
}

// numpy's add.reduce itself: the reduction runs through numpy's iterator in pieces of its
// buffer size (8192 elements, np.getbufsize()), each piece summed pairwise as above and the
// piece sums accumulated left to right -- np.sum over a contiguous run of more than 8192
// values is NOT one pairwise tree (found with a 8193-bin region; 60 / 60 sizes up to 55 337
// agree with this rule on numpy 2.2.6).
#define WC_NPY_BUFSIZE 8192
template <bool GROUP8, class F> __device__ inline double pairwise_sum(F f, int64_t n, int sub) {
    if (n <= WC_NPY_BUFSIZE) return pairwise_tree<GROUP8>(f, n, sub);
    double res = 0.0;
    for (int64_t off = 0; off < n; off += WC_NPY_BUFSIZE) {
        const int64_t m = n - off < WC_NPY_BUFSIZE ? n - off : WC_NPY_BUFSIZE;
        res = res + pairwise_tree<GROUP8>([&](int64_t i) { return f(off + i); }, m, sub);
    }
    return res;
}
template <class F> __device__ inline double pairwise_sum_wave(F f, int64_t n, int lane, PwWaveScratch &sc) {
    if (n <= WC_NPY_BUFSIZE) return pairwise_tree_wave(f, n, lane, sc);
    double res = 0.0;
    for (int64_t off = 0; off < n; off += WC_NPY_BUFSIZE) {
        const int64_t m = n - off < WC_NPY_BUFSIZE ? n - off : WC_NPY_BUFSIZE;
        res = res + pairwise_tree_wave([&](int64_t i) { return f(off + i); }, m, lane, sc);
    }
    return res;
}

// Small per-lane value stack addressed by a wave-uniform index; the switch keeps the
// array in registers (a runtime subscript would send it to scratch memory).
__device__ inline void stack_set(double (&v)[10], int i, double x) {
    switch (i) {
        case 0: v[0] = x; break; case 1: v[1] = x; break; case 2: v[2] = x; break; case 3: v[3] = x; break;
        case 4: v[4] = x; break; case 5: v[5] = x; break; case 6: v[6] = x; break; case 7: v[7] = x; break;
        case 8: v[8] = x; break; default: v[9] = x; break;
    }
}
__device__ inline double stack_get(const double (&v)[10], int i) {
    switch (i) {
        case 0: return v[0]; case 1: return v[1]; case 2: return v[2]; case 3: return v[3]; case 4: return v[4];
        case 5: return v[5]; case 6: return v[6]; case 7: return v[7]; case 8: return v[8]; default: return v[9];
    }
}

}  // namespace wc
