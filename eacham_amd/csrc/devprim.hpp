// devprim.hpp — the two data-parallel primitives the device-side problem construction of the bundle adjuster is made of
// (ba.hip, eacham_ba_prepare): an exclusive prefix sum and a STABLE least-significant-digit radix sort of (key, value)
// pairs, hand-written for gfx950 (64-wide wavefronts: the in-tile ranks of the sort come from wave ballots).
//
// Why stable matters here: RefineBA's graph construction (modules/sfm/reconstruction/BundleAdjuster.cpp:57-178) visits
// landmarks and observers in a fixed order, and every fp64 sum of the solver runs in the order of these lists — a stable sort
// reproduces the order a sequential host loop produces, so the device-built structure is bit-identical with the host-built
// one (tests/test_ba_gpu.py holds them against each other) and two runs are bit-identical with each other.
//
// Everything is enqueued on the caller's stream; nothing synchronises. Workspace sizes are given by the *_ws_elems helpers.
#pragma once

#include <hip/hip_runtime.h>

#include <cstdint>

namespace eacham {
namespace prim {

struct I3 {  // three counters scanned together (entries, present blocks, chunks)
    int a, b, c;
};
__host__ __device__ inline I3 operator+(I3 x, I3 y) { return I3{x.a + y.a, x.b + y.b, x.c + y.c}; }
__host__ __device__ inline void zero(int& v) { v = 0; }
__host__ __device__ inline void zero(long long& v) { v = 0; }
__host__ __device__ inline void zero(I3& v) { v = I3{0, 0, 0}; }

constexpr int SCAN_THREADS = 256;
constexpr int SCAN_ITEMS = 8;
constexpr int SCAN_TILE = SCAN_THREADS * SCAN_ITEMS;  // elements per workgroup

// exclusive scan of the SCAN_THREADS per-thread sums of a workgroup (Hillis-Steele through LDS); returns this thread's
// prefix, *total = the workgroup's sum
template <class T>
__device__ __forceinline__ T block_exclusive(T v, T* lds /* [SCAN_THREADS] */, T* total) {
    const int tid = threadIdx.x;
    lds[tid] = v;
    __syncthreads();
    for (int off = 1; off < SCAN_THREADS; off <<= 1) {
        T u;
        zero(u);
        if (tid >= off) u = lds[tid - off];
        __syncthreads();
        lds[tid] = lds[tid] + u;
        __syncthreads();
    }
    const T all = lds[SCAN_THREADS - 1];
    T excl;
    zero(excl);
    if (tid > 0) excl = lds[tid - 1];
    __syncthreads();  // (the next call overwrites lds)
    if (total) *total = all;
    return excl;
}

template <class T>
__global__ __launch_bounds__(SCAN_THREADS) void scan_tile_sums(const T* __restrict__ in, int n, T* __restrict__ sums) {
    __shared__ T lds[SCAN_THREADS];
    const int base = blockIdx.x * SCAN_TILE + threadIdx.x * SCAN_ITEMS;
    T s;
    zero(s);
#pragma unroll
    for (int i = 0; i < SCAN_ITEMS; ++i)
        if (base + i < n) s = s + in[base + i];
    T total;
    (void)block_exclusive(s, lds, &total);
    if (threadIdx.x == 0) sums[blockIdx.x] = total;
}

// single workgroup: exclusive scan of the tile sums in place, grand total to *total (may be null)
template <class T>
__global__ __launch_bounds__(SCAN_THREADS) void scan_sums(T* __restrict__ sums, int ntiles, T* __restrict__ total) {
    __shared__ T lds[SCAN_THREADS];
    T carry;
    zero(carry);
    for (int t0 = 0; t0 < ntiles; t0 += SCAN_TILE) {
        const int base = t0 + threadIdx.x * SCAN_ITEMS;
        T v[SCAN_ITEMS], s;
        zero(s);
#pragma unroll
        for (int i = 0; i < SCAN_ITEMS; ++i) {
            zero(v[i]);
            if (base + i < ntiles) v[i] = sums[base + i];
            s = s + v[i];
        }
        T all;
        T run = carry + block_exclusive(s, lds, &all);
#pragma unroll
        for (int i = 0; i < SCAN_ITEMS; ++i) {
            if (base + i < ntiles) sums[base + i] = run;
            run = run + v[i];
        }
        carry = carry + all;
    }
    if (threadIdx.x == 0 && total) *total = carry;
}

// single workgroup, any (small) n: out[i] = exclusive prefix of in (out may alias in), grand total to *total (may be null).
// The three-launch form below costs three launch latencies (14 us) whatever n; up to SCAN_SMALL elements one workgroup is faster.
constexpr int SCAN_SMALL = 8 * SCAN_TILE;
template <class T>
__global__ __launch_bounds__(SCAN_THREADS) void scan_small(const T* __restrict__ in, T* __restrict__ out, int n, T* __restrict__ total) {
    __shared__ T lds[SCAN_THREADS];
    T carry;
    zero(carry);
    for (int t0 = 0; t0 < n; t0 += SCAN_TILE) {
        const int base = t0 + threadIdx.x * SCAN_ITEMS;
        T v[SCAN_ITEMS], s;
        zero(s);
#pragma unroll
        for (int i = 0; i < SCAN_ITEMS; ++i) {
            zero(v[i]);
            if (base + i < n) v[i] = in[base + i];
            s = s + v[i];
        }
        T all;
        T run = carry + block_exclusive(s, lds, &all);
#pragma unroll
        for (int i = 0; i < SCAN_ITEMS; ++i) {
            if (base + i < n) out[base + i] = run;
            run = run + v[i];
        }
        carry = carry + all;
    }
    if (threadIdx.x == 0 && total) *total = carry;
}

// out[i] = sums[tile] + exclusive prefix inside the tile (out may alias in); out[n] is NOT written (see `total`)
template <class T>
__global__ __launch_bounds__(SCAN_THREADS) void scan_apply(const T* __restrict__ in, T* __restrict__ out, int n,
                                                           const T* __restrict__ sums) {
    __shared__ T lds[SCAN_THREADS];
    const int base = blockIdx.x * SCAN_TILE + threadIdx.x * SCAN_ITEMS;
    T v[SCAN_ITEMS], s;
    zero(s);
#pragma unroll
    for (int i = 0; i < SCAN_ITEMS; ++i) {
        zero(v[i]);
        if (base + i < n) v[i] = in[base + i];
        s = s + v[i];
    }
    T run = sums[blockIdx.x] + block_exclusive(s, lds, (T*)nullptr);
#pragma unroll
    for (int i = 0; i < SCAN_ITEMS; ++i) {
        if (base + i < n) out[base + i] = run;
        run = run + v[i];
    }
}

inline size_t scan_ws_elems(size_t n) { return (n + SCAN_TILE - 1) / SCAN_TILE + 1; }

// out[0..n) = exclusive prefix sums of in[0..n); *total_dev (optional) = the sum. ws: scan_ws_elems(n) elements of T.
template <class T>
inline void exclusive_scan(hipStream_t st, const T* in, T* out, int n, T* ws, T* total_dev) {
    if (n <= 0) {
        if (total_dev) (void)hipMemsetAsync(total_dev, 0, sizeof(T), st);
        return;
    }
    if (n <= SCAN_SMALL) {
        scan_small<T><<<1, SCAN_THREADS, 0, st>>>(in, out, n, total_dev);
        return;
    }
    const int ntiles = (n + SCAN_TILE - 1) / SCAN_TILE;
    scan_tile_sums<T><<<ntiles, SCAN_THREADS, 0, st>>>(in, n, ws);
    scan_sums<T><<<1, SCAN_THREADS, 0, st>>>(ws, ntiles, total_dev);
    scan_apply<T><<<ntiles, SCAN_THREADS, 0, st>>>(in, out, n, ws);
}

// ---- stable LSD radix sort of (u32 key, value) pairs ---------------------------------------------------------------
// A SEGMENT is RS_SEG consecutive elements owned by ONE wave, which walks it in rounds of 64 in index order; a pass is
//   radix_hist     per segment: digit histogram (LDS atomics) -> hist[digit][segment]
//   exclusive_scan over hist (digit-major: every (digit, segment) gets the global position of its first element)
//   radix_scatter  per segment, round by round: lanes with the same digit find each other with `bits` ballots, the rank
//                  among them is a popcount below the lane, the lowest such lane advances the segment's running
//                  position of that digit in LDS — equal digits leave in index order: the pass is stable.
constexpr int RS_WAVES = 4;
constexpr int RS_THREADS = 64 * RS_WAVES;
constexpr int RS_ROUNDS = 8;    // (32 until round 5: 245 waves walking 32 dependent rounds each on S200 — 27 us per scatter of 500 k pairs; 8: 980 waves, 8 rounds)
constexpr int RS_SEG = 64 * RS_ROUNDS;  // elements per wave
constexpr int RS_MAX_BITS = 10;

__device__ __forceinline__ void wave_lds_fence() {
    __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
    __builtin_amdgcn_wave_barrier();
    __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
}

__global__ __launch_bounds__(RS_THREADS) void radix_hist(const uint32_t* __restrict__ keys, int n, int shift, int bits,
                                                         int nseg, int* __restrict__ hist) {
    __shared__ int cnt[RS_WAVES][1 << RS_MAX_BITS];
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const int seg = blockIdx.x * RS_WAVES + wave;
    const int rb = 1 << bits;
    for (int d = lane; d < rb; d += 64) cnt[wave][d] = 0;
    wave_lds_fence();
    if (seg < nseg) {
        const int base = seg * RS_SEG;
        for (int r = 0; r < RS_ROUNDS; ++r) {
            const int i = base + r * 64 + lane;
            if (i < n) atomicAdd(&cnt[wave][(keys[i] >> shift) & (rb - 1)], 1);
        }
        wave_lds_fence();
        for (int d = lane; d < rb; d += 64) hist[(size_t)d * nseg + seg] = cnt[wave][d];
    }
}

template <class V>
__global__ __launch_bounds__(RS_THREADS) void radix_scatter(const uint32_t* __restrict__ keys, const V* __restrict__ vals,
                                                            uint32_t* __restrict__ keys_out, V* __restrict__ vals_out, int n,
                                                            int shift, int bits, int nseg, const int* __restrict__ hist) {
    __shared__ int pos[RS_WAVES][1 << RS_MAX_BITS];
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const int seg = blockIdx.x * RS_WAVES + wave;
    const int rb = 1 << bits;
    if (seg >= nseg) return;  // wave-uniform; no workgroup barrier below
    for (int d = lane; d < rb; d += 64) pos[wave][d] = hist[(size_t)d * nseg + seg];
    wave_lds_fence();
    const int base = seg * RS_SEG;
    const unsigned long long below = (1ull << lane) - 1ull;
    for (int r = 0; r < RS_ROUNDS; ++r) {
        const int i = base + r * 64 + lane;
        const bool valid = i < n;
        uint32_t key = 0;
        V val{};
        if (valid) {
            key = keys[i];
            val = vals[i];
        }
        const int digit = (int)((key >> shift) & (uint32_t)(rb - 1));
        unsigned long long peers = __ballot(valid);
        for (int k = 0; k < bits; ++k) {
            const unsigned long long b = __ballot((digit >> k) & 1);
            peers &= ((digit >> k) & 1) ? b : ~b;
        }
        const int rank = __popcll(peers & below);
        const int leader = valid ? __ffsll((long long)peers) - 1 : lane;
        int first = 0;
        if (valid && rank == 0) {
            first = pos[wave][digit];
            pos[wave][digit] = first + __popcll(peers);
        }
        first = __shfl(first, leader);
        if (valid) {
            keys_out[first + rank] = key;
            vals_out[first + rank] = val;
        }
        wave_lds_fence();
    }
}

inline int radix_nseg(int n) { return (n + RS_SEG - 1) / RS_SEG; }
// ints of workspace for one sort of n elements: the histogram of the widest pass + its scan workspace
inline size_t radix_ws_ints(int n) {
    const size_t h = (size_t)radix_nseg(n) << RS_MAX_BITS;
    return h + scan_ws_elems(h) + 8;
}

// Sorts n (key, value) pairs by the low `key_bits` bits of the key, stable. Buffers a and b are both n elements; the
// input is in (ka, va); returns 0 if the result is in (ka, va), 1 if it is in (kb, vb).
template <class V>
inline int radix_sort_pairs(hipStream_t st, uint32_t* ka, V* va, uint32_t* kb, V* vb, int n, int key_bits, int* ws) {
    if (n <= 0 || key_bits <= 0) return 0;
    const int npass = (key_bits + RS_MAX_BITS - 1) / RS_MAX_BITS;
    const int bits = (key_bits + npass - 1) / npass;
    const int nseg = radix_nseg(n);
    const int nwg = (nseg + RS_WAVES - 1) / RS_WAVES;
    int* hist = ws;
    int* sws = ws + ((size_t)nseg << RS_MAX_BITS);
    int where = 0;
    for (int p = 0; p < npass; ++p) {
        const int shift = p * bits;
        const int b = (key_bits - shift) < bits ? (key_bits - shift) : bits;
        const uint32_t* kin = where ? kb : ka;
        const V* vin = where ? vb : va;
        uint32_t* kout = where ? ka : kb;
        V* vout = where ? va : vb;
        radix_hist<<<nwg, RS_THREADS, 0, st>>>(kin, n, shift, b, nseg, hist);
        exclusive_scan<int>(st, hist, hist, nseg << b, sws, nullptr);
        radix_scatter<V><<<nwg, RS_THREADS, 0, st>>>(kin, vin, kout, vout, n, shift, b, nseg, hist);
        where ^= 1;
    }
    return where;
}

}  // namespace prim
}  // namespace eacham
